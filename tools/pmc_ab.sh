#!/bin/bash
# PMC counters of one kernel for every build/ab/lib_<name>.so: tools/pmc_ab.sh <kernel-substring> "<counters>"   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-k_ff_viterbi}
C=${2:-"SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"}
cd /tmp && export TMPDIR=/tmp
for f in $R/build/ab/lib_*.so; do
  v=$(basename $f .so); v=${v#lib_}
  cp $f $R/scrappie_amd/libscrappie_hip.so
  rm -rf /tmp/pmc_$v
  env $ABENV timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$v -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
  python - "$v" "$K" /tmp/pmc_$v <<'PY'
import csv, glob, sys, collections
v, k, d = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
print(v, k, {c: '%.4g' % (s / n) for c, (n, s) in sorted(acc.items())})
PY
done
