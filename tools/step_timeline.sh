#!/bin/bash
# kernel timeline of one steady-state step (rocprofv3 --kernel-trace), on the GPU box: bash tools/step_timeline.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/err.txt
python - $OUT <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/tr/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
short = lambda k: re.sub(r'\(.*', '', k).replace('void ', '')[:34]
idx = [i for i, r in enumerate(rows) if 'k_ff_viterbi' in r['Kernel_Name']]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'fillBuffer' in r['Kernel_Name']: continue
    print("%-36s start %8.3f  end %8.3f  dur %7.3f  q %s" % (short(r['Kernel_Name']), (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get('Queue_Id', '')))
PY
