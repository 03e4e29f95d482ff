#!/bin/bash
# k_gru_proj32 (experiments build, SH_GRU32=1): kernel-trace stats, two PMC passes and cycle stamps of the bench step -> gpurun_out/prof_gru32/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_gru32
rm -rf $OUT; mkdir -p $OUT
export SCRAPPIE_HIP_LIB=$R/scrappie_amd/libscrappie_hip_exp.so SH_GRU32=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_trace.json 2> $OUT/trace.err
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_$tag.err
done
cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench.json 2>/dev/null
SH_GRU32_STAMP=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "stamp wave" > $OUT/stamps.txt
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re, json
out = sys.argv[1]
def short(k): return re.sub(r'\(.*', '', k).replace('void ', '')[:48]
with open(out + '/summary.txt', 'w') as fh:
    for f in glob.glob(out + '/trace/*/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_gru_proj32' in r['Name'] or 'k_ff_viterbi' in r['Name']:
                fh.write('kernel-trace stats: %s calls %s avg %.1f us min %.1f max %.1f\n' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(out + '/pmc_*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_gru_proj32' in r['Kernel_Name']:
                a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for c, (n, v) in sorted(acc.items()):
        fh.write('pmc k_gru_proj32 %-28s %.6g (mean of %d dispatches)\n' % (c, v / n, n))
    d = json.load(open(out + '/bench.json'))
    fh.write('bench --steps 20 --warmup 5: %.2f ms per step, stages %s\n' % (d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms_per_step'].items()}))
    fh.write(open(out + '/stamps.txt').read())
print(open(out + '/summary.txt').read())
PY
