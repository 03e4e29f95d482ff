#!/bin/bash
# Collect the profiles committed under profiles/: kernel-trace stats + PMC passes of one bench step.
# Usage (on the GPU box, via gpurun): bash tools/profile_round.sh <tag>
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_summary.csv,bench.json}
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
TRACE="python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra"   # kernel-trace pass: enough launches (225 of the recurrent layer) that the cold ones after process start do not move the average
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace_bench.json 2> $OUT/trace.err
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$tag -- $BENCH > /dev/null 2> $OUT/pmc_$tag.err
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, re
out = sys.argv[1]
def short(k): return re.sub(r'\(.*', '', k).replace('void ', '')[:48]
# kernel stats (rocprofv3 --stats) + quartiles from the kernel trace of the same run
rows = []
for f in glob.glob(out + '/trace/*/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(out + '/kernel_stats.csv', 'w') as fh:
    if rows:
        fh.write('# rocprofv3 --kernel-trace --stats of bench.py --steps 40 --warmup 5; p25 / median / p75 from the kernel trace of the same run.\n')
        fh.write('# The mean launch duration of k_gru_proj is what bench.py measures with HIP events in an untraced run (roofline.avg_launch_ms); one launch\n')
        fh.write('# in five -- the first layer of a group, beside the previous group\'s traceback walk and k_stitch and the next group\'s convolution -- is\n')
        fh.write('# slower than the median.  Results and metadata move by kernels (k_results_out, k_upload_words), not by the copy engine.\n')
        w = csv.writer(fh)
        w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'pct', 'min_ns', 'max_ns', 'p25_ns', 'median_ns', 'p75_ns'])
        for r in rows:
            v = sorted(dur.get(short(r['Name']), []))
            q = [v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4]] if v else ['', '', '']
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']] + q)
# pmc
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + '/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (short(r['Kernel_Name']), r['Counter_Name'])
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
with open(out + '/pmc_summary.csv', 'w') as fh:
    fh.write('# rocprofv3 --pmc, one pass per counter group, bench.py --steps 2 --warmup 1 (10000 reads x 4000 samples); per-dispatch mean\n')
    fh.write('# FETCH_SIZE / WRITE_SIZE in KiB as reported; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts wide coalesced reads at 1/2 of their bytes\n')
    w = csv.writer(fh)
    fh.write('# registers / LDS per kernel: profiles/*_kernel_resources.csv (compiler remarks; rocprofv3 metadata columns on gfx950 are unreliable and omitted)\n')
    w.writerow(['kernel', 'counter', 'dispatches', 'mean_value'])
    for (k, c), (n, v) in sorted(acc.items()):
        w.writerow([k, c, n, '%.6g' % (v / n)])
print(open(out + '/kernel_stats.csv').read())
PY
# the traffic record of THIS tree (the PMC passes above), so that the bench line below carries roofline.traffic: the same file is committed as profiles/<tag>_traffic.json
cd $R && python tools/make_traffic_json.py $OUT/pmc_summary.csv profiles/${TAG}_traffic.json > /dev/null && cp profiles/${TAG}_traffic.json $OUT/traffic.json
cd $R && timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err    # the defaults (40 steps after 10 of warm-up)
tail -c 2500 $OUT/bench.json
