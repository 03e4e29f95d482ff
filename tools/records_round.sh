#!/bin/bash
# The records DESIGN.md quotes for the side models and configs, on the GPU box via gpurun:
#   bash tools/records_round.sh r3   ->  gpurun_out/rec_r3/*
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/rec_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
for m in rnnrf_r94 rgrgr_r10 raw_r94; do
  timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$m.json 2> $OUT/bench_$m.err
done
timeout 300 python bench.py --model nanonet_events --samples 800 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nanonet_events.json 2> $OUT/bench_nanonet_events.err
for m in rgrgr_r10 rnnrf_r94; do
  for n in 3000 8000 16000; do timeout 400 python tools/mixed_rate.py $n 1000 40000 $m 4 2; done > $OUT/mixed_rate_$m.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for m in rnnrf_r94 nanonet_events; do
  S=4000; [ $m = nanonet_events ] && S=800
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$m -- python $R/bench.py --model $m --samples $S --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace_$m.json 2> $OUT/trace_$m.err
  f=$(ls $OUT/trace_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$m.csv
  rm -rf $OUT/trace_$m
done
cd $R
python - "$OUT" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), "%.2f ms/step" % d["ms_per_step"], "%.3e %s" % (d["value"], d["unit"]), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "roofline %s frac %.3f" % (d["roofline"]["kernel"][:12], d["roofline"]["frac"]))
    except Exception as e:
        print(f, "FAILED", e)
for f in sorted(glob.glob(out + "/mixed_rate_*.txt")):
    print(open(f).read().strip())
PY
