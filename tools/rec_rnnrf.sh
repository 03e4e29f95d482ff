R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s7/rec; mkdir -p $OUT; cd $R
(timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $OUT/gputest.log
timeout 300 python bench.py --model rnnrf_r94 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rnnrf_r94.json 2> $OUT/bench_rnnrf_r94.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --model rnnrf_r94 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace.json 2> $OUT/trace.err
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_rnnrf_r94.csv; rm -rf $OUT/trace
cd $R; cat $OUT/gputest.log; python -c "
import json; d=json.load(open('$OUT/bench_rnnrf_r94.json')); print(d['ms_per_step'], d['value'], d['stage_ms_per_step'], d['roofline'])"
head -4 $OUT/kernel_stats_rnnrf_r94.csv | cut -c1-220
