#!/bin/bash
# rocprofv3 kernel trace of `scrappie raw` itself (N synthetic .f32 reads, all defaults): which kernels the command line spends the GPU on.
#   bash tools/cli_profile.sh [N=200000]   ->  gpurun_out/cli_prof/{kernel_stats.csv,stats.txt}
N=${1:-200000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cli_prof; rm -rf $OUT; mkdir -p $OUT
W=/tmp/cli_prof; rm -rf $W; mkdir -p $W/f32
gcc -O2 -o $W/make_reads $R/tools/make_reads.c -lm || exit 1
$W/make_reads f32 $W/f32 $N 4000
python - <<PY
import sys; sys.path.insert(0, "$R")
from scrappie_amd import model
model.save_model(model.synthetic_model("rgrgr_r94", seed=1), "$W/rgrgr_r94.scrm")
PY
$R/scrappie_amd/scrappie raw --model-file $W/rgrgr_r94.scrm --stats -o $W/out.fa $W/f32 2> $OUT/stats_untraced.txt      # warm: page cache
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $R/scrappie_amd/scrappie raw --model-file $W/rgrgr_r94.scrm --stats -o $W/out.fa $W/f32 2> $OUT/stats.txt
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv
t=$(ls $OUT/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
python - "$t" >> $OUT/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print("kernel trace: %d launches; first launch to last end %.3f s; union of kernel intervals (GPU not idle) %.3f s = %.1f %%" % (len(iv), span * 1e-9, busy * 1e-9, 100.0 * busy / span))
PY
rm -rf $OUT/trace $W
grep "scrappie stats" $OUT/stats_untraced.txt; cat $OUT/stats.txt | grep -v "No basecall"; head -12 $OUT/kernel_stats.csv | cut -c1-140
