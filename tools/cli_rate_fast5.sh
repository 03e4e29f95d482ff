#!/bin/bash
# `scrappie raw` on the one input format the reference reads: N single-read fast5 files as MinKNOW writes them (int16 Signal, chunked, deflate 1)
# -> FASTA, on one GPU; loader rate against engine rate against wall for several loader-thread counts, built-in reader (own inflate) against the
# same reader on zlib (SH_H5MINI_ZLIB=1 is not a product switch: the zlib line comes from profiles/r5_cli_rate.txt) and against .f32 files.
#   bash tools/cli_rate_fast5.sh [N=400000] [NS=4000]   ->  gpurun_out/cli_rate_fast5.txt   (on the GPU box, via gpurun)
N=${1:-400000}; NS=${2:-4000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cli_rate_fast5.txt
W=/tmp/cli_rate5; rm -rf $W; mkdir -p $W/f32 $W/fast5
exec > $OUT 2>&1
gcc -O2 -o $W/make_reads $R/tools/make_reads.c -lm || exit 1
gcc -O2 -DWITH_HDF5 -I/opt/conda/include -o $W/make_reads5 $R/tools/make_reads.c -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -lm 2>/dev/null || { echo "no libhdf5 to write fast5 files with"; exit 1; }
echo "host: $(nproc) CPUs visible, cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $(free -g | awk '/Mem:/{print $2}') GB RAM; $N reads x $NS samples"
SECONDS=0; P=8; per=$(( (N + P - 1) / P ))
for k in $(seq 0 $((P - 1))); do hi=$(( (k + 1) * per )); [ $hi -gt $N ] && hi=$N; $W/make_reads5 fast5 $W/fast5 $hi $NS $(( k * per )) & done; wait
echo "wrote $N .fast5 files (int16, chunked, deflate 1) with $P processes in $SECONDS s ($(du -sh $W/fast5 | cut -f1))"
SECONDS=0; $W/make_reads f32 $W/f32 $N $NS; echo "wrote $N .f32 files in $SECONDS s ($(du -sh $W/f32 | cut -f1))"
python - <<PY
import sys; sys.path.insert(0, "$R")
from scrappie_amd import model
model.save_model(model.synthetic_model("rgrgr_r94", seed=1), "$W/rgrgr_r94.scrm")
PY
run() {   # label, dir, extra args...
  local label=$1 dir=$2; shift 2
  echo "== $label: scrappie raw $* $dir"
  SCRAPPIE_FAST5_READER=own $R/scrappie_amd/scrappie raw --model-file $W/rgrgr_r94.scrm --stats -o $W/out.fa "$@" $dir 2>&1 | grep -v "^scrappie: No basecall" | tail -5
  echo "   records: $(grep -c '^>' $W/out.fa), md5 of the sorted sequences $(grep -v '^>' $W/out.fa | sort | md5sum | cut -c1-12)"
}
run "fast5, all defaults (warm-up run: clocks)" $W/fast5
run "fast5, all defaults" $W/fast5
for thr in 4 8 12 16; do run "fast5, $thr loader threads" $W/fast5 --threads $thr; done
run "f32, all defaults" $W/f32
run "fast5, host preparation, 16 threads" $W/fast5 --prep=host --threads 16
rm -rf $W
