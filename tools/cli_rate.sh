#!/bin/bash
# `scrappie raw` end to end on one GPU: N synthetic 4000-sample reads as files -> FASTA, loader rate
# against engine rate against wall, for --prep=host / device and several host thread counts.
#   bash tools/cli_rate.sh [N=200000] [NS=4000]   ->  gpurun_out/cli_rate.txt   (on the GPU box, via gpurun)
N=${1:-200000}; NS=${2:-4000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cli_rate.txt
W=/tmp/cli_rate; rm -rf $W; mkdir -p $W/f32 $W/fast5
exec > $OUT 2>&1
gcc -O2 -o $W/make_reads $R/tools/make_reads.c -lm || exit 1
H5=0; gcc -O2 -DWITH_HDF5 -I/opt/conda/include -o $W/make_reads5 $R/tools/make_reads.c -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -lm 2>/dev/null && H5=1
echo "host: $(nproc) CPUs visible, cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $(free -g | awk '/Mem:/{print $2}') GB RAM; $N reads x $NS samples"
SECONDS=0; $W/make_reads f32 $W/f32 $N $NS; echo "wrote $N .f32 files in $SECONDS s ($(du -sh $W/f32 | cut -f1))"
N5=$((N / 4))
if [ $H5 = 1 ]; then SECONDS=0; $W/make_reads5 fast5 $W/fast5 $N5 $NS; echo "wrote $N5 .fast5 files (int16, chunked, deflate 1) in $SECONDS s ($(du -sh $W/fast5 | cut -f1))"; fi
python - <<PY
import sys; sys.path.insert(0, "$R")
from scrappie_amd import model
model.save_model(model.synthetic_model("rgrgr_r94", seed=1), "$W/rgrgr_r94.scrm")
PY
run() {   # label, dir, extra args...
  local label=$1 dir=$2; shift 2
  echo "== $label: scrappie raw $* $dir"
  $R/scrappie_amd/scrappie raw --model-file $W/rgrgr_r94.scrm --stats -o $W/out.fa "$@" $dir 2>&1 | grep -v "^scrappie: No basecall" | tail -5
  echo "   records: $(grep -c '^>' $W/out.fa), md5 $(md5sum < $W/out.fa | cut -c1-12)"
}
# (the files were written a moment ago: they are in the page cache -- what is measured is the software, not the disk)
for thr in 2 8 16; do
  run "f32 host prep, $thr threads" $W/f32 --prep=host --threads $thr --batch 16384
  run "f32 device prep, $thr threads" $W/f32 --prep=device --threads $thr --batch 16384
done
run "f32, all defaults (device prep, batch 16384, loader threads = CPUs of the process up to 16)" $W/f32
run "f32 device prep, 4 threads" $W/f32 --prep=device --threads 4
run "f32 device prep, 8 threads, batch 65536" $W/f32 --prep=device --threads 8 --batch 65536
if [ -n "$BIG" ]; then       # a longer run: start-up (the ramp of batch sizes, the first and the last batch) is paid once
  mkdir -p $W/big; $W/make_reads f32 $W/big $BIG $NS
  run "f32 device prep, 8 threads, $BIG reads" $W/big --prep=device --threads 8
  run "f32, all defaults, $BIG reads" $W/big
  rm -rf $W/big
fi
if [ $H5 = 1 ]; then
  for thr in 8 16; do
    SCRAPPIE_FAST5_READER=hdf5 run "fast5 (libhdf5) device prep, $thr threads" $W/fast5 --prep=device --threads $thr --batch 16384
    SCRAPPIE_FAST5_READER=own run "fast5 (built-in reader) device prep, $thr threads" $W/fast5 --prep=device --threads $thr --batch 16384
  done
  SCRAPPIE_FAST5_READER=own run "fast5 (built-in reader) host prep, 16 threads" $W/fast5 --prep=host --threads 16 --batch 16384
fi
rm -rf $W
