#!/bin/bash
# `scrappie raw` on N fast5 files under two environments, interleaved (tools/cli_ab_env.sh "ENV_A" "ENV_B" [N=300000] [REPS=3]); prints the --stats wall line of each run
A="$1"; B="$2"; N=${3:-300000}; REPS=${4:-3}; NS=4000
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/cli_ab; rm -rf $W; mkdir -p $W/fast5
gcc -O2 -DWITH_HDF5 -I/opt/conda/include -o $W/make_reads5 $R/tools/make_reads.c -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -lm 2>/dev/null || { echo "no libhdf5"; exit 1; }
P=8; per=$(( (N + P - 1) / P ))
for k in $(seq 0 $((P - 1))); do hi=$(( (k + 1) * per )); [ $hi -gt $N ] && hi=$N; $W/make_reads5 fast5 $W/fast5 $hi $NS $(( k * per )) & done; wait
python - <<PY
import sys; sys.path.insert(0, "$R")
from scrappie_amd import model
model.save_model(model.synthetic_model("rgrgr_r94", seed=1), "$W/m.scrm")
PY
thr() { grep -h "nr_throttled\|^usage_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
env $A $R/scrappie_amd/scrappie raw --model-file $W/m.scrm --stats -o $W/out.fa $W/fast5 2>&1 | grep -i "wall" | head -1 | sed 's/^/[warm-up] /'
for r in $(seq $REPS); do for v in "$A" "$B"; do
  c0=$(thr)
  echo "[${v:-default}] $(env $v $R/scrappie_amd/scrappie raw --model-file $W/m.scrm --stats -o $W/out.fa $W/fast5 2>&1 | grep -i 'wall' | head -1)   cpu.stat $c0 -> $(thr)"
done; done
rm -rf $W
