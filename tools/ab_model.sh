#!/bin/bash
# build/ab/lib_<name>.so timed on one model, interleaved: tools/ab_model.sh <model> [reps]   (wall ms per step, device total, their gap)
cd $GRAFT_REPO_ROOT
M=${1:-rnnrf_r94}; REPS=${2:-2}
for r in $(seq $REPS); do for f in build/ab/lib_*.so; do
  v=$(basename $f .so); v=${v#lib_}
  cp $f scrappie_amd/libscrappie_hip.so
  echo "$v $M $(timeout 200 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms per step, device %.2f, gap %.2f' % (d['ms_per_step'], d['stage_ms_per_step']['total_ms'], d['ms_per_step'] - d['stage_ms_per_step']['total_ms']))")"
done; done
