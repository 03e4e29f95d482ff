#!/bin/bash
# A/B library builds on one GPU box (experiment helper): every build/ab/lib_<name>.so is timed,
# interleaved, REPS times (default 2).
cd $GRAFT_REPO_ROOT
REPS=${1:-2}
for r in $(seq $REPS); do for f in build/ab/lib_*.so; do
  v=$(basename $f .so); v=${v#lib_}
  cp $f scrappie_amd/libscrappie_hip.so
  echo "$v $(env $ABENV timeout 100 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})")"
done; done
