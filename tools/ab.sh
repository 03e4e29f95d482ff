#!/bin/bash
# A/B two builds of the library on one GPU box (experiment helper): tools/ab.sh <stage_key>
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in base new; do
  cp build/ab/lib_$v.so scrappie_amd/libscrappie_hip.so
  echo "$v $(timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})")"
done; done
