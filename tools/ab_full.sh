#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for f in build/ab/lib_*.so; do
  v=$(basename $f .so); v=${v#lib_}
  cp $f scrappie_amd/libscrappie_hip.so
  echo "$v $(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, 'h2h', round(d['host_to_host']['ms_per_step'],2), 'hmm', round(d['hmm_posteriors']['ms_per_step'],2), {k: round(v,2) for k,v in d['hmm_posteriors']['stage_ms_per_step'].items()})")"
done; done
