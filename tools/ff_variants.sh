#!/bin/bash
# time S1 kernel variants (experiment)
cd $GRAFT_REPO_ROOT
for v in 3_512 2_512 2_768 3_768 4_512; do
  cp build/variants/lib_$v.so scrappie_amd/libscrappie_hip.so
  echo "V=$v $(timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step']['ff_ms'])")"
done
