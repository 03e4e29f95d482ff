#!/bin/bash
# like tools/ab.sh, plus k_gru_proj32's cycle stamps of every variant: build/ab/lib_<name>.so timed once each, then stamped
cd $GRAFT_REPO_ROOT
for f in build/ab/lib_*.so; do
  v=$(basename $f .so); v=${v#lib_}
  cp $f scrappie_amd/libscrappie_hip.so
  echo "== $v $(env $ABENV timeout 100 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k in ('gru_ms','decode_ms','conv_ms')})")"
  env $ABENV SH_GRU32_STAMP=1 timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "stamp wave" | sed 's/gru32 stamp wave //; s/cycles per step (979 steps)//'
done
