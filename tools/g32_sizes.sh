#!/bin/bash
# recurrent-layer forms at several launch-group sizes (experiments build): tools/g32_sizes.sh "10000 16384 32768"
cd $GRAFT_REPO_ROOT
export SCRAPPIE_HIP_LIB=$PWD/scrappie_amd/libscrappie_hip_exp.so
for n in ${1:-10000 16384}; do for f in 0 1 2; do
  if [ $f = 0 ]; then env="SH_GRU16=1"; else env="SH_GRU32=$f"; fi
  env $env timeout 300 python bench.py --reads $n --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reads $n form $f: %.2f ms per step, layers %.2f, decoder %.2f -> %.3e samples/s' % (d['ms_per_step'], d['stage_ms_per_step']['gru_ms'], d['stage_ms_per_step']['decode_ms'], d['value']))"
done; done
