#!/bin/bash
# A/B of ONE library under two environments on one GPU box: tools/ab_env.sh REPS "ENV_A" "ENV_B" [bench args]   (e.g. 3 "" "SH_FOLD_TAIL=1" --model rgrgr_r10)
cd $GRAFT_REPO_ROOT
REPS=${1:-3}; A="$2"; B="$3"; shift 3
for r in $(seq $REPS); do for v in "$A" "$B"; do
  echo "[${v:-default}] $(env $v timeout 100 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()})")"
done; done
