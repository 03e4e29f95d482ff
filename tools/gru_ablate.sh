for d in 0 1 2 4 8 3 7 15; do
  echo "dbg=$d $(SH_GRU12_DEBUG=$d timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stage_ms_per_step']['gru_ms'])")"
done
