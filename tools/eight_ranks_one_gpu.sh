#!/bin/bash
# bench.py as eight torchrun ranks sharing cuda:0 (gloo), 1250 reads x 4000 samples per rank: what eight ranks cost the host (cpu.stat) under the two ways of waiting
cd $GRAFT_REPO_ROOT
thr() { grep -h "nr_throttled\|^usage_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
for v in "" "SCRAPPIE_HIP_SPIN_WAIT=1" "" "SCRAPPIE_HIP_SPIN_WAIT=1"; do
  c0=$(thr); t0=$(date +%s.%N)
  out=$(env $v BENCH_BACKEND=gloo BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 8 --steps 40 --warmup 5 --reads 1250 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['ms_per_step_per_rank']; print('value %.4g  ms/step max %.2f min %.2f' % (d['value'], p['max'], p['min']))")
  echo "[${v:-query + nap}] $out   wall $(python -c "print('%.1f' % ($(date +%s.%N) - $t0))") s   cpu.stat $c0 -> $(thr)"
done
