# long-tailed lengths: device-resident streams of calls of 12 000 reads, lognormal(20 000, 0.8) clipped to [1000, hi] (profiles/r5_long_tail.txt)
export MIXED_LOGNORMAL=20000,0.8 MIXED_H2H=0 MIXED_ENGINES=2
for K in 12 24 48; do MIXED_DSTREAM=$K python tools/mixed_rate.py 12000 1000 400000 rgrgr_r10 2>&1 | grep -v amdgpu.ids; done
MIXED_DSTREAM=24 python tools/mixed_rate.py 12000 1000 100000 rgrgr_r10 2>&1 | grep -v amdgpu.ids
unset MIXED_LOGNORMAL
MIXED_DSTREAM=12 python tools/mixed_rate.py 16000 1000 40000 rgrgr_r10 2>&1 | grep -v amdgpu.ids
