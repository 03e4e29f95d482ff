#!/bin/bash
# diagnostic: tools/batch64_omp.c with BATCH64_TRACE=1 (when each of the 156 callers went in and came out), a few processes in a row
cd $GRAFT_REPO_ROOT
T=$(mktemp -d)
python - "$T" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from scrappie_amd import model, synth
t = sys.argv[1]
model.save_model(model.synthetic_model("rgrgr_r94", seed=1), t + "/m.scrm")
np.stack([synth.medmad_normalise(synth.synthetic_signal(4000, 100 + i)) for i in range(64)] * 156).astype(np.float32).tofile(t + "/s.f32")
PY
gcc -O2 -std=gnu11 -fopenmp -Iinclude tools/batch64_omp.c -o $T/b64 -Lscrappie_amd -l${LIB:-scrappie_hip} -Wl,-rpath,$PWD/scrappie_amd -lm || exit 1
thr() { grep -h "nr_throttled\|throttled_usec\|^usage_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo; }
for i in 1 2 3; do
  echo "cpu.stat before: $(thr)"
  BATCH64_TRACE=${TRACE:-1} OMP_WAIT_POLICY=passive GOMP_SPINCOUNT=0 $T/b64 $T/m.scrm $T/s.f32 9984 4000 ${NTHR:-64} ${REPS:-2} 2>&1
  echo "cpu.stat after:  $(thr)"
done
rm -rf $T
