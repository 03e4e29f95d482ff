"""How much host CPU the engine's waiting costs: process CPU time over wall time while one thread drives device-resident launch groups the way bench.py does
(tools/cpu_use_probe.py [steps])."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scrappie_amd as sa
from scrappie_amd import model, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e = sa.Engine(0)
e.load_model("m", model.synthetic_model("rgrgr_r94", seed=1))
n, ns = 10000, 4000
sig = np.stack([synth.medmad_normalise(synth.synthetic_signal(ns, 100 + i)) for i in range(64)] * (n // 64 + 1))[:n].astype(np.float32)
d = e.upload(sig.reshape(-1))
off = np.arange(n, dtype=np.uint64) * np.uint64(ns)
ln = np.full(n, ns, np.uint32)
p = e.default_params()
e.set_max_launch_reads(16384)
for _ in range(3):
    e.run_device(d, off, ln, "m", p); e.collect(n, p, raw=True)
e.synchronize()
def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            v = f[f.rindex(")") + 2:].split()
            out[t] = (comm, (int(v[11]) + int(v[12])) / os.sysconf("SC_CLK_TCK"))
        except OSError:
            pass
    return out
th0 = threads()
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
e.run_device(d, off, ln, "m", p)
for _ in range(steps - 1):
    e.run_device(d, off, ln, "m", p)
    e.collect(n, p, raw=True)
e.collect(n, p, raw=True)
e.synchronize()
t1 = time.perf_counter(); r1 = resource.getrusage(resource.RUSAGE_SELF)
th1 = threads()
for t, (comm, c) in sorted(th1.items(), key=lambda kv: -(kv[1][1] - th0.get(kv[0], ("", 0))[1]))[:6]:
    print("  thread %s (%s%s): %.3f s" % (t, comm, ", main" if int(t) == os.getpid() else "", c - th0.get(t, ("", 0))[1]))
cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
print("wall %.3f s, process CPU %.3f s (user %.3f, sys %.3f) = %.2f CPUs; %.2f ms per step" % (t1 - t0, cpu, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, cpu / (t1 - t0), 1e3 * (t1 - t0) / steps))
