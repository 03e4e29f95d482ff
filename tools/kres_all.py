#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in libscrappie_hip.so, from the compiler's own
-Rpass-analysis=kernel-resource-usage remarks (device-only compile, same flags as the Makefile).
usage: python tools/kres_all.py [out.csv]"""
import re
import subprocess
import sys

CSRC = "/root/repo/scrappie_amd/csrc"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-fno-slp-vectorize", "-I../../include", "-I.", "--cuda-device-only", "-S", "-o", "/dev/null",
       "scrappie_hip.hip", "-Rpass-analysis=kernel-resource-usage"]
txt = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
txt += subprocess.run([c if c != "scrappie_hip.hip" else "sh_p0.hip" for c in cmd], cwd=CSRC, capture_output=True, text=True).stderr      # (the second HIP translation unit: k_p0)
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
out.write("kernel,VGPRs,AGPRs,SGPRs,scratch_bytes_per_lane,occupancy_waves_per_SIMD,static_LDS_bytes\n")
for r in rows:
    out.write("%s,%s,%s,%s,%s,%s,%s\n" % (r["name"].replace("void ", ""), r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                        r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
