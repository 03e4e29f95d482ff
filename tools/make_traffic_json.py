#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.csv -> profiles/<round>_traffic.json (HBM bytes per launch per kernel:
FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KiB -> bytes)."""
import csv
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "profiles/r2_pmc_summary.csv"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r2_traffic.json"
rows = [r for r in csv.reader(open(src)) if r and not r[0].startswith("#")]
d = {}
for r in rows[1:]:
    d.setdefault(r[0], {})[r[1]] = float(r[3])
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402  (csrc_tree_hash: the kernel sources this measurement is valid for)
out = {"csrc_sha256": bench.csrc_tree_hash(),
       "source": src + " (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1)",
       "workload": {"model": "rgrgr_r94", "reads": 10000, "samples": 4000},
       "correction": "FETCH_SIZE doubled (gfx950 counts 16 B/lane coalesced reads at half their bytes); WRITE_SIZE as reported; KiB -> bytes",
       "bytes_per_launch": {}}
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k.startswith("k_"):
        key = k.split("<")[0]
        out["bytes_per_launch"][key] = {"kernel": k, "fetch": v["FETCH_SIZE"] * 2 * 1024, "write": v["WRITE_SIZE"] * 1024,
                                        "total": (v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1024}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["bytes_per_launch"].items():
    print("%-22s %-40s %.3f GB" % (k, v["kernel"], v["total"] / 1e9))
