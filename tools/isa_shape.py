#!/usr/bin/env python3
"""Print the instruction shape of a kernel's MFMA-heavy basic blocks from hipcc -S output:
M = MFMA, v = VALU, D = LDS, G = global, w = s_waitcnt, n = s_nop, s = other scalar.
usage: isa_shape.py file.s mangled-name-prefix [min_mfma=20]"""
import sys
path, prefix = sys.argv[1], sys.argv[2]
min_m = int(sys.argv[3]) if len(sys.argv) > 3 else 20
on = False
blocks, cur, name = [], [], "entry"
for raw in open(path):
    l = raw.strip()
    if not on:
        if l.startswith(prefix) and l.split()[0].endswith(":"):
            on = True
        continue
    if not l or l.startswith(";"):
        continue
    op = l.split()[0]
    if l.startswith(".") and not op.endswith(":"):
        continue
    if op.endswith(":"):
        blocks.append((name, cur)); cur = []; name = op
        continue
    cur.append(op)
    if op == "s_endpgm":
        break
blocks.append((name, cur))
for name, ops in blocks:
    nm = sum(o.startswith("v_mfma") for o in ops)
    if nm < min_m:
        continue
    s = ""
    for o in ops:
        s += ("M" if o.startswith("v_mfma") else "v" if o.startswith("v_") else "D" if o.startswith("ds_") else
              "G" if o.startswith(("global_", "buffer_", "flat_", "scratch_")) else "w" if o.startswith("s_waitcnt") else
              "n" if o.startswith("s_nop") else "s")
    print("%s  %d instructions, %d MFMA, %d VALU" % (name, len(ops), nm, s.count("v")))
    for i in range(0, len(s), 120):
        print("   " + s[i:i + 120])
