#!/usr/bin/env python3
"""tests/golden/fast5/: the reference's three bundled reads (data files its own tests run on: /root/reference/reads/*.fast5,
copied byte for byte) and re-encodings of the smallest of them made with libhdf5's own h5repack, which exercise the other
storage forms the built-in reader (scrappie_amd/csrc/sh_h5mini.c) supports or must refuse:

    variant_shuf_gzip_c256.fast5   chunks of 256 samples (114 chunks: a two-level chunk B-tree), shuffle + deflate
    variant_latest.fast5           written with --latest (superblock v2, new-style groups): must be REFUSED with a message

Needs the build container (/root/reference and /opt/conda/bin/h5repack); the files are committed."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "fast5")
REPACK = "/opt/conda/bin/h5repack"

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for f in sorted(glob.glob("/root/reference/reads/*.fast5")):
        shutil.copyfile(f, os.path.join(OUT, os.path.basename(f)))
    small = "/root/reference/reads/read_ch228_file118.fast5"
    subprocess.run([REPACK, "-f", "SHUF", "-f", "GZIP=5", "-l", "CHUNK=256", small, os.path.join(OUT, "variant_shuf_gzip_c256.fast5")], check=True)
    subprocess.run([REPACK, "--latest", small, os.path.join(OUT, "variant_latest.fast5")], check=True)
    import provenance
    print(provenance.write())
