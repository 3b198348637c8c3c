#!/usr/bin/env python3
"""Builds openjph_amd/csrc/_build/lib_<name>.so: the product library with ONE source compiled with extra flags
(A/B experiments inside one GPU-box visit, tools/ab_variants.sh, tools/enc_only.py).
    python tools/build_variant.py <name> <source file in csrc> [flags ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openjph_amd import build as b

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
bdir = os.path.join(b.CSRC, "_build")
obj = os.path.join(bdir, "%s.%s.o" % (src, name))
subprocess.check_call([b.HIPCC, "-x", "hip"] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(bdir, s + ".o") for s in b.SOURCES]
out = os.path.join(ROOT, "openjph_amd", "variants", "lib_%s.so" % name); os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + objs)
print(out)
