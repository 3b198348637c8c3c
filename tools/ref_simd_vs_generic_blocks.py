#!/usr/bin/env python3
"""The reference against itself: its generic C++ and its AVX2 32-bit block decoders on the damaged blocks of tools/fuzz_blocks_cpu.py.
They agree on blocks of even size and differ on odd-sized ones (the parity pin for damaged input is the generic decoder, DESIGN.md
section 2.1).  CPU only, needs oracle/_ref.      python tools/ref_simd_vs_generic_blocks.py [seeds]"""
import os
import sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import refbind
from fuzz_blocks_cpu import damaged_blocks
r = refbind.Ref(generic=False); r.lib.ref_set_verbose(0)
print("simd level", r.simd_level())
n = dv = da = both_ok = 0
for seed in range(730000, 730000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 300)):
    for (wide, w, h, kmax, mmsb, data, len2, passes, causal) in damaged_blocks(seed):
        if wide or passes > 3: continue
        st = (w + 15) & ~15
        ok0, d0 = r.decode_block(data, mmsb, w, h, st, len2=len2, num_passes=passes, variant=0, stripe_causal=causal)
        ok1, d1 = r.decode_block(data, mmsb, w, h, st, len2=len2, num_passes=passes, variant=1, stripe_causal=causal)
        n += 1
        if ok0 != ok1: dv += 1
        elif ok0:
            both_ok += 1
            if not np.array_equal(d0[:, :w], d1[:, :w]): da += 1
print(n, "damaged blocks: verdicts differ on", dv, "; of", both_ok, "decoded by both,", da, "decoded differently")
