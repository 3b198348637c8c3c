#!/usr/bin/env python3
"""Codestreams of random parameter sets (tests/random_cases.py; written by the LIVE reference) cut at every byte of the main header
and the first tile-part header, at every byte around each later SOT, and at a stride through the data: the parser raises exactly
when the reference raises (with and without resilience) and otherwise reconstructs the same image.  CPU only.
    python tools/fuzz_trunc_cpu.py [seconds] [first seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openjph_amd import capi
from openjph_amd.plan import parse_codestream
from tests import cpu_pipeline as cp
from tests.random_cases import random_case
from oracle import refbind

t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
refs = {True: refbind.Ref(generic=False), False: refbind.Ref(generic=True)}
n = bad = streams = 0
while time.time() < t_end:
    planes, kw, size = random_case(seed); seed += 1
    if any(q.size == 0 for q in planes) or sum(q.size for q in planes) > 40000:
        continue
    r = refs[bool(kw["reversible"])]
    k2 = dict(kw); bd, sg = k2.pop("bit_depth"), k2.pop("is_signed")
    try:
        cs = r.encode(planes, bd, is_signed=sg, size=size, **k2)
    except RuntimeError:
        continue
    streams += 1
    sots = [i for i in range(len(cs) - 1) if cs[i] == 0xFF and cs[i + 1] == 0x90]
    cuts = set(range(2, min(len(cs), sots[0] + 20)))
    for s in sots[1:4]:
        cuts |= set(range(max(2, s - 3), min(len(cs), s + 16)))
    cuts |= set(range(sots[0], len(cs), max(1, len(cs) // 40))) | {len(cs) - 1, len(cs) - 2, len(cs) - 3}
    for k in sorted(cuts):
        for resilient in (False, True):
            try:
                want, _ = r.decode(cs[:k], resilient=resilient)
            except RuntimeError:
                want = None
            try:
                pl = parse_codestream(cs[:k], resilient=resilient)
                got = cp.inverse_stages(pl, cp.decode_blocks(pl, cs[:k]))
            except (capi.OjphError, RuntimeError):
                got = None
            n += 1
            same = (want is None) == (got is None) and (want is None or (all(np.array_equal(a, b) for a, b in zip(got, want)) if isinstance(want, list) else np.array_equal(got, want)))
            if not same:
                bad += 1
                print("DIFFERS: seed %d cut at %d of %d (SOTs at %s), resilient=%s: reference %s, here %s  %s" %
                      (seed - 1, k, len(cs), sots[:4], resilient, "raises" if want is None else "decodes", "raises" if got is None else "decodes", kw), flush=True)
print("%d cuts of %d codestreams of random parameter sets: %d handled differently from the live reference" % (n, streams, bad))
