#!/usr/bin/env python3
"""Codestreams of random parameter sets (written by the LIVE reference) with one to three bytes changed behind the first SOD -- packet
headers, code-block bytes, later SOT segments -- read with and without resilience: the reference's verdict (raise / decode) and its
image are the oracle pipeline's.  CPU only.      python tools/fuzz_flip_cpu.py [seconds] [first seed]"""
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
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
rng = np.random.default_rng(seed)
refs = {True: refbind.Ref(generic=True), False: refbind.Ref(generic=True)}   # (the generic C++ block decoder: the AVX2 one decodes DAMAGED blocks differently from it)
n = bad = streams = raised = 0
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
    sod = cs.find(b"\xff\x93")
    if sod < 0 or len(cs) - sod < 8:
        continue
    streams += 1
    for trial in range(40):
        b = bytearray(cs)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(sod + 2, len(b)))] = int(rng.choice([0xFF, 0x00, 0x90, 0x7F, int(rng.integers(0, 256))]))
        part = bytes(b)
        for resilient in (False, True):
            try:
                want, _ = r.decode(part, resilient=resilient)
            except RuntimeError:
                want = None
            try:
                pl = parse_codestream(part, resilient=resilient)
                got = cp.inverse_stages(pl, cp.decode_blocks(pl, part, resilient=resilient) if "resilient" in cp.decode_blocks.__code__.co_varnames else cp.decode_blocks(pl, part))
            except (capi.OjphError, RuntimeError):
                got = None
            n += 1; raised += want is None
            same = (want is None) == (got is None) and (want is None or (all(np.array_equal(a, c) for a, c in zip(got, want)) if isinstance(want, list) else np.array_equal(got, want)))
            if not same:
                bad += 1
                diff = [i for i in range(len(cs)) if cs[i] != part[i]]
                print("DIFFERS: seed %d bytes changed at %s of %d (SOD at %d), resilient=%s: reference %s, here %s  %s" %
                      (seed - 1, diff, len(cs), sod, resilient, "raises" if want is None else "decodes", "raises" if got is None else "decodes", kw), flush=True)
print("%d damaged codestreams (%d sources; the reference raised on %d): %d handled differently from the live reference" % (n, streams, raised, bad))
