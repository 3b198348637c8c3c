#!/usr/bin/env python3
"""tests/random_cases.py (random parameter sets: sub-sampling, precincts, tile-parts, progression orders, bit depths, qfactor,
COC segments ...) over many seeds against the LIVE reference: same bytes from the reference's encoder and the oracle pipeline,
same samples from both decoders.  CPU only.      python tools/fuzz_params_cpu.py [seconds] [first seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cpu_pipeline as cp
from tests.random_cases import random_case, random_coc_case
from oracle import refbind


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    refs = {True: refbind.Ref(generic=False), False: refbind.Ref(generic=True)}
    n = bad = 0
    from openjph_amd import capi
    refused = 0
    while time.time() < t_end:
        for gen in (random_case, random_coc_case):
            planes, kw, size = gen(seed)
            if any(q.size == 0 for q in planes):
                continue
            r = refs[bool(kw["reversible"])] if gen is random_case else refs[False]     # (COC sets mix wavelets: the generic build is the 9/7 pin)
            k2 = dict(kw); bd, sg = k2.pop("bit_depth"), k2.pop("is_signed")
            try:
                want = r.encode(planes, bd, is_signed=sg, size=size, **k2)
            except RuntimeError:
                want = None
            try:
                got, *_ = cp.encode(planes, size=size, **kw)
            except capi.OjphError:
                got = None
            if (want is None) != (got is None):
                bad += 1; print("ONE SIDE REFUSES (reference %s, here %s)" % (want is None, got is None), gen.__name__, seed, kw, flush=True); continue
            if want is None:
                refused += 1; continue
            n += 1
            if got != want:
                bad += 1; print("BYTES DIFFER", gen.__name__, seed, kw, flush=True); continue
            try:
                rdec, _ = r.decode(want)
            except RuntimeError:
                rdec = None
            try:
                dec, _ = cp.decode(want)
            except capi.OjphError:
                dec = None
            if (rdec is None) != (dec is None) or (rdec is not None and not all(np.array_equal(dec[c], rdec[c]) for c in range(len(planes)))):
                bad += 1; print("DECODE DIFFERS", gen.__name__, seed, kw, flush=True)
        seed += 1
    print("%d random parameter sets (seeds up to %d): bytes and samples of the live reference, %d differ; %d sets refused by both" % (n, seed - 1, bad, refused))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
