#!/usr/bin/env python3
"""Part-2 codestreams (ATK / DFS marker segments; written here) cut at every byte of the main header and the first tile-part header
and at many points of the data: the parser raises exactly when the LIVE reference raises (with and without resilience) and
otherwise reconstructs the same image.  CPU only.   python tools/fuzz_trunc_part2_cpu.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openjph_amd import capi
from openjph_amd.plan import parse_codestream
from tests import cpu_pipeline as cp
from tests.part2_cases import CASES, split, image
from oracle import refbind

refs = {True: refbind.Ref(generic=False), False: refbind.Ref(generic=True)}
n = bad = 0
rng = np.random.default_rng(4)
for case in CASES:
    nc, h, w, bd, kw = split(case)
    if bd > 16:
        continue
    cs, plan, *_ = cp.encode(image(nc, h, w, bd), **kw)
    rev_all = all(plan.comp_style(i)["reversible"] for i in range(nc))
    r = refs[rev_all]
    sot = cs.find(b"\xff\x90")
    trials = [("cut", k, cs[:k]) for k in list(range(2, sot + 14)) + [len(cs) * c // 12 for c in range(1, 12)] + [len(cs) - 1, len(cs) - 2]]
    for kind, k, part in trials:
        for resilient in (False, True):
            try:
                want, _ = r.decode(part, resilient=resilient)
            except RuntimeError:
                want = None
            try:
                pl = parse_codestream(part, resilient=resilient)
                got = cp.inverse_stages(pl, cp.decode_blocks(pl, part))
            except (capi.OjphError, RuntimeError):
                got = None
            n += 1
            same = (want is None) == (got is None) and (want is None or (all(np.array_equal(a, b) for a, b in zip(got, want)) if isinstance(want, list) else np.array_equal(got, want)))
            if not same:
                bad += 1
                print("DIFFERS: %s at %d of %d (main header %d), resilient=%s: reference %s, here %s   %s" %
                      (kind, k, len(cs), sot, resilient, "raises" if want is None else "decodes", "raises" if got is None else "decodes", {a: b for a, b in kw.items() if a not in ("atk",)}), flush=True)
print("%d damaged Part-2 codestreams: %d handled differently from the live reference" % (n, bad))
