"""Wider randomized parity sweep than the test-suite runs: GPU codec against the oracle pipeline
(bytes and samples) over seeded random parameter sets (tests/random_cases.py)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from openjph_amd import capi, codec
from openjph_amd.plan import make_params
from tests import cpu_pipeline as cp
from tests.random_cases import random_case, random_coc_case


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    done = skipped = 0
    for gen in (random_case, random_coc_case):
        for seed in range(lo, hi):
            planes, kw, size = gen(seed)
            if any(q.size == 0 for q in planes):
                continue
            try:
                want, plan, *_ = cp.encode(planes, size=size, **kw)
            except capi.OjphError:
                skipped += 1
                continue
            got = codec.Encoder(make_params(size[0], size[1], len(planes), **kw)).encode(planes)
            assert got == want, "%s seed %d: %s" % (gen.__name__, seed, kw)
            try:
                wdec, _ = cp.decode(want)
            except capi.OjphError:
                skipped += 1
                continue
            dec = codec.Decoder(want)
            out = dec.plan.unpack_frame(dec.decode())
            for c in range(len(planes)):
                assert np.array_equal(out[c], wdec[c]), "%s seed %d component %d: %s" % (gen.__name__, seed, c, kw)
            done += 1
    print("sweep %d..%d: %d parameter sets identical, %d refused by both, %.0f s" % (lo, hi, done, skipped, time.time() - t0))


if __name__ == "__main__":
    main()
