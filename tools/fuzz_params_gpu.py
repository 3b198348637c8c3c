#!/usr/bin/env python3
"""Seeded random parameter sets (tests/random_cases.py: odd sizes, offsets, sub-sampling, mixed formats, every progression
order, tiles, tile-parts, COC segments) through the HIP codec against the oracle pipeline, byte for byte and sample for
sample -- the GPU suite runs seeds 0..119 of this; here as many as the time allows.
    python tools/fuzz_params_gpu.py [seconds] [first seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from openjph_amd import capi, codec
from openjph_amd.plan import make_params
from tests import cpu_pipeline as cp
from tests.random_cases import random_case, random_coc_case

t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
done = skipped = differ = 0
while time.time() < t_end:
    for maker in (random_case, random_coc_case):
        planes, kw, size = maker(seed)
        if any(q.size == 0 for q in planes):
            skipped += 1
            continue
        try:
            want, plan, *_ = cp.encode(planes, size=size, **kw)
        except capi.OjphError:
            skipped += 1
            continue
        got = codec.Encoder(make_params(size[0], size[1], len(planes), **kw)).encode(planes)
        try:
            wdec, _ = cp.decode(want)
        except capi.OjphError:                            # tile-part numbers with gaps: read resiliently only (the reference's parser, too)
            ok = got == want
        else:
            dec = codec.Decoder(want)
            out = dec.plan.unpack_frame(dec.decode())
            ok = got == want and all(np.array_equal(out[c], wdec[c]) for c in range(len(planes)))
        if not ok:
            differ += 1
            print("DIFFERS: %s seed %d %s" % (maker.__name__, seed, kw), flush=True)
        done += 1
    seed += 1
print("%d random parameter sets through the HIP codec (seeds up to %d; %d rejected by the parameter checks): %d differ from the oracle pipeline"
      % (done, seed - 1, skipped, differ))
