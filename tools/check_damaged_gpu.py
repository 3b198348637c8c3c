#!/usr/bin/env python3
"""The HIP decoder on the damaged codestreams of tests/damaged_cases.py against the live reference's committed verdicts
(tests/golden/damaged.json): "raises" or the digest of the picture, with and without resilience.  Needs a GPU
(tests/test_gpu_damaged.py is the same comparison as -m gpu tests).      python tools/check_damaged_gpu.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openjph_amd import capi, codec
from tests.damaged_cases import cases, digest

gold = json.load(open(os.path.join(ROOT, "tests", "golden", "damaged.json")))["cases"]
n = bad = 0
for name, part in cases():
    for resilient in (False, True):
        key = "%s_%d" % (name, int(resilient))
        if key not in gold:
            continue
        try:
            dec = codec.Decoder(part, resilient=resilient)
            frame = np.asarray(dec.decode())
            got = digest(frame if len(dec.plan.frame_shape) == 3 else dec.plan.unpack_frame(frame))   # (a damaged SIZ may give the components different sizes)
        except (capi.OjphError, RuntimeError):
            got = "raises"
        n += 1
        if got != gold[key]:
            bad += 1
            print("DIFFERS: %s: reference %s, HIP decoder %s" % (key, gold[key][:16], got[:16]), flush=True)
print("%d damaged codestreams through the HIP decoder: %d differ from the reference's committed verdicts" % (n, bad))
sys.exit(1 if bad else 0)
