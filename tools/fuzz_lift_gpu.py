#!/usr/bin/env python3
"""Random lifting kernels through the stage-level general transform on the GPU against the oracle (the body of
tests/test_gpu_wide.py::test_general_lifting_vs_oracle with random step lists).   python tools/fuzz_lift_gpu.py [kernels] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_wide import test_general_lifting_vs_oracle as run

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n):
    ns = int(rng.choice([2, 3, 4, 6]))
    fsteps = [float(np.round(rng.uniform(-0.8, 0.45), 6)) or 0.25 for _ in range(ns)]
    K = float(np.round(rng.uniform(0.8, 1.4), 6))
    isteps = []
    for _ in range(ns):
        e = int(rng.integers(0, 5)); amax = max(1, (1 << e) // 2)
        isteps.append((int(rng.integers(-amax, amax + 1)) or 1, int(rng.integers(0, 1 << e)) if e else 0, e))
    for dt, steps, k in ((np.float32, fsteps, K), (np.int32, isteps, 1.0), (np.int64, isteps, 1.0)):
        for horz, vert in ((True, True), (True, False), (False, True)):
            try:
                run("fuzz", dt, steps, k, horz, vert)
            except AssertionError as e:
                bad += 1
                print("DIFFERS", dt.__name__, steps, k, horz, vert, str(e)[:100], flush=True)
print("%d random kernels x 3 sample types x 3 direction sets through the general transform on the GPU: %d differ from the oracle" % (n, bad))
