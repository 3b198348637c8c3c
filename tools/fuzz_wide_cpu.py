#!/usr/bin/env python3
"""Random frames of 25..32-bit samples (the 64-bit sample path and its border with the 32-bit one) through the oracle pipeline and
the LIVE reference: the reference's encoder must write the very bytes the oracle pipeline writes, both decode them to the
same samples, reversible frames come back exactly.  CPU only.     python tools/fuzz_wide_cpu.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cpu_pipeline as cp
from tests.test_gpu_wide import deep_image
from oracle import refbind


def main(seconds=None, seed=None):
    t_end = time.time() + (seconds if seconds is not None else float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    rng = np.random.default_rng(seed if seed is not None else int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ref = refbind.Ref(generic=False)
    n = bad = refused = ref_lossy = 0
    while time.time() < t_end:
        nc = int(rng.choice([1, 1, 2, 3, 3, 4]))
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 170))
        bd = int(rng.integers(25, 33))
        signed = bool(rng.random() < 0.4)
        kw = dict(bit_depth=bd, is_signed=signed, num_decomps=int(rng.integers(0, 6)))
        if nc >= 3 and rng.random() < 0.6: kw["color_transform"] = True
        if rng.random() < 0.4: kw["tile"] = (int(rng.integers(20, 120)), int(rng.integers(20, 120)))
        if rng.random() < 0.4: kw["block"] = [(64, 64), (32, 32), (128, 32), (16, 16), (4, 64), (1024, 4)][int(rng.integers(0, 6))]
        if rng.random() < 0.3: kw["prog_order"] = ["LRCP", "RLCP", "RPCL", "PCRL", "CPRL"][int(rng.integers(0, 5))]
        if rng.random() < 0.2: kw["image_offset"] = (int(rng.integers(0, 7)), int(rng.integers(0, 7)))
        if rng.random() < 0.2: kw["tlm"] = True
        img = deep_image(nc, h, w, bd, signed, seed=int(rng.integers(0, 1000)))
        if rng.random() < 0.3:                               # sparse / small-valued frames: blocks with many missing MSBs, empty blocks
            img = (img >> int(rng.integers(8, bd))).astype(img.dtype)
        rkw = dict(kw); rkw.pop("bit_depth")
        try:
            want = ref.encode(img, bd, **rkw)
        except Exception as e:
            refused += 1
            try:
                cp.encode(img, **kw); bad += 1; print("ONLY THE REFERENCE REFUSES", str(e)[:80], nc, h, w, kw, flush=True)
            except Exception:
                pass
            continue
        n += 1
        try:
            got, plan, *_ = cp.encode(img, **kw)
            ok = got == want
            if ok:
                dec, _ = cp.decode(want); rdec, _ = ref.decode(want)
                ok = np.array_equal(dec, rdec)
                # (lossless where the reference is: 32-bit unsigned samples of a component WITHOUT decompositions come back from the
                #  reference's own decoder 2^31 off where they exceed 2^31 - 1 -- its 32-bit line conversion -- and from here alike)
                if not np.array_equal(np.asarray(rdec).reshape(-1).astype(np.int64), np.asarray(img).reshape(-1).astype(np.int64)):
                    ref_lossy += 1
            if not ok:
                bad += 1; print("MISMATCH", len(got), len(want), nc, h, w, kw, flush=True)
        except Exception as e:
            bad += 1; print("ERROR %s: %s" % (type(e).__name__, str(e)[:160]), nc, h, w, kw, flush=True)
    print("%d random frames of 25..32-bit samples: codestream byte-identical to the live reference's, decoded alike; %d differ; "
          "%d configurations the reference refuses; %d frames the reference itself does not return exactly (and neither does the oracle pipeline)" % (n, bad, refused, ref_lossy))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
