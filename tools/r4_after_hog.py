"""After the held-chip test, in the same process: are decodes slow / wrong?  (diagnosis of a flaky run)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from openjph_amd import codec
from openjph_amd.plan import Plan, make_params
from tests import cpu_pipeline as cp
from tests.synth import synth_image

def rounds(tag):
    img = synth_image(3, 100, 150, 7, seed=21, signed=True)
    want, *_ = cp.encode(img, bit_depth=7, is_signed=True)
    want_dec, _ = cp.decode(want)
    bad = 0; t0 = time.time(); retries = 0
    for it in range(40):
        dec = codec.Decoder(want)
        for tdt in (torch.int32, torch.int16, torch.int8):
            out = dec.run_device(dtype=tdt).cpu().numpy().astype(np.int64)
            ref = want_dec.astype(np.int64)
            if tdt != torch.int32:
                bits = 16 if tdt == torch.int16 else 8
                ref = np.clip(ref, -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
            if not np.array_equal(out, ref):
                bad += 1
                d = np.argwhere(out != ref)
                print(tag, "MISMATCH it", it, tdt, len(d), d[:3].tolist(), "failed", dec.failed_blocks(), "retries", dec.fused_retries())
        retries += dec.fused_retries()
    print("%s: 120 decodes, %d wrong, %d retries, %.2f s" % (tag, bad, retries, time.time() - t0))

rounds("before")
if len(sys.argv) > 1 and sys.argv[1] == "hog":
    from tests import test_gpu_contention as t
    t0 = time.time(); t.test_fused_launches_beside_each_other_and_under_a_held_chip(); print("held-chip test: %.1f s" % (time.time() - t0))
    rounds("after")
