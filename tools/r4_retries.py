"""Does the fused decoder launch ever ask for the repeat (a wait that ran out) on an idle chip?  It must not.
   OJPHGPU_DEC_FUSED=2 python tools/r4_retries.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from openjph_amd import codec
from tests.synth import synth_image
for (nc, h, w, bd, rev) in [(1, 256, 256, 8, True), (3, 1080, 1920, 8, True), (3, 2160, 3840, 10, False), (3, 4320, 7680, 12, False), (1, 777, 1333, 12, True)]:
    img = synth_image(nc, h, w, bd, seed=7)
    cs = codec.encode(img, bit_depth=bd, reversible=rev)
    dec = codec.Decoder(cs)
    t0 = time.time()
    for _ in range(6):
        dec.run_device()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 6
    print("%dx%dx%d %2d-bit %s: fused launch used %s, retries %d, failed blocks %d, %.2f ms per decode" %
          (nc, h, w, bd, "5/3" if rev else "9/7", getattr(dec, "last_fused", "?"), dec.fused_retries(), dec.failed_blocks(), dt * 1e3))
