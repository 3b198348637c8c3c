#!/usr/bin/env python3
"""Device-resident encode / decode time of 8K frames with sub-sampled chroma (4:4:4, 4:2:2, 4:2:0), 10-bit, 9/7.
    python tools/subsampled.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    w, h = 7680, 4320
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (512 + 300 * np.sin(xx / 97.0) * np.cos(yy / 61.0) + rng.normal(0, 20, (h, w))).clip(0, 1023).astype(np.int32)
    for name, ds in (("4:4:4", [(1, 1)] * 3), ("4:2:2", [(1, 1), (2, 1), (2, 1)]), ("4:2:0", [(1, 1), (2, 2), (2, 2)])):
        planes = [np.ascontiguousarray(base[::dy, ::dx]) for dx, dy in ds]
        plan = Plan(make_params(w, h, 3, bit_depth=10, reversible=False, qstep=0.002, downsampling=ds))
        flat = np.concatenate([p.reshape(-1) for p in planes]).astype(np.int16)
        d = torch.from_numpy(flat).cuda()
        if tuple(plan.frame_shape) != tuple(d.shape):
            d = d.reshape(plan.frame_shape)
        enc = codec.Encoder(plan=plan)
        cs = enc.encode(d)
        dec = codec.Decoder(cs)
        for _ in range(3):
            enc.run_device(d); dec.run_device(dtype=torch.int16)
        torch.cuda.synchronize()
        te = td = 0.0
        n = 10
        for _ in range(n):
            enc.run_device(d); te += enc.timing()["total_ms"]
            dec.run_device(dtype=torch.int16); td += dec.timing()["total_ms"]
        ns = sum(p.size for p in planes)
        print("%s  %.1f Msamples  encode %.3f ms  decode %.3f ms  -> %.1f Gsamples/s encode+decode  (%.2f bytes/sample)" % (
            name, ns / 1e6, te / n, td / n, ns / ((te + td) / n) / 1e6, len(cs) / ns), flush=True)
        del enc, dec


if __name__ == "__main__":
    main()
