#!/usr/bin/env python3
"""Device-resident encode / decode time of an RGB 8-bit frame with the colour transform fused into the top DWT level, for checking
the chunk heights kernels_dwt.hip's fit_rounds picks against a fixed height (OJPHGPU_DWT_RP_COLOUR=8 in the environment).
    python tools/colour_chunks.py [width height [reversible]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
    rev = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (3, h // 8 + 1, w // 8 + 1)).astype(np.float32)
    img = np.clip(np.kron(base, np.ones((8, 8), np.float32))[:, :h, :w] + rng.normal(0, 3, (3, h, w)), 0, 255).astype(np.int32)
    plan = Plan(make_params(w, h, 3, bit_depth=8, reversible=rev, color_transform=True, qstep=-1.0 if rev else 0.01))
    d = torch.from_numpy(img.astype(np.int8)).cuda()
    enc = codec.Encoder(plan=plan)
    cs = enc.encode(d)
    dec = codec.Decoder(cs)
    for _ in range(3):
        enc.run_device(d); dec.run_device(dtype=torch.int8)
    torch.cuda.synchronize()
    te = td = fe = fd = 0.0
    n = 20
    for _ in range(n):
        enc.run_device(d); t = enc.timing(); te += t["total_ms"]; fe += t["dwt_levels_ms"][0]
        dec.run_device(dtype=torch.int8); t = dec.timing(); td += t["total_ms"]; fd += t["dwt_levels_ms"][-1]
    print("%dx%d rev=%d RP_COLOUR=%s: encode %.3f ms (top level %.4f)  decode %.3f ms (top level %.4f)  %d bytes" % (
        w, h, rev, os.environ.get("OJPHGPU_DWT_RP_COLOUR", "fitted"), te / n, fe / n, td / n, fd / n, len(cs)), flush=True)


if __name__ == "__main__":
    main()
