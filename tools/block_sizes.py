#!/usr/bin/env python3
"""Device-resident encode / decode time of the 8K bench frame for several code-block sizes.   python tools/block_sizes.py [64x64 32x32 ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bench import WORKLOADS, workload_image
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    name = "c3_8k_444_12b_irv97"
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    img = workload_image(name)
    d = torch.from_numpy(img.astype(np.int16)).cuda()
    blocks = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(64, 64), (32, 32), (128, 32), (64, 32), (16, 16)]
    for block in blocks:
        plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep, block=block))
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
        t = dec.timing()
        print("block %3dx%-3d  encode %.3f ms  decode %.3f ms  (%d bytes)  dec stages %s" % (
            block[0], block[1], te / n, td / n, len(cs), {k: round(v, 3) for k, v in t.items() if k.endswith("_ms") and isinstance(v, (int, float))}), flush=True)
        del enc, dec


if __name__ == "__main__":
    main()
