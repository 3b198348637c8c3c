#!/usr/bin/env python3
"""Device-resident encode / decode time of the 8K bench frame for several tile sizes.   python tools/tile_sizes.py"""
import os
import sys
import time

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
    for tile in ((0, 0), (2048, 2048), (1024, 1024), (512, 512), (256, 256)):
        t0 = time.perf_counter()
        plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep, tile=tile))
        enc = codec.Encoder(plan=plan)
        t_create = time.perf_counter() - t0
        t0 = time.perf_counter()
        cs = enc.encode(d)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        dec = codec.Decoder(cs)
        t_dec_create = time.perf_counter() - t0
        for _ in range(3):
            enc.run_device(d); dec.run_device(dtype=torch.int16)
        torch.cuda.synchronize()
        te = td = 0.0
        n = 10
        for _ in range(n):
            enc.run_device(d); te += enc.timing()["total_ms"]
            dec.run_device(dtype=torch.int16); td += dec.timing()["total_ms"]
        print("tile %4dx%-4d (%4d tiles)  encode %.3f ms  decode %.3f ms  | plan+encoder create %.2f s, first encode + finish %.3f s, parse + decoder create %.2f s" % (
            tile[0], tile[1], plan.num_tiles, te / n, td / n, t_create, t_first, t_dec_create), flush=True)
        del enc, dec


if __name__ == "__main__":
    main()
