#!/usr/bin/env python3
"""Does replaying the step's launches from a HIP graph shorten it?  Captures encoder.run_device + decoder.run_device
(their launches, the side-stream fork / join, the counters' memset) into a graph with torch's capture on the stream the
codec objects were created on, and times plain submission against graph replay.   python tools/graph_probe.py"""
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
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = torch.from_numpy(img.astype(np.int16)).cuda()
        plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep))
        enc = codec.Encoder(plan=plan)
        cs = enc.encode(img)
        dec = codec.Decoder(cs)
        out = torch.empty_like(d)
        for _ in range(5):
            enc.run_device(d); dec.run_device(out)
        s.synchronize()

        def timed(fn, n=300):
            fn(); s.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            s.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        def plain():
            enc.run_device(d); dec.run_device(out)
        a = timed(plain)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            enc.run_device(d); dec.run_device(out)
        b = timed(g.replay)
        a2 = timed(plain); b2 = timed(g.replay)
        ref = dec.run_device(out)
        s.synchronize()
        print("plain %.4f %.4f ms per step, graph replay %.4f %.4f ms per step" % (a, a2, b, b2))


if __name__ == "__main__":
    main()
