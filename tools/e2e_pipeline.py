#!/usr/bin/env python3
"""Steady-state end-to-end throughput of the frame pipelines (host memory -> codestream in host memory
and back), PCIe and host Tier-2 included: the figure a capture / playback process sees.

    python tools/e2e_pipeline.py [--workload c3|c5|c2] [--frames 48] [--depth 4] [--threads 2] [--container 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--container", type=int, default=16, help="bits per sample in host memory: 8, 16 or 32")
    ap.add_argument("--packed", type=int, default=0, help="frames cross PCIe as bit-packed planes: 10, 12 or 14 bits per sample")
    args = ap.parse_args()
    import torch
    from bench import WORKLOADS, workload_image, pcie_bandwidth, run_encoder_pipe, run_decoder_pipe
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    name = [k for k in WORKLOADS if k.startswith(args.workload)][0]
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile))
    img = workload_image(name)
    nsamp = img.size
    res = {"workload": name, "frames": args.frames, "depth": args.depth, "host_threads": args.threads, "pcie_GBps": pcie_bandwidth(torch)}
    want = codec.Encoder(plan=plan).encode(img)
    dt, st = run_encoder_pipe(plan, img, args.frames, args.depth, args.threads, container=args.container, want=want, packed=args.packed or None)
    res["encode"] = {"Msamples_s": round(nsamp * args.frames / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3 / args.frames, 3), **st,
                     "h2d_GBps": round(img.size * ((args.packed or args.container) / 8) * args.frames / dt / 1e9, 1)}
    ref = codec.decode(want)
    dt, st = run_decoder_pipe(want, args.frames, args.depth, args.threads, container=args.container, want=ref, packed=args.packed or None)
    res["decode"] = {"Msamples_s": round(nsamp * args.frames / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3 / args.frames, 3), **st,
                     "d2h_GBps": round(img.size * ((args.packed or args.container) / 8) * args.frames / dt / 1e9, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
