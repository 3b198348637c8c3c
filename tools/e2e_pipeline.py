#!/usr/bin/env python3
"""Steady-state end-to-end throughput of the frame pipelines (host memory -> codestream in host memory
and back), PCIe and host Tier-2 included: the figure a capture / playback process sees.

    python tools/e2e_pipeline.py [--workload c3|c5|c2] [--frames 48] [--depth 4] [--threads 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pcie_bandwidth(torch, nbytes=256 << 20, reps=5):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = {}
    for name, (dst, src) in dict(h2d=(d, h), d2h=(h, d)).items():
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[name] = nbytes * reps / (time.perf_counter() - t0) / 1e9
    return out


def run_encoder(plan, frames, n, depth, threads, container=16):
    from openjph_amd.pipeline import EncoderPipe
    pipe = EncoderPipe(plan=plan, depth=depth, container=container, host_threads=threads)
    k = 0
    while True:                                       # every slot gets a frame once; later submissions re-send the slot as it is
        buf = pipe.acquire()
        if buf is None or k >= depth:
            break
        buf[:] = frames[k % len(frames)].astype(buf.dtype)
        pipe.submit(); k += 1
    outs = []
    while pipe.in_flight:
        outs.append(pipe.collect())
    # timed: n frames
    lens = []
    t0 = time.perf_counter()
    sub = col = 0
    while col < n:
        while sub < n and pipe.acquire() is not None:
            pipe.submit(); sub += 1
        lens.append(len(pipe.collect(copy=False))); col += 1
    dt = time.perf_counter() - t0
    st = pipe.stats()
    pipe.close()
    return dt, outs, st, lens


def run_decoder(streams, n, depth, threads, container=16):
    from openjph_amd.pipeline import DecoderPipe
    pipe = DecoderPipe(streams[0], depth=depth, container=container, host_threads=threads)
    k = 0
    while k < depth:
        cs = streams[k % len(streams)]
        buf = pipe.acquire(len(cs))
        if buf is None:
            break
        buf[:] = np.frombuffer(cs, np.uint8)
        pipe.submit(); k += 1
    first = None
    while pipe.in_flight:
        f = pipe.collect()
        first = f if first is None else first
    ln = len(streams[0])
    t0 = time.perf_counter()
    sub = col = 0
    while col < n:
        while sub < n and pipe.acquire(len(streams[sub % len(streams)])) is not None:      # the slot still holds a codestream of the sequence
            pipe.submit(); sub += 1
        pipe.collect(copy=False); col += 1
    dt = time.perf_counter() - t0
    st = pipe.stats()
    pipe.close()
    return dt, first, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--threads", type=int, default=2)
    args = ap.parse_args()
    import torch
    from bench import WORKLOADS, workload_image
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    name = [k for k in WORKLOADS if k.startswith(args.workload)][0]
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile))
    img = workload_image(name)
    nsamp = img.size
    res = {"workload": name, "frames": args.frames, "depth": args.depth, "host_threads": args.threads, "pcie_GBps": pcie_bandwidth(torch)}
    want = codec.Encoder(plan=plan).encode(img)
    dt, outs, st, lens = run_encoder(plan, [img], args.frames, args.depth, args.threads)
    assert outs[0] == want, "pipeline codestream differs from the one-frame encoder's"
    assert all(l == len(want) for l in lens)
    res["encode"] = {"Msamples_s": round(nsamp * args.frames / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3 / args.frames, 3), **st,
                     "h2d_GBps": round(img.size * 2 * args.frames / dt / 1e9, 1)}
    ref = codec.decode(want)
    dt, first, st = run_decoder([want], args.frames, args.depth, args.threads)
    assert np.array_equal(first.astype(np.int64), ref.astype(np.int64)), "pipeline frame differs from the one-frame decoder's"
    res["decode"] = {"Msamples_s": round(nsamp * args.frames / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3 / args.frames, 3), **st,
                     "d2h_GBps": round(img.size * 2 * args.frames / dt / 1e9, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
