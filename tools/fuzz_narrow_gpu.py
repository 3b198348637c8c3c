#!/usr/bin/env python3
"""Random frames whose code-blocks are all at most 32 (or 16) columns wide through the HIP decoder -- step 2 with two / four blocks
to a wavefront (kernels_ht_dec.hip: ht_dec_step2_multi_kernel) -- against the oracle pipeline: clean streams (samples equal), then
the same streams with bytes of their block data changed, read resiliently (same picture, blocks refused or not).
GPU box (the CPU fuzzers pin the oracle pipeline to the live reference).      python tools/fuzz_narrow_gpu.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cpu_pipeline as cp
from tests.synth import synth_image
from openjph_amd import codec
from openjph_amd.plan import parse_codestream


def rand_case(rng):
    bw = int(rng.choice([4, 8, 16, 32])); bh = int(rng.choice([4, 8, 16, 32, 64, 128, 256]))
    while bw * bh > 4096:
        bh //= 2
    nc = int(rng.choice([1, 1, 2, 3]))
    h, w = int(rng.integers(1, 160)), int(rng.integers(1, 200))
    bd = int(rng.integers(2, 15))
    rev = bool(rng.integers(0, 2))
    kw = dict(bit_depth=bd, block=(bw, bh), num_decomps=int(rng.integers(0, 5)), reversible=rev)
    if not rev:
        kw["qstep"] = float(rng.choice([0.1, 0.02, 0.004]))
    if nc == 3 and rng.integers(0, 2):
        kw["color_transform"] = True
    if rng.integers(0, 3) == 0:
        kw["tile"] = (int(rng.integers(16, 128)), int(rng.integers(16, 128)))
    return nc, h, w, bd, kw


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    n = nd = bad = refused = 0
    while time.time() < t_end:
        nc, h, w, bd, kw = rand_case(rng)
        img = synth_image(nc, h, w, bd, seed=int(rng.integers(0, 1000)))
        try:
            cs = bytes(cp.encode(img, **kw)[0])
            want, _ = cp.decode(cs)
        except Exception:
            continue
        n += 1
        dec = codec.Decoder(cs)
        got = dec.run_device().cpu().numpy()
        if dec.failed_blocks() or not np.array_equal(got, want):
            bad += 1; print("clean stream differs:", nc, h, w, kw, flush=True)
            continue
        sod = cs.find(b"\xff\x93")
        if sod < 0 or len(cs) - sod < 60:
            continue
        for _ in range(3):
            c2 = bytearray(cs)
            for _ in range(int(rng.integers(1, 8))):
                c2[int(rng.integers(sod + 2, len(c2) - 2))] = int(rng.integers(0, 256))
            c2 = bytes(c2)
            try:
                plan = parse_codestream(c2, resilient=True)
                want = cp.inverse_stages(plan, cp.decode_blocks(plan, c2, resilient=True))
            except Exception:
                continue
            nd += 1
            try:
                d2 = codec.Decoder(c2, resilient=True)
                got = d2.run_device().cpu().numpy()
                refused += d2.failed_blocks() != 0
                if not np.array_equal(got, want):
                    bad += 1; print("damaged stream differs:", nc, h, w, kw, flush=True)
            except Exception as e:
                bad += 1; print("damaged stream raised on the device path:", nc, h, w, kw, repr(e)[:200], flush=True)
    print("%d random frames of narrow code-blocks and %d damaged copies (%d with refused blocks) through the HIP decoder: %d differ from the oracle pipeline" % (n, nd, refused, bad))


if __name__ == "__main__":
    main()
