"""Repeats the hot path many times and checks that every run gives the very same bytes / samples:
a race (LDS hand-over, atomics, cross-stream ordering) would show up as a rare difference."""
import hashlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from openjph_amd import codec
from openjph_amd.plan import make_params
from tests.synth import synth_image


def main():
    t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
    img = synth_image(3, 4320, 7680, 12, seed=1234)
    enc = codec.Encoder(make_params(7680, 4320, 3, bit_depth=12, reversible=False, qstep=0.001))
    d_img = torch.from_numpy(img.astype(np.int16)).cuda()
    ref_cs = enc.encode(d_img)
    ref_h = hashlib.sha256(ref_cs).hexdigest()
    dec = codec.Decoder(ref_cs)
    ref_out = dec.run_device(dtype=torch.int16).clone()
    runs = checks = 0
    while time.time() < t_end:
        for _ in range(8):
            enc.run_device(d_img)
            out = dec.run_device(dtype=torch.int16)
            runs += 1
        assert torch.equal(out, ref_out), "decode differs after %d runs" % runs
        assert hashlib.sha256(enc.finish()).hexdigest() == ref_h, "codestream differs after %d runs" % runs
        checks += 1
    assert dec.failed_blocks() == 0 and dec.fused_retries() == 0, "the fused launch asked for %d repeats" % dec.fused_retries()
    print("8K frame: %d runs, %d checks, all identical, no fused launch repeated" % (runs, checks))
    # small frames of many shapes, fresh codec objects every time (allocation / table upload paths)
    rng = np.random.default_rng(0)
    n = 0
    t_end = time.time() + 20
    while time.time() < t_end:
        w, h, nc = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 4))
        rev = bool(rng.integers(0, 2))
        im = synth_image(nc, h, w, 10, seed=int(rng.integers(0, 1000)))
        kw = dict(bit_depth=10, reversible=rev, num_decomps=int(rng.integers(0, 6)))
        a = codec.encode(im, **kw)
        b = codec.encode(im, **kw)
        assert a == b
        d1 = codec.decode(a)
        d2 = codec.decode(a)
        assert np.array_equal(d1, d2) and (not rev or np.array_equal(d1, im))
        n += 1
    print("small frames: %d shapes, all identical" % n)


if __name__ == "__main__":
    main()
