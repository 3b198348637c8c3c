#!/usr/bin/env python3
"""The random Part-2 configurations of tools/fuzz_part2_cpu.py through the HIP path: the codec writes the oracle pipeline's
bytes and decodes the oracle pipeline's samples (full and half resolution).  GPU box (no reference there: the CPU fuzzer pins
the oracle pipeline to the live reference).      python tools/fuzz_part2_gpu.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_part2_cpu import rand_case
from tests import cpu_pipeline as cp
from tests.synth import synth_image
from openjph_amd import codec


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    n = bad = 0
    while time.time() < t_end:
        nc, h, w, bd, kw = rand_case(rng)
        img = synth_image(nc, h, w, bd, seed=int(rng.integers(0, 1000)))
        try:
            want, plan, *_ = cp.encode(img, **kw)
        except Exception:
            continue
        n += 1
        try:
            got = codec.encode(img, **kw)
            ok = got == want
            why = "" if ok else "encode: %d vs %d bytes, first difference at %d" % (len(got), len(want), next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1))
            wd, _ = cp.decode(want)
            gd = codec.decode(want)
            if not np.array_equal(gd, wd):
                ok = False; why += " decode: %d samples differ (max %d)" % (int((np.asarray(gd) != np.asarray(wd)).sum()), int(np.abs(np.asarray(gd, np.int64) - np.asarray(wd, np.int64)).max()))
            L0 = min(plan.comp_style(c)["num_decomps"] for c in range(nc))
            if ok and L0 >= 1 and min(h, w) >= 8:
                try:
                    w1, _ = cp.decode(want, skip=(1, 1))
                except Exception:
                    w1 = None
                if w1 is not None:
                    d = codec.Decoder(want, skip_res=(1, 1))
                    d1 = d.plan.unpack_frame(d.decode())
                    ok = all(np.array_equal(d1[c], w1[c]) for c in range(nc))
                    why = why or ("" if ok else "reduced resolution differs")
            if not ok:
                bad += 1
                print("MISMATCH [%s]" % why.strip(), nc, h, w, bd, kw, flush=True)
        except Exception as e:
            bad += 1
            print("ERROR %s: %s" % (type(e).__name__, str(e)[:200]), nc, h, w, bd, kw, flush=True)
    print("%d random Part-2 configurations through the HIP path: bytes and samples (full and half resolution) of the oracle pipeline, %d differ" % (n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
