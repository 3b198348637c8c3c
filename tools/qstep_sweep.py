#!/usr/bin/env python3
"""Device-resident encode / decode time of the 8K bench frame over quantisation step sizes (rates from visually
lossless down to heavy compression).   python tools/qstep_sweep.py"""
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
    w, h, nc, bd, rev, ct, _, tile = WORKLOADS[name]
    img = workload_image(name)
    d = torch.from_numpy(img.astype(np.int16)).cuda()
    for qstep in (0.001, 0.004, 0.01, 0.03, 0.1):
        plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep))
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
        print("qstep %.3f  %.3f bytes/sample  encode %.3f ms  decode %.3f ms" % (qstep, len(cs) / img.size, te / n, td / n), flush=True)
        if os.environ.get("SWEEP_STAGES"):                    # where the time goes: the stages of the last timed run, and the same loops untimed
            t = enc.timing()
            print("    encode stages: dwt %.3f  ht %.3f  ht launches %s  levels %s" % (
                t["dwt_ms"], t["ht_ms"], ["%.3f" % x for x in t["ht_launches_ms"]], ["%.3f" % x for x in t["dwt_levels_ms"]]), flush=True)
            for what, run in (("encode", lambda: enc.run_device(d)), ("decode", lambda: dec.run_device(dtype=torch.int16))):
                for _ in range(20):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(torch.cuda.current_stream())
                for _ in range(100):
                    run()
                e1.record(torch.cuda.current_stream())
                torch.cuda.synchronize()
                print("    %s alone, 100 runs back to back: %.3f ms each" % (what, e0.elapsed_time(e1) / 100), flush=True)
        del enc, dec


if __name__ == "__main__":
    main()
