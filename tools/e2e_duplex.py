#!/usr/bin/env python3
"""Both frame pipes at once (a transcoder: per step one frame up and one codestream down through the encoder pipe, one
codestream up and one frame down through the decoder pipe), for every pairing of the pipes' copy modes
(OJPHGPU_ENC_COPY_MODE / OJPHGPU_DEC_COPY_MODE: 0 upload SDMA + download copy kernel, 1 upload copy kernel + download SDMA,
2 both SDMA).  The timed regions start together (a barrier behind the slot filling).

    python tools/e2e_duplex.py [--frames 48] [--packed 12]
"""
import argparse
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--packed", type=int, default=0)
    ap.add_argument("--modes", default="00,01,02,10,11,12,20,21,22")
    args = ap.parse_args()
    from bench import WORKLOADS, workload_image, run_encoder_pipe, run_decoder_pipe
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    name = "c3_8k_444_12b_irv97"
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile))
    img = workload_image(name)
    cs = codec.Encoder(plan=plan).encode(img)
    n, depth, threads = args.frames, 6, 4
    pk = args.packed or None
    for wgs in ("8", "32"):
        os.environ["OJPHGPU_COPY_WGS"] = wgs
        for m in args.modes.split(","):
            os.environ["OJPHGPU_ENC_COPY_MODE"], os.environ["OJPHGPU_DEC_COPY_MODE"] = m[0], m[1]
            alone_e = run_encoder_pipe(plan, img, n, depth, threads, container=16, packed=pk)[0]
            alone_d = run_decoder_pipe(cs, n, depth, threads, container=16, packed=pk)[0]
            res = {}
            gate = threading.Barrier(2)
            te = threading.Thread(target=lambda: res.__setitem__("e", run_encoder_pipe(plan, img, n, depth, threads, container=16, packed=pk, start=gate)[0]))
            td = threading.Thread(target=lambda: res.__setitem__("d", run_decoder_pipe(cs, n, depth, threads, container=16, packed=pk, start=gate)[0]))
            te.start(); td.start(); te.join(); td.join()
            if "e" not in res or "d" not in res:
                print("enc mode %s dec mode %s wgs %s: FAILED" % (m[0], m[1], wgs), flush=True)
                continue
            wall = max(res["e"], res["d"])
            print("enc mode %s dec mode %s copy wgs %2s | alone: enc %.3f dec %.3f ms/frame | together: enc %.3f dec %.3f -> %.3f ms per step, %.1f Gsamples/s per direction"
                  % (m[0], m[1], wgs, alone_e * 1e3 / n, alone_d * 1e3 / n, res["e"] * 1e3 / n, res["d"] * 1e3 / n, wall * 1e3 / n, img.size * n / wall / 1e9), flush=True)


if __name__ == "__main__":
    main()
