"""How many HIP streams overlap: two codec pairs alive at once (each object owns a side stream), the
second pair on two user streams -- step time of the second pair with the default number of hardware
queues and with GPU_MAX_HW_QUEUES raised (run as: python tools/hw_queues.py; it re-executes itself)."""
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")


def measure():
    import numpy as np
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    from tests.synth import synth_image
    img = synth_image(3, 4320, 7680, 12, seed=1234)
    d_img = torch.from_numpy(img.astype(np.int16)).cuda()
    plan = Plan(make_params(7680, 4320, 3, bit_depth=12, reversible=False, qstep=0.001))
    enc = codec.Encoder(plan=plan)                       # first pair: on the default stream, kept alive
    cs = enc.encode(d_img)
    dec = codec.Decoder(cs)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        enc2 = codec.Encoder(plan=plan)
    with torch.cuda.stream(s2):
        dec2 = codec.Decoder(cs)
    out = torch.empty_like(d_img)
    for objs in ((enc2, dec2),):
        for _ in range(5):
            objs[0].run_device(d_img); objs[1].run_device(out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            objs[0].run_device(d_img); objs[1].run_device(out)
        torch.cuda.synchronize()
        print("GPU_MAX_HW_QUEUES=%s: %.4f ms per step on two user streams (first pair alive)"
              % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), (time.perf_counter() - t0) * 1e3 / 30))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        measure()
    else:
        for q in (None, "8"):
            env = dict(os.environ)
            if q:
                env["GPU_MAX_HW_QUEUES"] = q
            subprocess.run([sys.executable, __file__, "run"], env=env)
