import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from openjph_amd import codec
from openjph_amd.plan import make_params, Plan
from tests.synth import synth_image
img = synth_image(3, 4320, 7680, 12, seed=1234)
plan = Plan(make_params(7680, 4320, 3, bit_depth=12, reversible=False, qstep=0.001))
enc = codec.Encoder(plan=plan)
pin = torch.from_numpy(img).pin_memory()
def t(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ms, d = t(lambda: torch.from_numpy(img).cuda()); print("H2D pageable int32 400MB: %.2f ms" % ms)
ms, d = t(lambda: pin.cuda(non_blocking=True)); print("H2D pinned int32: %.2f ms" % ms)
pin16 = torch.from_numpy(img.astype(np.int16)).pin_memory()
ms, d16 = t(lambda: pin16.cuda(non_blocking=True).to(torch.int32)); print("H2D pinned int16 + widen on device: %.2f ms" % ms)
ms, _ = t(lambda: enc.run_device(d)); print("run_device: %.2f ms" % ms)
ms, cs = t(lambda: enc.finish()); print("finish (D2H + T2): %.2f ms, %d bytes" % (ms, len(cs)))
t0 = time.perf_counter(); p2 = codec.parse_codestream(cs); print("parse: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
dec = codec.Decoder(cs)
ms, o = t(lambda: dec.run_device()); print("decode run_device: %.2f ms" % ms)
ms, h = t(lambda: o.cpu()); print("D2H image pageable: %.2f ms" % ms)
