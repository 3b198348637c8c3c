import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from openjph_amd import codec
from openjph_amd.plan import make_params
from oracle import refbind
W, H = 24001, 20003
rng = np.random.default_rng(5)
yy = (np.arange(H, dtype=np.int32)[:, None] * 3) & 0xFF
xx = (np.arange(W, dtype=np.int32)[None, :] * 7) & 0xFF
img = ((yy + xx + rng.integers(0, 9, (H, W), dtype=np.int32)) & 0xFF)[None]
t = time.time(); cs = codec.Encoder(make_params(W, H, 1, bit_depth=8)).encode(img); print("gpu encode", len(cs), round(time.time() - t, 2), "s")
t = time.time(); back = codec.Decoder(cs).decode(); print("gpu decode lossless:", np.array_equal(back, img), round(time.time() - t, 2), "s")
r = refbind.Ref()
t = time.time(); rd, _ = r.decode(cs); print("reference decodes the GPU stream losslessly:", np.array_equal(rd, img), round(time.time() - t, 2), "s")
t = time.time(); rcs = r.encode(img, 8); print("reference stream identical:", rcs == cs, round(time.time() - t, 2), "s")
