import sys, os, subprocess, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openjph_amd import codec
from tests.synth import synth_image
if len(sys.argv) > 1:                       # child: reference decode with the separate launches
    cs = open("/tmp/cs.bin", "rb").read()
    np.save("/tmp/want.npy", codec.Decoder(cs).run_device().cpu().numpy())
    sys.exit(0)
img = synth_image(1, 256, 256, 8, seed=1234)
cs = codec.encode(img, bit_depth=8)
open("/tmp/cs.bin", "wb").write(cs)
subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, OJPHGPU_DEC_FUSED="0"), check=True)
want = np.load("/tmp/want.npy")
dec = codec.Decoder(cs)
got = dec.run_device().cpu().numpy()
print("failed blocks:", dec.failed_blocks(), " equal:", np.array_equal(got, want), " lossless:", np.array_equal(want, img))
pl = dec.plan
bad = []
for i, b in enumerate(pl.blocks):
    band = pl.bands[int(b["band"])]
    bad.append((i, int(band["res"]), int(band["band"]), int(b["w"]), int(b["h"])))
print("blocks (index, res, band, w, h):", bad)
d = np.argwhere(got != want)
print("differing samples:", len(d), d[:5].tolist() if len(d) else "")
got2 = dec.run_device().cpu().numpy()
print("second run failed:", dec.failed_blocks(), np.array_equal(got2, want))
