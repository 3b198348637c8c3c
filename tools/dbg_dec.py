import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openjph_amd import codec
from oracle import oraclebind as ob
from tests.synth import random_block
from tests.test_gpu_stages import _block_cases
rng = np.random.default_rng(9)
cases = _block_cases(rng, 92)
descs = np.zeros(len(cases), codec.cb_desc_dtype)
datas, expect, off, doff = [], [], 0, 0
for i, (w, h, kmax, dens, amp) in enumerate(cases):
    pitch = (w + 63) & ~63
    sm, v = random_block(rng, w, h, w, kmax, dens, amp)
    coded = ob.ht_encode(sm, w, h, w, kmax - 1, 0) if np.any(np.abs(v[:, :w]) > 0) else b""
    d = descs[i]
    d["coef_off"], d["pitch"], d["w"], d["h"] = off, pitch, w, h
    d["K_max"], d["reversible"], d["missing_msbs"] = kmax, 1, kmax - 1
    d["num_passes"], d["len1"], d["len2"], d["data_off"] = (1 if coded else 0), len(coded), 0, doff
    if coded:
        ok, dec = ob.ht_decode(coded, w, h, w, kmax - 1)
        expect.append(ob.dequant_rev(dec, kmax))
    else:
        expect.append(np.zeros((h, w), np.int32))
    datas.append(np.frombuffer(coded, np.uint8))
    off += pitch * h; doff += len(coded)
coef = torch.full((off + 64,), 0x5A5A5A5A, dtype=torch.int32).cuda()
status = codec.ht_decode(descs, np.concatenate(datas), coef)
got = coef.cpu().numpy()
for i, (w, h, kmax, dens, amp) in enumerate(cases):
    d = descs[i]
    g = np.lib.stride_tricks.as_strided(got[int(d["coef_off"]):], (h, w), (int(d["pitch"]) * 4, 4))
    bad = (g != expect[i])
    if status[i] or bad.any():
        ys, xs = np.nonzero(bad)
        scup = 0
        if len(datas[i]) >= 2: scup = (int(datas[i][-1]) << 4) | (int(datas[i][-2]) & 15)
        print(i, cases[i], "len", len(datas[i]), "scup", scup, "status", status[i], "nbad", int(bad.sum()),
              "first", (int(ys[0]), int(xs[0])) if len(ys) else None)
print("done")
