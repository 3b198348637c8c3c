"""GPU parity of every hot-path stage against the oracle, called through the C ABI
(ojphgpu_dwt_forward/inverse, ojphgpu_ht_encode, ojphgpu_ht_decode).  Bit-exact for the integer
5/3 path and the block coder; the 9/7 float path is compared bit-exactly too (same fp32
add-mul-add order, no contraction) with the stated fallback tolerance of 1 ulp-level absolute
error 2^-20 of full scale documented in DESIGN.md."""
import numpy as np
import pytest

from tests.synth import random_block

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _plane_layout(shapes):
    """allocates planes in a flat arena; returns offsets (elements) and total size"""
    offs, total = [], 0
    for (h, w) in shapes:
        pitch = (max(w, 1) + 63) & ~63
        offs.append((total, pitch))
        total += pitch * max(h, 1) + 64
        total = (total + 63) & ~63
    return offs, total


@pytest.mark.parametrize("reversible", [True, False])
def test_dwt_forward_inverse_vs_oracle(reversible):
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(11)
    cases = [(64, 64, 1, 1), (130, 257, 1, 1), (257, 130, 0, 1), (131, 77, 1, 0), (96, 200, 0, 0),
             (1, 37, 1, 1), (1, 37, 0, 1), (40, 1, 1, 1), (40, 1, 1, 0), (2, 2, 1, 1), (3, 5, 0, 0),
             (513, 1031, 1, 1), (300, 121, 1, 1), (300, 122, 0, 1), (7, 300, 1, 1)]
    dt = np.int32 if reversible else np.float32
    descs = np.zeros(len(cases), codec.dwt_desc_dtype)
    shapes = []
    for (h, w, xe, ye) in cases:
        lw, hw, lh, hh = ob.band_dims(w, h, bool(xe), bool(ye))
        shapes += [(h, w), (lh, lw), (lh, hw), (hh, lw), (hh, hw)]
    offs, total = _plane_layout(shapes)
    arena = np.zeros(total, np.uint32)
    inputs, expect = [], []
    for i, (h, w, xe, ye) in enumerate(cases):
        if reversible:
            src = rng.integers(-40000, 40000, size=(h, w)).astype(np.int32)
        else:
            src = (rng.random((h, w)) - 0.5).astype(np.float32)
        inputs.append(src)
        o = offs[5 * i:5 * i + 5]
        d = descs[i]
        d["src_off"], d["src_pitch"] = o[0]
        for k, name in enumerate(("ll", "hl", "lh", "hh")):
            d[name + "_off"], d[name + "_pitch"] = o[1 + k]
        d["w"], d["h"], d["x_even"], d["y_even"] = w, h, xe, ye
        a = arena.view(dt)
        np.lib.stride_tricks.as_strided(a[o[0][0]:], (h, w), (o[0][1] * 4, 4))[:] = src
        expect.append((ob.dwt53_fwd if reversible else ob.dwt97_fwd)(src, bool(xe), bool(ye)))
    d_arena = torch.from_numpy(arena.view(np.int32)).cuda()
    max_w = max(c[1] for c in cases); max_h = max(c[0] for c in cases)
    codec.dwt("forward", reversible, descs, d_arena, max_w, max_h)
    got = d_arena.cpu().numpy().view(dt)
    for i, (h, w, xe, ye) in enumerate(cases):
        o = offs[5 * i:5 * i + 5]
        for k, name in enumerate(("ll", "hl", "lh", "hh")):
            e = expect[i][k]
            if e.size == 0:
                continue
            g = np.lib.stride_tricks.as_strided(got[o[1 + k][0]:], e.shape, (o[1 + k][1] * 4, 4))
            assert np.array_equal(g.view(np.uint32), e.view(np.uint32)), \
                "forward case %d %s band %s: max diff %g" % (i, cases[i], name, np.abs(g - e).max())
    # inverse: wipe the source planes, synthesise from the (oracle-exact) bands
    for i, (h, w, xe, ye) in enumerate(cases):
        o = offs[5 * i]
        np.lib.stride_tricks.as_strided(got[o[0]:], (h, w), (o[1] * 4, 4))[:] = 0
    d_arena = torch.from_numpy(got.view(np.int32).copy()).cuda()
    codec.dwt("inverse", reversible, descs, d_arena, max_w, max_h)
    got = d_arena.cpu().numpy().view(dt)
    for i, (h, w, xe, ye) in enumerate(cases):
        o = offs[5 * i]
        g = np.lib.stride_tricks.as_strided(got[o[0]:], (h, w), (o[1] * 4, 4))
        e = (ob.dwt53_inv if reversible else ob.dwt97_inv)(*expect[i], w, h, bool(xe), bool(ye))
        assert np.array_equal(g.view(np.uint32), e.view(np.uint32)), \
            "inverse case %d %s: max diff %g" % (i, cases[i], np.abs(g - e).max())
        if reversible:
            assert np.array_equal(g, inputs[i])


WIDE_UP_TO_128 = [(128, 32), (72, 40), (128, 2), (96, 8), (65, 63), (127, 31), (100, 1), (128, 32), (66, 5), (128, 17)]


def _block_cases(rng, n, shapes=None):
    shapes = shapes or ([(64, 64)] * 6 + [(32, 32), (128, 32), (32, 128), (4, 1024), (1024, 4), (64, 17), (17, 64), (1, 1),
                                          (3, 3), (5, 64), (64, 5), (2, 64), (63, 63), (33, 31), (8, 8), (1, 64), (64, 1)])
    out = []
    for i in range(n):
        w, h = shapes[i % len(shapes)]
        kmax = int(rng.integers(2, 22))
        dens = float(rng.choice([0.0, 0.002, 0.02, 0.2, 0.6, 1.0]))
        amp = int(min(2 ** kmax - 1, rng.choice([1, 2, 5, 40, 700, 2 ** kmax - 1])))
        out.append((w, h, kmax, dens, amp))
    return out


def test_ht_encode_vs_oracle_reversible_blocks():
    """Random sign-magnitude blocks fed as reversible coefficients (K5 + K8 fused)."""
    torch = _torch()
    from openjph_amd import codec
    from openjph_amd.csrc_consts import block_scratch_bytes
    from oracle import oraclebind as ob
    rng = np.random.default_rng(5)
    cases = _block_cases(rng, 92)
    descs = np.zeros(len(cases), codec.cb_desc_dtype)
    coefs, expect, off, soff = [], [], 0, 0
    for i, (w, h, kmax, dens, amp) in enumerate(cases):
        pitch = (w + 63) & ~63
        sm, v = random_block(rng, w, h, pitch, kmax, dens, amp)
        plane = np.zeros((h, pitch), np.int32); plane[:, :w] = v[:, :w]
        coefs.append(plane.ravel())
        q, mx = ob.quant_rev(plane[:, :w], kmax)
        expect.append(ob.ht_encode(q, w, h, w, kmax - 1, 0) if mx >= (1 << (31 - kmax)) else b"")
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = off, pitch, w, h
        d["K_max"], d["reversible"], d["delta"] = kmax, 1, 0.0
        d["data_off"], d["scratch_cap"] = soff, block_scratch_bytes(w, h, kmax)
        off += plane.size; soff += int(d["scratch_cap"])
    coef = torch.from_numpy(np.concatenate(coefs)).cuda()
    res, out, status = codec.ht_encode(descs, coef, soff, soff)
    assert status == 0
    bad = []
    for i, e in enumerate(expect):
        o, n = int(res[i, 0]), int(res[i, 1])
        g = out[o:o + n].tobytes()
        if g != e:
            first = next((k for k in range(min(len(g), len(e))) if g[k] != e[k]), min(len(g), len(e)))
            bad.append((i, cases[i], len(g), len(e), first))
    assert not bad, "HT encode mismatches (idx, case, got_len, want_len, first_diff): %s" % bad[:8]


def test_ht_encode_irreversible_quantisation():
    """Float coefficients with the irreversible quantiser in front (K6 + K8 fused)."""
    torch = _torch()
    from openjph_amd import codec
    from openjph_amd.csrc_consts import block_scratch_bytes
    from oracle import oraclebind as ob
    rng = np.random.default_rng(6)
    n = 24
    descs = np.zeros(n, codec.cb_desc_dtype)
    coefs, expect, off, soff = [], [], 0, 0
    for i in range(n):
        w, h = (64, 64) if i % 3 else (37, 50)
        kmax = int(rng.integers(8, 20))
        delta = np.float32(2.0 ** -int(rng.integers(3, 10)) * (1.0 + rng.random())) / np.float32(1 << (31 - kmax))
        pitch = (w + 63) & ~63
        plane = np.zeros((h, pitch), np.float32)
        plane[:, :w] = ((rng.random((h, w)) - 0.5) * (rng.random((h, w)) < 0.5) * 0.2).astype(np.float32)
        coefs.append(plane.view(np.int32).ravel())
        delta_inv = np.float32(1.0) / np.float32(delta)
        q, mx = ob.quant_irv(plane[:, :w], float(delta_inv))
        expect.append(ob.ht_encode(q, w, h, w, kmax - 1, 0) if mx >= (1 << (31 - kmax)) else b"")
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = off, pitch, w, h
        d["K_max"], d["reversible"], d["delta"] = kmax, 0, delta
        d["data_off"], d["scratch_cap"] = soff, block_scratch_bytes(w, h, kmax)
        off += plane.size; soff += int(d["scratch_cap"])
    coef = torch.from_numpy(np.concatenate(coefs)).cuda()
    res, out, status = codec.ht_encode(descs, coef, soff, soff)
    assert status == 0
    for i, e in enumerate(expect):
        o, nn = int(res[i, 0]), int(res[i, 1])
        assert out[o:o + nn].tobytes() == e, "block %d (%d vs %d bytes)" % (i, nn, len(e))


def test_ht_encode_coefficients_beyond_K_max():
    """Coefficients with more magnitude bits than K_max (Part-2 kernels whose gain outruns the guard bits) leave the
    reference's transfer the way its 32-bit arithmetic has it (ojph_codestream_gen.cpp:59-121): reversible, the bits above
    K_max are shifted out and bit K_max lands on the sign; irreversible, the conversion of a product beyond 2^31 returns
    INT_MIN -- a zero.  Either leaves a bit in max_val, and a block with a non-zero max_val is coded even when none of its
    samples is significant (ojph_codeblock.cpp:142-175).  The narrow and the wide kernel against the oracle's transfer +
    coder (pinned to the reference on such words: tests/test_cpu_wide.py)."""
    torch = _torch()
    from openjph_amd import codec
    from openjph_amd.csrc_consts import block_scratch_bytes
    from oracle import oraclebind as ob
    rng = np.random.default_rng(77)
    shapes = [(64, 64), (32, 32), (37, 50), (64, 17), (5, 7), (128, 32), (16, 16), (63, 63)]
    trials = []
    for it in range(96):
        w, h = shapes[it % len(shapes)]
        rev = (it // len(shapes)) % 2 == 0
        kind = (it // (2 * len(shapes))) % 3        # 0: one overflowing sample in an otherwise zero block, 1: a few among ordinary samples, 2: many
        kmax = int(rng.integers(6, 14))
        trials.append((w, h, rev, kind, kmax))
    descs = np.zeros(len(trials), codec.cb_desc_dtype)
    coefs, expect, off, soff = [], [], 0, 0
    for i, (w, h, rev, kind, kmax) in enumerate(trials):
        pitch = (w + 63) & ~63
        dens = 0.0 if kind == 0 else 0.3
        mag = (rng.integers(0, 1 << kmax, size=(h, w)) * (rng.random((h, w)) < dens)).astype(np.int64)
        nover = 1 if kind == 0 else (3 if kind == 1 else max(1, w * h // 6))
        for _ in range(nover):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            mag[y, x] = (1 << kmax) * int(rng.choice([1, 1, 2, 3, 5, 64])) + int(rng.integers(0, 1 << kmax)) * int(rng.integers(0, 2))
        sign = rng.integers(0, 2, size=(h, w))
        v = np.where(sign == 1, -mag, mag)
        d = descs[i]
        if rev:
            plane = np.zeros((h, pitch), np.int32); plane[:, :w] = v
            q, mx = ob.quant_rev(np.ascontiguousarray(plane[:, :w]), kmax)
            d["delta"] = 0.0
            coefs.append(plane.ravel())
        else:
            delta = np.float32(2.0 ** -5 * 1.25) / np.float32(1 << (31 - kmax))
            delta_inv = np.float32(1.0) / np.float32(delta)
            plane = np.zeros((h, pitch), np.float32)
            plane[:, :w] = (v.astype(np.float64) * float(np.float32(2.0 ** -5 * 1.25)) * 1.0001).astype(np.float32)
            if kind == 2:
                plane[0, 0] = np.float32(np.inf); plane[h - 1, w - 1] = np.float32(-3.0e38)
            q, mx = ob.quant_irv(np.ascontiguousarray(plane[:, :w]), float(delta_inv))
            d["delta"] = delta
            coefs.append(plane.view(np.int32).ravel())
        expect.append(ob.ht_encode(q, w, h, w, kmax - 1, 0) if mx >= (1 << (31 - kmax)) else b"")
        d["coef_off"], d["pitch"], d["w"], d["h"] = off, pitch, w, h
        d["K_max"], d["reversible"] = kmax, 1 if rev else 0
        d["data_off"], d["scratch_cap"] = soff, block_scratch_bytes(w, h, kmax)
        off += plane.size; soff += int(d["scratch_cap"])
    assert sum(1 for e in expect if 0 < len(e) < 24) >= 8          # (the "coded although nothing is significant" blocks are among them)
    coef = torch.from_numpy(np.concatenate(coefs)).cuda()
    res, out, status = codec.ht_encode(descs, coef, soff, soff)
    assert status == 0
    bad = []
    for i, e in enumerate(expect):
        o, n = int(res[i, 0]), int(res[i, 1])
        g = out[o:o + n].tobytes()
        if g != e:
            bad.append((i, trials[i], len(g), len(e)))
    assert not bad, "HT encode mismatches (idx, (w, h, rev, kind, kmax), got_len, want_len): %s" % bad[:8]


@pytest.mark.parametrize("shapes", [None, WIDE_UP_TO_128], ids=["every-shape", "65-to-128-columns"])
def test_ht_decode_vs_oracle(shapes):
    """(the second set: every block of the launch between 65 and 128 columns wide -- the chains keep the significance of the row
    above in a 128-bit mask per lane, step1_rows<2>; with wider blocks in the wavefront, as in the first set, they re-read it
    from the records)"""
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(9)
    cases = _block_cases(rng, 92 if shapes is None else 130, shapes)
    descs = np.zeros(len(cases), codec.cb_desc_dtype)
    datas, expect, off, doff, max_len = [], [], 0, 0, 0
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
            assert ok
            expect.append(ob.dequant_rev(dec, kmax))
        else:
            expect.append(np.zeros((h, w), np.int32))
        datas.append(np.frombuffer(coded, np.uint8))
        off += pitch * h; doff += len(coded); max_len = max(max_len, len(coded))
    coef = torch.full((off + 64,), 0x5A5A5A5A, dtype=torch.int32).cuda()
    status = codec.ht_decode(descs, np.concatenate(datas), coef)
    got = coef.cpu().numpy()
    assert not status.any(), "failed blocks: %s" % np.nonzero(status)[0][:10]
    for i, (w, h, kmax, dens, amp) in enumerate(cases):
        d = descs[i]
        g = np.lib.stride_tricks.as_strided(got[int(d["coef_off"]):], (h, w), (int(d["pitch"]) * 4, 4))
        assert np.array_equal(g, expect[i]), "decode block %d %s: %d samples differ" % (
            i, cases[i], int((g != expect[i]).sum()))


def test_ht_decode_corrupt_segments_match_oracle():
    """Corrupted / truncated cleanup segments: same accept/reject verdict as the oracle (itself pinned
    to ojph_decode_codeblock32 on such inputs) and, when accepted, the same samples."""
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(21)
    w = h = 64
    kmax = 11
    sm, v = random_block(rng, w, h, w, kmax, 0.5, 700)
    good = ob.ht_encode(sm, w, h, w, kmax - 1, 0)
    trials = [good, good[:2], good[:len(good) // 2], good[:-1], b"\x00\x00", b"\xff\xff\xff\xff", good + b"\x00"]
    for _ in range(120):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 5))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        trials.append(bytes(b))
    for _ in range(30):                           # damage concentrated in the MEL/VLC tail
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[len(b) - 1 - int(rng.integers(0, min(200, len(b))))] = int(rng.integers(0, 256))
        trials.append(bytes(b))
    descs = np.zeros(len(trials), codec.cb_desc_dtype)
    pitch = 64
    off = doff = 0
    expect = []
    for i, t in enumerate(trials):
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = off, pitch, w, h
        d["K_max"], d["reversible"], d["missing_msbs"] = kmax, 1, kmax - 1
        d["num_passes"], d["len1"], d["len2"], d["data_off"] = 1, len(t), 0, doff
        ok, dec = ob.ht_decode(t, w, h, w, kmax - 1)
        expect.append((ok, ob.dequant_rev(dec, kmax) if ok else np.zeros((h, w), np.int32)))
        off += pitch * h; doff += len(t)
    coef = torch.full((off + 64,), 0x5A5A5A5A, dtype=torch.int32).cuda()
    status = codec.ht_decode(descs, np.frombuffer(b"".join(trials), np.uint8), coef)
    got = coef.cpu().numpy()
    n_ok = 0
    for i, (ok, want) in enumerate(expect):
        assert (status[i] == 0) == ok, "trial %d: GPU status %d, oracle ok=%s" % (i, status[i], ok)
        g = got[i * pitch * h:(i + 1) * pitch * h].reshape(h, pitch)[:, :w]
        assert np.array_equal(g, want), "trial %d: %d samples differ" % (i, int((g != want).sum()))
        n_ok += ok
    assert 1 <= n_ok < len(trials)


def test_ht_decode_refinement_passes_match_oracle():
    """SigProp + MagRef passes (the fourth launch, ht_dec_refine_kernel): cleanup segments followed
    by random refinement bytes, 2 and 3 passes, normal and vertically causal, all block shapes."""
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(31)
    shapes = [(64, 64)] * 4 + [(32, 32), (17, 64), (64, 17), (5, 7), (4, 1024), (1024, 4), (63, 63), (128, 32), (1, 1), (2, 3)]
    trials = []
    for it in range(84):
        w, h = shapes[it % len(shapes)]
        kmax = int(rng.integers(3, 22))
        sm, v = random_block(rng, w, h, w, kmax, float(rng.choice([0.02, 0.2, 0.6, 1.0])), int(min(2 ** kmax - 1, rng.choice([3, 40, 700]))))
        if not np.any(v[:, :w]):
            continue
        cup = ob.ht_encode(sm, w, h, w, kmax - 1, 0)
        tail = bytes(rng.integers(0, 256, size=int(rng.integers(1, 2046 if it % 9 == 0 else 300)), dtype=np.uint8))
        trials.append((w, h, kmax, cup, tail, int(rng.integers(2, 4)), bool(rng.integers(0, 2)), bool(it % 2)))
    descs = np.zeros(len(trials), codec.cb_desc_dtype)
    off = doff = 0
    datas, expect = [], []
    for i, (w, h, kmax, cup, tail, npass, causal, rev) in enumerate(trials):
        pitch = (w + 63) & ~63
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = off, pitch, w, h
        d["K_max"], d["reversible"], d["missing_msbs"] = kmax, (1 if rev else 0) | (2 if causal else 0), kmax - 1
        d["delta"] = 0.37 / (1 << 20)
        d["num_passes"], d["len1"], d["len2"], d["data_off"] = npass, len(cup), len(tail), doff
        ok, dec = ob.ht_decode(cup + tail, w, h, w, kmax - 1, len2=len(tail), num_passes=npass, stripe_causal=causal)
        assert ok
        expect.append(ob.dequant_rev(dec, kmax) if rev else ob.dequant_irv(dec, float(d["delta"])).view(np.int32))
        datas.append(np.frombuffer(cup + tail, np.uint8))
        off += pitch * h; doff += len(cup) + len(tail)
    coef = torch.full((off + 64,), 0x5A5A5A5A, dtype=torch.int32).cuda()
    status = codec.ht_decode(descs, np.concatenate(datas), coef)
    got = coef.cpu().numpy()
    assert not status.any()
    for i, (w, h, kmax, cup, tail, npass, causal, rev) in enumerate(trials):
        d = descs[i]
        g = np.lib.stride_tricks.as_strided(got[int(d["coef_off"]):], (h, w), (int(d["pitch"]) * 4, 4))
        assert np.array_equal(g, expect[i][:, :w]), "trial %d %s: %d samples differ" % (
            i, (w, h, kmax, npass, causal, rev), int((g != expect[i][:, :w]).sum()))
