"""GPU parity of the 64-bit sample path and of the general lifting kernels.

Components that need more than 32 bits of precision (param_qcd::propose_precision, ojph_params.cpp:1684-1706: reversible
samples deeper than about 26 bits) take the reference's 64-bit functions -- ojph_encode_codeblock64
(ojph_block_encoder.cpp:1026), ojph_decode_codeblock64 (ojph_block_decoder64.cpp:766), gen_rev_vert_step64 /
gen_rev_horz_ana64 / _syn64 (ojph_transform.cpp:261,415,593), gen_rev_tx_to_cb64 / _from_cb64
(ojph_codestream_gen.cpp:81,140).  The HIP path has kernels for them (ht_encode_wide_kernel<true>, the ht_dec64_*
kernels, kernels_lift.hip); here they are compared, through the C ABI, with the oracle's 64-bit functions (pinned
against the live reference by tests/test_cpu_wide.py) stage by stage, and whole codestreams with the oracle pipeline
and -- where oracle/_ref travelled to this box -- with the reference itself: bytes identical, decode lossless."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def wide_block(rng, w, h, kmax, density, amp_bits):
    mag = (rng.integers(0, 1 << min(amp_bits, 62), size=(h, w), dtype=np.int64) * (rng.random((h, w)) < density)).astype(np.int64)
    mag = np.minimum(mag, (1 << kmax) - 1)
    sign = rng.integers(0, 2, size=(h, w)).astype(np.int64)
    v = np.where(sign == 1, -mag, mag)
    sm = (np.where(v < 0, np.uint64(1) << np.uint64(63), np.uint64(0)) | (np.abs(v).astype(np.uint64) << np.uint64(63 - kmax))).astype(np.uint64)
    return sm, v


SHAPES = [(64, 64)] * 5 + [(32, 32), (128, 32), (32, 128), (4, 1024), (1024, 4), (64, 17), (17, 64), (1, 1), (3, 3), (5, 64),
                            (2, 64), (63, 63), (33, 31), (1, 64), (64, 1)]


def _cases(rng, n):
    out = []
    for i in range(n):
        w, h = SHAPES[i % len(SHAPES)]
        kmax = int(rng.integers(31, 39))
        dens = float(rng.choice([0.0, 0.01, 0.2, 0.7, 1.0]))
        amp = int(rng.choice([1, 3, 12, 30, 34, kmax]))
        out.append((w, h, kmax, dens, min(amp, kmax)))
    return out


def test_ht_encode64_vs_oracle():
    torch = _torch()
    from openjph_amd import codec
    from openjph_amd.csrc_consts import block_scratch_bytes
    from oracle import oraclebind as ob
    rng = np.random.default_rng(64)
    cases = _cases(rng, 60)
    descs = np.zeros(len(cases), codec.cb_desc_dtype)
    coefs, expect, off, soff = [], [], 0, 0
    for i, (w, h, kmax, dens, amp) in enumerate(cases):
        pitch = (w + 63) & ~63
        sm, v = wide_block(rng, w, h, kmax, dens, amp)
        plane = np.zeros((h, pitch), np.int64); plane[:, :w] = v
        coefs.append(plane.ravel())
        mx = int(np.bitwise_or.reduce((np.abs(v).astype(np.uint64) << np.uint64(63 - kmax)).ravel())) if v.size else 0
        expect.append(ob.ht_encode64(sm, w, h, w, kmax - 1, 0) if mx >= (1 << (63 - kmax)) else b"")
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = 2 * off, pitch, w, h           # offsets count 32-bit elements
        d["K_max"], d["reversible"], d["delta"] = kmax, 1 | 4, 0.0
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
    assert not bad, "64-bit HT encode mismatches (idx, case, got_len, want_len, first_diff): %s" % bad[:8]


def test_ht_encode64_coefficients_beyond_K_max():
    """gen_rev_tx_to_cb64 (ojph_codestream_gen.cpp:81-100) on |v| >= 2^K_max: the shift drops what does not fit, bit K_max lands
    on the sign position and counts in max_val -- a block whose only non-zero word is such a sign is coded all the same
    (ojph_codeblock.cpp:161-172)"""
    torch = _torch()
    from openjph_amd import codec
    from openjph_amd.csrc_consts import block_scratch_bytes
    from oracle import oraclebind as ob
    rng = np.random.default_rng(65)
    cases = [(w, h, int(rng.integers(31, 39)), k) for k in range(3) for (w, h) in [(64, 64), (32, 32), (17, 64), (5, 7), (128, 32)]]
    descs = np.zeros(len(cases), codec.cb_desc_dtype)
    coefs, expect, off, soff = [], [], 0, 0
    for i, (w, h, kmax, kind) in enumerate(cases):
        pitch = (w + 63) & ~63
        sm, v = wide_block(rng, w, h, kmax, 0.0 if kind == 0 else 0.3, kmax)
        for _ in range(1 if kind == 0 else 4 if kind == 1 else w * h // 5):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            v[y, x] = ((1 << kmax) * int(rng.choice([1, 1, 3])) + int(rng.integers(0, 1 << 20))) * int(rng.choice([-1, 1]))
        val = (np.abs(v).astype(np.uint64) << np.uint64(63 - kmax)).astype(np.uint64)
        sm = (np.where(v < 0, np.uint64(1) << np.uint64(63), np.uint64(0)) | val).astype(np.uint64)
        plane = np.zeros((h, pitch), np.int64); plane[:, :w] = v
        coefs.append(plane.ravel())
        mx = int(np.bitwise_or.reduce(val.ravel()))
        expect.append(bytes(ob.ht_encode64(sm, w, h, w, kmax - 1, 0)) if mx >= (1 << (63 - kmax)) else b"")
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = 2 * off, pitch, w, h
        d["K_max"], d["reversible"], d["delta"] = kmax, 1 | 4, 0.0
        d["data_off"], d["scratch_cap"] = soff, block_scratch_bytes(w, h, kmax)
        off += plane.size; soff += int(d["scratch_cap"])
    coef = torch.from_numpy(np.concatenate(coefs)).cuda()
    res, out, status = codec.ht_encode(descs, coef, soff, soff)
    assert status == 0
    bad = [(i, cases[i], int(res[i, 1]), len(e)) for i, e in enumerate(expect) if out[int(res[i, 0]):int(res[i, 0]) + int(res[i, 1])].tobytes() != e]
    assert not bad, bad[:6]


def _decode64_expect(ob, coded, w, h, kmax, len2=0, npass=1, causal=False):
    ok, dec = ob.ht_decode64(coded, w, h, w, kmax - 1, len2=len2, num_passes=npass, stripe_causal=causal)
    if not ok:
        return False, np.zeros((h, w), np.int64)
    dq = np.empty(dec.shape, np.int64)
    ob.lib().ojo_dequant_rev64(np.ascontiguousarray(dec).ctypes.data, dq.ctypes.data, dec.size, kmax)
    return True, dq[:, :w]


def test_ht_decode64_vs_oracle_incl_corrupt_and_refinement():
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(65)
    trials = []                                            # (w, h, kmax, bytes, len2, passes, causal)
    for (w, h, kmax, dens, amp) in _cases(rng, 50):
        sm, v = wide_block(rng, w, h, kmax, dens, amp)
        if not np.any(v):
            trials.append((w, h, kmax, b"", 0, 0, False))
            continue
        trials.append((w, h, kmax, ob.ht_encode64(sm, w, h, w, kmax - 1, 0), 0, 1, False))
    # corrupted cleanup segments of one block: the 64-bit decoder's own byte readers decide what comes out
    w = h = 64; kmax = 34
    sm, v = wide_block(rng, w, h, kmax, 0.6, 30)
    good = ob.ht_encode64(sm, w, h, w, kmax - 1, 0)
    for t in (good[:2], good[:len(good) // 2], good[:-1], b"\x00\x00", b"\xff\xff\xff\xff", good + b"\x00"):
        trials.append((w, h, kmax, t, 0, 1, False))
    for k in range(110):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 5))):
            pos = int(rng.integers(0, len(b))) if k % 3 else len(b) - 1 - int(rng.integers(0, min(200, len(b))))
            b[pos] = int(rng.choice([0xFF, 0x7F, 0x8F, 0x90, int(rng.integers(0, 256))]))
        trials.append((w, h, kmax, bytes(b), 0, 1, False))
    # refinement passes behind cleanup segments
    for k in range(24):
        ww, hh = SHAPES[k % len(SHAPES)]
        km = int(rng.integers(31, 38))
        sm, v = wide_block(rng, ww, hh, km, 0.5, 20)
        if not np.any(v):
            continue
        cup = ob.ht_encode64(sm, ww, hh, ww, km - 1, 0)
        tail = bytes(rng.integers(0, 256, size=int(rng.integers(1, 300)), dtype=np.uint8))
        trials.append((ww, hh, km, cup + tail, len(tail), int(rng.integers(2, 4)), bool(k & 1)))
    descs = np.zeros(len(trials), codec.cb_desc_dtype)
    off = doff = 0
    expect = []
    for i, (w, h, kmax, t, len2, npass, causal) in enumerate(trials):
        pitch = (w + 63) & ~63
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = 2 * off, pitch, w, h
        d["K_max"], d["reversible"], d["missing_msbs"] = kmax, 1 | 4 | (2 if causal else 0), kmax - 1
        d["num_passes"], d["len1"], d["len2"], d["data_off"] = npass, len(t) - len2, len2, doff
        expect.append(_decode64_expect(ob, t, w, h, kmax, len2, npass, causal) if t else (True, np.zeros((h, w), np.int64)))
        off += pitch * h; doff += len(t)
    coef = torch.full((off + 64,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64).cuda()
    status = codec.ht_decode(descs, np.frombuffer(b"".join(t[3] for t in trials), np.uint8), coef)
    got = coef.cpu().numpy()
    n_rej = 0
    for i, (ok, want) in enumerate(expect):
        w, h = trials[i][0], trials[i][1]
        d = descs[i]
        assert (status[i] == 0) == ok, "trial %d %s: GPU status %d, oracle ok=%s" % (i, trials[i][:3], status[i], ok)
        g = np.lib.stride_tricks.as_strided(got[int(d["coef_off"]) // 2:], (h, w), (int(d["pitch"]) * 8, 8))
        assert np.array_equal(g, want), "trial %d %s: %d samples differ" % (i, trials[i][:3] + trials[i][4:], int((g != want).sum()))
        n_rej += not ok
    assert n_rej >= 1


def test_ht_decode64_with_61_missing_msbs():
    """p = 62 - 61 = 1: ojph_decode_codeblock64 has no test on missing_msbs (block_decoder64.cpp:792-827) and decodes the
    cleanup pass; with refinement passes behind it (3 << (p - 2)) the block is refused here and by the oracle
    (tests/test_cpu_wide.py::test_64bit_decoder_with_61_missing_msbs pins the oracle to the reference)"""
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(61)
    trials = []
    for it in range(24):
        w, h = [(64, 64), (32, 32), (5, 7), (3, 3), (17, 20), (128, 32)][it % 6]
        st, mm = (w + 7) & ~7, 61
        mag = (rng.random((h, st)) < 0.3).astype(np.uint64)
        sign = (rng.random((h, st)) < 0.5).astype(np.uint64) << np.uint64(63)
        buf = ((mag << np.uint64(1)) | (sign * mag)).astype(np.uint64)
        cs = bytes(ob.ht_encode64(buf, w, h, st, mm))
        if cs:
            trials.append((w, h, cs, 0, 1))
            if it % 4 == 0:
                trials.append((w, h, cs + b"\x12\x34", 2, 2))     # refinement bytes behind it: refused
    descs = np.zeros(len(trials), codec.cb_desc_dtype)
    off = doff = 0
    expect = []
    for i, (w, h, t, len2, npass) in enumerate(trials):
        pitch = (w + 63) & ~63
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = 2 * off, pitch, w, h
        d["K_max"], d["reversible"], d["missing_msbs"] = 62, 1 | 4, 61
        d["num_passes"], d["len1"], d["len2"], d["data_off"] = npass, len(t) - len2, len2, doff
        ok, dec = ob.ht_decode64(t, w, h, (w + 7) & ~7, 61, len2=len2, num_passes=npass)
        dq = np.zeros((h, w), np.int64)
        if ok:
            dec = np.ascontiguousarray(dec[:, :w])
            ob.lib().ojo_dequant_rev64(dec.ctypes.data, dq.ctypes.data, dec.size, 62)
        expect.append((ok, dq))
        off += pitch * h; doff += len(t)
    coef = torch.full((off + 64,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64).cuda()
    status = codec.ht_decode(descs, np.frombuffer(b"".join(t[2] for t in trials), np.uint8), coef)
    got = coef.cpu().numpy()
    for i, (ok, want) in enumerate(expect):
        w, h = trials[i][0], trials[i][1]
        d = descs[i]
        assert (status[i] == 0) == ok == (trials[i][4] == 1), (i, int(status[i]), ok)
        g = np.lib.stride_tricks.as_strided(got[int(d["coef_off"]) // 2:], (h, w), (int(d["pitch"]) * 8, 8))
        assert np.array_equal(g, want), "trial %d: %d samples differ" % (i, int((g != want).sum()))
        assert not ok or np.any(want)


def _layout(shapes, elems):
    offs, total = [], 0
    for (h, w) in shapes:
        pitch = (max(w, 1) + 63) & ~63
        offs.append((total, pitch))
        total += (pitch * max(h, 1) + 64) * elems
        total = (total + 63) & ~63
    return offs, total


KERNELS = [
    ("rev53-int64", np.int64, [(1, 2, 2), (-1, 1, 1)], 1.0),
    ("rev53-int32", np.int32, [(1, 2, 2), (-1, 1, 1)], 1.0),
    ("rev-3steps", np.int32, [(1, 2, 2), (-1, 1, 1), (3, 4, 3)], 1.0),
    ("rev-general-a", np.int64, [(-3, 8, 4), (5, 16, 5), (1, 1, 1), (-1, 2, 2)], 1.0),
    ("irv97", np.float32, [0.443506852043971, 0.882911075530934, -0.052980118572961, -1.586134342059924], 1.230174104914001),
    ("irv-2steps", np.float32, [0.25, -0.5], 1.41421356),
    ("irv-3steps", np.float32, [0.2, -0.4, 0.1], 1.1),
    # one to four steps that transform both directions take the register pipeline (kernels_dwt.hip: WvGen), the rest the
    # element-wise launches (kernels_lift.hip): every step count of the one, a kernel beyond it for the other
    ("rev-1step", np.int32, [(1, 1, 1)], 1.0),
    ("rev-3steps-int64", np.int64, [(-1, 1, 1), (3, 4, 3), (1, 0, 2)], 1.0),
    ("irv-1step", np.float32, [0.3], 0.9),
    ("rev-5steps", np.int32, [(1, 2, 2), (-1, 1, 1), (1, 4, 3), (-1, 2, 2), (1, 1, 1)], 1.0),
    ("irv-5steps", np.float32, [0.1, -0.2, 0.3, -0.15, 0.05], 1.05),
]


@pytest.mark.parametrize("name,dt,steps,K", KERNELS, ids=[k[0] for k in KERNELS])
@pytest.mark.parametrize("horz,vert", [(True, True), (True, False), (False, True)], ids=["bidir", "horz-only", "vert-only"])
def test_general_lifting_vs_oracle(name, dt, steps, K, horz, vert):
    """one level in the general form -- any lifting kernel, one direction only, int32 / int64 / float -- equals the
    oracle's general form value for value (itself equal to the 5/3 and 9/7 oracles where they apply, and to the
    reference's decoder on Part-2 codestreams: tests/test_cpu_part2.py)"""
    torch = _torch()
    from openjph_amd import codec
    from oracle import oraclebind as ob
    rng = np.random.default_rng(77)
    cases = [(64, 64, 1, 1), (130, 257, 1, 1), (257, 130, 0, 1), (131, 77, 1, 0), (96, 200, 0, 0), (1, 37, 1, 1), (1, 37, 0, 1),
             (40, 1, 1, 1), (40, 1, 1, 0), (2, 2, 1, 1), (3, 5, 0, 0), (300, 121, 1, 1), (1, 1, 0, 0), (1, 1, 1, 1)]
    elems = np.dtype(dt).itemsize // 4
    descs = np.zeros(len(cases), codec.dwt_desc_dtype)
    shapes = []
    for (h, w, xe, ye) in cases:
        lw, hw, lh, hh = ob.band_dims(w, h, bool(xe), bool(ye))
        if not horz:
            lw, hw = w, 0
        if not vert:
            lh, hh = h, 0
        shapes += [(h, w), (lh, lw), (lh, hw), (hh, lw), (hh, hw)]
    offs, total = _layout(shapes, elems)
    arena = np.zeros(total, np.uint32)

    def view(o, hh_, ww_):
        a = arena.view(dt)
        return np.lib.stride_tricks.as_strided(a[o[0] // elems:], (hh_, ww_), (o[1] * dt().itemsize, dt().itemsize))

    inputs, expect = [], []
    for i, (h, w, xe, ye) in enumerate(cases):
        if dt == np.float32:
            src = (rng.random((h, w)) - 0.5).astype(np.float32)
        else:
            src = rng.integers(-(1 << (40 if dt == np.int64 else 20)), 1 << (40 if dt == np.int64 else 20), size=(h, w)).astype(dt)
        inputs.append(src)
        o = offs[5 * i:5 * i + 5]
        d = descs[i]
        d["src_off"], d["src_pitch"] = o[0]
        for k, nm in enumerate(("ll", "hl", "lh", "hh")):
            d[nm + "_off"], d[nm + "_pitch"] = o[1 + k]
        d["w"], d["h"], d["x_even"], d["y_even"] = w, h, xe, ye
        view(o[0], h, w)[:] = src
        expect.append(ob.dwt_fwd_gen(src, steps, K, horz, vert, bool(xe), bool(ye)))
    d_arena = torch.from_numpy(arena.view(np.int32)).cuda()
    max_w = max(c[1] for c in cases); max_h = max(c[0] for c in cases)
    elem = {np.int32: 0, np.int64: 1, np.float32: 2}[dt]
    codec.dwt_general("forward", steps, elem, descs, d_arena, max_w, max_h, K, horz, vert)
    arena = d_arena.cpu().numpy().view(np.uint32).copy()
    for i, (h, w, xe, ye) in enumerate(cases):
        o = offs[5 * i:5 * i + 5]
        for k, nm in enumerate(("ll", "hl", "lh", "hh")):
            e = expect[i][k]
            if e.size == 0:
                continue
            g = view(o[1 + k], e.shape[0], e.shape[1])
            assert np.array_equal(g.view(np.uint8), np.ascontiguousarray(e).view(np.uint8).reshape(g.view(np.uint8).shape)), \
                "forward case %d %s band %s" % (i, cases[i], nm)
    for i, (h, w, xe, ye) in enumerate(cases):
        view(offs[5 * i], h, w)[:] = 0
    d_arena = torch.from_numpy(arena.view(np.int32)).cuda()
    codec.dwt_general("inverse", steps, elem, descs, d_arena, max_w, max_h, K, horz, vert)
    arena = d_arena.cpu().numpy().view(np.uint32).copy()
    for i, (h, w, xe, ye) in enumerate(cases):
        g = view(offs[5 * i], h, w)
        e = ob.dwt_inv_gen(*expect[i], w, h, steps, K, horz, vert, bool(xe), bool(ye))
        assert np.array_equal(g.view(np.uint8), e.view(np.uint8).reshape(g.view(np.uint8).shape)), "inverse case %d %s" % (i, cases[i])
        if dt != np.float32 and len(steps) % 2 == 0:
            assert np.array_equal(g, inputs[i])


def deep_image(nc, h, w, bd, signed, seed=5):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 17.0) + np.cos(yy / 11.0)) * 0.2 + 0.5
    out = []
    for c in range(nc):
        v = base * (2.0 ** bd - 1) * 0.9 + rng.normal(0, 2.0 ** (bd - 6), (h, w))
        v = np.clip(v, 0, 2.0 ** bd - 1).astype(np.int64)
        if signed:
            v -= 1 << (bd - 1)
        out.append(v.astype(np.uint64).astype(np.uint32).astype(np.int32))          # the si32 container
    return np.stack(out)


WIDE_CASES = [
    dict(nc=1, h=96, w=130, bd=32, signed=False),
    dict(nc=1, h=200, w=300, bd=31, signed=False, num_decomps=3, block=(32, 32)),
    dict(nc=3, h=70, w=90, bd=30, signed=True, color_transform=True),
    dict(nc=3, h=140, w=190, bd=29, signed=False, color_transform=True, tile=(128, 64)),
    dict(nc=2, h=64, w=64, bd=32, signed=True, num_decomps=1),
    dict(nc=1, h=260, w=200, bd=30, signed=False, block=(128, 32), prog_order="CPRL"),
    dict(nc=1, h=128, w=128, bd=27, signed=False),           # deeper than 26 bits, still the 32-bit path
    dict(nc=3, h=100, w=120, bd=28, signed=False, color_transform=True),
    dict(nc=1, h=90, w=110, bd=28, signed=True, reversible=False, qstep=0.00001),
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items()))
def test_deep_samples_end_to_end(case):
    """frames of 27..32-bit samples: codestream byte-identical to the oracle pipeline's (and the reference's), decode
    equal to the oracle's, lossless for the reversible ones"""
    from openjph_amd import codec
    from tests import cpu_pipeline as cp
    c = dict(case)
    nc, h, w, bd, signed = c.pop("nc"), c.pop("h"), c.pop("w"), c.pop("bd"), c.pop("signed")
    img = deep_image(nc, h, w, bd, signed)
    kw = dict(c, bit_depth=bd, is_signed=signed)
    got = codec.encode(img, **kw)
    want, plan, *_ = cp.encode(img, **kw)
    if got != want:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        pytest.fail("codestream differs: %d vs %d bytes, first difference at %d" % (len(got), len(want), first))
    from oracle import refbind
    generic = not kw.get("reversible", True)               # the 9/7 pin is the reference's generic build
    if refbind.available(generic=generic):
        rkw = dict(kw); rkw.pop("bit_depth")
        assert refbind.Ref(generic=generic).encode(img, bd, **rkw) == got
    dec = codec.decode(want)
    want_dec, _ = cp.decode(want)
    assert np.array_equal(dec, want_dec)
    if kw.get("reversible", True):
        assert np.array_equal(dec, img)


def test_general_lifting_elementwise_form_too():
    """the same cases with OJPHGPU_LIFT_ELEMENTWISE=1: every level through the element-wise launches of kernels_lift.hip (the
    form kernels of five and more steps always take) instead of the register pipeline -- the two forms are pinned to the same
    oracle, so to each other.  The switch is read once per process: a child process runs the test above."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OJPHGPU_LIFT_ELEMENTWISE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_wide.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_general_lifting_vs_oracle", "-p", "no:cacheprovider"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0 and ("%d passed" % (3 * len(KERNELS))).encode() in r.stdout, r.stdout[-2000:]
