"""The 64-bit sample path and the general lifting form on the CPU side: the oracle's functions against the live reference
(oracle/_ref, compiled from /root/reference by oracle/Makefile), and the oracle pipeline (oracle stages + this repo's plan
and Tier-2) against the reference's whole codestreams for samples of 27..32 bits.

Reference: ojph_encode_codeblock64 (ojph_block_encoder.cpp:1026-1520), ojph_decode_codeblock64
(ojph_block_decoder64.cpp:766-1660), param_qcd::propose_precision (ojph_params.cpp:1684-1706), the 64-bit lines of
resolution / subband / codeblock (ojph_resolution.cpp:209-229, ojph_subband.cpp:94-110, ojph_codeblock.cpp:57-266)."""
import numpy as np
import pytest

from oracle import oraclebind as ob
from tests.test_gpu_wide import SHAPES, WIDE_CASES, deep_image, wide_block


def test_general_lifting_form_equals_the_53_and_97_oracles():
    rng = np.random.default_rng(1)
    for (h, w) in [(1, 1), (1, 7), (8, 1), (5, 7), (16, 16), (33, 18), (2, 2), (3, 2), (64, 65)]:
        for xe in (True, False):
            for ye in (True, False):
                a = rng.integers(-1000, 1000, (h, w)).astype(np.int32)
                g = ob.dwt_fwd_gen(a, ob.REV53, x_even=xe, y_even=ye)
                assert all(np.array_equal(x, y) for x, y in zip(g, ob.dwt53_fwd(a, xe, ye)))
                assert np.array_equal(ob.dwt_inv_gen(*g, w, h, ob.REV53, x_even=xe, y_even=ye), a)
                a64 = a.astype(np.int64) * (1 << 20) + 12345
                g = ob.dwt_fwd_gen(a64, ob.REV53, x_even=xe, y_even=ye)
                assert np.array_equal(ob.dwt_inv_gen(*g, w, h, ob.REV53, x_even=xe, y_even=ye), a64)
                f = rng.standard_normal((h, w)).astype(np.float32)
                g = ob.dwt_fwd_gen(f, ob.IRV97, ob.K97, x_even=xe, y_even=ye)
                s = ob.dwt97_fwd(f, xe, ye)
                assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(g, s))
                r1 = ob.dwt_inv_gen(*g, w, h, ob.IRV97, ob.K97, x_even=xe, y_even=ye)
                assert np.array_equal(r1.view(np.uint32), ob.dwt97_inv(*s, w, h, xe, ye).view(np.uint32))


def test_oracle_64bit_blocks_equal_the_reference(ref):
    rng = np.random.default_rng(2)
    for i in range(60):
        w, h = SHAPES[i % len(SHAPES)]
        kmax = int(rng.integers(31, 39))
        sm, v = wide_block(rng, w, h, kmax, float(rng.choice([0.01, 0.2, 0.7, 1.0])), int(rng.choice([1, 3, 12, 30, kmax])))
        if not np.any(v):
            continue
        want = ref.encode_block64(sm, kmax - 1, w, h, w)
        assert ob.ht_encode64(sm, w, h, w, kmax - 1, 0) == want and ob.ht_encode64(sm, w, h, w, kmax - 1, 1) == want, (w, h, kmax)
        ok1, d1 = ob.ht_decode64(want, w, h, w, kmax - 1)
        ok2, d2 = ref.decode_block64(want, kmax - 1, w, h, w)
        assert ok1 and ok2 and np.array_equal(d1, d2[:, :w])


def test_oracle_64bit_decoder_on_corrupt_and_multi_pass_segments(ref):
    """the 64-bit function has byte readers of its own (one byte at a time, stuffed bits masked): same verdict, same
    samples on damaged segments; SigProp / MagRef bytes behind a cleanup segment"""
    rng = np.random.default_rng(3)
    w = h = 64; kmax = 34
    sm, v = wide_block(rng, w, h, kmax, 0.6, 30)
    good = ob.ht_encode64(sm, w, h, w, kmax - 1, 0)
    trials = [good[:2], good[:len(good) // 2], good[:-1], b"\x00\x00", b"\xff\xff\xff\xff", good + b"\x00"]
    for k in range(300):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 5))):
            pos = int(rng.integers(0, len(b))) if k % 3 else len(b) - 1 - int(rng.integers(0, min(200, len(b))))
            b[pos] = int(rng.choice([0xFF, 0x7F, 0x8F, 0x90, int(rng.integers(0, 256))]))
        trials.append(bytes(b))
    rejected = 0
    for t in trials:
        ok1, d1 = ob.ht_decode64(t, w, h, w, kmax - 1)
        ok2, d2 = ref.decode_block64(t, kmax - 1, w, h, w)
        assert ok1 == ok2
        if ok1:
            assert np.array_equal(d1, d2[:, :w])
        rejected += not ok1
    assert 0 < rejected < len(trials)
    for k in range(60):
        ww, hh = SHAPES[k % len(SHAPES)]
        km = int(rng.integers(31, 38))
        sm, v = wide_block(rng, ww, hh, km, 0.5, 20)
        if not np.any(v):
            continue
        cup = ob.ht_encode64(sm, ww, hh, ww, km - 1, 0)
        tail = bytes(rng.integers(0, 256, size=int(rng.integers(1, 300)), dtype=np.uint8))
        npass, causal = int(rng.integers(2, 4)), bool(k & 1)
        ok1, d1 = ob.ht_decode64(cup + tail, ww, hh, ww, km - 1, len2=len(tail), num_passes=npass, stripe_causal=causal)
        ok2, d2 = ref.decode_block64(cup + tail, km - 1, ww, hh, ww, len2=len(tail), num_passes=npass, stripe_causal=causal)
        assert ok1 and ok2 and np.array_equal(d1, d2[:, :ww]), (ww, hh, km, npass, causal)


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items()))
def test_deep_sample_codestreams_equal_the_reference(case, ref, refgen):
    from tests import cpu_pipeline as cp
    c = dict(case)
    nc, h, w, bd, signed = c.pop("nc"), c.pop("h"), c.pop("w"), c.pop("bd"), c.pop("signed")
    img = deep_image(nc, h, w, bd, signed)
    kw = dict(c, bit_depth=bd, is_signed=signed)
    rkw = dict(kw); rkw.pop("bit_depth")
    r = ref if kw.get("reversible", True) else refgen
    want = r.encode(img, bd, **rkw)
    got, plan, *_ = cp.encode(img, **kw)
    assert got == want, "%d vs %d bytes" % (len(got), len(want))
    dec, _ = cp.decode(want)
    rdec, _ = r.decode(want)
    assert np.array_equal(dec, rdec)
    if kw.get("reversible", True):
        assert np.array_equal(dec, img)


def test_random_deep_frames_equal_the_reference(ref):
    """a few seconds of tools/fuzz_wide_cpu.py: random 25..32-bit frames (components, signedness, colour transform, tiles, block
    sizes, progression orders) -- the reference's encoder writes the oracle pipeline's bytes and both decode them alike
    (profiles/r04_a_wide_fuzz.txt: 23 000 frames)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_wide_cpu
    assert fuzz_wide_cpu.main(seconds=5.0, seed=9) == 0


def test_damaged_blocks_decode_like_the_reference(refgen):
    """a few seconds of tools/fuzz_blocks_cpu.py: damaged cleanup segments (changed bytes, Scup, stuffing runs, cut and random
    segments; any block shape, odd widths and heights; any missing_msbs; SigProp / MagRef bytes behind them) through the oracle's
    32- and 64-bit block decoders and the live reference's generic ones: same verdict, same samples
    (profiles/r04_b_block_fuzz.txt)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_blocks_cpu
    assert fuzz_blocks_cpu.main(seconds=6.0, seed=700000) == 0


def test_gpu_block_fuzz_tool_bookkeeping():
    """tools/fuzz_blocks_gpu.py with the oracle in the device's place (FUZZ_BLOCKS_SELFTEST): descriptors, offsets and the comparison
    of the tool that is to run first on the next GPU visit"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_blocks_gpu.py"), "3"], capture_output=True, text=True,
                         env=dict(os.environ, FUZZ_BLOCKS_SELFTEST="1"), timeout=300)
    assert out.returncode == 0 and "0 decoded differently" in out.stdout, out.stdout[-400:] + out.stderr[-400:]


def test_blocks_without_a_significant_sample_are_coded_like_the_reference(refgen):
    """What codeblock::encode (ojph_codeblock.cpp:142-175) hands the block coder when only the transfer's overflow made max_val
    non-zero (a coefficient beyond K_max bits: the reversible shift puts bit K_max on the sign, the irreversible conversion
    returns 0x80000000): words that are zero or sign-only.  The oracle's coder writes the reference's bytes for them, alone and
    among ordinary samples -- the case the HIP coders' `over` path (kernels_ht_enc.hip) is compared against on the GPU."""
    from oracle import oraclebind as ob
    for (w, h) in [(64, 64), (32, 32), (5, 7), (1, 1), (17, 64), (128, 32)]:
        for fill in (0, 0x80000000):
            for mm in (3, 9):
                st = (w + 7) & ~7
                buf = np.full((h, st), fill, np.uint32)
                assert bytes(ob.ht_encode(buf, w, h, st, mm)) == bytes(refgen.encode_block(buf, mm, w, h, st))
                rng = np.random.default_rng(w * h + mm)
                for _ in range(3):
                    buf[rng.integers(0, h), rng.integers(0, w)] = (int(rng.integers(1, 4)) << (30 - mm)) | (0x80000000 if rng.random() < 0.5 else 0)
                assert bytes(ob.ht_encode(buf, w, h, st, mm)) == bytes(refgen.encode_block(buf, mm, w, h, st))


def test_64bit_decoder_with_61_missing_msbs(refgen):
    """ojph_decode_codeblock64 has no test on missing_msbs (block_decoder64.cpp:792-827): p = 62 - 61 = 1 decodes with the
    cleanup pass alone; with refinement passes (3 << (p - 2)) there is nothing to match and the oracle refuses"""
    from oracle import oraclebind as ob
    rng = np.random.default_rng(5)
    n = 0
    for it in range(60):
        w, h = [(64, 64), (32, 32), (5, 7), (3, 3), (17, 20)][it % 5]
        st, mm = (w + 7) & ~7, 61
        mag = (rng.random((h, st)) < 0.3).astype(np.uint64)
        sign = (rng.random((h, st)) < 0.5).astype(np.uint64) << np.uint64(63)
        buf = ((mag << np.uint64(62 - mm)) | (sign * mag)).astype(np.uint64)
        cs = bytes(ob.ht_encode64(buf, w, h, st, mm))
        if not cs:
            continue
        ok, d = ob.ht_decode64(cs, w, h, st, mm)
        ok2, d2 = refgen.decode_block64(cs, mm, w, h, st)
        assert ok and ok2 and np.array_equal(d[:, :w], d2[:, :w])
        assert not ob.ht_decode64(cs + b"\x00\x00", w, h, st, mm, len2=2, num_passes=2)[0]
        n += 1
    assert n > 40
