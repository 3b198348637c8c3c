"""End-to-end GPU parity through the whole-frame C ABI (ojphgpu_encoder_* / ojphgpu_decoder_*):
the emitted codestream must be byte-identical to the oracle-built one (which is pinned
byte-for-byte against the real reference by tests/test_cpu_parity.py), and -- when
oracle/_ref/*.so travelled to this box -- to the reference's own output."""
import os

import numpy as np
import pytest

from tests.synth import synth_image, c1_image

pytestmark = pytest.mark.gpu

CASES = [
    dict(nc=1, h=256, w=256, bd=8),
    dict(nc=3, h=200, w=300, bd=8, color_transform=True),
    dict(nc=3, h=131, w=257, bd=10),
    dict(nc=1, h=517, w=389, bd=12, num_decomps=3),
    dict(nc=1, h=300, w=500, bd=16, tile=(128, 128)),
    dict(nc=3, h=260, w=260, bd=8, prog_order="CPRL", precinct=(128, 128)),
    dict(nc=1, h=64, w=1, bd=8),
    dict(nc=1, h=1, w=64, bd=8),
    dict(nc=1, h=5, w=7, bd=8),
    dict(nc=1, h=200, w=200, bd=8, block=(128, 32)),
    dict(nc=1, h=200, w=200, bd=8, block=(4, 1024)),
    # blocks of 65..128 columns, none wider: the narrow encoder's 32-pairs-by-2-rows layout (LOGP 5), step 1's 128-bit masks
    dict(nc=1, h=333, w=517, bd=10, block=(128, 32), num_decomps=2),
    dict(nc=3, h=203, w=395, bd=12, block=(128, 8), reversible=False, qstep=0.002, num_decomps=1),
    dict(nc=1, h=97, w=260, bd=16, block=(128, 16), num_decomps=0),
    dict(nc=2, h=70, w=300, bd=8, block=(128, 4), num_decomps=1, tile=(200, 64)),
    dict(nc=1, h=256, w=256, bd=8, signed=True),
    dict(nc=1, h=256, w=256, bd=8, num_decomps=0),
    dict(nc=1, h=256, w=256, bd=12, reversible=False),
    dict(nc=3, h=200, w=300, bd=8, reversible=False, color_transform=True),
    dict(nc=3, h=240, w=320, bd=12, reversible=False, qstep=0.001),
    dict(nc=1, h=300, w=500, bd=10, reversible=False, tile=(128, 128), qstep=0.01),
]


def _split(case):
    c = dict(case)
    nc, h, w, bd = c.pop("nc"), c.pop("h"), c.pop("w"), c.pop("bd")
    signed = c.pop("signed", False)
    return nc, h, w, bd, signed, c


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items()))
def test_encode_decode_matches_oracle(case):
    from openjph_amd import codec
    from tests import cpu_pipeline as cp
    nc, h, w, bd, signed, kw = _split(case)
    img = synth_image(nc, h, w, bd, seed=3, signed=signed)
    kw = dict(kw, bit_depth=bd, is_signed=signed)
    got = codec.encode(img, **kw)
    want, plan, arena, data, coded = cp.encode(img, **kw)
    if got != want:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        pytest.fail("codestream differs: %d vs %d bytes, first difference at %d" % (len(got), len(want), first))
    dec = codec.decode(want)
    want_dec, _ = cp.decode(want)
    assert np.array_equal(dec, want_dec), "decode differs in %d samples, max %d" % (
        int((dec != want_dec).sum()), int(np.abs(dec - want_dec).max()))
    if kw.get("reversible", True):
        assert np.array_equal(dec, img)


def _sparse_plane(kind, h, w, bd, seed):
    """planes whose code-blocks (no decomposition: the samples are the blocks') exercise the MEL coder's regimes: long runs
    of context-0 quads without a significant sample (the coder climbs to k = 12 and stays), isolated significant samples at
    fixed and at random distances (k going up and down), both mixed with dense rows"""
    rng = np.random.default_rng(seed)
    mid = 1 << (bd - 1)
    img = np.full((1, h, w), mid, np.int64)
    if kind == "one-late":                 # one sample at the very end of every block's last row
        img[0, 63::64, 63::64] += 5
    elif kind.startswith("every"):         # a significant sample every n-th quad of the raster
        n = int(kind[5:])
        q = np.arange((h // 2) * (w // 2))
        ys, xs = np.divmod(q[::n], w // 2)
        img[0, 2 * ys, 2 * xs] += rng.integers(1, 40, ys.size) * rng.choice([-1, 1], ys.size)
    elif kind.startswith("p"):             # significant with probability 1 / n
        n = int(kind[1:])
        m = rng.random((h, w)) < 1.0 / n
        img[0][m] += (rng.integers(1, 1 << (bd - 2), int(m.sum())) * rng.choice([-1, 1], int(m.sum())))
    elif kind == "bands":                  # dense rows between long empty stretches
        for y0 in range(0, h, 24):
            img[0, y0:y0 + 2] += rng.integers(-200, 200, (2, w))
    return np.clip(img, 1, (1 << bd) - 1)      # (not the most negative value: without a decomposition its magnitude has K_max + 1 bits)


@pytest.mark.parametrize("kind", ["one-late", "every1", "every2", "every3", "every7", "every33", "every100", "every1000",
                                  "p4", "p20", "p150", "p2000", "bands"])
def test_encoder_mel_regimes(kind):
    """(kernels_ht_enc.hip: mel_at, the event walk; ojph_block_encoder.cpp:317-362)"""
    from openjph_amd import codec
    from tests import cpu_pipeline as cp
    for bd, h, w, block in ((10, 128, 192, (64, 64)), (8, 70, 130, (32, 32)), (12, 64, 256, (128, 32))):
        img = _sparse_plane(kind, h, w, bd, seed=len(kind) + bd)
        kw = dict(bit_depth=bd, num_decomps=0, block=block)
        got = codec.encode(img, **kw)
        want = cp.encode(img, **kw)[0]
        assert got == want, "%s, %d-bit %dx%d: codestream differs (%d vs %d bytes)" % (kind, bd, w, h, len(got), len(want))
        assert np.array_equal(codec.decode(got), img)


@pytest.mark.parametrize("i", range(9), ids=lambda i: "grid%d" % i)
def test_grid_encode_decode_matches_oracle(i):
    """sub-sampled components, image and tile offsets (tests/golden_cases.py GRID_CASES): the GPU
    codec against the oracle pipeline and against the stored digests of the reference's codestreams"""
    import hashlib
    import json
    import os
    from openjph_amd import codec
    from openjph_amd.plan import make_params
    from tests import cpu_pipeline as cp
    from tests.golden_cases import GRID_CASES, grid_kwargs
    planes, kw, size = grid_kwargs(GRID_CASES[i])
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["grid"][i]
    enc = codec.Encoder(make_params(size[0], size[1], len(planes), **kw))
    got = enc.encode(planes)
    want, plan, *_ = cp.encode(planes, size=size, **kw)
    assert got == want
    assert hashlib.sha256(got).hexdigest() == gold["sha256"]
    dec = codec.Decoder(want)
    out = plan.unpack_frame(dec.decode())
    want_dec, _ = cp.decode(want)
    for c in range(len(planes)):
        assert np.array_equal(out[c], want_dec[c]), "component %d differs" % c
    assert hashlib.sha256(b"".join(np.ascontiguousarray(q, dtype=np.int32).tobytes() for q in out)).hexdigest() == gold["dec_sha256"]


@pytest.mark.parametrize("i", range(15), ids=lambda i: "skip%d" % i)
def test_reduced_resolution_decode_matches_oracle(i):
    """codestream::restrict_input_resolution on the GPU decoder against the oracle pipeline and the
    stored digests of the reference's output (tests/golden_cases.py SKIP_CASES)"""
    import hashlib
    import json
    import os
    from openjph_amd import codec
    from tests import cpu_pipeline as cp
    from tests.golden_cases import skip_case
    planes, kw, size, skip = skip_case(i)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["skip"][i]
    cs, *_ = cp.encode(planes if size else np.stack(planes), **(dict(kw, size=size) if size else kw))
    dec = codec.Decoder(cs, skip_res=skip)
    out = dec.plan.unpack_frame(dec.decode())
    assert [list(q.shape) for q in out] == gold["shapes"]
    want, _ = cp.decode(cs, skip=skip)
    for c in range(len(planes)):
        assert np.array_equal(out[c], want[c]), "component %d differs" % c
    assert hashlib.sha256(b"".join(np.ascontiguousarray(q, dtype=np.int32).tobytes() for q in out)).hexdigest() == gold["dec_sha256"]


@pytest.mark.parametrize("i", range(9), ids=lambda i: "fmt%d" % i)
def test_component_formats_and_qfactor_on_gpu(i):
    """per-component bit depth / signedness (the conversion kernels take the format from the
    descriptor) and qfactor quantisation: GPU codec == oracle pipeline == reference digests"""
    import hashlib
    import json
    import os
    from openjph_amd import codec
    from openjph_amd.plan import make_params
    from tests import cpu_pipeline as cp
    from tests.golden_cases import format_case
    planes, kw, size = format_case(i)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["formats"][i]
    got = codec.Encoder(make_params(size[0], size[1], len(planes), **kw)).encode(planes)
    assert hashlib.sha256(got).hexdigest() == gold["sha256"]
    dec = codec.Decoder(got)
    out = dec.plan.unpack_frame(dec.decode())
    assert hashlib.sha256(b"".join(np.ascontiguousarray(q, dtype=np.int32).tobytes() for q in out)).hexdigest() == gold["dec_sha256"]


@pytest.mark.parametrize("i", range(15), ids=lambda i: "coc%d" % i)
def test_component_coding_styles_on_gpu(i):
    """COC marker segments: components with their own decompositions / block size / precincts /
    wavelet (the DWT launches split by depth below each component's top and by wavelet, the block
    coder and the conversion take the wavelet from their descriptors): GPU codec == reference
    digests, also at reduced resolution.  Case 0 is the reference's tests/test_mixed_coc.cpp."""
    import hashlib
    import json
    import os
    from openjph_amd import codec
    from openjph_amd.plan import make_params
    from tests.golden_cases import coc_case
    planes, kw, size, skip, resilient = coc_case(i)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["coc"][i]
    got = codec.Encoder(make_params(size[0], size[1], len(planes), **kw)).encode(planes)
    assert hashlib.sha256(got).hexdigest() == gold["sha256"]
    dec = codec.Decoder(got, resilient=resilient, skip_res=skip)
    out = dec.plan.unpack_frame(dec.decode())
    assert [list(q.shape) for q in out] == gold["shapes"]
    assert hashlib.sha256(b"".join(np.ascontiguousarray(q, dtype=np.int32).tobytes() for q in out)).hexdigest() == gold["dec_sha256"]


@pytest.mark.parametrize("i", range(5), ids=lambda i: "cqf%d" % i)
def test_component_quality_factors_on_gpu(i):
    """quality factors of single components (a QCC each): GPU codec == reference digests"""
    import hashlib
    import json
    import os
    from openjph_amd import codec
    from openjph_amd.plan import make_params
    from tests.golden_cases import cqf_case
    img, kw = cqf_case(i)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["cqf"][i]
    got = codec.Encoder(make_params(img.shape[2], img.shape[1], img.shape[0], **kw)).encode(img)
    assert hashlib.sha256(got).hexdigest() == gold["sha256"]
    out = codec.Decoder(got).decode()
    assert hashlib.sha256(np.ascontiguousarray(out, dtype=np.int32).tobytes()).hexdigest() == gold["dec_sha256"]


@pytest.mark.parametrize("i", range(8), ids=lambda i: "nlt%d" % i)
def test_nonlinearity_type3_on_gpu(i):
    """NLT type 3 on signed components (conversion kernels, descriptor bit 0x800; the fused
    conversion of the DWT's top level steps aside): GPU codec == reference digests, with int32 and
    with 16-bit sample containers"""
    import hashlib
    import json
    import os
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import make_params
    from tests.golden_cases import nlt_case
    planes, kw, size = nlt_case(i)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["nlt"][i]
    enc = codec.Encoder(make_params(size[0], size[1], len(planes), **kw))
    got = enc.encode(planes)
    assert hashlib.sha256(got).hexdigest() == gold["sha256"]
    dec = codec.Decoder(got)
    out = dec.plan.unpack_frame(dec.decode())
    assert hashlib.sha256(b"".join(np.ascontiguousarray(q, dtype=np.int32).tobytes() for q in out)).hexdigest() == gold["dec_sha256"]
    same_shape = len({q.shape for q in planes}) == 1
    if same_shape and all(bd <= 16 for bd in kw["bit_depths"]) and len(set(kw["signs"])) == 1:   # 16-bit containers, one dtype per frame
        signed = kw["signs"][0]
        img = np.stack(planes)
        small = img.astype(np.int16) if signed else img.astype(np.uint16)
        assert enc.encode(small) == got
        out16 = dec.run_device(dtype=torch.int16).cpu().numpy()
        back = out16.astype(np.int32) if signed else out16.view(np.uint16).astype(np.int32)
        assert np.array_equal(back, np.stack(out))


@pytest.mark.parametrize("chunk", range(2))
def test_random_coc_parameter_sets_on_gpu(chunk):
    """seeded random parameter sets with COC marker segments on random components
    (tests/random_cases.py: random_coc_case): GPU codec == oracle pipeline"""
    from openjph_amd import capi, codec
    from openjph_amd.plan import make_params
    from tests import cpu_pipeline as cp
    from tests.random_cases import random_coc_case
    done = 0
    for seed in range(chunk * 40, chunk * 40 + 40):
        planes, kw, size = random_coc_case(seed)
        if any(q.size == 0 for q in planes):
            continue
        try:
            want, plan, *_ = cp.encode(planes, size=size, **kw)
        except capi.OjphError:
            continue                                      # a parameter set the reference rejects, too
        got = codec.Encoder(make_params(size[0], size[1], len(planes), **kw)).encode(planes)
        assert got == want, "seed %d: %s" % (seed, kw)
        try:
            wdec, _ = cp.decode(want)
        except capi.OjphError:
            continue                                      # tile-part numbers with gaps
        dec = codec.Decoder(want)
        out = dec.plan.unpack_frame(dec.decode())
        for c in range(len(planes)):
            assert np.array_equal(out[c], wdec[c]), "seed %d component %d: %s" % (seed, c, kw)
        done += 1
    assert done >= 20


@pytest.mark.parametrize("chunk", range(4))
def test_random_parameter_sets_on_gpu(chunk):
    """the seeded random parameter sets of tests/random_cases.py (odd sizes, offsets, sub-sampling,
    mixed formats, every progression order, tile-parts): GPU codec == oracle pipeline, byte for byte
    and sample for sample"""
    from openjph_amd import capi, codec
    from openjph_amd.plan import make_params
    from tests import cpu_pipeline as cp
    from tests.random_cases import random_case
    done = 0
    for seed in range(chunk * 30, chunk * 30 + 30):
        planes, kw, size = random_case(seed)
        if any(q.size == 0 for q in planes):
            continue
        try:
            want, plan, *_ = cp.encode(planes, size=size, **kw)
        except capi.OjphError:
            continue                                      # a parameter set the reference rejects, too
        got = codec.Encoder(make_params(size[0], size[1], len(planes), **kw)).encode(planes)
        assert got == want, "seed %d: %s" % (seed, kw)
        dec = codec.Decoder(want)
        out = dec.plan.unpack_frame(dec.decode())
        wdec, _ = cp.decode(want)
        for c in range(len(planes)):
            assert np.array_equal(out[c], wdec[c]), "seed %d component %d: %s" % (seed, c, kw)
        done += 1
    assert done >= 20


@pytest.mark.parametrize("case", [
    dict(nc=3, h=131, w=257, bd=10), dict(nc=1, h=300, w=501, bd=16, tile=(128, 128)), dict(nc=3, h=200, w=300, bd=8, color_transform=True),
    dict(nc=1, h=97, w=113, bd=12, signed=True, num_decomps=2), dict(nc=3, h=240, w=321, bd=12, reversible=False, qstep=0.001),
    dict(nc=3, h=200, w=300, bd=8, reversible=False, color_transform=True), dict(nc=1, h=64, w=1, bd=8), dict(nc=1, h=5, w=7, bd=8, num_decomps=0),
], ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items()))
def test_16bit_sample_containers(case):
    """the frame handed over as int16 / uint16 planes (ojphgpu_encoder_run_device16 /
    ojphgpu_decoder_run_device16): same codestream bytes and the same samples as with int32 planes"""
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import make_params
    nc, h, w, bd, signed, kw = _split(case)
    img = synth_image(nc, h, w, bd, seed=3, signed=signed)
    kw = dict(kw, bit_depth=bd, is_signed=signed)
    enc = codec.Encoder(make_params(w, h, nc, **kw))
    want = enc.encode(img)
    small = img.astype(np.int16) if signed else img.astype(np.uint16)
    assert enc.encode(small) == want
    dec = codec.Decoder(want)
    ref_out = dec.decode()
    out16 = dec.run_device(dtype=torch.int16).cpu().numpy()
    got = out16.astype(np.int32) if signed else out16.view(np.uint16).astype(np.int32)
    assert np.array_equal(got, ref_out)


def test_corrupted_block_bytes_decode_like_oracle():
    """bit flips and overwritten bytes inside the code-block bodies (the parser does not see them): the
    GPU block decoder must neither hang nor read out of bounds, refuse exactly the blocks the reference
    algorithm refuses (they stay zero when resilient) and produce the very same -- wrong -- samples for
    the ones it accepts"""
    import random
    from openjph_amd import capi, codec
    from openjph_amd.plan import parse_codestream
    from tests import cpu_pipeline as cp
    img = synth_image(3, 96, 128, 10, seed=9)
    rng = random.Random(3)
    checked = refused_blocks = 0
    for kw in (dict(num_decomps=3), dict(reversible=False, qstep=0.01, num_decomps=2, block=(32, 32))):
        cs = bytes(cp.encode(img, bit_depth=10, **kw)[0])
        body0 = cs.find(b"\xff\x93") + 2
        for _ in range(12):
            b = bytearray(cs)
            for _ in range(rng.randint(1, 6)):
                pos = rng.randrange(body0, len(b) - 2)
                if rng.random() < 0.6:
                    b[pos] ^= 1 << rng.randrange(8)
                else:
                    b[pos] = rng.choice([0xFF, 0x00, 0x7F, rng.randrange(256)])
            b = bytes(b)
            try:
                pl = parse_codestream(b, resilient=True)
            except capi.OjphError:
                continue
            want = cp.inverse_stages(pl, cp.decode_blocks(pl, b, resilient=True))
            dec = codec.Decoder(b, resilient=True)
            got = dec.decode()
            assert np.array_equal(got, want)
            refused_blocks += dec.failed_blocks()
            checked += 1
    assert checked >= 12


def test_truncated_codestream_decodes_like_oracle():
    """tests/test_truncated_decode.cpp on the GPU decoder: a full frame from whatever was received
    when resilient, an error for a cut the parser detects when not"""
    from openjph_amd import capi, codec
    from openjph_amd.plan import parse_codestream
    from tests import cpu_pipeline as cp
    y, x = np.mgrid[0:256, 0:256]
    img = ((x * 7 + y * 13 + ((x * y) >> 3)) & 0xFF).astype(np.int32)[None]
    cs, *_ = cp.encode(img, bit_depth=8, num_decomps=5)
    detected = 0
    for cut in range(1, 16):
        part = cs[:len(cs) * cut // 16]
        got = codec.Decoder(part, resilient=True).decode()
        pl = parse_codestream(part, resilient=True)
        want = cp.inverse_stages(pl, cp.decode_blocks(pl, part))
        assert got.shape == (1, 256, 256) and np.array_equal(got, want), "cut %d" % cut
        try:
            again = codec.Decoder(part).decode()
            assert np.array_equal(again, want)          # an undetectable cut decodes the same way in both modes
        except capi.OjphError:
            detected += 1
    assert 0 < detected < 15


def test_c1_matches_reference_bytes(ref):
    """BASELINE config #1 (256x256 8-bit, 5/3): identical bytes to the reference library."""
    from openjph_amd import codec
    img = c1_image()
    got = codec.encode(img, bit_depth=8)
    want = ref.encode(img, 8)
    assert len(want) == 54702
    assert got == want
    back, _ = ref.decode(got)
    assert np.array_equal(back, img)
    assert np.array_equal(codec.decode(want), img)


def test_4k_rgb_reversible_matches_reference(ref):
    """BASELINE config #2: 3840x2160 8-bit RGB, 5/3, 64x64 blocks, 5 levels (6 321 code-blocks)."""
    from openjph_amd import codec
    img = synth_image(3, 2160, 3840, 8, seed=1234)
    got = codec.encode(img, bit_depth=8, color_transform=True)
    want = ref.encode(img, 8, color_transform=True)
    assert got == want, "4K codestream differs (%d vs %d bytes)" % (len(got), len(want))
    assert np.array_equal(codec.decode(got), img)


def test_8k_irreversible_properties():
    """BASELINE config #3 at full size (7680x4320x3, 12 bit, 9/7, qstep 0.001): size-independent
    properties -- decode(encode(x)) stays within the quantiser's error bound and re-encoding the
    decoded image is stable in size."""
    from openjph_amd import codec
    img = synth_image(3, 4320, 7680, 12, seed=1234)
    enc = codec.Encoder(bit_depth=12, width=7680, height=4320, num_comps=3, reversible=False, qstep=0.001)
    cs = enc.encode(img)
    dec = codec.decode(cs)
    err = dec.astype(np.int64) - img
    mse = float((err * err).mean()); pae = int(np.abs(err).max())
    # the reference's own figures on its synthetic C3 input are MSE 1.81 / PAE 8
    assert mse < 4.0 and pae <= 16, (mse, pae)
    bps = len(cs) / img.size
    assert 0.3 < bps < 1.5, bps


@pytest.mark.parametrize("kw", [dict(nc=3, h=120, w=200, bd=8, color_transform=True),
                                dict(nc=1, h=150, w=130, bd=12, reversible=False, qstep=0.002),
                                dict(nc=1, h=200, w=200, bd=10, tile=(128, 128))],
                         ids=["rgb-rct", "gray-irv", "tiled"])
def test_frame_batch_equals_frame_by_frame(kw):
    """BASELINE config #5 in small: a batch of independent frames through one set of launches gives
    the same codestreams / images as coding them one by one (and as the oracle)."""
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    from tests import cpu_pipeline as cp
    nc, h, w, bd, signed, c = _split(kw)
    B = 3
    frames = np.stack([synth_image(nc, h, w, bd, seed=40 + f) for f in range(B)])
    plan = Plan(make_params(w, h, nc, bit_depth=bd, **c))
    enc = codec.Encoder(plan=plan, frames=B)
    streams = enc.encode(frames)
    assert len(streams) == B
    for f in range(B):
        want, *_ = cp.encode(frames[f], bit_depth=bd, **c)
        assert streams[f] == want, "frame %d" % f
    dec = codec.Decoder(streams)
    out = dec.run_device().cpu().numpy()
    assert dec.failed_blocks() == 0
    for f in range(B):
        want_dec, _ = cp.decode(streams[f])
        assert np.array_equal(out[f], want_dec), "frame %d" % f


@pytest.mark.parametrize("kw", [dict(nc=3, h=200, w=300, bd=8, color_transform=True),
                                dict(nc=3, h=131, w=257, bd=8, reversible=False, color_transform=True, qstep=0.01),
                                dict(nc=1, h=300, w=500, bd=8, tile=(128, 128)),
                                dict(nc=3, h=100, w=150, bd=7, signed=True),
                                dict(nc=4, h=120, w=160, bd=8, color_transform=True),
                                dict(nc=1, h=64, w=1, bd=8), dict(nc=3, h=1, w=64, bd=8, color_transform=True)],
                         ids=["rgb-rct", "rgb-ict", "tiled", "signed7", "rgba-rct", "w1", "h1-rct"])
def test_sample_containers_8_16_32_agree(kw):
    """the same frame handed over in int32, 16-bit and 8-bit containers gives the same codestream (the oracle's), and
    decoding into the three container widths gives the same samples -- including the colour-transformed frames whose
    RCT / ICT runs inside the top DWT level"""
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    from tests import cpu_pipeline as cp
    nc, h, w, bd, signed, c = _split(kw)
    img = synth_image(nc, h, w, bd, seed=21, signed=signed)
    plan = Plan(make_params(w, h, nc, bit_depth=bd, is_signed=signed, **c))
    want, *_ = cp.encode(img, bit_depth=bd, is_signed=signed, **c)
    enc = codec.Encoder(plan=plan)
    for dt in (np.int32, np.int16, np.int8):
        cs = enc.encode(img.astype(dt))                    # unsigned values wrap into the container: the bits are what counts
        assert cs == want, "container %s: %d vs %d bytes" % (dt.__name__, len(cs), len(want))
    want_dec, _ = cp.decode(want)
    dec = codec.Decoder(want)
    for tdt, bits in ((torch.int32, 32), (torch.int16, 16), (torch.int8, 8)):
        out = dec.run_device(dtype=tdt).cpu().numpy().astype(np.int64)
        ref = want_dec.astype(np.int64)
        if bits < 32:                                      # a narrower container saturates at its own range (the reference's
            if not signed:                                 # 9/7 decode can leave 2^B, e.g. 256 for 8 bits)
                out &= (1 << bits) - 1
            ref = np.clip(ref, -(1 << (bits - 1)) if signed else 0, (1 << (bits - 1)) - 1 if signed else (1 << bits) - 1)
        assert np.array_equal(out, ref), "container %s" % tdt


def test_unfused_colour_path_still_matches():
    """OJPHGPU_NO_COLOUR_FUSION=1 sends colour-transformed frames through the stand-alone conversion kernels (the path
    frames with an NLT or a colour component without decompositions take): same bytes, same samples"""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "from openjph_amd import codec\nfrom tests import cpu_pipeline as cp\nfrom tests.synth import synth_image\n"
        "for kw in (dict(bit_depth=8, color_transform=True), dict(bit_depth=10, color_transform=True, reversible=False, qstep=0.004)):\n"
        "    img = synth_image(3, 150, 210, kw['bit_depth'], seed=9)\n"
        "    want, *_ = cp.encode(img, **kw)\n"
        "    assert codec.encode(img, **kw) == want\n"
        "    assert np.array_equal(codec.decode(want), cp.decode(want)[0])\n"
        "    assert codec.encode(img.astype(np.int16), **kw) == want\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OJPHGPU_NO_COLOUR_FUSION="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr[-2000:]


def test_frame_batch_with_different_quantisation():
    """a batch decoder takes K_max / delta from each frame's own QCD (frames 1.. were
    decoded with frame 0's step sizes); frames whose code-block grid differs are refused"""
    from openjph_amd import capi, codec
    from tests import cpu_pipeline as cp
    frames = [synth_image(1, 150, 130, 12, seed=50 + f) for f in range(3)]
    streams = [bytes(cp.encode(frames[f], bit_depth=12, reversible=False, qstep=q)[0]) for f, q in enumerate((0.002, 0.01, 0.0005))]
    dec = codec.Decoder(streams)
    out = dec.run_device().cpu().numpy()
    assert dec.failed_blocks() == 0
    for f in range(3):
        want, _ = cp.decode(streams[f])
        assert np.array_equal(out[f], want), "frame %d" % f
    # guard bits / exponents of a reversible stream may differ too (different bit depth is a different frame format: refused)
    other = bytes(cp.encode(frames[0], bit_depth=12, reversible=False, qstep=0.002, precinct=(64, 64))[0])
    with pytest.raises(capi.OjphError):
        codec.Decoder([streams[0], other])


def test_multi_pass_codestream_decodes_like_the_oracle():
    """Foreign-style codestreams (blocks with SigProp / MagRef segments): GPU decode == oracle decode
    (the oracle is pinned to the reference on such streams by tests/test_cpu_parity.py)."""
    from openjph_amd import codec
    from tests import cpu_pipeline as cp
    rng = np.random.default_rng(5)
    for kw in (dict(bit_depth=8, color_transform=True), dict(bit_depth=10, reversible=False, qstep=0.004, tile=(128, 128))):
        img = synth_image(1 if "tile" in kw else 3, 200, 260, kw["bit_depth"], seed=2)
        cs0, plan, arena, data, coded = cp.encode(img, **kw)
        data2, coded2 = cp.add_refinement(data, coded, rng)
        cs = plan.t2_write(data2, coded2)
        want, _ = cp.decode(cs)
        got = codec.decode(cs)
        assert np.array_equal(got, want), "%d samples differ" % int((got != want).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("regions", [0, 2, 8])
def test_encoder_output_regions(regions, monkeypatch):
    """the block encoder's compacted output split into regions with a cursor each (claim_output): same codestream
    with one cursor, two and eight regions, on noise that fills the blocks' bounds; the host path (used parts of
    the regions gathered, offsets moved along) and the device path (offsets as they are) can be mixed on one run"""
    import torch
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    from tests import cpu_pipeline as cp
    monkeypatch.setenv("OJPHGPU_ENC_REGIONS", str(regions))
    rng = np.random.default_rng(77)
    img = rng.integers(0, 1 << 16, size=(1, 300, 500), dtype=np.int64).astype(np.int32)       # incompressible
    c = dict(tile=(256, 256), block=(32, 32), tlm=True)
    plan = Plan(make_params(500, 300, 1, bit_depth=16, **c))
    want, *_ = cp.encode(img, bit_depth=16, **c)
    enc = codec.Encoder(plan=plan)
    enc.run_device(torch.from_numpy(img).cuda())
    assert enc.finish() == want
    dpart, lens = enc.finish_tiles_device()
    hpart, lens2 = enc.finish_tiles()
    assert bytes(dpart.cpu().numpy().tobytes()) == hpart and np.array_equal(lens, lens2)
    assert enc.finish() == want
    assert enc.coded_bytes() >= sum(int(x) for x in lens) - 64 * plan.num_tiles - 4096
    out = codec.decode(want)
    assert np.array_equal(out, img)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 256), (190, 250)], ids=["aligned", "ragged"])
def test_blocks_larger_than_the_lds_output_stage(shape):
    """incompressible 16-bit samples code ~2 bytes per sample: every 64x64 block overflows the block encoder's 5 KB
    LDS stage and goes through its flush path (whole dwords to the block's scratch slot, coding goes on in LDS)"""
    from openjph_amd import codec
    from tests import cpu_pipeline as cp
    h, w = shape
    rng = np.random.default_rng(5)
    img = rng.integers(0, 1 << 16, size=(1, h, w), dtype=np.int64).astype(np.int32)
    want, *_ = cp.encode(img, bit_depth=16)
    assert len(want) > 1.6 * h * w                      # the blocks are > 5 KB
    got = codec.encode(img, bit_depth=16)
    assert got == want
    assert np.array_equal(codec.decode(got), img)


@pytest.mark.gpu
def test_foreign_codestream_through_the_hip_decoder():
    """the only foreign-encoder codestream of the reference tree (tests/golden/foreign_test.j2c: 9/7 + ICT,
    per-resolution precincts, 77 blocks with SigProp / MagRef passes) through ht_dec_refine_kernel and the rest of the HIP
    decoder: sample for sample what the reference's generic build decodes (digest made by tests/golden/make_foreign.py,
    ojph_block_decoder32.cpp:1318-1609), within 1 of its SIMD build; the frame pipeline gives the same frame"""
    import hashlib, json
    from openjph_amd import codec
    from openjph_amd.pipeline import DecoderPipe
    from oracle import refbind
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(gd, "foreign.json")))["test_j2c"]
    cs = open(os.path.join(gd, g["file"]), "rb").read()
    dec = codec.Decoder(cs)
    assert int((dec.plan.coded_blocks()["num_passes"] > 1).sum()) == 77
    out = dec.decode()
    assert list(out.shape) == g["shape"]
    assert hashlib.sha256(np.ascontiguousarray(out.astype(np.int32)).tobytes()).hexdigest() == g["decoded_sha256_generic"]
    if refbind.available():                                    # the SIMD build's own tolerance: PAE <= 1
        want, _ = refbind.Ref().decode(cs)
        assert int(np.abs(out.astype(np.int64) - want).max()) <= g["max_abs_diff_generic_vs_simd"] <= 1
    pipe = DecoderPipe(cs, depth=2, container=32)
    slot = pipe.acquire(len(cs)); slot[:] = np.frombuffer(cs, np.uint8); pipe.submit()
    assert np.array_equal(pipe.collect().astype(np.int64), out.astype(np.int64))
    pipe.close()


@pytest.mark.gpu
def test_encoder_fuzz_seed_through_the_hip_encoder():
    """the seed input of the reference's encoder fuzz target (signed 12-bit, two components, one decomposition): the HIP
    path writes the reference's bytes (digest by tests/golden/make_foreign.py) and reads them back"""
    import hashlib, json
    from openjph_amd import codec
    from tests.golden_cases import fuzz_seed_case
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(gd, "foreign.json")))["fuzz_seed"]
    img, kw = fuzz_seed_case(open(os.path.join(gd, g["file"]), "rb").read())
    kw.pop("planar")
    cs = codec.encode(img, **kw)
    assert len(cs) == g["bytes"] and hashlib.sha256(cs).hexdigest() == g["sha256"]
    assert np.array_equal(codec.decode(cs), img)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"OJPHGPU_DEC_FUSED": "0"}, {"OJPHGPU_DEC_PREP": "1"}, {"OJPHGPU_DEC_FUSED": "2"},
                                 {"OJPHGPU_DEC_FUSED": "2", "OJPHGPU_FUSED_SHAPE": "0"}, {"OJPHGPU_DEC_FUSED": "2", "OJPHGPU_FUSED_RINGS": "1"}],
                         ids=["separate-launches", "prep-launch", "fused-wherever-possible", "fused-8-wavefront-shape", "fused-one-ring-per-wavefront"])
def test_the_other_decoder_schedules_decode_the_same(env, tmp_path):
    """the block decoder's default is ONE launch for step 1 + step 2 where that pays (blocks of 64 rows, few enough for
    resident workers: the first stream below, not the 32x32 one); the separate launches (also what blocks wider than 64
    samples and refinement passes take), the round-2 form with a prep launch, the other workgroup shape of the fused
    launch and its workers with one un-stuffing ring per wavefront (what frames of more than ~25 000 blocks take) are chosen per process by environment switches: each decodes the oracle's samples -- ragged block heights (quad
    rows not a multiple of a slice), tiles, a lossy and a lossless stream, twice in a row on the same decoder object"""
    import subprocess, sys
    script = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openjph_amd import codec
from tests import cpu_pipeline as cp
from tests.synth import synth_image
for kw, shape in ((dict(bit_depth=8, num_decomps=3), (1, 333, 517)), (dict(bit_depth=12, reversible=False, qstep=0.002, tile=(256, 192)), (3, 401, 611)),
                  (dict(bit_depth=10, block=(32, 32)), (1, 200, 300)),
                  # tall blocks: many slices of quad rows, the last one cut (4 + 2 + 2 rows), heights that end inside every piece
                  (dict(bit_depth=8, block=(16, 256), num_decomps=2), (1, 1021, 90)), (dict(bit_depth=9, block=(8, 512), num_decomps=1, reversible=False, qstep=0.01), (2, 1500, 40)),
                  (dict(bit_depth=8, block=(64, 64), num_decomps=1), (1, 2 * 61, 70)), (dict(bit_depth=8, block=(64, 64), num_decomps=1), (1, 2 * 59, 70))):
    img = synth_image(shape[0], shape[1], shape[2], kw["bit_depth"], seed=11)
    cs = codec.encode(img, **kw)
    want, _ = cp.decode(cs)
    dec = codec.Decoder(cs)
    for _ in range(2):
        got = dec.run_device().cpu().numpy()
        assert dec.failed_blocks() == 0 and np.array_equal(got, want), kw
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"OJPHGPU_DEC_DUAL": "2"}, {"OJPHGPU_DEC_DUAL": "0"}], ids=["four-or-two-blocks-per-wavefront", "two", "one"])
def test_blocks_of_at_most_32_columns(env):
    """frames whose code-blocks are all at most 32 (16) columns wide take step 2 with TWO (FOUR) blocks to a wavefront (a
    segment of 32 / 16 lanes per block: ht_dec_step2_multi_kernel; OJPHGPU_DEC_DUAL=2: two at most, =0: one): the oracle's samples
    for square, tall, flat and tiny blocks, ragged right / bottom blocks (odd widths and heights), an odd number of blocks,
    blocks that are not coded (a flat component), lossless and lossy, tiles, sub-sampled components -- and the same verdicts
    as the oracle pipeline on streams whose block bytes are damaged (ojph_block_decoder32.cpp:1091-1316 per block)"""
    import subprocess, sys
    script = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openjph_amd import codec
from tests import cpu_pipeline as cp
from tests.synth import synth_image
from openjph_amd.plan import parse_codestream
cases = ((dict(bit_depth=8, block=(32, 32)), (1, 333, 517)), (dict(bit_depth=12, block=(32, 32), reversible=False, qstep=0.002, tile=(256, 192)), (3, 401, 611)),
         (dict(bit_depth=10, block=(32, 64), num_decomps=2), (1, 259, 301)), (dict(bit_depth=8, block=(16, 256), num_decomps=2), (1, 1021, 90)),
         (dict(bit_depth=9, block=(8, 512), num_decomps=1, reversible=False, qstep=0.01), (2, 1500, 40)), (dict(bit_depth=8, block=(32, 8), num_decomps=3), (1, 97, 203)),
         (dict(bit_depth=8, block=(4, 4), num_decomps=2), (1, 37, 41)), (dict(bit_depth=8, block=(16, 16), num_decomps=5, color_transform=True), (3, 130, 94)),
         (dict(bit_depth=8, block=(32, 32), num_decomps=1), (1, 33, 31)), (dict(bit_depth=8, block=(32, 32), num_decomps=0), (1, 21, 32)))
for kw, shape in cases:
    img = synth_image(shape[0], shape[1], shape[2], kw["bit_depth"], seed=17)
    if shape[0] == 2:
        img[1] = 7                                     # a flat component: most of its blocks are not coded
    cs = codec.encode(img, **kw)
    want, _ = cp.decode(cs)
    dec = codec.Decoder(cs)
    for _ in range(2):
        got = dec.run_device().cpu().numpy()
        assert dec.failed_blocks() == 0 and np.array_equal(got, want), kw
# damaged block bytes: the same blocks refused, the same picture from the resilient read
rng = np.random.default_rng(23)
img = synth_image(1, 200, 300, 8, seed=3)
cs = bytearray(codec.encode(img, bit_depth=8, block=(32, 32), num_decomps=3))
sod = cs.index(b"\xff\x93") + 2
bad = 0
for trial in range(24):
    c2 = bytearray(cs)
    for _ in range(6):
        at = int(rng.integers(sod + 40, len(c2) - 2)); c2[at] = int(rng.integers(0, 256))
    try:
        plan = parse_codestream(bytes(c2), resilient=True)
        want = cp.inverse_stages(plan, cp.decode_blocks(plan, bytes(c2), resilient=True))
    except Exception:
        continue
    try:
        dec = codec.Decoder(bytes(c2), resilient=True)
        got = dec.run_device().cpu().numpy()
    except Exception as e:
        raise AssertionError("the oracle pipeline reads this stream, the device path raised: %%r" %% (e,))
    assert np.array_equal(got, want), trial
    bad += dec.failed_blocks() != 0
print("OK", bad)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stderr[-2000:]
