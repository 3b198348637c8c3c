"""Part-2 wavelets (ATK lifting kernels, DFS decompositions) through the HIP path: the general lifting kernels under the
whole-frame codec objects.  The codestream equals the oracle pipeline's byte for byte, the decode equals the oracle
pipeline's -- which tests/test_cpu_part2.py pins to the live reference's decoder -- and, where oracle/_ref travelled to
this box, the reference's decode of the same bytes directly (ojph_resolution.cpp:713-949, ojph_params.cpp:2530-2896)."""
import numpy as np
import pytest

from tests.part2_cases import CASES, case_id, image, split

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_part2_codestreams_on_the_gpu(case):
    from openjph_amd import codec
    from oracle import refbind
    from tests import cpu_pipeline as cp
    nc, h, w, bd, kw = split(case)
    img = image(nc, h, w, bd)
    want, plan, *_ = cp.encode(img, **kw)
    got = codec.encode(img, **kw)
    if got != want:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        pytest.fail("codestream differs: %d vs %d bytes, first difference at %d" % (len(got), len(want), first))
    dec = codec.decode(want)
    want_dec, _ = cp.decode(want)
    assert np.array_equal(dec, want_dec), "%d samples differ" % int((dec != want_dec).sum())
    rev_all = all(plan.comp_style(i)["reversible"] for i in range(nc))
    if refbind.available(generic=not rev_all):
        rdec, _ = refbind.Ref(generic=not rev_all).decode(want)
        assert np.array_equal(dec, rdec)
    L0 = min(plan.comp_style(c)["num_decomps"] for c in range(nc))
    if L0 >= 1:                                             # reduced resolution: a skipped level may halve one direction only
        d = codec.Decoder(want, skip_res=(1, 1))
        d1 = d.plan.unpack_frame(d.decode())                # (components may come out in different sizes)
        w1, _ = cp.decode(want, skip=(1, 1))
        for c in range(nc):
            assert np.array_equal(d1[c], w1[c]), "reduced resolution: component %d differs" % c


@pytest.mark.parametrize("shape", [(3, 270, 350, 12), (1, 129, 67, 10)])
def test_the_97_as_an_atk_segment_is_the_97(shape):
    """bench.py's c7 workload in small: the 9/7's four lifting steps and K written as an ATK marker segment go through the
    general lifting kernels and must code and decode EXACTLY what the built-in 9/7 kernels do -- the same code-block bytes
    (the codestreams differ in their marker segments only) and the same samples"""
    from openjph_amd import codec
    from openjph_amd.plan import parse_codestream
    from tests.synth import synth_image
    from bench import ATK97
    nc, h, w, bd = shape
    img = synth_image(nc, h, w, bd, seed=97)
    kw = dict(bit_depth=bd, reversible=False, qstep=0.002, num_decomps=4)
    plain = codec.encode(img, **kw)
    atk = codec.encode(img, atk=ATK97, wavelet=2, **kw)
    pa, pb = parse_codestream(plain), parse_codestream(atk)
    ca, cb = pa.coded_blocks(), pb.coded_blocks()
    assert len(ca) == len(cb)
    for x, y in zip(ca, cb):
        assert int(x["len1"]) == int(y["len1"]) and plain[int(x["offset"]):int(x["offset"]) + int(x["len1"])] == atk[int(y["offset"]):int(y["offset"]) + int(y["len1"])]
    assert np.array_equal(codec.decode(plain), codec.decode(atk))
