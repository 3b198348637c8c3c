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
