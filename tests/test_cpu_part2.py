"""Part-2 wavelets on the CPU side: codestreams written by this repository (oracle stages + plan + Tier-2) are read by
the live reference and decode to the same samples as the oracle pipeline; the parser reads back what the writer wrote."""
import numpy as np
import pytest

from tests.part2_cases import CASES, case_id, image, split


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_reference_decodes_part2_codestreams_like_the_oracle(case, ref, refgen):
    from openjph_amd.plan import parse_codestream
    from tests import cpu_pipeline as cp
    nc, h, w, bd, kw = split(case)
    img = image(nc, h, w, bd)
    cs, plan, *_ = cp.encode(img, **kw)
    dec, dplan = cp.decode(cs)
    rev_all = all(plan.comp_style(i)["reversible"] for i in range(nc))
    want, _ = (ref if rev_all else refgen).decode(cs)
    assert np.array_equal(dec, want), "oracle pipeline and reference decode %d samples differently" % int((dec != want).sum())
    for c in range(nc):                                     # the parsed plan describes the wavelets the writer's plan had
        L = plan.comp_style(c)["num_decomps"]
        assert dplan.comp_style(c) == plan.comp_style(c)
        for d in range(1, L + 1):
            assert dplan.comp_lift(c, d) == plan.comp_lift(c, d)
    odd = any(len(a["steps"]) % 2 for a in kw.get("atk", {}).values())
    if rev_all and not odd:                                 # (the reference's analysis and synthesis disagree for an odd number of steps)
        assert np.array_equal(dec, img)
    # reduced resolutions: a skipped level may halve one direction only (param_dfs::get_res_downsamp)
    L0 = min(plan.comp_style(c)["num_decomps"] for c in range(nc))
    if L0 >= 1:
        d1, _ = cp.decode(cs, skip=(1, 1))
        w1, _ = (ref if rev_all else refgen).decode(cs, skip=(1, 1))
        if isinstance(w1, list):
            assert all(np.array_equal(a, b) for a, b in zip(d1, w1))
        else:
            assert np.array_equal(d1, w1)


def test_random_part2_configurations_decode_like_the_reference(ref, refgen):
    """a few seconds of tools/fuzz_part2_cpu.py (random ATK kernels, DFS level kinds, component styles, tiles, offsets): the live
    reference decodes what this repository writes to the oracle pipeline's samples (profiles/r04_a_part2_fuzz.txt: 25 000 cases)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_part2_cpu
    assert fuzz_part2_cpu.main(seconds=6.0, seed=5) == 0


def test_cut_part2_codestreams_behave_like_the_reference(ref, refgen):
    """Part-2 codestreams cut at every byte of the main header (ATK / DFS segments included) and of the first tile-part header,
    and at a dozen points of the data: the parser raises exactly when the reference raises, with and without resilience, and
    otherwise reconstructs the same image (tools/fuzz_trunc_part2_cpu.py runs every case; this test three of them)"""
    from openjph_amd import capi
    from openjph_amd.plan import parse_codestream
    from tests import cpu_pipeline as cp
    for case in (CASES[2], CASES[5], CASES[8]):
        nc, h, w, bd, kw = split(case)
        cs, plan, *_ = cp.encode(image(nc, h, w, bd), **kw)
        r = ref if all(plan.comp_style(i)["reversible"] for i in range(nc)) else refgen
        sot = cs.find(b"\xff\x90")
        raised = 0
        for k in list(range(2, sot + 14)) + [len(cs) * c // 12 for c in range(1, 12)] + [len(cs) - 1, len(cs) - 2]:
            for resilient in (False, True):
                try:
                    want, _ = r.decode(cs[:k], resilient=resilient)
                except RuntimeError:
                    want = None
                try:
                    pl = parse_codestream(cs[:k], resilient=resilient)
                    got = cp.inverse_stages(pl, cp.decode_blocks(pl, cs[:k]))
                except capi.OjphError:
                    got = None
                assert (want is None) == (got is None), "cut at %d of %d (main header %d), resilient=%s" % (k, len(cs), sot, resilient)
                if want is not None:
                    assert all(np.array_equal(a, b) for a, b in zip(got, want)) if isinstance(want, list) else np.array_equal(got, want)
                raised += want is None
        assert raised > 0
