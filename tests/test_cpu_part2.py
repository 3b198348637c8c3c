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
