"""Part-2 (T.801) wavelet configurations the tests share: ATK marker segments (lifting kernels) and DFS marker segments
(levels that transform one direction only, or none).  The reference only reads these (param_atk::read / param_dfs::read,
ojph_params.cpp:2596-2866; resolution::pull_line's HORZ_TRX / VERT_TRX, ojph_resolution.cpp:713-949), so the codestreams
come from this repository's own writer and the pin is "the reference decodes them to the same samples"."""
A97 = [0.443506852043971, 0.882911075530934, -0.052980118572961, -1.586134342059924]
K97 = 1.230174104914001

CASES = [
    # the two Part-1 wavelets re-expressed as ATK kernels
    dict(nc=1, h=96, w=130, bd=8, wavelet=2, atk={2: dict(steps=[(1, 2, 2), (-1, 1, 1)])}),
    dict(nc=1, h=100, w=90, bd=10, wavelet=3, atk={3: dict(steps=A97, K=K97)}, qstep=0.01),
    # one component with a decomposition from a DFS marker segment: both directions, horizontal only, vertical only, none
    dict(nc=2, h=96, w=130, bd=8, coc={1: dict(reversible=True, dfs=0, block=(32, 32))}, dfs={0: [1, 2, 3, 1, 0]}),
    dict(nc=3, h=80, w=120, bd=8, num_decomps=3,
         coc={0: dict(reversible=True, dfs=2, num_decomps=3), 2: dict(reversible=True, dfs=5, num_decomps=3, wavelet=7)},
         dfs={2: [2, 2, 1], 5: [3, 1]}, atk={7: dict(steps=[(1, 2, 2), (-1, 1, 1)])}),
    # other kernels: four reversible steps, general coefficients, two irreversible steps, an odd number of steps
    dict(nc=1, h=64, w=64, bd=12, wavelet=9, atk={9: dict(steps=[(1, 4, 3), (-1, 1, 1), (1, 2, 2), (-1, 1, 1)])}),
    dict(nc=1, h=77, w=91, bd=9, wavelet=5, atk={5: dict(steps=[(3, 8, 4), (-5, 4, 3)])}, tile=(64, 48)),
    dict(nc=1, h=70, w=50, bd=8, wavelet=4, atk={4: dict(steps=[0.25, -0.5], K=1.2)}, qstep=0.02, reversible=False),
    dict(nc=1, h=70, w=50, bd=8, wavelet=4, atk={4: dict(steps=[(1, 2, 2), (-1, 1, 1), (1, 1, 1)])}),
    dict(nc=1, h=60, w=80, bd=8, wavelet=6, atk={6: dict(steps=[0.2, -0.4, 0.1], K=1.1)}, qstep=0.02, reversible=False),
    # an irreversible component with horizontal-only levels next to a 5/3 one; odd origins
    dict(nc=2, h=90, w=70, bd=8, reversible=False, qstep=0.01, num_decomps=4, image_offset=(3, 5),
         coc={0: dict(reversible=False, dfs=1, num_decomps=4, wavelet=8), 1: dict(reversible=True, num_decomps=2)},
         dfs={1: [2, 3, 1, 2]}, atk={8: dict(steps=A97, K=K97)}),
    # 64-bit samples under a DFS decomposition
    dict(nc=1, h=64, w=96, bd=31, num_decomps=3, coc={0: dict(reversible=True, dfs=3, num_decomps=3)}, dfs={3: [1, 3, 2]}),
]


def split(case):
    c = dict(case)
    nc, h, w, bd = c.pop("nc"), c.pop("h"), c.pop("w"), c.pop("bd")
    return nc, h, w, bd, dict(c, bit_depth=bd)


def image(nc, h, w, bd):
    from tests.synth import synth_image
    if bd <= 16:
        return synth_image(nc, h, w, bd, seed=3)
    from tests.test_gpu_wide import deep_image
    return deep_image(nc, h, w, bd, False)


def case_id(c):
    return "-".join("%s%s" % (k, str(v).replace(" ", "")[:24]) for k, v in c.items() if k not in ("atk", "dfs"))
