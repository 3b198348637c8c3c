"""Case lists shared by tests/golden/make_golden.py (which runs the real reference once, in the
build container) and tests/test_cpu_parity.py (which checks the oracle + host pipeline against the
stored reference outputs anywhere)."""
import numpy as np

from tests.synth import synth_image

# (w, h, K_max, density, amplitude, seed) -- single HT code-blocks (ojph_encode_codeblock32 inputs)
BLOCK_CASES = [
    (64, 64, 10, 0.5, 300, 1), (64, 64, 8, 0.05, 100, 2), (64, 64, 16, 1.0, 30000, 3),
    (64, 64, 12, 0.9, 3, 4), (32, 32, 9, 0.3, 200, 5), (128, 32, 11, 0.5, 900, 6),
    (32, 128, 11, 0.5, 900, 7), (4, 1024, 10, 0.4, 400, 8), (1024, 4, 10, 0.4, 400, 9),
    (1, 1, 6, 1.0, 20, 10), (1, 17, 7, 1.0, 60, 11), (17, 1, 7, 1.0, 60, 12), (2, 2, 5, 1.0, 15, 13),
    (3, 5, 8, 0.7, 120, 14), (63, 61, 10, 0.6, 500, 15), (64, 3, 9, 0.2, 250, 16),
    (5, 64, 9, 0.8, 250, 17), (64, 64, 20, 0.7, 500000, 18), (64, 64, 30, 0.6, 500000000, 19),
    (64, 64, 6, 0.02, 30, 20), (16, 16, 7, 1.0, 1, 21), (47, 33, 13, 0.95, 4000, 22),
]

# whole codestreams: small images over a spread of parameters
STREAM_CASES = [
    dict(nc=1, h=256, w=256, bd=8),
    dict(nc=3, h=200, w=300, bd=8, color_transform=True),
    dict(nc=3, h=131, w=257, bd=10),
    dict(nc=1, h=517, w=389, bd=12, num_decomps=3),
    dict(nc=1, h=300, w=500, bd=16, tile=(128, 128)),
    dict(nc=3, h=260, w=260, bd=8, prog_order="CPRL", precinct=(128, 128)),
    dict(nc=1, h=200, w=333, bd=8, prog_order="LRCP"),
    dict(nc=1, h=200, w=333, bd=8, prog_order="RLCP", tile=(100, 200)),
    dict(nc=3, h=150, w=150, bd=8, prog_order="PCRL", precinct=(64, 64)),
    dict(nc=1, h=64, w=1, bd=8),
    dict(nc=1, h=1, w=64, bd=8),
    dict(nc=1, h=1, w=1, bd=8),
    dict(nc=1, h=5, w=7, bd=8),
    dict(nc=1, h=200, w=200, bd=8, block=(128, 32)),
    dict(nc=1, h=200, w=200, bd=8, block=(4, 1024)),
    dict(nc=1, h=256, w=256, bd=8, signed=True),
    dict(nc=1, h=256, w=256, bd=8, num_decomps=0),
    dict(nc=1, h=256, w=256, bd=8, tlm=True, tile=(64, 64)),
    dict(nc=1, h=256, w=256, bd=12, reversible=False),
    dict(nc=3, h=200, w=300, bd=8, reversible=False, color_transform=True),
    dict(nc=3, h=240, w=320, bd=12, reversible=False, qstep=0.001),
    dict(nc=1, h=300, w=500, bd=10, reversible=False, tile=(128, 128), qstep=0.01),
    dict(nc=1, h=97, w=113, bd=8, reversible=False, num_decomps=2, qstep=0.05),
]


# codestreams on a general reference grid: sub-sampled components, image offset, tile offset
# (the reference's own tests: -downsamp {1,1},{2,2},{2,2}, -image_offset {1,0}, tiles of 33x33)
GRID_CASES = [
    dict(w=352, h=288, bd=8, ds=[(1, 1), (2, 2), (2, 2)], reversible=False, qstep=0.1),           # CIF 4:2:0
    dict(w=101, h=67, bd=8, ds=[(1, 1), (2, 2), (2, 2)], num_decomps=3),
    dict(w=130, h=70, bd=10, ds=[(1, 1), (2, 1), (2, 1)], num_decomps=2, tile=(64, 32), prog_order="PCRL"),   # 4:2:2
    dict(w=97, h=53, bd=8, ds=[(1, 1)], image_offset=(1, 0), reversible=False, num_decomps=4, qstep=0.1),
    dict(w=1, h=300, bd=8, ds=[(1, 1)], image_offset=(1, 0), num_decomps=5),                      # tall and narrow at an odd origin
    dict(w=97, h=53, bd=8, ds=[(1, 1)] * 3, image_offset=(5, 3), tile=(33, 33), tile_offset=(2, 1),
         color_transform=True, num_decomps=3, prog_order="CPRL"),
    dict(w=120, h=90, bd=12, ds=[(1, 1), (2, 2), (4, 1)], image_offset=(3, 7), tile=(50, 40), tile_offset=(1, 2),
         num_decomps=3, precinct=(32, 32)),
    dict(w=200, h=150, bd=8, ds=[(1, 1), (2, 2), (2, 2), (1, 1)], reversible=False, qstep=0.02, tile=(128, 128),
         prog_order="RLCP", tlm=True),
    dict(w=64, h=64, bd=8, ds=[(1, 1), (3, 5)], image_offset=(7, 11), num_decomps=2, prog_order="LRCP"),
]


# components of different bit depth / signedness (QCC marker segments) and the qfactor mode
# (visually weighted steps, a QCC for every component): w, h, sub-sampling, depths, signs, kwargs
FORMAT_CASES = [
    dict(w=120, h=90, ds=[(1, 1)] * 3, depths=[8, 10, 12], signs=[False, False, True], kw=dict(num_decomps=3)),
    dict(w=120, h=90, ds=[(1, 1)] * 2, depths=[12, 8], signs=[False, True], kw=dict(reversible=False, num_decomps=4)),
    dict(w=100, h=80, ds=[(1, 1)] * 4, depths=[8, 8, 8, 16], signs=[False] * 4, kw=dict(tile=(64, 64), prog_order="CPRL", color_transform=True)),
    dict(w=128, h=96, ds=[(1, 1), (2, 2), (2, 2)], depths=[8, 8, 8], signs=[False] * 3, kw=dict(reversible=False, qfactor=50)),
    dict(w=128, h=96, ds=[(1, 1), (2, 1), (2, 1)], depths=[10, 10, 10], signs=[False] * 3, kw=dict(reversible=False, qfactor=85, num_decomps=6)),
    dict(w=200, h=150, ds=[(1, 1)] * 3, depths=[8, 8, 8], signs=[False] * 3, kw=dict(reversible=False, qfactor=30, color_transform=True)),
    dict(w=100, h=80, ds=[(1, 1)], depths=[12], signs=[False], kw=dict(reversible=False, qfactor=99)),
    dict(w=100, h=80, ds=[(1, 1)] * 3, depths=[8, 8, 8], signs=[False] * 3, kw=dict(qfactor=70, color_transform=True)),
    dict(w=90, h=70, ds=[(1, 1)] * 3, depths=[8, 16, 10], signs=[False, True, False], kw=dict(reversible=False, qstep=0.01, tile=(50, 50))),
]


def format_case(i, seed=8):
    """-> (planes, kwargs for plan.make_params / refbind.Ref.encode (bit_depth / is_signed = component 0's), (W, H))"""
    import numpy as np
    c = FORMAT_CASES[i]
    rng = np.random.default_rng(seed + i)
    planes = []
    for (dx, dy), bd, sg in zip(c["ds"], c["depths"], c["signs"]):
        cw, ch = -(-c["w"] // dx), -(-c["h"] // dy)
        lo, hi = (-(1 << (bd - 1)), 1 << (bd - 1)) if sg else (0, 1 << bd)
        yy, xx = np.mgrid[0:ch, 0:cw]
        base = ((np.sin(xx / 9.0) + np.cos(yy / 7.0)) * 0.2 + 0.5) * (hi - lo) + lo
        planes.append(np.clip(base + rng.integers(-3, 4, (ch, cw)), lo, hi - 1).astype(np.int32))
    kw = dict(c["kw"], bit_depth=c["depths"][0], is_signed=c["signs"][0], downsampling=c["ds"], bit_depths=c["depths"], signs=c["signs"])
    return planes, kw, (c["w"], c["h"])


# per-component coding styles (COC marker segments, param_cod's comp_idx setters): decompositions, block
# size, precincts and wavelet of single components; the first one is the reference's own
# tests/test_mixed_coc.cpp (4 components, the last one reversible inside an irreversible codestream).
# (components, w, h, depths or one depth, kwargs incl. coc={component: settings}, skip or None)
COC_CASES = [
    dict(nc=4, w=64, h=64, bd=8, kw=dict(reversible=False, qstep=0.01, coc={3: dict(reversible=True)})),
    dict(nc=3, w=200, h=150, bd=8, kw=dict(reversible=True, coc={1: dict(reversible=True, num_decomps=2, block=(32, 32))})),
    dict(nc=3, w=200, h=150, bd=10, kw=dict(reversible=False, prog_order="LRCP",
                                            coc={0: dict(num_decomps=3), 2: dict(reversible=True, num_decomps=1, block=(16, 64))})),
    dict(nc=2, w=130, h=97, bd=8, kw=dict(reversible=True, prog_order="CPRL", num_decomps=2,
                                          coc={1: dict(reversible=True, num_decomps=4, precincts=[(32, 32), (64, 64)])})),
    dict(nc=3, w=130, h=97, bd=12, kw=dict(reversible=True, prog_order="PCRL", num_decomps=3, coc={2: dict(reversible=False, num_decomps=0)})),
    dict(nc=3, w=130, h=97, bd=8, kw=dict(reversible=False, prog_order="RLCP", tileparts="RC", tlm=True, num_decomps=3,
                                          coc={2: dict(reversible=False, num_decomps=1)}), resilient=True),   # tile-part numbers with gaps
    dict(nc=3, w=130, h=97, bd=8, kw=dict(reversible=False, prog_order="RPCL", tileparts="R", tlm=True, num_decomps=2, tile=(64, 64),
                                          coc={1: dict(reversible=True, num_decomps=4)})),
    dict(nc=3, w=160, h=120, bd=8, kw=dict(reversible=True, num_decomps=4, tile=(96, 96), prog_order="CPRL", tileparts="C",
                                           coc={2: dict(reversible=True, num_decomps=2), 0: dict(reversible=True, num_decomps=4, block=(128, 16))})),   # creation order 2, 0
    dict(nc=4, w=100, h=80, bd=8, kw=dict(reversible=True, color_transform=True, num_decomps=3,
                                          coc={3: dict(reversible=False, num_decomps=2)})),                   # RCT on 0..2, a 9/7 alpha-like plane
    dict(nc=3, w=128, h=96, bd=8, kw=dict(reversible=False, qfactor=60, coc={1: dict(reversible=False, num_decomps=3)})),
    dict(nc=3, w=128, h=96, depths=[8, 12, 10], signs=[False, True, False],
         kw=dict(reversible=False, qstep=0.02, coc={1: dict(reversible=True, num_decomps=3), 2: dict(reversible=False, num_decomps=5, block=(32, 32))})),
    dict(nc=3, w=128, h=96, bd=8, ds=[(1, 1), (2, 2), (2, 2)], kw=dict(reversible=True, coc={1: dict(reversible=True, num_decomps=3), 2: dict(reversible=True, num_decomps=3)})),
    dict(nc=3, w=150, h=110, bd=8, kw=dict(reversible=True, num_decomps=4, coc={2: dict(reversible=True, num_decomps=3)}), skip=(1, 1)),
    dict(nc=3, w=150, h=110, bd=8, kw=dict(reversible=False, num_decomps=3, prog_order="LRCP", coc={0: dict(reversible=False, num_decomps=5)}), skip=(2, 1)),
    dict(nc=2, w=64, h=64, bd=8, kw=dict(reversible=True, num_decomps=0, coc={1: dict(reversible=True, num_decomps=0, block=(32, 32))})),
]


def coc_case(i, seed=21):
    """-> (planes, kwargs for plan.make_params / refbind.Ref.encode, (W, H), skip or None, resilient)"""
    import numpy as np
    c = COC_CASES[i]
    nc = c["nc"]
    depths = c.get("depths", [c.get("bd", 8)] * nc)
    signs = c.get("signs", [False] * nc)
    ds = c.get("ds", [(1, 1)] * nc)
    rng = np.random.default_rng(seed + i)
    planes = []
    for (dx, dy), bd, sg in zip(ds, depths, signs):
        cw, ch = -(-c["w"] // dx), -(-c["h"] // dy)
        lo, hi = (-(1 << (bd - 1)), 1 << (bd - 1)) if sg else (0, 1 << bd)
        yy, xx = np.mgrid[0:ch, 0:cw]
        base = ((np.sin(xx / 8.0) + np.cos(yy / 6.0)) * 0.2 + 0.5) * (hi - lo) + lo
        planes.append(np.clip(base + rng.integers(-4, 5, (ch, cw)), lo, hi - 1).astype(np.int32))
    kw = dict(c["kw"], bit_depth=depths[0], is_signed=signs[0])
    if "depths" in c:
        kw.update(bit_depths=depths, signs=signs)
    if "ds" in c:
        kw.update(downsampling=ds)
    return planes, kw, (c["w"], c["h"]), c.get("skip"), c.get("resilient", False)


# param_qcd::set_qfactor(comp_idx, ctype, qfactor): quality factors of single components -- their QCCs
# come first, in creation order, then the ones the library adds (3 or 4 components of 120x90, 8 bit)
CQF_CASES = [
    dict(nc=3, kw=dict(reversible=False, qfactors={1: ("Cb", 40), 0: ("Y", 80)})),
    dict(nc=3, kw=dict(reversible=False, qfactor=60, qfactors={2: ("Cr", 30)})),
    dict(nc=3, kw=dict(reversible=False, qstep=0.02, qfactors={2: ("Y", 90)}, coc={1: dict(num_decomps=3)})),
    dict(nc=3, kw=dict(reversible=True, qfactors={1: ("Y", 50)})),
    dict(nc=4, kw=dict(reversible=False, color_transform=True, qfactors={0: ("Y", 70), 1: ("Cb", 70), 2: ("Cr", 70)})),
]


def cqf_case(i):
    c = CQF_CASES[i]
    return synth_image(c["nc"], 90, 120, 8, seed=4), dict(c["kw"], bit_depth=8)


# NLT marker segments (param_nlt::set_nonlinear_transform): the type 3 non-linearity on signed
# components, the ALL_COMPS entry with components of one / of different formats (the library then writes
# one segment or one per component, ojph_params.cpp:2087-2170), explicit type 0 entries, creation order
NLT_CASES = [
    dict(w=100, h=80, depths=[8] * 3, signs=[True] * 3, kw=dict(reversible=True, nlt={"all": 3})),
    dict(w=100, h=80, depths=[8, 10, 12], signs=[True, False, True], kw=dict(reversible=True, nlt={"all": 3}, num_decomps=3)),
    dict(w=100, h=80, depths=[12] * 3, signs=[True] * 3, kw=dict(reversible=False, qstep=0.001, nlt={"all": 3, 1: 0})),
    dict(w=100, h=80, depths=[10] * 3, signs=[True] * 3, kw=dict(reversible=True, color_transform=True, nlt={2: 3, 0: 3, 1: 3})),
    dict(w=64, h=64, depths=[16], signs=[True], kw=dict(reversible=True, nlt={0: 3}, num_decomps=0)),
    dict(w=64, h=64, depths=[8, 8], signs=[False, True], kw=dict(reversible=True, nlt={"all": 0, 1: 3}, tile=(40, 40))),
    dict(w=90, h=70, depths=[8, 8, 8], signs=[True] * 3, kw=dict(reversible=False, color_transform=True, qstep=0.01, nlt={"all": 3})),
    dict(w=90, h=70, depths=[12, 12], signs=[True, True], ds=[(1, 1), (2, 2)],
         kw=dict(reversible=False, nlt={"all": 3}, coc={1: dict(reversible=True, num_decomps=2)})),
]


def nlt_case(i, seed=33):
    """-> (planes with negative samples, kwargs for plan.make_params / refbind.Ref.encode, (W, H))"""
    import numpy as np
    c = NLT_CASES[i]
    rng = np.random.default_rng(seed + i)
    planes = []
    ds = c.get("ds", [(1, 1)] * len(c["depths"]))
    for (dx, dy), bd, sg in zip(ds, c["depths"], c["signs"]):
        cw, ch = -(-c["w"] // dx), -(-c["h"] // dy)
        lo, hi = (-(1 << (bd - 1)), 1 << (bd - 1)) if sg else (0, 1 << bd)
        yy, xx = np.mgrid[0:ch, 0:cw]
        base = ((np.sin(xx / 9.0) + np.cos(yy / 7.0)) * 0.3 + 0.5) * (hi - lo) + lo
        planes.append(np.clip(base + rng.integers(-5, 6, (ch, cw)), lo, hi - 1).astype(np.int32))
    kw = dict(c["kw"], bit_depth=c["depths"][0], is_signed=c["signs"][0], bit_depths=c["depths"], signs=c["signs"])
    if "ds" in c:
        kw["downsampling"] = ds
    return planes, kw, (c["w"], c["h"])


# tile-part divisions (codestream::set_tilepart_divisions) on a 3-component 150x200 image:
# (progression order, divisions, further kwargs)
TILEPART_CASES = [
    ("LRCP", "R", {}), ("LRCP", "C", {}), ("RLCP", "RC", dict(tile=(64, 64), tlm=True)), ("RPCL", "R", dict(precinct=(64, 64), num_decomps=3)),
    ("RPCL", "RC", {}), ("PCRL", "RC", {}), ("CPRL", "C", dict(tile=(64, 64), tlm=True)), ("CPRL", "R", {}),
    ("RPCL", "R", dict(tlm=True, reversible=False, qstep=0.05)),
]


def tilepart_case(i):
    po, tp, extra = TILEPART_CASES[i]
    img = synth_image(3, 150, 200, 8, seed=2)
    return img, dict(bit_depth=8, prog_order=po, tileparts=tp, **extra)


# reduced-resolution decoding (codestream::restrict_input_resolution): (family, case index,
# skipped_res_for_data, skipped_res_for_recon)
SKIP_CASES = [
    ("grid", 0, 1, 1), ("grid", 0, 2, 1), ("grid", 0, 5, 5), ("grid", 0, 5, 0), ("grid", 2, 2, 2), ("grid", 5, 1, 1),
    ("grid", 5, 3, 3), ("grid", 6, 2, 1), ("grid", 7, 2, 2), ("stream", 1, 1, 1), ("stream", 3, 3, 2), ("stream", 4, 1, 1),
    ("stream", 19, 2, 2), ("stream", 9, 1, 1), ("stream", 12, 5, 5),
]


def skip_case(i):
    """-> (planes list, encode kwargs, size or None, (skip_read, skip_recon))"""
    fam, k, a, b = SKIP_CASES[i]
    if fam == "grid":
        planes, kw, size = grid_kwargs(GRID_CASES[k])
        return planes, kw, size, (a, b)
    img, kw = stream_kwargs(STREAM_CASES[k])
    return [img[c] for c in range(img.shape[0])], kw, None, (a, b)


def grid_kwargs(case, seed=5):
    """-> (list of per-component int32 planes, kwargs for plan.make_params / refbind.Ref.encode,
    (W, H) on the reference grid)"""
    import numpy as np
    c = dict(case)
    w, h, bd, ds = c.pop("w"), c.pop("h"), c.pop("bd"), c.pop("ds")
    ox, oy = c.get("image_offset", (0, 0))
    rng = np.random.default_rng(seed)
    planes = []
    for i, (dx, dy) in enumerate(ds):
        cw = -(-(ox + w) // dx) - -(-ox // dx)
        ch = -(-(oy + h) // dy) - -(-oy // dy)
        yy, xx = np.mgrid[0:ch, 0:cw]
        smooth = (np.sin(xx / (7.0 + i)) + np.cos(yy / (5.0 + 2 * i))) * (1 << (bd - 3)) + (1 << (bd - 1))
        noise = rng.integers(-(1 << max(bd - 5, 0)), (1 << max(bd - 5, 0)) + 1, size=(ch, cw))
        planes.append(np.clip(smooth + noise, 0, (1 << bd) - 1).astype(np.int32))
    return planes, dict(c, bit_depth=bd, downsampling=ds), (w, h)


def stream_kwargs(case, seed=3):
    """-> (image int32 [C,H,W], kwargs understood by plan.make_params / refbind.Ref.encode)"""
    c = dict(case)
    nc, h, w, bd = c.pop("nc"), c.pop("h"), c.pop("w"), c.pop("bd")
    signed = c.pop("signed", False)
    img = synth_image(nc, h, w, bd, seed=seed, signed=signed)
    return img, dict(c, bit_depth=bd, is_signed=signed)


# code-blocks that also carry SigProp (+ MagRef) segments of random bytes:
# (w, h, K_max, density, amplitude, seed, num_passes, len2, stripe_causal)
REFINE_CASES = [
    (64, 64, 10, 0.5, 300, 31, 2, 200, False), (64, 64, 10, 0.5, 300, 31, 3, 200, False),
    (64, 64, 6, 0.05, 20, 32, 3, 500, False), (64, 64, 14, 0.9, 5000, 33, 3, 1500, True),
    (32, 32, 9, 0.3, 200, 34, 2, 90, True), (17, 64, 8, 0.4, 100, 35, 3, 77, False),
    (64, 17, 8, 0.4, 100, 36, 3, 77, False), (4, 1024, 10, 0.4, 400, 37, 3, 300, False),
    (1024, 4, 10, 0.4, 400, 38, 3, 300, True), (1, 1, 6, 1.0, 20, 39, 3, 5, False),
    (5, 7, 7, 0.6, 60, 40, 2, 9, False), (63, 61, 12, 0.2, 900, 41, 3, 1, False),
    (128, 32, 11, 0.5, 900, 42, 3, 2046, False), (64, 64, 30, 0.6, 500000000, 43, 3, 64, False),
]


def refine_case(i):
    """-> (cleanup bytes from the oracle encoder is NOT used here: callers pass them) parameters + refinement bytes"""
    import numpy as np
    w, h, kmax, density, amp, seed, npass, len2, causal = REFINE_CASES[i]
    rng = np.random.default_rng(seed)
    from tests.synth import random_block
    stride = (w + 15) // 16 * 16
    q, _ = random_block(rng, w, h, stride, kmax, density, amp)
    q[:, w:] = 0
    tail = bytes(rng.integers(0, 256, size=len2, dtype=np.uint8))
    if i % 3 == 0:                       # runs of the bytes the stuffing rules care about
        tail = bytes((0xFF, 0x7F, 0x8F, 0x90, 0xFF, 0xFF)[j % 6] if (j // 7) % 2 else tail[j] for j in range(len2))
    return q, w, h, stride, kmax, npass, causal, tail


def fuzz_seed_case(blob: bytes):
    """An input of the reference's encoder fuzz target, decoded the way fuzzing/fuzz_targets/ojph_compress_fuzz_target.cpp
    :46-124 reads it: 4 control bytes (width-1, height-1, components / depth selector / signed / reversible / colour
    transform, decompositions / planar), then one sample per byte, wrapping around, pushed in exchange() order (planar:
    component after component; otherwise row by row with the components of a row in turn).  -> (image [C,H,W] int32,
    encoder keyword arguments)"""
    d = np.frombuffer(blob, np.uint8)
    w, h = int(d[0] & 0x7F) + 1, int(d[1] & 0x7F) + 1
    nc = int(d[2] & 3) + 1
    bd = (8, 10, 12, 16)[(int(d[2]) >> 2) & 3]
    sg, rev, ct = bool((d[2] >> 4) & 1), bool((d[2] >> 5) & 1), bool((d[2] >> 6) & 1)
    nd = min(int(d[3] & 7), 5)
    planar = bool((d[3] >> 3) & 1)
    if nc < 3:
        ct = False
    if ct:
        planar = False
    pix = d[4:].astype(np.int32)
    seq = pix[np.arange(nc * h * w) % len(pix)] - (128 if sg else 0)
    img = seq.reshape(nc, h, w) if planar else seq.reshape(h, nc, w).transpose(1, 0, 2)
    return np.ascontiguousarray(img), dict(bit_depth=bd, is_signed=sg, reversible=rev, num_decomps=nd, color_transform=ct,
                                           planar=planar, qstep=-1.0 if rev else 0.0005)
