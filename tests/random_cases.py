"""Seeded random parameter sets over everything the path supports: image / tile geometry on the
reference grid, sub-sampling, component formats, precincts, block sizes, progression orders,
tile-parts, both wavelets.  Used by the CPU parity test (against the live reference) and by the GPU
test (against the oracle pipeline)."""
import numpy as np


def random_case(seed):
    """-> (planes, kwargs for plan.make_params / refbind.Ref.encode / cpu_pipeline.encode, (W, H))"""
    rng = np.random.default_rng(1000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 160))
    if rng.random() < 0.15:
        w = pick([1, 2, 3])
    if rng.random() < 0.15:
        h = pick([1, 2, 3])
    nc = pick([1, 1, 2, 3, 3, 4])
    reversible = bool(rng.random() < 0.6)
    color = nc >= 3 and rng.random() < 0.4
    ds = []
    for c in range(nc):
        if color and c < 3:
            ds.append((1, 1))
        else:
            ds.append(pick([(1, 1), (1, 1), (2, 2), (2, 1), (1, 2), (3, 1)]))
    if color:                              # interleaved line exchange: every component as tall as the first
        ds = [(dx, 1) for dx, _ in ds]
        ds[:3] = [ds[0]] * 3
    depths = [pick([8, 8, 10, 12, 16, 5]) for _ in range(nc)]
    signs = [bool(rng.random() < 0.2) for _ in range(nc)]
    if color:
        depths[:3] = [depths[0]] * 3
        signs[:3] = [signs[0]] * 3
    ox, oy = (int(rng.integers(0, 40)), int(rng.integers(0, 40))) if rng.random() < 0.4 else (0, 0)
    kw = dict(reversible=reversible, color_transform=color, num_decomps=int(rng.integers(0, 6)),
              block=pick([(64, 64), (32, 32), (16, 64), (128, 32), (4, 256), (8, 8)]),
              prog_order=pick(["LRCP", "RLCP", "RPCL", "PCRL", "CPRL"]), image_offset=(ox, oy),
              bit_depth=depths[0], is_signed=signs[0], downsampling=ds, bit_depths=depths, signs=signs)
    if rng.random() < 0.5:
        tw, th = int(rng.integers(17, 120)), int(rng.integers(17, 120))
        kw["tile"] = (tw, th)
        if ox or oy:
            kw["tile_offset"] = (int(rng.integers(0, ox + 1)), int(rng.integers(0, oy + 1)))
            # the first tile has to reach into the image
            kw["tile"] = (max(tw, ox - kw["tile_offset"][0] + 1), max(th, oy - kw["tile_offset"][1] + 1))
    if rng.random() < 0.4 and kw["num_decomps"] > 0:
        kw["precincts"] = [(pick([32, 64, 128]), pick([32, 64, 128])) for _ in range(int(rng.integers(1, 4)))]
    if rng.random() < 0.3:
        kw["tileparts"] = pick(["R", "C", "RC"])
    if rng.random() < 0.3:
        kw["tlm"] = True
    if not reversible and rng.random() < 0.6:
        kw["qstep"] = float(pick([0.1, 0.02, 0.005]))
    planes = []
    for (dx, dy), bd, sg in zip(ds, depths, signs):
        cw = -(-(ox + w) // dx) - -(-ox // dx)
        ch = -(-(oy + h) // dy) - -(-oy // dy)
        lo, hi = (-(1 << (bd - 1)), 1 << (bd - 1)) if sg else (0, 1 << bd)
        yy, xx = np.mgrid[0:max(ch, 1), 0:max(cw, 1)]
        base = ((np.sin(xx / 5.0 + seed) + np.cos(yy / 3.0)) * 0.22 + 0.5) * (hi - lo) + lo
        q = np.clip(base + rng.integers(-(hi - lo) // 16 - 1, (hi - lo) // 16 + 2, base.shape), lo, hi - 1).astype(np.int32)
        planes.append(q[:ch, :cw])
    return planes, kw, (w, h)


def random_coc_case(seed):
    """random_case plus COC marker segments on a random subset of the components (their own
    decompositions, block size, precincts, wavelet), in a random creation order"""
    planes, kw, size = random_case(7000 + seed)
    rng = np.random.default_rng(99000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    nc = len(planes)
    order = [int(c) for c in rng.permutation(nc)][:int(rng.integers(1, nc + 1))]
    coc = {}
    for c in order:
        st = {}
        if rng.random() < 0.8:
            st["num_decomps"] = int(rng.integers(0, 6))
        if rng.random() < 0.5:
            st["block"] = pick([(64, 64), (32, 32), (16, 64), (128, 32), (8, 8)])
        if rng.random() < 0.3 and st.get("num_decomps", 5) > 0:
            st["precincts"] = [(pick([32, 64, 128]), pick([32, 64, 128])) for _ in range(int(rng.integers(1, 3)))]
        if rng.random() < 0.7:
            # mostly the wavelet the colour transform needs on its three components
            st["reversible"] = kw["reversible"] if (kw["color_transform"] and c < 3 and rng.random() < 0.9) else bool(rng.random() < 0.5)
        elif kw["color_transform"] and c < 3 and kw["reversible"]:
            st["reversible"] = True                        # (a COC starts from the 9/7)
        if not st:                                         # a COC exists once one of its setters was called
            st["num_decomps"] = int(rng.integers(0, 6))
        coc[c] = st
    kw["coc"] = coc
    return planes, kw, size
