"""CPU model of the whole encode / decode path, glued from the ORACLE stages (oracle/) and the
product's host-side plan + Tier-2 (through the C ABI).  Test infrastructure only: it exists so
that (a) the oracle, the plan geometry and the Tier-2 writer/parser can be pinned against the
real reference codestream byte for byte without a GPU, and (b) the GPU stages have an exact
stage-by-stage expectation on the GPU box.
"""
import numpy as np

from openjph_amd.plan import Plan, make_params, parse_codestream, coded_dtype
from oracle import oraclebind as ob


def _view(arena, off, pitch, w, h, dtype):
    """a plane of the arena (offsets in 32-bit elements, pitch in samples); int64: the 64-bit sample path"""
    size = np.dtype(dtype).itemsize
    a = arena.view(dtype)
    return np.lib.stride_tricks.as_strided(a[off * 4 // size:], shape=(h, w), strides=(pitch * size, size))


def _wrap32(a):
    """two's-complement wrap to 32 bits, as the reference's si32 arithmetic does"""
    return a.astype(np.int64).astype(np.uint32).astype(np.int32) if a.dtype != np.int32 else a


def _plane_dtype(rev, wide):
    return (np.int64 if wide else np.int32) if rev else np.float32


def forward_stages(plan: Plan, image: np.ndarray, tiles=None):
    """image: int32 [C,H,W], or a list of per-component planes (sub-sampled components).  Returns
    the arena (uint32) after convert + all DWT levels.
    tiles=(first, count) restricts the work to a run of tiles (sharding tests)."""
    p = plan.params
    comps = [plan.comp_info(c) for c in range(p.num_comps)]
    revs = [plan.comp_style(c)["reversible"] for c in range(p.num_comps)]      # per component (COC)
    nlt3 = [plan.comp_style(c)["nlt3"] for c in range(p.num_comps)]
    wides = [plan.comp_style(c)["wide"] for c in range(p.num_comps)]
    gens = [plan.comp_style(c)["general"] for c in range(p.num_comps)]
    arena = np.zeros(plan.arena_elems, np.uint32)
    lib = ob.lib()
    t_first, t_count = (0, plan.num_tiles) if tiles is None else tiles
    for t in range(t_first, t_first + t_count):
        planes = []
        for c in range(p.num_comps):
            off, pitch, (x0, y0, w, h) = plan.comp_plane(t, c)
            x0 -= comps[c]["x0"]; y0 -= comps[c]["y0"]           # position inside the component's own plane
            src = np.ascontiguousarray(image[c][y0:y0 + h, x0:x0 + w], dtype=np.int32)
            bd, sg = plan.comp_format(c)
            if revs[c]:
                # gen_rev_convert / _nlt_type3 (ojph_colour.cpp:238-311).  A component on the 64-bit sample path gets 64-bit
                # lines -- unless it goes through the colour transform, whose input lines are always 32 bits wide
                # (ojph_tile.cpp:312-322) and which widens itself (gen_rct_forward :467-489)
                shift = 0 if sg else -(1 << (bd - 1))
                s64 = src.astype(np.int64)
                if nlt3[c]:
                    dst = np.where(s64 >= 0, s64, -s64 - ((1 << (bd - 1)) + 1))
                else:
                    dst = s64 + shift
                if not wides[c] or (p.color_transform and c < 3):
                    dst = dst.astype(np.uint64).astype(np.uint32).astype(np.int32)      # 32-bit arithmetic wraps
            else:
                if nlt3[c]:                                      # irv :406-412
                    src = np.where(src >= 0, src, -src - ((1 << (bd - 1)) + 1)).astype(np.int32)
                dst = np.empty(src.shape, np.float32)
                lib.ojo_irv_to_float(src.ctypes.data, dst.ctypes.data, src.size, bd, int(sg))
            planes.append((off, pitch, w, h, dst))
        if p.color_transform:
            r, g, b = [np.ascontiguousarray(pl[4]) for pl in planes[:3]]
            odt = np.int64 if (revs[0] and wides[0]) else r.dtype
            y = np.empty(r.shape, odt); cb = np.empty(r.shape, odt); cr = np.empty(r.shape, odt)
            f = (lib.ojo_rct_fwd64 if wides[0] else lib.ojo_rct_fwd) if revs[0] else lib.ojo_ict_fwd
            f(r.ctypes.data, g.ctypes.data, b.ctypes.data, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, r.size)
            for i, v in enumerate((y, cb, cr)):
                planes[i] = planes[i][:4] + (v,)
        for c, (off, pitch, w, h, v) in enumerate(planes):
            _view(arena, off, pitch, w, h, _plane_dtype(revs[c], wides[c]))[:] = v
    for lv in plan.levels:
        w, h = int(lv["w"]), int(lv["h"])
        if w == 0 or h == 0 or not (t_first <= int(lv["tile"]) < t_first + t_count):
            continue
        rev = revs[int(lv["comp"])]
        dt = _plane_dtype(rev, wides[int(lv["comp"])])
        src = np.ascontiguousarray(_view(arena, int(lv["src_off"]), int(lv["src_pitch"]), w, h, dt))
        xe, ye = bool(lv["x_even"]), bool(lv["y_even"])
        if gens[int(lv["comp"])]:                                # 64-bit samples (gen_rev_vert_step64 / horz_ana64, ojph_transform.cpp:261,415),
            k = plan.comp_lift(int(lv["comp"]), plan.comp_style(int(lv["comp"]))["num_decomps"] - int(lv["res"]) + 1)   # ATK kernels, DFS levels
            ll, hl, lh, hh = ob.dwt_fwd_gen(src, k["steps"], k["K"], k["horz"], k["vert"], xe, ye)
        else:
            ll, hl, lh, hh = (ob.dwt53_fwd if rev else ob.dwt97_fwd)(src, xe, ye)
        for name, b in (("ll", ll), ("hl", hl), ("lh", lh), ("hh", hh)):
            if b.size:
                _view(arena, int(lv[name + "_off"]), int(lv[name + "_pitch"]), b.shape[1], b.shape[0], dt)[:] = b
    return arena


def quantise_block(plan, arena, k):
    """Returns (sign-magnitude block [h, stride] -- uint32, or uint64 on the 64-bit sample path --, OR of the magnitudes)
    for block index k."""
    blk = plan.blocks[k]
    band = plan.bands[int(blk["band"])]
    w, h = int(blk["w"]), int(blk["h"])
    off = int(band["plane_off"]) + int(blk["y0"]) * int(band["pitch"]) + int(blk["x0"]) * (2 if _is_wide(plan, band) else 1)
    rev = plan.comp_style(int(band["comp"]))["reversible"]
    if _is_wide(plan, band):
        raw = np.ascontiguousarray(_view(arena, int(band["plane_off"]), int(band["pitch"]), int(band["w"]), int(band["h"]), np.int64)
                                   [int(blk["y0"]):int(blk["y0"]) + h, int(blk["x0"]):int(blk["x0"]) + w])
        q = np.empty(raw.shape, np.uint64)
        mx = ob.lib().ojo_quant_rev64(raw.ctypes.data, q.ctypes.data, raw.size, int(band["K_max"]))
        return q, int(mx)
    raw = np.ascontiguousarray(_view(arena, off, int(band["pitch"]), w, h, np.int32 if rev else np.float32))
    if rev:
        q, mx = ob.quant_rev(raw, int(band["K_max"]))
    else:
        q, mx = ob.quant_irv(raw, float(band["delta_inv"]))
    return q, mx


def _is_wide(plan, band):
    return plan.comp_style(int(band["comp"]))["wide"]


def encode_blocks(plan: Plan, arena, tiles=None):
    """HT-encodes every block (of the tile run) with the oracle. Returns (data bytes, coded table)."""
    coded = np.zeros(plan.num_blocks, coded_dtype)
    chunks = []
    pos = 0
    t_first, t_count = (0, plan.num_tiles) if tiles is None else tiles
    for k in range(plan.num_blocks):
        blk = plan.blocks[k]
        band = plan.bands[int(blk["band"])]
        if not (t_first <= int(band["tile"]) < t_first + t_count):
            continue
        K = int(blk["K_max"])
        q, mx = quantise_block(plan, arena, k)
        wide = q.dtype == np.uint64
        if mx >= (1 << ((63 if wide else 31) - K)):     # ojph_codeblock.cpp:146-175
            w, h = int(blk["w"]), int(blk["h"])
            b = (ob.ht_encode64 if wide else ob.ht_encode)(q, w, h, w, K - 1)
            assert len(b) > 0
            coded[k] = (pos, len(b), 0, K - 1, 1)
            chunks.append(b); pos += len(b)
    data = np.frombuffer(b"".join(chunks) if chunks else b"", dtype=np.uint8)
    return data, coded


def encode(image, size=None, **kw):
    """image: int32 [C,H,W], or a list of per-component planes with size=(W, H) on the reference grid"""
    if isinstance(image, (list, tuple)):
        nc = len(image)
        w, h = size if size is not None else (image[0].shape[1], image[0].shape[0])
    else:
        image = np.ascontiguousarray(image, dtype=np.int32)
        nc, h, w = image.shape
    plan = Plan(make_params(w, h, nc, **kw))
    arena = forward_stages(plan, image)
    data, coded = encode_blocks(plan, arena)
    return plan.t2_write(data, coded), plan, arena, data, coded


def encode_tiles(plan: Plan, image, first, count):
    """Oracle pipeline over a run of tiles -> (tile-part bytes, Psot per tile)."""
    arena = forward_stages(plan, image, (first, count))
    data, coded = encode_blocks(plan, arena, (first, count))
    return plan.t2_write_tiles(data, coded, first, count)


def add_refinement(data, coded, rng, fraction=0.6):
    """Turns a cleanup-only block table into one where a share of the blocks also carries SigProp
    (+ MagRef) segments made of random bytes -- what a foreign encoder may emit and the reference's
    own never does.  The refinement bytes follow the cleanup bytes (ojph_block_decoder32.cpp:742)."""
    out = coded.copy()
    chunks, pos = [], 0
    for k in range(len(coded)):
        n1 = int(coded[k]["len1"])
        if n1 == 0:
            continue
        o = int(coded[k]["offset"])
        chunks.append(data[o:o + n1].tobytes())
        out[k]["offset"] = pos
        pos += n1
        if rng.random() < fraction:
            n2 = int(rng.integers(1, 300))
            chunks.append(bytes(rng.integers(0, 256, size=n2, dtype=np.uint8)))
            out[k]["len2"] = n2
            out[k]["num_passes"] = int(rng.integers(2, 4))
            pos += n2
    return np.frombuffer(b"".join(chunks), dtype=np.uint8), out


def decode_blocks(plan: Plan, cs: bytes, resilient=False):
    """Oracle HT decode + dequantise of every block into a fresh arena (blocks of resolutions the
    plan was told not to read stay zero).  resilient: a block the decoder refuses stays zero
    (ojph_codeblock.cpp:214-224) instead of raising."""
    coded = plan.coded_blocks()
    nc = int(plan.params.num_comps)
    styles = [plan.comp_style(c) for c in range(nc)]
    # resolutions read: counted from the component's own top (ojph_resolution.cpp:254-255)
    top_read = [st["num_decomps"] - plan.skip[0] for st in styles]
    arena = np.zeros(plan.arena_elems, np.uint32)
    buf = np.frombuffer(cs, dtype=np.uint8)
    # blocks whose tile-part ran out of bytes: the reference decodes what there is, padded with zeros
    # (ojphgpu_plan_padded_blocks; bb_read_chunk, ojph_bitbuffer_read.h:134-150)
    padded = {int(q["block"]): q for q in plan.padded_blocks()}
    for k in range(plan.num_blocks):
        cbk = coded[k]
        pad = 0
        if k in padded:
            cbk = padded[k]; pad = int(cbk["len1"]) + int(cbk["len2"]) - int(cbk["got"])
        if cbk["len1"] == 0:
            continue
        blk = plan.blocks[k]
        band = plan.bands[int(blk["band"])]
        if int(band["res"]) > top_read[int(band["comp"])]:
            continue
        rev = styles[int(band["comp"])]["reversible"]
        w, h = int(blk["w"]), int(blk["h"])
        o = int(cbk["offset"]); n = int(cbk["len1"]) + int(cbk["len2"])
        wide = styles[int(band["comp"])]["wide"]
        ok, sm = (ob.ht_decode64 if wide else ob.ht_decode)(buf[o:o + n - pad].tobytes() + bytes(pad), w, h, w, int(cbk["missing_msbs"]),
                                                          len2=int(cbk["len2"]), num_passes=int(cbk["num_passes"]),
                                                          stripe_causal=bool(plan.params.reserved[0] & 1))
        if not ok:
            if resilient:
                continue
            raise RuntimeError("oracle failed to decode block %d" % k)
        off = int(band["plane_off"]) + int(blk["y0"]) * int(band["pitch"]) + int(blk["x0"])
        if wide:                                                 # gen_rev_tx_from_cb64 (ojph_codestream_gen.cpp:140-153)
            sm = np.ascontiguousarray(sm)
            dq = np.empty(sm.shape, np.int64)
            ob.lib().ojo_dequant_rev64(sm.ctypes.data, dq.ctypes.data, sm.size, int(band["K_max"]))
            _view(arena, int(band["plane_off"]), int(band["pitch"]), int(band["w"]), int(band["h"]), np.int64)[
                int(blk["y0"]):int(blk["y0"]) + h, int(blk["x0"]):int(blk["x0"]) + w] = dq
        elif rev:
            _view(arena, off, int(band["pitch"]), w, h, np.int32)[:] = ob.dequant_rev(sm, int(band["K_max"]))
        else:
            _view(arena, off, int(band["pitch"]), w, h, np.float32)[:] = ob.dequant_irv(sm, float(band["delta"]))
    return arena


def inverse_stages(plan: Plan, arena):
    p = plan.params
    styles = [plan.comp_style(c) for c in range(p.num_comps)]
    revs = [st["reversible"] for st in styles]
    lib = ob.lib()
    order = sorted(range(len(plan.levels)), key=lambda i: int(plan.levels[i]["res"]))
    for lv in (plan.levels[i] for i in order):
        w, h = int(lv["w"]), int(lv["h"])
        if w == 0 or h == 0 or int(lv["res"]) > styles[int(lv["comp"])]["recon_decomps"]:   # the resolution that is reconstructed
            continue
        rev = revs[int(lv["comp"])]
        dt = _plane_dtype(rev, styles[int(lv["comp"])]["wide"])
        xe, ye = bool(lv["x_even"]), bool(lv["y_even"])
        lw, hw, lh_, hh_ = ob.band_dims(w, h, xe, ye)
        kind = int(lv["kind"])                                   # DFS: a level may leave a direction alone
        if kind in (0, 3):
            lw, hw = w, 0
        if kind in (0, 2):
            lh_, hh_ = h, 0
        ll = _view(arena, int(lv["ll_off"]), int(lv["ll_pitch"]), lw, lh_, dt)
        hl = _view(arena, int(lv["hl_off"]), int(lv["hl_pitch"]), hw, lh_, dt)
        lh = _view(arena, int(lv["lh_off"]), int(lv["lh_pitch"]), lw, hh_, dt)
        hh = _view(arena, int(lv["hh_off"]), int(lv["hh_pitch"]), hw, hh_, dt)
        if styles[int(lv["comp"])]["general"]:
            k = plan.comp_lift(int(lv["comp"]), styles[int(lv["comp"])]["num_decomps"] - int(lv["res"]) + 1)
            dst = ob.dwt_inv_gen(ll, hl, lh, hh, w, h, k["steps"], k["K"], k["horz"], k["vert"], xe, ye)
        else:
            dst = (ob.dwt53_inv if rev else ob.dwt97_inv)(ll, hl, lh, hh, w, h, xe, ye)
        _view(arena, int(lv["src_off"]), int(lv["src_pitch"]), w, h, dt)[:] = dst
    comps = [plan.comp_info(c) for c in range(p.num_comps)]
    image = [np.zeros((ci["h"], ci["w"]), np.int32) for ci in comps]
    for t in range(plan.num_tiles):
        planes = []
        for c in range(p.num_comps):
            off, pitch, (x0, y0, w, h) = plan.comp_plane(t, c)
            planes.append(np.ascontiguousarray(_view(arena, off, pitch, w, h, _plane_dtype(revs[c], styles[c]["wide"]))))
        if p.color_transform:
            y, cb, cr = planes[:3]
            odt = np.int32 if revs[0] else np.float32            # the colour transform's output lines are 32 bits wide
            r = np.empty(y.shape, odt); g = np.empty(y.shape, odt); b = np.empty(y.shape, odt)
            f = (lib.ojo_rct_inv64 if styles[0]["wide"] else lib.ojo_rct_inv) if revs[0] else lib.ojo_ict_inv
            f(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, r.ctypes.data, g.ctypes.data, b.ctypes.data, y.size)
            planes[:3] = [r, g, b]
        for c in range(p.num_comps):
            off, pitch, (x0, y0, w, h) = plan.comp_plane(t, c)
            v = planes[c]
            bd, sg = plan.comp_format(c)
            if revs[c]:
                v64 = v.astype(np.int64)                         # gen_rev_convert / _nlt_type3 on the way out (ojph_tile.cpp:439-465)
                if styles[c]["nlt3"]:
                    out = np.where(v64 >= 0, v64, -v64 - ((1 << (bd - 1)) + 1))
                else:
                    out = v64 + (0 if sg else (1 << (bd - 1)))
                out = out.astype(np.uint64).astype(np.uint32).astype(np.int32)          # (si32) / 32-bit arithmetic
            else:
                out = np.empty(v.shape, np.int32)
                lib.ojo_irv_to_int(v.ctypes.data, out.ctypes.data, v.size, bd, int(sg))
                if styles[c]["nlt3"]:                            # the same mapping on the way out (ojph_tile.cpp:446-448)
                    out = np.where(out >= 0, out, -out - ((1 << (bd - 1)) + 1)).astype(np.int32)
            x0 -= comps[c]["x0"]; y0 -= comps[c]["y0"]
            image[c][y0:y0 + h, x0:x0 + w] = out
    return np.stack(image) if len(plan.frame_shape) == 3 else image


def decode(cs: bytes, skip=None):
    """skip = (skipped_res_for_data, skipped_res_for_recon): reduced-resolution decoding"""
    plan = parse_codestream(cs)
    if skip:
        plan.restrict_resolution(*skip)
    arena = decode_blocks(plan, cs)
    return inverse_stages(plan, arena), plan
