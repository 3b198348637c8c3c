"""Host-side plan + Tier-2 wrappers over the C ABI (numpy views of the plan tables)."""
import ctypes as C
import numpy as np

from . import capi
from .capi import Params, BandInfo, BlockInfo, LevelInfo, CodedBlock, PaddedBlock, check, PROG_ORDERS

band_dtype = np.dtype(BandInfo)
block_dtype = np.dtype(BlockInfo)
level_dtype = np.dtype(LevelInfo)
coded_dtype = np.dtype(CodedBlock)
padded_dtype = np.dtype(PaddedBlock)


def make_params(width, height, num_comps=1, bit_depth=8, is_signed=False, reversible=True,
                num_decomps=5, block=(64, 64), color_transform=False, tile=(0, 0),
                prog_order="RPCL", qstep=-1.0, precinct=(0, 0), tlm=False, precincts=None,
                downsampling=None, image_offset=(0, 0), tile_offset=(0, 0), tileparts="", bit_depths=None, signs=None,
                qfactor=0, coc=None, nlt=None, qfactors=None, atk=None, dfs=None, wavelet=0):
    """qfactors: {component: (ctype, qfactor)} with ctype "Y", "Cb" or "Cr" -- param_qcd::set_qfactor(comp_idx,
    ctype, qfactor) calls in the order of the dict (a QCC per named component).
    nlt: {component or "all": 0 or 3} -- param_nlt::set_nonlinear_transform calls in the order of the
    dict ("all" = the ALL_COMPS entry; 3 = binary complement <-> sign magnitude, 0 = none).
    coc: {component: dict(reversible=, num_decomps=, block=(w, h), precincts=[(w, h), ...])} -- COC
    marker segments in the order of the dict (param_cod's comp_idx setters); whatever a dict leaves
    out keeps the reference's COC defaults (9/7, 5 decompositions, 64x64, no precincts), NOT the COD's.
    Part 2 (the reference only READS these; this library also writes them, which is where the test codestreams come
    from): atk = {index 2..255: dict(steps=[(a, b, e), ...] for a reversible kernel or [A, ...] + K= for an irreversible
    one, synthesis order)} -- ATK marker segments; dfs = {index 0..15: [kind per decomposition level, 1 = the first one
    applied: 1 both directions, 2 horizontal, 3 vertical, 0 none]} -- DFS marker segments; wavelet = the ATK index the
    COD names (0: `reversible` decides); a coc entry may carry wavelet= and dfs= (a DFS index; the number of
    decompositions is then the COD's).
    width/height: the image SIZE (the reference's extent is offset + size); downsampling: list of
    (dx, dy) per component (param_siz::set_component), default 1,1; tileparts: "", "R", "C" or "RC"
    (codestream::set_tilepart_divisions); bit_depths / signs: per-component lists where they differ
    from bit_depth / is_signed; qfactor: 1..100 (param_qcd::set_qfactor), 0 = not set."""
    p = Params()
    p.width, p.height, p.num_comps = width, height, num_comps
    p.bit_depth, p.is_signed = bit_depth, int(is_signed)
    p.reversible, p.num_decomps = int(reversible), num_decomps
    p.block_w, p.block_h = block
    p.color_transform = int(color_transform)
    p.tile_w, p.tile_h = tile
    p.prog_order = PROG_ORDERS[prog_order] if isinstance(prog_order, str) else int(prog_order)
    p.qstep = float(qstep)
    p.precinct_w, p.precinct_h = precinct
    if precincts:                          # list of (w, h) from the lowest resolution up, last one repeated
        for i in range(num_decomps + 1):
            pw, ph = precincts[min(i, len(precincts) - 1)]
            p.precinct_exps[i] = (int(pw).bit_length() - 1) | ((int(ph).bit_length() - 1) << 4)
    p.tlm = int(tlm)
    p.reserved[1] = (1 if "R" in tileparts else 0) | (2 if "C" in tileparts else 0)
    p.reserved[2] = int(qfactor)
    for c, bd in enumerate(bit_depths or []):
        p.comp_depth[c] = int(bd)
    for c, sg in enumerate(signs or []):
        p.comp_sign[c] = 2 if sg else 1
    p.image_x0, p.image_y0 = image_offset
    p.tile_x0, p.tile_y0 = tile_offset
    p.wavelet = int(wavelet)
    for i, (idx, a) in enumerate((atk or {}).items()):
        k = p.atk[i]
        steps = a["steps"]
        k.index, k.num_steps = int(idx), len(steps)
        k.reversible = int(isinstance(steps[0], (tuple, list)))
        k.coeff_type = int(a.get("coeff_type", 1 if k.reversible else 2))
        k.K = float(a.get("K", 1.0))
        for j, st in enumerate(steps):
            if k.reversible:
                k.steps[j].a, k.steps[j].b, k.steps[j].e = [int(v) for v in st]
            else:
                k.steps[j].A = float(st)
    for i, (idx, kinds) in enumerate((dfs or {}).items()):
        f = p.dfs[i]
        f.used, f.index, f.num_levels = 1, int(idx), len(kinds)
        for j, kd in enumerate(kinds):
            f.types[j] = int(kd)
    for rank, (c, st) in enumerate((coc or {}).items()):
        if not 0 <= int(c) < 16:
            raise ValueError("per-component coding styles can be given for the first 16 components")
        k = p.coc[int(c)]
        k.rank = rank + 1
        k.reversible = int(bool(st.get("reversible", False)))
        k.num_decomps = int(st.get("num_decomps", 5))
        if st.get("dfs") is not None:
            k.reserved[0] |= 0x80 | (int(st["dfs"]) << 1)
        k.reserved[1] = int(st.get("wavelet", 0))
        bw, bh = st.get("block", (64, 64))
        k.log_block_w, k.log_block_h = int(bw).bit_length() - 1, int(bh).bit_length() - 1
        pl = st.get("precincts")
        if pl:
            k.has_precincts = 1
            for i in range(k.num_decomps + 1):
                pw, ph = pl[min(i, len(pl) - 1)]
                k.precinct_exps[i] = (int(pw).bit_length() - 1) | ((int(ph).bit_length() - 1) << 4)
    for rank, (c, (ctype, qf)) in enumerate((qfactors or {}).items()):
        if not 0 <= int(c) < 16:
            raise ValueError("per-component quality factors can be given for the first 16 components")
        p.qcc_qfactor[int(c)], p.qcc_ctype[int(c)], p.qcc_rank[int(c)] = int(qf), ("Y", "Cb", "Cr").index(ctype), rank + 1
    rank = 0
    for c, t in (nlt or {}).items():
        if int(t) not in (0, 3):
            raise ValueError("only the non-linearity types 0 and 3 exist in the reference")
        if c == "all":
            p.nlt_default = int(t) + 1
            continue
        if not 0 <= int(c) < 16:
            raise ValueError("NLT entries can be given for the first 16 components")
        if p.nlt_comp[int(c)] == 0:
            rank += 1
            p.nlt_rank[int(c)] = rank
        p.nlt_comp[int(c)] = int(t) + 1
    if downsampling:
        if len(downsampling) > 16:
            raise ValueError("sub-sampling factors can be given for the first 16 components")
        for c, (dx, dy) in enumerate(downsampling):
            p.comp_dx[c], p.comp_dy[c] = int(dx), int(dy)
    return p


class Plan:
    """Owns an ojphgpu_plan*. Tables are exposed as numpy structured arrays."""

    def __init__(self, params=None, handle=None, owned=True):
        self._lib = capi.lib()
        self._owned = owned                # False: the handle belongs to another object (a decoder pipe)
        if handle is None:
            h = C.c_void_p()
            check(self._lib.ojphgpu_plan_create(C.byref(params), C.byref(h)), "plan_create")
            handle = h
        self.handle = handle
        self.skip = (0, 0)
        cnt = (C.c_uint64 * 8)()
        check(self._lib.ojphgpu_plan_counts(self.handle, cnt))
        (self.num_tiles, self.num_bands, self.num_blocks, self.num_levels, self.arena_elems,
         self.max_block_bytes, self.num_precincts, self.num_tcomps) = [int(v) for v in cnt]
        self.params = Params()
        check(self._lib.ojphgpu_plan_params(self.handle, C.byref(self.params)))
        ppt = C.c_uint32()
        check(self._lib.ojphgpu_plan_tile_parts(self.handle, C.byref(ppt)))
        self.parts_per_tile = int(ppt.value)
        self.bands = np.zeros(self.num_bands, band_dtype)
        self.blocks = np.zeros(self.num_blocks, block_dtype)
        self.levels = np.zeros(self.num_levels, level_dtype)
        if self.num_bands:
            check(self._lib.ojphgpu_plan_bands(self.handle, self.bands.ctypes.data, self.num_bands))
        if self.num_blocks:
            check(self._lib.ojphgpu_plan_blocks(self.handle, self.blocks.ctypes.data, self.num_blocks))
        if self.num_levels:
            check(self._lib.ojphgpu_plan_levels(self.handle, self.levels.ctypes.data, self.num_levels))

    def __del__(self):
        try:
            if self.handle and self._owned:
                self._lib.ojphgpu_plan_destroy(self.handle)
            self.handle = None
        except Exception:
            pass

    def set_comments(self, comments):
        """user COM segments of the main header: a list of str (Latin text, Rcom = 1) or bytes (Rcom = 0)"""
        n = len(comments)
        raw = [c.encode("latin-1") if isinstance(c, str) else bytes(c) for c in comments]
        data = (C.c_char_p * max(n, 1))(*raw)
        lens = (C.c_uint16 * max(n, 1))(*[len(b) for b in raw])
        rcom = (C.c_uint16 * max(n, 1))(*[1 if isinstance(c, str) else 0 for c in comments])
        check(self._lib.ojphgpu_plan_set_comments(self.handle, data, lens, rcom, n), "plan_set_comments")

    def restrict_resolution(self, skipped_res_for_data, skipped_res_for_recon=None):
        """codestream::restrict_input_resolution on a parsed plan (before a Decoder is created from it)"""
        if skipped_res_for_recon is None:
            skipped_res_for_recon = skipped_res_for_data
        check(self._lib.ojphgpu_plan_restrict_resolution(self.handle, int(skipped_res_for_data), int(skipped_res_for_recon)),
              "plan_restrict_resolution")
        self.skip = (int(skipped_res_for_data), int(skipped_res_for_recon))

    def comp_info(self, comp):
        """-> dict(x0, y0, w, h, frame_off, dx, dy) of component `comp` (see ojphgpu_plan_comp_info)"""
        out = (C.c_uint32 * 8)()
        check(self._lib.ojphgpu_plan_comp_info(self.handle, comp, out))
        v = [int(x) for x in out]
        return dict(x0=v[0], y0=v[1], w=v[2], h=v[3], frame_off=v[4] | (v[5] << 32), dx=v[6], dy=v[7])

    def comp_format(self, comp):
        """-> (bit depth, is_signed) of component `comp`"""
        bd, sg = C.c_uint32(), C.c_uint32()
        check(self._lib.ojphgpu_plan_comp_format(self.handle, comp, C.byref(bd), C.byref(sg)))
        return int(bd.value), bool(sg.value)

    def comp_style(self, comp):
        """coding style of component `comp` (its COC, else the COD): dict(num_decomps, reversible,
        log_block=(w, h), has_coc, recon_decomps = levels left after restrict_resolution, nlt3 = the type 3
        non-linearity applies, wide = the component takes the 64-bit sample path: int64 planes of two arena elements per
        sample)"""
        out = (C.c_uint32 * 8)()
        check(self._lib.ojphgpu_plan_comp_style(self.handle, comp, out))
        return dict(num_decomps=int(out[0]), reversible=bool(out[1]), log_block=(int(out[2]), int(out[3])),
                    has_coc=bool(out[4]), recon_decomps=int(out[5]), nlt3=bool(out[6]), wide=bool(out[7] & 1),
                    general=bool(out[7] & 2))

    def comp_lift(self, comp, level):
        """the wavelet of decomposition level `level` (1 = the first applied) of a component, for the general lifting
        form: dict(steps=[(a, b, e) | A, ...] in synthesis order, K, elem (0 int32, 1 int64, 2 float), horz, vert)"""
        k = capi.Lift()
        check(self._lib.ojphgpu_plan_comp_lift(self.handle, comp, level, C.byref(k)))
        rev = k.elem != 2
        steps = [(k.steps[i].a, k.steps[i].b, k.steps[i].e) if rev else float(k.steps[i].A) for i in range(k.num_steps)]
        return dict(steps=steps, K=float(k.K), elem=int(k.elem), horz=bool(k.horz), vert=bool(k.vert))

    @property
    def frame_elems(self):
        out = (C.c_uint32 * 8)()
        check(self._lib.ojphgpu_plan_comp_info(self.handle, int(self.params.num_comps), out))
        return int(out[4]) | (int(out[5]) << 32)

    @property
    def frame_shape(self):
        """[C,H,W] when every component has the same size, else the flat (frame_elems,)"""
        ci = [self.comp_info(c) for c in range(int(self.params.num_comps))]
        if all((c["w"], c["h"]) == (ci[0]["w"], ci[0]["h"]) for c in ci):
            return (len(ci), ci[0]["h"], ci[0]["w"])
        return (self.frame_elems,)

    def pack_frame(self, planes):
        """list of per-component 2-D arrays -> the frame layout every codec call takes"""
        if len(self.frame_shape) == 3:
            return np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.int32) for q in planes]))
        out = np.empty(self.frame_elems, np.int32)
        for c, q in enumerate(planes):
            i = self.comp_info(c)
            assert tuple(np.shape(q)) == (i["h"], i["w"]), "component %d must be %dx%d" % (c, i["h"], i["w"])
            out[i["frame_off"]:i["frame_off"] + i["w"] * i["h"]] = np.asarray(q, dtype=np.int32).ravel()
        return out

    def unpack_frame(self, frame):
        """frame (numpy, either layout) -> list of per-component 2-D arrays"""
        flat = np.asarray(frame).reshape(-1)
        out = []
        for c in range(int(self.params.num_comps)):
            i = self.comp_info(c)
            out.append(flat[i["frame_off"]:i["frame_off"] + i["w"] * i["h"]].reshape(i["h"], i["w"]))
        return out

    def comp_plane(self, tile, comp):
        off, pitch, rect = C.c_uint64(), C.c_uint32(), (C.c_uint32 * 4)()
        check(self._lib.ojphgpu_plan_comp_plane(self.handle, tile, comp, C.byref(off), C.byref(pitch), rect))
        return int(off.value), int(pitch.value), tuple(int(v) for v in rect)

    def coded_blocks(self):
        out = np.zeros(self.num_blocks, coded_dtype)
        check(self._lib.ojphgpu_plan_coded_blocks(self.handle, out.ctypes.data, self.num_blocks))
        return out

    def padded_blocks(self):
        """ojphgpu_plan_padded_blocks: (block, got, offset, len1, len2, missing_msbs, num_passes) of the blocks a damaged
        codestream's packet headers promise more bytes for than their tile-part holds"""
        n = C.c_size_t()
        check(self._lib.ojphgpu_plan_padded_blocks(self.handle, None, 0, C.byref(n)))
        out = np.zeros(int(n.value), padded_dtype)
        if n.value:
            check(self._lib.ojphgpu_plan_padded_blocks(self.handle, out.ctypes.data, int(n.value), C.byref(n)))
        return out

    def t2_write(self, block_data: np.ndarray, coded: np.ndarray) -> bytes:
        """block_data: uint8 array; coded: coded_dtype array (offset/len1/...) in plan order."""
        block_data = np.ascontiguousarray(block_data, dtype=np.uint8)
        coded = np.ascontiguousarray(coded, dtype=coded_dtype)
        need = C.c_size_t()
        cap = int(block_data.size + 64 * self.num_blocks + 4096 * (self.num_tiles + 1) + (1 << 16))
        out = np.empty(cap, np.uint8)
        rc = self._lib.ojphgpu_t2_write(self.handle, block_data.ctypes.data, coded.ctypes.data,
                                        out.ctypes.data, cap, C.byref(need))
        if rc == capi.E_OVERFLOW:
            cap = int(need.value)
            out = np.empty(cap, np.uint8)
            rc = self._lib.ojphgpu_t2_write(self.handle, block_data.ctypes.data, coded.ctypes.data,
                                            out.ctypes.data, cap, C.byref(need))
        check(rc, "t2_write")
        return out[:need.value].tobytes()


    def t2_write_tiles(self, block_data: np.ndarray, coded: np.ndarray, tile_first: int, tile_count: int):
        """Tile-parts of tiles [tile_first, tile_first+tile_count) -> (bytes, Psot per tile)."""
        block_data = np.ascontiguousarray(block_data, dtype=np.uint8)
        coded = np.ascontiguousarray(coded, dtype=coded_dtype)
        lens = np.zeros(max(tile_count, 1) * self.parts_per_tile, np.uint32)
        need = C.c_size_t()
        rc = self._lib.ojphgpu_t2_write_tiles(self.handle, block_data.ctypes.data, coded.ctypes.data, tile_first,
                                              tile_count, None, 0, C.byref(need), lens.ctypes.data)
        if rc not in (capi.OK, capi.E_OVERFLOW):
            check(rc, "t2_write_tiles")
        out = np.empty(max(int(need.value), 1), np.uint8)
        check(self._lib.ojphgpu_t2_write_tiles(self.handle, block_data.ctypes.data, coded.ctypes.data, tile_first,
                                               tile_count, out.ctypes.data, out.size, C.byref(need), lens.ctypes.data),
              "t2_write_tiles")
        return out[:need.value].tobytes(), lens[:tile_count * self.parts_per_tile].copy()

    def t2_main_header(self, tile_part_len=None) -> bytes:
        """SOC .. end of the main header; tile_part_len (Psot of every tile) feeds the TLM marker."""
        lens = None if tile_part_len is None else np.ascontiguousarray(tile_part_len, dtype=np.uint32)
        if lens is not None and lens.size != self.num_tiles * self.parts_per_tile:
            raise ValueError("tile_part_len must have one entry per tile-part")
        need = C.c_size_t()
        ptr = None if lens is None else lens.ctypes.data
        rc = self._lib.ojphgpu_t2_write_main_header(self.handle, ptr, None, 0, C.byref(need))
        if rc not in (capi.OK, capi.E_OVERFLOW):
            check(rc, "t2_write_main_header")
        out = np.empty(int(need.value), np.uint8)
        check(self._lib.ojphgpu_t2_write_main_header(self.handle, ptr, out.ctypes.data, out.size, C.byref(need)),
              "t2_write_main_header")
        return out[:need.value].tobytes()


def parse_codestream(data: bytes, resilient=False, skip=None) -> Plan:
    """skip = (skipped_res_for_data, skipped_res_for_recon): restrict_input_resolution BEFORE the tile-parts are read, as the
    reference orders it (ojphgpu_t2_parse_restricted) -- on undamaged codestreams the same as Plan.restrict_resolution afterwards"""
    buf = np.frombuffer(data, dtype=np.uint8)
    h = C.c_void_p()
    if skip is None:
        check(capi.lib().ojphgpu_t2_parse(buf.ctypes.data, len(data), int(resilient), C.byref(h)), "t2_parse")
        return Plan(handle=h)
    check(capi.lib().ojphgpu_t2_parse_restricted(buf.ctypes.data, len(data), int(resilient), int(skip[0]), int(skip[1]), C.byref(h)), "t2_parse")
    pl = Plan(handle=h)
    pl.skip = (int(skip[0]), int(skip[1]))
    return pl
