"""ctypes binding to oracle/_ref/libojph_ref*.so (the real reference, built by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never from the product package (openjph_amd/).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PROG_ORDERS = {"LRCP": 0, "RLCP": 1, "RPCL": 2, "PCRL": 3, "CPRL": 4}


class TooLarge(ValueError):
    pass


class RefCoc(C.Structure):
    _fields_ = [("comp", C.c_uint8), ("mask", C.c_uint8), ("reversible", C.c_uint8), ("num_decomps", C.c_uint8),
                ("log_bw", C.c_uint8), ("log_bh", C.c_uint8), ("pad", C.c_uint8 * 2), ("precinct_exps", C.c_uint8 * 36)]


class RefCqf(C.Structure):
    _fields_ = [("comp", C.c_uint8), ("ctype", C.c_uint8), ("qfactor", C.c_uint8), ("pad", C.c_uint8)]


class RefNlt(C.Structure):
    _fields_ = [("comp", C.c_uint16), ("type", C.c_uint8), ("pad", C.c_uint8)]


class RefParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("num_comps", C.c_uint32),
        ("bit_depth", C.c_uint32), ("is_signed", C.c_uint32),
        ("reversible", C.c_uint32), ("num_decomps", C.c_uint32),
        ("block_w", C.c_uint32), ("block_h", C.c_uint32),
        ("color_transform", C.c_uint32),
        ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
        ("prog_order", C.c_uint32), ("planar", C.c_uint32),
        ("qstep", C.c_float),
        ("precinct_w", C.c_uint32), ("precinct_h", C.c_uint32),
        ("tlm", C.c_uint32),
        ("precinct_exps", C.c_uint8 * 36),
        ("image_x0", C.c_uint32), ("image_y0", C.c_uint32), ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32),
        ("comp_dx", C.c_uint8 * 16), ("comp_dy", C.c_uint8 * 16),
        ("tilepart_div", C.c_uint32),
        ("comp_depth", C.c_uint8 * 16), ("comp_sign", C.c_uint8 * 16), ("qfactor", C.c_uint32),
        ("coc", RefCoc * 16), ("num_coc", C.c_uint32),
        ("nlt", RefNlt * 17), ("num_nlt", C.c_uint32),
        ("cqf", RefCqf * 16), ("num_cqf", C.c_uint32),
    ]


def fnv1a64(data: bytes) -> str:
    h = 0xcbf29ce484222325
    for b in bytes(data):
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def available(generic=False):
    return os.path.exists(os.path.join(_HERE, "_ref",
                                       "libojph_refgen.so" if generic else "libojph_ref.so"))


class Ref:
    """The reference library. generic=True -> the -DOJPH_DISABLE_SIMD build."""

    def __init__(self, generic=False):
        name = "libojph_refgen.so" if generic else "libojph_ref.so"
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", name))
        L = self.lib
        L.ref_encode.restype = C.c_long
        L.ref_encode.argtypes = [C.POINTER(RefParams), C.POINTER(C.c_void_p), C.c_void_p, C.c_long]
        L.ref_encode_ex.restype = C.c_long
        L.ref_encode_ex.argtypes = [C.POINTER(RefParams), C.POINTER(C.c_void_p), C.c_void_p, C.c_long, C.c_char_p, C.c_char_p]
        L.ref_decode.restype = C.c_int
        L.ref_decode.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.ref_decode_skip.restype = C.c_int
        L.ref_decode_skip.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]
        L.ref_encode_block32.restype = C.c_long
        L.ref_encode_block32.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_void_p, C.c_long]
        L.ref_decode_block32.restype = C.c_int
        L.ref_decode_block32.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_int]
        L.ref_simd_level.restype = C.c_int
        if hasattr(L, "ref_encode_block64"):              # (a prebuilt library of an earlier round may lack them)
            L.ref_encode_block64.restype = C.c_long
            L.ref_encode_block64.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_long]
            L.ref_decode_block64.restype = C.c_int
            L.ref_decode_block64.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_void_p, C.c_int]

    def simd_level(self):
        return self.lib.ref_simd_level()

    def encode(self, planes, bit_depth, is_signed=False, reversible=True, num_decomps=5,
               block=(64, 64), color_transform=False, tile=(0, 0), prog_order="RPCL",
               planar=None, qstep=-1.0, precinct=(0, 0), tlm=False, precincts=None,
               downsampling=None, image_offset=(0, 0), tile_offset=(0, 0), size=None, tileparts="",
               profile=None, com=None, bit_depths=None, signs=None, qfactor=0, coc=None, nlt=None, qfactors=None):
        """planes: int32 array [num_comps, H, W], or a list of per-component 2-D arrays when the
        components are sub-sampled (then size=(W, H) is the image size on the reference grid).
        Returns codestream bytes."""
        if isinstance(planes, (list, tuple)):
            planes = [np.ascontiguousarray(q, dtype=np.int32) for q in planes]
            nc = len(planes)
            w, h = size if size is not None else (planes[0].shape[1], planes[0].shape[0])
        else:
            planes = np.ascontiguousarray(planes, dtype=np.int32)
            nc, h, w = planes.shape
            planes = [planes[c] for c in range(nc)]
        if planar is None:
            planar = not color_transform
        p = RefParams(w, h, nc, bit_depth, int(is_signed), int(reversible), num_decomps,
                      block[0], block[1], int(color_transform), tile[0], tile[1],
                      PROG_ORDERS[prog_order], int(planar), float(qstep),
                      precinct[0], precinct[1], int(tlm))
        if precincts:                      # list of (w, h) from the lowest resolution up, last one repeated
            for i in range(num_decomps + 1):
                pw, ph = precincts[min(i, len(precincts) - 1)]
                p.precinct_exps[i] = (pw.bit_length() - 1) | ((ph.bit_length() - 1) << 4)
        p.image_x0, p.image_y0 = image_offset
        p.tile_x0, p.tile_y0 = tile_offset
        p.tilepart_div = (1 if "R" in tileparts else 0) | (2 if "C" in tileparts else 0)
        for c, bd in enumerate(bit_depths or []):
            p.comp_depth[c] = bd
        for c, sg in enumerate(signs or []):
            p.comp_sign[c] = 2 if sg else 1
        p.qfactor = int(qfactor)
        for k, (c, (ctype, qf)) in enumerate((qfactors or {}).items()):   # set_qfactor(comp, ctype, qfactor) calls, in order
            p.cqf[k].comp, p.cqf[k].ctype, p.cqf[k].qfactor = int(c), ("Y", "Cb", "Cr").index(ctype), int(qf)
            p.num_cqf = k + 1
        for k, (c, t) in enumerate((nlt or {}).items()):       # set_nonlinear_transform calls, in order
            p.nlt[k].comp, p.nlt[k].type = (65535 if c == "all" else int(c)), int(t)
            p.num_nlt = k + 1
        for k, (c, st) in enumerate((coc or {}).items()):      # only the setters the dict names are called
            q = p.coc[k]
            q.comp = int(c)
            if "num_decomps" in st:
                q.mask |= 1; q.num_decomps = int(st["num_decomps"])
            if "block" in st:
                q.mask |= 2; q.log_bw, q.log_bh = st["block"][0].bit_length() - 1, st["block"][1].bit_length() - 1
            if st.get("precincts"):
                q.mask |= 4
                q.pad[0] = len(st["precincts"])
                for i, (pw, ph) in enumerate(st["precincts"]):
                    q.precinct_exps[i] = (pw.bit_length() - 1) | ((ph.bit_length() - 1) << 4)
            if "reversible" in st:
                q.mask |= 8; q.reversible = int(bool(st["reversible"]))
            p.num_coc = k + 1
        for c, (dx, dy) in enumerate(downsampling or []):
            p.comp_dx[c], p.comp_dy[c] = dx, dy
        ptrs = (C.c_void_p * nc)(*[planes[c].ctypes.data for c in range(nc)])
        cap = sum(q.size for q in planes) * 5 + (1 << 20)
        out = np.empty(cap, dtype=np.uint8)
        n = self.lib.ref_encode_ex(C.byref(p), ptrs, out.ctypes.data, cap,
                                   profile.encode() if profile else None, com.encode("latin-1") if com is not None else None)
        if n <= 0:
            raise RuntimeError("reference encode failed (%d)" % n)
        return out[:n].tobytes()

    def decode(self, data: bytes, resilient=False, skip=(0, 0), max_samples=None):
        """skip = (skipped_res_for_data, skipped_res_for_recon) of codestream::restrict_input_resolution;
        max_samples: raise TooLarge instead of decoding a frame of more samples (fuzzers of damaged SIZ segments)"""
        buf = np.frombuffer(data, dtype=np.uint8)
        info = np.zeros(8 + 32, dtype=np.uint32)
        r = self.lib.ref_decode_skip(buf.ctypes.data, len(data), None, info.ctypes.data, int(resilient), skip[0], skip[1])
        if r != 0:
            raise RuntimeError("reference read_headers failed (%d)" % r)
        w, h, nc = int(info[0]), int(info[1]), int(info[2])
        dims = [(int(info[9 + 2 * c]), int(info[8 + 2 * c])) if c < 16 else (h, w) for c in range(nc)]
        uniform = all(d == dims[0] for d in dims)
        if max_samples is not None and sum(a * b for a, b in dims) > max_samples:
            raise TooLarge("%d components of %s" % (nc, dims[:4]))
        if uniform:
            planes = np.zeros((nc, h, w), dtype=np.int32)
            views = [planes[c] for c in range(nc)]
        else:                              # sub-sampled components: a list of 2-D arrays
            planes = [np.zeros(d, dtype=np.int32) for d in dims]
            views = planes
        ptrs = (C.c_void_p * nc)(*[v.ctypes.data for v in views])
        r = self.lib.ref_decode_skip(buf.ctypes.data, len(data), ptrs, info.ctypes.data, int(resilient), skip[0], skip[1])
        if r != 0:
            raise RuntimeError("reference decode failed (%d)" % r)
        return planes, dict(bit_depth=int(info[3]), is_signed=bool(info[4]),
                            reversible=bool(info[5]))

    def encode_block(self, buf, missing_msbs, width, height, stride, variant=0):
        buf = np.ascontiguousarray(buf, dtype=np.uint32)
        out = np.empty(65536 * 4, dtype=np.uint8)
        n = self.lib.ref_encode_block32(variant, buf.ctypes.data, missing_msbs, width, height,
                                        stride, out.ctypes.data, out.size)
        if n <= 0:
            raise RuntimeError("reference block encode failed (%d)" % n)
        return out[:n].tobytes()

    def decode_block(self, coded: bytes, missing_msbs, width, height, stride, len2=0,
                     num_passes=1, variant=0, stripe_causal=False):
        data = np.frombuffer(coded, dtype=np.uint8)
        out = np.zeros((height + 2) * stride, dtype=np.uint32)
        len1 = len(coded) - len2
        r = self.lib.ref_decode_block32(variant, data.ctypes.data, len1, len2, missing_msbs,
                                        num_passes, width, height, stride, out.ctypes.data,
                                        int(stripe_causal))
        return r == 0, out[:height * stride].reshape(height, stride)

    def encode_block64(self, buf, missing_msbs, width, height, stride):
        """ojph_encode_codeblock64 on uint64 sign-magnitude samples"""
        buf = np.ascontiguousarray(buf, dtype=np.uint64)
        out = np.empty(65536 * 4, dtype=np.uint8)
        n = self.lib.ref_encode_block64(buf.ctypes.data, missing_msbs, width, height, stride, out.ctypes.data, out.size)
        if n <= 0:
            raise RuntimeError("reference 64-bit block encode failed (%d)" % n)
        return out[:n].tobytes()

    def decode_block64(self, coded: bytes, missing_msbs, width, height, stride, len2=0, num_passes=1, stripe_causal=False):
        data = np.frombuffer(coded, dtype=np.uint8)
        out = np.zeros((height + 2) * stride, dtype=np.uint64)
        len1 = len(coded) - len2
        r = self.lib.ref_decode_block64(data.ctypes.data, len1, len2, missing_msbs, num_passes, width, height, stride,
                                        out.ctypes.data, int(stripe_causal))
        return r == 0, out[:height * stride].reshape(height, stride)
