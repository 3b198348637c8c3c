"""ctypes binding to oracle/libht_oracle.so (our plain-C restatement of the hot path).

TEST INFRASTRUCTURE ONLY -- see oracle/ht_oracle.h.  Importable from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(); never from the product package.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(_HERE, "libht_oracle.so"))
        L.ojo_ht_encode.restype = C.c_int
        L.ojo_ht_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int]
        L.ojo_ht_decode.restype = C.c_int
        L.ojo_ht_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_int]
        for name in ("ojo_dwt53_fwd", "ojo_dwt97_fwd"):
            f = getattr(L, name); f.restype = None
            f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p, C.c_int] * 4
        for name in ("ojo_dwt53_inv", "ojo_dwt97_inv"):
            f = getattr(L, name); f.restype = None
            f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p, C.c_int] * 4
        L.ojo_quant_rev.restype = C.c_uint32
        L.ojo_quant_rev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ojo_quant_irv.restype = C.c_uint32
        L.ojo_quant_irv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.ojo_dequant_rev.restype = None
        L.ojo_dequant_rev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ojo_dequant_irv.restype = None
        L.ojo_dequant_irv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.ojo_rev_convert.restype = None
        L.ojo_rev_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int32]
        L.ojo_irv_to_float.restype = None
        L.ojo_irv_to_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ojo_irv_to_int.restype = None
        L.ojo_irv_to_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        for name in ("ojo_rct_fwd", "ojo_rct_inv", "ojo_ict_fwd", "ojo_ict_inv"):
            f = getattr(L, name); f.restype = None
            f.argtypes = [C.c_void_p] * 6 + [C.c_int]
        _lib = L
    return _lib


def ht_encode(buf, width, height, stride, missing_msbs, variant=0):
    buf = np.ascontiguousarray(buf, dtype=np.uint32)
    out = np.empty(24576, dtype=np.uint8)
    n = lib().ojo_ht_encode(buf.ctypes.data, width, height, stride, missing_msbs,
                            out.ctypes.data, out.size, variant)
    return out[:n].tobytes()


def ht_decode(coded: bytes, width, height, stride, missing_msbs, len2=0, num_passes=1,
              stripe_causal=False):
    data = np.frombuffer(coded, dtype=np.uint8)
    out = np.zeros((height, stride), dtype=np.uint32)
    ok = lib().ojo_ht_decode(data.ctypes.data, len(coded) - len2, len2, num_passes, missing_msbs,
                             width, height, stride, out.ctypes.data, int(stripe_causal))
    return bool(ok), out


def band_dims(w, h, x_even=True, y_even=True):
    lw = (w + (1 if x_even else 0)) >> 1
    hw = (w + (0 if x_even else 1)) >> 1
    lh = (h + (1 if y_even else 0)) >> 1
    hh = (h + (0 if y_even else 1)) >> 1
    return lw, hw, lh, hh


def _dwt_fwd(name, src, dtype, x_even, y_even):
    src = np.ascontiguousarray(src, dtype=dtype)
    h, w = src.shape
    lw, hw, lh, hh = band_dims(w, h, x_even, y_even)
    ll = np.zeros((lh, max(lw, 1)), dtype); hl = np.zeros((lh, max(hw, 1)), dtype)
    lhb = np.zeros((hh, max(lw, 1)), dtype); hhb = np.zeros((hh, max(hw, 1)), dtype)
    getattr(lib(), name)(src.ctypes.data, w, w, h, int(x_even), int(y_even),
                         ll.ctypes.data, ll.shape[1], hl.ctypes.data, hl.shape[1],
                         lhb.ctypes.data, lhb.shape[1], hhb.ctypes.data, hhb.shape[1])
    return ll[:, :lw], hl[:, :hw], lhb[:, :lw], hhb[:, :hw]


def _dwt_inv(name, ll, hl, lh, hh, w, h, dtype, x_even, y_even):
    bands = [np.ascontiguousarray(b, dtype=dtype) if b.size else np.zeros((1, 1), dtype)
             for b in (ll, hl, lh, hh)]
    dst = np.zeros((h, w), dtype)
    args = []
    for b in bands:
        args += [b.ctypes.data, b.shape[1]]
    getattr(lib(), name)(dst.ctypes.data, w, w, h, int(x_even), int(y_even), *args)
    return dst


def dwt53_fwd(src, x_even=True, y_even=True):
    return _dwt_fwd("ojo_dwt53_fwd", src, np.int32, x_even, y_even)


def dwt53_inv(ll, hl, lh, hh, w, h, x_even=True, y_even=True):
    return _dwt_inv("ojo_dwt53_inv", ll, hl, lh, hh, w, h, np.int32, x_even, y_even)


def dwt97_fwd(src, x_even=True, y_even=True):
    return _dwt_fwd("ojo_dwt97_fwd", src, np.float32, x_even, y_even)


def dwt97_inv(ll, hl, lh, hh, w, h, x_even=True, y_even=True):
    return _dwt_inv("ojo_dwt97_inv", ll, hl, lh, hh, w, h, np.float32, x_even, y_even)


def quant_rev(src, K_max):
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.empty(src.shape, np.uint32)
    mx = lib().ojo_quant_rev(src.ctypes.data, dst.ctypes.data, src.size, K_max)
    return dst, mx


def quant_irv(src, delta_inv):
    src = np.ascontiguousarray(src, dtype=np.float32)
    dst = np.empty(src.shape, np.uint32)
    mx = lib().ojo_quant_irv(src.ctypes.data, dst.ctypes.data, src.size, delta_inv)
    return dst, mx


def dequant_rev(src, K_max):
    src = np.ascontiguousarray(src, dtype=np.uint32)
    dst = np.empty(src.shape, np.int32)
    lib().ojo_dequant_rev(src.ctypes.data, dst.ctypes.data, src.size, K_max)
    return dst


def dequant_irv(src, delta):
    src = np.ascontiguousarray(src, dtype=np.uint32)
    dst = np.empty(src.shape, np.float32)
    lib().ojo_dequant_irv(src.ctypes.data, dst.ctypes.data, src.size, delta)
    return dst
