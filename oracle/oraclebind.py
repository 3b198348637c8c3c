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
        L.ojo_ht_encode64.restype = C.c_int
        L.ojo_ht_encode64.argtypes = L.ojo_ht_encode.argtypes
        L.ojo_ht_decode64.restype = C.c_int
        L.ojo_ht_decode64.argtypes = L.ojo_ht_decode.argtypes
        for name in ("ojo_dwt_fwd_gen", "ojo_dwt_inv_gen"):
            f = getattr(L, name); f.restype = None
            f.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_int, C.c_float] + [C.c_void_p, C.c_int] * 4
        L.ojo_quant_rev64.restype = C.c_uint64
        L.ojo_quant_rev64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ojo_dequant_rev64.restype = None
        L.ojo_dequant_rev64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        for name in ("ojo_rev_convert_to64", "ojo_rev_convert_from64"):
            f = getattr(L, name); f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]
        for name in ("ojo_rct_fwd64", "ojo_rct_inv64"):
            f = getattr(L, name); f.restype = None
            f.argtypes = [C.c_void_p] * 6 + [C.c_int]
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


def ht_encode64(buf, width, height, stride, missing_msbs, variant=0):
    """uint64 sign-magnitude samples (sign in bit 63): the 64-bit sample path"""
    buf = np.ascontiguousarray(buf, dtype=np.uint64)
    out = np.empty(40960, dtype=np.uint8)
    n = lib().ojo_ht_encode64(buf.ctypes.data, width, height, stride, missing_msbs, out.ctypes.data, out.size, variant)
    return out[:n].tobytes()


def ht_decode64(coded: bytes, width, height, stride, missing_msbs, len2=0, num_passes=1, stripe_causal=False):
    data = np.frombuffer(coded, dtype=np.uint8)
    out = np.zeros((height, stride), dtype=np.uint64)
    ok = lib().ojo_ht_decode64(data.ctypes.data, len(coded) - len2, len2, num_passes, missing_msbs,
                               width, height, stride, out.ctypes.data, int(stripe_causal))
    return bool(ok), out


class LiftStep(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("e", C.c_int32), ("A", C.c_float)]


ELEM = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float32): 2}
REV53 = [(1, 2, 2), (-1, 1, 1)]                                   # param_atk::init_rev53 (ojph_params.cpp:2883-2896), synthesis order
IRV97 = [0.443506852043971, 0.882911075530934, -0.052980118572961, -1.586134342059924]
K97 = 1.230174104914001


def _steps(steps):
    arr = (LiftStep * max(1, len(steps)))()
    for i, st in enumerate(steps):
        if isinstance(st, (tuple, list)):
            arr[i].a, arr[i].b, arr[i].e = st
        else:
            arr[i].A = float(st)
    return arr


def dwt_fwd_gen(src, steps, K=1.0, horz=True, vert=True, x_even=True, y_even=True):
    """one level in the general form; src int32 / int64 / float32 2-D; -> (ll, hl, lh, hh), absent bands empty"""
    src = np.ascontiguousarray(src)
    dt = src.dtype
    h, w = src.shape
    lw, hw, lh_, hh_ = band_dims(w, h, x_even, y_even)
    if not horz:
        lw, hw = w, 0
    if not vert:
        lh_, hh_ = h, 0
    mk = lambda r, c: np.zeros((max(r, 1), max(c, 1)), dt)
    ll, hl, lh, hh = mk(lh_, lw), mk(lh_, hw), mk(hh_, lw), mk(hh_, hw)
    lib().ojo_dwt_fwd_gen(src.ctypes.data, w, w, h, int(x_even), int(y_even), ELEM[dt], int(horz), int(vert),
                          _steps(steps), len(steps), float(K), ll.ctypes.data, ll.shape[1], hl.ctypes.data, hl.shape[1],
                          lh.ctypes.data, lh.shape[1], hh.ctypes.data, hh.shape[1])
    return ll[:lh_, :lw], hl[:lh_, :hw], lh[:hh_, :lw], hh[:hh_, :hw]


def dwt_inv_gen(ll, hl, lh, hh, w, h, steps, K=1.0, horz=True, vert=True, x_even=True, y_even=True):
    dt = ll.dtype
    bands = [np.ascontiguousarray(b, dtype=dt) if b.size else np.zeros((1, 1), dt) for b in (ll, hl, lh, hh)]
    dst = np.zeros((h, w), dt)
    args = []
    for b in bands:
        args += [b.ctypes.data, b.shape[1]]
    lib().ojo_dwt_inv_gen(dst.ctypes.data, w, w, h, int(x_even), int(y_even), ELEM[dt], int(horz), int(vert),
                          _steps(steps), len(steps), float(K), *args)
    return dst


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
