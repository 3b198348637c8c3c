"""Frame pipelines (C ABI section 6): sequences of same-shaped frames with the PCIe copies, the kernels
and the host Tier-2 of consecutive frames overlapped.  Mirrors how one ojph::codestream object is re-used
through restart() for the frames of a sequence (ojph_codestream.h:204): the caller writes samples into
memory the library hands out and receives finished codestreams -- or the other way round -- `depth`
frames in flight.  Plumbing only: ctypes views of the pipe's pinned host memory, no arithmetic here.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check
from .plan import Plan, make_params


_CONTAINER = {8: np.uint8, 16: np.uint16, 32: np.int32}      # numpy view of a frame slot by container width


def _view(ptr, nbytes, dtype=np.uint8):
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


def pack_bits(samples, bits):
    """numpy: unsigned samples (any shape, flattened in C order) -> the little-endian bit string ojphgpu_unpack_bits
    reads (sample i = bits [i * bits, (i + 1) * bits)), padded to a whole group of 32 samples"""
    v = np.ascontiguousarray(samples).reshape(-1)
    n = v.size
    pad = (-n) % 32
    if bits == 12:                                    # two samples in three bytes, without the bit matrix
        a = np.zeros(n + pad, np.uint16); a[:n] = v
        lo, hi = a[0::2], a[1::2]
        out = np.empty((lo.size, 3), np.uint8)
        out[:, 0] = lo & 0xFF; out[:, 1] = (lo >> 8) | ((hi & 0xF) << 4); out[:, 2] = hi >> 4
        return out.reshape(-1)
    v = v.astype(np.uint64)
    if pad:
        v = np.concatenate([v, np.zeros(pad, np.uint64)])
    b = ((v[:, None] >> np.arange(bits, dtype=np.uint64)[None, :]) & 1).astype(np.uint8)       # LSB first
    return np.packbits(b.reshape(-1), bitorder="little")


def unpack_bits(packed, bits, n):
    """numpy inverse of pack_bits: -> n samples (uint32)"""
    b = np.unpackbits(np.ascontiguousarray(packed).view(np.uint8), bitorder="little")[: n * bits].reshape(n, bits).astype(np.uint32)
    return (b << np.arange(bits, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)


class EncoderPipe:
    def __init__(self, plan: Plan = None, params=None, device=0, depth=4, container=16, host_threads=0, pixels=None, packed=None, **kw):
        """pixels=(bits, big_endian): the frames are handed over pixel-interleaved ([H,W,C] of 8- or 16-bit samples, the
        order of .ppm files / capture buffers; 16-bit samples byte-swapped when big_endian) and turned into planes on
        the device"""
        from .codec import _torch
        _torch()
        self.plan = plan if plan is not None else Plan(params if params is not None else make_params(**kw))
        self.container = int(container)
        self.depth = int(depth)
        self._lib = capi.lib()
        self._h = C.c_void_p()
        check(self._lib.ojphgpu_enc_pipe_create(self.plan.handle, device, self.depth, self.container, host_threads,
                                                C.byref(self._h)), "enc_pipe_create")
        self.pixels = None
        if pixels is not None:
            check(self._lib.ojphgpu_enc_pipe_set_pixels(self._h, int(pixels[0]), int(bool(pixels[1]))), "enc_pipe_set_pixels")
            self.pixels = (int(pixels[0]), bool(pixels[1]))
        self.packed = None
        if packed:                                        # planes of bit-packed samples (10 / 12 / 14 bits): acquire() -> uint8 view
            check(self._lib.ojphgpu_enc_pipe_set_packed(self._h, int(packed)), "enc_pipe_set_packed")
            self.packed = int(packed)
        self.in_flight = 0

    def close(self):
        if self._h:
            self._lib.ojphgpu_enc_pipe_destroy(self._h)
            self._h = None

    __del__ = close

    def acquire(self):
        """-> writable numpy view ([C,H,W] uint16 / int32; flat when components differ in size) of the pinned
        memory the next frame goes into, or None when every slot is in flight (collect first)"""
        ptr, n = C.c_void_p(), C.c_size_t()
        rc = self._lib.ojphgpu_enc_pipe_acquire(self._h, C.byref(ptr), C.byref(n))
        if rc == capi.E_AGAIN:
            return None
        check(rc, "enc_pipe_acquire")
        if self.packed:
            return _view(ptr.value, n.value, np.uint8)
        if self.pixels is not None:                     # [H,W,C] in the file's sample type (big endian: the raw bytes)
            c, h, w = self.plan.frame_shape
            dt = np.uint8 if self.pixels[0] == 8 else np.dtype(">u2" if self.pixels[1] else "<u2")
            return _view(ptr.value, n.value, np.uint8).view(dt).reshape(h, w, c)
        return _view(ptr.value, n.value, _CONTAINER[self.container]).reshape(self.plan.frame_shape)

    def submit(self):
        check(self._lib.ojphgpu_enc_pipe_submit(self._h), "enc_pipe_submit")
        self.in_flight += 1

    def collect(self, copy=True):
        """the oldest frame's codestream; copy=False returns a view of pinned memory valid until the next collect"""
        ptr, n = C.c_void_p(), C.c_size_t()
        if self.in_flight <= 0:              # nothing to collect: the counter stays where it is (the C call would say E_INVALID too)
            raise capi.OjphError(capi.E_INVALID, "enc_pipe_collect: nothing in flight")
        rc = self._lib.ojphgpu_enc_pipe_collect(self._h, C.byref(ptr), C.byref(n))
        self.in_flight -= 1                  # the oldest frame is taken whatever its outcome (a frame of its own may fail with E_INVALID)
        check(rc, "enc_pipe_collect")
        v = _view(ptr.value, n.value)
        return v.tobytes() if copy else v

    def stats(self):
        out = (C.c_double * 4)()
        check(self._lib.ojphgpu_enc_pipe_stats(self._h, out), "enc_pipe_stats")
        return dict(frames=int(out[0]), host_tier2_ms=out[1], latency_ms=out[2], tier2_threads=int(out[3]))

    def encode_sequence(self, frames):
        """frames: iterable of [C,H,W] arrays -> generator of codestreams, in order"""
        for f in frames:
            buf = self.acquire()
            while buf is None:
                yield self.collect()
                buf = self.acquire()
            np.copyto(buf, np.asarray(f).astype(buf.dtype, copy=False).reshape(buf.shape), casting="unsafe")
            self.submit()
        while self.in_flight:
            yield self.collect()


class DecoderPipe:
    def __init__(self, first_codestream: bytes, device=0, depth=4, container=16, host_threads=0, resilient=False, pixels=None, packed=None):
        """pixels=(bits, big_endian): decoded frames come back pixel-interleaved ([H,W,C]), clamped to the bit depth"""
        from .codec import _torch
        _torch()
        self.container = int(container)
        self._lib = capi.lib()
        self._h = C.c_void_p()
        buf = np.frombuffer(first_codestream, dtype=np.uint8)
        check(self._lib.ojphgpu_dec_pipe_create(buf.ctypes.data, len(first_codestream), int(resilient), device, depth,
                                                self.container, host_threads, C.byref(self._h)), "dec_pipe_create")
        h = C.c_void_p()
        check(self._lib.ojphgpu_dec_pipe_plan(self._h, C.byref(h)), "dec_pipe_plan")
        self.plan = Plan(handle=h, owned=False)
        self.pixels = None
        if pixels is not None:
            check(self._lib.ojphgpu_dec_pipe_set_pixels(self._h, int(pixels[0]), int(bool(pixels[1]))), "dec_pipe_set_pixels")
            self.pixels = (int(pixels[0]), bool(pixels[1]))
        self.packed = None
        if packed:
            check(self._lib.ojphgpu_dec_pipe_set_packed(self._h, int(packed)), "dec_pipe_set_packed")
            self.packed = int(packed)
        self.in_flight = 0

    def close(self):
        if self._h:
            self._lib.ojphgpu_dec_pipe_destroy(self._h)
            self._h = None

    __del__ = close

    def acquire(self, nbytes):
        """-> writable uint8 view of pinned memory for the next codestream, or None when every slot is in flight"""
        ptr = C.c_void_p()
        rc = self._lib.ojphgpu_dec_pipe_acquire(self._h, nbytes, C.byref(ptr))
        if rc == capi.E_AGAIN:
            return None
        check(rc, "dec_pipe_acquire")
        return _view(ptr.value, nbytes)

    def submit(self):
        check(self._lib.ojphgpu_dec_pipe_submit(self._h), "dec_pipe_submit")
        self.in_flight += 1

    def collect(self, copy=True):
        """the oldest frame.  When code-blocks failed on a non-resilient pipe the C ABI still hands the frame out
        (failed blocks zeroed) with OJPHGPU_E_BLOCK: the exception raised here carries it as .frame and the count as
        .failed_blocks (ojphgpu.h, ojphgpu_dec_pipe_collect)"""
        ptr, n, failed = C.c_void_p(), C.c_size_t(), C.c_uint32()
        if self.in_flight <= 0:              # nothing to collect: the counter stays where it is
            raise capi.OjphError(capi.E_INVALID, "dec_pipe_collect: nothing in flight")
        rc = self._lib.ojphgpu_dec_pipe_collect(self._h, C.byref(ptr), C.byref(n), C.byref(failed))
        self.in_flight -= 1                  # the oldest frame is taken whatever its outcome (a codestream of another geometry fails with E_INVALID)
        if rc != capi.E_BLOCK:
            check(rc, "dec_pipe_collect")
        if self.packed:
            v = _view(ptr.value, n.value, np.uint8)
        elif self.pixels is not None:
            c, h, w = self.plan.frame_shape
            dt = np.uint8 if self.pixels[0] == 8 else np.dtype(">u2" if self.pixels[1] else "<u2")
            v = _view(ptr.value, n.value, np.uint8).view(dt).reshape(h, w, c)
        else:
            v = _view(ptr.value, n.value, _CONTAINER[self.container]).reshape(self.plan.frame_shape)
        if rc == capi.E_BLOCK:
            err = capi.OjphError(rc, "dec_pipe_collect: %d code-blocks" % failed.value)
            err.frame, err.failed_blocks = v.copy(), int(failed.value)
            raise err
        return v.copy() if copy else v

    def stats(self):
        out = (C.c_double * 4)()
        check(self._lib.ojphgpu_dec_pipe_stats(self._h, out), "dec_pipe_stats")
        n = C.c_uint32()
        check(self._lib.ojphgpu_dec_pipe_fused_retries(self._h, C.byref(n)), "dec_pipe_fused_retries")
        return dict(frames=int(out[0]), host_parse_ms=out[1], latency_ms=out[2], host_threads=int(out[3]), fused_retries=int(n.value))

    def decode_sequence(self, codestreams):
        for cs in codestreams:
            buf = self.acquire(len(cs))
            while buf is None:
                yield self.collect()
                buf = self.acquire(len(cs))
            buf[:] = np.frombuffer(cs, dtype=np.uint8)
            self.submit()
        while self.in_flight:
            yield self.collect()
