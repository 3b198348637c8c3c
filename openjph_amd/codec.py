"""Python host-side wrapper of the GPU codec objects (C ABI sections 4 and 5).

PyTorch is used only as plumbing: device memory (torch tensors on ``cuda:N``) and streams.  All
arithmetic happens in the HIP kernels behind libojphgpu.so; there is no CPU fallback -- without the
library or without a GPU these calls raise.
"""
import ctypes as C
import numpy as np

from . import capi
from .capi import check, Params, DwtDesc, CbDesc, CbResult, ConvertDesc
from .plan import Plan, make_params, parse_codestream

dwt_desc_dtype = np.dtype(DwtDesc)
cb_desc_dtype = np.dtype(CbDesc)
cb_result_dtype = np.dtype(CbResult)
convert_desc_dtype = np.dtype(ConvertDesc)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("openjph_amd: no GPU visible (torch.cuda.is_available() is False); "
                           "the HTJ2K hot path has no CPU fallback")
    return torch


def _stream_ptr(torch, device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def to_device(arr: np.ndarray, device=0):
    """numpy (any dtype / structured) -> uint8 torch tensor on the device holding the same bytes."""
    torch = _torch()
    raw = np.frombuffer(np.ascontiguousarray(arr).tobytes(), dtype=np.uint8)
    if raw.size == 0:
        return torch.zeros(16, dtype=torch.uint8, device="cuda:%d" % device)
    return torch.from_numpy(raw.copy()).to("cuda:%d" % device)


class Encoder:
    """Whole-frame encoder for one frame shape / parameter set (ojphgpu_encoder)."""

    def __init__(self, params: Params = None, device=0, plan: Plan = None, tiles=None, frames=1, **kw):
        """tiles=(first, count) restricts the encoder to a run of tiles (multi-GPU sharding);
        frames=B makes it code a batch of B independent frames per run ([B,C,H,W] input)."""
        torch = _torch()
        self.device = device
        self.plan = plan if plan is not None else Plan(params if params is not None else make_params(**kw))
        self.tiles = (0, self.plan.num_tiles) if tiles is None else (int(tiles[0]), int(tiles[1]))
        self.frames = int(frames)
        self._lib = capi.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(device):
            if self.frames > 1:
                assert tiles is None, "a batch encoder codes whole frames"
                check(self._lib.ojphgpu_encoder_create_batch(self.plan.handle, device, _stream_ptr(torch, device),
                                                             self.frames, C.byref(self._h)), "encoder_create_batch")
            else:
                check(self._lib.ojphgpu_encoder_create_tiles(self.plan.handle, device, _stream_ptr(torch, device),
                                                             self.tiles[0], self.tiles[1], C.byref(self._h)),
                      "encoder_create")
        fs = self.plan.frame_shape          # [C,H,W], or flat (frame_elems,) when components differ in size
        self.shape = fs if self.frames == 1 else (self.frames,) + fs

    def __del__(self):
        try:
            if self._h:
                self._lib.ojphgpu_encoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def run_device(self, d_image):
        """d_image: int32 torch tensor [C,H,W] (flat plan.frame_shape for sub-sampled components)
        resident on the device. Asynchronous."""
        torch = _torch()
        assert d_image.is_cuda and tuple(d_image.shape) == self.shape and d_image.is_contiguous()
        if d_image.dtype in (torch.int16, torch.uint16):     # 16-bit containers: int16 for signed components, else uint16
            check(self._lib.ojphgpu_encoder_run_device16(self._h, C.c_void_p(d_image.data_ptr())), "encoder_run_device16")
            return
        if d_image.dtype in (torch.int8, torch.uint8):       # 8-bit containers
            check(self._lib.ojphgpu_encoder_run_device8(self._h, C.c_void_p(d_image.data_ptr())), "encoder_run_device8")
            return
        assert d_image.dtype == torch.int32
        check(self._lib.ojphgpu_encoder_run_device(self._h, C.c_void_p(d_image.data_ptr())), "encoder_run_device")

    def set_timing(self, per_launch: bool):
        check(self._lib.ojphgpu_encoder_set_timing(self._h, int(per_launch)), "encoder_set_timing")

    def coded_bytes(self):
        n = C.c_uint64()
        check(self._lib.ojphgpu_encoder_coded_bytes(self._h, C.byref(n)), "encoder_coded_bytes")
        return int(n.value)

    def finish(self, frame=0) -> bytes:
        """codestream of frame `frame` of the last run"""
        cap = self.coded_bytes() // self.frames * 2 + 64 * self.plan.num_blocks + (1 << 20)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        rc = self._lib.ojphgpu_encoder_finish_frame(self._h, frame, out.ctypes.data, cap, C.byref(n))
        if rc == capi.E_OVERFLOW and n.value > cap:
            cap = int(n.value)
            out = np.empty(cap, np.uint8)
            rc = self._lib.ojphgpu_encoder_finish_frame(self._h, frame, out.ctypes.data, cap, C.byref(n))
        check(rc, "encoder_finish")
        return out[:n.value].tobytes()

    def finish_tiles(self):
        """-> (tile-part bytes of this encoder's tile range, Psot per tile)"""
        cap = self.plan.frame_elems * 3 + (1 << 20)
        lens = np.zeros(max(self.tiles[1], 1) * self.plan.parts_per_tile, np.uint32)
        n = C.c_size_t()
        out = np.empty(cap, np.uint8)
        rc = self._lib.ojphgpu_encoder_finish_tiles(self._h, out.ctypes.data, cap, C.byref(n), lens.ctypes.data)
        if rc == capi.E_OVERFLOW and n.value > cap:
            cap = int(n.value)
            out = np.empty(cap, np.uint8)
            rc = self._lib.ojphgpu_encoder_finish_tiles(self._h, out.ctypes.data, cap, C.byref(n), lens.ctypes.data)
        check(rc, "encoder_finish_tiles")
        return out[:n.value].tobytes(), lens[:self.tiles[1] * self.plan.parts_per_tile].copy()

    def finish_tiles_device(self):
        """-> (tile-part bytes of this encoder's tile range as a uint8 torch tensor ON THE DEVICE, Psot per tile-part):
        the bytes are assembled in HBM and never visit the host (input of shard.gather_bytes over RCCL)"""
        torch = _torch()
        cap = self.coded_bytes() + 64 * self.plan.num_blocks + 4096 * max(self.tiles[1], 1) + (1 << 16)
        lens = np.zeros(max(self.tiles[1], 1) * self.plan.parts_per_tile, np.uint32)
        n = C.c_size_t()
        out = torch.empty(cap, dtype=torch.uint8, device="cuda:%d" % self.device)
        rc = self._lib.ojphgpu_encoder_finish_tiles_device(self._h, C.c_void_p(out.data_ptr()), cap, C.byref(n), lens.ctypes.data)
        if rc == capi.E_OVERFLOW and n.value > cap:
            cap = int(n.value)
            out = torch.empty(cap, dtype=torch.uint8, device="cuda:%d" % self.device)
            rc = self._lib.ojphgpu_encoder_finish_tiles_device(self._h, C.c_void_p(out.data_ptr()), cap, C.byref(n), lens.ctypes.data)
        check(rc, "encoder_finish_tiles_device")
        return out[:n.value], lens[:self.tiles[1] * self.plan.parts_per_tile].copy()

    def encode(self, image):
        """image: numpy int32 [C,H,W] (host), a list of per-component 2-D arrays (sub-sampled
        components), or a torch int32 tensor on the device -> codestream bytes; for a batch encoder
        [B,C,H,W] -> list of B codestreams."""
        torch = _torch()
        if isinstance(image, (list, tuple)):
            image = self.plan.pack_frame(image)
        if isinstance(image, np.ndarray):
            if image.dtype in (np.int16, np.uint16):         # stays 16 bits wide on its way to and in HBM
                image = torch.from_numpy(np.ascontiguousarray(image).view(np.int16)).to("cuda:%d" % self.device)
            elif image.dtype in (np.int8, np.uint8):         # ... or 8
                image = torch.from_numpy(np.ascontiguousarray(image).view(np.int8)).to("cuda:%d" % self.device)
            else:
                image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.int32)).to("cuda:%d" % self.device)
        self.run_device(image)
        if self.frames > 1:
            return [self.finish(f) for f in range(self.frames)]
        return self.finish()

    def timing(self):
        t = (C.c_float * 4)()
        check(self._lib.ojphgpu_encoder_timing(self._h, t), "encoder_timing")
        lv = (C.c_float * 40)(); n = C.c_uint32()
        check(self._lib.ojphgpu_encoder_level_timing(self._h, lv, 40, C.byref(n)), "encoder_level_timing")
        ht = (C.c_float * 4)(); nh = C.c_uint32(); ntop = C.c_uint32()
        check(self._lib.ojphgpu_encoder_ht_timing(self._h, ht, 4, C.byref(nh), C.byref(ntop)), "encoder_ht_timing")
        return dict(convert_ms=t[0], dwt_ms=t[1], ht_ms=t[2], total_ms=t[3], dwt_levels_ms=[lv[i] for i in range(n.value)],
                    ht_launches_ms=[ht[i] for i in range(nh.value)])

    def top_blocks(self):
        """number of block descriptors coded on the side stream (0 = no overlap)"""
        ht = (C.c_float * 4)(); nh = C.c_uint32(); ntop = C.c_uint32()
        check(self._lib.ojphgpu_encoder_ht_timing(self._h, ht, 4, C.byref(nh), C.byref(ntop)), "encoder_ht_timing")
        return int(ntop.value)


class Decoder:
    """Whole-frame decoder bound to one parsed codestream layout (ojphgpu_decoder)."""

    def __init__(self, codestream, device=0, resilient=False, tiles=None, skip_res=None):
        """codestream: bytes, or a list of codestreams of same-shaped frames (batch decoder, output
        [B,C,H,W]).  tiles=(first, count) restricts the decoder to a run of tiles (multi-GPU sharding).
        skip_res=n or (for_data, for_recon): reduced-resolution decoding (codestream::restrict_input_resolution)."""
        torch = _torch()
        self.device = device
        self.resilient = resilient
        streams = list(codestream) if isinstance(codestream, (list, tuple)) else [codestream]
        self.frames = len(streams)
        self.plans = [parse_codestream(cs, resilient) for cs in streams]
        if skip_res:
            a, b = (skip_res, skip_res) if isinstance(skip_res, int) else skip_res
            for pl in self.plans:
                pl.restrict_resolution(a, b)
        self.plan = self.plans[0]
        self.tiles = (0, self.plan.num_tiles) if tiles is None else (int(tiles[0]), int(tiles[1]))
        self._lib = capi.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(device):
            if self.frames > 1:
                assert tiles is None, "a batch decoder decodes whole frames"
                arr = (C.c_void_p * self.frames)(*[pl.handle.value if hasattr(pl.handle, "value") else pl.handle for pl in self.plans])
                check(self._lib.ojphgpu_decoder_create_batch(arr, self.frames, device, _stream_ptr(torch, device),
                                                             C.byref(self._h)), "decoder_create_batch")
            else:
                check(self._lib.ojphgpu_decoder_create_tiles(self.plan.handle, device, _stream_ptr(torch, device),
                                                             self.tiles[0], self.tiles[1], C.byref(self._h)),
                      "decoder_create")
        fs = self.plan.frame_shape
        self.shape = fs if self.frames == 1 else (self.frames,) + fs
        for f, cs in enumerate(streams):
            self.upload(cs, f)

    def __del__(self):
        try:
            if self._h:
                self._lib.ojphgpu_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def upload(self, codestream: bytes, frame=0):
        buf = np.frombuffer(codestream, dtype=np.uint8)
        check(self._lib.ojphgpu_decoder_upload_frame(self._h, frame, buf.ctypes.data, len(codestream)), "decoder_upload")
        _torch().cuda.synchronize(self.device)   # the host buffer may go away after this call

    def run_device(self, d_image=None, dtype=None):
        """dtype / d_image.dtype torch.int16 / uint16 or torch.int8 / uint8: samples in 16- / 8-bit containers.
        Asynchronous: the launches are enqueued on the decoder's stream.  The run is COLLECTED by failed_blocks() (decode()
        does it): that is where a block's verdict is read, and where a frame is decoded once more through the separate
        launches if the one-launch block decoder ran out of patience on a chip held by others (fused_retries() counts; it
        has not happened on an idle or a shared chip so far) -- a caller that reads d_image on the device without
        collecting the run takes that frame as it is."""
        torch = _torch()
        if d_image is None:
            alloc = torch.empty if self.tiles == (0, self.plan.num_tiles) else torch.zeros
            d_image = alloc(self.shape, dtype=dtype or torch.int32, device="cuda:%d" % self.device)
        if d_image.dtype in (torch.int16, torch.uint16):
            check(self._lib.ojphgpu_decoder_run_device16(self._h, C.c_void_p(d_image.data_ptr())), "decoder_run_device16")
        elif d_image.dtype in (torch.int8, torch.uint8):
            check(self._lib.ojphgpu_decoder_run_device8(self._h, C.c_void_p(d_image.data_ptr())), "decoder_run_device8")
        else:
            check(self._lib.ojphgpu_decoder_run_device(self._h, C.c_void_p(d_image.data_ptr())), "decoder_run_device")
        return d_image

    def set_timing(self, per_launch: bool):
        check(self._lib.ojphgpu_decoder_set_timing(self._h, int(per_launch)), "decoder_set_timing")

    def failed_blocks(self):
        n = C.c_uint32()
        check(self._lib.ojphgpu_decoder_failed_blocks(self._h, C.byref(n)), "decoder_failed_blocks")
        return int(n.value)

    def fused_retries(self):
        """runs of this decoder that were repeated through the separate launches (see ojphgpu_decoder_failed_blocks)"""
        n = C.c_uint32()
        check(self._lib.ojphgpu_decoder_fused_retries(self._h, C.byref(n)), "decoder_fused_retries")
        return int(n.value)

    def giveup_epoch(self):
        """-> (last_giveup, current): the number of one-launch block-decoder runs enqueued so far, and the number of the newest
        one whose wait ran out (0: none); synchronises the stream.  Brackets a series of uncollected runs (bench.py)."""
        g, c = C.c_uint32(), C.c_uint32()
        check(self._lib.ojphgpu_decoder_giveup_epoch(self._h, C.byref(g), C.byref(c)), "decoder_giveup_epoch")
        return int(g.value), int(c.value)

    def decode(self) -> np.ndarray:
        img = self.run_device()
        failed = self.failed_blocks()
        if failed and not self.resilient:
            raise capi.OjphError(capi.E_BLOCK, "%d code-blocks" % failed)   # ojph_codeblock.cpp:214-224
        return img.cpu().numpy()

    def timing(self):
        t = (C.c_float * 4)()
        check(self._lib.ojphgpu_decoder_timing(self._h, t), "decoder_timing")
        lv = (C.c_float * 40)(); n = C.c_uint32()
        check(self._lib.ojphgpu_decoder_level_timing(self._h, lv, 40, C.byref(n)), "decoder_level_timing")
        ht = (C.c_float * 3)()
        check(self._lib.ojphgpu_decoder_ht_timing(self._h, ht), "decoder_ht_timing")
        return dict(ht_ms=t[0], dwt_ms=t[1], convert_ms=t[2], total_ms=t[3], dwt_levels_ms=[lv[i] for i in range(n.value)],
                    ht_prep_ms=ht[0], ht_step1_ms=ht[1], ht_step2_ms=ht[2])


class MultiEncoder:
    """One frame over several GPUs of this process (ojphgpu_multi_encoder: include/ojphgpu.h section 8): contiguous runs
    of tiles, one host thread + one encoder object per device, tile-parts copied from every GPU straight to their place
    in one pinned host buffer.  devices: list of device numbers (a number may repeat)."""

    def __init__(self, params=None, plan=None, devices=(0,)):
        torch = _torch()
        self.plan = plan if plan is not None else Plan(params)
        self._lib = capi.lib()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(self._lib.ojphgpu_multi_encoder_create(self.plan.handle, devs, len(devices), C.byref(self._h)), "multi_encoder_create")
        self._out = torch.empty(max(self.plan.frame_elems * 4, 1 << 20) + (1 << 20), dtype=torch.uint8).pin_memory()
        n, per = C.c_uint32(), (C.c_uint32 * 64)()
        check(self._lib.ojphgpu_multi_encoder_workers(self._h, C.byref(n), per, 64), "multi_encoder_workers")
        self.tiles_per_worker = [int(per[i]) for i in range(n.value)]

    def __del__(self):
        try:
            if self._h:
                self._lib.ojphgpu_multi_encoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def encode(self, image, copy=True):
        """image: numpy array / pinned torch tensor in the frame layout, int32 or in 16- / 8-bit containers (the dtype says
        which: half / a quarter of the bytes over every device's link) -> the codestream (bytes; copy=False: a view of the
        encoder's pinned buffer, valid until the next call)"""
        torch = _torch()
        if hasattr(image, "data_ptr"):
            t = image
        else:
            a = np.ascontiguousarray(image)
            t = torch.from_numpy(a if a.dtype.itemsize in (1, 2) and a.dtype.kind in "iu" else a.astype(np.int32))
        bits = 8 * t.element_size()
        n = C.c_size_t()
        check(self._lib.ojphgpu_multi_encode_container(self._h, C.c_void_p(t.data_ptr()), bits, C.c_void_p(self._out.data_ptr()),
                                                       self._out.numel(), C.byref(n)), "multi_encode")
        v = self._out[:n.value].numpy()
        return v.tobytes() if copy else v


class MultiDecoder:
    """The decoding mirror of MultiEncoder (ojphgpu_multi_decoder)."""

    def __init__(self, codestream: bytes, devices=(0,), resilient=False, skip_res=None):
        torch = _torch()
        self._lib = capi.lib()
        self._h = C.c_void_p()
        self._cs = np.frombuffer(codestream, np.uint8)
        a, b = (0, 0) if not skip_res else ((skip_res, skip_res) if isinstance(skip_res, int) else skip_res)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(self._lib.ojphgpu_multi_decoder_create(self._cs.ctypes.data, len(codestream), int(resilient), a, b, devs, len(devices),
                                                     C.byref(self._h)), "multi_decoder_create")
        ph = C.c_void_p()
        check(self._lib.ojphgpu_multi_decoder_plan(self._h, C.byref(ph)), "multi_decoder_plan")
        self.plan = Plan(handle=ph, owned=False)
        self.resilient = resilient
        self._img = torch.zeros(self.plan.frame_shape, dtype=torch.int32).pin_memory()

    def __del__(self):
        try:
            if self._h:
                self._lib.ojphgpu_multi_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def decode(self, copy=True, dtype=None):
        """dtype np.int16 / np.uint16 / np.int8 / np.uint8: the samples come down in 16- / 8-bit containers"""
        torch = _torch()
        dt = np.dtype(dtype or np.int32)
        if dt.itemsize != 4:
            key = "_img%d" % dt.itemsize
            if not hasattr(self, key):
                setattr(self, key, torch.zeros(self.plan.frame_shape, dtype=torch.int16 if dt.itemsize == 2 else torch.int8).pin_memory())
            img = getattr(self, key)
        else:
            img = self._img
        failed = C.c_uint32()
        check(self._lib.ojphgpu_multi_decode_container(self._h, self._cs.ctypes.data, len(self._cs), C.c_void_p(img.data_ptr()),
                                                       8 * dt.itemsize, C.byref(failed)), "multi_decode")
        v = img.numpy().view(dt) if dt.itemsize != 4 else img.numpy()
        return v.copy() if copy else v


def encode(image: np.ndarray, device=0, **kw) -> bytes:
    """One-shot helper: image int32 [C,H,W]; keyword args as in plan.make_params (minus sizes)."""
    nc, h, w = image.shape
    return Encoder(make_params(w, h, nc, **kw), device=device).encode(image)


def decode(codestream: bytes, device=0, resilient=False, skip_res=None) -> np.ndarray:
    return Decoder(codestream, device=device, resilient=resilient, skip_res=skip_res).decode()


# -------------------------------------------------------------------------------------------------
# stage-level entry points (used by the parity tests; same C ABI the codec objects call)
# -------------------------------------------------------------------------------------------------
def dwt(direction, reversible, descs: np.ndarray, arena, max_w, max_h):
    """descs: dwt_desc_dtype array (host); arena: torch int32/float32/uint32 tensor on the device."""
    torch = _torch()
    dev = arena.device.index
    d = to_device(descs, dev)
    f = capi.lib().ojphgpu_dwt_forward if direction == "forward" else capi.lib().ojphgpu_dwt_inverse
    check(f(_stream_ptr(torch, dev), int(reversible), C.c_void_p(d.data_ptr()), len(descs), max_w, max_h,
            C.c_void_p(arena.data_ptr())), "dwt_" + direction)
    torch.cuda.synchronize(dev)


class _LiftStep(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("e", C.c_int32), ("A", C.c_float)]


class _Lift(C.Structure):
    _fields_ = [("num_steps", C.c_uint32), ("elem", C.c_uint32), ("horz", C.c_uint32), ("vert", C.c_uint32), ("K", C.c_float),
                ("steps", _LiftStep * 16)]


def dwt_general(direction, steps, elem, descs: np.ndarray, arena, max_w, max_h, K=1.0, horz=True, vert=True):
    """ojphgpu_dwt_forward_general / _inverse_general: steps in synthesis order -- (a, b, e) tuples for a reversible
    kernel, floats for an irreversible one; elem 0 = int32, 1 = int64, 2 = float planes in `arena`."""
    torch = _torch()
    dev = arena.device.index
    d = to_device(descs, dev)
    k = _Lift()
    k.num_steps, k.elem, k.horz, k.vert, k.K = len(steps), int(elem), int(bool(horz)), int(bool(vert)), float(K)
    for i, st in enumerate(steps):
        if isinstance(st, (tuple, list)):
            k.steps[i].a, k.steps[i].b, k.steps[i].e = st
        else:
            k.steps[i].A = float(st)
    f = capi.lib().ojphgpu_dwt_forward_general if direction == "forward" else capi.lib().ojphgpu_dwt_inverse_general
    check(f(_stream_ptr(torch, dev), C.byref(k), C.c_void_p(d.data_ptr()), len(descs), max_w, max_h, C.c_void_p(arena.data_ptr())),
          "dwt_general_" + direction)
    torch.cuda.synchronize(dev)


def ht_encode(descs: np.ndarray, coef, scratch_bytes, out_cap):
    """Returns (results array, out bytes tensor (host numpy), status)."""
    torch = _torch()
    dev = coef.device.index
    d = to_device(descs, dev)
    scratch = torch.zeros(max(int(scratch_bytes), 16), dtype=torch.uint8, device=coef.device)
    out = torch.zeros(max(int(out_cap), 16), dtype=torch.uint8, device=coef.device)
    results = torch.zeros(len(descs) * 2 + 2, dtype=torch.int32, device=coef.device)
    counters = torch.zeros(4, dtype=torch.int32, device=coef.device)
    check(capi.lib().ojphgpu_ht_encode(_stream_ptr(torch, dev), C.c_void_p(d.data_ptr()), len(descs),
                                       C.c_void_p(coef.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                       C.c_void_p(out.data_ptr()), int(out_cap), C.c_void_p(results.data_ptr()),
                                       C.c_void_p(counters.data_ptr()), C.c_void_p(counters.data_ptr() + 4)),
          "ht_encode")
    torch.cuda.synchronize(dev)
    res = results.cpu().numpy()[:len(descs) * 2].view(np.uint32).reshape(-1, 2)
    cnt = counters.cpu().numpy().view(np.uint32)
    return res, out.cpu().numpy()[:int(cnt[0])], int(cnt[1])


def ht_decode(descs: np.ndarray, data: np.ndarray, coef):
    torch = _torch()
    dev = coef.device.index
    L = capi.lib()
    descs = np.ascontiguousarray(descs.copy())
    q, a = C.c_uint64(), C.c_uint64()             # per-quad record / aux offsets (see include/ojphgpu.h)
    check(L.ojphgpu_ht_decode_layout(descs.ctypes.data, len(descs), C.byref(q), C.byref(a)), "ht_decode_layout")
    qoff, aoff = int(q.value), int(a.value)
    d = to_device(descs, dev)
    dd = to_device(np.concatenate([np.asarray(data, np.uint8), np.zeros(64, np.uint8)]), dev)
    status = torch.zeros(len(descs) + 16, dtype=torch.uint8, device=coef.device)
    quads = torch.zeros(qoff + 16, dtype=torch.int32, device=coef.device)
    aux = torch.zeros(aoff + 16, dtype=torch.int32, device=coef.device)
    check(L.ojphgpu_ht_decode(_stream_ptr(torch, dev), C.c_void_p(d.data_ptr()), len(descs),
                              C.c_void_p(dd.data_ptr()), C.c_void_p(coef.data_ptr()),
                              C.c_void_p(quads.data_ptr()), C.c_void_p(aux.data_ptr()),
                              C.c_void_p(status.data_ptr())),
          "ht_decode")
    torch.cuda.synchronize(dev)
    return status.cpu().numpy()[:len(descs)]


def unpack_pixels(pixels, num_comps=None, big_endian=False, dtype=None):
    """pixel-interleaved samples on the device ([H,W,C] uint8 / uint16 tensor -- for big-endian 16-bit data the tensor
    holds the file's bytes as they are) -> planes [C,H,W] in `dtype` (uint8 / int16 / uint16 / int32; default: as the
    input).  ojphgpu_unpack_pixels: what the reference's image readers do sample by sample on the host."""
    torch = _torch()
    assert pixels.is_cuda and pixels.is_contiguous() and pixels.dim() == 3
    h, w, c = pixels.shape
    bits = pixels.element_size() * 8
    assert bits in (8, 16)
    out_dt = dtype if dtype is not None else (torch.uint8 if bits == 8 else torch.int16)
    out = torch.empty((c, h, w), dtype=out_dt, device=pixels.device)
    check(capi.lib().ojphgpu_unpack_pixels(_stream_ptr(torch, pixels.device.index or 0), C.c_void_p(pixels.data_ptr()),
                                           C.c_void_p(out.data_ptr()), w, h, c, bits, int(bool(big_endian)), out.element_size() * 8),
          "unpack_pixels")
    return out


def pack_pixels(planes, bit_depth, pixel_bits=None, big_endian=False):
    """planes [C,H,W] on the device -> pixel-interleaved [H,W,C] (uint8 for pixel_bits 8, else int16 holding the bytes
    of uint16 samples, byte-swapped when big_endian), clamped to [0, 2^bit_depth - 1] as the reference's writers do."""
    torch = _torch()
    assert planes.is_cuda and planes.is_contiguous() and planes.dim() == 3
    c, h, w = planes.shape
    pb = int(pixel_bits) if pixel_bits else (8 if bit_depth <= 8 else 16)
    out = torch.empty((h, w, c), dtype=torch.uint8 if pb == 8 else torch.int16, device=planes.device)
    check(capi.lib().ojphgpu_pack_pixels(_stream_ptr(torch, planes.device.index or 0), C.c_void_p(planes.data_ptr()),
                                         C.c_void_p(out.data_ptr()), w, h, c, planes.element_size() * 8, pb, int(bool(big_endian)),
                                         int(bit_depth)), "pack_pixels")
    return out

