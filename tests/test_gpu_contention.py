"""The one-launch block decoder (ht_dec_fused_kernel: step-1 chains and persistent step-2 workers in one grid) must not
depend on the order in which workgroups are dispatched, nor on having the chip to itself: roles are dealt by a ticket
taken when a workgroup STARTS, a worker's wait for its chain is bounded by time, and a wait that runs out marks the run
for a repeat through the separate launches instead of failing blocks (kernels_ht_dec.hip, ojphgpu_codec.cpp).  These
tests run the launch beside other fused launches, beside an encoder pipe and on a chip that a third stream holds for
tens of milliseconds, and drive the repeat path with a test switch.  Parity bar as everywhere: the oracle's samples
(reference: ojph_decode_codeblock32, ojph_block_decoder32.cpp:742-1316; verdict handling ojph_codeblock.cpp:190-224)."""
import ctypes
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from tests.synth import synth_image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hog():
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "support", "libhog.so"))
    lib.ojph_test_hog.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ojph_test_hog.restype = ctypes.c_int
    return lib


def test_fused_launches_beside_each_other_and_under_a_held_chip():
    """a decoder pipe with two decoder objects (two fused launches in flight on two streams), an encoder pipe and a
    kernel that holds every wavefront slot of every CU for 20 ms at a time on a third stream (60 times a fused launch;
    every launch of the two pipes may queue behind one such hold, which is what the test's time goes to): every frame comes back
    as the oracle decodes it, no block fails, nothing had to be repeated"""
    import torch
    from openjph_amd import codec
    from openjph_amd.pipeline import DecoderPipe, EncoderPipe
    from openjph_amd.plan import Plan, make_params
    from tests import cpu_pipeline as cp
    w, h, nc, bd = 1024, 768, 3, 10
    kw = dict(bit_depth=bd, reversible=False, qstep=0.004)
    plan = Plan(make_params(w, h, nc, **kw))
    frames = [synth_image(nc, h, w, bd, seed=700 + f) for f in range(4)]
    streams = [codec.Encoder(plan=plan).encode(f) for f in frames]
    want = [cp.decode(cs)[0] for cs in streams]
    n = 12
    hog = _hog()
    hs = torch.cuda.Stream()
    stop = threading.Event()

    def hold():
        torch.cuda.set_device(0)
        while not stop.is_set():
            assert hog.ojph_test_hog(ctypes.c_void_p(hs.cuda_stream), 20, 2, 48) == 0
            hs.synchronize()

    got = {}

    def dec():
        pipe = DecoderPipe(streams[0], depth=4)
        got["d"] = list(pipe.decode_sequence(streams[i % 4] for i in range(n)))
        got["retries"] = pipe.stats()["fused_retries"]
        pipe.close()

    def enc():
        got["e"] = list(EncoderPipe(plan=plan, depth=3).encode_sequence(frames[i % 4] for i in range(n)))

    th = threading.Thread(target=hold)
    te, td = threading.Thread(target=enc), threading.Thread(target=dec)
    th.start(); te.start(); td.start()
    te.join(); td.join()
    stop.set(); th.join()
    assert len(got["d"]) == n and len(got["e"]) == n
    for i in range(n):
        assert got["e"][i] == streams[i % 4], "frame %d: codestream differs under contention" % i
        assert np.array_equal(np.asarray(got["d"][i]).astype(np.int64), want[i % 4].astype(np.int64)), "frame %d differs" % i
    assert got["retries"] == 0
    # and afterwards: small fused launches of fresh decoder objects (five workgroups) find their step-1 workgroup at once --
    # after the streams above had been busy such a launch no longer started on the XCDs it starts on in a fresh process, and
    # a form of the tickets that tied the step-1 numbers to XCDs waited two seconds per frame for a workgroup that never came
    import time
    small = synth_image(3, 100, 150, 7, seed=21, signed=True)
    cs_small = codec.encode(small, bit_depth=7, is_signed=True)
    want_small, _ = cp.decode(cs_small)
    t0 = time.time()
    for _ in range(10):
        d = codec.Decoder(cs_small)
        img = d.run_device()
        assert d.failed_blocks() == 0 and d.fused_retries() == 0
        assert np.array_equal(img.cpu().numpy().astype(np.int64), want_small.astype(np.int64))
    assert time.time() - t0 < 5.0


SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from openjph_amd import codec
from openjph_amd.pipeline import DecoderPipe
from tests import cpu_pipeline as cp
from tests.synth import synth_image
for kw, shape in ((dict(bit_depth=8, num_decomps=2), (1, 700, 900)), (dict(bit_depth=12, reversible=False, qstep=0.002), (3, 520, 640))):
    img = synth_image(shape[0], shape[1], shape[2], kw["bit_depth"], seed=21)
    cs = codec.encode(img, **kw)
    want, _ = cp.decode(cs)
    dec = codec.Decoder(cs)
    for run in range(2):
        d_img = dec.run_device()
        assert dec.failed_blocks() == 0                     # collects the run: the repeat happens here
        assert np.array_equal(d_img.cpu().numpy(), want), (kw, run)
    assert dec.fused_retries() == 2, dec.fused_retries()
    pipe = DecoderPipe(cs, depth=3)
    frames = list(pipe.decode_sequence([cs] * 5))
    st = pipe.stats(); pipe.close()
    assert st["fused_retries"] == 5, st
    for f in frames:
        assert np.array_equal(np.asarray(f).astype(np.int64), want.astype(np.int64))
    # eight repeats in a row and a decoder object stops using the one launch
    dec = codec.Decoder(cs)
    for run in range(11):
        d_img = dec.run_device()
        assert dec.failed_blocks() == 0 and np.array_equal(d_img.cpu().numpy(), want), run
    assert dec.fused_retries() == 8, dec.fused_retries()
    # a caller that never collects: the NEXT run of the object says that the one before it had asked for the repeat
    import torch
    from openjph_amd import capi
    dec = codec.Decoder(cs)
    dec.run_device(); torch.cuda.synchronize()
    try:
        dec.run_device()
        raise SystemExit("an uncollected repeat went unnoticed")
    except capi.OjphError as e:
        assert e.code == capi.E_UNCOLLECTED, e
    d_img = dec.run_device()                                # said once; this run is enqueued
    assert dec.failed_blocks() == 0 and np.array_equal(d_img.cpu().numpy(), want)
    # ... also when the give-up lands late: runs enqueued back to back before any of them has said anything (a real wait runs
    # out after two seconds).  Every give-up that was not collected is reported by SOME later run, once; the epochs bracket it
    dec = codec.Decoder(cs)
    g0, c0 = dec.giveup_epoch()
    assert (g0, c0) == (0, 0)
    notices = 0
    for _ in range(3):
        try:
            dec.run_device()
        except capi.OjphError as e:
            assert e.code == capi.E_UNCOLLECTED, e
            notices += 1
    g1, c1 = dec.giveup_epoch()                             # synchronises: every enqueued run has given up by now
    assert c1 == 3 - notices and g1 == c1 > c0, (g1, c1, notices)
    try:
        dec.run_device()
        assert notices >= 1, "three uncollected give-ups went unnoticed"
    except capi.OjphError as e:
        assert e.code == capi.E_UNCOLLECTED, e
        notices += 1
    assert notices >= 1
    d_img = dec.run_device() if dec.giveup_epoch()[1] == c1 else d_img
    assert dec.failed_blocks() == 0
print("OK")
''' % ROOT


def test_a_wait_that_runs_out_repeats_the_run_through_the_separate_launches():
    """OJPHGPU_FUSED_DBG=4 makes one worker wavefront of every fused launch behave as if its wait for the chains had run
    out in the second slice: the run is marked, the decoder object / the pipe decode the frame again through the
    separate step 1 / step 2 launches when they collect it, and the caller sees the oracle's samples and no failed block;
    after eight repeats in a row a decoder object keeps to the separate launches"""
    env = dict(os.environ, OJPHGPU_FUSED_DBG="4", OJPHGPU_DEC_FUSED="2")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stderr[-3000:]
