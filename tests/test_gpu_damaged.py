"""The HIP decode path on damaged input, against the LIVE REFERENCE's committed verdicts.

* tests/golden/damaged.json (made by tests/golden/make_damaged.py where /root/reference exists) holds, for each of the 240
  damaged codestreams of tests/damaged_cases.py and for both values of `resilient`, what the reference's generic build does
  with it: "raises", or the digest of the picture it returns.  The expected value here is THAT record -- not the CPU
  pipeline's result (tests/test_cpu_damaged.py pins the CPU pipeline to the same file).
* a slice of tools/fuzz_blocks_gpu.py: damaged cleanup / refinement segments through the block decoder launches, the
  oracle's verdict and samples expected (tools/fuzz_blocks_cpu.py pins the oracle to the reference on the same blocks).
"""
import json
import os
import sys

import numpy as np
import pytest

from tests.damaged_cases import SOURCES, cases, digest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "damaged.json")))["cases"]
KEYS = ["s%d_t%02d_%d" % (i, t, r) for i in range(len(SOURCES)) for t in range(60) for r in (0, 1)]
KEYS = [k for k in KEYS if k in GOLD]


@pytest.fixture(scope="module")
def streams():
    return dict(cases())


@pytest.mark.parametrize("key", KEYS)
def test_damaged_codestream_reads_like_the_reference(key, streams):
    from openjph_amd import capi, codec
    name, resilient = key[:-2], key.endswith("_1")
    try:
        dec = codec.Decoder(streams[name], resilient=resilient)
        frame = np.asarray(dec.decode())
        # (a damaged SIZ may give the components different sizes: the reference's planes one by one, like the committed digest)
        got = digest(frame if len(dec.plan.frame_shape) == 3 else dec.plan.unpack_frame(frame))
    except (capi.OjphError, RuntimeError):
        got = "raises"
    assert got == GOLD[key], "reference: %s, HIP decoder: %s" % (GOLD[key][:16], got[:16])


def test_the_fixture_set_is_complete():
    assert len(KEYS) == 2 * 60 * len(SOURCES) == 480


@pytest.mark.parametrize("seed", [700000, 700100])
def test_damaged_blocks_through_the_block_decoder_launches(seed):
    """ten seconds of tools/fuzz_blocks_gpu.py per seed range: 32- and 64-bit sample paths, cleanup and refinement launches"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_blocks_gpu
    sys.argv = ["fuzz_blocks_gpu.py", "10", str(seed)]
    assert fuzz_blocks_gpu.main() == 0


def test_padded_block_is_decoded_on_the_device():
    """a tile-part one byte short of its last block's cleanup segment, with a plausible Scup: the reference decodes the block from
    the bytes there are and a zero (bb_read_chunk, ojph_bitbuffer_read.h:134-150) -- and so does the device, from the padded
    copy the upload places behind the codestream's byte range (ojphgpu_decoder_upload_pads).  The oracle pipeline is the
    expected value (tests/test_cpu_damaged.py::test_padded_blocks_are_listed pins it to the live reference)."""
    from openjph_amd import codec
    from openjph_amd.plan import parse_codestream
    from tests import cpu_pipeline as cp
    from tests.synth import synth_image
    img = synth_image(1, 72, 88, 8, seed=5)
    cs = bytes(cp.encode(img, bit_depth=8, num_decomps=3, block=(16, 16), prog_order="LRCP", tileparts="R")[0])
    sots = [i for i in range(len(cs) - 12) if cs[i:i + 4] == b"\xff\x90\x00\x0a"]
    hits = 0
    for t in range(4):
        for short in (1, 2):
            # tile-part t claims to be `short` bytes shorter than it is: its last block gets that many bytes less
            part = bytearray(cs)
            psot = int.from_bytes(cs[sots[t] + 6:sots[t] + 10], "big") - short
            part[sots[t] + 6:sots[t] + 10] = psot.to_bytes(4, "big")
            part = bytes(part)
            for resilient in (False, True):
                try:
                    pl = parse_codestream(part, resilient=resilient)
                    want = np.asarray(cp.inverse_stages(pl, cp.decode_blocks(pl, part, resilient=resilient)))
                except Exception:
                    want = None
                try:
                    got = np.asarray(codec.Decoder(part, resilient=resilient).decode())
                except Exception:
                    got = None
                assert (want is None) == (got is None), (t, short, resilient)
                if want is not None:
                    assert np.array_equal(got, want), (t, short, resilient)
                    hits += len(pl.padded_blocks()) > 0
    assert hits == 8               # (one byte short: the block is decoded from its bytes and a zero, in every tile-part, both modes)
