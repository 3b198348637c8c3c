"""Damaged codestreams whose reading by the LIVE reference is pinned in tests/golden/damaged.json (made by
tests/golden/make_damaged.py where /root/reference exists).  The sources come from the oracle pipeline's encoder (byte-identical
to the reference's, tests/test_cpu_parity.py), so the cases can be rebuilt where the reference is absent: only its verdicts
and picture digests are stored."""
import hashlib
import numpy as np

from tests import cpu_pipeline as cp
from tests.synth import synth_image

SOURCES = [
    dict(nc=1, h=72, w=88, bd=8, seed=5, kw=dict(num_decomps=3, block=(16, 16), prog_order="LRCP", tileparts="R")),
    dict(nc=3, h=40, w=52, bd=10, seed=7, kw=dict(num_decomps=2, block=(32, 32), prog_order="RPCL", color_transform=True, tile=(32, 32))),
    dict(nc=1, h=61, w=47, bd=8, seed=9, kw=dict(num_decomps=4, block=(64, 64), prog_order="CPRL", reversible=False, qstep=0.02)),
    dict(nc=2, h=33, w=90, bd=12, seed=3, kw=dict(num_decomps=1, block=(8, 8), prog_order="PCRL", tlm=True, tileparts="C")),
]


def source(i):
    s = SOURCES[i]
    return bytes(cp.encode(synth_image(s["nc"], s["h"], s["w"], s["bd"], seed=s["seed"]), bit_depth=s["bd"], **s["kw"])[0])


def cases():
    """-> (name, bytes) in a fixed order"""
    for i in range(len(SOURCES)):
        cs = source(i)
        sot = cs.find(b"\xff\x90\x00\x0a")
        rng = np.random.default_rng(1000 + i)
        for t in range(60):
            b = bytearray(cs)
            kind = t % 4
            if kind == 0:                                    # behind the first SOD
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(sot + 14, len(b)))] = int(rng.choice([0xFF, 0x00, 0x90, 0x7F, int(rng.integers(0, 256))]))
            elif kind == 1:                                  # SOT segments
                sots = [k for k in range(sot, len(cs) - 14) if cs[k:k + 4] == b"\xff\x90\x00\x0a"]
                at = sots[int(rng.integers(0, len(sots)))]
                b[at + int(rng.integers(0, 14))] = int(rng.choice([0xFF, 0x00, 0x01, 0x90, 0x93, int(rng.integers(0, 256))]))
            elif kind == 2:                                  # a cut
                b = b[:int(rng.integers(sot - 4, len(b)))]
            else:                                            # the main header (bytes that decide sizes are left alone: offsets 4..40 of SIZ)
                siz = cs.find(b"\xff\x51")
                while True:
                    at = int(rng.integers(0, sot + 2))
                    if not (siz + 4 <= at < siz + 40):
                        break
                b[at] = int(rng.choice([0xFF, 0x00, 0x01, 0x52, 0x90, int(rng.integers(0, 256))]))
            yield "s%d_t%02d" % (i, t), bytes(b)


def digest(planes):
    h = hashlib.sha256()
    for p in (planes if isinstance(planes, list) else [planes]):
        a = np.ascontiguousarray(np.asarray(p), dtype=np.int32)
        h.update(np.asarray(a.shape, np.int64).tobytes()); h.update(a.tobytes())
    return h.hexdigest()
