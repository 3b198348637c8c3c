#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json from the REAL reference (oracle/_ref/libojph_ref*.so built from
/root/reference by oracle/Makefile).  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The fixtures pin (a) single code-block HT cleanup encodes (input seed -> coded bytes), (b) whole
codestream digests for small images over a spread of parameters, (c) 5/3 and 9/7 one-level DWT
outputs, so that `-m "not gpu"` tests can check the oracle on machines where the reference is
absent (the GPU box).  Inputs are regenerated from seeds by tests/synth.py; only outputs are stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refbind                      # noqa: E402
from tests.synth import synth_image, c1_image, random_block, ka2_block   # noqa: E402
from tests.golden_cases import (BLOCK_CASES, STREAM_CASES, REFINE_CASES, GRID_CASES, SKIP_CASES, TILEPART_CASES,  # noqa: E402
                                FORMAT_CASES, COC_CASES, coc_case, NLT_CASES, nlt_case, CQF_CASES, cqf_case, stream_kwargs, refine_case, grid_kwargs, skip_case, tilepart_case, format_case)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    ref = refbind.Ref()
    refgen = refbind.Ref(generic=True)
    out = {"reference": "aous72/OpenJPH 0.31.0 (oracle/_ref, simd level %d)" % ref.simd_level(),
           "blocks": [], "streams": []}
    blobs = {}
    # (a) code-blocks
    buf = ka2_block()
    b = ref.encode_block(buf, 9, 64, 64, 64)
    assert b == refgen.encode_block(buf, 9, 64, 64, 64)
    out["ka2"] = {"len": len(b), "sha256": sha(b), "fnv1a64": refbind.fnv1a64(b)}
    blobs["ka2"] = np.frombuffer(b, np.uint8)
    for i, (w, h, kmax, density, amp, seed) in enumerate(BLOCK_CASES):
        rng = np.random.default_rng(seed)
        stride = (w + 15) // 16 * 16
        q, _ = random_block(rng, w, h, stride, kmax, density, amp)
        q[:, w:] = 0
        b = ref.encode_block(q, kmax - 1, w, h, stride)
        ok, dec = ref.decode_block(b, kmax - 1, w, h, stride)
        assert ok
        out["blocks"].append({"case": i, "len": len(b), "sha256": sha(b),
                              "dec_sha256": sha(np.ascontiguousarray(dec[:, :w]).tobytes())})
        if w * h <= 1024:
            blobs["block%d" % i] = np.frombuffer(b, np.uint8)
    # (a2) blocks with SigProp / MagRef segments (random refinement bytes behind a reference-coded cleanup pass)
    out["refine"] = []
    for i in range(len(REFINE_CASES)):
        q, w, h, stride, kmax, npass, causal, tail = refine_case(i)
        cup = ref.encode_block(q, kmax - 1, w, h, stride)
        ok, dec = ref.decode_block(cup + tail, kmax - 1, w, h, stride, len2=len(tail), num_passes=npass, stripe_causal=causal)
        ok1, dec1 = ref.decode_block(cup + tail, kmax - 1, w, h, stride, len2=len(tail), num_passes=npass, variant=1,
                                     stripe_causal=causal)
        assert ok and ok1 and np.array_equal(dec[:, :w], dec1[:, :w])
        out["refine"].append({"case": i, "dec_sha256": sha(np.ascontiguousarray(dec[:, :w]).tobytes())})
    # (b) codestreams
    for i, case in enumerate(STREAM_CASES):
        img, kw = stream_kwargs(case)
        r = ref if kw.get("reversible", True) else refgen        # 9/7: the generic build is the pin
        cs = r.encode(img, **kw)
        dec, _ = r.decode(cs)
        out["streams"].append({"case": i, "len": len(cs), "sha256": sha(cs),
                               "dec_sha256": sha(dec.astype(np.int32).tobytes())})
    # (b2) general reference grid: sub-sampling, image / tile offsets
    out["grid"] = []
    for i, case in enumerate(GRID_CASES):
        planes, kw, size = grid_kwargs(case)
        r = ref if kw.get("reversible", True) else refgen
        cs = r.encode(planes, size=size, **kw)
        dec, _ = r.decode(cs)
        dec = [dec[c] for c in range(len(planes))]
        assert [d.shape for d in dec] == [q.shape for q in planes]
        out["grid"].append({"case": i, "len": len(cs), "sha256": sha(cs),
                            "dec_sha256": sha(b"".join(np.ascontiguousarray(d, dtype=np.int32).tobytes() for d in dec))})
    # (b4) tile-part divisions, user COM segment, BROADCAST profile
    out["tileparts"] = []
    for i in range(len(TILEPART_CASES)):
        img, kw = tilepart_case(i)
        r = ref if kw.get("reversible", True) else refgen
        cs = r.encode(img, **kw)
        out["tileparts"].append({"case": i, "len": len(cs), "sha256": sha(cs)})
    # (b5) per-component bit depth / signedness (QCC) and qfactor
    out["formats"] = []
    for i in range(len(FORMAT_CASES)):
        planes, kw, size = format_case(i)
        kw = dict(kw)
        bd, sg = kw.pop("bit_depth"), kw.pop("is_signed")
        r = ref if kw.get("reversible", True) else refgen
        cs = r.encode(planes, bd, is_signed=sg, size=size, **kw)
        dec, _ = r.decode(cs)
        dec = [dec[c] for c in range(len(planes))]
        out["formats"].append({"case": i, "len": len(cs), "sha256": sha(cs),
                               "dec_sha256": sha(b"".join(np.ascontiguousarray(d, dtype=np.int32).tobytes() for d in dec))})
    # (b6) per-component coding styles (COC)
    out["coc"] = []
    for i in range(len(COC_CASES)):
        planes, kw, size, skip, resilient = coc_case(i)
        kw = dict(kw)
        bd, sg = kw.pop("bit_depth"), kw.pop("is_signed")
        all_rev = kw.get("reversible", True) and all(st.get("reversible", False) for st in kw["coc"].values())
        r = ref if all_rev else refgen
        cs = r.encode(planes, bd, is_signed=sg, size=size, **kw)
        assert len(cs) > 0
        dec, _ = r.decode(cs, resilient=resilient, skip=skip or (0, 0))
        dec = [dec[c] for c in range(len(planes))]
        out["coc"].append({"case": i, "len": len(cs), "sha256": sha(cs), "shapes": [list(d.shape) for d in dec],
                           "dec_sha256": sha(b"".join(np.ascontiguousarray(d, dtype=np.int32).tobytes() for d in dec))})
    # (b7) NLT marker segments (type 3 non-linearity)
    out["nlt"] = []
    for i in range(len(NLT_CASES)):
        planes, kw, size = nlt_case(i)
        kw = dict(kw)
        bd, sg = kw.pop("bit_depth"), kw.pop("is_signed")
        all_rev = kw.get("reversible", True) and all(st.get("reversible", False) for st in kw.get("coc", {}).values())
        r = ref if all_rev else refgen
        cs = r.encode(planes, bd, is_signed=sg, size=size, **kw)
        assert len(cs) > 0
        dec, _ = r.decode(cs)
        dec = [dec[c] for c in range(len(planes))]
        out["nlt"].append({"case": i, "len": len(cs), "sha256": sha(cs),
                           "dec_sha256": sha(b"".join(np.ascontiguousarray(d, dtype=np.int32).tobytes() for d in dec))})
    # (b8) quality factors of single components
    out["cqf"] = []
    for i in range(len(CQF_CASES)):
        img, kw = cqf_case(i)
        kw = dict(kw)
        bd = kw.pop("bit_depth")
        cs = refgen.encode(img, bd, **kw)
        assert len(cs) > 0
        dec, _ = refgen.decode(cs)
        out["cqf"].append({"case": i, "len": len(cs), "sha256": sha(cs),
                           "dec_sha256": sha(np.ascontiguousarray(dec, dtype=np.int32).tobytes())})
    # (b3) reduced-resolution decoding
    out["skip"] = []
    for i in range(len(SKIP_CASES)):
        planes, kw, size, skip = skip_case(i)
        r = ref if kw.get("reversible", True) else refgen
        cs = r.encode(planes if size else np.stack(planes), **(dict(kw, size=size) if size else kw))
        dec, _ = r.decode(cs, skip=skip)
        dec = [dec[c] for c in range(len(planes))]
        out["skip"].append({"case": i, "shapes": [list(d.shape) for d in dec],
                            "dec_sha256": sha(b"".join(np.ascontiguousarray(d, dtype=np.int32).tobytes() for d in dec))})
    cs = ref.encode(c1_image(), 8)
    out["ka1"] = {"len": len(cs), "sha256": sha(cs), "fnv1a64": refbind.fnv1a64(cs)}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "golden_blocks.npz"), **blobs)
    print("wrote golden.json (%d blocks, %d streams) and golden_blocks.npz" % (len(out["blocks"]), len(out["streams"])))


if __name__ == "__main__":
    main()
