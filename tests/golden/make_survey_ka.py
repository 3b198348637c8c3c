#!/usr/bin/env python3
"""Makes tests/golden/survey_ka.json: the BASELINE workloads at their stated sizes (SURVEY.md section
8(d) inputs, tests/synth.py survey_c*) coded by the REAL reference built from /root/reference
(oracle/_ref, `make -C oracle ref`).  Run in the build container; the GPU box has no /root/reference
and checks the HIP path against these digests (tests/test_gpu_fullsize.py).

    python tests/golden/make_survey_ka.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refbind            # noqa: E402
from tests import synth               # noqa: E402


def sha(a):
    return hashlib.sha256(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()).hexdigest()


def quality(dec, img):
    d = dec.astype(np.int64) - img
    return float((d * d).mean()), int(np.abs(d).max())


def main():
    ref, gen = refbind.Ref(), refbind.Ref(generic=True)
    out = {"made_by": "tests/golden/make_survey_ka.py", "reference": "aous72/OpenJPH 0.31.0 built by oracle/Makefile"}

    img = synth.survey_c2()
    cs = ref.encode(img, 8, reversible=True, color_transform=True, planar=False)
    assert cs == gen.encode(img, 8, reversible=True, color_transform=True, planar=False)
    assert len(cs) == 16674994, len(cs)
    out["c2"] = {"image_sha256": sha(img), "bytes": len(cs), "sha256": sha(cs)}
    print("c2", out["c2"], flush=True)

    img = synth.survey_c3()
    e = {"image_sha256": sha(img)}
    for name, r in (("generic", gen), ("simd", ref)):
        cs = r.encode(img, 12, reversible=False, color_transform=False, qstep=0.001)
        dec, _ = r.decode(cs)
        mse, pae = quality(dec, img)
        e[name] = {"bytes": len(cs), "sha256": sha(cs), "decoded_sha256": sha(dec.astype(np.int32)), "mse": mse, "pae": pae}
    assert e["generic"]["bytes"] == 72601187 and e["simd"]["bytes"] == 72601177      # KA-4
    out["c3"] = e
    print("c3", e, flush=True)

    frames = []
    for f in range(8):
        img = synth.survey_c5(f)
        cs = gen.encode(img, 10, reversible=False, color_transform=False)
        dec, _ = gen.decode(cs)
        mse, pae = quality(dec, img)
        fr = {"image_sha256": sha(img), "irv": {"bytes": len(cs), "sha256": sha(cs), "decoded_sha256": sha(dec.astype(np.int32)),
                                                 "mse": mse, "pae": pae}}
        if f < 2:
            cs = ref.encode(img, 10, reversible=True, color_transform=False)
            fr["rev"] = {"bytes": len(cs), "sha256": sha(cs)}
        frames.append(fr)
        print("c5 frame", f, fr, flush=True)
    out["c5"] = frames

    img = synth.survey_c4()
    cs = ref.encode(img, 16, reversible=True, tile=(1024, 1024))
    out["c4"] = {"image_sha256": sha(img), "bytes": len(cs), "sha256": sha(cs)}
    print("c4", out["c4"], flush=True)

    with open(os.path.join(ROOT, "tests", "golden", "survey_ka.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
