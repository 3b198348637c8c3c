#!/usr/bin/env python3
"""Makes tests/golden/foreign.json and copies the two data fixtures it describes out of the reference tree, so that the
GPU box (which has no /root/reference) can run them through the HIP path:

* foreign_test.j2c -- /root/reference/subprojects/js/html/test.j2c, the only codestream of a foreign encoder in the tree:
  512x512x3, 9/7 + ICT, per-resolution precincts, 77 code-blocks with SigProp / MagRef passes (the cleanup-only
  reference encoder never writes those).  The digest of what the reference's generic build decodes from it is the pin
  of ht_dec_refine_kernel (ojph_block_decoder32.cpp:1318-1609).
* fuzz_seed_w128_h128_b2_79_b3_09.bin -- /root/reference/fuzzing/seed_corpus/ojph_compress_fuzz_target/, an ENCODER
  input: 4 control bytes + sample bytes, laid out as fuzzing/fuzz_targets/ojph_compress_fuzz_target.cpp:46-56 reads
  them (here: 128x128, 2 components, 12-bit signed, reversible, 1 decomposition, planar).  The digest of the
  reference's codestream for it is the pin.

Data fixtures, not source.  Run in the build container:   python tests/golden/make_foreign.py
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refbind            # noqa: E402
from tests.golden_cases import fuzz_seed_case   # noqa: E402


def sha(a):
    return hashlib.sha256(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref, gen = refbind.Ref(), refbind.Ref(generic=True)
    out = {"made_by": "tests/golden/make_foreign.py", "reference": "aous72/OpenJPH 0.31.0 built by oracle/Makefile"}
    src = "/root/reference/subprojects/js/html/test.j2c"
    shutil.copyfile(src, os.path.join(HERE, "foreign_test.j2c"))
    os.chmod(os.path.join(HERE, "foreign_test.j2c"), 0o644)
    cs = open(src, "rb").read()
    dec, info = gen.decode(cs)
    dec_simd, _ = ref.decode(cs)
    out["test_j2c"] = {"file": "foreign_test.j2c", "from": "subprojects/js/html/test.j2c", "bytes": len(cs), "sha256": sha(cs),
                       "shape": list(dec.shape), "bit_depth": info["bit_depth"], "reversible": info["reversible"],
                       "decoded_sha256_generic": sha(dec.astype(np.int32)), "decoded_sha256_simd": sha(dec_simd.astype(np.int32)),
                       "max_abs_diff_generic_vs_simd": int(np.abs(dec.astype(np.int64) - dec_simd).max()),
                       "blocks_with_refinement_passes": 77}
    src = "/root/reference/fuzzing/seed_corpus/ojph_compress_fuzz_target/w128_h128_b2_79_b3_09.bin"
    dst = os.path.join(HERE, "fuzz_seed_w128_h128_b2_79_b3_09.bin")
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    img, kw = fuzz_seed_case(open(dst, "rb").read())
    cs = gen.encode(img, kw["bit_depth"], is_signed=kw["is_signed"], reversible=kw["reversible"], num_decomps=kw["num_decomps"],
                    color_transform=kw["color_transform"], planar=kw["planar"], qstep=kw["qstep"])
    assert cs == ref.encode(img, kw["bit_depth"], is_signed=kw["is_signed"], reversible=kw["reversible"], num_decomps=kw["num_decomps"],
                            color_transform=kw["color_transform"], planar=kw["planar"], qstep=kw["qstep"])
    dec, _ = gen.decode(cs)
    assert np.array_equal(dec, img)
    out["fuzz_seed"] = {"file": os.path.basename(dst), "from": "fuzzing/seed_corpus/ojph_compress_fuzz_target/", "params": kw,
                        "image_sha256": sha(img), "bytes": len(cs), "sha256": sha(cs)}
    json.dump(out, open(os.path.join(HERE, "foreign.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
