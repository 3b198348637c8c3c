"""A build from nothing: every source of the product library compiled for gfx950 into a scratch directory (the
in-tree library is mtime-gated and travels pre-built to the GPU box, so nothing else exercises a clean checkout) and
the result exports what include/ojphgpu.h declares."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clean_build_exports_the_abi(tmp_path):
    sys.path.insert(0, ROOT)
    from openjph_amd import build as b
    if not os.path.exists(b.HIPCC):
        import pytest
        pytest.skip("hipcc not installed")
    objs, procs = [], []
    for src in b.SOURCES:
        obj = str(tmp_path / (src + ".o"))
        procs.append((src, subprocess.Popen([b.HIPCC, "-x", "hip"] + b.FLAGS + ["-c", os.path.join(b.CSRC, src), "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, "%s: %s" % (src, out.decode(errors="replace")[-3000:])
    lib = str(tmp_path / "libojphgpu_clean.so")
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib] + objs)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = set(re.findall(r" T (ojphgpu_\w+)", syms))
    header = open(os.path.join(ROOT, "include", "ojphgpu.h")).read()
    declared = set(re.findall(r"\b(ojphgpu_\w+)\s*\(", header))
    declared = {d for d in declared if not d.endswith("_t")}
    missing = declared - exported
    assert not missing, "declared in include/ojphgpu.h but not exported by a clean build: %s" % sorted(missing)
    # the code object is for gfx950 and nothing else
    assert b"gfx950" in open(lib, "rb").read()
