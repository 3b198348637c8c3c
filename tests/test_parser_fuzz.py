"""The codestream parser under AddressSanitizer / UBSan (marker segments shorter than
their fixed fields made it read past the buffer).  Host-only sources, compiled here with g++."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.synth import synth_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openjph_amd", "csrc")


def _asan_works(tmp):
    src = os.path.join(tmp, "probe.cpp")
    with open(src, "w") as f:
        f.write("int main(){return 0;}\n")
    exe = os.path.join(tmp, "probe")
    if subprocess.call(["g++", "-fsanitize=address,undefined", src, "-o", exe], stderr=subprocess.DEVNULL) != 0:
        return False
    return subprocess.call([exe]) == 0


def test_parser_fuzz_under_asan(tmp_path):
    tmp = str(tmp_path)
    if shutil.which("g++") is None or not _asan_works(tmp):
        pytest.skip("g++ with -fsanitize=address is not usable here")
    exe = os.path.join(tmp, "t2_parse_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "fuzz", "t2_parse_fuzz.cpp"), os.path.join(CSRC, "ojph_plan.cpp"),
                           os.path.join(CSRC, "ojph_t2.cpp"), os.path.join(CSRC, "ojph_pool.cpp"), "-o", exe, "-pthread"])
    from tests import cpu_pipeline as cp
    from tests.test_cpu_parity import coc_case, nlt_case
    img = synth_image(3, 70, 90, 8, seed=1)
    seeds = [bytes(cp.encode(img, bit_depth=8, **kw)[0]) for kw in (
        dict(), dict(reversible=False, qstep=0.05), dict(tile=(32, 32), tlm=True, prog_order="CPRL", tileparts="C"),
        dict(color_transform=True, precinct=(32, 32), prog_order="PCRL"))]
    pl, kw, size, _, _ = coc_case(0)
    seeds.append(bytes(cp.encode(pl, size=size, **kw)[0]))
    pl, kw, size = nlt_case(1)
    seeds.append(bytes(cp.encode(pl, size=size, **kw)[0]))
    files = []
    for i, s in enumerate(seeds):
        fn = os.path.join(tmp, "seed%d.j2c" % i)
        with open(fn, "wb") as f:
            f.write(s)
        files.append(fn)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "1200"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-4000:]
    out = r.stdout.decode()
    parsed, refused = int(out.split()[1]), int(out.split()[3])
    assert parsed > 100 and refused > 100, out
