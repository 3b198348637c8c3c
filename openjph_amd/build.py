"""Builds openjph_amd/libojphgpu.so (HIP kernels for gfx950 + host plan/Tier-2 + C ABI) in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libojphgpu.so")
SOURCES = ["ojph_plan.cpp", "ojph_t2.cpp", "ojph_pool.cpp", "ht_tables.cpp", "ojphgpu_codec.cpp", "ojphgpu_pipe.cpp", "ojphgpu_multi.cpp",
           "kernels_dwt.hip", "kernels_lift.hip", "kernels_convert.hip", "kernels_assemble.hip", "kernels_pixels.hip", "kernels_ht_enc.hip", "kernels_ht_dec.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"]


def kernel_sources_digest():
    """sha256 over the device code of the library as written (every .hip source and the headers / tables they include, in
    name order; // comments, blank lines and indentation do not count): the identity of the kernels a counter pass was
    taken on.  tools/pmc_summary.py and tools/sq_round.sh stamp their JSON with it; bench.py refuses counters whose stamp is
    not the digest of the sources it runs."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.startswith("_"):
            continue                                      # scratch copies of experiments (never part of the library)
        if f.endswith((".hip", ".inc")) or (f.endswith(".h") and f.startswith(("ht_", "kernels_"))):
            h.update(f.encode())
            for line in open(os.path.join(CSRC, f), "r", errors="replace"):
                line = re.sub(r"//.*$", "", line).strip()                 # (no string literal of these sources holds "//")
                if line:
                    h.update(re.sub(r"\s+", " ", line).encode()); h.update(b"\n")
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith("_")]
    deps.append(os.path.join(HERE, "..", "include", "ojphgpu.h"))
    deps.append(os.path.join(HERE, "..", "include", "ojph_gpu_codestream.h"))
    deps += [os.path.join(HERE, "apps", f) for f in os.listdir(os.path.join(HERE, "apps")) if f.endswith((".cpp", ".h"))]
    fdir = os.path.join(HERE, "..", "tests", "facade")
    checks = sorted(f for f in os.listdir(fdir) if f.endswith(".cpp")) if os.path.isdir(fdir) else []
    deps += [os.path.join(fdir, f) for f in checks]
    outs = [os.path.join(HERE, "libopenjph_gpu.so"), os.path.join(HERE, "apps", "ojph_compress"), os.path.join(HERE, "apps", "ojph_expand")]
    outs += [os.path.join(HERE, "apps", "facade_" + f[:-4]) for f in checks]
    if not all(os.path.exists(x) for x in outs):
        return True
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objs = []
    procs = []
    bdir = os.path.join(HERE, "csrc", "_build")
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(bdir, src + ".o")
        cmd = [HIPCC, "-x", "hip"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % src)
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT] + objs
    subprocess.check_call(cmd)
    build_facade(verbose)
    return OUT


FACADE = os.path.join(HERE, "libopenjph_gpu.so")
APPS = os.path.join(HERE, "apps")


def build_facade(verbose=False):
    """libopenjph_gpu.so (the ojph::codestream-compatible C++ facade over the C ABI) and the two
    command-line tools, all linked against libojphgpu.so with an $ORIGIN rpath."""
    cmds = [[HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
             os.path.join(CSRC, "ojph_facade.cpp"), "-o", FACADE, "-L" + HERE, "-lojphgpu", "-Wl,-rpath,$ORIGIN"]]
    for app in ("ojph_compress", "ojph_expand"):
        cmds.append([HIPCC, "-O2", "-std=c++17", "-Wall", os.path.join(APPS, app + ".cpp"), "-o", os.path.join(APPS, app),
                     "-L" + HERE, "-lopenjph_gpu", "-lojphgpu", "-Wl,-rpath,$ORIGIN/.."])
    # facade scenarios the tests run on the GPU box (tests/test_cli.py)
    fdir = os.path.join(os.path.dirname(HERE), "tests", "facade")
    for f in sorted(os.listdir(fdir)) if os.path.isdir(fdir) else []:
        if f.endswith(".cpp"):
            cmds.append([HIPCC, "-O2", "-std=c++17", "-Wall", os.path.join(fdir, f), "-o", os.path.join(APPS, "facade_" + f[:-4]),
                         "-L" + HERE, "-lopenjph_gpu", "-lojphgpu", "-Wl,-rpath,$ORIGIN/.."])
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
