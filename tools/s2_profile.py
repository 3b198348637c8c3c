#!/usr/bin/env python3
"""experiment: where a worker wavefront of the fused block-decoder launch spends its time (8K bench frame).
The library variant comes from the product sources + tools/s2_profile.patch (s_memtime stamps around the wait for the
chain, the slice's set-up -- state, loads, un-stuffing -- and its row loop):
    patch -p0 < tools/s2_profile.patch && python tools/build_variant.py s2prof kernels_ht_dec.hip && git checkout openjph_amd/csrc/kernels_ht_dec.hip"""
import ctypes as C, os, shutil, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openjph_amd", "libojphgpu.so")
shutil.copy(LIB, "/tmp/lib_s2_orig.so")
shutil.copy(os.path.join(ROOT, "openjph_amd", "variants", "lib_s2prof.so"), LIB)
try:
    from bench import workload_image, WORKLOADS
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    name = "c3_8k_444_12b_irv97"
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    img = workload_image(name)
    d = torch.from_numpy(img.astype(np.int16)).cuda()
    enc = codec.Encoder(plan=Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep)))
    cs = enc.encode(d)
    dec = codec.Decoder(cs)
    out = torch.empty_like(d)
    L = C.CDLL(LIB)
    st = (C.c_ulonglong * 8)()
    for _ in range(3):
        dec.run_device(out)
    torch.cuda.synchronize()
    L.ojphgpu_debug_s2_prof(st, 1)
    dec.run_device(out); torch.cuda.synchronize()
    L.ojphgpu_debug_s2_prof(st, 1)
    wait, work, seg, setup, rows, wall, waves = [int(st[i]) for i in range(7)]
    # s_memtime ticks (the counter runs at the shader clock here, not at 100 MHz: shares are what counts)
    print("worker wavefronts %d, slices of a block decoded %d (%.1f per wavefront)" % (waves, seg, seg / max(waves, 1)))
    print("per wavefront: %.0f ticks from its first instruction to its last; waiting for the chains %.1f %%, decoding %.1f %%"
          % (wall / waves, 100.0 * wait / wall, 100.0 * work / wall))
    print("per slice (8 quad rows of one block): wait %.0f ticks, set-up (state, loads arrive, un-stuffing) %.0f, row loop %.0f"
          % (wait / seg, setup / seg, rows / seg))
    print(dec.timing())
    sys.stdout.flush(); os._exit(0)
finally:
    shutil.copy("/tmp/lib_s2_orig.so", LIB)
