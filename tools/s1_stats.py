#!/usr/bin/env python3
"""experiment: how often do the step-1 chains of the bench frame wait for their partner wavefronts (library built with -DS1_STATS)"""
import ctypes as C, os, shutil, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openjph_amd", "libojphgpu.so")
shutil.copy(LIB, "/tmp/lib_s1_orig.so")
shutil.copy(os.path.join(ROOT, "openjph_amd", "variants", "lib_s1stats.so"), LIB)
try:
    from bench import workload_image, WORKLOADS
    from openjph_amd import codec, capi
    from openjph_amd.plan import Plan, make_params
    name = "c3_8k_444_12b_irv97"
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    img = np.load("/tmp/c3.npy") if os.path.exists("/tmp/c3.npy") else workload_image(name)
    d = torch.from_numpy(img.astype(np.int16)).cuda()
    enc = codec.Encoder(plan=Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep)))
    cs = enc.encode(d)
    dec = codec.Decoder(cs)
    out = torch.empty_like(d)
    L = C.CDLL(LIB)
    st = (C.c_ulonglong * 8)()
    dec.run_device(out); torch.cuda.synchronize()
    L.ojphgpu_debug_s1_stats(st, 1)
    dec.run_device(out); torch.cuda.synchronize()
    L.ojphgpu_debug_s1_stats(st, 1)
    print("slow VLC fetches %d (poll rounds %d), slow MEL fetches %d (poll rounds %d) for %d blocks" % (st[0], st[1], st[2], st[4], dec.plan.num_blocks))
    n = max(int(st[3]), 1)
    print("per chain wavefront (%d of them), s_memtime ticks (100 MHz): VLC init %.1f, MEL init %.1f, rows %.1f" % (n, st[5] / n, st[6] / n, st[7] / n))
    print(dec.timing())
    sys.stdout.flush(); os._exit(0)
finally:
    shutil.copy("/tmp/lib_s1_orig.so", LIB)
