#!/usr/bin/env python3
"""Times the encoder's device part only (ablation / A-B runs of encoder kernel variants; results of ablated
builds are wrong on purpose and never decoded).   python tools/enc_only.py [lib.so ...]"""
import os
import shutil
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "openjph_amd", "libojphgpu.so")

CHILD = r'''
import sys, json, time, numpy as np, torch
sys.path.insert(0, %r)
from bench import workload_image, WORKLOADS
from openjph_amd import codec
from openjph_amd.plan import Plan, make_params
name = "c3_8k_444_12b_irv97"
w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
img = np.load("/tmp/c3.npy") if __import__("os").path.exists("/tmp/c3.npy") else workload_image(name)
np.save("/tmp/c3.npy", img)
d = torch.from_numpy(img.astype(np.int16)).cuda()
enc = codec.Encoder(plan=Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep)))
for _ in range(3): enc.run_device(d)
torch.cuda.synchronize()
acc = None
for _ in range(10):
    enc.run_device(d); t = enc.timing()
    v = [t["total_ms"], t["dwt_ms"]] + t["ht_launches_ms"]
    acc = v if acc is None else [a + b for a, b in zip(acc, v)]
print(json.dumps([round(a / 10, 4) for a in acc] + [enc.coded_bytes()]))
''' % ROOT

orig = "/tmp/lib_enc_only_orig.so"
shutil.copy(LIB, orig)
try:
    for rep in range(2):
        for v in sys.argv[1:] or ["orig"]:
            shutil.copy(orig if v == "orig" else os.path.join(ROOT, "openjph_amd", "variants", "lib_%s.so" % v), LIB)
            r = subprocess.run([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            out = r.stdout.decode().strip().splitlines()
            print("%-12s total/dwt/ht... %s" % (v, out[-1] if out else r.stderr.decode()[-300:]), flush=True)
finally:
    shutil.copy(orig, LIB)
