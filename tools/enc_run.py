#!/usr/bin/env python3
"""Runs the encoder's device part on the bench frame a few times (the process a counter pass of tools/enc_counters.sh profiles)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import workload_image, WORKLOADS
from openjph_amd import codec
from openjph_amd.plan import Plan, make_params
name = "c3_8k_444_12b_irv97"
w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
img = np.load("/tmp/c3.npy") if os.path.exists("/tmp/c3.npy") else workload_image(name)
if not os.path.exists("/tmp/c3.npy"):
    np.save("/tmp/c3.npy", img)
d = torch.from_numpy(img.astype(np.int16)).cuda()
os.environ["OJPHGPU_NO_OVERLAP"] = "1"          # one launch over all blocks
enc = codec.Encoder(plan=Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep)))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    enc.run_device(d)
torch.cuda.synchronize()
