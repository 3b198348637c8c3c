#!/usr/bin/env python3
"""The HIP block decoders (32- and 64-bit sample paths, cleanup + refinement launches) against the oracle on the DAMAGED blocks of
tools/fuzz_blocks_cpu.py (there the oracle is pinned to the live reference on the very same blocks): same refused / decoded
verdict per block, same de-quantised samples.  Needs a GPU.     python tools/fuzz_blocks_gpu.py [seconds] [first seed]
Written when round 4's GPU minutes were spent: NOT yet run on a GPU (DESIGN.md section 8 item 9 lists what it is expected to find;
FUZZ_BLOCKS_SELFTEST=1 runs its bookkeeping with the oracle in the device's place)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from openjph_amd import codec
SELFTEST = bool(os.environ.get("FUZZ_BLOCKS_SELFTEST"))     # no GPU: the oracle stands in for the device (checks this tool's own bookkeeping)
from oracle import oraclebind as ob
from fuzz_blocks_cpu import damaged_blocks


def run_batch(trials, wide):
    """trials: (seed, trial, w, h, kmax, mmsb, data, len2, passes, causal) of one sample width; returns the list of differing ones"""
    descs = np.zeros(len(trials), codec.cb_desc_dtype)
    off = doff = 0
    expect = []
    for i, (seed, trial, w, h, kmax, mmsb, t, len2, npass, causal) in enumerate(trials):
        pitch = (w + 63) & ~63
        d = descs[i]
        d["coef_off"], d["pitch"], d["w"], d["h"] = (2 * off if wide else off), pitch, w, h
        d["K_max"], d["reversible"], d["missing_msbs"] = kmax, 1 | (4 if wide else 0) | (2 if causal else 0), mmsb
        d["num_passes"], d["len1"], d["len2"], d["data_off"] = npass, len(t) - len2, len2, doff
        if wide:
            ok, dec = ob.ht_decode64(t, w, h, w + 8 & ~7, mmsb, len2=len2, num_passes=npass, stripe_causal=causal)
            dq = np.zeros((h, w), np.int64)
            if ok:
                dec = np.ascontiguousarray(dec[:, :w])
                ob.lib().ojo_dequant_rev64(dec.ctypes.data, dq.ctypes.data, dec.size, kmax)
        else:
            ok, dec = ob.ht_decode(t, w, h, w + 8 & ~7, mmsb, len2=len2, num_passes=npass, stripe_causal=causal)
            dq = ob.dequant_rev(np.ascontiguousarray(dec[:, :w]), kmax) if ok else np.zeros((h, w), np.int32)
        expect.append((ok, dq))
        off += pitch * h; doff += len(t)
    if wide:
        coef = torch.full((off + 64,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64)
    else:
        coef = torch.full((off + 64,), 0x5A5A5A5A, dtype=torch.int32)
    if SELFTEST:
        got = coef.numpy(); status = np.zeros(len(trials), np.uint8)
        for i, (ok, want) in enumerate(expect):
            d = descs[i]; w, h = trials[i][2], trials[i][3]; es = 8 if wide else 4
            status[i] = 0 if ok else 1
            np.lib.stride_tricks.as_strided(got[int(d["coef_off"]) // (2 if wide else 1):], (h, w), (int(d["pitch"]) * es, es))[:] = want
    else:
        dcoef = coef.cuda()
        status = codec.ht_decode(descs, np.frombuffer(b"".join(t[6] for t in trials), np.uint8), dcoef)
        got = dcoef.cpu().numpy()
    bad = []
    for i, (ok, want) in enumerate(expect):
        d = descs[i]; w, h = trials[i][2], trials[i][3]
        es = 8 if wide else 4
        g = np.lib.stride_tricks.as_strided(got[int(d["coef_off"]) // (2 if wide else 1):], (h, w), (int(d["pitch"]) * es, es))
        if (status[i] == 0) != ok or not np.array_equal(g, want):
            bad.append((trials[i], int(status[i]), ok, int((g != want).sum())))
    return bad


def main():
    t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
    n = nbad = 0
    while time.time() < t_end:
        batch = {False: [], True: []}
        while len(batch[False]) + len(batch[True]) < 4000:
            for trial, (wide, w, h, kmax, mmsb, data, len2, passes, causal) in enumerate(damaged_blocks(seed)):
                if passes > 3 or len(data) - len2 < 1:
                    continue                          # (more than three passes: the host refuses the packet before any launch)
                batch[wide].append((seed, trial, w, h, kmax, mmsb, data, len2, passes, causal))
            seed += 1
        for wide in (False, True):
            if not batch[wide]:
                continue
            bad = run_batch(batch[wide], wide)
            n += len(batch[wide]); nbad += len(bad)
            for (tr, st, ok, cnt) in bad[:6]:
                print("DIFFERS: seed %d trial %d wide=%s %dx%d kmax %d mmsb %d passes %d len2 %d causal %s: GPU status %d, oracle ok=%s, %d samples differ" %
                      (tr[0], tr[1], wide, tr[2], tr[3], tr[4], tr[5], tr[8], tr[7], tr[9], st, ok, cnt), flush=True)
    print("%d damaged blocks: %d decoded differently on the GPU" % (n, nbad))
    return nbad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
