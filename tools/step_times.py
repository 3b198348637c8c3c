#!/usr/bin/env python3
"""Per-step GPU time (HIP events between steps) of bursts of 20 encode + decode steps of the 8K bench frame, after idle gaps of
different lengths -- how much of a short timed region (the driver's --steps 20) is the chip coming out of idle."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from bench import WORKLOADS, workload_image
from openjph_amd import codec
from openjph_amd.plan import Plan, make_params
name = "c3_8k_444_12b_irv97"
w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
img = workload_image(name)
d = torch.from_numpy(img.astype(np.int16)).cuda()
plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep))
enc = codec.Encoder(plan=plan); cs = enc.encode(d); dec = codec.Decoder(cs); out = torch.empty_like(d)
enc.set_timing(False); dec.set_timing(False)


def burst(n, idle_s, busy_before=0):
    for _ in range(busy_before):
        enc.run_device(d); dec.run_device(out)
    if idle_s is not None:
        torch.cuda.synchronize(); time.sleep(idle_s)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        enc.run_device(d); dec.run_device(out); ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


burst(300, 0.0)
for idle, busy in ((None, 300), (0.0, 300), (0.0005, 300), (0.005, 300), (0.05, 300), (0.5, 300), (0.0, 5)):
    t = burst(20, idle, busy)
    print("idle %s s after %3d busy steps: mean %.4f  first 10: %s  last 5 mean %.4f" %
          ("none (queued behind them)" if idle is None else "%.4f" % idle, busy, sum(t) / len(t), " ".join("%.3f" % x for x in t[:10]), sum(t[-5:]) / 5))


def burst2(what):
    for _ in range(60):
        enc.run_device(d); dec.run_device(out)
    torch.cuda.synchronize()
    if "zero" in what: out.zero_()
    if "epoch" in what: dec.giveup_epoch()
    if "sync2" in what: torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        enc.run_device(d); dec.run_device(out); ev[i + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
    print("before the burst: %-18s mean %.4f  first 10: %s" % (what or "(sync only)", sum(t) / 20, " ".join("%.3f" % x for x in t[:10])))


if len(sys.argv) > 1:
    for what in ("", "zero", "epoch", "zero epoch sync2", "", "zero"):
        burst2(what)
