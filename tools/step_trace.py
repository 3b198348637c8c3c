#!/usr/bin/env python3
"""The launches of the last steps of a bench run from a rocprofv3 --kernel-trace CSV, on one time axis: start, end, duration,
queue, and the gap since the previous launch ended on ANY queue (where the chip sat idle between launches).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python bench.py --steps 6 --warmup 2 --plain --no-cpu-baseline
    python tools/step_trace.py gpurun_out/tr [launches to print, default 40]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
# the timed loop's launches are the ones before the per-launch-timing repeats: take a window in the middle of the run
codec = [r for r in rows if "ojphgpu" in r[2] or "anonymous" in r[2] or "fillBuffer" in r[2]]
mid = len(codec) // 3
sel = codec[mid:mid + n]
t0 = sel[0][0]
busy_until = sel[0][0]
for s, e, k, q in sel:
    name = k.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    gap = (s - busy_until) / 1e3
    print("%9.1f .. %9.1f  %7.1f us  q%-3s idle-before %6.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, gap if gap > 0 else 0.0, name))
    busy_until = max(busy_until, e)
