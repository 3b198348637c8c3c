#!/usr/bin/env python3
"""Per-frame timeline of the codec's launches from a rocprofv3 --kernel-trace CSV: start / end of every launch relative to
the first launch of its encode or decode job, and the idle gap before it on its stream (queue).
    python tools/timeline.py <dir with *_kernel_trace.csv> [frames to print]"""
import csv
import glob
import sys

d = sys.argv[1]
nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ojphgpu" in n or "anonymous" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()


def short(n):
    for k in ("dwt_forward", "dwt_inverse", "ht_encode", "ht_dec_prep", "ht_dec_step1", "ht_dec_step2", "ht_dec_refine", "convert", "assemble", "copy_to_host"):
        if k in n:
            return k + ("<" + n.split("<", 1)[1].split(">")[0] + ">" if "<" in n and k.startswith("dwt") else "")
    return n[:40]


# a job starts at the first dwt_forward<..., 16/8/32, ...> (encode) or ht_dec_prep (decode)
jobs, cur = [], None
for s, e, n, q in rows:
    k = short(n)
    starts = k == "ht_dec_prep" or (k.startswith("dwt_forward") and cur is not None and cur["kind"] == "dec") or cur is None
    if k.startswith("dwt_forward") and cur is not None and cur["kind"] == "enc" and cur["seen_ht"]:
        starts = True
    if starts:
        cur = {"kind": "dec" if k == "ht_dec_prep" else "enc", "rows": [], "seen_ht": False}
        jobs.append(cur)
    cur["rows"].append((s, e, k, q))
    if k == "ht_encode":
        cur["seen_ht"] = True
for kind in ("enc", "dec"):
    sel = [j for j in jobs if j["kind"] == kind][-nprint:]
    for j in sel:
        t0 = j["rows"][0][0]
        end = max(r[1] for r in j["rows"])
        print("%s job, %.1f us from first start to last end" % (kind, (end - t0) / 1e3))
        last_end = {}
        for s, e, k, q in j["rows"]:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            print("   %8.1f .. %8.1f  (%6.1f us, queue %s, idle before %5.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, gap, k))
            last_end[q] = e
