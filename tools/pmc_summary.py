#!/usr/bin/env python3
"""Per-kernel average FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc csv passes.
Usage: pmc_summary.py <dir_fetch> <dir_write>.  Values are the raw counter units (KiB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            did = row.get("Dispatch_Id") or row.get("Dispatch Id")
            acc[(name, did)][0] += val       # a counter may be split over several rows (per XCC)
            acc[(name, did)][1] = 1
    per = defaultdict(list)
    for (name, _), (v, _) in acc.items():
        per[name].append(v)
    return per


def main():
    fe, wr = load(sys.argv[1]), load(sys.argv[2])
    names = sorted(set(fe) | set(wr), key=lambda n: -sum(fe.get(n, [0])))
    print("| kernel | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB |")
    print("|---|---|---|---|")
    for n in names:
        f, w = fe.get(n, []), wr.get(n, [])
        short = n if len(n) < 100 else n[:97] + "..."
        print("| `%s` | %d | %.1f | %.1f |" % (short, max(len(f), len(w)), sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)))


if __name__ == "__main__":
    main()
