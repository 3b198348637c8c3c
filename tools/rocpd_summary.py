#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (kernel trace) into a table with one row per LAUNCH KIND: kernel name x grid
size x stream -- the same kernel launched over the top resolution's blocks and over the lower ones, or on the main and on
a side stream, are different rows (one per-kernel average over all of them matches no figure of the bench line).
Durations: median / min / mean / max in microseconds; the median is what to compare with bench.py's HIP-event figures.
Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import statistics
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, stream_id, end - start from kernels").fetchall()
    groups = {}
    for name, gx, gy, gz, wx, wy, wz, st, dur in rows:
        wgs = (int(gx) // max(int(wx), 1)) * (int(gy) // max(int(wy), 1)) * (int(gz) // max(int(wz), 1))
        groups.setdefault((short(name), wgs, int(wx) * int(wy) * int(wz), int(st)), []).append(dur / 1e3)
    total = sum(sum(v) for v in groups.values()) or 1.0
    lines = ["| kernel | workgroups x threads | stream | calls | median us | min us | mean us | max us | % of kernel time |",
             "|---|---|---|---|---|---|---|---|---|"]
    for (name, wgs, wx, st), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| `%s` | %d x %d | %d | %d | %.1f | %.1f | %.1f | %.1f | %.1f |" % (
            name, wgs, wx, st, len(v), statistics.median(v), min(v), sum(v) / len(v), max(v), 100.0 * sum(v) / total))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
