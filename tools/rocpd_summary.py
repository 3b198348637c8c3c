#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (kernel trace) into a per-kernel table:
calls, total / average / min / max duration.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, n, tot / 1e6, avg / 1e3, mn / 1e3,
                                                                          mx / 1e3, 100.0 * tot / total))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
