#!/usr/bin/env python
"""Per-dispatch timeline of the LAST repetition in a rocprofv3 kernel trace
(rocpd sqlite .db): start offset, duration, gap to the previous kernel, grid.

  python tools/trace_timeline.py <results.db> <launches_per_repetition | 0: up to the last accumulate kernel> [out.txt]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    per = int(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = "select name, start, end%s from kernels order by start" % (", " + gx if gx else "")
    rows = db.execute(q).fetchall()
    if per <= 0:   # one repetition = everything after the previous accumulate kernel
        ends = [i for i, r in enumerate(rows) if "accum_kernel" in r[0]]
        rows = rows[ends[-2] + 1: ends[-1] + 1]
    else:
        rows = rows[-per:]
    t0 = rows[0][1]
    lines = ["%9s %9s %8s %9s  %s" % ("start_us", "dur_us", "gap_us", "grid_x", "kernel")]
    prev_end = None
    busy = 0
    for r in rows:
        name, st, en = r[0], r[1], r[2]
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        busy += en - st
        lines.append("%9.2f %9.2f %8.2f %9s  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap,
                                                    r[3] if gx else "-", name[:90]))
        prev_end = en
    lines.append("span %.1f us, busy %.1f us, %d launches" % ((rows[-1][2] - t0) / 1e3, busy / 1e3, len(rows)))
    text = "\n".join(lines)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
