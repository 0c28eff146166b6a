#!/usr/bin/env python
"""Turn a rocprofv3 result (rocpd sqlite .db) into the small text/JSON
summaries committed under profiles/.

  python tools/rocprof_summary.py <results.db> <out_prefix>

writes <out_prefix>_kernels.txt : per-kernel calls / total / average / share
       <out_prefix>_kernels.json: the same as JSON (name, calls, total_ns, avg_ns, pct)
With --pmc it also aggregates PMC counter values per kernel.
"""
import json
import sqlite3
import sys


def main():
    db_path, prefix = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    out = []
    lines = ["%-110s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for name, calls, tot, avg, mn, mx in rows:
        out.append({"name": name, "calls": calls, "total_ns": tot, "avg_ns": avg,
                    "min_ns": mn, "max_ns": mx, "pct": 100.0 * tot / total})
        lines.append("%-110s %8d %14.3f %12.2f %12.2f %12.2f %6.2f%%" % (
            name[:110], calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    pmc = {}
    if "--pmc" in sys.argv:
        try:
            q = cur.execute(
                "select k.name, p.counter_name, sum(p.value), count(*) from pmc_events p "
                "join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name"
            ).fetchall()
            for name, cname, val, n in q:
                pmc.setdefault(name, {})[cname] = {"sum": val, "dispatches": n}
            lines.append("")
            lines.append("PMC counters (sum over dispatches):")
            for name, d in pmc.items():
                lines.append("  " + name[:120])
                for cname, v in d.items():
                    lines.append("      %-28s sum=%.6g over %d dispatches" % (cname, v["sum"], v["dispatches"]))
        except Exception as e:  # schema differences
            lines.append("PMC query failed: %r" % (e,))
    open(prefix + "_kernels.txt", "w").write("\n".join(lines) + "\n")
    json.dump({"kernels": out, "pmc": pmc}, open(prefix + "_kernels.json", "w"), indent=1)
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main()
