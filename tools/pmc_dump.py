#!/usr/bin/env python
"""Print per-kernel sums of every counter found in rocprofv3 --pmc output
directories (rocpd .db):  python tools/pmc_dump.py <dir> [<dir> ...] [--match substr]"""
import collections
import glob
import os
import sqlite3
import sys

match = None
dirs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    else:
        dirs.append(a)
res = collections.defaultdict(dict)
for d in dirs:
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute(
                "select kernel_name, counter_name, count(distinct dispatch_id), sum(value) "
                "from counters_collection group by kernel_name, counter_name").fetchall()
        except sqlite3.OperationalError as e:
            print(db, e)
            continue
        for kn, cn, nd, v in rows:
            k = kn.split("(")[0].replace("void ctg::", "")
            if match and match not in k:
                continue
            res[k][cn] = (nd, v)
for k, cs in res.items():
    print(k)
    for cn, (nd, v) in sorted(cs.items()):
        print(f"    {cn:40s} launches={nd:4d} sum={v:.6g} per_launch={v / nd:.6g}")
