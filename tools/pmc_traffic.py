#!/usr/bin/env python
"""HBM traffic of the MFMA pair kernels from two rocprofv3 PMC passes.

  python tools/pmc_traffic.py <fetch.db> <write.db> <slices_in_run> <out.json> [<tree file name>]

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) on
gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (64 B tallied per 128 B request), so reads are doubled;
WRITE_SIZE is used as is (it reproduces the algorithmic write bytes of the
plan to 4 digits, see the JSON).  The two counters need separate passes
(TCC slots).
"""
import collections
import json
import re
import sqlite3
import sys


def step_kernel_name(rocprof_name):
    """Normalise a rocprof kernel name to the ctg_exec_step_kernel() spelling."""
    # (the fast kernel carries a third template argument since round 2: GROUPED)
    m = re.match(r"(pair_mfma_(?:fast|c64)_kernel)<ctg::MfmaCfg<(\d+), (\d+), (\d+), \d+, \d+>, (true|false)(?:, (?:true|false))?>", rocprof_name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)}>,{m.group(5)}"
    m = re.match(r"pair_mfma_stream_kernel<(\d+), (true|false), (true|false), (true|false), (\d+)>", rocprof_name)
    if m:
        return f"pair_mfma_stream_kernel<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)},{m.group(5)}>"
    m = re.match(r"pair_mfma_kstream_kernel<(\d+), (true|false)>", rocprof_name)
    if m:
        return f"pair_mfma_kstream_kernel<{m.group(1)},{m.group(2)}>"
    m = re.match(r"pair_rowwise_kernel<(\d+), (?:true|false)>", rocprof_name)
    if m:
        return f"pair_rowwise_kernel<{m.group(1)}>"
    m = re.match(r"(stem2h?_kernel)<([^>]*)>", rocprof_name)
    if m:   # (template arguments spelled as csrc/ctg_stem.hip: stem2_kernel_name spells them; stem2h: fp16 x 2)
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    m = re.match(r"pair_skinny_kernel<(\d+), (\d+)>", rocprof_name)
    if m:
        return f"pair_skinny_kernel<{m.group(1)},{m.group(2)}>"
    return rocprof_name.split("<")[0]


def per_kernel(path, cname):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select dispatch_id, kernel_name, sum(value) from counters_collection "
        "where counter_name=? group by dispatch_id", (cname,)).fetchall()
    d = collections.defaultdict(lambda: [0, 0.0])
    for _, kn, v in rows:
        k = kn.split("(")[0].replace("void ctg::", "")
        d[k][0] += 1
        d[k][1] += v
    return d


def main():
    fdb, wdb, nsl, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    tree = sys.argv[5] if len(sys.argv) > 5 else None
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kernels = {}
    tot_f = tot_w = launches = 0
    for k in f:
        fb = 2.0 * f[k][1] * 1024.0
        wb = w.get(k, [0, 0.0])[1] * 1024.0
        kernels[step_kernel_name(k)] = {"rocprof_name": k, "launches": f[k][0],
                                        "fetch_bytes_corrected": fb, "write_bytes": wb,
                                        "hbm_bytes_per_launch": (fb + wb) / f[k][0]}
        if "pair_mfma" in k or "pair_skinny" in k or "stem2_kernel" in k or "stem2h_kernel" in k:
            tot_f += fb
            tot_w += wb
            launches += f[k][0]
    res = {
        "tree": tree,
        "counters": "FETCH_SIZE (x2 gfx950 correction), WRITE_SIZE; separate rocprofv3 --pmc passes",
        "slices_in_run": nsl,
        "mfma_launches": launches,
        "mfma_fetch_bytes_per_slice": tot_f / nsl,
        "mfma_write_bytes_per_slice": tot_w / nsl,
        "hbm_bytes_per_launch": (tot_f + tot_w) / max(launches, 1),
        "kernels": kernels,
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
