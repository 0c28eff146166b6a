#!/usr/bin/env python
"""Build experiment variants of the library: tools/build_variants.py name=-DFLAG[,-DFLAG2] ...
-> cotengra_amd/lib/exp/libctg_<name>.so (select with CTG_LIB=...)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

out = os.path.join(g.ROOT, "cotengra_amd", "lib", "exp")
os.makedirs(out, exist_ok=True)
jobs = []
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    jobs.append((name, [f for f in flags.split(",") if f]))
g.build()  # the default objects first: variants share every source the flags do not touch
KERNEL_SOURCES = os.environ.get("CTG_VARIANT_SOURCES", "ctg_pair_mfma.hip,ctg_pair_mfma_f64.hip,ctg_stem.hip,ctg_runtime.hip").split(",")
with ThreadPoolExecutor(8) as pool:
    list(pool.map(lambda j: g.build(extra_flags=j[1], lib=os.path.join(out, f"libctg_{j[0]}.so"),
                                    flag_sources=KERNEL_SOURCES), jobs))
print("built", [j[0] for j in jobs])
