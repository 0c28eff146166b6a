#!/usr/bin/env python
"""Drive cotengra_amd/csrc/tools/ctg_probe.hip: HBM rate of a streaming-kernel-shaped
copy for each store width / pattern, grid size and operand size."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "cotengra_amd", "lib", "exp", "libctg_probe.so"))
lib.ctg_probe_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int]
names = {0: "16B load, 16B store", 1: "16B load, 8B store", 2: "16B load, 8B store (epilogue rows)",
         3: "16B load only", 4: "16B store only", 5: "8B store only"}
for gib in (8,):
    n = gib << 30
    src = torch.empty(n // 4, device="cuda", dtype=torch.float32).normal_()
    dst = torch.empty_like(src)
    nt = n // 4096
    for blocks, tpw in ((256 * 3, 0), (0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (0, 64)):
        if tpw:
            blocks = (nt + 4 * tpw - 1) // (4 * tpw)
        for mode in (0, 3, 4):
            def run():
                lib.ctg_probe_copy(src.data_ptr(), dst.data_ptr(), n, mode, blocks, None, tpw)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            moved = n * (2 if mode < 3 else 1)
            print(f"{gib:2d} GiB  blocks {blocks:7d} tasks/wave {tpw:2d}  {names[mode]:36s} {ms:8.3f} ms  {moved / ms / 1e9:6.2f} TB/s", flush=True)
    # the elementwise reference on the same buffers
    torch.mul(src, 2.0, out=dst); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        torch.mul(src, 2.0, out=dst)
    e1.record(); torch.cuda.synchronize()
    print(f"{gib:2d} GiB  torch.mul 1R:1W {2 * n / (e0.elapsed_time(e1) / 3) / 1e9:6.2f} TB/s")
    del src, dst
