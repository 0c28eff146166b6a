#!/usr/bin/env python
"""Rates on this box for emulating fp32 products on the bf16 matrix cores (csrc/tools/ctg_probe.hip:
bf16x3_kernel): the issue rate of v_mfma_f32_32x32x16_bf16, and one "task" of the stem kernel's
first step done that way -- 16 fp32 values per lane split into 3 bf16 each + 24 MFMAs -- as
fp32-equivalent TFLOP/s (8 waves per CU, as the stem kernel runs)."""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "cotengra_amd", "lib", "exp", "libctg_probe.so"))
lib.ctg_probe_bf16x3.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
lib.ctg_probe_bf16x3.restype = C.c_double
out = torch.zeros(1024, device="cuda")
src = torch.randn(1024, device="cuda")
for which, nm in ((0, "v_mfma_f32_32x32x16_bf16, 4 chains"), (1, "fp32-equivalent task: split3 + 24 MFMAs")):
    for blocks in (256, 512):
        lib.ctg_probe_bf16x3(which, blocks, 100, out.data_ptr(), src.data_ptr(), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ctg_probe_bf16x3(which, blocks, 20000, out.data_ptr(), src.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        print(f"{nm:42s} blocks {blocks:4d} x 8 waves  {flops / e0.elapsed_time(e1) / 1e9:8.1f} TFLOP/s")
