#!/usr/bin/env python
"""Matrix-core issue rates on this box (csrc/tools/ctg_probe.hip): chains of
independent MFMAs, no memory traffic -- the achievable peak per instruction."""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "cotengra_amd", "lib", "exp", "libctg_probe.so"))
lib.ctg_probe_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ctg_probe_mfma.restype = C.c_double
out = torch.zeros(256, device="cuda")
names = {0: "v_mfma_f64_16x16x4_f64", 1: "v_mfma_f32_16x16x4_f32", 2: "v_mfma_f32_32x32x2_f32"}
for which, chains in ((0, 4), (0, 8), (1, 4), (1, 8), (2, 4)):
    for blocks in (256, 512, 1024):
        iters = 20000
        lib.ctg_probe_mfma(which, chains, blocks, 100, out.data_ptr(), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ctg_probe_mfma(which, chains, blocks, iters, out.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"{names[which]:26s} chains {chains} blocks {blocks:5d} ({blocks * 4 // 256 // 4} waves/SIMD)  {flops / ms / 1e9:7.1f} TFLOP/s")

lib.ctg_probe_valu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ctg_probe_valu.restype = C.c_double
buf = torch.zeros(512, device="cuda", dtype=torch.float64)
for which, nm in ((0, "v_fma_f64"), (1, "v_fma_f32")):
    for blocks in (1024, 2048, 4096):
        lib.ctg_probe_valu(which, blocks, 100, buf.data_ptr(), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ctg_probe_valu(which, blocks, 20000, buf.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        print(f"{nm:26s} 16 chains blocks {blocks:5d} ({blocks * 4 // 1024} waves/SIMD)  {flops / e0.elapsed_time(e1) / 1e9:7.1f} TFLOP/s")
