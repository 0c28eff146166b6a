#!/usr/bin/env python
"""Matrix-pipe rate under the instruction mixes of the stem kernel's step 2
(csrc/tools/ctg_probe_loop.hip), with shader clock and socket power sampled while each
variant runs (rocm-smi)."""
import ctypes as C
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "cotengra_amd", "lib", "exp", "libctg_probe_loop.so"))
lib.ctg_probe_loop.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ctg_probe_loop.restype = C.c_double
out = torch.zeros(4096, device="cuda")
NAMES = {
    0: "registers only, 2 chains (cx, cy)",
    1: "registers only, 2 chains + xor per pair",
    2: "A' from LDS, B' in registers (K2Q > 0 path)",
    3: "A', B' from LDS + xor (K2Q = 0 path)",
    4: "A', B' from LDS + xor, two items interleaved",
    5: "registers only, ONE chain (back-to-back dependent)",
    6: "registers only, 4 chains",
    7: "variant 2 + accumulator read-out per item",
    8: "A', B' from LDS, no xor",
    9: "variant 2, two items interleaved",
}


def smi():
    try:
        txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True,
                             timeout=20).stdout
        rows = [r for r in txt.splitlines() if r.strip()]
        hdr, val = rows[0].split(","), rows[1].split(",")
        d = dict(zip(hdr, val))
        pw = next((v for k, v in d.items() if "ower" in k), "?")
        ck = next((v for k, v in d.items() if "sclk" in k.lower()), "?")
        return f"sclk {ck} power {pw}"
    except Exception as e:  # noqa: BLE001
        return f"(rocm-smi: {e})"


for blocks in ():
    for v in sorted(NAMES):
        lib.ctg_probe_loop(v, blocks, 50, out.data_ptr(), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ctg_probe_loop(v, blocks, 2000, out.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = flops / ms / 1e9
        # long run for the power sample (about 1.5 s)
        items = int(2000 * 1500 / ms)
        lib.ctg_probe_loop(v, blocks, items, out.data_ptr(), None)
        s = smi()
        torch.cuda.synchronize()
        print(f"blocks {blocks} ({blocks * 8 // 1024} waves/SIMD)  v{v} {NAMES[v]:52s} {tf:7.1f} TFLOP/s = {tf / 157.3:.3f}   {s}",
              flush=True)

# ---- step 2 as it is vs the row-interleaved form, with the item's stores ----
lib.ctg_probe_step2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ctg_probe_step2.restype = C.c_double
big = torch.zeros(512 * 8 * 2048, device="cuda")
N2 = {
    0: "X/Y form, K2 = 64, B' from LDS + xor (today's K2Q = 0 path)",
    1: "row-interleaved, K2 = 64, B' in registers (64 floats)",
    2: "X/Y form, K2 = 32, B' in registers, sign folded (today's K2Q > 0 path)",
    3: "row-interleaved, K2 = 32, B' in registers (32 floats)",
    4: "X/Y form, K2 = 32, B' from LDS + xor",
    5: "X/Y form, K2 = 64, B' in registers (128 floats)",
    6: "row-interleaved, K2 = 64, planes interleaved per row, LD = K2 + 4",
    7: "row-interleaved, K2 = 64, planes interleaved per row, LD = K2 + 8",
    8: "row-interleaved, K2 = 64, planes interleaved per row, pitch 3 LD + 4",
    9: "row-interleaved, K2 = 64, three planes WITHOUT the bank shift",
    10: "row-interleaved, K2 = 32, planes interleaved per row",
    11: "row-interleaved, K2 = 32, planes interleaved per row, pitch 3 LD + 4",
}
for blocks in (256,):
    for v in sorted(N2):
        lib.ctg_probe_step2(v, blocks, 50, big.data_ptr(), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        flops = lib.ctg_probe_step2(v, blocks, 4000, big.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = flops / ms / 1e9
        items = int(4000 * 1500 / ms)
        lib.ctg_probe_step2(v, blocks, items, big.data_ptr(), None)
        s = smi()
        torch.cuda.synchronize()
        print(f"step2 blocks {blocks}  v{v} {N2[v]:72s} {tf:7.1f} TFLOP/s = {tf / 157.3:.3f}   {s}", flush=True)
