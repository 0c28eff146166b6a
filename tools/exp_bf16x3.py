#!/usr/bin/env python
"""Numerics of emulating fp32 products on the bf16 matrix cores: every fp32
operand is split exactly into three bfloat16 values (8 + 8 + 8 mantissa bits), products of bf16
values are exact in fp32, and a real MAC becomes 6 bf16 MACs (the three smallest cross terms are
dropped) accumulated in fp32 -- at 16x the fp32 MFMA rate, i.e. 2.7x the fp32 matrix peak.  Prints,
for GEMMs of several depths, the error of a plain fp32 GEMM against the TRUNCATION error of the
6-term (and 3-term) emulation, both relative to sqrt(K) (numpy, CPU)."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(x):
    a1 = bf16(x)
    r = (x - a1).astype(np.float32)
    a2 = bf16(r)
    return a1, a2, bf16((r - a2).astype(np.float32))


for K in (16, 64, 256, 2048):
    A = rng.standard_normal((2048, K)).astype(np.float32)
    B = rng.standard_normal((K, 64)).astype(np.float32)
    truth = A.astype(np.float64) @ B.astype(np.float64)
    a, b = split3(A), split3(B)
    assert np.max(np.abs((a[0].astype(np.float64) + a[1] + a[2]) - A)) == 0.0   # the split is exact

    def emu(terms):
        return sum(a[i].astype(np.float64) @ b[j].astype(np.float64) for i, j in terms)

    s = np.sqrt(K)
    e32 = np.max(np.abs((A @ B).astype(np.float64) - truth)) / s
    e6 = np.max(np.abs(emu([(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]) - truth)) / s
    e3 = np.max(np.abs(emu([(0, 0), (0, 1), (1, 0)]) - truth)) / s
    print(f"K = {K:5d}: fp32 GEMM {e32:.2e} | 6-term truncation {e6:.2e} | 3-term {e3:.2e}")
