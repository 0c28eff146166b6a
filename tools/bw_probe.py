#!/usr/bin/env python
"""HBM bandwidth ceilings of this box for the access mixes the streaming
kernels produce (torch elementwise kernels, 4 GiB operands): copy (1R:1W),
fill (0R:1W), sum (1R:0W), add (2R:1W)."""
import torch

n = 1 << 30  # float32 elements = 4 GiB
a = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
b = torch.empty_like(a).normal_()
c = torch.empty_like(a)


def timeit(fn, nbytes, label, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:24s} {ms:8.3f} ms  {nbytes / ms / 1e9:7.2f} TB/s")


timeit(lambda: c.copy_(a), 8 * n, "copy 1R:1W")
timeit(lambda: c.fill_(1.0), 4 * n, "fill 0R:1W")
timeit(lambda: a.sum(), 4 * n, "sum 1R:0W")
timeit(lambda: torch.add(a, b, out=c), 12 * n, "add 2R:1W")
timeit(lambda: torch.mul(a, 2.0, out=c), 8 * n, "scale 1R:1W")
