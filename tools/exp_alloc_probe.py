#!/usr/bin/env python
"""Two 100 GiB allocations side by side (the two physical regions a 112 GiB arena alternates between,
profiles/r6_process_alternation.txt): is there a simple access pattern that runs at different speeds on them?"""
import torch

def t_ms(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best

n = 100 * (1 << 30) // 8
X = torch.empty(n, dtype=torch.complex64, device="cuda"); X.zero_()
Y = torch.empty(n, dtype=torch.complex64, device="cuda"); Y.zero_()
print("X", hex(X.data_ptr()), "Y", hex(Y.data_ptr()))
m = 1 << 31   # 16 GiB pieces
def pats(T, off):
    a = T[off: off + m]
    b = T[off + 2 * m: off + 3 * m]
    out = {}
    out["copy"] = t_ms(lambda: b.copy_(a))
    for k in (5, 8, 11, 14):   # transposes: (2^k, m / 2^k) -> strided gather, contiguous store
        v = a.view(1 << k, -1)
        w = b.view(-1, 1 << k)
        out["T%d" % k] = t_ms(lambda: w.copy_(v.t()))
    v = a.view(-1, 32, 2, 16)   # permute of middle indices, like a kept index moved past a contracted one
    w = b.view(-1, 2, 32, 16)
    out["perm"] = t_ms(lambda: w.copy_(v.permute(0, 2, 1, 3)))
    return out
for name, T in (("X", X), ("Y", Y), ("X", X), ("Y", Y)):
    for off in (0, 3 * m):
        r = pats(T, off)
        print(name, "off", off // m, " ".join("%s %.2f" % kv for kv in r.items()), flush=True)
