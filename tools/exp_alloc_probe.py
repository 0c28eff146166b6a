import torch, time
def probe(x):
    best=1e9
    for _ in range(5):
        torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record(); y = x[: x.numel()//2].copy_(x[x.numel()//2:]); e.record(); torch.cuda.synchronize()
        best=min(best, s.elapsed_time(e))
    return best
n = 112 * (1<<30) // 4
for rep in range(6):
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    x.zero_()
    t = probe(x)
    print("alloc", rep, "ptr", hex(x.data_ptr()), "copy 56 GB ms", round(t,3), "TB/s", round(2*56*(1<<30)/t/1e9,3), flush=True)
    del x
    torch.cuda.empty_cache()
