"""Achievable HBM read rate at a given size (torch reduction / copy kernels over rotating buffers larger than the
Infinity Cache), for comparison with K1's rate on the same bytes.  python tools/hbm_probe.py [MiB ...]"""
import sys
import torch

def probe(mib, iters=40):
    n = mib * (1 << 20) // 4
    nbuf = max(2, int(600 / mib) + 1)
    bufs = [torch.randint(0, 100, (n,), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    out = torch.empty_like(bufs[0])
    res = {}
    for name, fn in (("sum", lambda b: b.sum()), ("copy", lambda b: out.copy_(b))):
        for b in bufs: fn(b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fn(bufs[i % nbuf])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        res[name] = (us, mib * (1 << 20) / us / 1e6)
    return res

if __name__ == "__main__":
    for mib in [int(a) for a in sys.argv[1:]] or [117, 1024, 6231]:
        r = probe(mib)
        print(mib, "MiB:", {k: "%.1f us, %.2f TB/s read" % v for k, v in r.items()}, flush=True)
