"""Shared timing helper of the GPU probes: mean time of one call of f in MICROSECONDS (HIP events on the current stream)."""
import torch


def bench(f, n=10, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
