"""GPU probe: GEMM time vs K at fixed (M, N) -> per-k-iteration cost and fixed (prologue + epilogue) cost per tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
bf = torch.bfloat16


def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for layout in (0, 1):
    for (M, N) in ((36096, 3072), (32768, 2048), (32768, 512)):
        row = []
        for Kd in (64, 128, 256, 512, 768, 1536, 3072):
            A = torch.randn(M, Kd, device="cuda").to(bf)
            Bm = (torch.randn(N, Kd, device="cuda") if layout == 0 else torch.randn(Kd, N, device="cuda")).to(bf)
            out = torch.empty(M, N, device="cuda", dtype=bf)
            us = t(lambda: K.gemm(layout, A, Bm, M, N, Kd, out=out))
            row.append(f"K={Kd}:{us:.0f}us")
        print("NT NN".split()[layout], M, N, " ".join(row))
