"""GPU probe: k-loop rate of the 256x256 NT kernel when operands are cache-resident vs streamed from HBM."""
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


for (M, N) in ((4096, 4096), (2048, 8192), (65536, 256), (65536, 1024), (16384, 1024)):
    res = []
    for Kd in (1024, 4096, 8192):
        A = torch.randn(M, Kd, device="cuda").to(bf); Bm = torch.randn(N, Kd, device="cuda").to(bf)
        out = torch.empty(M, N, device="cuda", dtype=bf)
        res.append((Kd, t(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out))))
    (k0, t0), (k1, t1), (k2, t2) = res
    tiles = (M // 256) * (N // 256)
    print(f"M={M} N={N} tiles={tiles}: " + " ".join(f"K={k}:{u:.0f}us({2.0*M*N*k/u/1e6:.0f}TF)" for k, u in res) +
          f" | slope {2.0*M*N*(k2-k0)/(t2-t0)/1e6:.0f} TF = {2.0*M*N*(k2-k0)/(t2-t0)/1e6/min(tiles,256)*256/2500*100:.0f}% of peak per busy CU")
