"""GPU probe: k-loop rate of the weight-gradient (TN) large-tile kernel from the slope of time vs K at fixed splits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K


def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, s) in ((3072, 768, 7), (2048, 2048, 4), (4096, 4096, 1), (512, 2048, 16)):
    res = []
    for Kd in (16384, 32768, 65536):
        A = torch.randn(Kd, M, device="cuda").bfloat16(); B = torch.randn(Kd, N, device="cuda").bfloat16()
        out = torch.zeros(M, N, device="cuda")
        res.append((Kd, t(lambda: K.gemm(2, A, B, M, N, Kd, out=out, accumulate=True, splits=s))))
    (k0, t0), (k1, t1), (k2, t2) = res
    print(f"TN M={M} N={N} splits={s}: " + " ".join(f"K={k}:{u:.0f}us({2.0*M*N*k/u/1e6:.0f}TF)" for k, u in res) + f" | slope {2.0*M*N*(k2-k0)/(t2-t0)/1e6:.0f} TF")
