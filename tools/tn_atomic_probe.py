"""GPU probe: how much of the weight-gradient (TN) GEMM time is the fp32 atomic flush?  Run twice:
   python tools/tn_atomic_probe.py ; LAV_GEMM_DBG_NOATOMIC=1 python tools/tn_atomic_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K

def bench(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [(3072, 768, 36096), (2304, 768, 36096), (768, 768, 36096), (768, 3072, 36096), (2048, 512, 31360), (1536, 512, 31360),
          (512, 512, 31360), (512, 2048, 31360), (1024, 256, 125440), (768, 256, 125440), (256, 256, 125440), (512, 128, 501760),
          (384, 128, 501760), (128, 128, 501760), (30528, 768, 4096)]
tot = 0.0
for (M, N, Kd) in shapes:
    A = torch.randn(Kd, M, device="cuda").bfloat16()
    B = torch.randn(Kd, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda")
    s = K.splits_for(M, N, Kd)
    t = bench(lambda: K.gemm(2, A, B, M, N, Kd, out=out, accumulate=True, splits=s))
    tot += t
    print(f"M={M:6d} N={N:5d} K={Kd:7d} splits={s:3d}: {t*1e3:7.0f} us  {2*M*N*Kd/t/1e9:5.0f} TF   atomics {M*N*s/1e6:.1f} M")
print("total", tot)
