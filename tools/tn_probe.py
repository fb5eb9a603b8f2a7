"""GPU probe: weight-gradient (TN) GEMM time vs split-K factor on the shapes of the pretrain step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K

def bench(f, n=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [(3072, 768, 36096), (3072, 768, 9024), (2304, 768, 36096), (768, 768, 36096), (768, 768, 9024), (2048, 512, 31360), (1536, 512, 31360),
          (512, 512, 31360), (1024, 256, 125440), (768, 256, 125440), (256, 256, 125440), (512, 128, 501760),
          (384, 128, 501760), (128, 128, 501760), (30528, 768, 4096), (30528, 768, 1024), (4096, 1024, 7840)]
for (M, N, Kd) in shapes:
    A = torch.randn(Kd, M, device="cuda").bfloat16()
    B = torch.randn(Kd, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda")
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    res = []
    for s in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256):
        if tiles * s > 4096 or Kd // s < 512: continue
        t = bench(lambda: K.gemm(2, A, B, M, N, Kd, out=out, accumulate=True, splits=s))
        res.append((t, s))
    best = min(res)
    cur = K.splits_for(M, N, Kd)
    tc = [t for t, s in res if s == cur]
    print(f"M={M:6d} N={N:5d} K={Kd:7d} tiles={tiles:4d}: best splits={best[1]:3d} ({best[0]*1e3:6.0f} us, {2*M*N*Kd/best[0]/1e9:5.0f} TF) blocks={tiles*best[1]:5d} | "
          + " ".join(f"{s}:{t*1e3:.0f}" for t, s in res) + f" | current {cur}")
