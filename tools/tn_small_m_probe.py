"""GPU probe: weight-gradient GEMMs of cfg4 (Swin-L stage 0, C = 192: 160 <= M < 256 output rows; fusion encoder with n * L = 30280 token
rows, not a multiple of 32) -- run once with the defaults and once with LAV_GEMM_TN_MINM=256 (old kernel choice for the small-M shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
bf = torch.bfloat16


def run(f, n=8):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, Kd in ((192, 192, 368640), (192, 768, 368640), (576, 192, 368640), (768, 192, 368640), (768, 768, 30280), (2304, 768, 30280), (3072, 768, 30280), (768, 3072, 30280)):
    A, B = torch.randn(Kd, M, device="cuda").to(bf), torch.randn(Kd, N, device="cuda").to(bf)
    C = torch.zeros(M, N, device="cuda")
    sp = K.splits_for(M, N, Kd)
    t = run(lambda: K.gemm(2, A, B, M, N, Kd, out=C, accumulate=True, splits=sp))
    print(f"TN {M:5d} x {N:5d} x {Kd:7d} splits {sp:3d}: {t:7.1f} us  {2.0 * M * N * Kd / t / 1e6:6.0f} TF/s  (LAV_GEMM_TN_MINM={os.environ.get('LAV_GEMM_TN_MINM', '160')})")
