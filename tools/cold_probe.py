"""GPU probe: GEMM time with COLD operands (rotating through > 256 MB of distinct buffers to defeat the Infinity
Cache) vs warm (same buffers every call) -- tells whether a shape is compute- or HBM/fabric-bound in the real step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K

def run(layout, M, N, Kd, splits, nset, iters=12):
    sets = []
    for i in range(nset):
        if layout == 0: A, B = torch.randn(M, Kd, device="cuda").bfloat16(), torch.randn(N, Kd, device="cuda").bfloat16()
        elif layout == 1: A, B = torch.randn(M, Kd, device="cuda").bfloat16(), torch.randn(Kd, N, device="cuda").bfloat16()
        else: A, B = torch.randn(Kd, M, device="cuda").bfloat16(), torch.randn(Kd, N, device="cuda").bfloat16()
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if layout == 2 else torch.bfloat16)
        sets.append((A, B, out))
    def call(i):
        A, B, out = sets[i % nset]
        K.gemm(layout, A, B, M, N, Kd, out=out, accumulate=(layout == 2), splits=splits)
    for i in range(nset): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): call(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, layout, M, N, Kd, sp in [("bert ffn1 dW", 2, 3072, 768, 36096, 3), ("swin s3 fc1 dW", 2, 2048, 512, 31360, 3),
                                   ("bert ffn1 fwd", 0, 36096, 3072, 768, 1), ("bert ffn2 fwd", 0, 36096, 768, 3072, 1),
                                   ("bert ffn1 dX", 1, 36096, 768, 3072, 1), ("swin s3 fc1 fwd", 0, 31360, 2048, 512, 1),
                                   ("swin s3 fc1 dX", 1, 31360, 512, 2048, 1)]:
    w = run(layout, M, N, Kd, sp, 1)
    c = run(layout, M, N, Kd, sp, 6)
    fl = 2.0 * M * N * Kd
    print(f"{name:18s} layout {layout} M={M} N={N} K={Kd}: warm {w*1e3:6.0f} us ({fl/w/1e9:5.0f} TF)   cold {c*1e3:6.0f} us ({fl/c/1e9:5.0f} TF)")
