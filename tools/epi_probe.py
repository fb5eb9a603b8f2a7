"""GPU probe: cost of the fused GEMM epilogues on the BERT FFN shapes (VTM pass)."""
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
R, H, F = 36096, 768, 3072
x = torch.randn(R, H, device="cuda").to(bf); W1 = torch.randn(F, H, device="cuda").to(bf); b1 = torch.randn(F, device="cuda")
h = torch.randn(R, F, device="cuda").to(bf); W2 = torch.randn(H, F, device="cuda").to(bf); b2 = torch.randn(H, device="cuda")
pre = torch.empty(R, F, device="cuda", dtype=bf); out1 = torch.empty(R, F, device="cuda", dtype=bf); out2 = torch.empty(R, H, device="cuda", dtype=bf)
cs = torch.zeros(F, device="cuda")
print("ffn1 fwd plain            ", round(t(lambda: K.gemm(0, x, W1, R, F, H, out=out1))))
print("ffn1 fwd +bias            ", round(t(lambda: K.gemm(0, x, W1, R, F, H, out=out1, bias=b1))))
print("ffn1 fwd +bias+gelu       ", round(t(lambda: K.gemm(0, x, W1, R, F, H, out=out1, bias=b1, act=1))))
print("ffn1 fwd +bias+gelu+preact", round(t(lambda: K.gemm(0, x, W1, R, F, H, out=out1, bias=b1, act=1, preact=pre))))
print("ffn2 fwd plain            ", round(t(lambda: K.gemm(0, h, W2, R, H, F, out=out2))))
print("ffn2 fwd +bias+res        ", round(t(lambda: K.gemm(0, h, W2, R, H, F, out=out2, bias=b2, residual=x))))
print("ffn2 fwd +bias+drop+res   ", round(t(lambda: K.gemm(0, h, W2, R, H, F, out=out2, bias=b2, residual=x, dropout_p=0.1, seed=5))))
print("ffn2 dX (->dh) plain      ", round(t(lambda: K.gemm(1, out2, W2, R, F, H, out=out1))))
print("ffn2 dX +gelu_in          ", round(t(lambda: K.gemm(1, out2, W2, R, F, H, out=out1, gelu_in=pre))))
print("ffn2 dX +gelu_in+colsum   ", round(t(lambda: K.gemm(1, out2, W2, R, F, H, out=out1, gelu_in=pre, colsum=cs))))
print("ffn1 dX plain             ", round(t(lambda: K.gemm(1, h, W1, R, H, F, out=out2))))
print("ffn1 dX +residual         ", round(t(lambda: K.gemm(1, h, W1, R, H, F, out=out2, residual=x))))
