"""GPU probe: the forward GEMMs of one fusion layer at the cfg2 shape (45120 rows) with the epilogues the step uses, against the same
products with a plain bf16 store -- what each fused epilogue costs in isolation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from tools._bench import bench
bf, f32 = torch.bfloat16, torch.float32
R, H, F = 45120, 768, 3072
dev = "cuda"
x = torch.randn(R, H, device=dev).to(bf); h = torch.randn(R, F, device=dev).to(bf)
Wq = (0.02 * torch.randn(3 * H, H, device=dev)).to(bf); Wa = (0.02 * torch.randn(H, H, device=dev)).to(bf)
W1 = (0.02 * torch.randn(F, H, device=dev)).to(bf); W2 = (0.02 * torch.randn(H, F, device=dev)).to(bf)
bq, ba, b1, b2 = (torch.randn(n, device=dev) for n in (3 * H, H, F, H))
pre32 = torch.randn(R, H, device=dev); mean = torch.zeros(R, device=dev); rstd = torch.ones(R, device=dev)
g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
o_q = torch.empty(R, 3 * H, device=dev, dtype=bf); o_h = torch.empty(R, F, device=dev, dtype=bf); o_b = torch.empty(R, H, device=dev, dtype=bf)
o_32 = torch.empty(R, H, device=dev, dtype=f32); hp = torch.empty(R, F, device=dev, dtype=bf)
rows = [
    ("qkv   plain", lambda: K.gemm(0, x, Wq, R, 3 * H, H, out=o_q)),
    ("qkv   +bias", lambda: K.gemm(0, x, Wq, R, 3 * H, H, out=o_q, bias=bq)),
    ("ao    plain", lambda: K.gemm(0, x, Wa, R, H, H, out=o_b)),
    ("ao    +bias+drop+res(fp32 pre-LN, LN recomputed)->fp32", lambda: K.gemm(0, x, Wa, R, H, H, out=o_32, bias=ba, dropout_p=0.1, seed=3, residual=pre32, res_ln=(mean, rstd, g, b))),
    ("ao    +bias+drop+res(fp32)->fp32", lambda: K.gemm(0, x, Wa, R, H, H, out=o_32, bias=ba, dropout_p=0.1, seed=3, residual=pre32)),
    ("ao    +bias+res(fp32)->fp32 (no dropout)", lambda: K.gemm(0, x, Wa, R, H, H, out=o_32, bias=ba, residual=pre32)),
    ("ff1   plain", lambda: K.gemm(0, x, W1, R, F, H, out=o_h)),
    ("ff1   +bias+gelu+gelu'", lambda: K.gemm(0, x, W1, R, F, H, out=o_h, bias=b1, act=1, preact=hp, preact_is_grad=True)),
    ("ff2   plain", lambda: K.gemm(0, h, W2, R, H, F, out=o_b)),
    ("ff2   +bias+drop+res(fp32 pre-LN, LN recomputed)->fp32", lambda: K.gemm(0, h, W2, R, H, F, out=o_32, bias=b2, dropout_p=0.1, seed=3, residual=pre32, res_ln=(mean, rstd, g, b))),
    ("ff2   +bias+drop+res(fp32)->fp32", lambda: K.gemm(0, h, W2, R, H, F, out=o_32, bias=b2, dropout_p=0.1, seed=3, residual=pre32)),
    ("ff2   +bias+res(fp32)->fp32 (no dropout)", lambda: K.gemm(0, h, W2, R, H, F, out=o_32, bias=b2, residual=pre32)),
    ("ff2   +bias+res(bf16)->bf16", lambda: K.gemm(0, h, W2, R, H, F, out=o_b, bias=b2, residual=x)),
]
pre16 = pre32.half(); o_16 = torch.empty(R, H, device=dev, dtype=torch.float16)
rows += [
    ("ao    +bias+drop+res(fp16 pre-LN, LN recomputed)->fp16", lambda: K.gemm(0, x, Wa, R, H, H, out=o_16, bias=ba, dropout_p=0.1, seed=3, residual=pre16, res_ln=(mean, rstd, g, b))),
    ("ff2   +bias+drop+res(fp16 pre-LN, LN recomputed)->fp16", lambda: K.gemm(0, h, W2, R, H, F, out=o_16, bias=b2, dropout_p=0.1, seed=3, residual=pre16, res_ln=(mean, rstd, g, b))),
    ("LayerNorm fwd, fp32 rows -> bf16", lambda: K.layernorm_fwd(pre32, R, H, g, b, 1e-12)),
    ("LayerNorm fwd, fp16 rows -> bf16", lambda: K.layernorm_fwd(pre16, R, H, g, b, 1e-12)),
]
dyb = torch.randn(R, H, device=dev).to(bf); dgz, dbz, csz = (torch.zeros(H, device=dev) for _ in range(3)); dx2 = torch.empty(R, H, device=dev, dtype=bf)
rows += [
    ("LayerNorm bwd (+dx2, dropout, colsum), fp32 rows", lambda: K.layernorm_bwd(dyb, pre32, R, H, g, mean, rstd, dgz, dbz, dx2=dx2, dropout_p=0.1, seed=5, colsum=csz)),
    ("LayerNorm bwd (+dx2, dropout, colsum), fp16 rows", lambda: K.layernorm_bwd(dyb, pre16, R, H, g, mean, rstd, dgz, dbz, dx2=dx2, dropout_p=0.1, seed=5, colsum=csz)),
]
for name, f in rows:
    print(f"{name:62s} {bench(f, n=20):7.1f} us", flush=True)
