"""Runs the step's dominant GEMMs a few times each for rocprofv3 --pmc passes (SQ counters): the 256x256 K-contiguous kernel on
45120 x 3072 x 768 (plain store and bias + GELU + stored GELU') and 31360 x 2048 x 512, and the ping-pong weight-gradient kernel on
(3072 x 768, K = 45120)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
bf = torch.bfloat16
torch.manual_seed(0)
for M, N, Kd, fl in ((45120, 3072, 768, ""), (45120, 3072, 768, "bGp"), (31360, 2048, 512, "")):
    A = torch.randn(M, Kd, device="cuda").to(bf); Bm = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    kw = {}
    if "b" in fl: kw["bias"] = torch.randn(N, device="cuda")
    if "G" in fl: kw["act"] = 1
    if "p" in fl: kw["preact"] = torch.empty(M, N, device="cuda", dtype=bf); kw["preact_is_grad"] = 1
    out = torch.empty(M, N, device="cuda", dtype=bf)
    for _ in range(4):
        K.gemm(0, A, Bm, M, N, Kd, out=out, **kw)
dy = torch.randn(45120, 3072, device="cuda").to(bf); x = torch.randn(45120, 768, device="cuda").to(bf)
dw = torch.zeros(3072, 768, device="cuda")
for _ in range(4):
    K.gemm(2, dy, x, 3072, 768, 45120, out=dw, accumulate=True, splits=K.splits_for(3072, 768, 45120))
torch.cuda.synchronize()
