"""Runs one NT and one TN large-tile GEMM a few times (for rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE ...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
M, N, Kd = 4096, 4096, 8192
A = torch.randn(M, Kd, device="cuda").bfloat16(); B = torch.randn(N, Kd, device="cuda").bfloat16()
for _ in range(3):
    K.gemm(0, A, B, M, N, Kd)
At = torch.randn(Kd, M, device="cuda").bfloat16(); Bt = torch.randn(Kd, N, device="cuda").bfloat16()
out = torch.zeros(M, N, device="cuda")
for _ in range(3):
    K.gemm(2, At, Bt, M, N, Kd, out=out, accumulate=True, splits=1)
torch.cuda.synchronize()
