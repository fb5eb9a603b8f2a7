"""One grouped weight-gradient launch per group (fusion layer, Swin-B stage 2 / 3) for PMC passes (tools/pmc_fetch_one.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
GROUPS = {"stage2": ([(512, 2048, 31360), (2048, 512, 31360), (512, 512, 31360), (1536, 512, 31360)], 4),
          "stage3": ([(1024, 4096, 7840), (4096, 1024, 7840), (1024, 1024, 7840), (3072, 1024, 7840)], 1),
          "fusion": ([(768, 3072, 45120), (3072, 768, 45120), (768, 768, 45120), (2304, 768, 45120)], 2)}
which = sys.argv[1] if len(sys.argv) > 1 else "fusion"
shapes, gs = GROUPS[which]
jobs = []
for (M, N, Kd) in shapes:
    jobs.append(dict(A=torch.randn(Kd, M, device="cuda").bfloat16(), B=torch.randn(Kd, N, device="cuda").bfloat16(), out=torch.zeros(M, N, device="cuda"), fallback_splits=1))
for _ in range(6):
    K.gemm_tn_grouped(jobs, gs)
torch.cuda.synchronize()
