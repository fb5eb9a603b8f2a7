"""Runs the stage-2 window attention forward / one-pass backward / bias-table gradient a few times (for rocprofv3 --pmc: tools/sq_one.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
side, C, heads, B = 14, 512, 16, 32
M = B * 5 * side * side
qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
tbl = torch.randn(2535, heads, device="cuda") * 0.02
att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=3, sw=3, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
lse = torch.empty(att.lse_elems(), device="cuda")
out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(M, C, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
dtbl = torch.zeros_like(tbl)
for _ in range(6):
    att.fwd(qkv, out, lse)
    att.bwd(qkv, out, dout, lse, dqkv, dtbl)
torch.cuda.synchronize()
