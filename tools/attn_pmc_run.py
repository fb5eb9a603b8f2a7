"""Runs the attention kernels of the step a few times each for rocprofv3 --pmc passes: Swin-B stage-2 window attention (B = 32,
14 x 14 x 5 tokens, 16 heads, shifted) forward / backward with the bias-table gradient, and the fusion attention (160 sequences of
282 tokens, 12 heads, dropout 0.1) forward / backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
bf = torch.bfloat16
torch.manual_seed(0)
B, side, C, heads = 32, 14, 512, 16
M = B * 5 * side * side
qkv = torch.randn(M, 3 * C, device="cuda").to(bf)
tbl = torch.randn(2535, heads, device="cuda") * 0.02
att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=3, sw=3, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
lse = torch.empty(att.lse_elems(), device="cuda"); out = torch.empty(M, C, device="cuda", dtype=bf)
dout = torch.randn(M, C, device="cuda").to(bf); dqkv = torch.empty_like(qkv); dtbl = torch.zeros_like(tbl)
for _ in range(4):
    att.fwd(qkv, out, lse)
    att.bwd(qkv, out, dout, lse, dqkv, dtbl)
n, L, H = 160, 282, 12
qkv2 = torch.randn(n * L, 3 * H * 64, device="cuda").to(bf)
km = torch.ones(n, L, dtype=torch.int32, device="cuda")
att2 = K.Attn(1, H, 64, n_seq=n, L=L, key_mask=km, dropout_p=0.1, seed=7)
lse2 = torch.empty(att2.lse_elems(), device="cuda"); out2 = torch.empty(n * L, H * 64, device="cuda", dtype=bf)
dout2 = torch.randn(n * L, H * 64, device="cuda").to(bf); dqkv2 = torch.empty_like(qkv2)
for _ in range(4):
    att2.fwd(qkv2, out2, lse2)
    att2.bwd(qkv2, out2, dout2, lse2, dqkv2, None)
torch.cuda.synchronize()
