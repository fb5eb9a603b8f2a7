"""GPU probe: one Swin-B stage's window attention backward (no bias gradient) for timing ablations: tools/win_one.py [side C heads]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from tools.win_probe import bench
side, C, heads = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (14, 512, 16)
B = 32
M = B * 5 * side * side
qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
tbl = torch.randn(2535, heads, device="cuda") * 0.02
att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=3, sw=3, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
lse = torch.empty(att.lse_elems(), device="cuda")
out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(M, C, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
att.fwd(qkv, out, lse)
print(f"side {side} C {C} heads {heads}: fwd {bench(lambda: att.fwd(qkv, out, lse)):.1f} us, bwd (no bias gradient) {bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, None)):.1f} us", flush=True)
