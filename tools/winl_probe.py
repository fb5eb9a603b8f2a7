"""GPU probe: large-window attention (attention_winl.hip) on the Swin-L 384^2 stage shapes of cfg4 (B = 8, 720-token windows):
forward, dQ + dK/dV without and with the bias-table gradient.  LAV_WINL=0 in the environment times the generic kernels instead."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from tools._bench import bench

B = 8
tot = [0.0, 0.0, 0.0]
for side, C, heads, depth in ((96, 192, 6, 2), (48, 384, 12, 2), (24, 768, 24, 18), (12, 1536, 48, 2)):
    for shift in ((0, 6) if side > 12 else (0,)):
        M = B * 5 * side * side
        qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
        tbl = torch.randn(15 * 23 * 23, heads, device="cuda") * 0.02
        att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=12, ww=12, sd=0, sh=shift, sw=shift, cfg_wd=8, cfg_wh=12, cfg_ww=12, bias_table=tbl)
        lse = torch.empty(att.lse_elems(), device="cuda")
        out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
        dout = torch.randn(M, C, device="cuda").bfloat16()
        dqkv = torch.empty_like(qkv); dtbl = torch.zeros_like(tbl)
        tf = bench(lambda: att.fwd(qkv, out, lse), n=5)
        tb = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, None), n=5)
        tbb = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, dtbl), n=5)
        nw = B * (side // 12) ** 2
        fl = nw * heads * 4 * 720 * 720 * 32
        print(f"side={side} heads={heads} shift={shift}: fwd {tf:7.1f} us ({fl/tf/1e6:5.0f} TF/s)  dq+dkv {tb:7.1f} us ({2.5*fl/tb/1e6:5.0f} TF/s)  with bias-grad {tbb:7.1f} us", flush=True)
        n = depth / (2 if side > 12 else 1)
        tot[0] += tf * n; tot[1] += tb * n; tot[2] += tbb * n
print(f"per cfg4 step: window fwd {tot[0]/1e3:.2f} ms, dq+dkv {tot[1]/1e3:.2f} ms, dq+dkv with bias-grad {tot[2]/1e3:.2f} ms")
