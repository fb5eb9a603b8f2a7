"""GPU probe: window attention forward / backward per Swin-B stage, with HBM-bound floors (bytes / 6.3 TB/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K


def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def window(B, side, C, heads, shift, bias_grad=True):
    M = B * 5 * side * side
    qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
    tbl = torch.randn(2535, heads, device="cuda") * 0.02
    att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=shift, sw=shift, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
    lse = torch.empty(att.lse_elems(), device="cuda")
    out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn(M, C, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    dtbl = torch.zeros_like(tbl)
    t_f = bench(lambda: att.fwd(qkv, out, lse))
    t_b = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, dtbl if bias_grad else None))
    by_f = M * 4 * C * 2
    by_b = M * (3 + 1 + 1 + 3) * C * 2
    print(f"window B={B} side={side} C={C} heads={heads} shift={shift} bias_grad={bias_grad}: fwd {t_f:.0f} us (floor {by_f/6.3e6:.0f})  "
          f"bwd {t_b:.0f} us (floor {by_b/6.3e6:.0f})")
    return t_f, t_b


if __name__ == "__main__":
    tot_f = tot_b = 0.0
    for side, C, heads, depth in ((56, 128, 4, 2), (28, 256, 8, 2), (14, 512, 16, 18), (7, 1024, 32, 2)):
        for shift in ((0, 3) if side > 7 else (0,)):
            f, b = window(32, side, C, heads, shift)
            n = depth / (2 if side > 7 else 1)
            tot_f += f * n; tot_b += b * n
    print(f"per step: window fwd {tot_f/1e3:.2f} ms, bwd {tot_b/1e3:.2f} ms")
    window(32, 14, 512, 16, 3, bias_grad=False)
