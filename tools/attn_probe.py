"""GPU probe: time the attention kernels on the Swin-B / fusion shapes (forward, dQ pass, dK/dV pass)."""
import sys, os, time
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
    return e0.elapsed_time(e1) / n

def window(B, side, C, heads, shift, nobias_grad=False):
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
    t_b = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, None if nobias_grad else dtbl))
    nw = B * 5 // 5 * (side // 7) ** 2
    fl = nw * heads * 4 * 245 * 245 * 32
    print(f"window B={B} side={side} C={C} heads={heads} shift={shift} nobias={nobias_grad}: fwd {t_f*1e3:.0f} us ({fl/t_f/1e9:.0f} TF)  bwd {t_b*1e3:.0f} us ({2.5*fl/t_b/1e9:.0f} TF)")

def seq(n, L, heads, p):
    Hd = heads * 64
    qkv = torch.randn(n * L, 3 * Hd, device="cuda").bfloat16()
    km = torch.ones(n, L, dtype=torch.int32, device="cuda")
    att = K.Attn(1, heads, 64, n_seq=n, L=L, key_mask=km, dropout_p=p, seed=123)
    lse = torch.empty(att.lse_elems(), device="cuda")
    out = torch.empty(n * L, Hd, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn(n * L, Hd, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    t_f = bench(lambda: att.fwd(qkv, out, lse))
    t_b = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, None))
    fl = n * heads * 4 * L * L * 64
    print(f"seq n={n} L={L} heads={heads} p={p}: fwd {t_f*1e3:.0f} us ({fl/t_f/1e9:.0f} TF)  bwd {t_b*1e3:.0f} us ({2.5*fl/t_b/1e9:.0f} TF)")

if __name__ == "__main__":
    window(32, 56, 128, 4, 3); window(32, 56, 128, 4, 3, True); window(32, 56, 128, 4, 0)
    window(32, 14, 512, 16, 3); window(32, 14, 512, 16, 3, True)
    seq(128, 282, 12, 0.1); seq(128, 282, 12, 0.0); seq(32, 282, 12, 0.1)
