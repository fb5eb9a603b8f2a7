"""GPU probe: window attention variants (lav_gemm_select 16/17/18) on the Swin-B stage shapes, interleaved A/B in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from lavender_amd import _lib as L


def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def setup(B, side, C, heads, shift):
    M = B * 5 * side * side
    qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
    tbl = torch.randn(2535, heads, device="cuda") * 0.02
    att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=shift, sw=shift, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
    lse = torch.empty(att.lse_elems(), device="cuda")
    out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn(M, C, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    dtbl = torch.zeros_like(tbl)
    return att, qkv, out, lse, dout, dqkv, dtbl


def sel(which, v):
    return L.lib.lav_gemm_select(which, v)


if __name__ == "__main__":
    tot_f = tot_b = tot_bias = 0.0
    for side, C, heads, depth in ((14, 512, 16, 18), (56, 128, 4, 2), (28, 256, 8, 2), (7, 1024, 32, 2)):
        for shift in ((0, 3) if side > 7 else (0,)):
            att, qkv, out, lse, dout, dqkv, dtbl = setup(32, side, C, heads, shift)
            tf = bench(lambda: att.fwd(qkv, out, lse))
            tb = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, None))
            tbb = bench(lambda: att.bwd_bias(qkv, dout, lse, dtbl))
            nw = 32 * (side // 7) ** 2
            fl = nw * heads * 4 * 245 * 245 * 32
            print(f"side={side} shift={shift}: fwd {tf:7.1f} us ({fl/tf/1e6:5.0f} TF/s)  dq+dkv {tb:7.1f} us ({2.5*fl/tb/1e6:5.0f} TF/s)  bias-grad {tbb:7.1f} us", flush=True)
            n = depth / (2 if side > 7 else 1)
            tot_f += tf * n; tot_b += tb * n; tot_bias += tbb * n
    print(f"per step: window fwd {tot_f/1e3:.2f} ms, dq+dkv {tot_b/1e3:.2f} ms, bias-grad (side stream) {tot_bias/1e3:.2f} ms")
