"""GPU probe: window attention with the row-major (rows, 3C) qkv operand vs the head-major one ([q | k | v][head][token][32],
lav_attn_desc.qkv_headmajor) on the Swin-B stage shapes of the cfg2 step (B = 32): forward, dQ / dK / dV, split bias-table gradient;
interleaved rounds in one process, per-step totals weighted by the stage depths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K


def bench(f, n=8):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {0: [0.0, 0.0, 0.0], 1: [0.0, 0.0, 0.0]}
for side, C, heads, depth in ((56, 128, 4, 2), (28, 256, 8, 2), (14, 512, 16, 18), (7, 1024, 32, 2)):
    for shift in ((0, 3) if side > 7 else (0,)):
        B = 32
        M = B * 5 * side * side
        qkv = torch.randn(M, 3 * C, device="cuda").bfloat16()
        qkv_hm = qkv.view(M, 3, heads, 32).permute(1, 2, 0, 3).contiguous().view(M, 3 * C)
        tbl = torch.randn(2535, heads, device="cuda") * 0.02
        out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
        dout = torch.randn(M, C, device="cuda").bfloat16()
        dqkv = torch.empty_like(qkv)
        dtbl = torch.zeros_like(tbl)
        atts = {}
        for hm in (0, 1):
            atts[hm] = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=shift, sw=shift, cfg_wd=8, cfg_wh=7, cfg_ww=7,
                              bias_table=tbl, qkv_headmajor=hm)
        lse = torch.empty(atts[0].lse_elems(), device="cuda")
        t = {0: [[], [], []], 1: [[], [], []]}
        for rnd in range(5):
            for hm, x in ((0, qkv), (1, qkv_hm)):
                a = atts[hm]
                t[hm][0].append(bench(lambda: a.fwd(x, out, lse)))
                t[hm][1].append(bench(lambda: a.bwd(x, out, dout, lse, dqkv, None)))
                t[hm][2].append(bench(lambda: a.bwd_bias(x, dout, lse, dtbl)))
        med = {hm: [sorted(v)[len(v) // 2] for v in t[hm]] for hm in (0, 1)}
        n = depth / (2 if side > 7 else 1)
        for hm in (0, 1):
            for i in range(3):
                tot[hm][i] += med[hm][i] * n
        print(f"side {side:2d} C {C:4d} heads {heads:2d} shift {shift}: fwd {med[0][0]:6.1f} -> {med[1][0]:6.1f} us | dq+dkv {med[0][1]:6.1f} -> {med[1][1]:6.1f} us | "
              f"bias grad {med[0][2]:6.1f} -> {med[1][2]:6.1f} us   (row-major -> head-major)", flush=True)
print(f"per step: fwd {tot[0][0] / 1e3:.2f} -> {tot[1][0] / 1e3:.2f} ms, dq+dkv {tot[0][1] / 1e3:.2f} -> {tot[1][1] / 1e3:.2f} ms, "
      f"bias gradient (side stream) {tot[0][2] / 1e3:.2f} -> {tot[1][2] / 1e3:.2f} ms")
