"""GPU probe: per-step time of the 24 window bias-fragment builds (Swin-B, B = 32) with the index arithmetic in the kernel vs the gather through the per-geometry map."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from tools._bench import bench
for use_map in (False, True, False, True):
    K.BIAS_MAP = use_map
    tot = 0.0
    for (side, C, heads, n, sh) in ((56, 128, 4, 2, 3), (28, 256, 8, 2, 3), (14, 512, 16, 18, 3), (7, 1024, 32, 2, 0)):
        for shift in (0, sh):
            tbl = torch.randn(2535, heads, device="cuda") * 0.02
            att = K.Attn(0, heads, 32, B=32, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=shift, sw=shift, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
            t = bench(lambda: K.L.lib.lav_attention_build_bias(K._s(), K.C.byref(att.d)), n=20)
            tot += t * n / 2
    print(f"bias_map={use_map}: {tot:.1f} us of table builds per step (24 blocks)")
