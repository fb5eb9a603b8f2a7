"""GPU probe: s_memtime stamps inside win_bwd1<8> (stage-2 shape): where a wave's step goes.  Waves 0 and 4 of workgroup 0 (one SIMD), third sample.
stamps per step: 0 S/dP accumulators readable | 1 soft-max VALU done | 2 Q^T/dO^T fragments landed | 3 tile flag seen | 4 dQ tile in registers |
5 dS^T fragments back from LDS | 6 dQ MFMAs done (accumulator readable) | 7 tile + flag written"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K, _lib as L
side, Cn, heads, B = 14, 512, 16, 32
M = B * 5 * side * side
qkv = torch.randn(M, 3 * Cn, device="cuda").bfloat16()
tbl = torch.randn(2535, heads, device="cuda") * 0.02
att = K.Attn(0, heads, 32, B=B, D=5, H=side, W=side, wd=5, wh=7, ww=7, sd=0, sh=3, sw=3, cfg_wd=8, cfg_wh=7, cfg_ww=7, bias_table=tbl)
lse = torch.empty(att.lse_elems(), device="cuda")
out = torch.empty(M, Cn, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(M, Cn, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
att.fwd(qkv, out, lse)
for _ in range(3):
    att.bwd(qkv, out, dout, lse, dqkv, None)
prof = torch.zeros(2 * 8 * 8 + 32, dtype=torch.int64, device="cuda")
L.lib.lav_probe_win_prof(prof.data_ptr())
att.bwd(qkv, out, dout, lse, dqkv, None)
torch.cuda.synchronize()
L.lib.lav_probe_win_prof(None)
o = prof.cpu()[128:].view(2, 16)
p = prof.cpu()[:128].view(2, 8, 8)
for w in range(2):
    print(f"wave {4 * w}: per step, cycles since the step's stamp 0 (stamp 0 = cycles since the previous step's stamp 0)")
    prev = None
    for s in range(8):
        t = p[w, s].tolist()
        base = t[0]
        d0 = 0 if prev is None else base - prev
        prev = base
        print(f"  step {s}: +{d0:6d} | " + " ".join(f"{(x - base) if x else -1:6d}" for x in t[1:]))
for w in range(2):
    t = o[w].tolist()
    print(f"wave {4 * w} sample: top (strips, delta) {t[1]-t[0]}, DMA issue {t[2]-t[1]}, top barrier {t[3]-t[2]}, steps + stores {t[4]-t[3]}, DMA wait {t[5]-t[4]}, end barrier {t[6]-t[5]}; first step stamp0 at +{p[w,0,0].item()-t[3]}")
