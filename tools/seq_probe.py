"""GPU probe: fusion (sequence) attention forward / backward on the step's shapes, with and without dropout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from tools._bench import bench

SHAPES = ((160, 282, 0.1), (160, 282, 0.0), (128, 282, 0.1), (32, 282, 0.1), (64, 276, 0.1), (40, 757, 0.1), (40, 757, 0.0))   # last two: cfg4 (LAV_SEQL=0: generic kernels)
for n, L, p in SHAPES:
    heads, Hd = 12, 768
    qkv = torch.randn(n * L, 3 * Hd, device="cuda").bfloat16()
    km = torch.ones(n, L, dtype=torch.int32, device="cuda")
    att = K.Attn(1, heads, 64, n_seq=n, L=L, key_mask=km, dropout_p=p, seed=123)
    lse = torch.empty(att.lse_elems(), device="cuda")
    out = torch.empty(n * L, Hd, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn(n * L, Hd, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    tf = bench(lambda: att.fwd(qkv, out, lse))
    tb = bench(lambda: att.bwd(qkv, out, dout, lse, dqkv, None))
    fl = n * heads * 4 * L * L * 64
    print(f"seq n={n} L={L} p={p}: fwd {tf:7.1f} us ({fl/tf/1e6:5.0f} TF/s)  bwd {tb:7.1f} us ({2.5*fl/tb/1e6:5.0f} TF/s)", flush=True)
