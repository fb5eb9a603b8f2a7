"""Runs the fusion-attention kernels a few times (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
n, L, heads, p = 160, 282, 12, 0.1          # the cfg2 step: 32 MTM + 128 VTM sequences
Hd = heads * 64
qkv = torch.randn(n * L, 3 * Hd, device="cuda").bfloat16()
km = torch.ones(n, L, dtype=torch.int32, device="cuda")
att = K.Attn(1, heads, 64, n_seq=n, L=L, key_mask=km, dropout_p=p, seed=123)
lse = torch.empty(att.lse_elems(), device="cuda")
out = torch.empty(n * L, Hd, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(n * L, Hd, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
for _ in range(3):
    att.fwd(qkv, out, lse)
    att.bwd(qkv, out, dout, lse, dqkv, None)
torch.cuda.synchronize()


def _t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"n={n} L={L}: fwd {_t(lambda: att.fwd(qkv, out, lse)):.1f} us, bwd (dq + dkv) {_t(lambda: att.bwd(qkv, out, dout, lse, dqkv, None)):.1f} us")
