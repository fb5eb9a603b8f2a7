"""GPU probe: LayerNorm backward on the step's shapes (us per call, effective HBM GB/s on 3 tensors of rows x C bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
bf = torch.bfloat16


def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, C, extra in ((31360, 512, False), (31360, 512, True), (36096, 768, True), (125440, 256, False), (501760, 128, False), (7840, 1024, False)):
    x = torch.randn(rows, C, device="cuda").to(bf); dy = torch.randn(rows, C, device="cuda").to(bf)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, mean, rstd = K.layernorm_fwd(x, rows, C, g, b, 1e-5)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); cs = torch.zeros(C, device="cuda")
    add = torch.randn(rows, C, device="cuda").to(bf)
    if extra:
        dx2 = torch.empty(rows, C, device="cuda", dtype=bf)
        f = lambda: K.layernorm_bwd(dy, x, rows, C, g, mean, rstd, dg, db, dx2=dx2, dropout_p=0.1, seed=3, colsum=cs)
        nt = 4
    else:
        f = lambda: K.layernorm_bwd(dy, x, rows, C, g, mean, rstd, dg, db, add_in=add)
        nt = 4
    us = t(f)
    print(f"rows={rows:6d} C={C:4d} extra={extra}: {us:6.1f} us  {nt*rows*C*2/us/1e3:6.0f} GB/s;  fwd {t(lambda: K.layernorm_fwd(x, rows, C, g, b, 1e-5)):5.1f} us")

# Swin-L 384^2 (cfg4, B = 8) shapes, incl. the PatchMerging gather form (rows = merged rows, C = 4 * C0)
print("cfg4 shapes:")
for rows, C, gather in ((368640, 192, None), (92160, 768, (96, 96, 192)), (92160, 384, None), (23040, 1536, (48, 48, 384)), (23040, 768, None),
                        (5760, 3072, (24, 24, 768)), (5760, 1536, None)):
    if gather is None:
        x = torch.randn(rows, C, device="cuda").to(bf)
    else:
        x = torch.randn(rows * 4, C // 4, device="cuda").to(bf)
    dy = torch.randn(rows, C, device="cuda").to(bf)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, mean, rstd = K.layernorm_fwd(x, rows, C, g, b, 1e-5, gather=gather)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    us = t(lambda: K.layernorm_bwd(dy, x, rows, C, g, mean, rstd, dg, db, gather=gather))
    uf = t(lambda: K.layernorm_fwd(x, rows, C, g, b, 1e-5, gather=gather))
    print(f"rows={rows:6d} C={C:4d} gather={gather}: bwd {us:7.1f} us {3*rows*C*2/us/1e3:6.0f} GB/s;  fwd {uf:6.1f} us {2*rows*C*2/uf/1e3:6.0f} GB/s")
