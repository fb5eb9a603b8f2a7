"""GPU probe: persistent tile-walking GEMM kernels (gemm_p256 / gemm_p192l, lav_gemm_select(10, 1)) against the one-tile-per-workgroup
kernels (10, 0) on the step's forward / input-gradient shapes: outputs must be bit-identical (column sums: atomics, compared to 1e-3
relative); interleaved timing rounds in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from lavender_amd import _lib as L
bf = torch.bfloat16


def run(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(45120, 3072, 768, "b"), (45120, 3072, 768, "bGp"), (45120, 3072, 768, "gc"), (45120, 2304, 768, "b"), (45120, 768, 3072, "bdr"),
          (45120, 768, 3072, "bdrO"), (45120, 768, 3072, "r"), (45120, 768, 2304, "r"), (45120, 768, 768, "bdrO"), (45120, 768, 768, ""),
          (31360, 2048, 512, "bGp"), (31360, 2048, 512, "gsc"), (31360, 1536, 512, "b"), (31360, 512, 2048, "bsr"), (31360, 512, 2048, ""),
          (31360, 512, 1536, ""), (31360, 512, 512, "bsr"), (31360, 512, 512, "s"), (125440, 1024, 256, "bGp"), (125440, 256, 1024, "bsr"),
          (125440, 768, 256, "b"), (125440, 256, 256, "bsr"), (501760, 512, 128, "bGp"), (7840, 4096, 1024, "bGp"), (7840, 1024, 4096, "bsr"),
          (7840, 3072, 1024, "b"), (7840, 1024, 1024, "bsr"), (5120, 768, 768, "bGp"), (36096, 3072, 3072, "b"), (8192, 8192, 8192, ""),
          (300, 256, 64, "b"), (2500, 512, 128, "bdr"), (45121, 768, 192, "bdrO")]
torch.manual_seed(0)
tot0 = tot1 = 0.0
bad = 0
for M, N, Kd, fl in shapes:
    A = torch.randn(M, Kd, device="cuda").to(bf); Bm = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    kw = {}
    o32 = "O" in fl
    if "b" in fl: kw["bias"] = torch.randn(N, device="cuda")
    if "G" in fl: kw["act"] = 1
    if "g" in fl: kw["gelu_in"] = torch.rand(M, N, device="cuda").to(bf); kw["gelu_in_is_grad"] = 1
    if "d" in fl: kw["dropout_p"] = 0.1; kw["seed"] = 1234
    if "s" in fl: kw["row_scale"] = (torch.rand(32, device="cuda") > 0.2).float() / 0.8; kw["rows_per_group"] = (M + 31) // 32
    if "r" in fl: kw["residual"] = torch.randn(M, N, device="cuda").to(torch.float32 if o32 else bf)
    outs = {}
    times = {0: [], 1: []}
    for mode in (0, 1):
        L.lib.lav_gemm_select(10, mode)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if o32 else bf)
        k2 = dict(kw)
        if "p" in fl: k2["preact"] = torch.zeros(M, N, device="cuda", dtype=bf); k2["preact_is_grad"] = 1
        if "c" in fl: k2["colsum"] = torch.zeros(N, device="cuda")
        K.gemm(0, A, Bm, M, N, Kd, out=out, **k2)
        torch.cuda.synchronize()
        outs[mode] = (out, k2.get("preact"), k2.get("colsum"))
    same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    if outs[0][2] is not None:
        same = same and bool(((outs[0][2] - outs[1][2]).abs().max() <= 1e-3 * outs[0][2].abs().max().clamp_min(1e-6)))
    finite = bool(torch.isfinite(outs[1][0].float()).all()) and float(outs[1][0].float().abs().max()) > 0
    bad += int(not (same and finite))
    out = outs[1][0]
    k2 = dict(kw)
    if "p" in fl: k2["preact"] = outs[1][1]; k2["preact_is_grad"] = 1
    if "c" in fl: k2["colsum"] = outs[1][2]
    for rnd in range(5):
        for mode in (0, 1):
            L.lib.lav_gemm_select(10, mode)
            times[mode].append(run(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out, **k2), 5))
    t0, t1 = sorted(times[0])[len(times[0]) // 2], sorted(times[1])[len(times[1]) // 2]
    tot0 += t0; tot1 += t1
    print(f"{M:6d} {N:5d} {Kd:5d} {fl:5s} one-tile {t0:7.1f} us | persistent {t1:7.1f} us | x{t0 / t1:5.3f} | {2.0 * M * N * Kd / t1 / 1e6:6.0f} TF/s | "
          f"{'bit-identical' if same else 'MISMATCH'}{'' if finite else ' NONFINITE/ZERO'}", flush=True)
    del A, Bm, outs, out, kw, k2
L.lib.lav_gemm_select(10, 1)
print(f"sum one-tile {tot0:.0f} us, persistent {tot1:.0f} us (x{tot0 / tot1:.3f}); mismatching shapes: {bad}")
sys.exit(1 if bad else 0)
