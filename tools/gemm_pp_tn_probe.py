"""GPU probe: ping-pong weight-gradient (TN) kernel vs the 2-phase 256x256 TN kernel: correctness + time, same process."""
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


shapes = [(768, 3072, 45120, ""), (3072, 768, 45120, ""), (2304, 768, 45120, "R"), (512, 2048, 31360, "kR"), (2048, 512, 31360, ""),
          (1536, 512, 31360, "R"), (768, 768, 45120, ""), (512, 512, 31360, "kR"), (4096, 1024, 7840, ""), (256, 1024, 125440, "kR")]
torch.manual_seed(0)
for M, N, Kd, fl in shapes:
    A = torch.randn(Kd, M, device="cuda").to(bf)
    Bm = (torch.randn(Kd, N, device="cuda") * 0.05).to(bf)
    kw = dict(accumulate=True, splits=K.splits_for(M, N, Kd, "k" in fl))
    if "k" in fl:
        B_ = 32
        keep = (torch.rand(B_, device="cuda") > 0.2).float()
        kw.update(k_keep=keep, k_rows_per_group=Kd // B_, alpha=1.25)
    outs, ts, rs = [], [[], []], []
    for rnd in range(3):
        for pp in (0, 1):
            L.lib.lav_gemm_select(2, pp)
            out = torch.zeros(M, N, device="cuda")
            if "R" in fl:
                kw["rowsum_a"] = torch.zeros(M, device="cuda")
            K.gemm(2, A, Bm, M, N, Kd, out=out, **kw)
            if rnd == 0:
                outs.append(out.clone()); rs.append(kw["rowsum_a"].clone() if "R" in fl else None)
            ts[pp].append(run(lambda: K.gemm(2, A, Bm, M, N, Kd, out=out, **kw), 5))
    d = (outs[0] - outs[1]).abs().max().item()
    ref = outs[0].abs().max().item()
    dr = (rs[0] - rs[1]).abs().max().item() / (rs[0].abs().max().item() + 1e-9) if rs[0] is not None else 0.0
    t0, t1 = min(ts[0]), min(ts[1])
    fl_ = 2.0 * M * N * Kd
    print(f"{M:5d} {N:5d} {Kd:6d} {fl:3s} splits {kw['splits']:3d}  2-phase {t0:7.1f} us {fl_/t0/1e6:6.0f} TF | ping-pong {t1:7.1f} us {fl_/t1/1e6:6.0f} TF | x{t0/t1:.2f} | max|d| {d:.3g} (|out| {ref:.3g}) rowsum rel {dr:.2g}")
L.lib.lav_gemm_select(2, 0)
