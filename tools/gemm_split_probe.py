"""GPU probe: where a 256x256 NT GEMM launch spends its time -- full launch vs no epilogue (lav_gemm_select(5, 1)) vs no k-loop
(5, 2) vs neither (5, 3: launch + prologue + tile scheduling only); (5, 4) keeps everything but the LDS staging writes of the
epilogue (garbage results) = what the staging itself costs."""
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
shapes = [(45120, 3072, 768, ""), (45120, 3072, 768, "b"), (45120, 3072, 768, "bGp"), (45120, 3072, 768, "gc"), (45120, 768, 3072, "bdr"), (45120, 768, 3072, ""),
          (31360, 2048, 512, "bGp"), (31360, 2048, 512, ""), (31360, 512, 2048, "bsr"), (31360, 512, 2048, "")]
torch.manual_seed(0)
for M, N, Kd, fl in shapes:
    A = torch.randn(M, Kd, device="cuda").to(bf); Bm = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    kw = {}
    if "b" in fl: kw["bias"] = torch.randn(N, device="cuda")
    if "G" in fl: kw["act"] = 1
    if "p" in fl: kw["preact"] = torch.empty(M, N, device="cuda", dtype=bf); kw["preact_is_grad"] = 1
    if "g" in fl: kw["gelu_in"] = torch.rand(M, N, device="cuda").to(bf); kw["gelu_in_is_grad"] = 1
    if "d" in fl: kw["dropout_p"] = 0.1; kw["seed"] = 1234
    if "s" in fl: kw["row_scale"] = torch.ones(32, device="cuda"); kw["rows_per_group"] = (M + 31) // 32
    if "r" in fl: kw["residual"] = torch.randn(M, N, device="cuda").to(bf)
    if "c" in fl: kw["colsum"] = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=bf)
    res = {}
    for rnd in range(3):
        for d in (0, 1, 2, 3, 4):
            L.lib.lav_gemm_select(5, d)
            res.setdefault(d, []).append(run(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out, **kw), 5))
    t = {d: min(v) for d, v in res.items()}
    peak = 2.0 * M * N * Kd / 2.5e9
    print(f"{M:6d} {N:5d} {Kd:5d} {fl:4s} full {t[0]:6.1f} | no epilogue {t[1]:6.1f} | no k-loop {t[2]:6.1f} | neither {t[3]:5.1f} us | MFMA at peak {peak:6.1f} "
          f"-> k-loop {t[1]-t[3]:6.1f} ({peak/(t[1]-t[3])*100:4.1f} % of peak), epilogue {t[0]-t[1]:6.1f} of which LDS staging writes {t[0]-t[4]:5.1f}")
L.lib.lav_gemm_select(5, 0)
