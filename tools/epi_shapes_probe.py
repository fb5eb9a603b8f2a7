"""GPU probe: the epilogue-heavy GEMM shapes of the step, default kernel selection vs LAV_GEMM_SMALL=1 / LAV_GEMM_NO_HUGE=1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
bf = torch.bfloat16


def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(layout, M, N, Kd, flags):
    A = torch.randn(M, Kd, device="cuda").to(bf)
    Bm = (torch.randn(N, Kd, device="cuda") if layout == 0 else torch.randn(Kd, N, device="cuda")).to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    kw = {}
    if "b" in flags: kw["bias"] = torch.randn(N, device="cuda")
    if "G" in flags: kw["act"] = 1
    if "p" in flags: kw["preact"] = torch.empty(M, N, device="cuda", dtype=bf); kw["preact_is_grad"] = True
    if "g" in flags: kw["gelu_in"] = torch.randn(M, N, device="cuda").to(bf); kw["gelu_in_is_grad"] = True
    if "s" in flags: kw["row_scale"] = torch.ones(M // 1568 + 1, device="cuda"); kw["rows_per_group"] = 1568
    if "c" in flags: kw["colsum"] = torch.zeros(N, device="cuda")
    if "r" in flags: kw["residual"] = torch.randn(M, N, device="cuda").to(bf)
    if "d" in flags: kw["dropout_p"] = 0.1; kw["seed"] = 7
    us = t(lambda: K.gemm(layout, A, Bm, M, N, Kd, out=out, **kw))
    print(f"{'NT NN'.split()[layout]} {M:7d} {N:5d} {Kd:5d} {flags:5s}: {us:7.1f} us {2.0*M*N*Kd/us/1e6:6.0f} TF")


for c in [(1, 36096, 3072, 768, "gc"), (1, 36096, 3072, 768, ""), (1, 31360, 2048, 512, "gsc"), (1, 31360, 2048, 512, ""),
          (0, 36096, 3072, 768, "bGp"), (0, 36096, 3072, 768, "b"), (0, 31360, 2048, 512, "bGp"), (0, 31360, 2048, 512, "b"),
          (0, 36096, 768, 3072, "bdr"), (0, 36096, 768, 768, "bdr"), (0, 31360, 512, 512, "bsr"), (1, 31360, 512, 512, "s"),
          (0, 501760, 512, 128, "bGp"), (1, 501760, 512, 128, "gsc"), (1, 125440, 1024, 256, "gsc"), (0, 125440, 1024, 256, "bGp")]:
    case(*c)
