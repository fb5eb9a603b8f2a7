"""GPU probe: ping-pong GEMM kernel vs the 2-phase 256x256 kernel, same process, interleaved rounds (correctness + time)."""
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


shapes = [(45120, 3072, 768, "bGp"), (45120, 768, 3072, "bdr"), (45120, 3072, 768, "gc"), (45120, 768, 3072, "r"), (45120, 2304, 768, "b"),
          (45120, 768, 2304, "r"), (45120, 768, 768, "bdr"), (31360, 2048, 512, "bGp"), (31360, 512, 2048, "bsr"), (31360, 1536, 512, "b"),
          (31360, 512, 2048, ""), (31360, 512, 512, "bsr"), (8192, 8192, 8192, ""), (36096, 3072, 3072, "")]
torch.manual_seed(0)
for M, N, Kd, fl in shapes:
    A = torch.randn(M, Kd, device="cuda").to(bf)
    Bm = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    kw = {}
    if "b" in fl: kw["bias"] = torch.randn(N, device="cuda")
    if "G" in fl: kw["act"] = 1
    if "p" in fl: kw["preact"] = torch.empty(M, N, device="cuda", dtype=bf); kw["preact_is_grad"] = 1
    if "g" in fl: kw["gelu_in"] = torch.rand(M, N, device="cuda").to(bf); kw["gelu_in_is_grad"] = 1
    if "d" in fl: kw["dropout_p"] = 0.1; kw["seed"] = 1234
    if "s" in fl: kw["row_scale"] = torch.ones(32, device="cuda"); kw["rows_per_group"] = (M + 31) // 32
    if "r" in fl: kw["residual"] = torch.randn(M, N, device="cuda").to(bf)
    outs, ts = [], [[], []]
    for rnd in range(3):
        for pp in (0, 1):
            L.lib.lav_gemm_select(0, pp)
            out = torch.empty(M, N, device="cuda", dtype=bf)
            if "c" in fl: kw["colsum"] = torch.zeros(N, device="cuda")
            ts[pp].append(run(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out, **kw), 5))
            if rnd == 0: outs.append(out.float())
    d = (outs[0] - outs[1]).abs().max().item()
    ref = outs[0].abs().max().item()
    t0, t1 = min(ts[0]), min(ts[1])
    fl_ = 2.0 * M * N * Kd
    print(f"{M:6d} {N:5d} {Kd:5d} {fl:4s}  2-phase {t0:7.1f} us {fl_/t0/1e6:6.0f} TF | ping-pong {t1:7.1f} us {fl_/t1/1e6:6.0f} TF | x{t0/t1:.2f} | max|d| {d:.3g} (|out| {ref:.3g})")
L.lib.lav_gemm_select(0, 0)
