"""GPU probe: 192-row vs 256-row tiles (lav_gemm_select(7, v)) on the narrow-output GEMMs of the fusion encoder; checks results."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from lavender_amd import _lib as L
from tools._bench import bench
bf = torch.bfloat16
for (M, N, Kd, kw) in ((45120, 768, 3072, "r"), (45120, 768, 3072, "bdr"), (45120, 768, 2304, "r"), (45120, 768, 768, ""), (45120, 768, 768, "bdr"),
                       (45120, 2304, 768, "b"), (36096, 768, 3072, "r"), (31360, 512, 2048, "")):
    A = torch.randn(M, Kd, device="cuda").to(bf); B = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(bf)
    args = {}
    if "b" in kw: args["bias"] = bias
    if "r" in kw: args["residual"] = res
    if "d" in kw: args.update(dropout_p=0.1, seed=5)
    outs = []
    for v in (0, 1, 0, 1):
        L.lib.lav_gemm_select(7, v)
        t = bench(lambda: K.gemm(0, A, B, M, N, Kd, **args))
        o = K.gemm(0, A, B, M, N, Kd, **args).float()
        outs.append(o)
        print(f"{M}x{N}x{Kd} {kw:4s} h192={v}: {t:7.1f} us  {2*M*N*Kd/t/1e6:6.0f} TF/s   max|d vs first| {float((o - outs[0]).abs().max()):.3g}", flush=True)
L.lib.lav_gemm_select(7, 1)
