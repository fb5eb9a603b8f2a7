"""GPU probe: 192 x 256 GEMM tiles with two loader waves (lav_gemm_select(9, v): 0 off, 1 where the 192-row tile is chosen = default, 2 on every
step's forward / input-gradient GEMM shapes; checks results against the default path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from lavender_amd import _lib as L
from tools.win_var_probe import bench
bf = torch.bfloat16
SH = ((45120, 2304, 768, "b"), (45120, 768, 768, "bdr"), (45120, 3072, 768, "bg"), (45120, 768, 3072, "bdr"), (45120, 768, 2304, ""), (45120, 768, 768, ""),
      (45120, 3072, 768, "b"), (31360, 1536, 512, "b"), (31360, 512, 512, "b"), (31360, 2048, 512, "bg"), (31360, 512, 2048, "b"),
      (125440, 768, 256, "b"), (125440, 256, 256, "b"), (125440, 1024, 256, "bg"), (125440, 256, 1024, "b"),
      (501760, 384, 128, "b"), (501760, 128, 128, "b"), (501760, 512, 128, "bg"), (501760, 128, 512, "b"), (7840, 3072, 1024, "b"), (7840, 1024, 4096, "b"))
for (M, N, Kd, kw) in SH:
    A = torch.randn(M, Kd, device="cuda").to(bf); B = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(bf)
    args = {}
    if "b" in kw: args["bias"] = bias
    if "r" in kw: args["residual"] = res
    if "d" in kw: args.update(dropout_p=0.1, seed=5)
    if "g" in kw: args.update(act=1)
    ts, outs = {}, []
    for v in (0, 1, 2, 0, 1, 2):
        L.lib.lav_gemm_select(9, v)
        t = bench(lambda: K.gemm(0, A, B, M, N, Kd, **args))
        o = K.gemm(0, A, B, M, N, Kd, **args).float()
        outs.append(o)
        ts.setdefault(v, []).append(t)
    d = max(float((outs[1] - outs[0]).abs().max()), float((outs[2] - outs[0]).abs().max()))
    print(f"{M}x{N}x{Kd} {kw:4s}: without loader waves {min(ts[0]):7.1f} us {2*M*N*Kd/min(ts[0])/1e6:5.0f} TF/s | 192 rows + loaders where the 192-row tile is chosen {min(ts[1]):7.1f} us ({min(ts[1])/min(ts[0]):.3f}) | 192 rows + loaders (everywhere) {min(ts[2]):7.1f} us ({min(ts[2])/min(ts[0]):.3f})  max|d| {d:.3g}", flush=True)
L.lib.lav_gemm_select(9, 0)
