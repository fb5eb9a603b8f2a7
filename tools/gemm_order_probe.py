"""GPU probe: tile walk order of the 256x256 NT kernel -- n fastest over all column tiles (0) vs column groups of G tiles
(lav_gemm_select(6, G)): B panels of a group stay resident in the XCD's L2 while the row panels stream."""
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
shapes = [(45120, 3072, 768, "bGp"), (45120, 3072, 768, "gc"), (45120, 3072, 768, ""), (45120, 2304, 768, "b"), (31360, 2048, 512, "bGp"), (31360, 1536, 512, "b"),
          (125440, 1024, 256, "bGp"), (8192, 8192, 8192, ""), (36096, 3072, 3072, "")]
torch.manual_seed(0)
for M, N, Kd, fl in shapes:
    A = torch.randn(M, Kd, device="cuda").to(bf); Bm = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    kw = {}
    if "b" in fl: kw["bias"] = torch.randn(N, device="cuda")
    if "G" in fl: kw["act"] = 1
    if "p" in fl: kw["preact"] = torch.empty(M, N, device="cuda", dtype=bf); kw["preact_is_grad"] = 1
    if "g" in fl: kw["gelu_in"] = torch.rand(M, N, device="cuda").to(bf); kw["gelu_in_is_grad"] = 1
    if "c" in fl: kw["colsum"] = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=bf)
    res, outs = {}, {}
    for rnd in range(3):
        for G in (0, 2, 3, 4, 6):  # 0 = n-fastest
            if G and (N // 256) % G: continue
            L.lib.lav_gemm_select(6, G)
            res.setdefault(G, []).append(run(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out, **kw), 5))
            if rnd == 0: outs[G] = out.float().clone()
    t0 = min(res[0])
    ok = all(torch.equal(outs[0], o) for o in outs.values())
    print(f"{M:6d} {N:5d} {Kd:5d} {fl:4s} n-fastest {t0:7.1f} us | " + " | ".join(f"G={G} {min(v):7.1f} x{t0/min(v):.3f}" for G, v in res.items() if G) + f" | identical {ok}")
L.lib.lav_gemm_select(6, 0)
