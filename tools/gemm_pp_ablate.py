"""GPU probe: ablations of the ping-pong GEMM kernel (which part of a step paces it)."""
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


names = {0: "full", 1: "no refills", 2: "no fragment reads", 3: "no refills, no reads", 4: "no MFMAs", 5: "no refills, no MFMAs", 6: "no reads, no MFMAs"}
for M, N, Kd in ((8192, 8192, 8192), (36096, 3072, 3072), (45120, 3072, 768)):
    A = torch.randn(M, Kd, device="cuda").to(bf)
    Bm = (torch.randn(N, Kd, device="cuda") * 0.05).to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=bf)
    L.lib.lav_gemm_select(0, 0)
    t = min(run(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out, bias=bias), 5) for _ in range(3))
    print(f"{M} {N} {Kd}: 2-phase kernel {t:.1f} us = {2.0*M*N*Kd/t/1e6:.0f} TF")
    L.lib.lav_gemm_select(0, 1)
    for dbg in range(7):
        L.lib.lav_gemm_select(1, dbg)
        t = min(run(lambda: K.gemm(0, A, Bm, M, N, Kd, out=out, bias=bias), 5) for _ in range(3))
        steps = 2 * (Kd // 32) + 1
        tiles = ((M + 255) // 256) * (N // 256)
        rounds = (tiles + 255) // 256
        print(f"   ping-pong [{names[dbg]:24s}] {t:8.1f} us = {2.0*M*N*Kd/t/1e6:5.0f} TF-equivalent; {t/rounds/steps*1e3:6.0f} ns per step")
    L.lib.lav_gemm_select(1, 0)
L.lib.lav_gemm_select(0, 0)
