"""GPU probe: host cost of one C-ABI launch (ctypes marshalling + hipLaunchKernelGGL) against a torch op, queue kept short."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import hip as K
from lavender_amd import _lib as L
import ctypes as C

x = torch.randn(4096, device="cuda"); y = torch.empty(4096, device="cuda", dtype=torch.bfloat16)
A = torch.randn(256, 64, device="cuda").bfloat16(); B = torch.randn(256, 64, device="cuda").bfloat16(); out = torch.empty(256, 256, device="cuda", dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream


def loop(f, n=3000, every=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        f()
        if i % every == every - 1: torch.cuda.synchronize()      # keep the queue short: measure the call, not back-pressure
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(f"raw ctypes lav_cast_f32_to_bf16: {loop(lambda: L.lib.lav_cast_f32_to_bf16(s, 4096, x.data_ptr(), y.data_ptr())):6.1f} us / call")
print(f"K.gemm 256x256x64 (python wrapper + struct fill): {loop(lambda: K.gemm(0, A, B, 256, 256, 64, out=out)):6.1f} us / call")
print(f"torch mul_ (ATen launch): {loop(lambda: x.mul_(1.0)):6.1f} us / call")
print(f"torch.empty((256,256)): {loop(lambda: torch.empty((256, 256), device='cuda', dtype=torch.bfloat16), every=10**9):6.1f} us / call")
print(f"torch.cuda.current_stream().cuda_stream: {loop(lambda: torch.cuda.current_stream().cuda_stream, every=10**9):6.1f} us / call")
print(f"torch._C._cuda_getCurrentRawStream(current_device()): {loop(lambda: torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()), every=10**9):6.1f} us / call")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    assert torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()) == side.cuda_stream
assert torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()) == torch.cuda.current_stream().cuda_stream
print("raw stream getter follows torch.cuda.stream() contexts")
