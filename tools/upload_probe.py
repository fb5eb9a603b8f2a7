"""GPU probe: does a small host-to-device copy block the launch thread while earlier work is still queued on the stream?
pageable source with non_blocking=True, a fresh torch pinned block per call, and one slice of a long-lived pinned buffer."""
import time
import torch

dev = torch.device("cuda")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def busy(n=40):
    for _ in range(n):
        a @ a


def timed(f):
    torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    r = f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, r


src = torch.arange(45120, dtype=torch.int32)
ring = torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True)
busy(); torch.cuda.synchronize()
for name, f in (("pageable .to(non_blocking=True)", lambda: src.to(dev, non_blocking=True)),
                ("fresh torch.empty(pin_memory=True) + copy_ + .to(non_blocking=True)",
                 lambda: torch.empty(src.shape, dtype=src.dtype, pin_memory=True).copy_(src).to(dev, non_blocking=True)),
                ("slice of a long-lived pinned buffer + copy_ + .to(non_blocking=True)",
                 lambda: ring[:src.numel() * 4].view(torch.int32).copy_(src).to(dev, non_blocking=True))):
    for rep in range(3):
        call, rest, r = timed(f)
        assert int(r[-1]) == 45119
        print(f"{name}: call returned after {call:7.3f} ms; queued work needed another {rest:7.3f} ms", flush=True)
t0 = time.perf_counter()
for _ in range(20):
    p = torch.empty(45120, dtype=torch.int32, pin_memory=True)
    del p
print(f"torch.empty(pin_memory=True) alloc + free, idle GPU: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms each")
