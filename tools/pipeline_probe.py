"""GPU probe: input-pipeline throughput on the fixture frames (320x240 4:2:0 JPEGs, 5 per row) replicated to a B x T batch.
Reports host stage (base64 + Huffman, n threads) and device stage separately, against the CPU restatement (Pillow) per frame."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavender_amd import data as D

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
tsv = D.TsvFile(os.path.join(G, "msrvtt_2rows.tsv"), os.path.join(G, "msrvtt_2rows.lineidx"))
frames = [b for r in range(2) for b in tsv.fields(tsv.offset(r))[1:]]
B, T, S = 32, 4, 224
plans = []
for i in range(B * T):
    b = frames[i % len(frames)]
    w, h = D.jpeg_size(b)
    rw, rh = D.resized_size(w, h, S)
    plans.append(D.FramePlan(b, 0, 0, rw, rh, random.randrange(rw - S + 1), random.randrange(rh - S + 1)))
out = torch.empty((B * T, 3, S, S), device="cuda")
for nt in (1, 4, 8, 16, 32):
    dec = D.FrameDecoder(nt)
    for _ in range(3):
        dec.decode(plans, S, out=out)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = 0.0
    gpu = 0.0
    for _ in range(n):
        h0 = time.perf_counter()
        e0.record()
        dec.decode(plans, S, out=out)
        e1.record()
        host += time.perf_counter() - h0
        torch.cuda.synchronize()
        gpu += e0.elapsed_time(e1) / 1e3
    wall = (time.perf_counter() - t0) / n
    print(f"threads {nt:2d}: call {host/n*1e3:7.2f} ms (host stage + enqueue), device stage <= {gpu/n*1e3:6.2f} ms incl. H2D, wall {wall*1e3:7.2f} ms -> "
          f"{B*T/wall:8.0f} frames/s = {B/wall:7.0f} samples/s (T={T})")
    dec.close()
try:
    from oracle import pipeline_ref as PR
    import ctypes
    raw = [ctypes.string_at(p, n) for p, n in frames]
    t0 = time.perf_counter()
    for b in raw:
        PR.img_center_crop(PR.str2img(b), S)
    dt = (time.perf_counter() - t0) / len(raw)
    print(f"CPU restatement (Pillow decode + resize + crop + normalise, 1 thread): {dt*1e3:.2f} ms/frame = {1/dt:.0f} frames/s")
except Exception as e:
    print("cpu baseline skipped:", e)
