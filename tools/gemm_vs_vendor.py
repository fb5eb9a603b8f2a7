"""GPU probe (measurement only -- nothing vendor-made is on the product path): every distinct GEMM shape of one cfg2
pretrain step, timed in isolation with lav_gemm_bf16 (plain bf16 store, no epilogue; weight gradients with the step's
split-K and fp32 accumulate) against torch.matmul (hipBLASLt / rocBLAS) on the same operands, interleaved rounds in one
process, median of the rounds.  Tells per shape whether the tile design or the chip is the limit.
   python tools/gemm_vs_vendor.py [out.md] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lavender_amd as LA
from lavender_amd import hip as K
from lavender_amd.args import EasyDict
from lavender_amd.dist import set_seed
import bench as BN
import lavender_amd.engine as _E
_E.STAGE_C = False          # the K.gemm hook below has to see every launch: per-kernel host path (same kernels as the stage-level C entries)

out_md = sys.argv[1] if len(sys.argv) > 1 else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = dict(num_hidden_layers=12)
args = EasyDict(vis_backbone_size="base", size_img=224, vis_backbone_init="random", kinetics=600, txt_backbone=cfg,
                txt_backbone_embed_only=True, fusion_encoder=cfg, fusion_encoder_rand_init=True, use_checkpoint=False,
                size_patch=32, size_batch=B, tokenizer=cfg, enable_task_token=False, enable_prompt=False, temp=0.05,
                lr=2e-5, decay=1e-3, max_iter=10000, max_grad_norm=1.0, deepspeed=False, vis_backbone_lr_mul=1.0,
                dataset=["synthetic"], logging_steps=20, path_output="/tmp/lav_bench", task="pretrain", seed=88)


class Tok:
    cls_token = "[CLS]"; sep_token = "[SEP]"; pad_token = "[PAD]"; mask_token = "[MASK]"; unk_token = "[UNK]"
    ids = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103, "true": 2995, "false": 6270}

    def convert_tokens_to_ids(self, toks):
        return [self.ids[t] for t in toks]


set_seed(88)
model = LA.LAVENDER_Pretrain_MLM(args, Tok()).cuda()
model.arena()
agent = LA.Agent_Pretrain_MLM(args, model)
b = BN.synth_batch(B, 5, 224, 32, 0, "cuda")
torch.manual_seed(88)
b.update(agent.masking(b["txt"], b["mask"]))
batch = agent.prepare_batch(b)
np.random.seed(88)
agent.step(batch, True, sync=False)
torch.cuda.synchronize()
shapes = {}
orig = K.gemm


padw = set()                                       # shapes the step launches with c_pad_writable (ragged N, padding columns of the output writable)


def rec(layout, A, Bm, M, N, Kd, **kw):
    key = (layout, M, N, Kd, kw.get("splits", 1))
    shapes[key] = shapes.get(key, 0) + 1
    if kw.get("c_pad_writable"):
        padw.add(key)
    return orig(layout, A, Bm, M, N, Kd, **kw)


K.gemm = rec
agent.step(batch, True, sync=False)
torch.cuda.synchronize()
K.gemm = orig
del agent, model, batch, b
torch.cuda.empty_cache()

ROUNDS, INNER = 7, 4
bf = torch.bfloat16


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(INNER):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / INNER * 1e3          # us


rows = []
g = torch.Generator(device="cuda").manual_seed(1)
for (layout, M, N, Kd, sp), calls in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][1] * kv[0][2] * kv[0][3]):
    ash = (M, Kd) if layout != 2 else (Kd, M)
    bsh = (N, Kd) if layout == 0 else (Kd, N)
    pad8 = lambda sh: (sh[0], (sh[1] + 7) // 8 * 8)           # row strides are multiples of 8 elements in the step (16-byte rows)
    A = (torch.rand(pad8(ash), device="cuda", generator=g) * 2 - 1).to(bf)[:, :ash[1]]
    Bm = (torch.rand(pad8(bsh), device="cuda", generator=g) * 2 - 1).to(bf)[:, :bsh[1]]
    if layout == 2:
        Cf = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        lav = lambda: orig(2, A, Bm, M, N, Kd, out=Cf, accumulate=True, splits=sp)
        ven = lambda: torch.matmul(A.t(), Bm)
    else:
        Cb = torch.empty((M, (N + 7) // 8 * 8), dtype=bf, device="cuda")[:, :N]
        lav = lambda: orig(layout, A, Bm, M, N, Kd, out=Cb, splits=sp, c_pad_writable=(layout, M, N, Kd, sp) in padw)   # as the step launches it
        ven = (lambda: torch.matmul(A, Bm.t())) if layout == 0 else (lambda: torch.matmul(A, Bm))
    tl, tv = [], []
    for _ in range(ROUNDS):
        tl.append(timeit(lav)); tv.append(timeit(ven))
    tl, tv = float(np.median(tl)), float(np.median(tv))
    fl = 2.0 * M * N * Kd
    rows.append((layout, M, N, Kd, sp, calls, tl, tv, fl / tl / 1e6, fl / tv / 1e6))
    del A, Bm

hdr = "| layout | M | N | K | splits | calls/step | lav us | vendor us | lav TF/s | vendor TF/s | lav/vendor time |\n|---|---|---|---|---|---|---|---|---|---|---|"
lines = [hdr]
wl = wv = 0.0
for (layout, M, N, Kd, sp, calls, tl, tv, fl_, fv_) in rows:
    lines.append(f"| {'NT NN TN'.split()[layout]} | {M} | {N} | {Kd} | {sp} | {calls} | {tl:.1f} | {tv:.1f} | {fl_:.0f} | {fv_:.0f} | {tl / tv:.2f} |")
    wl += calls * tl; wv += calls * tv
within = sum(1 for r in rows if r[6] <= 1.10 * r[7])
txt = (f"GEMM shapes of one cfg2 step (B = {B}), isolated, plain stores (no fused epilogue on either side); uniform [-1, 1) operands; "
       f"median of {ROUNDS} interleaved rounds x {INNER} launches.\n\n" + "\n".join(lines) +
       f"\n\nsum over the step's launches: lav {wl / 1e3:.2f} ms, vendor {wv / 1e3:.2f} ms (ratio {wl / wv:.3f}); "
       f"lav within 10 % of (or faster than) the vendor library on {within} of {len(rows)} shapes\n")
print(txt)
if out_md:
    with open(out_md, "w") as f:
        f.write(txt)
