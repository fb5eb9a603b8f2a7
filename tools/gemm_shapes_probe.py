"""GPU probe: every distinct GEMM call of one pretrain step (cfg2) with its measured time -> where the GEMM time goes.
   python tools/gemm_shapes_probe.py [B]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
for _ in range(3):
    agent.step(batch, True, sync=False)
torch.cuda.synchronize()
rec = []
orig = K.gemm


def timed(layout, A, Bm, M, N, Kd, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(layout, A, Bm, M, N, Kd, **kw)
    e1.record()
    flags = "".join(c for c, k in (("b", "bias"), ("G", "act"), ("p", "preact"), ("g", "gelu_in"), ("d", "dropout_p"), ("s", "row_scale"),
                                   ("r", "residual"), ("c", "colsum"), ("k", "k_keep"), ("R", "rowsum_a"))
                    if kw.get(k) is not None and not (isinstance(kw.get(k), (int, float)) and not kw.get(k)))
    rec.append(((layout, M, N, Kd, flags, kw.get("splits", 1)), 2.0 * M * N * Kd, e0, e1))
    return r


K.gemm = timed
agent.step(batch, True, sync=False)
torch.cuda.synchronize()
K.gemm = orig
by = {}
for key, fl, e0, e1 in rec:
    d = by.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += fl; d[2] += e0.elapsed_time(e1) * 1e-3
tot = sum(v[2] for v in by.values())
print(f"total GEMM time {tot*1e3:.1f} ms, {sum(v[1] for v in by.values())/1e12:.2f} TFLOP, {sum(v[0] for v in by.values())} launches")
print("layout      M      N      K  flags  splits calls   ms_total  us/call   TF/s   %")
for key, (n, fl, tm) in sorted(by.items(), key=lambda kv: -kv[1][2]):
    L, M, N, Kd, flags, sp = key
    print(f"{'NT NN TN'.split()[L]:>4} {M:8d} {N:6d} {Kd:6d}  {flags:6s} {sp:4d} {n:5d} {tm*1e3:9.2f} {tm/n*1e6:8.1f} {fl/tm/1e12:6.0f} {100*tm/tot:5.1f}")
