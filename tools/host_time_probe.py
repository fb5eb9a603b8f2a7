"""GPU probe: host (Python + ctypes launch) time per step vs GPU time per step -- how far the launch thread runs ahead."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lavender_amd as LA
from lavender_amd.args import EasyDict
from lavender_amd.dist import set_seed
import bench as BN

B = 32
cfg = dict(num_hidden_layers=12)
args = EasyDict(vis_backbone_size="base", size_img=224, vis_backbone_init="random", kinetics=600, txt_backbone=cfg, txt_backbone_embed_only=True,
                fusion_encoder=cfg, fusion_encoder_rand_init=True, use_checkpoint=False, size_patch=32, size_batch=B, tokenizer=cfg,
                enable_task_token=False, enable_prompt=False, temp=0.05, lr=2e-5, decay=1e-3, max_iter=1000, max_grad_norm=1.0, deepspeed=False,
                vis_backbone_lr_mul=1.0, dataset=["x"], logging_steps=20, path_output="/tmp/x", task="pretrain", seed=88)


class Tok:
    cls_token = "[CLS]"; sep_token = "[SEP]"; pad_token = "[PAD]"; mask_token = "[MASK]"; unk_token = "[UNK]"
    ids = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103, "true": 2995, "false": 6270}
    def convert_tokens_to_ids(self, t): return [self.ids[x] for x in t]


set_seed(88)
m = LA.LAVENDER_Pretrain_MLM(args, Tok()).cuda(); m.arena(); ag = LA.Agent_Pretrain_MLM(args, m)
b = BN.synth_batch(B, 5, 224, 32, 0, "cuda"); torch.manual_seed(88); b.update(ag.masking(b["txt"], b["mask"])); batch = ag.prepare_batch(b)
for _ in range(3):
    ag.step(batch, True, sync=False)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    ag.step(batch, True, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.1f} ms/step, total {1e3*(t2-t0)/n:.1f} ms/step  (OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')})")

# pure host cost: the same steps with every C-ABI call replaced by a no-op (torch allocations / autograd / ctypes marshalling stay)
import lavender_amd._lib as L


class _Null:
    def __getattr__(self, name):
        if name in ("lav_attention_lse_elems", "lav_last_error", "lav_abi_version"):
            return getattr(real, name)
        return lambda *a: 0


real = L.lib
L.lib = _Null()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ag.step(batch, True, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
L.lib = real
print(f"host-only (C-ABI calls stubbed out) {1e3*(t1-t0)/n:.1f} ms/step")

# where the host time goes: cProfile over a few steps (cumulative time, top entries)
if os.environ.get("LAV_HOST_PROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(5):
        ag.step(batch, True, sync=False)
    pr.disable()
    torch.cuda.synchronize()
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
    print(st.getvalue()[:6000])
