"""GPU check: weight gradients of the full-size model (Swin-B + 12L, B=8) with the dW side stream on vs off -- bitwise equal."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lavender_amd as LA, lavender_amd.engine as E
from lavender_amd import hip as K
from lavender_amd.args import EasyDict
from lavender_amd.dist import set_seed
import bench as BN

B = 8
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
m = LA.LAVENDER_Pretrain_MLM(args, Tok()).cuda(); ar = m.arena(); ag = LA.Agent_Pretrain_MLM(args, m)
b = BN.synth_batch(B, 5, 224, 32, 0, "cuda"); torch.manual_seed(88); b.update(ag.masking(b["txt"], b["mask"])); batch = ag.prepare_batch(b)
wn = [n for n, p in m.named_parameters() if p.dim() == 2 and n.endswith("weight") and "embeddings" not in n]


def run(side):
    E._DW_SIDE = side; K.reseed(99); np.random.seed(1); m.train(); ar.zero_grad()
    out = m(batch)
    ls = ag.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten(), batch["_n_mtm"]) + ag.loss_func(out["out_vtm"].flatten(0, 1), out["ans_vtm"].flatten(), B * 4)
    ls.backward(); torch.cuda.synchronize()
    return {n: ar.params[n].grad.clone() for n in wn}


ref = run(False)
for rep in range(5):
    got = run(True)
    bad = [n for n in wn if not torch.equal(got[n], ref[n])]
    print("rep", rep, "mismatching weight gradients:", len(bad), bad[:3])
