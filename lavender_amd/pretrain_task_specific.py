"""LAVENDER_Pretrain / Agent_Pretrain -- mirror of the reference's main_pretrain_task_specific.py:124-262.

Same fusion-side kernels as the MLM variant; the video-text matching objective is a scalar score per
(video, text) pair from `self.fc` on the first text position, softmaxed over the O = min(B, 4) pairs of each
video (label 0 = the true pair)."""
import math
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

from . import engine as E
from .agent import Agent_Base
from .bert import BertConfigLite, BertOnlyMLMHead, load_hf_into, load_hf_state
from .model import LAVENDER_Base
from .pretrain_mlm import masking, vtm_pairs


class ScoreHead(nn.Sequential):
    """self.fc (main_pretrain_task_specific.py:128-133): Dropout(0.1), Linear(H, 2H), ReLU, Linear(2H, 1).
    State-dict keys fc.1.* / fc.3.* as in the reference; the forward runs the HIP stage."""

    def __init__(self, hidden):
        super().__init__(nn.Dropout(0.1), nn.Linear(hidden, hidden * 2), nn.ReLU(inplace=True), nn.Linear(hidden * 2, 1))
        self._arena_of = None

    def forward(self, x, O=1, temp=1.0):
        """x (n, H) -> (n // O, O) logits = fc(x).view(n // O, O) / temp."""
        if self._arena_of is None:
            raise RuntimeError("ScoreHead is not attached to a parameter arena: call model.cuda() / model.arena() first")
        arena = self._arena_of()
        p = self[0].p if self.training else 0.0
        return E.ScoreHeadFn.apply(arena.anchor, x, self, int(O), 1.0 / float(temp), p)


class LAVENDER_Pretrain(LAVENDER_Base):
    def __init__(self, args, tokzr=None):
        super().__init__(args, tokzr)
        self.patch_size = args.size_patch
        self.fc = ScoreHead(self.hidden_size)
        cfg = BertConfigLite.from_pretrained(args.tokenizer)
        self.fc_mtm = BertOnlyMLMHead(cfg)
        sd = load_hf_state(args.tokenizer, [("cls.", "")])
        if sd:
            load_hf_into(self.fc_mtm, sd, "MLM head (HF checkpoint)")

    def arena(self):
        a = super().arena()
        if self.fc._arena_of is None or self.fc._arena_of() is not a:
            import weakref
            self.fc._arena_of = weakref.ref(a)
        return a

    build_arena = arena

    def forward(self, img, txt, mask, ans_mtm):
        """main_pretrain_task_specific.py:139-177."""
        (_B, _T, _, _H, _W), (_, _X) = img.shape, txt.shape
        _h, _w = _H // self.patch_size, _W // self.patch_size
        _O = min(_B, 4)
        Lv = (1 + _h * _w) * _T

        feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img, txt, mask)
        # the B true pairs (MLM head) and the B*O matching pairs (score head) go through the fusion encoder as one batch
        # (the reference runs it twice, :147 and :165); same arithmetic per sequence, half the launches
        vi, ti, _ = vtm_pairs(_B, _O)
        out, _ = self.go_cross_pairs(feat_img, mask_img, feat_txt, mask_txt, np.concatenate([np.arange(_B), vi]),
                                     np.concatenate([np.arange(_B), ti]))
        out_mtm = self.fc_mtm(out[:_B, Lv:])
        out_vtm = self.fc(out[_B:, Lv, :], O=_O, temp=self.args.temp)
        ans_vtm = torch.zeros(_B, dtype=torch.long, device=txt.device)
        return {"out_vtm": out_vtm, "out_mtm": out_mtm, "ans_vtm": ans_vtm, "ans_mtm": ans_mtm}


class Agent_Pretrain(Agent_Base):
    def __init__(self, args, model):
        super().__init__(args, model)
        self.patch_size = self._unwrapped().patch_size
        self.log = {dataset: defaultdict(list) for dataset in getattr(self.args, "dataset", [])}

    def save_model(self, ep, dataset="init", part=0):
        """main_pretrain_task_specific.py:282-297: one checkpoint per (dataset, part, epoch) of the pre-training loop,
        ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt -- the names the downstream configs of the reference load -- and, as in
        the reference, nothing else (no log.json).  Call on EVERY rank: under ZeRO-1 the masters are gathered collectively first."""
        self._save_state(f"ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt", write_log=False)

    def masking(self, txt, mask, p_mask=0.15):
        """main_pretrain_task_specific.py:186-209 (same procedure as the MLM agent)."""
        return masking(txt, mask, (self.cls_token_id, self.sep_token_id, self.pad_token_id, self.mask_token_id), p_mask)

    def prepare_batch(self, batch):
        if isinstance(batch.get("ans_mtm"), torch.Tensor) and not batch["ans_mtm"].is_cuda:
            batch["_n_mtm"] = int((batch["ans_mtm"] != -1).sum())
        return super().prepare_batch(batch)

    def step(self, batch, is_train=True, sync=True):
        """main_pretrain_task_specific.py:211-248."""
        self._set_mode(is_train)
        img, txt, mask = [batch[key] for key in ["img", "txt", "mask"]]
        ans_mtm = batch["ans_mtm"]
        n_mtm = batch.get("_n_mtm")
        with torch.set_grad_enabled(is_train):
            out = self.forward_step((img, txt, mask, ans_mtm))
            out_mtm, out_vtm, ans_mtm, ans_vtm = out["out_mtm"], out["out_vtm"], out["ans_mtm"], out["ans_vtm"]
            if not is_train:
                pred_mtm, pred_vtm = torch.argmax(out_mtm, dim=-1), torch.argmax(out_vtm, dim=-1)
            ls_mtm = self.loss_func(out_mtm.flatten(0, len(out_mtm.shape) - 2), ans_mtm.flatten(0, len(ans_mtm.shape) - 1), n_mtm)
            ls_vtm = self.loss_func(out_vtm, ans_vtm, ans_vtm.shape[0] if is_train else None)
        if is_train:
            ls = ls_mtm + ls_vtm
            self.backward_step(ls)
            if not sync:
                return {'mtm': ls_mtm.detach(), 'vtm': ls_vtm.detach()}
            return {'mtm': ls_mtm.item(), 'vtm': ls_vtm.item()}
        ac = [float((o == a).sum() / (a != -1).sum()) if (a != -1).sum() > 0 else -1
              for o, a in zip([pred_mtm, pred_vtm], [ans_mtm, ans_vtm])]
        return {'mtm': ac[0], 'vtm': ac[1]}

    def go_dl(self, ep, dl, is_train):
        """main_pretrain_task_specific.py:250-262 (+ masking as in its collate path)."""
        self._set_mode(is_train)
        ret = defaultdict(list)
        for batch in dl:
            batch = dict(batch)
            if "ans_mtm" not in batch:
                batch.update(self.masking(batch["txt"], batch["mask"]))
            batch = self.prepare_batch(batch)
            r = self.step(batch, is_train)
            ret = {k: ret[k] + [l] for k, l in r.items()}
        return {k: self.reduce_mean(float(np.average([v for v in l if not math.isnan(v)]))) for k, l in ret.items()}
