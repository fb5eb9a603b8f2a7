"""LAVENDER_Pretrain_MLM / Agent_Pretrain_MLM -- mirror of the reference's main_pretrain_mlm.py:42-232."""
import math
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

from . import engine as E
from .agent import Agent_Base
from .bert import BertConfigLite, BertOnlyMLMHead, load_hf_into, load_hf_state
from .model import LAVENDER_Base


import os as _os
_MERGE_PASSES = _os.environ.get("LAV_MERGE_PASSES", "1") != "0"


def fp32_validation(args):
    """args.validate_fp32 = True or LAV_FP32=1 selects the fp32-I/O validation forward (lavender_amd/validate.py)."""
    return bool(getattr(args, "validate_fp32", False)) or _os.environ.get("LAV_FP32", "0") == "1"


def vtm_pairs(B, O):
    """Pair list of main_pretrain_mlm.py:74-106: per sample i -> (i,i,true) then O-1 negatives drawn with ONE
    np.random.permutation([j != i]) -- same RNG call sequence, hence the same negatives, as the reference."""
    vi, ti, tr = [], [], []
    for i in range(B):
        vi.append(i); ti.append(i); tr.append(True)
        neg = np.random.permutation([j for j in range(B) if j != i])
        for j in range(O - 1):
            vi.append(i); ti.append(int(neg[j])); tr.append(False)
    return np.asarray(vi), np.asarray(ti), np.asarray(tr)


def masking(txt, mask, ids, p_mask=0.15):
    """Agent_Pretrain_MLM.masking (main_pretrain_mlm.py:178-200), vectorised per row but consuming the torch CPU
    RNG exactly like the reference (one T.rand(X) per row, in row order) => bit-identical outputs."""
    cls_id, sep_id, pad_id, mask_id = ids
    _B, _X = txt.shape
    spc = (txt == cls_id) | (txt == sep_id) | (txt == pad_id) | (txt == mask_id)
    ans_mtm = torch.ones(txt.shape).long() * -1
    if p_mask <= 0:
        return {"txt": txt, "mask": mask, "ans_mtm": ans_mtm}
    for i in range(_B):
        hit = torch.logical_and(torch.logical_not(spc[i]), torch.rand(_X) < p_mask)
        ans_mtm[i][hit] = txt[i][hit]
        txt[i][hit] = mask_id
    return {"txt": txt, "mask": mask, "ans_mtm": ans_mtm}


class LAVENDER_Pretrain_MLM(LAVENDER_Base):
    def __init__(self, args, tokzr=None):
        super().__init__(args, tokzr)
        self.patch_size = args.size_patch
        cfg = BertConfigLite.from_pretrained(args.tokenizer)
        self.fc_mtm = BertOnlyMLMHead(cfg)
        sd = load_hf_state(args.tokenizer, [("cls.", "")])
        if sd:
            load_hf_into(self.fc_mtm, sd, "MLM head (HF checkpoint)")
        self.vtm_batch = min(self.args.size_batch, 4)
        self.task_tok2id = {"vtm": 0, "mc": 1, "oe": 2, "cap": 3}
        self.emb_task = nn.Parameter(0.02 * torch.randn(10, self.hidden_size))

    def forward(self, batch):
        """main_pretrain_mlm.py:55-119.  Same outputs; the B*O python loop of slices + T.cat is replaced by an
        index list (same numpy RNG draws) consumed by one gather kernel."""
        if fp32_validation(self.args):
            # tier-T2 validation mode (north_star "MLM logits within 1e-3 of reference"): fp32 activations end to end,
            # forward only, eval arithmetic -- lavender_amd/validate.py
            if self.training:
                raise RuntimeError("the fp32 validation mode is forward-only (eval arithmetic): call model.eval() first")
            from . import validate
            with torch.no_grad():
                return validate.pretrain_mlm_forward(self, batch)
        batch = defaultdict(lambda: None, batch)
        img, txt, mask = [batch[key] for key in ["img", "txt", "mask"]]
        vt_mask, ans_mtm = batch["vt_mask"], batch["ans_mtm"]
        (_B, _T, _, _H, _W), (_, _X) = img.shape, txt.shape
        _h, _w = _H // self.patch_size, _W // self.patch_size
        _O = min(_B, self.vtm_batch)
        Lv = (1 + _h * _w) * _T

        feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img, txt, mask, vt_mask=vt_mask)
        # The reference runs the fusion encoder + head twice (main_pretrain_mlm.py:67-69 on the B true pairs, :111-115 on
        # the B*O matching pairs).  Here both groups go through as ONE batch of B + B*O sequences (same arithmetic per
        # sequence, half the kernel launches, no under-filled 32-sequence GEMMs); the head returns the two logit tensors.
        vi, ti, tr = vtm_pairs(_B, _O)
        lab_vtm = torch.where(torch.from_numpy(tr), self.true_token_id, self.false_token_id)
        if not _MERGE_PASSES:
            out, _ = self.go_cross(feat_img, mask_img, feat_txt, mask_txt)
            out_mtm = self.fc_mtm(out[:, Lv:])
            out, _ = self.go_cross_pairs(feat_img, mask_img, feat_txt, mask_txt, vi, ti)
            out_vtm = self.fc_mtm(out[:, Lv:])
            ans_vtm = torch.full((_B * _O, _X), -1, dtype=torch.long)
            ans_vtm[:, -1] = lab_vtm
            return {"out_vtm": out_vtm, "out_mtm": out_mtm, "ans_vtm": ans_vtm.to(txt.device, non_blocking=True), "ans_mtm": ans_mtm}
        vi_all = np.concatenate([np.arange(_B), vi])
        ti_all = np.concatenate([np.arange(_B), ti])
        out, _ = self.go_cross_pairs(feat_img, mask_img, feat_txt, mask_txt, vi_all, ti_all)
        _L, _n = Lv + _X, _B + _B * _O
        # opt-in (args.loss_aware_head, training only): head + loss on the supervised positions only -- same loss and
        # gradients, ~8x less vocabulary GEMM and no (n, X, vocab) logits; the default keeps the reference's full outputs
        lab_cpu = batch["_ans_mtm_cpu"]
        aware = self.training and bool(getattr(self.args, "loss_aware_head", False)) and lab_cpu is not None
        rows_b, rows_x = (np.nonzero(lab_cpu.numpy() != -1) if aware else (None, None))
        aware = aware and len(rows_b) > 0
        if aware:
            rows = np.concatenate([rows_b * _L + Lv + rows_x, (_B + np.arange(_B * _O)) * _L + _L - 1])
            logits = self.fc_mtm(E.RowGatherFn.apply(out.reshape(_n * _L, -1), rows))
            out_mtm, out_vtm = logits[:len(rows_b)], logits[len(rows_b):]
            ans_mtm = lab_cpu[rows_b, rows_x].to(txt.device, non_blocking=True)
            ans_vtm = lab_vtm.to(txt.device, non_blocking=True)
        else:
            out_mtm, out_vtm = self.fc_mtm(out[:, Lv:], split=_B)
            ans_vtm = torch.full((_B * _O, _X), -1, dtype=torch.long)
            ans_vtm[:, -1] = lab_vtm
            ans_vtm = ans_vtm.to(txt.device, non_blocking=True)
        return {"out_vtm": out_vtm, "out_mtm": out_mtm, "ans_vtm": ans_vtm, "ans_mtm": ans_mtm}


class Agent_Pretrain_MLM(Agent_Base):
    def __init__(self, args, model):
        super().__init__(args, model)
        self.patch_size = self._unwrapped().patch_size
        self.log = {dataset: defaultdict(list) for dataset in getattr(self.args, "dataset", [])}

    def save_model(self, ep, dataset="init", part=0):
        """main_pretrain_task_specific.py:282-297: one checkpoint per (dataset, part, epoch) of the pre-training loop,
        ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt -- the names the downstream configs of the reference load -- and, as in
        the reference, nothing else (no log.json).  Call on EVERY rank: under ZeRO-1 the masters are gathered collectively first."""
        self._save_state(f"ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt", write_log=False)

    def cal_vtm_loss(self, txt, out, ans, is_train=True, count=None):
        if is_train:
            return self.loss_func(out.flatten(0, len(out.shape) - 2), ans.flatten(0, len(ans.shape) - 1), count)
        _B, _ = txt.shape
        p_true = out[:, :, self.true_token_id].float()
        p_false = out[:, :, self.false_token_id].float()
        out_vtm = p_true / (p_true + p_false)
        out_vtm = out_vtm[ans != -1].view(_B, -1)
        ans_vtm = ans[ans != -1].view(_B, -1)
        out_vtm = torch.argmax(out_vtm, dim=-1)
        ans_idx = (ans_vtm == self.true_token_id).nonzero()[:, 1]
        return float((out_vtm == ans_idx).float().sum() / _B)

    def step(self, batch, is_train=True, sync=True):
        """main_pretrain_mlm.py:145-176.  sync=False returns the two losses as device scalars (no host round trip)."""
        self._set_mode(is_train)
        n_mtm = batch.get("_n_mtm") if isinstance(batch, dict) else None
        with torch.set_grad_enabled(is_train):
            out = self.forward_step(batch)
            out_mtm, out_vtm, ans_mtm, ans_vtm = out["out_mtm"], out["out_vtm"], out["ans_mtm"], out["ans_vtm"]
            ls_mtm = self.loss_func(out_mtm.flatten(0, len(out_mtm.shape) - 2), ans_mtm.flatten(0, len(ans_mtm.shape) - 1), n_mtm)
            ls_vtm = self.cal_vtm_loss(batch["txt"], out_vtm, ans_vtm, is_train, count=ans_vtm.shape[0] if is_train else None)
        if is_train:
            ls = ls_mtm + ls_vtm
            self.backward_step(ls)
            if not sync:
                return {'mtm': ls_mtm.detach(), 'vtm': ls_vtm.detach()}
            return {'mtm': ls_mtm.item(), 'vtm': ls_vtm.item()}
        pred = torch.argmax(out_mtm, dim=-1)
        n = (ans_mtm != -1).sum()
        ac_mtm = float((pred == ans_mtm).sum() / n) if n > 0 else -1
        return {'mtm': ac_mtm, 'vtm': ls_vtm}

    def masking(self, txt, mask, p_mask=0.15):
        return masking(txt, mask, (self.cls_token_id, self.sep_token_id, self.pad_token_id, self.mask_token_id), p_mask)

    def prepare_batch(self, batch):
        # the labels are still on the host here: count them once so the loss kernel needs no device round trip
        lab = batch.get("ans_mtm")
        if isinstance(lab, torch.Tensor) and not lab.is_cuda:
            batch["_n_mtm"] = int((lab != -1).sum())
        out = super().prepare_batch(batch)
        if isinstance(lab, torch.Tensor) and not lab.is_cuda:
            out["_ans_mtm_cpu"] = lab                      # host copy of the labels: the loss-aware head indexes with it (no device sync)
        return out

    def go_dl(self, ep, dl, is_train):
        """main_pretrain_mlm.py:202-232."""
        self._set_mode(is_train)
        ret = defaultdict(list)
        idx = 0
        for idx, batch in enumerate(dl):
            batch = dict(batch)
            masked = self.masking(batch["txt"], batch["mask"])
            batch.update(masked)
            batch = self.prepare_batch(batch)
            r = self.step(batch, is_train)
            ret = {k: ret[k] + [l] for k, l in r.items()}
        return {k: self.reduce_mean(float(np.average([v for v in l if not math.isnan(v)]))) for k, l in ret.items()}
