"""LAVENDER_Retrieval_MLM / Agent_Retrieval_MLM -- mirror of the reference's main_retrieval_mlm.py:30-118.

Text-to-video retrieval posed as masked-LM: every one of the B x B (video i, text j) pairs is run through the
fusion encoder and the MLM head; the appended [MASK] position is trained to say "true" when vid[i] == vid[j]."""
import numpy as np
import torch
import torch.nn as nn

from .agent import Agent_Base
from .bert import BertConfigLite, BertOnlyMLMHead, load_hf_into, load_hf_state
from .model import LAVENDER_Base


def all_pairs(B):
    """Pair order of the double loop main_retrieval_mlm.py:63-64: i (video) outer, j (text) inner."""
    vi = np.repeat(np.arange(B), B)
    ti = np.tile(np.arange(B), B)
    return vi, ti


class LAVENDER_Retrieval_MLM(LAVENDER_Base):
    def __init__(self, args, tokzr=None):
        super().__init__(args, tokzr)
        cfg = BertConfigLite.from_pretrained(args.tokenizer)
        self.fc_mtm = BertOnlyMLMHead(cfg)
        sd = load_hf_state(args.tokenizer, [("cls.", "")])
        if sd:
            load_hf_into(self.fc_mtm, sd, "MLM head (HF checkpoint)")
        self.task_tok2id = {"vtm": 0, "mc": 1, "oe": 2, "cap": 3}
        self.emb_task = nn.Parameter(0.02 * torch.randn(10, self.hidden_size))

    def forward(self, batch):
        """main_retrieval_mlm.py:50-91: same (B*B, X, vocab) logits and (B*B, X) labels; the B^2 python list of
        slices + T.cat is one gather kernel over the pair index list."""
        img, txt, mask, vid = [batch.get(key) for key in ["img", "txt", "mask", "vid"]]
        (_B, _T, _, _H, _W) = img.shape
        _h, _w = _H // 32, _W // 32
        _X = txt.shape[1]
        Lv = (1 + _h * _w) * _T
        feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img, txt, mask)
        if batch.get("task_name") is not None or batch.get("prompt") is not None:
            self.prepro_txt_inputs(txt[0], mask_txt[0], feat_txt[0], task_name=batch.get("task_name"), prompt=batch.get("prompt"))
        vi, ti = all_pairs(_B)
        out, _ = self.go_cross_pairs(feat_img, mask_img, feat_txt, mask_txt, vi, ti)
        out = self.fc_mtm(out[:, Lv:])
        vid = list(vid.tolist()) if isinstance(vid, torch.Tensor) else list(vid)
        same = torch.tensor([vid[i] == vid[j] for i, j in zip(vi, ti)])
        mtm_ans = torch.full((_B * _B, _X), -1, dtype=torch.long)
        mtm_ans[:, -1] = torch.where(same, self.true_token_id, self.false_token_id)
        return out, mtm_ans.to(txt.device, non_blocking=True)


class Agent_Retrieval_MLM(Agent_Base):
    def __init__(self, args, model):
        super().__init__(args, model)
        self.log = {'ls_tr': [], 'ac_vl': [], 'ac_ts': []}

    def step(self, batch, is_train, sync=True):
        """main_retrieval_mlm.py:99-118.  sync=False returns the training loss as a device scalar (no host round trip per step)."""
        self._set_mode(is_train)
        with torch.set_grad_enabled(is_train):
            out, ans = self.forward_step(batch)
            if is_train:
                out = out.flatten(0, len(out.shape) - 2)
                ans = ans.flatten(0, len(ans.shape) - 1)
                ls = self.loss_func(out, ans, out.shape[0] // batch["txt"].shape[1])
                self.backward_step(ls)
                return ls.item() if sync else ls.detach()
        _B = len(batch["vid"])
        p_true = out[:, :, self.true_token_id].float()
        p_false = out[:, :, self.false_token_id].float()
        out_mtm = p_true / (p_true + p_false)
        out_mtm = out_mtm[ans != -1].view(_B, _B)
        ans_mtm = ans[ans != -1].view(_B, _B)
        out_mtm = torch.argmax(out_mtm, dim=-1)
        ans_idx = (ans_mtm == self.true_token_id).nonzero()[:, 1]
        return (out_mtm == ans_idx).float().tolist()

    def go_dl(self, ep, dl, is_train):
        """main_retrieval_mlm.py:120-150."""
        self._set_mode(is_train)
        ret = []
        for batch in dl:
            batch = self.prepare_batch(dict(batch))
            r = self.step(batch, is_train)
            ret.extend(r) if isinstance(r, list) else ret.append(r)
        return self.reduce_mean(float(np.average(ret)))


class LAVENDER_RetrievalMlmEval(LAVENDER_Retrieval_MLM):
    """Two-phase retrieval inference of the reference (eval_retrieval_mlm.py:10-47): 'feat' encodes every video once
    (mean over its clips) and every caption once; 'cross' runs the fusion encoder + MLM head on (video, text) pairs
    of cached features.  Forward only."""

    def forward(self, typ, batch):
        from . import hip as K
        if typ == 'feat':
            img, txt, mask = [batch.get(key) for key in ["img", "txt", "mask"]]
            _B, _Clips, _T, _C, _H, _W = img.shape
            feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img.view(-1, _T, _C, _H, _W), txt, mask)
            Lv, Hd = feat_img.shape[1], feat_img.shape[2]
            if _Clips > 1:                                # mean over the clips of a video: gather-sum kernel + 1/Clips scale kernel
                rows = feat_img.reshape(_B * _Clips * Lv, Hd)
                idx = (torch.arange(_B * Lv).view(_B, 1, Lv) // Lv * (_Clips * Lv) + torch.arange(Lv).view(1, 1, Lv)
                       + torch.arange(_Clips).view(1, _Clips, 1) * Lv)                      # (B, Clips, Lv) source rows
                lst = idx.permute(0, 2, 1).reshape(-1).to(torch.int32).to(rows.device)      # per output row: its Clips sources
                start = (torch.arange(_B * Lv + 1, dtype=torch.int32) * _Clips).to(rows.device)
                summed = K.gather_sum_rows(rows, start, lst, _B * Lv, Hd)
                scale = torch.full((1,), 1.0 / _Clips, dtype=torch.float32, device=rows.device)
                mean = torch.empty_like(summed)
                K.scale_mask_rows(summed, _B * Lv, Hd, out=mean, row_scale=scale, rows_per_group=_B * Lv)
                feat_img = mean.view(_B, Lv, Hd)
            else:
                feat_img = feat_img.view(_B, Lv, Hd)
            mask_img = mask_img.view(_B, _Clips, -1)[:, 0, :]
            txt, mask_txt, feat_txt = self.prepro_txt_inputs(txt, mask_txt, feat_txt, task_name=batch.get("task_name"),
                                                             prompt=None)
            return feat_img, mask_img, feat_txt, mask_txt, txt
        elif typ == 'cross':
            feat_img, mask_img, feat_txt, mask_txt = [batch.get(key) for key in ["feat_img", "mask_img", "feat_txt", "mask_txt"]]
            out, _ = self.go_cross(feat_img, mask_img, feat_txt, mask_txt)
            return self.fc_mtm(out[:, feat_img.shape[1]:]), batch.get("txt")
        raise ValueError(f"typ must be 'feat' or 'cross', got {typ!r}")
