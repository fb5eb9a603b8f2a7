"""Video question answering posed as masked-LM -- mirrors of the reference's MLM-head QA callers:

    LAVENDER_QAOE_MLM / Agent_QAOE_MLM   main_qaoe_mlm_lsmdc_fib.py:64-125, main_qaoe_mlm.py:92-125   open-ended: the answer token is
                                         predicted at the "[MASK]" of "... answer: [MASK]" (X = 26 in the shipped configs)
    LAVENDER_QAMC_MLM / Agent_QAMC_MLM   main_qamc_mlm.py:109-209                                     multiple choice: question and all
                                         options as ONE text (X = 101), the [MASK] is trained to the option index token

    LAVENDER_RetMC_MLM / Agent_RetMC_MLM main_retmc_mlm.py:70-140                                      retrieval multiple choice: O candidate texts per
                                         video, each scored true / false at its [MASK]; the video prefix is shared by the O sequences
                                         through the pair-index gather (the reference expand()s and flattens feat_img)

The first two run one (video, text) sequence per sample through go_feat -> go_cross -> fc_mtm over the text positions -- the same HIP
kernels as the pre-training path ("they inherit the speed-up because they sit on the same modules").
Task tokens / prompts (enable_task_token / enable_prompt) are outside the built paths and raise in the base class."""
import numpy as np
import torch
import torch.nn as nn

from .agent import Agent_Base
from .bert import BertConfigLite, BertOnlyMLMHead, load_hf_into, load_hf_state
from .model import LAVENDER_Base


class LAVENDER_QAOE_MLM(LAVENDER_Base):
    def __init__(self, args, tokzr=None):
        super().__init__(args, tokzr)
        assert getattr(args, "size_vocab", -1) == -1, "the MLM-head QA model answers with the tokenizer's vocabulary (size_vocab = -1)"
        cfg = BertConfigLite.from_pretrained(args.tokenizer)
        self.fc_mtm = BertOnlyMLMHead(cfg)
        sd = load_hf_state(args.tokenizer, [("cls.", "")])
        if sd:
            load_hf_into(self.fc_mtm, sd, "MLM head (HF checkpoint)")
        self.task_tok2id = {"vtm": 0, "mc": 1, "oe": 2, "cap": 3}
        self.emb_task = nn.Parameter(0.02 * torch.randn(10, self.hidden_size))

    def forward(self, batch):
        """main_qaoe_mlm_lsmdc_fib.py:79-93 / main_qamc_mlm.py:124-140: (B, X, vocab) logits over the text positions and the
        (B, X) labels (-1 everywhere but the [MASK])."""
        img, txt, mask, ans = [batch.get(key) for key in ["img", "txt", "mask", "mask_ans"]]
        (_B, _T, _, _H, _W) = img.shape
        _h, _w = _H // 32, _W // 32
        feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img, txt, mask)
        ans, mask_txt, feat_txt = self.prepro_txt_inputs(ans, mask_txt, feat_txt, task_name=batch.get("task_name"), prompt=batch.get("prompt"))
        out, _ = self.go_cross(feat_img, mask_img, feat_txt, mask_txt)
        out = self.fc_mtm(out[:, (1 + _h * _w) * _T:])
        return out, ans


class LAVENDER_QAMC_MLM(LAVENDER_QAOE_MLM):
    """main_qamc_mlm.py:109-140: the multiple-choice model is the same graph (its task-specific `fc` is deleted in favour of the
    MLM head); only the text it is fed and the agent's evaluation differ."""


class LAVENDER_RetMC_MLM(LAVENDER_QAOE_MLM):
    def forward(self, batch):
        """main_retmc_mlm.py:89-113: txt / mask / mask_ans are (B, O, X); logits (B * O, X, vocab), labels back as (B, O, X)."""
        img, txt, mask, ans = [batch.get(key) for key in ["img", "txt", "mask", "mask_ans"]]
        (_B, _T, _, _H, _W), (_, _O, _X) = img.shape, txt.shape
        _h, _w = _H // 32, _W // 32
        feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img, txt.flatten(0, 1), mask.flatten(0, 1))
        ans = ans.flatten(0, 1)
        ans, mask_txt, feat_txt = self.prepro_txt_inputs(ans, mask_txt, feat_txt, task_name=batch.get("task_name"), prompt=batch.get("prompt"))
        vi, ti = np.repeat(np.arange(_B), _O), np.arange(_B * _O)       # video i with each of its O texts (expand + flatten, :100-103)
        out, _ = self.go_cross_pairs(feat_img, mask_img, feat_txt, mask_txt, vi, ti)
        out = self.fc_mtm(out[:, (1 + _h * _w) * _T:])
        return out, ans.view(_B, _O, -1)


class Agent_RetMC_MLM(Agent_Base):
    def __init__(self, args, model):
        super().__init__(args, model)
        self.log = {'ls_tr': [], 'ac_vl': [], 'ac_ts': []}

    def prepare_batch(self, batch):
        lab = batch.get("mask_ans")
        if isinstance(lab, torch.Tensor) and not lab.is_cuda:
            batch["_n_lab"] = int((lab != -1).sum())
        return super().prepare_batch(batch)

    def step(self, batch, is_train):
        """main_retmc_mlm.py:120-140."""
        self._set_mode(is_train)
        n_lab = batch.pop("_n_lab", None) if isinstance(batch, dict) else None
        with torch.set_grad_enabled(is_train):
            out, ans = self.forward_step(batch)
            if is_train:
                ls = self.loss_func(out.flatten(0, len(out.shape) - 2), ans.flatten(0, 1).flatten(0, 1), n_lab)
                self.backward_step(ls)
                return ls.item()
        _B, _O, _L = ans.shape
        p_true = out[:, :, self.true_token_id].float()
        p_false = out[:, :, self.false_token_id].float()
        out_mtm = p_true / (p_true + p_false)
        ans_mtm = ans.view(_B * _O, _L)
        assert ans_mtm.shape == out_mtm.shape
        out_mtm = torch.argmax(out_mtm[ans_mtm != -1].view(_B, _O), dim=-1)
        ans_idx = (ans_mtm[ans_mtm != -1].view(_B, _O) == self.true_token_id).nonzero()[:, 1]
        return (out_mtm == ans_idx).float().tolist()

    def go_dl(self, ep, dl, is_train):
        self._set_mode(is_train)
        ret = []
        for batch in dl:
            r = self.step(self.prepare_batch(dict(batch)), is_train)
            ret.extend(r) if isinstance(r, list) else ret.append(r)
        return self.reduce_mean(float(np.average(ret)))


class Agent_QAOE_MLM(Agent_Base):
    def __init__(self, args, model):
        super().__init__(args, model)
        self.log = {}

    def step(self, batch, is_train):
        """main_qaoe_mlm_lsmdc_fib.py:100-113."""
        self._set_mode(is_train)
        n_lab = batch.pop("_n_lab", None) if isinstance(batch, dict) else None
        with torch.set_grad_enabled(is_train):
            out, ans = self.forward_step(batch)
            if is_train:
                ls = self.loss_func(out.flatten(0, len(out.shape) - 2), ans.flatten(0, len(ans.shape) - 1), n_lab)
                self.backward_step(ls)
                return {'ls': ls.item()}
        return {'ac_1': self.get_top_k_acc(out, ans, k=1), 'ac_5': self.get_top_k_acc(out, ans, k=5)}

    def get_top_k_acc(self, out, ans, k=5):
        """main_qaoe_mlm_lsmdc_fib.py:115-126."""
        _B = out.shape[0]
        ans_mtm = ans[ans != -1].view(-1, 1)
        out_mtm = out[ans != -1].view(ans_mtm.shape[0], -1).float()
        _, out_mtm_i = torch.topk(out_mtm, k=k, dim=-1)
        ac = (out_mtm_i == ans_mtm).any(dim=-1).float().tolist()
        if len(ac) < _B:
            ac += [0.] * (_B - len(ac))
        return ac

    def prepare_batch(self, batch):
        lab = batch.get("mask_ans")
        if isinstance(lab, torch.Tensor) and not lab.is_cuda:
            batch["_n_lab"] = int((lab != -1).sum())       # counted on the host: the loss kernel needs no device round trip
        return super().prepare_batch(batch)

    def go_dl(self, ep, dl, is_train):
        """main_qaoe_mlm.py:96-125."""
        self._set_mode(is_train)
        ret = {}
        for batch in dl:
            r = self.step(self.prepare_batch(dict(batch)), is_train)
            for k, l in r.items():
                ret.setdefault(k, []).extend(l if isinstance(l, list) else [l])
        return {k: self.reduce_mean(float(np.average(v))) for k, v in ret.items()}


class Agent_QAMC_MLM(Agent_Base):
    def __init__(self, args, model, ans_tok_ids):
        super().__init__(args, model)
        self.ans_tok_ids = list(ans_tok_ids)
        self.log = {'ls_tr': [], 'ac_vl': [], 'ac_ts': []}

    def prepare_batch(self, batch):
        lab = batch.get("mask_ans")
        if isinstance(lab, torch.Tensor) and not lab.is_cuda:
            batch["_n_lab"] = int((lab != -1).sum())
        return super().prepare_batch(batch)

    def step(self, batch, is_train):
        """main_qamc_mlm.py:148-170."""
        self._set_mode(is_train)
        n_lab = batch.pop("_n_lab", None) if isinstance(batch, dict) else None
        with torch.set_grad_enabled(is_train):
            out, ans = self.forward_step(batch)
            if is_train:
                ls = self.loss_func(out.flatten(0, len(out.shape) - 2), ans.flatten(0, len(ans.shape) - 1), n_lab)
                self.backward_step(ls)
                return ls.item()
        _B = ans.shape[0]
        p_all = out[:, :, self.ans_tok_ids].float()
        out_mtm = p_all[ans != -1]
        out_mtm = (out_mtm / out_mtm.sum(dim=-1).view(_B, 1)).view(_B, -1)
        return (torch.argmax(out_mtm, dim=-1) == batch["ans_idx"].to(out_mtm.device)).float().tolist()

    def go_dl(self, ep, dl, is_train):
        """main_qamc_mlm.py:172-207."""
        self._set_mode(is_train)
        ret = []
        for batch in dl:
            r = self.step(self.prepare_batch(dict(batch)), is_train)
            ret.extend(r) if isinstance(r, list) else ret.append(r)
        return self.reduce_mean(float(np.average(ret)))
