"""LAVENDER_Captioning -- the training forward of the reference's model_for_captioning.py:40-95 on the HIP engine.

Captioning is MLM under a sequence-to-sequence attention mask (model.py:208-218): every position sees the video tokens,
a text position additionally sees the text up to itself, video positions see no text.  The mask is not materialised as a
(B, L, L) tensor on the device path: the fusion attention kernels take (key mask, number of prefix keys)
(lav_attn_desc.causal_from).  Generation (beam / greedy decode, model_for_captioning.py:97-500) is host-side search on
top of the same forward and is outside the built path."""
from collections import defaultdict

import torch
import torch.nn as nn

from .bert import BertConfigLite, BertOnlyMLMHead, load_hf_into, load_hf_state
from .model import LAVENDER_Base


class LAVENDER_Captioning(LAVENDER_Base):
    def __init__(self, args, tokzr=None, is_decoder=True):
        super().__init__(args, tokzr)
        self.config.is_decoder = is_decoder
        cfg = BertConfigLite.from_pretrained(args.tokenizer)
        self.fc_mtm = BertOnlyMLMHead(cfg)
        sd = load_hf_state(args.tokenizer, [("cls.", "")])
        if sd:
            load_hf_into(self.fc_mtm, sd, "MLM head (HF checkpoint)")
        self.task_tok2id = {"vtm": 0, "mc": 1, "oe": 2, "cap": 3}
        self.emb_task = nn.Parameter(0.02 * torch.randn(10, self.hidden_size))
        self.cap_prompt_txt_L = 0

    def forward(self, batch, is_decode=False):
        batch = defaultdict(lambda: None, batch)
        if is_decode:
            raise NotImplementedError("caption generation (model_for_captioning.py:97-500) is host-side search outside the built path")
        return self.encode_forward(batch)

    def encode_forward(self, batch):
        """model_for_captioning.py:54-95, branch input_ids is None, prompt / task token off."""
        if batch["input_ids"] is not None:
            raise NotImplementedError("the incremental-decoding branch (model_for_captioning.py:96-130) belongs to generation")
        img, txt, mask, ans_mtm = batch["img"], batch["txt"], batch["mask"], batch["ans_mtm"]
        (_B, _T, _, _H, _W) = img.shape
        _h, _w = _H // 32, _W // 32
        feat_img, mask_img, feat_txt, mask_txt = self.go_feat(img, txt, mask)
        ans_mtm, _, feat_txt = self.prepro_txt_inputs(ans_mtm, mask_txt, feat_txt, task_name="cap", prompt=batch["prompt"])
        out, _ = self.go_cross(feat_img, mask_img, feat_txt, mask_txt, attn_mask_type=batch["attn_mask_type"] or "seq2seq")
        out = self.fc_mtm(out[:, (1 + _h * _w) * _T:])
        return {"out": out, "ans": ans_mtm}
