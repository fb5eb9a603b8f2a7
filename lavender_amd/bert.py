"""BERT pieces the reference borrows from HF `transformers` (model.py:100-110,152-165; main_pretrain_mlm.py:46-48),
restated as parameter holders with the SAME state_dict key names, executed by lavender_amd.engine:
  * BertEmbeddings            -> EncTxt.emb_txt
  * BertEncoder / BertLayer   -> LAVENDER_Base.trsfr   (post-LN, erf-GELU, LN eps 1e-12, dropout 0.1/0.1)
  * BertOnlyMLMHead           -> LAVENDER_Pretrain_MLM.fc_mtm
`transformers` is only used (optionally) to read a config / checkpoint directory.
"""
import json
import os

import torch
import torch.nn as nn

from .video_swin import LayerNorm, Linear, _Params


class BertConfigLite:
    """bert-base-uncased defaults (== transformers.BertConfig())."""

    def __init__(self, **kw):
        self.vocab_size = 30522
        self.hidden_size = 768
        self.num_hidden_layers = 12
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.hidden_act = "gelu"
        self.hidden_dropout_prob = 0.1
        self.attention_probs_dropout_prob = 0.1
        self.max_position_embeddings = 512
        self.type_vocab_size = 2
        self.layer_norm_eps = 1e-12
        self.pad_token_id = 0
        self.initializer_range = 0.02
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path):
        """Accepts a dict, a BertConfigLite, a directory holding config.json, or a hub name (defaults)."""
        if isinstance(path, cls):
            return path
        if isinstance(path, dict):
            return cls(**path)
        f = os.path.join(str(path), "config.json")
        if os.path.exists(f):
            with open(f) as fh:
                d = json.load(fh)
            keys = cls().__dict__.keys()
            return cls(**{k: v for k, v in d.items() if k in keys})
        return cls()


class Embedding(_Params):
    def __init__(self, n, dim, std):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, dim) * std)


class BertEmbeddings(_Params):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = Embedding(cfg.vocab_size, cfg.hidden_size, cfg.initializer_range)
        self.position_embeddings = Embedding(cfg.max_position_embeddings, cfg.hidden_size, cfg.initializer_range)
        self.token_type_embeddings = Embedding(cfg.type_vocab_size, cfg.hidden_size, cfg.initializer_range)
        self.LayerNorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout_p = cfg.hidden_dropout_prob
        with torch.no_grad():
            self.word_embeddings.weight[cfg.pad_token_id].zero_()


def _dense(i, o, std):
    m = Linear(i, o)
    with torch.no_grad():
        m.weight.normal_(0, std)
        m.bias.zero_()
    return m


class BertSelfAttention(_Params):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.query, self.key, self.value = (_dense(H, H, cfg.initializer_range) for _ in range(3))


class BertSelfOutput(_Params):
    def __init__(self, cfg, in_dim=None):
        super().__init__()
        self.dense = _dense(in_dim or cfg.hidden_size, cfg.hidden_size, cfg.initializer_range)
        self.LayerNorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class BertAttention(_Params):
    def __init__(self, cfg):
        super().__init__()
        self.self = BertSelfAttention(cfg)
        self.output = BertSelfOutput(cfg)


class BertIntermediate(_Params):
    def __init__(self, cfg):
        super().__init__()
        self.dense = _dense(cfg.hidden_size, cfg.intermediate_size, cfg.initializer_range)


class BertLayer(_Params):
    def __init__(self, cfg):
        super().__init__()
        assert cfg.hidden_size // cfg.num_attention_heads == 64, "the fusion attention kernel is specialised for head_dim 64"
        assert cfg.hidden_act == "gelu"
        self.num_heads = cfg.num_attention_heads
        self.attention = BertAttention(cfg)
        self.intermediate = BertIntermediate(cfg)
        self.output = BertSelfOutput(cfg, in_dim=cfg.intermediate_size)
        self._arena_of = None

    def _arena(self):
        return self._arena_of()


class BertEncoder(_Params):
    """`self.trsfr` of LAVENDER_Base (model.py:160-164)."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.layer = nn.ModuleList([BertLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class BertPredictionHeadTransform(_Params):
    def __init__(self, cfg):
        super().__init__()
        self.dense = _dense(cfg.hidden_size, cfg.hidden_size, cfg.initializer_range)
        self.LayerNorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class BertLMPredictionHead(_Params):
    def __init__(self, cfg):
        super().__init__()
        self.transform = BertPredictionHeadTransform(cfg)
        self.decoder = Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
        with torch.no_grad():
            self.decoder.weight.normal_(0, cfg.initializer_range)
        self.bias = nn.Parameter(torch.zeros(cfg.vocab_size))
        self.decoder.bias = self.bias          # tied, as in HF (state_dict shows both keys)


class BertOnlyMLMHead(nn.Module):
    """`self.fc_mtm` (main_pretrain_mlm.py:46-48).  Callable like the HF head: (..., H) -> (..., vocab) logits."""

    def __init__(self, cfg):
        super().__init__()
        self.predictions = BertLMPredictionHead(cfg)
        self._arena_of = None

    def forward(self, sequence_output, split=None):
        from . import engine as E
        arena = self._arena_of()
        return E.MLMHeadFn.apply(arena.anchor, sequence_output, self, split)


def load_hf_state(path, prefix_map):
    """Read a HF checkpoint directory (model.safetensors / pytorch_model.bin) -> {our key: tensor}.
    prefix_map: list of (hf_prefix, our_prefix).  Returns {} when `path` is not a local directory with weights.
    Mirrors what `from_pretrained` does for the keys the reference keeps (model.py:100-102,152-165, main_pretrain_mlm.py:46-48):
    legacy TF-style LayerNorm names (`LayerNorm.gamma` / `LayerNorm.beta`, as in the stock bert-base-uncased file) become
    `.weight` / `.bias`, and a checkpoint without `cls.predictions.decoder.weight` gets it from the tied word embeddings."""
    if not isinstance(path, str) or not os.path.isdir(path):
        return {}
    sd = None
    st = os.path.join(path, "model.safetensors")
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    elif os.path.exists(pt):
        sd = torch.load(pt, map_location="cpu")
    if sd is None:
        return {}
    sd = {k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias"): v for k, v in sd.items()}
    if "cls.predictions.decoder.weight" not in sd:
        for emb in ("bert.embeddings.word_embeddings.weight", "embeddings.word_embeddings.weight"):
            if emb in sd and any(k.startswith("cls.predictions.") for k in sd):
                sd["cls.predictions.decoder.weight"] = sd[emb]
                break
    if "cls.predictions.decoder.bias" not in sd and "cls.predictions.bias" in sd:
        sd["cls.predictions.decoder.bias"] = sd["cls.predictions.bias"]
    out = {}
    for k, v in sd.items():
        for hp, op in prefix_map:
            if k.startswith(hp):
                out[op + k[len(hp):]] = v
                break
    return out


def load_hf_into(module, sd, what, ignore=("position_ids", "token_type_ids")):
    """module.load_state_dict(sd, strict=False) that REPORTS what did not match instead of dropping it silently."""
    sd = {k: v for k, v in sd.items() if not any(s in k for s in ignore)}
    res = module.load_state_dict(sd, strict=False)
    if res.missing_keys or res.unexpected_keys:
        print(f"[lavender_amd] {what}: missing keys {sorted(res.missing_keys)}; unexpected keys {sorted(res.unexpected_keys)}")
    return res
