"""Golden-vector generator (CONTAINER ONLY -- needs /root/reference; never runs on the GPU box).

Imports the real microsoft/LAVENDER reference with stub modules for its missing
third-party deps (recipe: SURVEY.md section 8c / Appendix E), fills every parameter
deterministically from its state_dict key (oracle.lavender_ref.fill_tensor), runs the
reference on seeded inputs and writes small .npz fixtures next to this file.
The fixtures are data (inputs are re-derived from seeds; outputs are sub-sampled
reference outputs) -- no reference source is stored.

    python tests/golden/make_goldens.py
"""
import sys
sys.dont_write_bytecode = True
import hashlib
import importlib
import os
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import transformers
from transformers import (BertConfig, BertForMaskedLM, AutoModel, AutoModelForMaskedLM,  # noqa: F401
                          AutoConfig, AutoTokenizer, RobertaForMaskedLM)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import lavender_ref as R  # noqa: E402

REF = "/root/reference"
TMP = "/tmp/lav_golden"

BERT_CFGS = {
    "micro": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512, vocab_size=8192),
    "b2l": dict(num_hidden_layers=2),
}


def hf_dir(name):
    d = f"{TMP}/{name}"
    if not os.path.exists(d):
        BertForMaskedLM(BertConfig(**BERT_CFGS[name])).save_pretrained(d)
    return d


def import_reference():
    for n in BERT_CFGS:
        hf_dir(n)
    missing = ["easydict", "skimage", "skimage.feature", "skimage.transform", "torchvision", "torchvision.transforms",
               "torchvision.transforms.functional", "matplotlib.pyplot", "cv2", "fairscale", "fairscale.nn",
               "fairscale.nn.misc", "toolz", "toolz.sandbox", "tensorboardX", "addict", "yapf", "yapf.yapflib",
               "yapf.yapflib.yapf_api", "deepspeed", "apex", "progressbar", "future", "future.utils", "ete3",
               "deprecated", "av"]
    stubbed = []
    for m in missing:
        try:
            importlib.import_module(m)
        except Exception:
            sys.modules[m] = MagicMock(name=m)
            stubbed.append(m)

    class EasyDict(dict):
        def __getattr__(s, k):
            try:
                return s[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = dict.__setitem__
        __delattr__ = dict.__delitem__
    sys.modules["easydict"].EasyDict = EasyDict
    sys.modules["toolz.sandbox"].unzip = lambda seq: zip(*seq)
    sys.path.insert(0, REF)
    os.chdir(REF)
    import visbackbone.video_swin as VS
    import model as M
    import main_pretrain_mlm as PM
    import agent as AG
    for m in stubbed:
        if isinstance(sys.modules.get(m), MagicMock):
            del sys.modules[m]

    class _Cfg:
        def __init__(s, model):
            s.model = model

        @staticmethod
        def fromfile(path):
            k = os.environ.get("LAV_SWIN_SIZE") or next(k for k in R.SWIN_SIZES if f"swin_{k}_" in path)
            c = R.SWIN_SIZES[k]
            return _Cfg({"backbone": dict(patch_size=(2, 4, 4), patch_norm=True, embed_dim=c["embed_dim"],
                                          depths=list(c["depths"]), num_heads=list(c["num_heads"]),
                                          window_size=c["window_size"])})
    VS.Config = _Cfg
    torch.Tensor.cuda = lambda self, *a, **k: self
    return types.SimpleNamespace(VS=VS, M=M, PM=PM, AG=AG, EasyDict=EasyDict)


class Tok:
    cls_token = "[CLS]"; sep_token = "[SEP]"; pad_token = "[PAD]"; mask_token = "[MASK]"; unk_token = "[UNK]"
    ids = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103, "true": 2995, "false": 6270}

    def convert_tokens_to_ids(s, toks):
        return [s.ids[t] for t in toks]


def build_reference(ref, swin, bert, B):
    os.environ["LAV_SWIN_SIZE"] = swin
    d = hf_dir(bert)
    args = ref.EasyDict(vis_backbone_size="base" if swin in ("micro", "micro12") else swin, size_img=224,
                        vis_backbone_init="random", kinetics=400, txt_backbone=d, txt_backbone_embed_only=True,
                        fusion_encoder=d, fusion_encoder_rand_init=False, use_checkpoint=False, size_patch=32,
                        size_batch=B, tokenizer=d, enable_task_token=False, enable_prompt=False, temp=0.05)
    m = ref.PM.LAVENDER_Pretrain_MLM(args, Tok())
    sd = m.state_dict()
    keys = {k: tuple(v.shape) for k, v in sd.items()}
    new = {k: R.fill_tensor(k, v.shape) for k, v in sd.items() if v.is_floating_point()}
    # decoder.bias is tied to predictions.bias
    if "fc_mtm.predictions.decoder.bias" in new:
        new["fc_mtm.predictions.decoder.bias"] = new["fc_mtm.predictions.bias"]
    m.load_state_dict(new, strict=False)
    _orig = m.mask_ext
    m.mask_ext = lambda mk, shp, dev=None: _orig(mk, shp)

    class _Enc(torch.nn.Module):
        def __init__(s, enc):
            super().__init__()
            s.enc = enc

        def forward(s, feat, mask, output_attentions=False):
            return {"last_hidden_state": s.enc(feat, mask).last_hidden_state, "attentions": None}
    m.trsfr = _Enc(m.trsfr)
    return m, keys, args


def make_batch(B, T=5, S=224, X=32, vocab=30522, seed=1):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, T, 3, S, S, generator=g)
    txt = torch.zeros(B, X, dtype=torch.long)
    for b in range(B):
        k = int(torch.randint(6, X - 4, (1,), generator=g))
        body = torch.randint(1000, min(30000, vocab), (k,), generator=g)
        txt[b, 0] = 101
        txt[b, 1:1 + k] = body
        txt[b, 1 + k] = 102
        txt[b, -1] = 103
    mask = (txt != 0).long()
    return dict(img=img, txt=txt, mask=mask)


def sub(t, n=4096, seed=7):
    """Deterministic sub-sample of a tensor: (values at fixed flat indices)."""
    flat = t.detach().reshape(-1)
    g = torch.Generator().manual_seed(seed + flat.numel() % 9973)
    idx = torch.randperm(flat.numel(), generator=g)[:n]
    return flat[idx].numpy().astype(np.float32)


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().max().item(), t.pow(2).mean().sqrt().item()])


def run_model_case(ref, name, swin, bert, B, S=224, with_grads=True, T=5, X=32):
    print(f"== {name}: swin={swin} bert={bert} B={B} S={S} T={T} X={X}")
    m, keys, args = build_reference(ref, swin, bert, B)
    vocab = BERT_CFGS[bert].get("vocab_size", 30522)
    heads = BERT_CFGS[bert].get("num_attention_heads", 12)
    batch = make_batch(B, T=T, S=S, X=X, vocab=vocab)
    torch.manual_seed(88)
    txt_m, ans = R.masking(batch["txt"])
    batch["txt"], batch["ans_mtm"] = txt_m, ans
    taps = {}
    sw = m.enc_img.swin
    hooks = [sw.patch_embed.register_forward_hook(
        lambda mod, i, o: taps.__setitem__("patch_embed", o.permute(0, 2, 3, 4, 1)))]
    for s in range(4):
        hooks.append(sw.layers[s].blocks[-1].register_forward_hook(
            lambda mod, i, o, s=s: taps.__setitem__(f"stage{s}", o)))
    m.eval()
    np.random.seed(88)
    out = m({k: v.clone() for k, v in batch.items()})
    f_img, m_img, f_txt, _ = m.go_feat(batch["img"], batch["txt"], batch["mask"])
    for h in hooks:
        h.remove()
    V = out["out_mtm"].shape[-1]
    lf = torch.nn.CrossEntropyLoss(ignore_index=-1)
    l_mtm = lf(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    l_vtm = lf(out["out_vtm"].flatten(0, 1), out["ans_vtm"].flatten())
    G = {}
    if with_grads:
        m.zero_grad()
        (l_mtm + l_vtm).backward()
        unwrap = lambda k: k.replace("trsfr.enc.", "trsfr.")
        gn, gs = {}, {}
        for k, p in m.named_parameters():
            k = unwrap(k)
            if p.grad is None:
                gn[k] = -1.0
                continue
            gn[k] = p.grad.double().norm().item()
        pick = ["enc_img.swin.patch_embed.proj.weight", "enc_img.swin.layers.0.blocks.1.attn.qkv.weight",
                "enc_img.swin.layers.0.blocks.1.attn.relative_position_bias_table",
                "enc_img.swin.layers.2.blocks.1.attn.qkv.weight", "enc_img.swin.layers.1.downsample.reduction.weight",
                "enc_img.swin.layers.2.blocks.0.mlp.fc1.weight", "enc_img.swin.layers.0.blocks.0.norm1.weight",
                "enc_img.fc.weight", "enc_img.emb_pos", "enc_img.emb_len", "enc_img.emb_cls",
                "enc_txt.emb_txt.word_embeddings.weight", "enc_txt.emb_txt.position_embeddings.weight",
                "trsfr.layer.0.attention.self.query.weight", "trsfr.layer.1.intermediate.dense.weight",
                "trsfr.layer.1.output.LayerNorm.weight", "fc_mtm.predictions.decoder.weight",
                "fc_mtm.predictions.bias", "fc_mtm.predictions.transform.dense.bias"]
        named = {unwrap(k): p for k, p in m.named_parameters()}
        for k in pick:
            if k in named and named[k].grad is not None:
                gs[k] = sub(named[k].grad, 2048)
        G["grad_norm_keys"] = np.array(list(gn.keys()))
        G["grad_norm_vals"] = np.array(list(gn.values()))
        for k, v in gs.items():
            G["grad_sub::" + k] = v
    cols = torch.randperm(V, generator=torch.Generator().manual_seed(5))[:256]
    res = dict(
        keys=np.array(list(keys.keys())), shapes=np.array([str(v) for v in keys.values()]),
        txt=batch["txt"].numpy(), ans_mtm=batch["ans_mtm"].numpy(), ans_vtm=out["ans_vtm"].numpy(),
        f_img_sub=sub(f_img), f_img_stats=stats(f_img), f_txt_sub=sub(f_txt), f_txt_stats=stats(f_txt),
        out_mtm_cols=out["out_mtm"][:, :, cols].detach().numpy().astype(np.float32), cols=cols.numpy(),
        out_vtm_cols=out["out_vtm"][:, :, cols].detach().numpy().astype(np.float32),
        out_mtm_lse=torch.logsumexp(out["out_mtm"], -1).detach().numpy(),
        out_mtm_argmax=out["out_mtm"].argmax(-1).numpy(),
        out_vtm_lse=torch.logsumexp(out["out_vtm"], -1).detach().numpy(),
        out_mtm_stats=stats(out["out_mtm"]), out_vtm_stats=stats(out["out_vtm"]),
        loss=np.array([l_mtm.item(), l_vtm.item()]), meta=np.array([swin, bert, str(B), str(S), str(heads), str(T), str(X)]),
        **{f"tap_{k}_sub": sub(v) for k, v in taps.items()},
        **{f"tap_{k}_stats": stats(v) for k, v in taps.items()}, **G)
    np.savez_compressed(f"{HERE}/{name}.npz", **res)

    # immediate oracle-vs-reference check (T1): fail loudly if the restatement drifted
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    np.random.seed(88)
    o = R.pretrain_forward(P, batch, swin, heads)
    d = (o["out_mtm"] - out["out_mtm"]).abs().max().item()
    d2 = (o["out_vtm"] - out["out_vtm"]).abs().max().item()
    print(f"   oracle vs reference: max|d| mtm {d:.2e} vtm {d2:.2e}; loss {l_mtm.item():.4f} {l_vtm.item():.4f}")
    assert d < 2e-5 and d2 < 2e-5
    return m


def run_swin_pad_case(ref):
    """5x64^2 input: exercises the pad branches video_swin.py:211-215,241-242, 273-276."""
    os.environ["LAV_SWIN_SIZE"] = "micro"
    args = ref.EasyDict(vis_backbone_size="base", size_img=224, vis_backbone_init="random", kinetics=400)
    sw = ref.VS.get_vidswin_model(args)
    sd = sw.state_dict()
    sw.load_state_dict({k: R.fill_tensor("enc_img.swin." + k, v.shape) for k, v in sd.items() if v.is_floating_point()},
                       strict=False)
    sw.eval()
    res = {}
    for T, S in ((5, 64), (4, 96), (1, 224), (6, 224)):
        x = torch.randn(1, 3, T, S, S, generator=torch.Generator().manual_seed(3))
        y = sw(x).permute(0, 2, 3, 4, 1)
        P = {"enc_img.swin." + k: v for k, v in sw.state_dict().items()}
        yo = R.swin_forward(P, "enc_img.swin", x, "micro")
        d = (y - yo).abs().max().item()
        print(f"   swin pad case T={T} S={S}: out {tuple(y.shape)} oracle max|d| {d:.2e}")
        assert d < 2e-5
        res[f"T{T}_S{S}_sub"] = sub(y, 2048)
        res[f"T{T}_S{S}_stats"] = stats(y)
    np.savez_compressed(f"{HERE}/swin_shapes.npz", **res)


def run_int_cases(ref):
    res = {}
    VS = ref.VS
    # get_window_size
    cases = [((5, 56, 56), (8, 7, 7), (4, 3, 3)), ((5, 7, 7), (8, 7, 7), (4, 3, 3)), ((16, 56, 56), (8, 7, 7), (4, 3, 3)),
             ((5, 96, 96), (8, 12, 12), (4, 6, 6)), ((1, 56, 56), (8, 7, 7), (4, 3, 3)), ((5, 12, 12), (8, 12, 12), (4, 6, 6)),
             ((4, 16, 16), (8, 7, 7), (4, 3, 3))]
    res["gws_in"] = np.array(cases)
    res["gws_out"] = np.array([VS.get_window_size(*c) for c in cases])
    # relative_position_index
    for w in ((8, 7, 7), (8, 12, 12)):
        att = VS.WindowAttention3D(32, w, 1, qkv_bias=True)
        idx = att.relative_position_index.numpy()
        tag = "x".join(map(str, w))
        res[f"rpi_{tag}_sha"] = np.array(hashlib.sha256(idx.astype(np.int64).tobytes()).hexdigest())
        res[f"rpi_{tag}_corner"] = idx[:50, :50]
    # compute_mask
    for (D, H, W, win, sh) in ((5, 56, 56, (5, 7, 7), (0, 3, 3)), (5, 14, 14, (5, 7, 7), (0, 3, 3)),
                               (5, 96, 96, (5, 12, 12), (0, 6, 6)), (16, 14, 14, (8, 7, 7), (4, 3, 3)),
                               (5, 21, 21, (5, 7, 7), (0, 3, 3))):
        mk = VS.compute_mask(D, H, W, win, sh, torch.device("cpu"))
        tag = f"{D}_{H}_{W}_" + "x".join(map(str, win)) + "_" + "x".join(map(str, sh))
        bits = (mk != 0).numpy()
        assert set(np.unique(mk.numpy()).tolist()) <= {-100.0, 0.0}
        res[f"mask_{tag}_sha"] = np.array(hashlib.sha256(np.packbits(bits).tobytes()).hexdigest())
        res[f"mask_{tag}_shape"] = np.array(mk.shape)
        res[f"mask_{tag}_frac"] = np.array(bits.mean())
        if bits.size < 1_000_000:
            res[f"mask_{tag}_bits"] = np.packbits(bits)
    # masking
    ns = types.SimpleNamespace(cls_token_id=101, sep_token_id=102, pad_token_id=0, mask_token_id=103)
    for seed in (88, 0, 1):
        for (B, X) in ((2, 33), (8, 32), (32, 32)):
            b = make_batch(B, S=8, X=X, seed=seed + 100)
            torch.manual_seed(seed)
            o = ref.PM.Agent_Pretrain_MLM.masking(ns, b["txt"].clone(), b["mask"])
            res[f"masking_s{seed}_B{B}_X{X}_in"] = b["txt"].numpy()
            res[f"masking_s{seed}_B{B}_X{X}_txt"] = o["txt"].numpy()
            res[f"masking_s{seed}_B{B}_X{X}_ans"] = o["ans_mtm"].numpy()
    # survey appendix C case
    txt = torch.tensor([[101] + list(range(2000, 2020)) + [102] + [0] * 10 + [103]] * 2)
    torch.manual_seed(88)
    o = ref.PM.Agent_Pretrain_MLM.masking(ns, txt.clone(), (txt != 0).long())
    res["masking_appC_ans"] = o["ans_mtm"].numpy()
    # LR schedule
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=2e-5)
    sch = ref.AG.WarmupLinearLR(opt, 100)
    lrs = []
    for _ in range(110):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    res["lr_max_iter100_lr2e-5"] = np.array(lrs)
    np.savez_compressed(f"{HERE}/ints.npz", **res)


def run_agent_case(ref, m):
    """Optimizer groups + one optimizer step of the reference Agent on Tiny+2L (CPU)."""
    args = m.args
    args.update(lr=2e-5, decay=1e-3, max_iter=100, max_grad_norm=1.0, deepspeed=False, vis_backbone_lr_mul=1,
                dataset=["x"], logging_steps=10, path_output="/tmp/lav_golden/out", task="pretrain")
    m.trsfr = m.trsfr.enc  # unwrap for clean names; re-wrap below
    ag = ref.PM.Agent_Pretrain_MLM(args, m)
    names = {id(p): n for n, p in m.named_parameters()}
    groups = [[names[id(p)] for p in g["params"]] for g in ag.optzr.param_groups]
    res = {f"group{i}": np.array(g) for i, g in enumerate(groups)}
    res["group_sizes"] = np.array([len(g) for g in groups])

    class _Enc(torch.nn.Module):
        def __init__(s, enc):
            super().__init__()
            s.enc = enc

        def forward(s, feat, mask, output_attentions=False):
            return {"last_hidden_state": s.enc(feat, mask).last_hidden_state, "attentions": None}
    m.trsfr = _Enc(m.trsfr)
    np.savez_compressed(f"{HERE}/agent.npz", **res)
    print("   optimizer group sizes", res["group_sizes"])


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = import_reference()
    run_int_cases(ref)
    run_swin_pad_case(ref)
    run_model_case(ref, "micro_b2", "micro", "micro", 2)
    run_model_case(ref, "micro_b5", "micro", "micro", 5, with_grads=False)
    m = run_model_case(ref, "tiny2l_b2", "tiny", "b2l", 2)
    run_agent_case(ref, m)
    print("goldens written to", HERE)
