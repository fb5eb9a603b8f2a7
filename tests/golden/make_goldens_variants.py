"""Golden vectors for the callers next to the MLM pre-training model (CONTAINER ONLY -- needs /root/reference):
  * LAVENDER_Pretrain (main_pretrain_task_specific.py:124-177): MLM + scalar video-text-matching score head;
  * LAVENDER_Retrieval_MLM (main_retrieval_mlm.py:30-91): all B x B pairs through the MLM head.
Same recipe as make_goldens.py (stubbed third-party imports, parameters filled from their state_dict keys,
seeded inputs); writes ts_micro_b5.npz, retr_micro_b3.npz and micro12_s384_b2.npz (MLM model on 384^2 frames with the
Large-384 window geometry) next to this file.

    python tests/golden/make_goldens_variants.py
"""
import sys
sys.dont_write_bytecode = True
import importlib
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402
from make_goldens import R, Tok, make_batch, sub, stats, hf_dir, BERT_CFGS  # noqa: E402


def build(ref, cls, swin, bert, B, **extra):
    os.environ["LAV_SWIN_SIZE"] = swin
    d = hf_dir(bert)
    args = ref.EasyDict(vis_backbone_size="base", size_img=224, vis_backbone_init="random", kinetics=400, txt_backbone=d,
                        txt_backbone_embed_only=True, fusion_encoder=d, fusion_encoder_rand_init=False, use_checkpoint=False,
                        size_patch=32, size_batch=B, tokenizer=d, enable_task_token=False, enable_prompt=False, temp=0.05)
    args.update(extra)
    m = cls(args, Tok())
    sd = m.state_dict()
    new = {k: R.fill_tensor(k, v.shape) for k, v in sd.items() if v.is_floating_point()}
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
    return m, {k: tuple(v.shape) for k, v in sd.items()}


def grads_of(m):
    unwrap = lambda k: k.replace("trsfr.enc.", "trsfr.")
    gn = {unwrap(k): (p.grad.double().norm().item() if p.grad is not None else -1.0) for k, p in m.named_parameters()}
    return np.array(list(gn.keys())), np.array(list(gn.values()))


def run_task_specific(ref):
    TS = importlib.import_module("main_pretrain_task_specific")
    B, swin, bert = 5, "micro", "micro"
    m, keys = build(ref, TS.LAVENDER_Pretrain, swin, bert, B)
    vocab, heads = BERT_CFGS[bert]["vocab_size"], BERT_CFGS[bert]["num_attention_heads"]
    batch = make_batch(B, vocab=vocab)
    torch.manual_seed(88)
    batch["txt"], ans = R.masking(batch["txt"])
    m.eval()
    np.random.seed(88)
    out = m(batch["img"], batch["txt"].clone(), batch["mask"], ans)
    lf = torch.nn.CrossEntropyLoss(ignore_index=-1)
    l_mtm = lf(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten())
    l_vtm = lf(out["out_vtm"], out["ans_vtm"])
    m.zero_grad()
    (l_mtm + l_vtm).backward()
    gk, gv = grads_of(m)
    named = {k.replace("trsfr.enc.", "trsfr."): p for k, p in m.named_parameters()}
    V = out["out_mtm"].shape[-1]
    cols = torch.randperm(V, generator=torch.Generator().manual_seed(5))[:256]
    np.savez_compressed(
        f"{HERE}/ts_micro_b5.npz", keys=np.array(list(keys.keys())), shapes=np.array([str(v) for v in keys.values()]),
        txt=batch["txt"].numpy(), ans_mtm=ans.numpy(), out_vtm=out["out_vtm"].detach().numpy(), ans_vtm=out["ans_vtm"].numpy(),
        out_mtm_cols=out["out_mtm"][:, :, cols].detach().numpy().astype(np.float32), cols=cols.numpy(),
        out_mtm_lse=torch.logsumexp(out["out_mtm"], -1).detach().numpy(), out_mtm_stats=stats(out["out_mtm"]),
        loss=np.array([l_mtm.item(), l_vtm.item()]), grad_norm_keys=gk, grad_norm_vals=gv,
        **{"grad_sub::" + k: sub(named[k].grad, 2048) for k in ("fc.1.weight", "fc.1.bias", "fc.3.weight", "fc.3.bias",
                                                               "trsfr.layer.1.output.dense.weight", "enc_img.emb_cls")},
        meta=np.array([swin, bert, str(B), "224", str(heads), "0.05"]))
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    np.random.seed(88)
    o = R.pretrain_ts_forward(P, dict(batch, ans_mtm=ans), swin, heads, 0.05)
    d1 = (o["out_mtm"] - out["out_mtm"]).abs().max().item()
    d2 = (o["out_vtm"] - out["out_vtm"]).abs().max().item()
    print(f"   task-specific: oracle vs reference max|d| mtm {d1:.2e} vtm {d2:.2e}; loss {l_mtm.item():.4f} {l_vtm.item():.4f}")
    assert d1 < 2e-5 and d2 < 2e-4


def run_retrieval(ref):
    RM = importlib.import_module("main_retrieval_mlm")
    B, swin, bert = 3, "micro", "micro"
    m, keys = build(ref, RM.LAVENDER_Retrieval_MLM, swin, bert, B)
    vocab, heads = BERT_CFGS[bert]["vocab_size"], BERT_CFGS[bert]["num_attention_heads"]
    batch = make_batch(B, vocab=vocab, seed=4)
    batch["vid"] = [7, 9, 7]                      # two captions of the same video: two positives per row 0 / 2
    m.eval()
    out, ans = m(batch)
    lf = torch.nn.CrossEntropyLoss(ignore_index=-1)
    ls = lf(out.flatten(0, 1), ans.flatten())
    m.zero_grad()
    ls.backward()
    gk, gv = grads_of(m)
    V = out.shape[-1]
    cols = torch.cat([torch.tensor([2995, 6270]), torch.randperm(V, generator=torch.Generator().manual_seed(5))[:254]])
    np.savez_compressed(
        f"{HERE}/retr_micro_b3.npz", keys=np.array(list(keys.keys())), vid=np.array(batch["vid"]), txt=batch["txt"].numpy(),
        ans=ans.numpy(), out_cols=out[:, :, cols].detach().numpy().astype(np.float32), cols=cols.numpy(),
        out_lse=torch.logsumexp(out, -1).detach().numpy(), out_stats=stats(out), loss=np.array([ls.item()]),
        grad_norm_keys=gk, grad_norm_vals=gv, meta=np.array([swin, bert, str(B), "224", str(heads)]))
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    o, a = R.retrieval_forward(P, batch, swin, heads)
    d = (o - out).abs().max().item()
    print(f"   retrieval: oracle vs reference max|d| {d:.2e}; labels equal {bool((a == ans).all())}; loss {ls.item():.4f}")
    assert d < 2e-5 and bool((a == ans).all())


def run_retrieval_eval(ref):
    """Two-phase retrieval inference (eval_retrieval_mlm.py:10-47): 'feat' with 2 clips per video, then 'cross' on every
    (caption p, video q) pair in the order of Dataset_Product (p outer, q inner)."""
    EV = importlib.import_module("eval_retrieval_mlm")
    B, Cl, swin, bert = 2, 2, "micro", "micro"
    m, keys = build(ref, EV.LAVENDER_RetrievalMlmEval, swin, bert, B)
    vocab, heads = BERT_CFGS[bert]["vocab_size"], BERT_CFGS[bert]["num_attention_heads"]
    b = make_batch(B * Cl, T=4, vocab=vocab, seed=9)
    img = b["img"].view(B, Cl, 4, 3, 224, 224)
    txt, mask = b["txt"][:B], b["mask"][:B]
    m.eval()
    with torch.no_grad():
        f_img, m_img, f_txt, m_txt, _ = m('feat', {"img": img, "txt": txt, "mask": mask})
        pi = torch.tensor([p for p in range(B) for q in range(B)]); qi = torch.tensor([q for p in range(B) for q in range(B)])
        out, _ = m('cross', {"feat_img": f_img[qi], "mask_img": m_img[qi], "feat_txt": f_txt[pi], "mask_txt": m_txt[pi], "txt": txt[pi]})
    V = out.shape[-1]
    cols = torch.cat([torch.tensor([2995, 6270]), torch.randperm(V, generator=torch.Generator().manual_seed(5))[:254]])
    np.savez_compressed(f"{HERE}/retr_eval_micro.npz", txt=txt.numpy(), f_img_sub=sub(f_img), f_img_stats=stats(f_img),
                        f_txt_sub=sub(f_txt), out_cols=out[:, :, cols].numpy().astype(np.float32), cols=cols.numpy(),
                        out_lse=torch.logsumexp(out, -1).numpy(), meta=np.array([swin, bert, str(B), str(Cl), str(heads), "4"]))
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    with torch.no_grad():
        of_img, om_img, of_txt = R.retrieval_eval_feat(P, img, txt, swin)
        o = R.retrieval_eval_cross(P, of_img[qi], om_img[qi], of_txt[pi], mask[pi], heads)
    d1, d2 = (of_img - f_img).abs().max().item(), (o - out).abs().max().item()
    print(f"   retrieval eval: oracle vs reference max|d| feat {d1:.2e} logits {d2:.2e}")
    assert d1 < 2e-5 and d2 < 2e-5


def run_captioning(ref):
    """LAVENDER_Captioning.encode_forward (model_for_captioning.py:54-95) under the seq2seq attention mask of
    LAVENDER_Base.get_attn_mask (model.py:208-218): logits on the text positions, MLM loss, gradients."""
    CAP = importlib.import_module("model_for_captioning")
    B, swin, bert = 2, "micro", "micro"
    m, keys = build(ref, CAP.LAVENDER_Captioning, swin, bert, B)
    vocab, heads = BERT_CFGS[bert]["vocab_size"], BERT_CFGS[bert]["num_attention_heads"]
    batch = make_batch(B, vocab=vocab, seed=6)
    torch.manual_seed(88)
    batch["txt"], ans = R.masking(batch["txt"])
    m.eval()
    mask3 = m.get_attn_mask(torch.ones(B, 250, dtype=torch.long), batch["mask"], attn_mask_type="seq2seq")
    out = m({"img": batch["img"], "txt": batch["txt"].clone(), "mask": batch["mask"], "ans_mtm": ans.clone(), "attn_mask_type": "seq2seq"})
    lf = torch.nn.CrossEntropyLoss(ignore_index=-1)
    ls = lf(out["out"].flatten(0, 1), out["ans"].flatten())
    m.zero_grad()
    ls.backward()
    gk, gv = grads_of(m)
    named = {k.replace("trsfr.enc.", "trsfr."): p for k, p in m.named_parameters()}
    V = out["out"].shape[-1]
    cols = torch.randperm(V, generator=torch.Generator().manual_seed(5))[:256]
    np.savez_compressed(
        f"{HERE}/cap_micro_b2.npz", txt=batch["txt"].numpy(), ans=ans.numpy(), mask_rows=mask3[:, [0, 249, 250, 260, 281]].numpy().astype(np.int8),
        out_cols=out["out"][:, :, cols].detach().numpy().astype(np.float32), cols=cols.numpy(),
        out_lse=torch.logsumexp(out["out"], -1).detach().numpy(), loss=np.array([ls.item()]), grad_norm_keys=gk, grad_norm_vals=gv,
        **{"grad_sub::" + k: sub(named[k].grad, 2048) for k in ("trsfr.layer.0.attention.self.key.weight", "enc_img.emb_pos",
                                                               "enc_txt.emb_txt.position_embeddings.weight")},
        meta=np.array([swin, bert, str(B), "224", str(heads)]))
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    o = R.captioning_encode_forward(P, dict(batch, ans_mtm=ans, attn_mask_type="seq2seq"), swin, heads)
    d = (o["out"] - out["out"]).abs().max().item()
    same_mask = bool((R.attn_mask(torch.ones(B, 250, dtype=torch.long), batch["mask"], "seq2seq") == mask3).all())
    print(f"   captioning (seq2seq mask): oracle vs reference max|d| {d:.2e}; mask identical {same_mask}; loss {ls.item():.4f}")
    assert d < 2e-5 and same_mask


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = MG.import_reference()
    only = set(sys.argv[1:])                               # e.g. `make_goldens_variants.py shapes` regenerates one group
    if not only or "large384" in only:
        # BASELINE config 4 geometry (384^2 frames, window (8,12,12) -> (5,12,12), N = 720 tokens per window) at micro widths
        MG.run_model_case(ref, "micro12_s384_b2", "micro12", "micro", 2, S=384)
    if not only or "shapes" in only:
        # shapes of the shipped pre-training json (4 frames, 32 + 1 text positions) at batch 1 (no negatives: O = min(B, 4) = 1),
        # and the frame-count maximum (6) with short, mostly padded captions at an odd batch
        MG.run_model_case(ref, "micro_b1_t4_x33", "micro", "micro", 1, T=4, X=33)
        MG.run_model_case(ref, "micro_b3_t6_x20", "micro", "micro", 3, T=6, X=20)
    if not only or "task_specific" in only:
        run_task_specific(ref)
    if not only or "retrieval" in only:
        run_retrieval(ref)
    if not only or "retrieval_eval" in only:
        run_retrieval_eval(ref)
    if not only or "captioning" in only:
        run_captioning(ref)
    print("variant goldens written to", HERE)
