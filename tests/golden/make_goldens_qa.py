"""Golden vectors for the MLM-head question-answering callers (CONTAINER ONLY -- needs /root/reference):
  * LAVENDER_QAOE_MLM (main_qaoe_mlm_lsmdc_fib.py:64-93): open-ended QA, X = 26 ("... answer: [MASK]");
  * LAVENDER_QAMC_MLM (main_qamc_mlm.py:109-140): multiple choice, question + options as one text (X = 41 here), evaluated
    over the option-index tokens (Agent_QAMC_MLM.step, :160-170).
Same recipe as make_goldens.py / make_goldens_variants.py; writes qaoe_micro_b3.npz and qamc_micro_b3.npz;
  * LAVENDER_RetMC_MLM (main_retmc_mlm.py:70-113): retrieval multiple choice, O candidate texts per video -> retmc_micro_b2.npz.

    python tests/golden/make_goldens_qa.py
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
from make_goldens import R, make_batch, sub, stats, BERT_CFGS  # noqa: E402
from make_goldens_variants import build, grads_of  # noqa: E402

ANS_TOK_IDS = [1014, 1015, 1016, 1017, 1018]       # bert-base-uncased ids of "0" .. "4" (Dataset_QAMC_MLM.ans_tok_ids)


def qa_batch(B, X, vocab, seed, answers):
    """Text with one [MASK] whose label is the answer token (the datasets' mask_ans, main_qaoe_mlm.py:68-69, main_qamc_mlm.py:83-84)."""
    b = make_batch(B, X=X, vocab=vocab, seed=seed)
    txt = b["txt"]
    for i in range(B):
        k = int((txt[i] != 0).sum()) - 2               # make_batch ends every row with a [MASK]; put one before [SEP] as well
        txt[i, max(2, k - 1)] = 103
        txt[i, -1] = 0
    b["mask"] = (txt != 0).long()
    ans = torch.full(txt.shape, -1, dtype=torch.long)
    for i in range(B):
        ans[i][txt[i] == 103] = answers[i]
    b["mask_ans"] = ans
    return b


def run(ref, name, cls_path, X, answers):
    mod, cls = cls_path
    M = importlib.import_module(mod)
    B, swin, bert = 3, "micro", "micro"
    m, keys = build(ref, getattr(M, cls), swin, bert, B, size_vocab=-1, size_option=5)
    vocab, heads = BERT_CFGS[bert]["vocab_size"], BERT_CFGS[bert]["num_attention_heads"]
    batch = qa_batch(B, X, vocab, 6, answers)
    m.eval()
    out, ans = m(batch)
    lf = torch.nn.CrossEntropyLoss(ignore_index=-1)
    ls = lf(out.flatten(0, 1), ans.flatten())
    m.zero_grad()
    ls.backward()
    gk, gv = grads_of(m)
    V = out.shape[-1]
    cols = torch.cat([torch.tensor(ANS_TOK_IDS), torch.randperm(V, generator=torch.Generator().manual_seed(5))[:251]])
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    o, a = R.qa_mlm_forward(P, batch, swin, heads)
    d = (o - out).abs().max().item()
    ac1, ac5 = R.qa_top_k_acc(out.detach(), ans, 1), R.qa_top_k_acc(out.detach(), ans, 5)
    mc = R.qamc_choice_acc(out.detach().softmax(-1), ans, ANS_TOK_IDS, torch.tensor([a_ - 1014 if 1014 <= a_ <= 1018 else 0 for a_ in answers]))
    print(f"   {name}: oracle vs reference max|d| {d:.2e}; labels equal {bool((a == ans).all())}; loss {ls.item():.4f}; top-1 {ac1} top-5 {ac5}")
    assert d < 2e-5 and bool((a == ans).all())
    named = {k.replace("trsfr.enc.", "trsfr."): p for k, p in m.named_parameters()}
    np.savez_compressed(
        f"{HERE}/{name}.npz", keys=np.array(list(keys.keys())), txt=batch["txt"].numpy(), mask_ans=ans.numpy(),
        out_cols=out[:, :, cols].detach().numpy().astype(np.float32), cols=cols.numpy(), out_lse=torch.logsumexp(out, -1).detach().numpy(),
        out_stats=stats(out), loss=np.array([ls.item()]), ac_1=np.array(ac1), ac_5=np.array(ac5), grad_norm_keys=gk, grad_norm_vals=gv,
        **{"grad_sub::" + k: sub(named[k].grad, 2048) for k in ("fc_mtm.predictions.transform.dense.weight", "trsfr.layer.1.output.dense.weight",
                                                               "enc_img.emb_cls", "enc_img.swin.layers.1.blocks.0.attn.qkv.weight")},
        meta=np.array([swin, bert, str(B), "224", str(heads), str(X)]))


def run_retmc(ref):
    """LAVENDER_RetMC_MLM (main_retmc_mlm.py:70-113): B = 2 videos x O = 3 candidate texts of X = 26 tokens, a [MASK] at the end of each."""
    M = importlib.import_module("main_retmc_mlm")
    B, O, X, swin, bert = 2, 3, 26, "micro", "micro"
    m, keys = build(ref, M.LAVENDER_RetMC_MLM, swin, bert, B, size_vocab=-1, size_option=O)
    vocab, heads = BERT_CFGS[bert]["vocab_size"], BERT_CFGS[bert]["num_attention_heads"]
    b = make_batch(B * O, X=X, vocab=vocab, seed=12)
    txt = b["txt"].view(B, O, X)
    ans = torch.full(txt.shape, -1, dtype=torch.long)
    ans[:, :, -1] = torch.tensor([[2995, 6270, 6270], [6270, 6270, 2995]])     # one true candidate per video
    batch = {"img": make_batch(B, vocab=vocab, seed=13)["img"], "txt": txt, "mask": (txt != 0).long(), "mask_ans": ans}
    m.eval()
    out, a = m(batch)
    lf = torch.nn.CrossEntropyLoss(ignore_index=-1)
    ls = lf(out.flatten(0, 1), a.flatten())
    m.zero_grad()
    ls.backward()
    gk, gv = grads_of(m)
    P = {k.replace("trsfr.enc.", "trsfr."): v.detach() for k, v in m.state_dict().items()}
    o, a2 = R.retmc_mlm_forward(P, batch, swin, heads)
    d = (o - out).abs().max().item()
    ac = R.retmc_acc(out.detach().softmax(-1), a)
    print(f"   retmc: oracle vs reference max|d| {d:.2e}; labels equal {bool((a2 == a).all())}; loss {ls.item():.4f}; acc {ac}")
    assert d < 2e-5 and bool((a2 == a).all())
    V = out.shape[-1]
    cols = torch.cat([torch.tensor([2995, 6270]), torch.randperm(V, generator=torch.Generator().manual_seed(5))[:254]])
    np.savez_compressed(
        f"{HERE}/retmc_micro_b2.npz", keys=np.array(list(keys.keys())), txt=txt.numpy(), mask_ans=a.numpy(),
        out_cols=out[:, :, cols].detach().numpy().astype(np.float32), cols=cols.numpy(), out_lse=torch.logsumexp(out, -1).detach().numpy(),
        out_stats=stats(out), loss=np.array([ls.item()]), acc=np.array(ac), grad_norm_keys=gk, grad_norm_vals=gv,
        meta=np.array([swin, bert, str(B), str(O), str(heads), str(X)]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = MG.import_reference()
    run(ref, "qaoe_micro_b3", ("main_qaoe_mlm_lsmdc_fib", "LAVENDER_QAOE_MLM"), 26, [2023, 3899, 2158])
    run(ref, "qamc_micro_b3", ("main_qamc_mlm", "LAVENDER_QAMC_MLM"), 41, [1016, 1014, 1018])
    run_retmc(ref)
