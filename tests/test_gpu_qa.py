"""The MLM-head question-answering callers on the HIP engine (lavender_amd/qa_mlm.py) against the CPU oracle and the reference
fixtures (tests/golden/qaoe_micro_b3.npz, qamc_micro_b3.npz: main_qaoe_mlm_lsmdc_fib.py:64-125, main_qamc_mlm.py:109-170)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import BERT_CFGS, make_batch
from tests.test_gpu_variants import _grad_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,cls", [("qaoe_micro_b3", "LAVENDER_QAOE_MLM"), ("qamc_micro_b3", "LAVENDER_QAMC_MLM")])
def test_qa_forward_loss_gradients(golden_dir, name, cls):
    import lavender_amd as LA
    from oracle import lavender_ref as R
    from tests.helpers import build_filled_model
    from lavender_amd.agent import CrossEntropyIgnore
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    swin, bert, B, S, heads, X = g["meta"].tolist()
    B, heads, X = int(B), int(heads), int(X)
    bc = BERT_CFGS[bert]
    P = R.filled_params(swin, hidden=bc["hidden"], layers=bc["layers"], ffn=bc["ffn"], vocab=bc["vocab"])
    for v in P.values():
        v.requires_grad_(True)
    batch = make_batch(B, X=X, vocab=bc["vocab"], seed=6)
    batch["txt"] = torch.from_numpy(g["txt"]); batch["mask"] = (batch["txt"] != 0).long(); batch["mask_ans"] = torch.from_numpy(g["mask_ans"])
    ref, ans_ref = R.qa_mlm_forward(P, batch, swin, heads)
    l_ref = torch.nn.functional.cross_entropy(ref.flatten(0, 1), ans_ref.flatten(), ignore_index=-1)
    l_ref.backward()
    P["emb_task"].grad = None

    m = build_filled_model(swin, bert, B, cls=getattr(LA, cls)).eval()
    m.arena().zero_grad()
    out, ans = m({k: v.cuda() for k, v in batch.items()})
    assert out.shape == (B, X, bc["vocab"]) and (ans.cpu() == batch["mask_ans"]).all()
    a = out.float().cpu()
    d = (a - ref).abs()
    print(name, "logits max", d.max().item(), "mean", d.mean().item())
    assert d.max() < 3e-2 and d.mean() < 5e-3
    np.testing.assert_allclose(a[:, :, torch.from_numpy(g["cols"])].detach().numpy(), g["out_cols"], atol=3e-2)
    assert np.abs(torch.logsumexp(a, -1).detach().numpy() - g["out_lse"]).max() < 2e-2
    ls = CrossEntropyIgnore()(out.flatten(0, 1), ans.flatten(), count=int((batch["mask_ans"] != -1).sum()))
    ls.backward()
    torch.cuda.synchronize()
    assert abs(ls.item() - g["loss"][0]) < 1e-2 and abs(ls.item() - l_ref.item()) < 1e-2
    _grad_check(m, P)


def test_qa_agents_train_and_evaluate():
    """Agent_QAOE_MLM (loss dict / top-1, top-5 lists) and Agent_QAMC_MLM (float loss / per-sample hits over the option tokens):
    the loss falls on a repeated batch and the evaluation outputs have the reference's shapes and agree with the oracle's
    restatement of the same formulas on the model's own logits."""
    import lavender_amd as LA
    from oracle import lavender_ref as R
    from tests.helpers import Tok, make_args
    from lavender_amd.dist import set_seed
    set_seed(88)
    B, X = 4, 26
    opt_ids = [1014, 1015, 1016, 1017, 1018]
    b = make_batch(B, X=X, vocab=BERT_CFGS["micro"]["vocab"], seed=6)
    ans = torch.full(b["txt"].shape, -1, dtype=torch.long)
    ans[:, -1] = torch.tensor([1016, 1014, 1018, 1015])                 # make_batch ends every row with [MASK]
    b["mask_ans"] = ans
    b["ans_idx"] = torch.tensor([2, 0, 4, 1])
    for kind in ("oe", "mc"):
        args = make_args("micro", "micro", B, lr=2e-3, max_iter=40, size_vocab=-1, size_option=5)
        m = (LA.LAVENDER_QAOE_MLM if kind == "oe" else LA.LAVENDER_QAMC_MLM)(args, Tok()).cuda()
        m.arena()
        ag = LA.Agent_QAOE_MLM(args, m) if kind == "oe" else LA.Agent_QAMC_MLM(args, m, opt_ids)
        losses = []
        for _ in range(12):
            r = ag.step(ag.prepare_batch(dict(b)), True)
            losses.append(r["ls"] if kind == "oe" else r)
        print(kind, "losses", [round(x, 3) for x in losses])
        assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 1.0
        ev = ag.step(ag.prepare_batch(dict(b)), False)
        m.eval()
        with torch.no_grad():
            out, a = m({k: v.cuda() for k, v in b.items()})
        out = out.float().cpu()
        if kind == "oe":
            assert set(ev) == {"ac_1", "ac_5"} and len(ev["ac_1"]) == B
            assert ev["ac_1"] == R.qa_top_k_acc(out, ans, 1) and ev["ac_5"] == R.qa_top_k_acc(out, ans, 5)
            assert all(t5 >= t1 for t1, t5 in zip(ev["ac_1"], ev["ac_5"]))
        else:
            assert isinstance(ev, list) and len(ev) == B and set(ev) <= {0.0, 1.0}
            assert ev == R.qamc_choice_acc(out, ans, opt_ids, b["ans_idx"])


def test_retmc_forward_loss_gradients_and_agent(golden_dir):
    """LAVENDER_RetMC_MLM: O candidate texts per video through the pair-index gather (no expanded feat_img copies) vs oracle and the
    reference fixture; Agent_RetMC_MLM trains and returns one hit flag per video."""
    import lavender_amd as LA
    from oracle import lavender_ref as R
    from tests.helpers import Tok, build_filled_model, make_args
    from lavender_amd.agent import CrossEntropyIgnore
    g = np.load(os.path.join(golden_dir, "retmc_micro_b2.npz"))
    swin, bert, B, O, heads, X = g["meta"].tolist()
    B, O, heads, X = int(B), int(O), int(heads), int(X)
    bc = BERT_CFGS[bert]
    P = R.filled_params(swin, hidden=bc["hidden"], layers=bc["layers"], ffn=bc["ffn"], vocab=bc["vocab"])
    for v in P.values():
        v.requires_grad_(True)
    txt = torch.from_numpy(g["txt"])
    batch = {"img": make_batch(B, vocab=bc["vocab"], seed=13)["img"], "txt": txt, "mask": (txt != 0).long(), "mask_ans": torch.from_numpy(g["mask_ans"])}
    ref, ans_ref = R.retmc_mlm_forward(P, batch, swin, heads)
    l_ref = torch.nn.functional.cross_entropy(ref.flatten(0, 1), ans_ref.flatten(), ignore_index=-1)
    l_ref.backward()
    P["emb_task"].grad = None
    m = build_filled_model(swin, bert, B, cls=LA.LAVENDER_RetMC_MLM).eval()
    m.arena().zero_grad()
    out, ans = m({k: v.cuda() for k, v in batch.items()})
    assert out.shape == (B * O, X, bc["vocab"]) and ans.shape == (B, O, X) and (ans.cpu() == batch["mask_ans"]).all()
    a = out.float().cpu()
    d = (a - ref).abs()
    print("retmc logits max", d.max().item(), "mean", d.mean().item())
    assert d.max() < 3e-2 and d.mean() < 5e-3
    np.testing.assert_allclose(a[:, :, torch.from_numpy(g["cols"])].detach().numpy(), g["out_cols"], atol=3e-2)
    ls = CrossEntropyIgnore()(out.flatten(0, 1), ans.flatten(), count=B * O)
    ls.backward()
    torch.cuda.synchronize()
    assert abs(ls.item() - g["loss"][0]) < 1e-2
    _grad_check(m, P)
    # agent: train on the same batch, then the eval branch
    args = make_args("micro", "micro", B, lr=2e-3, max_iter=40, size_vocab=-1)
    m2 = LA.LAVENDER_RetMC_MLM(args, Tok()).cuda()
    m2.arena()
    ag = LA.Agent_RetMC_MLM(args, m2)
    losses = [ag.step(ag.prepare_batch(dict(batch)), True) for _ in range(10)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 1.0, losses
    ev = ag.step(ag.prepare_batch(dict(batch)), False)
    m2.eval()
    with torch.no_grad():
        o2, a2 = m2({k: v.cuda() for k, v in batch.items()})
    assert isinstance(ev, list) and len(ev) == B and ev == R.retmc_acc(o2.float().cpu(), a2.cpu())
