"""CPU: the oracle (oracle/lavender_ref.py) against golden vectors captured from the real
reference (tests/golden/make_goldens.py).  Tolerances: integer paths bit-exact; fp32 <= 1e-5
on logits (T1 of SURVEY.md section 8c)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import lavender_ref as R
from tests.helpers import make_batch, sub, stats, BERT_CFGS


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def test_get_window_size(golden_dir):
    g = _load(golden_dir, "ints")
    for c, o in zip(g["gws_in"], g["gws_out"]):
        w, s = R.use_window(tuple(c[0]), tuple(c[1]), tuple(c[2]))
        assert (list(w), list(s)) == (list(o[0]), list(o[1]))


@pytest.mark.parametrize("win", [(8, 7, 7), (8, 12, 12)])
def test_relative_position_index(golden_dir, win):
    g = _load(golden_dir, "ints")
    idx = R.rel_pos_index(win).numpy().astype(np.int64)
    tag = "x".join(map(str, win))
    assert hashlib.sha256(idx.tobytes()).hexdigest() == str(g[f"rpi_{tag}_sha"])
    assert (idx[:50, :50] == g[f"rpi_{tag}_corner"]).all()


@pytest.mark.parametrize("case", [(5, 56, 56, (5, 7, 7), (0, 3, 3)), (5, 14, 14, (5, 7, 7), (0, 3, 3)),
                                  (5, 96, 96, (5, 12, 12), (0, 6, 6)), (16, 14, 14, (8, 7, 7), (4, 3, 3)),
                                  (5, 21, 21, (5, 7, 7), (0, 3, 3))])
def test_shift_mask(golden_dir, case):
    g = _load(golden_dir, "ints")
    D, H, W, win, sh = case
    mk = R.shift_mask(D, H, W, win, sh)
    tag = f"{D}_{H}_{W}_" + "x".join(map(str, win)) + "_" + "x".join(map(str, sh))
    assert list(mk.shape) == list(g[f"mask_{tag}_shape"])
    bits = (mk != 0).numpy()
    assert set(np.unique(mk.numpy()).tolist()) <= {-100.0, 0.0}
    assert hashlib.sha256(np.packbits(bits).tobytes()).hexdigest() == str(g[f"mask_{tag}_sha"])


@pytest.mark.parametrize("seed", [88, 0, 1])
@pytest.mark.parametrize("BX", [(2, 33), (8, 32), (32, 32)])
def test_masking_bit_exact(golden_dir, seed, BX):
    g = _load(golden_dir, "ints")
    B, X = BX
    tin = torch.from_numpy(g[f"masking_s{seed}_B{B}_X{X}_in"])
    torch.manual_seed(seed)
    txt, ans = R.masking(tin)
    assert (txt.numpy() == g[f"masking_s{seed}_B{B}_X{X}_txt"]).all()
    assert (ans.numpy() == g[f"masking_s{seed}_B{B}_X{X}_ans"]).all()


def test_masking_known_answer(golden_dir):
    # seed 88 masks {5,14} in row 0 and {5,9,12} in row 1
    g = _load(golden_dir, "ints")
    txt = torch.tensor([[101] + list(range(2000, 2020)) + [102] + [0] * 10 + [103]] * 2)
    torch.manual_seed(88)
    _, ans = R.masking(txt)
    assert (ans.numpy() == g["masking_appC_ans"]).all()
    assert sorted(torch.nonzero(ans[0] != -1).flatten().tolist()) == [5, 14]
    assert sorted(torch.nonzero(ans[1] != -1).flatten().tolist()) == [5, 9, 12]


def test_vtm_pairs_known_answer():
    np.random.seed(88)
    vi, ti, tr = R.vtm_pairs(4, 4)
    assert ti.reshape(4, 4)[:, 1:].tolist() == [[2, 3, 1], [3, 0, 2], [3, 1, 0], [0, 1, 2]]
    assert tr.reshape(4, 4)[:, 0].all() and not tr.reshape(4, 4)[:, 1:].any()


def test_lr_schedule(golden_dir):
    g = _load(golden_dir, "ints")
    lrs = [max(1e-8, 2e-5 * R.warmup_linear_factor(s, 100)) for s in range(110)]
    np.testing.assert_allclose(lrs, g["lr_max_iter100_lr2e-5"], rtol=1e-12)


def test_param_groups_and_state_spec(golden_dir):
    g = _load(golden_dir, "agent")
    t = _load(golden_dir, "tiny2l_b2")
    spec = R.state_spec("tiny", layers=2)
    ref_shapes = dict(zip(t["keys"].tolist(), t["shapes"].tolist()))
    for k, s in spec.items():
        assert str(tuple(s)) == ref_shapes[k], k
    extra = set(ref_shapes) - set(spec)
    assert all(k.endswith("relative_position_index") or k == "fc_mtm.predictions.decoder.bias" for k in extra)
    for i in range(4):
        for n in g[f"group{i}"].tolist():
            assert R.param_group_of(n) == i, n
    assert g["group_sizes"].tolist() == [81, 23, 90, 27]


def _run_case(golden_dir, name, with_grads):
    g = _load(golden_dir, name)
    swin, bert, B, S, heads, T, X = (g["meta"].tolist() + ["5", "32"])[:7]
    B, S, heads, T, X = int(B), int(S), int(heads), int(T), int(X)
    bc = BERT_CFGS[bert]
    P = R.filled_params(swin, hidden=bc["hidden"], layers=bc["layers"], ffn=bc["ffn"], vocab=bc["vocab"])
    if with_grads:
        for k, v in P.items():
            v.requires_grad_(True)
    batch = make_batch(B, T=T, S=S, X=X, vocab=bc["vocab"])
    torch.manual_seed(88)
    batch["txt"], batch["ans_mtm"] = R.masking(batch["txt"])
    assert (batch["txt"].numpy() == g["txt"]).all() and (batch["ans_mtm"].numpy() == g["ans_mtm"]).all()
    np.random.seed(88)
    taps = {}
    out = R.pretrain_forward(P, batch, swin, heads, taps=taps)
    assert (out["ans_vtm"].numpy() == g["ans_vtm"]).all()
    cols = torch.from_numpy(g["cols"])
    np.testing.assert_allclose(out["out_mtm"][:, :, cols].detach().numpy(), g["out_mtm_cols"], atol=1e-5)
    np.testing.assert_allclose(out["out_vtm"][:, :, cols].detach().numpy(), g["out_vtm_cols"], atol=1e-5)
    np.testing.assert_allclose(torch.logsumexp(out["out_mtm"], -1).detach().numpy(), g["out_mtm_lse"], atol=1e-5)
    assert (out["out_mtm"].argmax(-1).numpy() == g["out_mtm_argmax"]).mean() > 0.999
    np.testing.assert_allclose(sub(taps["f_img"]), g["f_img_sub"], atol=1e-5)
    np.testing.assert_allclose(sub(taps["f_txt"]), g["f_txt_sub"], atol=1e-5)
    for k in ("patch_embed", "stage0", "stage1", "stage2", "stage3"):
        np.testing.assert_allclose(sub(taps[k]), g[f"tap_{k}_sub"], atol=2e-5)
        np.testing.assert_allclose(stats(taps[k]), g[f"tap_{k}_stats"], rtol=1e-4, atol=1e-6)
    l_mtm, l_vtm = R.pretrain_loss(out)
    np.testing.assert_allclose([l_mtm.item(), l_vtm.item()], g["loss"], atol=1e-5)
    if with_grads:
        (l_mtm + l_vtm).backward()
        ref = dict(zip(g["grad_norm_keys"].tolist(), g["grad_norm_vals"].tolist()))
        for k, v in ref.items():
            if v < 0:
                assert P[k].grad is None, k                 # emb_task, enc_img.emb_odr: unused
            else:
                assert abs(P[k].grad.double().norm().item() - v) <= 1e-4 * v + 1e-7, k
        for k in g.files:
            if k.startswith("grad_sub::"):
                np.testing.assert_allclose(sub(P[k[10:]].grad, 2048), g[k], atol=1e-6, rtol=1e-3)


def test_micro_forward_backward(golden_dir):
    _run_case(golden_dir, "micro_b2", True)


def test_micro_b5_forward(golden_dir):
    _run_case(golden_dir, "micro_b5", False)


def test_tiny2l_forward_backward(golden_dir):
    _run_case(golden_dir, "tiny2l_b2", True)


@pytest.mark.parametrize("case", ["micro_b1_t4_x33", "micro_b3_t6_x20"])
def test_odd_shapes_forward_backward(golden_dir, case):
    """Batch 1 (no VTM negatives) with the shipped json's 4 frames / 33 text positions; 6 frames (the frame-embedding
    maximum) with 20 text positions at batch 3."""
    _run_case(golden_dir, case, True)


def test_micro12_s384_forward_backward(golden_dir):
    """BASELINE config 4 geometry (384^2 frames, window (8,12,12) -> (5,12,12)) at micro widths."""
    _run_case(golden_dir, "micro12_s384_b2", True)


def test_swin_shapes_pad_branches(golden_dir):
    g = _load(golden_dir, "swin_shapes")
    P = {k: v for k, v in R.filled_params("micro", hidden=128, layers=0, ffn=512, vocab=64).items()}
    for T, S in ((5, 64), (4, 96), (1, 224), (6, 224)):
        x = torch.randn(1, 3, T, S, S, generator=torch.Generator().manual_seed(3))
        y = R.swin_forward(P, "enc_img.swin", x, "micro")
        np.testing.assert_allclose(sub(y, 2048), g[f"T{T}_S{S}_sub"], atol=2e-5)


def _check_grads(P, g):
    ref = dict(zip(g["grad_norm_keys"].tolist(), g["grad_norm_vals"].tolist()))
    for k, v in ref.items():
        if v < 0:
            assert P[k].grad is None, k
        else:
            assert abs(P[k].grad.double().norm().item() - v) <= 1e-4 * v + 2e-6, k   # fc.3.bias: exact value 0
    for k in g.files:
        if k.startswith("grad_sub::"):
            np.testing.assert_allclose(sub(P[k[10:]].grad, 2048), g[k], atol=2e-6, rtol=1e-3)


def _variant_params(bert, swin, extra=()):
    bc = BERT_CFGS[bert]
    P = R.filled_params(swin, hidden=bc["hidden"], layers=bc["layers"], ffn=bc["ffn"], vocab=bc["vocab"])
    for k, shp in extra:
        P[k] = R.fill_tensor(k, shp)
    for v in P.values():
        v.requires_grad_(True)
    return P, bc


def test_task_specific_variant(golden_dir):
    """LAVENDER_Pretrain (main_pretrain_task_specific.py:124-177): MLM logits + (B, O) matching scores, losses, grads."""
    g = _load(golden_dir, "ts_micro_b5")
    swin, bert, B, S, heads, temp = g["meta"].tolist()
    B, heads, temp = int(B), int(heads), float(temp)
    H = BERT_CFGS[bert]["hidden"]
    P, bc = _variant_params(bert, swin, [("fc.1.weight", (2 * H, H)), ("fc.1.bias", (2 * H,)), ("fc.3.weight", (1, 2 * H)),
                                         ("fc.3.bias", (1,))])
    P.pop("emb_task")                                       # this variant has no task-token table
    assert set(g["keys"].tolist()) - set(P) <= {"fc_mtm.predictions.decoder.bias"} | {k for k in g["keys"].tolist() if "relative_position_index" in k or "position_ids" in k}
    batch = make_batch(B, vocab=bc["vocab"])
    torch.manual_seed(88)
    batch["txt"], batch["ans_mtm"] = R.masking(batch["txt"])
    assert (batch["txt"].numpy() == g["txt"]).all() and (batch["ans_mtm"].numpy() == g["ans_mtm"]).all()
    np.random.seed(88)
    out = R.pretrain_ts_forward(P, batch, swin, heads, temp)
    cols = torch.from_numpy(g["cols"])
    np.testing.assert_allclose(out["out_mtm"][:, :, cols].detach().numpy(), g["out_mtm_cols"], atol=1e-5)
    np.testing.assert_allclose(out["out_vtm"].detach().numpy(), g["out_vtm"], atol=2e-5)
    assert (out["ans_vtm"].numpy() == g["ans_vtm"]).all()
    l_mtm, l_vtm = R.pretrain_ts_loss(out)
    np.testing.assert_allclose([l_mtm.item(), l_vtm.item()], g["loss"], atol=1e-5)
    (l_mtm + l_vtm).backward()
    _check_grads(P, g)


def test_retrieval_variant(golden_dir):
    """LAVENDER_Retrieval_MLM (main_retrieval_mlm.py:50-91): B x B pair order, labels from vid equality, logits, grads."""
    g = _load(golden_dir, "retr_micro_b3")
    swin, bert, B, S, heads = g["meta"].tolist()
    B, heads = int(B), int(heads)
    P, bc = _variant_params(bert, swin)
    batch = make_batch(B, vocab=bc["vocab"], seed=4)
    assert (batch["txt"].numpy() == g["txt"]).all()
    batch["vid"] = g["vid"].tolist()
    out, ans = R.retrieval_forward(P, batch, swin, heads)
    assert (ans.numpy() == g["ans"]).all()
    assert (ans[:, -1].view(B, B).numpy() == np.where(np.equal.outer(g["vid"], g["vid"]), 2995, 6270)).all()
    cols = torch.from_numpy(g["cols"])
    np.testing.assert_allclose(out[:, :, cols].detach().numpy(), g["out_cols"], atol=1e-5)
    np.testing.assert_allclose(torch.logsumexp(out, -1).detach().numpy(), g["out_lse"], atol=1e-5)
    ls = torch.nn.functional.cross_entropy(out.flatten(0, 1), ans.flatten(), ignore_index=-1)
    np.testing.assert_allclose(ls.item(), g["loss"][0], atol=1e-5)
    ls.backward()
    P["emb_task"].grad = None
    _check_grads(P, g)


def test_retrieval_eval_two_phase(golden_dir):
    """LAVENDER_RetrievalMlmEval (eval_retrieval_mlm.py:10-47): clip-averaged video features, then all (caption, video) pairs."""
    g = _load(golden_dir, "retr_eval_micro")
    swin, bert, B, Cl, heads, T = g["meta"].tolist()
    B, Cl, heads, T = int(B), int(Cl), int(heads), int(T)
    bc = BERT_CFGS[bert]
    P = R.filled_params(swin, hidden=bc["hidden"], layers=bc["layers"], ffn=bc["ffn"], vocab=bc["vocab"])
    b = make_batch(B * Cl, T=T, vocab=bc["vocab"], seed=9)
    img = b["img"].view(B, Cl, T, 3, 224, 224)
    txt, mask = b["txt"][:B], b["mask"][:B]
    assert (txt.numpy() == g["txt"]).all()
    with torch.no_grad():
        f_img, m_img, f_txt = R.retrieval_eval_feat(P, img, txt, swin)
        pi = torch.tensor([p for p in range(B) for q in range(B)]); qi = torch.tensor([q for p in range(B) for q in range(B)])
        out = R.retrieval_eval_cross(P, f_img[qi], m_img[qi], f_txt[pi], mask[pi], heads)
    np.testing.assert_allclose(sub(f_img), g["f_img_sub"], atol=1e-5)
    np.testing.assert_allclose(sub(f_txt), g["f_txt_sub"], atol=1e-5)
    np.testing.assert_allclose(out[:, :, torch.from_numpy(g["cols"])].numpy(), g["out_cols"], atol=1e-5)
    np.testing.assert_allclose(torch.logsumexp(out, -1).numpy(), g["out_lse"], atol=1e-5)


def test_captioning_seq2seq_variant(golden_dir):
    """LAVENDER_Captioning.encode_forward under the seq2seq mask (model.py:208-218, model_for_captioning.py:54-95): the oracle
    against the vectors captured from the real reference (logits, loss, gradient norms); the (B, L, L) mask rows bit-exact."""
    g = _load(golden_dir, "cap_micro_b2")
    P, bc = _variant_params("micro", "micro")
    B = 2
    batch = make_batch(B, vocab=bc["vocab"], seed=6)
    torch.manual_seed(88)
    batch["txt"], ans = R.masking(batch["txt"])
    assert (batch["txt"].numpy() == g["txt"]).all() and (ans.numpy() == g["ans"]).all()
    m3 = R.attn_mask(torch.ones(B, 250, dtype=torch.long), batch["mask"], "seq2seq")
    assert (m3[:, [0, 249, 250, 260, 281]].numpy() == g["mask_rows"]).all()
    out = R.captioning_encode_forward(P, dict(batch, ans_mtm=ans), "micro", bc["heads"])
    cols = torch.from_numpy(g["cols"])
    np.testing.assert_allclose(out["out"][:, :, cols].detach().numpy(), g["out_cols"], atol=1e-5)
    ls = torch.nn.functional.cross_entropy(out["out"].flatten(0, 1), ans.flatten(), ignore_index=-1)
    assert abs(ls.item() - g["loss"][0]) < 1e-5
    ls.backward()
    _check_grads(P, g)


def test_swin_pad_branch_gradients(golden_dir):
    """Oracle vs the reference on padded token grids, forward AND backward (fixture: tests/golden/make_goldens_pad.py):
    5x64^2 / 4x96^2 (window padding) and 2x40^2 (odd H / W in PatchMerging, video_swin.py:273-276)."""
    g = _load(golden_dir, "swin_pad_grads")
    for B, T, S in ((2, 5, 64), (2, 2, 40), (1, 4, 96)):
        tag = f"B{B}_T{T}_S{S}"
        P = {k: v.requires_grad_(True) for k, v in R.filled_params("micro", hidden=128, layers=0, ffn=512, vocab=64).items()}
        x = torch.randn(B, 3, T, S, S, generator=torch.Generator().manual_seed(3))
        y = R.swin_forward(P, "enc_img.swin", x, "micro")
        assert tuple(y.shape) == tuple(g[f"{tag}_shape"])
        np.testing.assert_allclose(sub(y, 2048), g[f"{tag}_sub"], atol=2e-5)
        w = torch.randn(y.shape, generator=torch.Generator().manual_seed(11))
        (y * w).sum().backward()
        for k, n in zip(g[f"{tag}_grad_keys"].tolist(), g[f"{tag}_grad_norms"].tolist()):
            assert abs(P["enc_img.swin." + k].grad.double().norm().item() - n) <= 1e-4 * n + 2e-6, k
        for k in g.files:
            if k.startswith(f"{tag}_grad_sub::"):
                np.testing.assert_allclose(sub(P["enc_img.swin." + k.split("::")[1]].grad, 1024), g[k], atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("name", ["qaoe_micro_b3", "qamc_micro_b3"])
def test_qa_mlm_variants(golden_dir, name):
    """LAVENDER_QAOE_MLM (main_qaoe_mlm_lsmdc_fib.py:79-93) / LAVENDER_QAMC_MLM (main_qamc_mlm.py:124-140): logits at sampled
    vocabulary columns, log-sum-exp, loss, accuracies and every gradient norm against the reference fixture."""
    g = _load(golden_dir, name)
    swin, bert, B, S, heads, X = g["meta"].tolist()
    P, bc = _variant_params(bert, swin)
    batch = make_batch(int(B), X=int(X), vocab=bc["vocab"], seed=6)
    batch["txt"], batch["mask"], batch["mask_ans"] = torch.from_numpy(g["txt"]), (torch.from_numpy(g["txt"]) != 0).long(), torch.from_numpy(g["mask_ans"])
    out, ans = R.qa_mlm_forward(P, batch, swin, int(heads))
    np.testing.assert_allclose(out[:, :, torch.from_numpy(g["cols"])].detach().numpy(), g["out_cols"], atol=2e-5)
    np.testing.assert_allclose(torch.logsumexp(out, -1).detach().numpy(), g["out_lse"], atol=2e-5)
    ls = torch.nn.functional.cross_entropy(out.flatten(0, 1), ans.flatten(), ignore_index=-1)
    assert abs(ls.item() - g["loss"][0]) < 1e-5
    assert R.qa_top_k_acc(out.detach(), ans, 1) == g["ac_1"].tolist() and R.qa_top_k_acc(out.detach(), ans, 5) == g["ac_5"].tolist()
    ls.backward()
    _check_grads(P, g)


def test_retmc_mlm_variant(golden_dir):
    """LAVENDER_RetMC_MLM (main_retmc_mlm.py:89-113) + the agent's candidate accuracy (:130-140) against the reference fixture."""
    g = _load(golden_dir, "retmc_micro_b2")
    swin, bert, B, O, heads, X = g["meta"].tolist()
    B, O, heads, X = int(B), int(O), int(heads), int(X)
    P, bc = _variant_params(bert, swin)
    txt = torch.from_numpy(g["txt"])
    batch = {"img": make_batch(B, vocab=bc["vocab"], seed=13)["img"], "txt": txt, "mask": (txt != 0).long(), "mask_ans": torch.from_numpy(g["mask_ans"])}
    out, ans = R.retmc_mlm_forward(P, batch, swin, heads)
    np.testing.assert_allclose(out[:, :, torch.from_numpy(g["cols"])].detach().numpy(), g["out_cols"], atol=2e-5)
    np.testing.assert_allclose(torch.logsumexp(out, -1).detach().numpy(), g["out_lse"], atol=2e-5)
    ls = torch.nn.functional.cross_entropy(out.flatten(0, 1), ans.flatten(), ignore_index=-1)
    assert abs(ls.item() - g["loss"][0]) < 1e-5
    assert R.retmc_acc(out.detach().softmax(-1), ans) == g["acc"].tolist()
    ls.backward()
    _check_grads(P, g)


def test_pipeline_oracle_against_the_reference_dataset_class(golden_dir):
    """oracle/pipeline_ref.py against outputs of the REFERENCE's own dataset.Dataset_Base (sampling, temporal_sample, str2img,
    vid_center_crop, vid_rand_crop, get_img_or_video with img_transform == ['vid_rand_crop']) captured by
    tests/golden/make_goldens_pipeline_ref.py on the two fixture rows: integers and decoded bytes bit-exact, fp32 tensors exact."""
    import random
    import zlib
    from oracle import pipeline_ref as PR
    g = _load(golden_dir, "pipeline_ref_pin")

    def sub(a, n=4096, seed=7):                       # the generator's numpy sub-sampler (helpers.sub is the torch one)
        flat = np.asarray(a).reshape(-1)
        return flat[np.random.RandomState(seed + flat.size % 9973).permutation(flat.size)[:n]]

    out, pos = g["sampling_out"], 0
    for s, e, n in g["sampling_args"].tolist():
        got = PR.sampling(s, e, n)
        assert got == out[pos:pos + len(got)].tolist(), (s, e, n)
        pos += len(got)
    assert pos == len(out)
    pe = pt = 0
    for ci, (L, sf) in enumerate(g["temporal_cases"].tolist()):
        ne, nt = int(g["temporal_eval_len"][ci]), int(g["temporal_train_len"][ci])
        assert PR.temporal_sample(list(range(L)), sf, False) == g["temporal_eval_out"][pe:pe + ne].tolist(), (L, sf)
        random.seed(100 + ci)
        assert PR.temporal_sample(list(range(L)), sf, True, random) == g["temporal_train_out"][pt:pt + nt].tolist(), (L, sf)
        pe += ne; pt += nt
    tsvp = os.path.join(golden_dir, "msrvtt_2rows.tsv")
    offs = [int(x) for x in open(os.path.join(golden_dir, "msrvtt_2rows.lineidx"))]

    def same(x, name):
        x = x.numpy()
        assert np.array_equal(sub(x), g[name + "_sub"]), name
        assert np.array_equal(np.array([x.astype(np.float64).sum(), np.abs(x.astype(np.float64)).sum()]), g[name + "_sum"]), name

    for r, p in enumerate(offs):
        frames = PR.read_row(tsvp, p)[1:]
        for fi, b in enumerate(frames):
            rgb = np.array(PR.str2img(b))
            assert list(rgb.shape) == g[f"ref_{r}_{fi}_rgb_shape"].tolist()
            assert np.array_equal(sub(rgb), g[f"ref_{r}_{fi}_rgb_sub"])
            assert [int(rgb.astype(np.int64).sum()), zlib.adler32(rgb.tobytes())] == g[f"ref_{r}_{fi}_rgb_sum"].tolist()
        imgs = [PR.str2img(b) for b in frames[:4]]
        same(PR.vid_center_crop(list(imgs), 224), f"ref_vid_center_{r}")
        random.seed(21 + r)
        same(PR.vid_rand_crop(list(imgs), 224, random), f"ref_vid_rand_{r}")
        random.seed(9 + r)
        x = PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop"], "train", random, None)
        assert list(x.shape) == g[f"ref_sample_train_{r}_shape"].tolist()
        same(x, f"ref_sample_train_{r}")
        same(PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop"], "val", random, None), f"ref_sample_val_{r}")
        # three-entry transform list: random.choice (dataset.py:225) draws from the same stream between the frame sampling and the crop
        random.seed(33 + r)
        same(PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop", "vid_rand_crop", "vid_rand_crop"], "train", random, None), f"ref_sample_train2_{r}")
