"""Two optimizer steps of the REFERENCE Agent (tests/golden/agent_steps.npz: forward, both losses, backward, clip_grad_norm_,
AdamW with the reference's parameter groups, WarmupLinearLR; stochastic modules set to p = 0) against Agent_Pretrain_MLM on the
HIP engine with the same inputs: losses, learning rates and the parameter updates themselves (agent.py:13-43,96-140,235-250)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import BERT_CFGS, Tok, hf_cfg, make_args, make_batch, sub

pytestmark = pytest.mark.gpu


def test_two_agent_steps_match_the_reference(golden_dir):
    import lavender_amd as LA
    from oracle import lavender_ref as R
    g = np.load(os.path.join(golden_dir, "agent_steps.npz"))
    B = 2
    cfg = dict(hf_cfg("micro"), hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    args = make_args("micro", "micro", B, txt_backbone=cfg, fusion_encoder=cfg, tokenizer=cfg, lr=2e-5, decay=1e-3, max_iter=100,
                     max_grad_norm=1.0, vis_backbone_lr_mul=1)
    m = LA.LAVENDER_Pretrain_MLM(args, Tok())
    sd = m.state_dict()
    new = {k: R.fill_tensor(k, v.shape) for k, v in sd.items() if v.is_floating_point()}
    new["fc_mtm.predictions.decoder.bias"] = new["fc_mtm.predictions.bias"]
    m.load_state_dict(new, strict=False)
    m.cuda()
    for layer in m.enc_img.swin.layers:                    # stochastic depth off, as in the fixture
        for blk in layer.blocks:
            blk.drop_prob, blk.keep_prob = 0.0, 1.0
    ag = LA.Agent_Pretrain_MLM(args, m)
    before = {k: p.detach().float().cpu().clone() for k, p in m.named_parameters()}
    for step in range(2):
        batch = make_batch(B, vocab=BERT_CFGS["micro"]["vocab"], seed=1 + step)
        torch.manual_seed(88 + step)
        batch["txt"], batch["ans_mtm"] = R.masking(batch["txt"])
        np.random.seed(88 + step)
        out = ag.step(ag.prepare_batch(batch), True)
        ref_l = g["losses"][step]
        print(f"step {step}: loss {out['mtm']:.4f} {out['vtm']:.4f}  reference {ref_l[0]:.4f} {ref_l[1]:.4f}")
        assert abs(out["mtm"] - ref_l[0]) < 1e-2 and abs(out["vtm"] - ref_l[1]) < 1e-2
        lrs = [grp["lr"] for grp in ag.optzr.param_groups]
        assert np.allclose(sorted(set(lrs)), sorted(set(g["lrs"][step].tolist())), rtol=1e-6), (lrs, g["lrs"][step])
    torch.cuda.synchronize()
    after = {k: p.detach().float().cpu() for k, p in m.named_parameters()}
    keys, norms = g["step1_keys"].tolist(), g["step1_delta_norms"]
    # after the second step (the first runs at the 1e-8 floor of the warm-up): every parameter's total update has the reference's
    # size -- an Adam step is +-lr per element wherever the gradient is not tiny -- and the picked updates agree in direction
    bad = []
    for k, n in zip(keys, norms):
        if k not in after or k.endswith("attention.self.key.bias"):
            # key.bias: softmax is shift-invariant, the true gradient is 0.  The reference's fp32 round-off (~1e-12) is below Adam's
            # eps and gives a ~1e-9 update; bf16 round-off (~1e-6) is above it and gives a full +-lr step.  Either way the function
            # the model computes does not depend on this parameter.
            continue
        d = (after[k] - before[k]).double().norm().item()
        if n < 1e-12:
            assert d < 1e-9, (k, d)
            continue
        if abs(d - n) > 0.05 * n:
            bad.append((k, d, n))
    assert not bad, bad[:10]
    worst = 1.0
    for key in g.files:
        if key.startswith("step1_delta_sub::"):
            k = key.split("::")[1]
            a, b = sub(after[k] - before[k], 2048).astype(np.float64), g[key].astype(np.float64)
            cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            worst = min(worst, cos)
            print(f"{k}: cosine of the update vs the reference {cos:.4f}")
            assert cos > 0.99, (k, cos)                     # measured: >= 0.9998
    print("worst update cosine", worst)


def test_pretrain_agents_write_the_reference_checkpoint_names(tmp_path):
    """main_pretrain_task_specific.py:282-297 / main_pretrain_mlm.py:326: one file per (dataset, part, epoch),
    ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt -- parts of an epoch must not overwrite each other; the generic agents keep
    agent.py:164-180's ckpt_violet_{task}_{ep}.pt."""
    import lavender_amd as LA
    cfg = dict(hf_cfg("micro"), hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    args = make_args("micro", "micro", 2, txt_backbone=cfg, fusion_encoder=cfg, tokenizer=cfg, lr=2e-5, decay=1e-3, max_iter=100,
                     max_grad_norm=1.0, vis_backbone_lr_mul=1)
    args.path_output, args.task, args.dataset = str(tmp_path), "pretrain", ["webvid", "cc3m"]
    m = LA.LAVENDER_Pretrain_MLM(args, Tok()).cuda()
    ag = LA.Agent_Pretrain_MLM(args, m)
    ag.log["webvid"]["ls_mtm"].append(1.0)
    ag.save_model(0)
    ag.save_model(1, "webvid", 0)
    ag.save_model(1, "webvid", 1)
    names = sorted(os.listdir(tmp_path))
    # ... and nothing else: Agent_Pretrain.save_model of the reference writes no log.json (only agent.py:164-180 does)
    assert names == ["ckpt_violet_pretrain_init_0_0.pt", "ckpt_violet_pretrain_webvid_0_1.pt", "ckpt_violet_pretrain_webvid_1_1.pt"], names
    sd = torch.load(os.path.join(tmp_path, "ckpt_violet_pretrain_webvid_1_1.pt"))
    assert set(sd) == set(m.state_dict())
