"""Golden vectors of the reference Agent's optimizer steps (CONTAINER ONLY -- needs /root/reference).

Two full `Agent_Pretrain_MLM.step(batch, is_train=True)` calls of the real reference on the CPU (micro model, B = 2): forward,
both losses, backward, clip_grad_norm_, AdamW with the reference's parameter groups, WarmupLinearLR (agent.py:13-43,96-140,
235-250; main_pretrain_mlm.py:145-176).  Stochastic layers are neutralised on the MODULES (nn.Dropout.p = 0, DropPath.drop_prob = 0)
so that the step is a function of the inputs; nothing in the reference's source is changed.  Writes agent_steps.npz: the losses
of both steps, the learning rates the scheduler leaves behind, every parameter's update norm after step 1 and after step 2 and
sub-samples of a few updates.

    python tests/golden/make_goldens_agent.py
"""
import sys
sys.dont_write_bytecode = True
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402
from make_goldens import R, make_batch, sub  # noqa: E402

PICK = ["enc_img.swin.patch_embed.proj.weight", "enc_img.swin.layers.0.blocks.1.attn.qkv.weight",
        "enc_img.swin.layers.0.blocks.1.attn.relative_position_bias_table", "enc_img.swin.layers.2.blocks.0.mlp.fc1.weight",
        "enc_img.emb_pos", "enc_txt.emb_txt.word_embeddings.weight", "trsfr.layer.0.attention.self.query.weight",
        "trsfr.layer.1.output.LayerNorm.weight", "fc_mtm.predictions.bias", "fc_mtm.predictions.transform.dense.weight"]


def main():
    ref = MG.import_reference()
    B = 2
    m, keys, args = MG.build_reference(ref, "micro", "micro", B)
    args.update(lr=2e-5, decay=1e-3, max_iter=100, max_grad_norm=1.0, deepspeed=False, vis_backbone_lr_mul=1, dataset=["x"],
                logging_steps=10, path_output="/tmp/lav_golden/out", task="pretrain")
    n_drop = 0
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
            n_drop += 1
        if type(mod).__name__ == "DropPath":
            mod.drop_prob = 0.0
            n_drop += 1
    inner = m.trsfr.enc
    m.trsfr = inner                                        # the Agent's group rules match unwrapped names
    ag = ref.PM.Agent_Pretrain_MLM(args, m)

    class _Enc(torch.nn.Module):
        def __init__(s, enc):
            super().__init__()
            s.enc = enc

        def forward(s, feat, mask, output_attentions=False):
            return {"last_hidden_state": s.enc(feat, mask).last_hidden_state, "attentions": None}
    m.trsfr = _Enc(inner)                                  # re-wrapped for the forward (as make_goldens.build_reference does)
    unwrap = lambda k: k.replace("trsfr.enc.", "trsfr.")
    res = {"n_neutralised": np.array(n_drop)}
    before = {unwrap(k): p.detach().clone() for k, p in m.named_parameters()}
    losses, lrs = [], []
    for step in range(2):
        batch = make_batch(B, vocab=MG.BERT_CFGS["micro"]["vocab_size"], seed=1 + step)
        torch.manual_seed(88 + step)
        txt_m, ans = R.masking(batch["txt"])
        batch["txt"], batch["ans_mtm"] = txt_m, ans
        np.random.seed(88 + step)
        out = ag.step(batch, is_train=True)
        losses.append([out["mtm"], out["vtm"]])
        lrs.append([g["lr"] for g in ag.optzr.param_groups])
        after = {unwrap(k): p.detach().clone() for k, p in m.named_parameters()}
        res[f"step{step}_keys"] = np.array(list(after.keys()))
        res[f"step{step}_delta_norms"] = np.array([(after[k] - before[k]).double().norm().item() for k in after])
        for k in PICK:
            res[f"step{step}_delta_sub::{k}"] = sub(after[k] - before[k], 2048)
        print(f"   step {step}: loss mtm {out['mtm']:.5f} vtm {out['vtm']:.5f}  lr {lrs[-1]}  total update norm "
              f"{np.sqrt((res[f'step{step}_delta_norms'] ** 2).sum()):.6f}")
    res["losses"] = np.array(losses)
    res["lrs"] = np.array(lrs)
    np.savez_compressed(f"{HERE}/agent_steps.npz", **res)
    print("written", f"{HERE}/agent_steps.npz", os.path.getsize(f"{HERE}/agent_steps.npz"), "bytes;", n_drop, "stochastic modules neutralised")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
