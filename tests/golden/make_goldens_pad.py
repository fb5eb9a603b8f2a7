"""Golden vectors for the Swin pad branches WITH gradients (CONTAINER ONLY -- needs /root/reference).

Token grids that are not window multiples (zero padding after norm1, crop after window_reverse: video_swin.py:211-215,
241-242) and odd H / W in PatchMerging (video_swin.py:273-276): the reference SwinTransformer3D at micro width, parameters
filled from their state_dict keys, seeded input, scalar loss = sum(out * fixed weights).  Writes swin_pad_grads.npz with
sub-sampled outputs, every parameter's gradient norm and sub-samples of a few gradients.

    python tests/golden/make_goldens_pad.py
"""
import sys
sys.dont_write_bytecode = True
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402
from make_goldens import R, sub, stats  # noqa: E402

CASES = ((2, 5, 64), (2, 2, 40), (1, 4, 96))          # (B, T, S): 16^2 grid -> pad 21; 10^2 -> pad 14, merges 5 -> 3 -> 2; 24^2 -> 28
PICK = ["patch_embed.proj.weight", "layers.0.blocks.0.attn.qkv.bias", "layers.0.blocks.1.attn.qkv.weight",
        "layers.0.blocks.1.attn.relative_position_bias_table", "layers.0.downsample.norm.weight",
        "layers.1.blocks.0.attn.qkv.bias", "layers.1.downsample.reduction.weight", "layers.2.blocks.0.mlp.fc1.weight"]


def loss_weights(shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(11))


def run(ref):
    os.environ["LAV_SWIN_SIZE"] = "micro"
    args = ref.EasyDict(vis_backbone_size="base", size_img=224, vis_backbone_init="random", kinetics=400)
    sw = ref.VS.get_vidswin_model(args)
    sd = sw.state_dict()
    sw.load_state_dict({k: R.fill_tensor("enc_img.swin." + k, v.shape) for k, v in sd.items() if v.is_floating_point()},
                       strict=False)
    sw.eval()
    res = {}
    for B, T, S in CASES:
        tag = f"B{B}_T{T}_S{S}"
        x = torch.randn(B, 3, T, S, S, generator=torch.Generator().manual_seed(3))
        sw.zero_grad()
        y = sw(x).permute(0, 2, 3, 4, 1)
        w = loss_weights(y.shape)
        (y * w).sum().backward()
        # oracle-vs-reference check, forward and backward, before anything is written
        P = {"enc_img.swin." + k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sw.state_dict().items()}
        yo = R.swin_forward(P, "enc_img.swin", x, "micro")
        (yo * w).sum().backward()
        d = (y - yo).abs().max().item()
        gd = max(((p.grad - P["enc_img.swin." + k].grad).abs().max() / (p.grad.abs().max() + 1e-12)).item()
                 for k, p in sw.named_parameters())
        print(f"   {tag}: out {tuple(y.shape)} oracle max|d| {d:.2e}, worst relative gradient difference {gd:.2e}")
        assert d < 2e-5 and gd < 1e-4
        res[f"{tag}_shape"] = np.array(y.shape)
        res[f"{tag}_sub"] = sub(y, 2048)
        res[f"{tag}_stats"] = stats(y)
        res[f"{tag}_grad_keys"] = np.array([k for k, _ in sw.named_parameters()])
        res[f"{tag}_grad_norms"] = np.array([p.grad.double().norm().item() for _, p in sw.named_parameters()])
        named = dict(sw.named_parameters())
        for k in PICK:
            res[f"{tag}_grad_sub::{k}"] = sub(named[k].grad, 1024)
    np.savez_compressed(f"{HERE}/swin_pad_grads.npz", **res)
    print("written", f"{HERE}/swin_pad_grads.npz")


if __name__ == "__main__":
    torch.set_num_threads(8)
    run(MG.import_reference())
