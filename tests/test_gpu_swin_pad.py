"""Swin pad branches on the PRODUCTION (bf16, HIP engine) path: token grids that are not window multiples
(video_swin.py:211-215,241-242) and odd H / W in PatchMerging (video_swin.py:273-276), forward and backward against the
reference fixture tests/golden/swin_pad_grads.npz and against the oracle on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import sub

pytestmark = pytest.mark.gpu

CASES = ((2, 5, 64), (2, 2, 40), (1, 4, 96))


@pytest.mark.parametrize("case", CASES)
def test_pad_geometry_forward_backward(golden_dir, case):
    from tests.helpers import build_filled_model
    from oracle import lavender_ref as R
    B, T, S = case
    tag = f"B{B}_T{T}_S{S}"
    g = np.load(os.path.join(golden_dir, "swin_pad_grads.npz"))
    x = torch.randn(B, 3, T, S, S, generator=torch.Generator().manual_seed(3))
    w = torch.randn(tuple(g[f"{tag}_shape"]), generator=torch.Generator().manual_seed(11))

    P = {k: v.requires_grad_(True) for k, v in R.filled_params("micro", hidden=128, layers=0, ffn=512, vocab=64).items()
         if k.startswith("enc_img.swin.")}
    yo = R.swin_forward(P, "enc_img.swin", x, "micro")
    (yo * w).sum().backward()

    m = build_filled_model("micro", "micro", B).eval()
    m.arena().zero_grad()
    tok, (b, D, h, wd) = m.enc_img.swin.forward_tokens(x.cuda(), frame_major=False)
    y = tok.view(b, D, h, wd, -1)
    assert tuple(y.shape) == tuple(g[f"{tag}_shape"])
    (y.float() * w.cuda()).sum().backward()
    torch.cuda.synchronize()

    d = np.abs(sub(y, 2048) - g[f"{tag}_sub"])
    scale = float(g[f"{tag}_stats"][2])                                  # rms of the reference output
    print(f"{tag}: output max|d| vs reference {d.max():.3e} mean {d.mean():.3e} (rms {scale:.3f})")
    assert d.max() < 6e-2 * max(scale, 1.0) and d.mean() < 1e-2 * max(scale, 1.0)      # bf16 activations, 4 stages (tier T3)
    do = (y.float().cpu() - yo.detach()).abs()
    assert do.max() < 6e-2 * max(scale, 1.0)

    norms = dict(zip(g[f"{tag}_grad_keys"].tolist(), g[f"{tag}_grad_norms"].tolist()))
    bad, worst = [], 0.0
    for name, p in m.enc_img.swin.named_parameters():
        a, ref = p.grad.float().cpu(), P["enc_img.swin." + name].grad
        if ref.norm() < 1e-7:
            assert a.norm() < 1e-3, name
            continue
        rel = ((a - ref).norm() / ref.norm()).item()
        cos = torch.nn.functional.cosine_similarity(a.flatten(), ref.flatten(), dim=0).item()
        worst = max(worst, rel)
        if not (rel < 0.05 and cos > 0.998 and abs(a.norm().item() - norms[name]) <= 0.05 * norms[name] + 1e-6):
            bad.append((name, rel, cos, a.norm().item(), norms[name]))
    print(f"{tag}: worst relative gradient error {worst:.4f}")
    assert not bad, bad[:12]


def test_fp32_validation_mode_odd_patch_merging(golden_dir):
    """The fp32-I/O validation path on the odd-H/W PatchMerging geometry (tier T2 bound, forward only)."""
    from tests.helpers import build_filled_model
    from lavender_amd import validate
    g = np.load(os.path.join(golden_dir, "swin_pad_grads.npz"))
    m = build_filled_model("micro", "micro", 2).eval()
    x = torch.randn(2, 3, 2, 40, 40, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        tok, (B, D, h, w) = validate.swin_tokens(m.enc_img.swin, x.cuda(), frame_major=False)
    d = np.abs(sub(tok.view(B, D, h, w, -1), 2048) - g["B2_T2_S40_sub"]).max()
    print(f"fp32 validation, 2x40^2: max|d| vs reference {d:.2e}")
    assert d <= 2e-4
