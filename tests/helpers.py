"""Shared test helpers: seeded synthetic batches (same generator as tests/golden/make_goldens.py)."""
import numpy as np
import torch


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
    flat = t.detach().reshape(-1)
    g = torch.Generator().manual_seed(seed + flat.numel() % 9973)
    idx = torch.randperm(flat.numel(), generator=g)[:n]
    return flat[idx].float().cpu().numpy().astype(np.float32)


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().max().item(), t.pow(2).mean().sqrt().item()])


BERT_CFGS = {
    "micro": dict(hidden=128, layers=2, heads=4, ffn=512, vocab=8192),
    "b2l": dict(hidden=768, layers=2, heads=12, ffn=3072, vocab=30522),
    "b12l": dict(hidden=768, layers=12, heads=12, ffn=3072, vocab=30522),
}
