"""CPU oracle for the LAVENDER pretrain hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 *restatement* of the reference algorithm
(microsoft/LAVENDER: Video-Swin encoder + BERT-style fusion encoder + MLM
head + pretrain step), written functionally over a flat ``{state_dict key:
tensor}`` mapping.  It exists so the HIP path can be checked on any box:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it -- never ``lavender_amd`` itself;
  * parity status: PINNED.  ``tests/golden/make_goldens.py`` imports the real
    reference (container only), runs it on key-hashed deterministic weights
    and stores its outputs under ``tests/golden/*.npz``;
    ``tests/test_oracle_golden.py`` checks this file against those vectors
    (fp32, max|d| <= 1e-5 on logits; integer paths bit-exact).
  * The fusion encoder / MLM head / text embedding arithmetic lives in the
    third-party ``transformers`` package (un-pinned by the reference,
    README.md:28); it is restated here from the published BERT algorithm
    (post-LN encoder, erf-GELU, LN eps 1e-12) and anchored on the reference's
    call sites model.py:100-110,152-165,223-243 and main_pretrain_mlm.py:46-48.

Every function cites the reference file:line it follows (paths relative to
the reference root).
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# size table: visbackbone/swin_{tiny,base,large}.py (only the backbone keys
# that video_swin.py:616-634 reads)
# --------------------------------------------------------------------------
SWIN_SIZES = {
    "micro": dict(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), window_size=(8, 7, 7)),
    "micro12": dict(embed_dim=32, depths=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), window_size=(8, 12, 12)),  # test size: Large-384 geometry
    "tiny": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=(8, 7, 7)),
    "base": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=(8, 7, 7)),
    "large": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48), window_size=(8, 12, 12)),
}
DROP_PATH_RATE = 0.2  # hard-coded in video_swin.py:630


# --------------------------------------------------------------------------
# deterministic parameter fill: every tensor is a pure
# function of its state_dict key, so reference / oracle / HIP hold the same
# weights without shipping any
# --------------------------------------------------------------------------
def fill_tensor(key, shape):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32) * 0.02
    leaf = key.rsplit(".", 1)[-1]
    is_norm = ("norm" in key.lower()) and leaf == "weight"
    if is_norm:
        t = t + 1.0
    return t


def fill_state_dict(spec):
    """spec: ordered {key: shape}.  Integer buffers (relative_position_index) are
    rebuilt from geometry, not filled."""
    return {k: fill_tensor(k, s) for k, s in spec.items()}


# --------------------------------------------------------------------------
# integer geometry
# --------------------------------------------------------------------------
def use_window(dims, window, shift=None):
    """get_window_size, video_swin.py:93-106: clamp window to the input, zero the
    shift on clamped axes."""
    w = [min(d, ws) if d <= ws else ws for d, ws in zip(dims, window)]
    if shift is None:
        return tuple(w)
    s = [0 if d <= ws else sh for d, ws, sh in zip(dims, window, shift)]
    return tuple(w), tuple(s)


def rel_pos_index(window):
    """relative_position_index buffer, video_swin.py:118-135.
    index[i,j] = code(i) - code(j) + const with code = d*(2wh-1)(2ww-1) + h*(2ww-1) + w."""
    wd, wh, ww = window
    d, h, w = np.meshgrid(np.arange(wd), np.arange(wh), np.arange(ww), indexing="ij")
    code = (d * (2 * wh - 1) * (2 * ww - 1) + h * (2 * ww - 1) + w).reshape(-1)
    const = (wd - 1) * (2 * wh - 1) * (2 * ww - 1) + (wh - 1) * (2 * ww - 1) + (ww - 1)
    return torch.from_numpy(code[:, None] - code[None, :] + const).long()


def region_ids(D, H, W, window, shift):
    """img_mask of compute_mask, video_swin.py:290-299, in closed form: along an
    axis of length n the region is (x >= n-w) + (x >= n-s)."""
    def axis(n, w, s):
        x = np.arange(n)
        if s == 0:
            # slice(-w), slice(-w,-0) (empty), slice(-0,None) (everything): the last
            # assignment wins, so the whole axis is region 2
            return np.full(n, 2)
        return (x >= n - w).astype(np.int64) + (x >= n - s).astype(np.int64)
    rd, rh, rw = axis(D, window[0], shift[0]), axis(H, window[1], shift[1]), axis(W, window[2], shift[2])
    return torch.from_numpy(rd[:, None, None] * 9 + rh[None, :, None] * 3 + rw[None, None, :]).float()


def partition(x, window):
    """window_partition, video_swin.py:82-86."""
    B, D, H, W, C = x.shape
    wd, wh, ww = window
    x = x.reshape(B, D // wd, wd, H // wh, wh, W // ww, ww, C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, wd * wh * ww, C)


def unpartition(wins, window, B, D, H, W):
    """window_reverse, video_swin.py:88-91."""
    wd, wh, ww = window
    x = wins.reshape(B, D // wd, H // wh, W // ww, wd, wh, ww, -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def shift_mask(D, H, W, window, shift):
    """compute_mask, video_swin.py:290-305: (nW, N, N) of {-100, 0}."""
    reg = region_ids(D, H, W, window, shift)[None, ..., None]
    mw = partition(reg, window).squeeze(-1)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


# --------------------------------------------------------------------------
# Swin
# --------------------------------------------------------------------------
def _ln(x, P, pre, eps):
    return F.layer_norm(x, (x.shape[-1],), P[pre + ".weight"], P[pre + ".bias"], eps)


def _lin(x, P, pre):
    return F.linear(x, P[pre + ".weight"], P.get(pre + ".bias"))


def patch_embed(P, pre, x):
    """PatchEmbed3D.forward, video_swin.py:388-405. x: (B,3,T,H,W)."""
    ph, pw = 4, 4
    if x.shape[-1] % pw:
        x = F.pad(x, (0, pw - x.shape[-1] % pw))
    if x.shape[-2] % ph:
        x = F.pad(x, (0, 0, 0, ph - x.shape[-2] % ph))
    x = F.pad(x, (0, 0, 0, 0, 0, 1))                       # one zero frame at the END of T (:396)
    x = F.conv3d(x, P[pre + ".proj.weight"], P[pre + ".proj.bias"], stride=(1, 4, 4))
    x = x.permute(0, 2, 3, 4, 1)                           # channels-last tokens
    return _ln(x, P, pre + ".norm", 1e-5)


def window_attention(P, pre, xw, heads, cfg_window, mask):
    """WindowAttention3D.forward, video_swin.py:145-170."""
    Bw, N, C = xw.shape
    hd = C // heads
    qkv = _lin(xw, P, pre + ".qkv").reshape(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    att = q @ k.transpose(-2, -1)
    idx = rel_pos_index(cfg_window)[:N, :N].reshape(-1)
    bias = P[pre + ".relative_position_bias_table"][idx].reshape(N, N, heads).permute(2, 0, 1)
    att = att + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        att = (att.reshape(Bw // nW, nW, heads, N, N) + mask[None, :, None]).reshape(-1, heads, N, N)
    att = att.softmax(-1)
    out = (att @ v).transpose(1, 2).reshape(Bw, N, C)
    return _lin(out, P, pre + ".proj")


def swin_block(P, pre, x, heads, cfg_window, cfg_shift, dp=None):
    """SwinTransformerBlock3D.forward, video_swin.py:204-261.
    dp: None or (scale_attn, scale_mlp) per-sample stochastic-depth factors (B,)."""
    B, D, H, W, C = x.shape
    window, shift = use_window((D, H, W), cfg_window, cfg_shift)
    h = _ln(x, P, pre + ".norm1", 1e-5)
    pd = (window[0] - D % window[0]) % window[0]
    pb = (window[1] - H % window[1]) % window[1]
    pr = (window[2] - W % window[2]) % window[2]
    h = F.pad(h, (0, 0, 0, pr, 0, pb, 0, pd))             # zeros AFTER the norm (:209-215)
    Dp, Hp, Wp = h.shape[1:4]
    mask = None
    if any(shift):
        h = torch.roll(h, (-shift[0], -shift[1], -shift[2]), (1, 2, 3))
        mask = shift_mask(Dp, Hp, Wp, window, shift)
    a = window_attention(P, pre + ".attn", partition(h, window), heads, cfg_window, mask)
    h = unpartition(a, window, B, Dp, Hp, Wp)
    if any(shift):
        h = torch.roll(h, shift, (1, 2, 3))
    h = h[:, :D, :H, :W]
    if dp is not None:
        h = h * dp[0].view(B, 1, 1, 1, 1)
    x = x + h
    m = _lin(F.gelu(_lin(_ln(x, P, pre + ".norm2", 1e-5), P, pre + ".mlp.fc1")), P, pre + ".mlp.fc2")
    if dp is not None:
        m = m * dp[1].view(B, 1, 1, 1, 1)
    return x + m


def patch_merge(P, pre, x):
    """PatchMerging.forward, video_swin.py:271-287."""
    H, W = x.shape[2], x.shape[3]
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    return F.linear(_ln(x, P, pre + ".norm", 1e-5), P[pre + ".reduction.weight"])


def swin_forward(P, pre, x, size, droppath=None, taps=None):
    """SwinTransformer3D.forward, video_swin.py:468-480 (+BasicLayer.forward :350-368).
    x: (B,3,T,H,W) -> channels-last (B,T,h,w,8E).  droppath: optional list (per block)
    of (scale_attn, scale_mlp) tensors."""
    cfg = SWIN_SIZES[size]
    win = cfg["window_size"]
    shift = tuple(i // 2 for i in win)
    x = patch_embed(P, pre + ".patch_embed", x)
    if taps is not None:
        taps["patch_embed"] = x
    blk = 0
    for s, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        for b in range(depth):
            x = swin_block(P, f"{pre}.layers.{s}.blocks.{b}", x, heads, win,
                           (0, 0, 0) if b % 2 == 0 else shift,
                           None if droppath is None else droppath[blk])
            blk += 1
        if taps is not None:
            taps[f"stage{s}"] = x
        if s < len(cfg["depths"]) - 1:
            x = patch_merge(P, f"{pre}.layers.{s}.downsample", x)
    return _ln(x, P, pre + ".norm", 1e-5)


# --------------------------------------------------------------------------
# EncVideo / EncTxt / fusion / head
# --------------------------------------------------------------------------
def enc_video(P, img, size, droppath=None, taps=None):
    """EncVideo.forward, model.py:37-93 (odr=None, vt_mask=None branch).
    img: (B,T,3,H,W) -> f_img (B, T(1+hw), hid), m_img (B, T(1+hw)) int64."""
    B, T, _, H, W = img.shape
    hw = (H // 32) * (W // 32)
    f = swin_forward(P, "enc_img.swin", img.transpose(1, 2), size, droppath, taps)
    f = f.reshape(B, T, hw, f.shape[-1])
    if "enc_img.fc.weight" in P:
        f = _lin(f, P, "enc_img.fc")
    hid = f.shape[-1]
    f = torch.cat([P["enc_img.emb_cls"].expand(B, T, 1, hid), f], 2)
    f = f + P["enc_img.emb_pos"][:, :, :1 + hw]
    f = f + P["enc_img.emb_len"][:, :T]
    f = _ln(f, P, "enc_img.norm", 1e-5).reshape(B, T * (1 + hw), hid)
    return f, torch.ones(B, T * (1 + hw), dtype=torch.long)


def enc_txt(P, txt, drop=None):
    """EncTxt.forward embed-only branch, model.py:125-142 -> BERT embeddings
    (word + token_type[0] + position[0..X), LN eps 1e-12, dropout)."""
    X = txt.shape[1]
    pre = "enc_txt.emb_txt"
    e = P[pre + ".word_embeddings.weight"][txt] + P[pre + ".token_type_embeddings.weight"][0] \
        + P[pre + ".position_embeddings.weight"][:X]
    e = _ln(e, P, pre + ".LayerNorm", 1e-12)
    return e if drop is None else e * drop


def extended_mask(mask, dtype=torch.float32):
    """get_extended_attention_mask as used at model.py:239: (B,L) 0/1 -> additive (B,1,1,L); a (B,L,L) mask (seq2seq)
    -> (B,1,L,L)."""
    m = mask[:, None, None, :] if mask.dim() == 2 else mask[:, None, :, :]
    return (1.0 - m.to(dtype)) * torch.finfo(dtype).min


def attn_mask(m_img, m_txt, attn_mask_type="full"):
    """LAVENDER_Base.get_attn_mask, model.py:194-221 (mask_pretxt=None)."""
    if attn_mask_type == "seq2seq":
        B, Lv = m_img.shape
        Lt = m_txt.shape[1]
        L = Lv + Lt
        mask = torch.zeros((B, L, L), dtype=torch.long)
        mask[:, :, :Lv] = m_img[:, None, :]
        mask[:, Lv:, Lv:] = torch.tril(torch.ones((B, Lt, Lt), dtype=torch.long))
        return mask
    return torch.cat([m_img, m_txt], 1)


def bert_layer(P, pre, x, add_mask, heads, drop=None):
    """One post-LN BERT layer (the fusion encoder, model.py:242).  drop: None (eval) or the sampled dropout masks of a
    training step as multipliers (0 or 1/(1-p)): (attention probabilities (B,heads,L,L), attention-output hidden (B,L,H),
    FFN-output hidden (B,L,H)) -- HF BertSelfAttention / BertSelfOutput / BertOutput dropout placement."""
    B, L, Hd = x.shape
    hd = Hd // heads
    d_att, d_h1, d_h2 = drop if drop is not None else (None, None, None)

    def split(t):
        return t.reshape(B, L, heads, hd).transpose(1, 2)
    q = split(_lin(x, P, pre + ".attention.self.query"))
    k = split(_lin(x, P, pre + ".attention.self.key"))
    v = split(_lin(x, P, pre + ".attention.self.value"))
    s = q @ k.transpose(-1, -2) * hd ** -0.5 + add_mask
    pr = s.softmax(-1)
    if d_att is not None:
        pr = pr * d_att
    ctx = (pr @ v).transpose(1, 2).reshape(B, L, Hd)
    a = _lin(ctx, P, pre + ".attention.output.dense")
    if d_h1 is not None:
        a = a * d_h1
    x = _ln(a + x, P, pre + ".attention.output.LayerNorm", 1e-12)
    h = _lin(F.gelu(_lin(x, P, pre + ".intermediate.dense")), P, pre + ".output.dense")
    if d_h2 is not None:
        h = h * d_h2
    return _ln(h + x, P, pre + ".output.LayerNorm", 1e-12)


def n_fusion_layers(P):
    n = 0
    while f"trsfr.layer.{n}.attention.self.query.weight" in P:
        n += 1
    return n


def go_cross(P, f_img, m_img, f_txt, m_txt, heads, drops=None, attn_mask_type="full"):
    """LAVENDER_Base.go_cross, model.py:223-243.  drops: per-layer dropout multipliers (see bert_layer)."""
    x = torch.cat([f_img, f_txt], 1)
    add = extended_mask(attn_mask(m_img, m_txt, attn_mask_type))
    for i in range(n_fusion_layers(P)):
        x = bert_layer(P, f"trsfr.layer.{i}", x, add, heads, None if drops is None else drops[i])
    return x


def mlm_head(P, x):
    """BertOnlyMLMHead (main_pretrain_mlm.py:46-48,69)."""
    pre = "fc_mtm.predictions"
    h = _ln(F.gelu(_lin(x, P, pre + ".transform.dense")), P, pre + ".transform.LayerNorm", 1e-12)
    return F.linear(h, P[pre + ".decoder.weight"], P[pre + ".decoder.bias"])


# --------------------------------------------------------------------------
# integer paths of the pretrain step
# --------------------------------------------------------------------------
SPECIAL = dict(cls=101, sep=102, pad=0, mask=103, unk=100, true=2995, false=6270)


def masking(txt, p_mask=0.15, ids=SPECIAL):
    """Agent_Pretrain_MLM.masking, main_pretrain_mlm.py:178-200.  Consumes the global
    torch CPU RNG exactly like the reference: one T.rand(X) per row."""
    txt = txt.clone()
    B, X = txt.shape
    spc = (txt == ids["cls"]) | (txt == ids["sep"]) | (txt == ids["pad"]) | (txt == ids["mask"])
    ans = torch.full(txt.shape, -1, dtype=torch.long)
    if p_mask <= 0:
        return txt, ans
    for i in range(B):
        hit = (~spc[i]) & (torch.rand(X) < p_mask)
        ans[i, hit] = txt[i, hit]
        txt[i, hit] = ids["mask"]
    return txt, ans


def vtm_pairs(B, O):
    """Pair construction of LAVENDER_Pretrain_MLM.forward, main_pretrain_mlm.py:74-106.
    Consumes numpy's global RNG like the reference (one permutation per sample).
    Returns (video_idx, text_idx, is_true) each of length B*O."""
    vi, ti, tr = [], [], []
    for i in range(B):
        vi.append(i); ti.append(i); tr.append(True)
        neg = np.random.permutation([j for j in range(B) if j != i])
        for j in range(O - 1):
            vi.append(i); ti.append(int(neg[j])); tr.append(False)
    return np.array(vi), np.array(ti), np.array(tr)


def pretrain_forward(P, batch, size, heads, droppath=None, vtm_batch=4, ids=SPECIAL, taps=None, drop=None):
    """LAVENDER_Pretrain_MLM.forward, main_pretrain_mlm.py:55-119.  Default: eval-mode arithmetic.  A training step is
    reproduced by passing the masks it sampled: droppath = per-block (scale_attn, scale_mlp) stochastic-depth factors and
    drop = dict(txt=(B,X,H) multiplier of the text-embedding dropout, mtm=[per layer], vtm=[per layer]) (see bert_layer)."""
    img, txt, mask = batch["img"], batch["txt"], batch["mask"]
    B, X = txt.shape
    O = min(B, vtm_batch)
    f_img, m_img = enc_video(P, img, size, droppath, taps)
    f_txt = enc_txt(P, txt, None if drop is None else drop["txt"])
    Lv = f_img.shape[1]
    out = go_cross(P, f_img, m_img, f_txt, mask, heads, None if drop is None else drop["mtm"])
    out_mtm = mlm_head(P, out[:, Lv:])
    vi, ti, tr = vtm_pairs(B, O)
    out = go_cross(P, f_img[vi], m_img[vi], f_txt[ti], mask[ti], heads, None if drop is None else drop["vtm"])
    out_vtm = mlm_head(P, out[:, Lv:])
    ans_vtm = torch.full((B * O, X), -1, dtype=torch.long)
    ans_vtm[:, -1] = torch.where(torch.from_numpy(tr), ids["true"], ids["false"])
    if taps is not None:
        taps.update(f_img=f_img, f_txt=f_txt)
    return dict(out_mtm=out_mtm, out_vtm=out_vtm, ans_vtm=ans_vtm, ans_mtm=batch.get("ans_mtm"))


def pretrain_loss(out):
    """Agent_Pretrain_MLM.step train branch, main_pretrain_mlm.py:158-163 with
    CrossEntropyLoss(ignore_index=-1) (agent.py:72)."""
    V = out["out_mtm"].shape[-1]
    l_mtm = F.cross_entropy(out["out_mtm"].reshape(-1, V), out["ans_mtm"].reshape(-1), ignore_index=-1)
    l_vtm = F.cross_entropy(out["out_vtm"].reshape(-1, V), out["ans_vtm"].reshape(-1), ignore_index=-1)
    return l_mtm, l_vtm


def captioning_encode_forward(P, batch, size, heads):
    """LAVENDER_Captioning.encode_forward, model_for_captioning.py:54-95 (prompt / task token off): go_feat, go_cross with the
    seq2seq mask, MLM head on the text positions."""
    img, txt, mask = batch["img"], batch["txt"], batch["mask"]
    f_img, m_img = enc_video(P, img, size)
    f_txt = enc_txt(P, txt)
    Lv = f_img.shape[1]
    out = go_cross(P, f_img, m_img, f_txt, mask, heads, attn_mask_type=batch.get("attn_mask_type", "seq2seq"))
    return dict(out=mlm_head(P, out[:, Lv:]), ans=batch.get("ans_mtm"))


def score_head(P, x):
    """self.fc of LAVENDER_Pretrain, main_pretrain_task_specific.py:128-133 (eval: Dropout is the identity):
    Linear(H, 2H) -> ReLU -> Linear(2H, 1)."""
    return _lin(torch.relu(_lin(x, P, "fc.1")), P, "fc.3")


def pretrain_ts_forward(P, batch, size, heads, temp, droppath=None):
    """LAVENDER_Pretrain.forward, main_pretrain_task_specific.py:139-177 (eval-mode arithmetic)."""
    img, txt, mask = batch["img"], batch["txt"], batch["mask"]
    B, X = txt.shape
    O = min(B, 4)
    f_img, m_img = enc_video(P, img, size, droppath)
    f_txt = enc_txt(P, txt)
    Lv = f_img.shape[1]
    out = go_cross(P, f_img, m_img, f_txt, mask, heads)
    out_mtm = mlm_head(P, out[:, Lv:])
    vi, ti, _ = vtm_pairs(B, O)
    out = go_cross(P, f_img[vi], m_img[vi], f_txt[ti], mask[ti], heads)
    out_vtm = score_head(P, out[:, Lv, :]).squeeze().view(B, O) / temp
    return dict(out_mtm=out_mtm, out_vtm=out_vtm, ans_vtm=torch.zeros(B, dtype=torch.long), ans_mtm=batch.get("ans_mtm"))


def pretrain_ts_loss(out):
    """Agent_Pretrain.step, main_pretrain_task_specific.py:222-229."""
    V = out["out_mtm"].shape[-1]
    l_mtm = F.cross_entropy(out["out_mtm"].reshape(-1, V), out["ans_mtm"].reshape(-1), ignore_index=-1)
    l_vtm = F.cross_entropy(out["out_vtm"], out["ans_vtm"], ignore_index=-1)
    return l_mtm, l_vtm


def retrieval_forward(P, batch, size, heads, ids=SPECIAL):
    """LAVENDER_Retrieval_MLM.forward, main_retrieval_mlm.py:50-91: all B x B (video i, text j) pairs, i outer."""
    img, txt, mask, vid = batch["img"], batch["txt"], batch["mask"], list(batch["vid"])
    B, X = txt.shape
    f_img, m_img = enc_video(P, img, size)
    f_txt = enc_txt(P, txt)
    Lv = f_img.shape[1]
    vi = np.repeat(np.arange(B), B)
    ti = np.tile(np.arange(B), B)
    out = go_cross(P, f_img[vi], m_img[vi], f_txt[ti], mask[ti], heads)
    out = mlm_head(P, out[:, Lv:])
    ans = torch.full((B * B, X), -1, dtype=torch.long)
    ans[:, -1] = torch.tensor([ids["true"] if vid[i] == vid[j] else ids["false"] for i, j in zip(vi, ti)])
    return out, ans


def qa_mlm_forward(P, batch, size, heads):
    """LAVENDER_QAOE_MLM.forward (main_qaoe_mlm_lsmdc_fib.py:79-93) = LAVENDER_QAMC_MLM.forward (main_qamc_mlm.py:124-140) with
    task tokens / prompts off: one (video, text) sequence per sample, MLM-head logits over the text positions, labels passed through."""
    f_img, m_img = enc_video(P, batch["img"], size)
    f_txt = enc_txt(P, batch["txt"])
    out = go_cross(P, f_img, m_img, f_txt, batch["mask"], heads)
    return mlm_head(P, out[:, f_img.shape[1]:]), batch["mask_ans"]


def retmc_mlm_forward(P, batch, size, heads):
    """LAVENDER_RetMC_MLM.forward, main_retmc_mlm.py:89-113: txt / mask / mask_ans (B, O, X); every video with each of its O texts."""
    txt, mask, ans = batch["txt"], batch["mask"], batch["mask_ans"]
    B, O, X = txt.shape
    f_img, m_img = enc_video(P, batch["img"], size)
    f_txt = enc_txt(P, txt.flatten(0, 1))
    vi = np.repeat(np.arange(B), O)
    out = go_cross(P, f_img[vi], m_img[vi], f_txt, mask.flatten(0, 1), heads)
    return mlm_head(P, out[:, f_img.shape[1]:]), ans


def retmc_acc(out, ans, ids=SPECIAL):
    """Agent_RetMC_MLM.step, eval branch, main_retmc_mlm.py:130-140."""
    B, O, L = ans.shape
    p = out[:, :, ids["true"]] / (out[:, :, ids["true"]] + out[:, :, ids["false"]])
    a = ans.view(B * O, L)
    pick = torch.argmax(p[a != -1].view(B, O), dim=-1)
    return (pick == (a[a != -1].view(B, O) == ids["true"]).nonzero()[:, 1]).float().tolist()


def qa_top_k_acc(out, ans, k):
    """Agent_QAOE_MLM_LSMDC.get_top_k_acc, main_qaoe_mlm_lsmdc_fib.py:115-126."""
    B = out.shape[0]
    a = ans[ans != -1].view(-1, 1)
    o = out[ans != -1].view(a.shape[0], -1)
    ac = (torch.topk(o, k=k, dim=-1).indices == a).any(dim=-1).float().tolist()
    return ac + [0.] * (B - len(ac))


def qamc_choice_acc(out, ans, ans_tok_ids, ans_idx):
    """Agent_QAMC_MLM.step, eval branch, main_qamc_mlm.py:160-170."""
    B = ans.shape[0]
    p = out[:, :, ans_tok_ids][ans != -1]
    p = (p / p.sum(dim=-1).view(B, 1)).view(B, -1)
    return (torch.argmax(p, dim=-1) == ans_idx).float().tolist()


def retrieval_eval_feat(P, img, txt, size):
    """LAVENDER_RetrievalMlmEval.forward('feat'), eval_retrieval_mlm.py:19-37: img (B, Clips, T, C, H, W); the
    video features are averaged over the clips of a video."""
    B, Cl = img.shape[0], img.shape[1]
    f_img, m_img = enc_video(P, img.reshape((-1,) + tuple(img.shape[2:])), size)
    f_img = f_img.view(B, Cl, f_img.shape[1], f_img.shape[2]).mean(dim=1)
    m_img = m_img.view(B, Cl, -1)[:, 0, :]
    return f_img, m_img, enc_txt(P, txt)


def retrieval_eval_cross(P, f_img, m_img, f_txt, m_txt, heads):
    """LAVENDER_RetrievalMlmEval.forward('cross'), eval_retrieval_mlm.py:39-47."""
    out = go_cross(P, f_img, m_img, f_txt, m_txt, heads)
    return mlm_head(P, out[:, f_img.shape[1]:])


# --------------------------------------------------------------------------
# optimizer-side host logic
# --------------------------------------------------------------------------
def param_group_of(name):
    """Agent_Base.build_optimizer, agent.py:96-120: 0 swin/decay, 1 other/decay,
    2 swin/no-decay, 3 other/no-decay (substring rules reproduced verbatim)."""
    nd = any(s in name for s in ("bias", "LayerNorm.bias", "LayerNorm.weight"))
    return (2 if nd else 0) + (0 if "swin." in name else 1)


def warmup_linear_factor(step, max_iter, warmup_ratio=0.1):
    """WarmupLinearLR.get_lr_factor, agent.py:28-36."""
    wu = int(warmup_ratio * max_iter)
    if step < wu:
        return max(0, step / wu)
    step = min(step, max_iter)
    return max(0, (max_iter - step) / (max_iter - wu))


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.98, eps=1e-8):
    """torch.optim.AdamW single-tensor update as configured at agent.py:137-140."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    return p - lr * mh / (vh.sqrt() + eps), m, v


# --------------------------------------------------------------------------
# state_dict layout
# --------------------------------------------------------------------------
def state_spec(size, hidden=768, layers=12, ffn=3072, vocab=30522, max_pos=512,
               max_frame=6, max_patch=14):
    cfg = SWIN_SIZES[size]
    E, win = cfg["embed_dim"], cfg["window_size"]
    tbl = (2 * win[0] - 1) * (2 * win[1] - 1) * (2 * win[2] - 1)
    S = {}
    S["emb_task"] = (10, hidden)
    e = "enc_txt.emb_txt"
    S[e + ".word_embeddings.weight"] = (vocab, hidden)
    S[e + ".position_embeddings.weight"] = (max_pos, hidden)
    S[e + ".token_type_embeddings.weight"] = (2, hidden)
    S[e + ".LayerNorm.weight"] = (hidden,); S[e + ".LayerNorm.bias"] = (hidden,)
    for i in range(layers):
        p = f"trsfr.layer.{i}"
        for n in ("query", "key", "value"):
            S[f"{p}.attention.self.{n}.weight"] = (hidden, hidden); S[f"{p}.attention.self.{n}.bias"] = (hidden,)
        S[f"{p}.attention.output.dense.weight"] = (hidden, hidden); S[f"{p}.attention.output.dense.bias"] = (hidden,)
        S[f"{p}.attention.output.LayerNorm.weight"] = (hidden,); S[f"{p}.attention.output.LayerNorm.bias"] = (hidden,)
        S[f"{p}.intermediate.dense.weight"] = (ffn, hidden); S[f"{p}.intermediate.dense.bias"] = (ffn,)
        S[f"{p}.output.dense.weight"] = (hidden, ffn); S[f"{p}.output.dense.bias"] = (hidden,)
        S[f"{p}.output.LayerNorm.weight"] = (hidden,); S[f"{p}.output.LayerNorm.bias"] = (hidden,)
    s = "enc_img.swin"
    S[s + ".patch_embed.proj.weight"] = (E, 3, 2, 4, 4); S[s + ".patch_embed.proj.bias"] = (E,)
    S[s + ".patch_embed.norm.weight"] = (E,); S[s + ".patch_embed.norm.bias"] = (E,)
    for st, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        C = E * 2 ** st
        for b in range(depth):
            p = f"{s}.layers.{st}.blocks.{b}"
            S[p + ".norm1.weight"] = (C,); S[p + ".norm1.bias"] = (C,)
            S[p + ".attn.relative_position_bias_table"] = (tbl, heads)
            S[p + ".attn.qkv.weight"] = (3 * C, C); S[p + ".attn.qkv.bias"] = (3 * C,)
            S[p + ".attn.proj.weight"] = (C, C); S[p + ".attn.proj.bias"] = (C,)
            S[p + ".norm2.weight"] = (C,); S[p + ".norm2.bias"] = (C,)
            S[p + ".mlp.fc1.weight"] = (4 * C, C); S[p + ".mlp.fc1.bias"] = (4 * C,)
            S[p + ".mlp.fc2.weight"] = (C, 4 * C); S[p + ".mlp.fc2.bias"] = (C,)
        if st < 3:
            p = f"{s}.layers.{st}.downsample"
            S[p + ".reduction.weight"] = (2 * C, 4 * C)
            S[p + ".norm.weight"] = (4 * C,); S[p + ".norm.bias"] = (4 * C,)
    S[s + ".norm.weight"] = (8 * E,); S[s + ".norm.bias"] = (8 * E,)
    if 8 * E != hidden:
        S["enc_img.fc.weight"] = (hidden, 8 * E); S["enc_img.fc.bias"] = (hidden,)
    S["enc_img.emb_cls"] = (1, 1, 1, hidden)
    S["enc_img.emb_pos"] = (1, 1, 1 + max_patch ** 2, hidden)
    S["enc_img.emb_len"] = (1, max_frame, 1, hidden)
    S["enc_img.emb_odr"] = (1, 1, 1, hidden)
    S["enc_img.norm.weight"] = (hidden,); S["enc_img.norm.bias"] = (hidden,)
    h = "fc_mtm.predictions"
    S[h + ".bias"] = (vocab,)
    S[h + ".transform.dense.weight"] = (hidden, hidden); S[h + ".transform.dense.bias"] = (hidden,)
    S[h + ".transform.LayerNorm.weight"] = (hidden,); S[h + ".transform.LayerNorm.bias"] = (hidden,)
    S[h + ".decoder.weight"] = (vocab, hidden)
    return S


def filled_params(size, **kw):
    P = fill_state_dict(state_spec(size, **kw))
    # decoder.bias is tied to predictions.bias in the reference's HF head
    P["fc_mtm.predictions.decoder.bias"] = P["fc_mtm.predictions.bias"]
    return P
