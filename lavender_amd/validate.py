"""fp32-I/O validation mode (tier T2 of SURVEY.md section 8c; BASELINE.json north_star "MLM logits within 1e-3 of
reference").

The same forward path as the production engine -- same parameter holders, same index conventions, every operation a
HIP kernel behind the C ABI -- but with fp32 activations end to end: GEMMs on the exact-fp32 matrix instruction
(`lav_v_gemm_f32`, v_mfma_f32_32x32x2_f32), LayerNorm through the fp32 I/O mode of `lav_layernorm_fwd`, attention and
embeddings in fp32 (`lavender_amd/csrc/validate.hip`).  Forward only, eval arithmetic (no dropout / drop-path), speed
irrelevant.  Selected with `args.validate_fp32 = True` or LAV_FP32=1; `tests/test_gpu_validate.py` holds it to
max|dlogit| <= 1e-3 against the reference goldens.  It also implements the zero-pad branch of a Swin block
(video_swin.py:211-215,241-242) for every geometry, so the 5x64^2 / 4x96^2 reference fixture runs on the GPU through it.
"""
import numpy as np
import torch

from . import hip as K
from .video_swin import get_window_size

f32 = torch.float32


def _buf(rows, cols, dev):
    return torch.empty((rows, cols), dtype=f32, device=dev)


def linear(x, lin, act=0, residual=None, weight=None, bias=None):
    """x (M, K) fp32 -> (M, N) fp32 = act(x W^T + b) + residual, W / b the fp32 master parameters."""
    w = lin.weight.data if weight is None else weight
    b = (lin.bias.data if getattr(lin, "bias", None) is not None else None) if bias is None else bias
    N, Kd = w.shape[0], w[0].numel()
    M = x.shape[0]
    out = _buf(M, N, x.device)
    K.v_gemm(x, w.reshape(N, Kd), out, M, N, Kd, bias=b, act=act, residual=residual)
    return out


def layernorm(x, ln, eps, gather=None, rows=None):
    rows = x.shape[0] if rows is None else rows
    Cn = ln.weight.shape[0]
    out = _buf(rows, Cn, x.device)
    K.layernorm_fwd(x, rows, Cn, ln.weight.data, ln.bias.data, eps, gather=gather, want_stats=False, out32=out, want16=False)
    return out


def swin_tokens(swin, img, frame_major=True, taps=None):
    """SwinTransformer3D.forward (video_swin.py:468-480) on fp32 channels-last tokens."""
    if frame_major:
        B, T, _, H, W = img.shape
    else:
        B, _, T, H, W = img.shape
    assert H % 4 == 0 and W % 4 == 0, "frame sizes that are not multiples of the 4x4 patch are not supported"
    dev = img.device
    img = img.contiguous().float()
    D, Hc, Wc = T, H // 4, W // 4
    M = B * D * Hc * Wc
    E = swin.embed_dim
    cols = _buf(M, 96, dev)
    K.v_im2col(img, B, T, H, W, frame_major, cols)
    x = linear(cols, None, weight=swin.patch_embed.proj.weight.data.reshape(E, 96), bias=swin.patch_embed.proj.bias.data)
    x = layernorm(x, swin.patch_embed.norm, 1e-5)
    if taps is not None:
        taps["patch_embed"] = x
    for s, layer in enumerate(swin.layers):
        window, shift = get_window_size((D, Hc, Wc), layer.window_size, layer.shift_size)
        C = x.shape[1]
        for blk in layer.blocks:
            sh = shift if any(blk.shift_size) else (0, 0, 0)
            a = blk.attn
            y = layernorm(x, blk.norm1, 1e-5)
            qkv = linear(y, a.qkv)
            att = K.Attn(0, blk.num_heads, C // blk.num_heads, B=B, D=D, H=Hc, W=Wc, wd=window[0], wh=window[1], ww=window[2],
                         sd=sh[0], sh=sh[1], sw=sh[2], cfg_wd=layer.window_size[0], cfg_wh=layer.window_size[1],
                         cfg_ww=layer.window_size[2], bias_table=a.relative_position_bias_table.data, fast=False)
            ao = _buf(x.shape[0], C, dev)
            K.v_attention(att, qkv, ao, pad_qkv=a.qkv.bias.data)
            x = linear(ao, a.proj, residual=x)
            h = linear(layernorm(x, blk.norm2, 1e-5), blk.mlp.fc1, act=1)
            x = linear(h, blk.mlp.fc2, residual=x)
        if taps is not None:
            taps[f"stage{s}"] = x
        if layer.downsample is not None:
            if Hc % 2 or Wc % 2:                          # zero row / column at the far edge (video_swin.py:273-276)
                from .engine import pad_maps
                (_, Hc, Wc), Mp, to_pad, _ = pad_maps(dev, B * D, (1, Hc, Wc), (1, 2, 2))
                xp = _buf(Mp, C, dev)
                K.v_gather_rows(x, to_pad, Mp, C, xp)
                x = xp
            y = layernorm(x, layer.downsample.norm, 1e-5, gather=(Hc, Wc, C), rows=x.shape[0] // 4)
            x = linear(y, layer.downsample.reduction)
            Hc, Wc = Hc // 2, Wc // 2
    return layernorm(x, swin.norm, 1e-5), (B, D, Hc, Wc)


def enc_video(enc, img, taps=None):
    """EncVideo.forward (model.py:37-93): -> (B, T(1+hw), hidden) fp32."""
    tok, (B, T, h, w) = swin_tokens(enc.swin, img, frame_major=True, taps=taps)
    hw = h * w
    feat = linear(tok, enc.fc) if enc.fc is not None else tok
    Hd = feat.shape[1]
    Lv = T * (1 + hw)
    out = torch.empty((B, Lv, Hd), dtype=f32, device=img.device)
    K.v_video_embed(feat, B, T, hw, Hd, enc.emb_cls.data, enc.emb_pos.data, enc.emb_len.data, enc.norm.weight.data,
                    enc.norm.bias.data, 1e-5, out, Lv)
    return out


def enc_txt(enc, txt):
    """EncTxt.forward, embed-only branch (model.py:125-129), eval: no dropout."""
    emb = enc.emb_txt
    n, X = txt.shape
    Hd = emb.word_embeddings.weight.shape[1]
    out = torch.empty((n, X, Hd), dtype=f32, device=txt.device)
    K.v_text_embed(txt.contiguous(), n, X, Hd, emb.word_embeddings.weight.data, emb.position_embeddings.weight.data,
                   emb.token_type_embeddings.weight.data, emb.LayerNorm.weight.data, emb.LayerNorm.bias.data, emb.LayerNorm.eps, out)
    return out


def pair_sequences(f_img, f_txt, vi, ti):
    """[video rows of sample vi[k] | text rows of sample ti[k]] (model.py:235 + the pair lists of the callers)."""
    B, Lv, Hd = f_img.shape
    X = f_txt.shape[1]
    n, L = len(vi), Lv + X
    src = torch.cat([f_img.reshape(B * Lv, Hd), f_txt.reshape(-1, Hd)], 0)
    vi, ti = np.asarray(vi, dtype=np.int64), np.asarray(ti, dtype=np.int64)
    idx = np.empty((n, L), dtype=np.int32)
    idx[:, :Lv] = vi[:, None] * Lv + np.arange(Lv)[None, :]
    idx[:, Lv:] = B * Lv + ti[:, None] * X + np.arange(X)[None, :]
    out = _buf(n * L, Hd, f_img.device)
    K.v_gather_rows(src, torch.from_numpy(idx.reshape(-1)).to(f_img.device), n * L, Hd, out)
    return out.view(n, L, Hd)


def encode(model, feat, mask, causal_from=0):
    """The 12 post-LN BertLayers of go_cross (model.py:239-243) on fp32 rows."""
    n, L, Hd = feat.shape
    km = mask.to(torch.int32).contiguous()
    x = feat.reshape(n * L, Hd)
    arena = model.arena()
    for lyr in model.trsfr.layer:
        att_m, ao, inter, outp = lyr.attention.self, lyr.attention.output, lyr.intermediate, lyr.output
        _, _, wqkv = arena.fused_view(att_m.query.weight, 3 * Hd)
        _, _, bqkv = arena.fused_view(att_m.query.bias, 3 * Hd)
        qkv = linear(x, None, weight=wqkv, bias=bqkv)
        att = K.Attn(1, lyr.num_heads, Hd // lyr.num_heads, n_seq=n, L=L, key_mask=km, dropout_p=0.0, seed=0, causal_from=int(causal_from))
        cx = _buf(n * L, Hd, x.device)
        K.v_attention(att, qkv, cx)
        x = layernorm(linear(cx, ao.dense, residual=x), ao.LayerNorm, ao.LayerNorm.eps)
        h = linear(x, inter.dense, act=1)
        x = layernorm(linear(h, outp.dense, residual=x), outp.LayerNorm, outp.LayerNorm.eps)
    return x.view(n, L, Hd)


def mlm_head(head, x):
    """BertOnlyMLMHead (main_pretrain_mlm.py:46-48): (..., H) fp32 -> (..., vocab) fp32 logits."""
    shp = x.shape
    tr, dec = head.predictions.transform, head.predictions.decoder
    t = linear(x.reshape(-1, shp[-1]).contiguous(), tr.dense, act=1)
    t = layernorm(t, tr.LayerNorm, tr.LayerNorm.eps)
    return linear(t, dec).view(*shp[:-1], dec.weight.shape[0])


def pretrain_mlm_forward(model, batch, taps=None):
    """LAVENDER_Pretrain_MLM.forward (main_pretrain_mlm.py:55-119) in fp32; same outputs as the production forward."""
    from .pretrain_mlm import vtm_pairs
    img, txt, mask = batch["img"], batch["txt"], batch["mask"]
    _B, _X = txt.shape
    _O = min(_B, model.vtm_batch)
    model.arena().require_full_master("fp32 validation forward")      # reads the fp32 masters
    f_img = enc_video(model.enc_img, img, taps)
    f_txt = enc_txt(model.enc_txt, txt)
    Lv = f_img.shape[1]
    m_img = torch.ones((_B, Lv), dtype=torch.long, device=img.device)
    if taps is not None:
        taps["f_img"], taps["f_txt"] = f_img, f_txt
    ident = np.arange(_B)
    out = encode(model, pair_sequences(f_img, f_txt, ident, ident), torch.cat([m_img, mask], 1))
    out_mtm = mlm_head(model.fc_mtm, out[:, Lv:])
    vi, ti, tr = vtm_pairs(_B, _O)
    vi_t, ti_t = torch.as_tensor(vi, device=img.device), torch.as_tensor(ti, device=img.device)
    out = encode(model, pair_sequences(f_img, f_txt, vi, ti), torch.cat([m_img[vi_t], mask[ti_t]], 1))
    out_vtm = mlm_head(model.fc_mtm, out[:, Lv:])
    ans_vtm = torch.full((_B * _O, _X), -1, dtype=torch.long)
    ans_vtm[:, -1] = torch.where(torch.from_numpy(tr), model.true_token_id, model.false_token_id)
    return {"out_vtm": out_vtm, "out_mtm": out_mtm, "ans_vtm": ans_vtm.to(txt.device), "ans_mtm": batch.get("ans_mtm")}
