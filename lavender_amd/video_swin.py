"""Video Swin Transformer encoder -- MI355X-native mirror of the reference's visbackbone/video_swin.py.

Same public surface (class / attribute / state_dict key names, `get_vidswin_model(args)`), different engine:
tokens stay channels-last (B*T*H*W, C) bf16 from the patch embedding to the final norm (the reference ping-pongs
NCDHW <-> NDHWC per stage, video_swin.py:354,367,474), the cyclic shift / window partition / reverse
(video_swin.py:82-91,218-239) are index math inside the window-attention kernel, and every stage is a chain of
hand-written HIP kernels (lavender_amd.engine).  There is no eager fallback.
"""
import math
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import engine as E
from . import hip as K

# visbackbone/swin_{tiny,base,large}*.py -- only the keys video_swin.py:616-634 reads
SWIN_SIZES = {
    "micro": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 7, 7)),
    "micro12": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=(8, 12, 12)),  # test size: Large-384 windows
    "tiny": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=(8, 7, 7)),
    "base": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=(8, 7, 7)),
    "large": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=(8, 12, 12)),
}


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def get_window_size(x_size, window_size, shift_size=None):
    """video_swin.py:93-106."""
    use_w = [x if x <= w else w for x, w in zip(x_size, window_size)]
    if shift_size is None:
        return tuple(use_w)
    use_s = [0 if x <= w else s for x, w, s in zip(x_size, window_size, shift_size)]
    return tuple(use_w), tuple(use_s)


def relative_position_index(window_size):
    """The int64 buffer of video_swin.py:118-135 in closed form: code(i) - code(j) + const."""
    wd, wh, ww = window_size
    d, h, w = np.meshgrid(np.arange(wd), np.arange(wh), np.arange(ww), indexing="ij")
    code = (d * (2 * wh - 1) * (2 * ww - 1) + h * (2 * ww - 1) + w).reshape(-1)
    const = (wd - 1) * (2 * wh - 1) * (2 * ww - 1) + (wh - 1) * (2 * ww - 1) + (ww - 1)
    return torch.from_numpy(code[:, None] - code[None, :] + const).long()


class _Params(nn.Module):
    """Parameter holders: the arithmetic lives in lavender_amd.engine, these only own names and shapes."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is executed by its parent stage on the HIP engine; it has no standalone forward")


class Linear(_Params):
    def __init__(self, in_f, out_f, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.zeros(out_f)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            b = 1 / math.sqrt(in_f)
            nn.init.uniform_(self.bias, -b, b)


class LayerNorm(_Params):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.normalized_shape = (dim,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class Mlp(_Params):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = Linear(dim, hidden)
        self.fc2 = Linear(hidden, dim)


class WindowAttention3D(_Params):
    """video_swin.py:109-170 (parameters + the relative_position_index buffer of the checkpoint contract)."""

    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        n = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) * (2 * window_size[2] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = Linear(dim, dim * 3)
        self.proj = Linear(dim, dim)
        trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock3D(_Params):
    def __init__(self, dim, num_heads, window_size, shift_size, drop_path):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        assert dim // num_heads == 32, "the window-attention kernel is specialised for head_dim 32 (every Swin size in the reference)"
        self.drop_prob = float(drop_path)
        self.keep_prob = 1.0 - float(drop_path)
        self.norm1 = LayerNorm(dim)
        self.attn = WindowAttention3D(dim, window_size, num_heads)
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * 4))


class PatchMerging(_Params):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = Linear(4 * dim, 2 * dim, bias=False)
        self.norm = LayerNorm(4 * dim)


class BasicLayer(_Params):
    def __init__(self, dim, depth, num_heads, window_size, drop_path, downsample):
        super().__init__()
        self.window_size = window_size
        self.shift_size = tuple(i // 2 for i in window_size)
        self.depth = depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim, num_heads, window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size, drop_path[i])
            for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None


class PatchEmbed3D(_Params):
    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=96):
        super().__init__()
        assert tuple(patch_size) == (2, 4, 4) and in_chans == 3, "the im2col kernel is specialised for patch (2,4,4) x 3 channels"
        self.patch_size, self.in_chans, self.embed_dim = patch_size, in_chans, embed_dim
        self.proj = _Params()
        self.proj.weight = nn.Parameter(torch.empty(embed_dim, in_chans, *patch_size))
        self.proj.bias = nn.Parameter(torch.zeros(embed_dim))
        nn.init.kaiming_uniform_(self.proj.weight, a=math.sqrt(5))
        b = 1 / math.sqrt(in_chans * 32)
        nn.init.uniform_(self.proj.bias, -b, b)
        self.norm = LayerNorm(embed_dim)


class SwinTransformer3D(nn.Module):
    """video_swin.py:408-480.  forward(x: (B,3,T,H,W)) -> (B, 8E, T, H/32, W/32) like the reference;
    forward_tokens(img: (B,T,3,H,W)) -> channels-last tokens (B*T*h*w, 8E) is what EncVideo uses."""

    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
                 window_size=(8, 7, 7), drop_path_rate=0.2, patch_norm=True, **unused):
        super().__init__()
        assert patch_norm, "patch_norm=False is not supported"
        self.num_layers = len(depths)
        self.embed_dim, self.window_size, self.patch_size = embed_dim, tuple(window_size), patch_size
        self.depths = list(depths)
        self.patch_embed = PatchEmbed3D(patch_size, in_chans, embed_dim)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], tuple(window_size),
                                          dpr[sum(depths[:i]):sum(depths[:i + 1])], downsample=i < self.num_layers - 1))
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.norm = LayerNorm(self.num_features)
        self._arena_ref = None
        self._keep_dev = None

    # ---- arena plumbing ------------------------------------------------------------------------
    def _arena(self):
        a = self._arena_ref() if self._arena_ref is not None else None
        if a is None:
            from .arena import ParamArena
            dev = self.norm.weight.device
            if dev.type != "cuda":
                raise RuntimeError("lavender_amd runs on the MI355X only: move the model to a cuda device first (no CPU path)")
            self._own_arena = ParamArena(self, dev)
            self._arena_ref = weakref.ref(self._own_arena)
            a = self._own_arena
        return a

    def init_weights(self):
        """video_swin.py:535-568 random-init branch: Linear trunc_normal(.02)/0, LayerNorm 1/0."""
        for m in self.modules():
            if isinstance(m, Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def _droppath_scales(self, B, device):
        """per-block, per-sample stochastic-depth factors (video_swin.py:46-54); None in eval mode."""
        if not self.training:
            return None
        nblk = sum(self.depths)
        if self._keep_dev is None or self._keep_dev.device != device:
            keep = [1.0 - blk.drop_prob for layer in self.layers for blk in layer.blocks for _ in (0, 1)]
            self._keep_dev = torch.tensor(keep, dtype=torch.float32, device=device)
        scale = torch.empty((2 * nblk, B), dtype=torch.float32, device=device)
        K.fill_droppath(2 * nblk, B, self._keep_dev, K.next_seed(), scale)
        return scale

    def forward_tokens(self, img, frame_major=True, droppath=None, taps=None):
        arena = self._arena()
        arena.sync_half_if_stale()
        anchor = arena.anchor
        if frame_major:
            B, T, _, H, W = img.shape
        else:
            B, _, T, H, W = img.shape
        x = E.PatchEmbedFn.apply(anchor, img, self.patch_embed, frame_major)
        D, Hc, Wc = T, H // 4, W // 4
        if droppath is None:
            droppath = self._droppath_scales(B, x.device)
        if taps is not None:
            taps["patch_embed"] = x
        bi = 0
        for s, layer in enumerate(self.layers):
            window, shift = get_window_size((D, Hc, Wc), layer.window_size, layer.shift_size)
            for i, blk in enumerate(layer.blocks):
                sh = shift if any(blk.shift_size) else (0, 0, 0)
                geo = dict(B=B, D=D, H=Hc, W=Wc, window=window, shift=sh, cfg_window=layer.window_size,
                           notify=f"swin_stage{s}_grads_final" if i == 0 else None, arena=arena)
                dpa = droppath[2 * bi] if droppath is not None and blk.drop_prob > 0 else None
                dpm = droppath[2 * bi + 1] if droppath is not None and blk.drop_prob > 0 else None
                x = E.SwinBlockFn.apply(anchor, x, blk, geo, dpa, dpm)
                bi += 1
            if taps is not None:
                taps[f"stage{s}"] = x
            if layer.downsample is not None:
                x = E.PatchMergeFn.apply(anchor, x, layer.downsample, B * D, Hc, Wc)
                Hc, Wc = (Hc + 1) // 2, (Wc + 1) // 2
        x = E.LayerNormFn.apply(anchor, x, self.norm, 1e-5)
        return x, (B, D, Hc, Wc)

    def forward(self, x):
        tok, (B, D, h, w) = self.forward_tokens(x, frame_major=False)
        return tok.view(B, D, h, w, -1).permute(0, 4, 1, 2, 3)


def get_vidswin_model(args):
    """video_swin.py:571-645.  The mmcv config files are replaced by the SWIN_SIZES table (only the backbone
    geometry was ever read from them); checkpoint selection / loading semantics are kept."""
    size = args.vis_backbone_size
    if int(args.size_img) == 384:
        assert size == "large"
        model_path = (f'./_models/swin_transformer/swin_{size}_patch4_window12_384_22k.pth' if args.vis_backbone_init == "2d"
                      else './_models/video_swin_transformer/swin_%s_384_patch244_window81212_kinetics%s_22k.pth' % (size, args.kinetics))
    elif size == "tiny":
        model_path = ('./_models/swin_transformer/swin_tiny_patch4_window7_224.pth' if args.vis_backbone_init == "2d"
                      else './_models/video_swin_transformer/swin_tiny_patch244_window877_kinetics400_1k.pth')
    else:
        model_path = ('./_models/swin_transformer/swin_%s_patch4_window7_224_22k.pth' % size if args.vis_backbone_init == "2d"
                      else './_models/video_swin_transformer/swin_%s_patch244_window877_kinetics%s_22k.pth' % (size, args.kinetics))
    if size not in SWIN_SIZES:
        raise ValueError(f"unknown vis_backbone_size {size!r}")
    cfg = SWIN_SIZES[size]
    if args.vis_backbone_init == "random":
        model_path = None
    args.vis_backbone_pretrained_weight = model_path
    if args.vis_backbone_init == "2d":
        raise NotImplementedError("2D->3D weight inflation (video_swin.py:482-533) is outside the hot path; "
                                  "load a 3D checkpoint or use vis_backbone_init='random'")
    swin = SwinTransformer3D(patch_size=(2, 4, 4), in_chans=3, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                             num_heads=cfg["num_heads"], window_size=cfg["window_size"], drop_path_rate=0.2, patch_norm=True)
    if args.vis_backbone_init == "3d" and model_path is not None:
        sd = load_checkpoint_3d(model_path)
        missing, unexpected = swin.load_state_dict(sd, strict=False)
        print(f"Missing keys in loaded video_swin_transformerr: {missing}")
        print(f"Unexpected keys in loaded video_swin_transformer: {unexpected}")
    else:
        swin.init_weights()
    return swin


def load_checkpoint_3d(model_path):
    """video_swin.py:648-654."""
    ck = torch.load(model_path, map_location='cpu')['state_dict']
    return {k.replace("backbone.", ""): v for k, v in ck.items()}
