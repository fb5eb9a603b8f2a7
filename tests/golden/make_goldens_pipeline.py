"""Golden vectors of the input pipeline (CONTAINER ONLY for the TSV part -- reads /root/reference/_tools/msrvtt.tsv).

Writes next to this file:
  * msrvtt_2rows.tsv / .lineidx   the first two rows of the reference's own sample data file (_tools/msrvtt.tsv: id + 5 base64
                                  JPEG frames per row, the on-disk format of _tools/extract_tsv.py:20-27) -- data, not source;
  * pipeline_frames.npz           expected outputs computed by oracle/pipeline_ref.py running on Pillow + torch (CPU):
        tsv_{r}_{f}_rgb_sub / _rgb_sum      decoded RGB frames (sub-sample + byte sum)
        tsv_{r}_{f}_{transform}_sub / _sum  final normalised tensors of the three per-image transforms at size_img 224
        tsv_train_{r}                       a seeded train-mode sample (temporal sampling + mixed transforms + random crops)
        tsv_vid_train_{r} / tsv_vid_val_{r} the clip transforms vid_rand_crop (seeded) / vid_center_crop
        syn_{i}_jpg / _rgb                  small synthetic JPEGs (4:4:4, 4:2:2, 4:2:0, grey, odd sizes, optimised Huffman
                                            tables, restart markers) and their full decoded frames
        syn_{i}_{transform}_{S}             full final tensors of the transforms on those frames

    python tests/golden/make_goldens_pipeline.py
"""
import base64
import io
import os
import random
import sys
import zlib

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pipeline_ref as PR  # noqa: E402

SRC = "/root/reference/_tools/msrvtt.tsv"


def sub(a, n=4096, seed=7):
    flat = np.asarray(a).reshape(-1)
    idx = np.random.RandomState(seed + flat.size % 9973).permutation(flat.size)[:n]
    return flat[idx]


def synthetic():
    """(name, jpeg bytes): smooth + noisy content so that every DCT coefficient class and both chroma phases are exercised."""
    rs = np.random.RandomState(3)
    out = []

    def image(w, h, grey=False):
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(x / 5.0 + c) * np.cos(y / 7.0 - c) for c in range(3)], -1)
        base += rs.randn(h, w, 3) * 25
        base[h // 3:h // 3 + 3] = 255
        base[:, w // 2:w // 2 + 2] = 0
        a = np.clip(base, 0, 255).astype(np.uint8)
        return Image.fromarray(a[:, :, 0], "L") if grey else Image.fromarray(a, "RGB")

    cases = [("444_17x13", image(17, 13), dict(quality=90, subsampling=0)),
             ("420_33x47", image(33, 47), dict(quality=75, subsampling=2)),
             ("422_64x48", image(64, 48), dict(quality=60, subsampling=1)),
             ("grey_100x75", image(100, 75, grey=True), dict(quality=80)),
             ("420_opt_97x61", image(97, 61), dict(quality=35, subsampling=2, optimize=True)),
             ("420_q100_48x48", image(48, 48), dict(quality=100, subsampling=2)),
             ("420_rst_81x50", image(81, 50), dict(quality=70, subsampling=2, restart_marker_blocks=3)),
             ("420_1x1", image(1, 1), dict(quality=75, subsampling=2)),
             ("420_16x16_flat", Image.new("RGB", (16, 16), (200, 30, 90)), dict(quality=75, subsampling=2))]
    for name, im, kw in cases:
        buf = io.BytesIO()
        im.save(buf, format="JPEG", **kw)
        out.append((name, buf.getvalue()))
    return out


def main():
    res = {}
    # ---- TSV fixture ------------------------------------------------------------------------------------------------------
    with open(SRC, "r") as f:
        rows = [f.readline() for _ in range(2)]
    with open(f"{HERE}/msrvtt_2rows.tsv", "w") as f, open(f"{HERE}/msrvtt_2rows.lineidx", "w") as fi:
        for r in rows:
            fi.write("%d\n" % f.tell())
            f.write(r)
    offs = [int(x) for x in open(f"{HERE}/msrvtt_2rows.lineidx")]
    ids = []
    for r, pos in enumerate(offs):
        item = PR.read_row(f"{HERE}/msrvtt_2rows.tsv", pos)
        ids.append(item[0])
        for fi_, b in enumerate(item[1:]):
            img = PR.str2img(b)
            rgb = np.array(img)
            res[f"tsv_{r}_{fi_}_rgb_sub"] = sub(rgb)
            res[f"tsv_{r}_{fi_}_rgb_sum"] = np.array([rgb.astype(np.int64).sum(), zlib.adler32(rgb.tobytes())], dtype=np.int64)
            for t in ("pad_resize", "img_center_crop"):
                x = getattr(PR, t)(img, 224).numpy()
                res[f"tsv_{r}_{fi_}_{t}_sub"] = sub(x)
                res[f"tsv_{r}_{fi_}_{t}_sum"] = np.array([x.astype(np.float64).sum(), np.abs(x.astype(np.float64)).sum()])
        random.seed(5 + r)
        g = torch.Generator()
        g.manual_seed(5 + r)
        x = PR.get_img_or_video(item[1:], 4, 224, ["img_rand_crop", "pad_resize", "img_center_crop"], "train", random, g).numpy()
        res[f"tsv_train_{r}_sub"] = sub(x)
        res[f"tsv_train_{r}_sum"] = np.array([x.astype(np.float64).sum(), np.abs(x.astype(np.float64)).sum()])
        random.seed(9 + r)                                  # the clip transforms (one crop window per clip, python RNG)
        x = PR.get_img_or_video(item[1:], 4, 224, ["vid_rand_crop"], "train", random, None).numpy()
        res[f"tsv_vid_train_{r}_sub"] = sub(x)
        res[f"tsv_vid_train_{r}_sum"] = np.array([x.astype(np.float64).sum(), np.abs(x.astype(np.float64)).sum()])
        x = PR.get_img_or_video(item[1:], 4, 224, ["vid_rand_crop"], "val", random, None).numpy()
        res[f"tsv_vid_val_{r}_sub"] = sub(x)
    res["tsv_ids"] = np.array(ids)
    # ---- synthetic JPEGs --------------------------------------------------------------------------------------------------
    names = []
    for name, jpg in synthetic():
        names.append(name)
        b = base64.b64encode(jpg)
        img = PR.str2img(b)
        res[f"syn_{name}_jpg"] = np.frombuffer(jpg, dtype=np.uint8)
        res[f"syn_{name}_rgb"] = np.array(img)
        w, h = img.size
        for S in (8, 24):
            res[f"syn_{name}_pad_resize_{S}"] = PR.pad_resize(img, S).numpy()
            if min(w, h) >= 2:
                res[f"syn_{name}_img_center_crop_{S}"] = PR.img_center_crop(img, S).numpy()
                g = torch.Generator()
                g.manual_seed(S)
                res[f"syn_{name}_img_rand_crop_{S}"] = PR.img_rand_crop(img, S, g).numpy()
    res["syn_names"] = np.array(names)
    np.savez_compressed(f"{HERE}/pipeline_frames.npz", **res)
    print("written", len(res), "arrays;", os.path.getsize(f"{HERE}/pipeline_frames.npz"), "bytes; tsv", os.path.getsize(f"{HERE}/msrvtt_2rows.tsv"))


if __name__ == "__main__":
    main()
