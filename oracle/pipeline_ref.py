"""CPU restatement of the reference's input pipeline for the pretrain path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(lavender_amd/data.py + lavender_amd/csrc/pipeline.hip) never does and has no CPU fallback.

What is restated, and against what it is pinned:
  * dataset.py:40-46, main_pretrain_task_specific.py:75-85   TSV seek + split                       -> read_row
  * dataset.py:177-186  str2img: base64 -> cv2.imdecode / PIL.Image.open -> RGB                      -> str2img.  cv2 is absent from
    this image; the function takes the reference's own `except` branch (PIL).  Both go through libjpeg(-turbo) with default
    settings (islow IDCT, fancy upsampling), whose output is bit-identical across builds by design.
  * dataset.py:107-118,120-130,164-175  pad_resize / img_center_crop / img_rand_crop; :132-162 vid_center_crop / vid_rand_crop
    (visbackbone/video_transform.py Resize / RandomCrop / CenterCrop / ClipToTensor / Normalize)     -> same names here.
    torchvision (v0.11+ semantics: Resize on a PIL image is PIL.Image.resize(BILINEAR), Pad fills 0, CenterCrop rounds with
    Python round(), RandomCrop draws torch.randint for the row then the column, ToTensor = uint8 -> fp32 / 255, Normalize =
    (x - mean) / std in fp32) is NOT installed here, so these few lines are restated from torchvision's published source and
    executed with Pillow + torch on the CPU: the pixel arithmetic is Pillow's and torch's own.
  * dataset.py:188-216  sampling / temporal_sample                                                   -> same names here.
Parity pin (round 3): the reference's dataset.Dataset_Base DOES import in the build container with the stub recipe of
tests/golden/make_goldens.py (cv2 / torchvision mocked: str2img then takes the reference's own PIL branch; the clip transforms of
visbackbone/video_transform.py need neither).  tests/golden/make_goldens_pipeline_ref.py runs the REFERENCE's sampling,
temporal_sample, str2img, vid_center_crop, vid_rand_crop and get_img_or_video(["vid_rand_crop"]) on the fixture rows and writes
tests/golden/pipeline_ref_pin.npz; tests/test_oracle_golden.py::test_pipeline_oracle_against_the_reference_dataset_class holds this
module to those vectors (exact).  PINNED: read_row's consumer chain, str2img, sampling, temporal_sample, vid_center_crop,
vid_rand_crop, get_img_or_video on the clip transforms.  UNPINNED ("restated from torchvision's published source"): pad_resize,
img_center_crop, img_rand_crop -- they call torchvision.transforms, which is not installed here, so their fixtures in
tests/golden/pipeline_frames.npz (make_goldens_pipeline.py) come from this module itself.
"""
import base64
import io
import math
import random

import numpy as np
import torch
from PIL import Image

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


def read_row(tsv_path, pos):
    """dataset.py:44-46: seek, readline, split at tabs, strip."""
    with open(tsv_path, "r") as f:
        f.seek(pos)
        return [s.strip() for s in f.readline().split("\t")]


def str2img(b):
    """dataset.py:177-186 (the PIL branch)."""
    return Image.open(io.BytesIO(base64.b64decode(b))).convert("RGB")


def _to_tensor_normalize(img):
    """torchvision ToTensor + Normalize (dataset.py:113-116)."""
    t = torch.from_numpy(np.array(img, np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(MEAN, dtype=torch.float32).view(-1, 1, 1)
    std = torch.as_tensor(STD, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(mean).div_(std)


def _resize_short(img, size):
    """torchvision Resize(int): shorter side -> size, keep the aspect with int() truncation; unchanged if already there."""
    w, h = img.size
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    if (w, h) == (nw, nh):
        return img
    return img.resize((nw, nh), Image.BILINEAR)


def pad_resize(img, size_img):
    """dataset.py:107-118."""
    w, h = img.size
    pl, pt = (0, (w - h) // 2) if w > h else ((h - w) // 2, 0)
    padded = Image.new("RGB", (w + 2 * pl, h + 2 * pt), 0)
    padded.paste(img, (pl, pt))
    if padded.size != (size_img, size_img):
        padded = padded.resize((size_img, size_img), Image.BILINEAR)
    return _to_tensor_normalize(padded)


def img_center_crop(img, size_img):
    """dataset.py:120-130."""
    img = _resize_short(img, size_img)
    w, h = img.size
    top, left = int(round((h - size_img) / 2.0)), int(round((w - size_img) / 2.0))
    return _to_tensor_normalize(img.crop((left, top, left + size_img, top + size_img)))


def img_rand_crop(img, size_img, generator=None):
    """dataset.py:164-175; RandomCrop.get_params draws the row offset, then the column offset."""
    img = _resize_short(img, size_img)
    w, h = img.size
    if (w, h) == (size_img, size_img):
        i = j = 0
    else:
        i = torch.randint(0, h - size_img + 1, size=(1,), generator=generator).item()
        j = torch.randint(0, w - size_img + 1, size=(1,), generator=generator).item()
    return _to_tensor_normalize(img.crop((j, i, j + size_img, i + size_img)))


def _clip_resize_crop(imgs, size_img, x1y1):
    """visbackbone/video_transform.py Resize (size rule video_functional.py:94-101; its default 'nearest' maps to
    PIL.Image.BILINEAR, video_functional.py:83-86) + one crop window for the clip + ClipToTensor + Normalize -> (T, 3, S, S)."""
    im_w, im_h = imgs[0].size
    if not ((im_w <= im_h and im_w == size_img) or (im_h <= im_w and im_h == size_img)):
        if im_w < im_h:
            ow, oh = size_img, int(size_img * im_h / im_w)
        else:
            oh, ow = size_img, int(size_img * im_w / im_h)
        imgs = [im.resize((ow, oh), Image.BILINEAR) for im in imgs]
    im_w, im_h = imgs[0].size
    x1, y1 = x1y1(im_w, im_h)
    imgs = [im.crop((x1, y1, x1 + size_img, y1 + size_img)) for im in imgs]
    clip = torch.stack([torch.from_numpy(np.array(im, np.uint8, copy=True)).permute(2, 0, 1) for im in imgs], 1).float().div(255)   # (C, T, H, W)
    mean = torch.as_tensor(MEAN, dtype=clip.dtype)
    std = torch.as_tensor(STD, dtype=clip.dtype)
    clip.sub_(mean[:, None, None, None]).div_(std[:, None, None, None])
    return clip.permute(1, 0, 2, 3)


def vid_rand_crop(imgs, size_img, rng=random):
    """dataset.py:147-162: RandomCrop draws x, then y, with python's random."""
    def pick(w, h):
        if size_img > w or size_img > h:
            raise ValueError("Initial image size should be larger then cropped size")
        x1 = rng.randint(0, w - size_img)
        y1 = rng.randint(0, h - size_img)
        return x1, y1
    return _clip_resize_crop(imgs, size_img, pick)


def vid_center_crop(imgs, size_img):
    """dataset.py:132-145."""
    return _clip_resize_crop(imgs, size_img, lambda w, h: (int(round((w - size_img) / 2.)), int(round((h - size_img) / 2.))))


def sampling(start, end, n):
    """dataset.py:188-194."""
    if n == 1:
        return [int(round((start + end) / 2.))]
    if n < 1:
        raise Exception("behaviour not defined for n<2")
    step = (end - start) / float(n - 1)
    return [int(round(start + x * step)) for x in range(n)]


def temporal_sample(list_of_b, size_frame_cfg, random_sample=False, rng=random):
    """dataset.py:196-216."""
    max_size_frame = len(list_of_b)
    if max_size_frame == 1 or size_frame_cfg == max_size_frame:
        return list_of_b
    size_frame = min(size_frame_cfg, max_size_frame)
    size_clips = int(math.ceil(max_size_frame / size_frame))
    if random_sample:
        sampled_start = rng.choice(range(size_clips))
        sampled_end = min(sampled_start + (size_frame - 1) * size_clips, max_size_frame - 1)
    else:
        sampled_start, sampled_end = 0, max_size_frame - 1
    return [list_of_b[i] for i in sampling(sampled_start, sampled_end, size_frame)]


def get_img_or_video(list_of_b, size_frame, size_img, img_transform, split="train", rng=random, generator=None):
    """dataset.py:218-256 for the per-image transforms: (T, 3, S, S) fp32."""
    bufs = temporal_sample(list_of_b, size_frame, random_sample=(split == "train"), rng=rng)
    out, raw, t = [], [], None
    for b in bufs:
        img = str2img(b)
        if split == "train":
            t = rng.choice(img_transform)
        elif img_transform == ["vid_rand_crop"]:
            t = "vid_center_crop"
        elif img_transform == ["pad_resize"]:
            t = "pad_resize"
        else:
            t = "img_center_crop"
        if t.startswith("vid_"):
            raw.append(img)
        elif t == "pad_resize":
            out.append(pad_resize(img, size_img).unsqueeze(0))
        elif t == "img_center_crop":
            out.append(img_center_crop(img, size_img).unsqueeze(0))
        elif t == "img_rand_crop":
            out.append(img_rand_crop(img, size_img, generator).unsqueeze(0))
        else:
            raise NotImplementedError(t)
    if t == "vid_rand_crop":
        return vid_rand_crop(raw, size_img, rng)
    if t == "vid_center_crop":
        return vid_center_crop(raw, size_img)
    return torch.cat(out, 0)
