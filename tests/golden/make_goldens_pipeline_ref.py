"""Reference pin of the input-pipeline oracle (CONTAINER ONLY -- imports /root/reference/dataset.py; never runs on the GPU box).

The reference's `dataset.Dataset_Base` imports here with the same stub recipe make_goldens.import_reference() uses (cv2 and
torchvision are absent: `str2img` then takes the reference's own `except` branch (PIL), and the clip transforms of
visbackbone/video_transform.py need neither).  This script runs the REFERENCE's
    sampling / temporal_sample (dataset.py:188-216), str2img (:177-186), vid_center_crop / vid_rand_crop (:132-162),
    get_img_or_video (:218-256, img_transform == ["vid_rand_crop"], train seeded and val; and a three-entry list, which pins the draw
    order of random.choice at :225)
on the two fixture rows of msrvtt_2rows.tsv and writes their outputs to pipeline_ref_pin.npz.  tests/test_oracle_golden.py holds
oracle/pipeline_ref.py to these vectors; tests/test_gpu_pipeline.py holds the HIP pipeline to the oracle.
pad_resize / img_center_crop / img_rand_crop call torchvision.transforms, which this image does not have: those three stay
"restated from torchvision's published source, unpinned" (oracle/pipeline_ref.py header).

    python tests/golden/make_goldens_pipeline_ref.py
"""
import sys
sys.dont_write_bytecode = True
import importlib
import os
import random
import zlib
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_goldens as MG  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402

STUBS = ["cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "skimage", "skimage.transform",
         "skimage.feature", "av", "easydict", "toolz", "toolz.sandbox", "tensorboardX", "fairscale", "fairscale.nn",
         "fairscale.nn.misc", "deepspeed", "apex", "progressbar", "future", "future.utils", "ete3", "deprecated",
         "matplotlib.pyplot", "addict", "yapf", "yapf.yapflib", "yapf.yapflib.yapf_api"]


def sub(a, n=4096, seed=7):
    flat = np.asarray(a).reshape(-1)
    idx = np.random.RandomState(seed + flat.size % 9973).permutation(flat.size)[:n]
    return flat[idx]


def sums(x):
    x = np.asarray(x, dtype=np.float64)
    return np.array([x.sum(), np.abs(x).sum()])


def main():
    ref = MG.import_reference()
    for m in STUBS:                                   # import_reference() removes its stubs again; dataset.py needs them back
        try:
            importlib.import_module(m)
        except Exception:
            sys.modules[m] = MagicMock(name=m)
    import dataset as DS                                # /root/reference/dataset.py

    def ds(split, size_frame, transforms):
        args = ref.EasyDict(size_img=224, img_transform=transforms, tokenizer=None)
        return DS.Dataset_Base(args, split, size_frame, MG.Tok())

    res = {}
    d_train, d_val = ds("train", 4, ["vid_rand_crop"]), ds("val", 4, ["vid_rand_crop"])
    d_train2 = ds("train", 4, ["vid_rand_crop", "vid_rand_crop", "vid_rand_crop"])
    # ---- integer paths: sampling / temporal_sample ----------------------------------------------------------------------
    grid = [(s, e, n) for s in (0, 1, 3) for e in (3, 4, 9, 31, 100) for n in (1, 2, 3, 4, 5, 8) if e >= s]
    res["sampling_args"] = np.array(grid, dtype=np.int64)
    res["sampling_out"] = np.concatenate([np.array(d_train.sampling(*g), dtype=np.int64) for g in grid])
    # L < size_frame (L > 1) is undefined in the reference: its error print reads `size_frame` before assignment (dataset.py:201-202)
    ts_cases = [(L, sf) for L in (1, 2, 4, 5, 6, 9, 10, 32) for sf in (1, 2, 4, 5, 8) if L == 1 or L >= sf]
    res["temporal_cases"] = np.array(ts_cases, dtype=np.int64)
    ev, tr = [], []
    for ci, (L, sf) in enumerate(ts_cases):
        frames = list(range(L))
        ev.append(np.array(ds("val", sf, ["vid_rand_crop"]).temporal_sample(frames, random_sample=False), dtype=np.int64))
        random.seed(100 + ci)
        tr.append(np.array(ds("train", sf, ["vid_rand_crop"]).temporal_sample(frames, random_sample=True), dtype=np.int64))
    res["temporal_eval_out"] = np.concatenate(ev)
    res["temporal_eval_len"] = np.array([len(x) for x in ev], dtype=np.int64)
    res["temporal_train_out"] = np.concatenate(tr)
    res["temporal_train_len"] = np.array([len(x) for x in tr], dtype=np.int64)
    # ---- byte / pixel paths on the two TSV fixture rows -------------------------------------------------------------------
    offs = [int(x) for x in open(f"{HERE}/msrvtt_2rows.lineidx")]
    for r, pos in enumerate(offs):
        item = PR.read_row(f"{HERE}/msrvtt_2rows.tsv", pos)           # TSV seek is the oracle's; everything after is the reference's
        frames = item[1:]
        for fi, b in enumerate(frames):
            rgb = np.array(d_val.str2img(b))
            res[f"ref_{r}_{fi}_rgb_sub"] = sub(rgb)
            res[f"ref_{r}_{fi}_rgb_sum"] = np.array([rgb.astype(np.int64).sum(), zlib.adler32(rgb.tobytes())], dtype=np.int64)
            res[f"ref_{r}_{fi}_rgb_shape"] = np.array(rgb.shape, dtype=np.int64)
        imgs = [d_val.str2img(b) for b in frames[:4]]
        x = d_val.vid_center_crop(list(imgs)).numpy()
        res[f"ref_vid_center_{r}_sub"], res[f"ref_vid_center_{r}_sum"] = sub(x), sums(x)
        random.seed(21 + r)
        x = d_train.vid_rand_crop(list(imgs)).numpy()
        res[f"ref_vid_rand_{r}_sub"], res[f"ref_vid_rand_{r}_sum"] = sub(x), sums(x)
        random.seed(9 + r)
        x = d_train.get_img_or_video(frames).numpy()
        res[f"ref_sample_train_{r}_sub"], res[f"ref_sample_train_{r}_sum"] = sub(x), sums(x)
        res[f"ref_sample_train_{r}_shape"] = np.array(x.shape, dtype=np.int64)
        x = d_val.get_img_or_video(frames).numpy()
        res[f"ref_sample_val_{r}_sub"], res[f"ref_sample_val_{r}_sum"] = sub(x), sums(x)
        # a THREE-entry transform list: random.choice (dataset.py:225) then draws from the same `random` stream per frame, between the
        # temporal sampling and the crop offsets -- one or two entries reject a 32-bit draw when its top bit is set, three entries when its top TWO bits are
        # set, so the number of draws consumed -- and the crop that follows -- differs: this pins the draw ORDER of the choice itself.  (Only lists of 'vid_rand_crop' are runnable
        # here: every other train-mode entry calls torchvision, and a mixed list would cat PIL images with tensors in the reference.)
        random.seed(33 + r)
        x2 = d_train2.get_img_or_video(frames).numpy()
        res[f"ref_sample_train2_{r}_sub"], res[f"ref_sample_train2_{r}_sum"] = sub(x2), sums(x2)
        random.seed(33 + r)
        y2 = PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop", "vid_rand_crop", "vid_rand_crop"], "train", random, None).numpy()
        random.seed(33 + r)
        y1 = PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop"], "train", random, None).numpy()
        print(f"row {r}: three-entry list: oracle vs reference max|d| {np.abs(x2 - y2).max():.3g}; differs from the one-entry draw: {not np.array_equal(y1, y2)}")
        assert np.array_equal(x2, y2)
        # self-check while the reference is in memory: the oracle must agree exactly
        random.seed(9 + r)
        y = PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop"], "train", random, None).numpy()
        z = PR.get_img_or_video(frames, 4, 224, ["vid_rand_crop"], "val", random, None).numpy()
        random.seed(9 + r)
        xr = d_train.get_img_or_video(frames).numpy()
        print(f"row {r}: oracle vs reference max|d| train {np.abs(xr - y).max():.3g}  val {np.abs(x - z).max():.3g}")
        assert np.array_equal(xr, y) and np.array_equal(x, z)
    np.savez_compressed(f"{HERE}/pipeline_ref_pin.npz", **res)
    print("written", len(res), "arrays;", os.path.getsize(f"{HERE}/pipeline_ref_pin.npz"), "bytes")


if __name__ == "__main__":
    main()
