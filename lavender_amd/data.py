"""Input pipeline of the pretrain path: TSV(+lineidx) rows of base64 JPEG frames -> normalised
fp32 frames in HBM.  Mirrors the reference's dataset.py / main_pretrain_task_specific.py surface for this path:

    Dataset_Base       dataset.py:18-276      token ids, sampling / temporal_sample, transform choice, str2txt, *_mask_tok2txt
    Dataset_Pretrain   main_pretrain_task_specific.py:15-122   TSV row -> (frames, txt, mask), collate_batch
    get_dl             dataset.py:279-292     Random / Sequential / Distributed sampler semantics, batches of size_batch

What differs is WHERE the pixels are made: `__getitem__` only plans a sample -- it picks the frames, the transform and the crop
offsets, drawing from `random` / `torch` in exactly the reference's order -- and the batch is decoded in one call of the C-ABI
decoder (include/lavender_pipeline.h): base64 + Huffman decode on host threads, then IDCT, chroma upsampling, colour
conversion, antialiased resize, crop and normalisation as HIP kernels writing the (B, T, 3, S, S) tensor in HBM.  A prefetch
thread with its own stream keeps one batch ahead of the training step.  The frames are bit-identical to the reference's
(cv2 / PIL decode + torchvision transforms); tests/test_gpu_pipeline.py holds that against fixtures made with Pillow.

The clip transforms `vid_rand_crop` / `vid_center_crop` (visbackbone/video_transform.py) are planned here as well: one target
size and ONE crop window for all frames of the clip (python `random.randint` for x, then y); their pixel arithmetic is the image
path's (the reference's `Resize` default 'nearest' selects PIL.Image.BILINEAR through the swapped test at
video_functional.py:83-86).
"""
import ctypes as C
import math
import queue
import random
import threading

import numpy as np
import torch

from . import _lib as L

MEAN = (0.485, 0.456, 0.406)          # dataset.py:114-116
STD = (0.229, 0.224, 0.225)


class TsvFile:
    """Memory-mapped TSV with the reference's .lineidx (dataset.py:40-46; _tools/extract_tsv.py:20-27 writes the format)."""

    def __init__(self, tsv_path, lineidx_path=None):
        self._h = L.lib.lav_tsv_open(str(tsv_path).encode(), str(lineidx_path).encode() if lineidx_path else None)
        if not self._h:
            L.check(-1, "lav_tsv_open")
        self.path = str(tsv_path)

    def __len__(self):
        return int(L.lib.lav_tsv_rows(self._h))

    def offset(self, row):
        off = L.lib.lav_tsv_row_offset(self._h, int(row))
        if off < 0:
            L.check(-1, "lav_tsv_row_offset")
        return int(off)

    def fields(self, pos, max_fields=64):
        """[(address, length)] of the tab-separated, whitespace-stripped fields of the line at byte offset `pos`."""
        ptr = (L.vp * max_fields)()
        ln = (L.i64 * max_fields)()
        n = L.lib.lav_tsv_fields(self._h, int(pos), max_fields, ptr, ln)
        if n < 0:
            L.check(n, "lav_tsv_fields")
        if n > max_fields:
            return self.fields(pos, n)
        return [(int(ptr[i] or 0), int(ln[i])) for i in range(n)]

    def seek(self, pos):
        """dataset.py:44-46 seek_img_tsv: the fields as strings."""
        return [C.string_at(p, n).decode() if n else "" for p, n in self.fields(pos)]

    def close(self):
        if self._h and L is not None:                      # L is None during interpreter shutdown
            L.lib.lav_tsv_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


def _ref(buf):
    """(address, length, keep-alive) of a base64 JPEG given as a TSV field reference, bytes or str."""
    if isinstance(buf, tuple):
        return buf[0], buf[1], None
    if isinstance(buf, str):
        buf = buf.encode()
    keep = C.create_string_buffer(buf, len(buf))
    return C.addressof(keep), len(buf), keep


def jpeg_size(buf):
    """(width, height) as PIL.Image.size would report after dataset.py:177-186 str2img."""
    p, n, _keep = _ref(buf)
    w, h = L.i32(), L.i32()
    L.check(L.lib.lav_jpeg_peek(p, n, C.byref(w), C.byref(h)), "lav_jpeg_peek")
    return w.value, h.value


def resized_size(w, h, size):
    """torchvision Resize(int) on a (w, h) image: the shorter side becomes `size`, the other int(size * long / short)."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


class FramePlan:
    """One frame of a sample: the JPEG and the geometric transform the decoder applies to it."""
    __slots__ = ("buf", "pad_left", "pad_top", "resize_w", "resize_h", "crop_x", "crop_y")

    def __init__(self, buf, pad_left, pad_top, resize_w, resize_h, crop_x, crop_y):
        self.buf, self.pad_left, self.pad_top = buf, pad_left, pad_top
        self.resize_w, self.resize_h, self.crop_x, self.crop_y = resize_w, resize_h, crop_x, crop_y


class FrameDecoder:
    """Batch JPEG decoder + transforms on the GPU (lav_decoder_* of include/lavender_pipeline.h)."""

    def __init__(self, n_threads=4):
        self._h = L.lib.lav_decoder_create(int(n_threads))
        if not self._h:
            L.check(-1, "lav_decoder_create")
        self._mean = (L.f32 * 3)(*MEAN)
        self._std = (L.f32 * 3)(*STD)

    def decode(self, plans, size, out=None, slots=None, stream=None):
        """plans: FramePlan list -> (len(plans), 3, size, size) fp32 on the GPU (or written into `out` at frame slots `slots`).
        Enqueued on `stream` (default: torch's current stream); the host part (base64, Huffman) runs inside the call."""
        n = len(plans)
        if not torch.cuda.is_available():
            raise RuntimeError("lavender_amd runs on the MI355X only: the frame decoder has no CPU path")
        if out is None:
            out = torch.empty((n, 3, size, size), dtype=torch.float32, device="cuda")
        ptr, ln, xf, keep = (L.vp * n)(), (L.i64 * n)(), (L.FrameXform * n)(), []
        for i, p in enumerate(plans):
            a, k, ka = _ref(p.buf)
            keep.append(ka)
            ptr[i], ln[i] = a, k
            xf[i] = L.FrameXform(p.pad_left, p.pad_top, p.resize_w, p.resize_h, p.crop_x, p.crop_y, i if slots is None else int(slots[i]))
        st = (stream or torch.cuda.current_stream()).cuda_stream
        L.check(L.lib.lav_decoder_decode(self._h, st, n, ptr, ln, xf, size, size, self._mean, self._std, out.data_ptr()), "lav_decoder_decode")
        return out

    def failed_frames(self):
        """Indices (into the plan list) of the frames the last failed decode() call could not read."""
        buf = (L.i32 * 4096)()
        n = L.lib.lav_decoder_failed_frames(self._h, buf, 4096)
        return [buf[i] for i in range(max(0, min(n, 4096)))]

    def last_rgb(self, i):
        """Full-resolution RGB (H, W, 3) uint8 of frame i of the last batch (test tap)."""
        cap = 1 << 26
        buf = (C.c_uint8 * cap)()
        w, h = L.i32(), L.i32()
        L.check(L.lib.lav_decoder_read_rgb(self._h, int(i), buf, cap, C.byref(w), C.byref(h)), "lav_decoder_read_rgb")
        return np.frombuffer(buf, dtype=np.uint8, count=3 * w.value * h.value).reshape(h.value, w.value, 3).copy()

    def close(self):
        if self._h and L is not None:
            L.lib.lav_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Dataset_Base:
    """dataset.py:18-276 for the pretrain path."""

    def __init__(self, args, split="train", size_frame=4, tokzr=None):
        self.args, self.size_frame, self.split = args, size_frame, split
        if tokzr is None:
            raise ValueError("pass the tokenizer object (no network / model hub access in this build)")
        self.tokzr = tokzr
        (self.cls_token_id, self.sep_token_id, self.pad_token_id, self.mask_token_id, self.unk_token_id) = tokzr.convert_tokens_to_ids(
            [tokzr.cls_token, tokzr.sep_token, tokzr.pad_token, tokzr.mask_token, tokzr.unk_token])
        self.true_token_id = tokzr.convert_tokens_to_ids(["true"])[0]
        self.false_token_id = tokzr.convert_tokens_to_ids(["false"])[0]
        self._py_rng, self._torch_gen = random, None      # the process-global generators, as in a reference worker process

    def set_rng(self, py_rng, torch_gen):
        """Private generators for the sampling / crop draws (the reference's DataLoader workers are separate processes with
        their own seeds; a prefetch THREAD must not interleave its draws with the trainer's masking / dropout draws)."""
        self._py_rng, self._torch_gen = py_rng, torch_gen

    # ---- text (dataset.py:83-105, 258-276) ----------------------------------------------------------------------------
    def append_mask_tok2txt(self, txt, mask):
        one = torch.LongTensor([1])
        return torch.cat([txt, self.mask_token_id * one], -1), torch.cat([mask, one], -1)

    def prepend_mask_tok2txt(self, txt, mask):
        one = torch.LongTensor([1])
        return torch.cat([self.mask_token_id * one, txt], -1), torch.cat([one, mask], -1)

    def replace_cls_w_mask(self, txt, mask):
        one = torch.LongTensor([1])
        return torch.cat([self.mask_token_id * one, txt[1:]], -1), torch.cat([one, mask[1:]], -1)

    def str2txt(self, s):
        txt = self.tokzr.encode(s, padding="max_length", max_length=self.args.size_txt, truncation=True)
        mask = torch.LongTensor([1 if w != self.pad_token_id else 0 for w in txt])
        txt = torch.LongTensor(txt)
        assert len(txt[txt == self.sep_token_id]) == 1, f"{txt}"
        return txt, mask

    # ---- frames (dataset.py:188-256) -----------------------------------------------------------------------------------
    def sampling(self, start, end, n):
        """n evenly spaced integer positions from start to end inclusive (python round(): halves go to the even neighbour); the
        midpoint for n == 1 (dataset.py:188-194)."""
        if n < 1:
            raise Exception("behaviour not defined for n<2")
        if n == 1:
            return [int(round(0.5 * (start + end)))]
        pitch = (end - start) / float(n - 1)
        return [int(round(start + i * pitch)) for i in range(n)]

    def temporal_sample(self, list_of_b, random_sample=False):
        """Which frames of a row become the clip (dataset.py:196-216): all of them when the row has exactly size_frame frames (or
        one); otherwise size_frame positions spread over the row -- from a random phase in [0, ceil(total / size_frame)) with that
        stride in training, over the whole row in evaluation."""
        total = len(list_of_b)
        if total == 1 or total == self.size_frame:
            return list_of_b
        want = min(self.size_frame, total)
        stride = int(math.ceil(total / want))
        first, last = 0, total - 1
        if random_sample:
            first = self._py_rng.choice(range(stride))
            last = min(first + (want - 1) * stride, total - 1)
        return [list_of_b[i] for i in self.sampling(first, last, want)]

    def _plan_one(self, buf, transform):
        S = int(self.args.size_img)
        w, h = jpeg_size(buf)
        if transform == "pad_resize":                     # dataset.py:107-118: Pad to a square, Resize([S, S])
            pl, pt = (0, (w - h) // 2) if w > h else ((h - w) // 2, 0)
            return FramePlan(buf, pl, pt, S, S, 0, 0)
        rw, rh = resized_size(w, h, S)                    # Resize(S): dataset.py:121,166
        if transform == "img_center_crop":                # CenterCrop: int(round((h - S) / 2.0)), Python rounding
            return FramePlan(buf, 0, 0, rw, rh, int(round((rw - S) / 2.0)), int(round((rh - S) / 2.0)))
        if transform == "img_rand_crop":                  # RandomCrop.get_params: torch.randint for the row, then the column
            assert self.split == "train"
            if rw == S and rh == S:
                return FramePlan(buf, 0, 0, rw, rh, 0, 0)
            i = torch.randint(0, rh - S + 1, size=(1,), generator=self._torch_gen).item()
            j = torch.randint(0, rw - S + 1, size=(1,), generator=self._torch_gen).item()
            return FramePlan(buf, 0, 0, rw, rh, j, i)
        raise ValueError(f"unknown img_transform {transform!r}")

    def _plan_clip(self, bufs, random_crop):
        """vid_rand_crop / vid_center_crop (dataset.py:132-162): Resize(size_img) with the size rule of video_functional.py:94-101
        taken from the FIRST frame, one crop window for the whole clip, ClipToTensor, Normalize."""
        S = int(self.args.size_img)
        w, h = jpeg_size(bufs[0])
        if (w <= h and w == S) or (h <= w and h == S):
            rw, rh = w, h
        elif w < h:
            rw, rh = S, int(S * h / w)
        else:
            rw, rh = int(S * w / h), S
        if S > rw or S > rh:
            raise ValueError(f"Initial image size should be larger then cropped size but got cropped sizes : ({S}, {S}) while initial image is ({rw}, {rh})")
        if random_crop:
            assert self.split == "train"
            x1 = self._py_rng.randint(0, rw - S)
            y1 = self._py_rng.randint(0, rh - S)
        else:
            x1, y1 = int(round((rw - S) / 2.)), int(round((rh - S) / 2.))
        return [FramePlan(b, 0, 0, rw, rh, x1, y1) for b in bufs]

    def get_img_or_video(self, list_of_b):
        """dataset.py:218-256, as a plan: the same frames, transforms and crop offsets (same RNG draws in the same order);
        the pixels are produced by FrameDecoder.decode at batch time."""
        bufs = self.temporal_sample(list_of_b, random_sample=(self.split == "train"))
        choice = []
        for _ in bufs:
            if self.split == "train":
                choice.append(self._py_rng.choice(self.args.img_transform))
            elif self.args.img_transform == ["vid_rand_crop"]:
                choice.append("vid_center_crop")
            elif self.args.img_transform == ["pad_resize"]:
                choice.append("pad_resize")
            else:
                choice.append("img_center_crop")
        if any(c.startswith("vid_") for c in choice):
            if len(set(choice)) != 1:                     # the reference would fail in T.cat / Compose on such a mix
                raise ValueError(f"clip transforms cannot be mixed with per-image transforms inside one clip: {choice}")
            return self._plan_clip(bufs, choice[0] == "vid_rand_crop")
        return [self._plan_one(b, c) for b, c in zip(bufs, choice)]


class Dataset_Pretrain(Dataset_Base):
    """main_pretrain_task_specific.py:15-122 with explicit file paths (the reference derives them from dataset names)."""

    def __init__(self, args, txt, tsv_path, lineidx_path, split="train", dataset="webvid2.5m", tokzr=None):
        super().__init__(args, split=split, size_frame=args.size_frame, tokzr=tokzr)
        if dataset in ["cc3m", "coco", "vg", "cc12m"]:
            self.size_frame = 1
        self.dataset = dataset
        self.txt = txt[self.split] if isinstance(txt, dict) and self.split in txt else txt
        self.tsv = TsvFile(tsv_path, lineidx_path)
        self.lineidx = [self.tsv.offset(i) for i in range(len(self.tsv))]

    def __len__(self):
        return len(self.lineidx)

    def __getitem__(self, idx):
        item = self.tsv.fields(self.lineidx[idx])
        vid = C.string_at(item[0][0], item[0][1]).decode()
        bufs = item[2:] if self.dataset in ["webvid10m", "webvid10m_filtered"] and self.split == "train" else item[1:]
        raw_txt = self.txt[vid][0] if vid in self.txt else ""
        try:
            plans = self.get_img_or_video(bufs)
        except L.LavenderHipError as e:                   # main_pretrain_task_specific.py:98-106: zeros for unreadable frames
            print(f"Failed to load image binaries for video {vid} for dataset {self.dataset}, split {self.split}, {e}")
            plans = None
        txt, mask = self.str2txt(raw_txt)
        return plans, txt, mask


def reference_paths(dataset, split, part, data_dir):
    """(tsv, lineidx) file names of main_pretrain_task_specific.py:29-70 for a dataset / split / part."""
    if dataset == "webvid10m":
        base = f"{data_dir}/_webvid10m-tsv_frame4/webvid10m-{part + 1:03d}.img" if split == "train" else f"{data_dir}/webvid2.5m_val"
    elif dataset == "webvid10m_filtered":
        base = f"{data_dir}/image-1{part:04d}" if split == "train" else f"{data_dir}/webvid2.5m_val"
    elif dataset == "cc12m":
        base = f"{data_dir}/train.{part}.62.img" if split == "train" else f"{data_dir}/cc3m_val"
    else:
        base = f"{data_dir}/{dataset}_train_{part}" if split == "train" else f"{data_dir}/{dataset}_val"
    return base + ".tsv", base + ".lineidx"


class Dataset_Pretrain_MLM(Dataset_Pretrain):
    """main_pretrain_mlm.py:15-25: the text gets one [MASK] appended (the VTM answer slot), so txt has size_txt + 1 tokens.
    Constructor arguments as in the reference; the files are located with its naming scheme."""

    def __init__(self, args, txt, dataset, split, part=None, data_dir=None, tokzr=None):
        tsv, idx = reference_paths(dataset, split, part, data_dir if data_dir is not None else args.data_dir)
        super().__init__(args, txt, tsv, idx, split=split, dataset=dataset, tokzr=tokzr)
        self.part, self.data_dir = part, data_dir if data_dir is not None else args.data_dir

    def str2txt(self, s):
        txt, mask = super().str2txt(s)
        return self.append_mask_tok2txt(txt, mask)


class _Sampler:
    """dataset.py:279-287: DistributedSampler(shuffle=train) / RandomSampler / SequentialSampler index order."""

    def __init__(self, n, train, distributed, rank=0, world=1, seed=0):
        self.n, self.train, self.distributed, self.rank, self.world, self.seed, self.epoch = n, train, distributed, rank, world, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        if self.distributed:
            if self.train:
                g = torch.Generator()
                g.manual_seed(self.seed + self.epoch)
                idx = torch.randperm(self.n, generator=g).tolist()
            else:
                idx = list(range(self.n))
            total = -(-self.n // self.world) * self.world
            pad = total - len(idx)
            if pad:
                idx += (idx * math.ceil(pad / len(idx)))[:pad]
            return idx[self.rank:total:self.world]
        if self.train:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
            return torch.randperm(self.n, generator=g).tolist()
        return list(range(self.n))


class PretrainLoader:
    """get_dl (dataset.py:279-292) for Dataset_Pretrain: batches {"img": (B, T, 3, S, S) fp32 in HBM, "txt", "mask" pinned host}.
    One batch is decoded ahead on a side stream by a prefetch thread (n_workers = host decode threads)."""

    def __init__(self, ds, args, rank=0, world=1, prefetch=True):
        self.ds, self.args = ds, args
        self.sampler = _Sampler(len(ds), ds.split == "train", bool(getattr(args, "distributed", False)), rank, world)
        self.decoder = FrameDecoder(max(1, int(getattr(args, "n_workers", 4))))
        self.prefetch = prefetch
        self._stream = None

    def __len__(self):
        """Batches per epoch on this rank (the last one may be short: drop_last is False in dataset.py:287-291)."""
        n = -(-len(self.ds) // self.sampler.world) if self.sampler.distributed else len(self.ds)
        return -(-n // int(self.args.size_batch))

    def _collate(self, items, stream):
        S, T = int(self.args.size_img), max(len(p) for p, _, _ in items if p is not None) if any(p for p, _, _ in items) else self.ds.size_frame
        B = len(items)
        with torch.cuda.stream(stream):                    # owned by the side stream's pool; the consumer records its own use
            img = torch.empty((B, T, 3, S, S), dtype=torch.float32, device="cuda")
        plans, slots, owner = [], [], []
        for b, (p, _, _) in enumerate(items):
            if p is None or len(p) != T:
                if p is not None:
                    raise ValueError(f"sample {b} has {len(p)} frames, the batch {T} (T.stack of the reference fails as well)")
                with torch.cuda.stream(stream):
                    img[b].zero_()
                continue
            plans += p
            slots += [b * T + t for t in range(T)]
            owner += [b] * T
        while plans:
            try:
                self.decoder.decode(plans, S, out=img, slots=slots, stream=stream)
                break
            except L.LavenderHipError as e:
                # a frame whose header parsed but whose entropy stream is truncated / corrupt only fails here: the reference
                # substitutes a zero clip for THAT sample (main_pretrain_task_specific.py:95-106), the rest of the batch decodes
                bad = {owner[i] for i in self.decoder.failed_frames()}
                if not bad:
                    raise
                print(f"Failed to decode image binaries of sample(s) {sorted(bad)} of a batch for dataset {getattr(self.ds, 'dataset', '?')}, "
                      f"split {self.ds.split}: {e}")
                with torch.cuda.stream(stream):
                    for b in bad:
                        img[b].zero_()
                keep = [i for i, b in enumerate(owner) if b not in bad]
                plans, slots, owner = [plans[i] for i in keep], [slots[i] for i in keep], [owner[i] for i in keep]
        # txt / mask stay on the host (pinned), as the reference's DataLoader hands them over: the trainer masks them on the
        # CPU before the copy (main_pretrain_mlm.py:215-220)
        batch = {"img": img, "txt": torch.stack([t for _, t, _ in items]).pin_memory(),
                 "mask": torch.stack([m for _, _, m in items]).pin_memory()}
        ev = torch.cuda.Event()
        ev.record(stream)
        return batch, ev

    def _batches(self):
        idx = self.sampler.indices()
        bs = int(self.args.size_batch)
        for i in range(0, len(idx), bs):
            yield [self.ds[j] for j in idx[i:i + bs]]

    def __iter__(self):
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        if not self.prefetch:
            for items in self._batches():
                batch, ev = self._collate(items, self._stream)
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                batch["img"].record_stream(cur)          # allocated in the side stream's pool, consumed on the current stream
                yield batch
            return
        q = queue.Queue(maxsize=2)
        dev = torch.cuda.current_device()
        if self.ds._torch_gen is None:                     # worker-style private seeds, drawn once from the global generator
            base = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(base)
            self.ds.set_rng(random.Random(base), g)

        stop = threading.Event()

        def put(x):                                        # gives up when the consumer has gone away
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                torch.cuda.set_device(dev)
                for items in self._batches():
                    if stop.is_set() or not put(self._collate(items, self._stream)):
                        return
                put(None)
            except BaseException as e:                     # surfaced in the consumer
                put(e)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                got = q.get()
                if got is None:
                    break
                if isinstance(got, BaseException):
                    raise got
                batch, ev = got
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                batch["img"].record_stream(cur)
                yield batch
        finally:
            # a consumer that stops early (break / exception / GeneratorExit) must not leave the worker blocked in q.put holding
            # GPU batches, sharing the decoder and the side stream with the next __iter__
            stop.set()
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            th.join()


def get_dl(ds, args, rank=0, world=1):
    return PretrainLoader(ds, args, rank=rank, world=world)
