"""Input pipeline on the GPU (include/lavender_pipeline.h, lavender_amd/data.py) against the Pillow / torch CPU restatement of
the reference's dataset.py (oracle/pipeline_ref.py) and the committed fixtures (tests/golden/pipeline_frames.npz,
msrvtt_2rows.tsv: two rows of the reference's own sample TSV).  Byte / integer work: the bar is bit-exact; the final fp32 frames
(three IEEE operations per value) must be equal as well."""
import base64
import os
import random
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class WordTok:
    """Whitespace tokenizer stub with the encode() signature dataset.py:270-272 uses."""
    cls_token = "[CLS]"; sep_token = "[SEP]"; pad_token = "[PAD]"; mask_token = "[MASK]"; unk_token = "[UNK]"
    ids = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103, "true": 2995, "false": 6270}

    def convert_tokens_to_ids(self, toks):
        return [self.ids[t] for t in toks]

    def encode(self, s, padding=None, max_length=None, truncation=None):
        ids = [101] + [1000 + (zlib.crc32(w.encode()) % 20000) for w in s.split()][:max_length - 2] + [102]
        return ids + [0] * (max_length - len(ids))


def sub(a, n=4096, seed=7):
    flat = np.asarray(a).reshape(-1)
    idx = np.random.RandomState(seed + flat.size % 9973).permutation(flat.size)[:n]
    return flat[idx]


def _args(**kw):
    from lavender_amd.args import EasyDict
    a = EasyDict(size_img=224, size_frame=4, size_txt=32, img_transform=["img_rand_crop"], size_batch=2, n_workers=4, distributed=False)
    a.update(kw)
    return a


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "pipeline_frames.npz"))


def test_synthetic_jpegs_decode_and_transforms_bit_exact(gold):
    """4:4:4 / 4:2:2 / 4:2:0 / grey, odd sizes, optimised Huffman tables, restart markers, 1x1: decoded frame and every
    transform output equal the CPU path's."""
    from lavender_amd import data as D
    dec = D.FrameDecoder(2)
    ds = D.Dataset_Base(_args(), split="train", tokzr=WordTok())
    for name in gold["syn_names"].tolist():
        b64 = base64.b64encode(gold[f"syn_{name}_jpg"].tobytes())
        ref = gold[f"syn_{name}_rgb"]
        h, w = ref.shape[:2]
        assert D.jpeg_size(b64) == (w, h)
        for S in (8, 24):
            ds.args.size_img = S
            for t in ("pad_resize", "img_center_crop", "img_rand_crop"):
                key = f"syn_{name}_{t}_{S}"
                if key not in gold.files:
                    continue
                g = torch.Generator()
                g.manual_seed(S)
                ds.set_rng(random, g)
                plan = ds._plan_one(b64, t)
                out = dec.decode([plan], S)
                torch.cuda.synchronize()
                rgb = dec.last_rgb(0)[plan.pad_top:plan.pad_top + h, plan.pad_left:plan.pad_left + w]
                assert np.array_equal(rgb, ref), f"{name}: decoded frame differs in {(rgb != ref).sum()} bytes (max {np.abs(rgb.astype(int) - ref).max()})"
                got = out[0].cpu().numpy()
                d = np.abs(got - gold[key]).max()
                assert d == 0.0, f"{key}: max|d| {d}"


def test_tsv_rows_match_reference_fixture(gold, golden_dir):
    """Two rows of the reference's sample TSV: ids, frame sizes, decoded frames and the eval transforms at 224."""
    from lavender_amd import data as D
    tsv = D.TsvFile(os.path.join(golden_dir, "msrvtt_2rows.tsv"), os.path.join(golden_dir, "msrvtt_2rows.lineidx"))
    assert len(tsv) == 2
    dec = D.FrameDecoder(4)
    ds = D.Dataset_Base(_args(), split="val", tokzr=WordTok())
    for r in range(2):
        f = tsv.fields(tsv.offset(r))
        assert tsv.seek(tsv.offset(r))[0] == gold["tsv_ids"][r] and len(f) == 6
        for t in ("pad_resize", "img_center_crop"):
            plans = [ds._plan_one(b, t) for b in f[1:]]
            out = dec.decode(plans, 224)
            torch.cuda.synchronize()
            for i in range(5):
                if t == "img_center_crop":
                    rgb = dec.last_rgb(i)
                    assert np.array_equal(sub(rgb), gold[f"tsv_{r}_{i}_rgb_sub"])
                    assert int(rgb.astype(np.int64).sum()) == int(gold[f"tsv_{r}_{i}_rgb_sum"][0]) and zlib.adler32(rgb.tobytes()) == int(gold[f"tsv_{r}_{i}_rgb_sum"][1])
                x = out[i].cpu().numpy()
                assert np.array_equal(sub(x), gold[f"tsv_{r}_{i}_{t}_sub"]), (r, i, t)
                s = gold[f"tsv_{r}_{i}_{t}_sum"]
                assert abs(x.astype(np.float64).sum() - s[0]) <= 1e-6 * s[1]


def test_train_sample_plan_and_loader(gold, golden_dir):
    """Dataset_Pretrain in train mode with the reference's draw order (temporal start, transform choice, crop row, crop column)
    reproduces the seeded CPU sample; the prefetching loader yields the same batches as the synchronous one."""
    from lavender_amd import data as D
    from oracle import pipeline_ref as PR
    tsvp, idxp = os.path.join(golden_dir, "msrvtt_2rows.tsv"), os.path.join(golden_dir, "msrvtt_2rows.lineidx")
    txt = {"train": {k: [f"a video of {k} playing"] for k in gold["tsv_ids"].tolist()}}
    args = _args(img_transform=["img_rand_crop", "pad_resize", "img_center_crop"])
    ds = D.Dataset_Pretrain(args, txt, tsvp, idxp, split="train", dataset="msrvtt", tokzr=WordTok())
    dec = D.FrameDecoder(4)
    for r in range(2):
        random.seed(5 + r)
        g = torch.Generator()
        g.manual_seed(5 + r)
        ds.set_rng(random, g)
        plans, t, m = ds[r]
        assert len(plans) == 4 and t.shape == (32,) and int(m.sum()) == 7 and int(t[0]) == 101
        x = dec.decode(plans, 224).cpu().numpy()
        assert np.array_equal(sub(x), gold[f"tsv_train_{r}_sub"])
        # and live against the CPU restatement on the same seeds
        random.seed(5 + r)
        g.manual_seed(5 + r)
        ref = PR.get_img_or_video(PR.read_row(tsvp, ds.lineidx[r])[1:], 4, 224, args.img_transform, "train", random, g).numpy()
        assert np.array_equal(x, ref)
    # clip transforms: one window per clip from python's RNG (train), centre window (eval)
    args_v = _args(img_transform=["vid_rand_crop"])
    for r in range(2):
        for split in ("train", "val"):
            dsv = D.Dataset_Pretrain(args_v, {split: txt["train"]}, tsvp, idxp, split=split, dataset="msrvtt", tokzr=WordTok())
            random.seed(9 + r)
            dsv.set_rng(random, None)
            plans, _, _ = dsv[r]
            assert len({(p.crop_x, p.crop_y, p.resize_w, p.resize_h) for p in plans}) == 1
            x = dec.decode(plans, 224).cpu().numpy()
            assert np.array_equal(sub(x), gold[f"tsv_vid_{split}_{r}_sub"]), (r, split)
    with pytest.raises(ValueError):
        dsm = D.Dataset_Pretrain(_args(img_transform=["vid_rand_crop", "pad_resize"]), txt, tsvp, idxp, split="train", dataset="msrvtt", tokzr=WordTok())
        for seed in range(20):                             # some draw mixes the two kinds inside one clip
            random.seed(seed)
            dsm.set_rng(random, None)
            dsm[0]
    # loader: synchronous vs prefetching, same private seeds
    batches = []
    for prefetch in (False, True):
        g = torch.Generator()
        g.manual_seed(11)
        ds.set_rng(random.Random(11), g)
        torch.manual_seed(3)                               # RandomSampler order
        dl = D.PretrainLoader(ds, args, prefetch=prefetch)
        got = []
        for b in dl:
            assert b["img"].is_cuda and b["txt"].is_pinned() and b["img"].dtype == torch.float32
            got.append({k: v.clone() for k, v in b.items()})
        assert len(got) == 1 and got[0]["img"].shape == (2, 4, 3, 224, 224) and got[0]["txt"].shape == (2, 32)
        batches.append(got[0])
    for k in ("img", "txt", "mask"):
        assert torch.equal(batches[0][k], batches[1][k]), k


def test_decoder_rejects_bad_input():
    from lavender_amd import data as D
    from lavender_amd._lib import LavenderHipError
    dec = D.FrameDecoder(1)
    with pytest.raises(LavenderHipError):
        dec.decode([D.FramePlan(base64.b64encode(b"not a jpeg at all"), 0, 0, 8, 8, 0, 0)], 8)
    with pytest.raises(LavenderHipError):
        D.jpeg_size(b"")


def test_loader_substitutes_a_zero_clip_for_a_sample_with_a_corrupt_entropy_stream(tmp_path, gold, golden_dir):
    """main_pretrain_task_specific.py:95-106: a sample whose frames cannot be decoded becomes a zero clip, the rest of the batch is
    untouched.  A JPEG whose header parses but whose entropy stream is truncated is only detected inside lav_decoder_decode: the
    loader asks lav_decoder_failed_frames which frames failed, zeroes their samples and decodes the others."""
    from lavender_amd import data as D
    tsvp = os.path.join(golden_dir, "msrvtt_2rows.tsv")
    rows = [l.rstrip("\n").split("\t") for l in open(tsvp)]
    jpg = base64.b64decode(rows[1][2])
    sos = jpg.index(b"\xff\xda")
    jpg = jpg[:sos + 300] + b"\xfe" * 3000 + jpg[sos + 3300:]             # header intact (D.jpeg_size works), the scan is garbage:
    rows[1][2] = base64.b64encode(jpg).decode()                            # long runs of one-bits are not a Huffman code
    assert D.jpeg_size(rows[1][2].encode()) == (320, 240)
    bad = os.path.join(tmp_path, "bad.tsv")
    with open(bad, "w") as f:
        for r in rows:
            f.write("\t".join(r) + "\n")
    txt = {"val": {k: [f"a video of {k} playing"] for k in gold["tsv_ids"].tolist()}}
    args = _args(img_transform=["img_center_crop"])
    out = {}
    for name, path in (("good", tsvp), ("bad", bad)):
        ds = D.Dataset_Pretrain(args, txt, path, None, split="val", dataset="msrvtt", tokzr=WordTok())
        for prefetch in (False, True):
            batches = list(D.PretrainLoader(ds, args, prefetch=prefetch))
            assert len(batches) == 1
            out[name, prefetch] = batches[0]["img"].clone()
    for prefetch in (False, True):
        g, b = out["good", prefetch], out["bad", prefetch]
        assert torch.equal(b[0], g[0]) and float(g[1].abs().sum()) > 0 and float(b[1].abs().sum()) == 0.0
    # a consumer that stops early does not leave the prefetch thread behind
    import threading
    n0 = threading.active_count()
    ds = D.Dataset_Pretrain(args, txt, tsvp, None, split="val", dataset="msrvtt", tokzr=WordTok())
    a1 = _args(img_transform=["img_center_crop"]); a1.size_batch = 1
    for _ in D.PretrainLoader(ds, a1, prefetch=True):
        break
    import time
    time.sleep(0.5)
    assert threading.active_count() <= n0


def test_agent_trains_and_evaluates_from_the_tsv_loader(tmp_path, golden_dir):
    """End to end as main_pretrain_mlm.py:251-328 drives it: Dataset_Pretrain_MLM over files named by the reference's scheme,
    the prefetching GPU loader, host masking, prepare_batch, step (train) and the eval branch, on the micro model."""
    import json
    import lavender_amd as LA
    from lavender_amd import data as D
    from tests.helpers import make_args
    rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(golden_dir, "msrvtt_2rows.tsv"))]
    txt = {"train": {}, "val": {}}
    for split, n in (("train_0", 6), ("val", 2)):
        with open(tmp_path / f"fix_{split}.tsv", "w") as f, open(tmp_path / f"fix_{split}.lineidx", "w") as fi:
            for i in range(n):
                r = rows[i % 2]
                fi.write("%d\n" % f.tell())
                f.write("\t".join([f"{split}_{i}"] + r[1:]) + "\n")
                txt[split.split("_")[0]][f"{split}_{i}"] = [f"clip {i} shows a person doing thing number {i}"]
    (tmp_path / "txt_fix.json").write_text(json.dumps(txt))
    args = make_args("micro", "micro", 2, size_txt=15, size_frame=4, img_transform=["img_rand_crop"], n_workers=2, data_dir=str(tmp_path),
                     distributed=False, max_iter=10)
    tok = WordTok()
    assert D.reference_paths("fix", "train", 0, str(tmp_path))[0].endswith("fix_train_0.tsv")
    ds_tr = D.Dataset_Pretrain_MLM(args, txt, "fix", "train", 0, data_dir=str(tmp_path), tokzr=tok)
    ds_vl = D.Dataset_Pretrain_MLM(args, txt, "fix", "val", data_dir=str(tmp_path), tokzr=tok)
    assert len(ds_tr) == 6 and len(ds_vl) == 2
    _, t, m = ds_tr[0]
    assert t.shape == (16,) and int(t[-1]) == 103 and int(m[-1]) == 1          # [MASK] appended: the VTM answer slot
    torch.manual_seed(0)
    model = LA.LAVENDER_Pretrain_MLM(args, tok).cuda()
    agent = LA.Agent_Pretrain_MLM(args, model)
    ls = agent.go_dl(1, D.get_dl(ds_tr, args), True)
    ac = agent.go_dl(1, D.get_dl(ds_vl, args), False)
    print("train", ls, "eval", ac)
    assert all(np.isfinite(v) for v in ls.values()) and set(ls) == {"mtm", "vtm"} and set(ac) == {"mtm", "vtm"}
