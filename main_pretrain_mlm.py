"""Entry point with the reference's command line (main_pretrain_mlm.py:235-328):

    python -m torch.distributed.run --nproc_per_node=N main_pretrain_mlm.py --config _args/args_pretrain_webvid.json --path_output D [--path_ckpt P]

The model / agent / optimizer / data-parallel path are the MI355X-native ones of lavender_amd.  When `--data_dir` holds the
reference's files (`txt_<dataset>.json`, `<dataset>_train_<part>.tsv/.lineidx`, `<dataset>_val.tsv/.lineidx`) and `--tokenizer`
is a local tokenizer directory, the batches come from the GPU input pipeline (lavender_amd/data.py: TSV -> JPEG decode ->
transforms in HBM), loop structure as main_pretrain_mlm.py:251-328; otherwise (no data here, no network for the tokenizer) the
script trains on synthetic clips of the configured shape, which is what the benchmark uses.
"""
import json
import os

import numpy as np
import torch

import lavender_amd as LA
from lavender_amd.args import get_args
from lavender_amd.dist import get_rank, get_world_size, is_main_process


class _Tok:
    cls_token = "[CLS]"; sep_token = "[SEP]"; pad_token = "[PAD]"; mask_token = "[MASK]"; unk_token = "[UNK]"
    ids = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103, "true": 2995, "false": 6270}

    def convert_tokens_to_ids(self, toks):
        return [self.ids[t] for t in toks]


class SyntheticPretrain(torch.utils.data.Dataset):
    """(img, txt, mask) triples with the shapes Dataset_Pretrain_MLM emits (main_pretrain_mlm.py:14-25)."""

    def __init__(self, args, n):
        self.a, self.n = args, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        X = self.a.size_txt + 1
        img = torch.randn(self.a.size_frame, 3, self.a.size_img, self.a.size_img, generator=g)
        k = int(torch.randint(6, X - 4, (1,), generator=g))
        txt = torch.zeros(X, dtype=torch.long)
        txt[0] = 101; txt[1:1 + k] = torch.randint(1000, 30000, (k,), generator=g); txt[1 + k] = 102; txt[-1] = 103
        return {"img": img, "txt": txt, "mask": (txt != 0).long()}


def real_data(args):
    """{dataset: txt json} and a tokenizer when the reference's data layout is present, else None."""
    dirs = {d: (args.dataset[d] if isinstance(args.dataset, dict) else getattr(args, "data_dir", None)) for d in args.dataset}   # :263-266
    if not all(v and os.path.exists(f"{v}/txt_{d}.json") for d, v in dirs.items()) or not os.path.isdir(str(args.tokenizer)):
        return None
    import transformers
    return ({d: json.load(open(f"{v}/txt_{d}.json")) for d, v in dirs.items()},
            transformers.AutoTokenizer.from_pretrained(args.tokenizer))


def train_on_tsv(args, txt_data, tokzr):
    """main_pretrain_mlm.py:251-328 with the GPU input pipeline."""
    from lavender_amd.data import Dataset_Pretrain_MLM, get_dl
    rank, world = get_rank(), get_world_size()
    loaders, n = {}, 0
    data_dirs = {}
    for d in args.dataset:
        size_part = args.size_part if isinstance(args.size_part, int) else args.size_part[d]
        data_dirs[d] = args.dataset[d] if isinstance(args.dataset, dict) else args.data_dir     # main_pretrain_mlm.py:263-266
        loaders[f"{d}-val"] = get_dl(Dataset_Pretrain_MLM(args, txt_data[d], d, "val", data_dir=data_dirs[d], tokzr=tokzr), args, rank, world)
        loaders[f"{d}-train-0"] = get_dl(Dataset_Pretrain_MLM(args, txt_data[d], d, "train", 0, data_dir=data_dirs[d], tokzr=tokzr), args, rank, world)
        n += len(loaders[f"{d}-train-0"]) * size_part
    args.max_iter = n * args.size_epoch
    model = LA.LAVENDER_Pretrain_MLM(args, tokzr)
    model.load_ckpt(args.path_ckpt)
    model.cuda()
    agent = LA.Agent_Pretrain_MLM(args, model)
    if args.distributed:
        agent.prepare_dist_model()
    agent.save_training_meta()
    for e in range(args.size_epoch):
        for d in args.dataset:
            size_part = args.size_part if isinstance(args.size_part, int) else args.size_part[d]
            for part in range(size_part):
                key = f"{d}-train-{part}"
                dl_tr = loaders.get(key) or get_dl(Dataset_Pretrain_MLM(args, txt_data[d], d, "train", part, data_dir=data_dirs[d], tokzr=tokzr),
                                                   args, rank, world)
                dl_tr.sampler.set_epoch(e + 1)
                ls_tr = agent.go_dl(e + 1, dl_tr, True)
                ac_vl = agent.go_dl(e + 1, loaders[f"{d}-val"], False)
                for k in ls_tr:                                # main_pretrain_mlm.py:321-323
                    agent.log[d]['ls_%s' % k].append(ls_tr[k])
                    agent.log[d]['ac_%s' % k].append(ac_vl[k])
                agent.save_model(e + 1, d, part)               # ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt (:326)
                if is_main_process():
                    print(f"Ep {e + 1}, dataset {d}, part {part}: {json.dumps(ls_tr)}, {json.dumps(ac_vl)}")


def name_the_run(args):
    """main_pretrain_mlm.py:239-244 / main_pretrain_task_specific.py: the task carries the dataset names and every run writes into its own
    `<path_output>/_<task>_<YYYYmmddHHMMSS>` directory (args.json, the checkpoints)."""
    from datetime import datetime
    for d in args.dataset:
        args.task += f"-{d}"
    args.path_output = '%s/_%s_%s' % (args.path_output, args.task, datetime.now().strftime('%Y%m%d%H%M%S'))


if __name__ == '__main__':
    args = get_args()
    name_the_run(args)
    real = real_data(args)
    if real is not None:
        train_on_tsv(args, *real)
        raise SystemExit(0)
    tokzr = _Tok()
    n_steps = int(os.environ.get("LAV_SYNTH_STEPS", 20))
    ds = SyntheticPretrain(args, n_steps * args.size_batch * get_world_size())
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=get_world_size(), rank=get_rank(), shuffle=True) \
        if args.distributed else None
    dl = torch.utils.data.DataLoader(ds, batch_size=args.size_batch, sampler=sampler, num_workers=args.n_workers, pin_memory=True,
                                     drop_last=True)
    args.max_iter = len(dl) * args.size_epoch
    model = LA.LAVENDER_Pretrain_MLM(args, tokzr)
    model.load_ckpt(args.path_ckpt)
    model.cuda()
    agent = LA.Agent_Pretrain_MLM(args, model)
    if args.distributed:
        agent.prepare_dist_model()
    agent.save_training_meta()
    for e in range(args.size_epoch):
        if sampler is not None:
            sampler.set_epoch(e)
        ls = agent.go_dl(e + 1, dl, True)
        if is_main_process():
            print(f"Ep {e + 1}: " + ", ".join(f"{k} {v:.4f}" for k, v in ls.items()))
        agent.save_model(e + 1)
