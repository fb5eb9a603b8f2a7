"""Entry point with the reference's command line (main_pretrain_task_specific.py:265-395):

    python -m torch.distributed.run --nproc_per_node=N main_pretrain_task_specific.py --config _args/args_pretrain_webvid.json --path_output D

Same MI355X-native kernels as main_pretrain_mlm.py; the video-text matching objective is the scalar score head
(`LAVENDER_Pretrain`, `Agent_Pretrain`).  The reference's TSV datasets are outside the hot path:
this script trains on synthetic clips of the configured shape.
"""
import os

import torch

import lavender_amd as LA
from lavender_amd.args import get_args
from lavender_amd.dist import get_rank, get_world_size, is_main_process
from main_pretrain_mlm import SyntheticPretrain, _Tok


class SyntheticPretrainTS(SyntheticPretrain):
    """Dataset_Pretrain (main_pretrain_task_specific.py:27-121) emits size_txt text positions (no appended [MASK])."""

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        X = self.a.size_txt
        img = torch.randn(self.a.size_frame, 3, self.a.size_img, self.a.size_img, generator=g)
        k = int(torch.randint(6, X - 4, (1,), generator=g))
        txt = torch.zeros(X, dtype=torch.long)
        txt[0] = 101; txt[1:1 + k] = torch.randint(1000, 30000, (k,), generator=g); txt[1 + k] = 102
        return {"img": img, "txt": txt, "mask": (txt != 0).long()}


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
    tokzr = _Tok()
    n_steps = int(os.environ.get("LAV_SYNTH_STEPS", 20))
    ds = SyntheticPretrainTS(args, n_steps * args.size_batch * get_world_size())
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=get_world_size(), rank=get_rank(), shuffle=True) \
        if args.distributed else None
    dl = torch.utils.data.DataLoader(ds, batch_size=args.size_batch, sampler=sampler, num_workers=args.n_workers, pin_memory=True,
                                     drop_last=True)
    args.max_iter = len(dl) * args.size_epoch
    model = LA.LAVENDER_Pretrain(args, tokzr)
    model.load_ckpt(args.path_ckpt)
    model.cuda()
    agent = LA.Agent_Pretrain(args, model)
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
