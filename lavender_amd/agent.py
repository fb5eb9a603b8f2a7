"""Training agent -- MI355X-native mirror of the reference's agent.py (Agent_Base, WarmupLinearLR).

Differences in mechanism, not in contract:
  * optimizer: the four AdamW groups of agent.py:96-140 (swin|other x decay|no-decay, betas (0.9, 0.98)) are
    applied by ONE fused kernel over the flat arena (global-norm clip + AdamW + bf16 refresh), see arena.py;
  * mixed precision: bf16 activations with fp32 master weights need no GradScaler (agent.py:76,240-248);
  * data parallelism: gradient buckets are contiguous arena slices all-reduced with RCCL on a side stream while
    the backward still runs (replaces DDP / DeepSpeed ZeRO-1 of agent.py:252-265), see dp.py.
"""
import json
import os

import torch

from . import engine as E
from .arena import param_group_of
from .dist import get_world_size, is_main_process


class WarmupLinearLR:
    """agent.py:13-43: linear warm-up over warmup_ratio*max_iter steps, then linear decay; floor min_lr.
    Stand-alone (no torch optimizer needed): get_lr(base_lrs) for the current step, step() advances."""

    def __init__(self, optimizer, max_iter, min_lr=1e-8, warmup_ratio=0.1, last_epoch=-1):
        self.optimizer = optimizer
        self.max_iter, self.min_lr, self.warmup_ratio = max_iter, min_lr, warmup_ratio
        self.warmup_iters = int(warmup_ratio * max_iter)
        self.base_lrs = [g["initial_lr"] if "initial_lr" in g else g["lr"] for g in optimizer.param_groups]
        for g, b in zip(optimizer.param_groups, self.base_lrs):
            g.setdefault("initial_lr", b)
        self.last_epoch = last_epoch
        self.step()

    def get_lr_factor(self):
        tot, warm, step = self.max_iter, self.warmup_iters, self.last_epoch
        if step < warm:
            return max(0, step / warm)
        elif step > tot:
            step = tot
        return max(0, (tot - step) / (tot - warm))

    def get_lr(self):
        f = self.get_lr_factor()
        return [max(self.min_lr, b * f) for b in self.base_lrs]

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]


class ArenaAdamW:
    """The optimizer object of Agent_Base.build_optimizer (agent.py:96-140): same four param_groups (names,
    lr, weight_decay), stepping = one fused kernel over the arena."""

    def __init__(self, model, groups, betas=(0.9, 0.98), eps=1e-8):
        self.model = model
        self.param_groups = groups
        self.betas, self.eps = betas, eps
        self.steps = 0

    def step(self, max_norm=-1.0, grad_div=1.0, dp=None):
        self.steps += 1
        a = self.model.arena()
        lr4, wd4 = [g["lr"] for g in self.param_groups], [g["weight_decay"] for g in self.param_groups]
        if dp is not None:
            dp.optimizer_step(a, lr4, wd4, self.steps, max_norm, self.betas, self.eps)      # DDP: plain step; ZeRO-1: sharded step + all-gather
        else:
            a.adamw_step(lr4, wd4, self.steps, max_norm, grad_div, self.betas, self.eps)

    def zero_grad(self, set_to_none=False):
        self.model.arena().zero_grad()


class CrossEntropyIgnore:
    """T.nn.CrossEntropyLoss(ignore_index=-1) of agent.py:72 on the HIP engine.  `count` (number of labelled
    rows) may be given when the host already knows it (labels are built on the host)."""

    def __init__(self, keep_logits=False):
        self.keep_logits = bool(keep_logits)              # args.keep_logits: leave the caller's logits intact in training (one extra copy)

    def __call__(self, logits, labels, count=None):
        return E.CrossEntropyFn.apply(logits, labels, count, self.keep_logits)

    def cuda(self):
        return self


class Agent_Base:
    def __init__(self, args, model):
        super().__init__()
        self.args, self.model = args, model
        self.loss_func = CrossEntropyIgnore(keep_logits=getattr(args, "keep_logits", False)).cuda()
        self.optzr = self.build_optimizer()
        self.lr_scheduler = WarmupLinearLR(self.optzr, args.max_iter)
        self.scaler = None                      # bf16 + fp32 masters: no loss scaling
        self.log = None
        self.dp = None
        self.tokzr = getattr(model, "tokzr", None)
        for k in ("cls", "sep", "pad", "mask", "unk", "true", "false"):
            if hasattr(model, f"{k}_token_id"):
                setattr(self, f"{k}_token_id", getattr(model, f"{k}_token_id"))
        self.global_step = 0

    def _unwrapped(self):
        return self.model.module if hasattr(self.model, 'module') else self.model

    def build_optimizer(self):
        """agent.py:96-140: group membership by substring on the unwrapped parameter name."""
        named = list(self._unwrapped().named_parameters())
        wd, lr, mul = self.args.decay, self.args.lr, self.args.vis_backbone_lr_mul
        groups = [dict(params=[], names=[], weight_decay=wd, lr=lr * mul), dict(params=[], names=[], weight_decay=wd, lr=lr),
                  dict(params=[], names=[], weight_decay=0.0, lr=lr * mul), dict(params=[], names=[], weight_decay=0.0, lr=lr)]
        for n, p in named:
            g = groups[param_group_of(n)]
            g["params"].append(p)
            g["names"].append(n)
        return ArenaAdamW(self._unwrapped(), groups)

    def _set_mode(self, is_train):
        """self.model.train() / .eval() as the reference's step() does on every call (main_pretrain_mlm.py:146) -- but nn.Module.train()
        walks and re-assigns all ~425 modules each time (1.6 ms of launch-thread time per step); the flags are only rewritten when some
        module actually is in the other mode (the module list is cached: the model's structure does not change under an agent)."""
        mods = self.__dict__.get("_mode_modules")
        if mods is None:
            mods = self.__dict__["_mode_modules"] = list(self.model.modules())
        is_train = bool(is_train)
        for m in mods:
            if m.training != is_train:
                self.model.train(is_train)
                break

    def reduce_mean(self, v):
        """agent.py:145-153."""
        world_size = get_world_size()
        if world_size < 2:
            return v
        import torch.distributed as DIST
        t = torch.tensor(v, dtype=torch.float32, device="cuda" if torch.cuda.is_available() else "cpu")
        DIST.all_reduce(t)
        return t.item() / world_size

    def save_training_meta(self):
        if is_main_process():
            os.makedirs(self.args.path_output, exist_ok=True)
            json.dump(dict(self.args), open(f'{self.args.path_output}/args.json', 'w'), indent=2)
            self.save_model(0)

    def _save_state(self, filename, write_log=True):
        """rank-0 torch.save of the CPU state_dict as `filename` under args.path_output.  EVERY RANK MUST CALL THIS (and therefore
        save_model): with ZeRO-1 (args.deepspeed) gather_master() below is a collective that re-assembles the fp32 masters, so a
        caller that guards save_model with is_main_process() would hang the other ranks; only the file write is rank-0-only."""
        if self.dp is not None:
            self.dp.gather_master()
        if is_main_process():
            output_dir = self.args.path_output
            os.makedirs(output_dir, exist_ok=True)
            sd = {k: v.cpu() if isinstance(v, torch.Tensor) else v for k, v in self._unwrapped().state_dict().items()}
            torch.save(sd, f"{output_dir}/{filename}")
            if write_log and self.log is not None:
                json.dump(self.log, open(f"{output_dir}/log.json", 'w'), indent=2)

    def save_model(self, ep):
        """agent.py:164-180: ckpt_violet_{task}_{ep}.pt (+ log.json).  Call on every rank (see _save_state)."""
        self._save_state(f"ckpt_violet_{self.args.task}_{ep}.pt")

    def log_memory(self, ep=-1, step=-1):
        step_str = f"global step: {self.global_step}," if ep == -1 and step == -1 else f"ep: {ep}, step: {step},"
        mem = torch.cuda.max_memory_allocated() / 2 ** 30 if torch.cuda.is_available() else 0.0
        return (f"{step_str} lr_swin: {self.optzr.param_groups[0]['lr']:.2e}, lr_bert: {self.optzr.param_groups[1]['lr']:.2e}, "
                f"max memory: {mem:.2f} GB")

    def prepare_batch(self, batch):
        """agent.py:197-201 (move_to_cuda, dataset.py:333): non-blocking H2D of every tensor in the batch."""
        out = {}
        for k, v in batch.items():
            out[k] = v.cuda(non_blocking=True) if isinstance(v, torch.Tensor) else v
        return out

    def forward_step(self, batch):
        """agent.py:203-233."""
        if isinstance(batch, dict):
            return self.model(batch)
        elif isinstance(batch, tuple):
            return self.model(*batch)
        raise TypeError(f"batch is either dict or tuple, {type(batch)}")

    def backward_step(self, loss):
        """agent.py:235-250: backward, (gradient exchange,) clip by global norm, AdamW, LR schedule, zero_grad."""
        if self.dp is not None:
            self.dp.begin_step()
        loss.backward()
        from .engine import dw_join
        dw_join()                                          # weight-gradient kernels run on a side stream
        if self.dp is not None:
            self.dp.finish()
        self.optzr.step(max_norm=self.args.max_grad_norm, dp=self.dp)
        self.lr_scheduler.step()
        self.optzr.zero_grad()
        self.global_step += 1

    def prepare_dist_model(self):
        """agent.py:252-265: instead of wrapping in DDP (default) or DeepSpeed ZeRO-1 (args.deepspeed), attach the matching
        arena gradient reducer (lavender_amd.dp)."""
        if get_world_size() > 1:
            from .dp import ArenaReducer, ZeroOneReducer
            # args.grad_comm_bf16: half-precision gradient exchange (the reference's ZeRO-1 config uses fp16 gradients,
            # utils/deepspeed.py:20-28); default fp32 = DDP's exact sum
            gd = "bf16" if getattr(self.args, "grad_comm_bf16", False) else None
            if getattr(self.args, "deepspeed", False):
                self.dp = ZeroOneReducer(self._unwrapped(), grad_dtype=gd)
            else:
                self.dp = ArenaReducer(self._unwrapped(), grad_dtype=gd)
