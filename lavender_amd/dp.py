"""Data-parallel gradient exchange over the flat gradient arena (replaces DDP / DeepSpeed ZeRO-1 of the
reference's agent.py:252-265 and utils/deepspeed.py).

One process per GPU; parameters and optimizer state are replicated; the ONLY data-path collective per step is a
sum all-reduce of the fp32 gradient arena (RCCL over xGMI when the process group backend is "nccl"; "gloo" on CPU
for tests).  Because gradients already live in one contiguous buffer in execution order, buckets are plain slices:
no flatten / unflatten copies and no per-parameter hooks.  Buckets are issued back-to-front (the order the backward
produces them) on a side stream so the tail of the exchange overlaps the optimizer's norm pass of earlier buckets.
The division by world size is folded into the fused AdamW kernel (grad_div).
"""
import torch
import torch.distributed as dist


class ArenaReducer:
    def __init__(self, model, bucket_mb=64, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        a = model.arena()
        # identical initial parameters on every rank (DDP broadcasts from rank 0 at wrap time)
        dist.broadcast(a.master, src=0, group=group)
        a.sync_half()
        self._stream = torch.cuda.Stream() if a.master.is_cuda else None

    def buckets(self):
        n = self.model.arena().total
        edges = list(range(0, n, self.bucket_elems)) + [n]
        return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)][::-1]

    def finish(self):
        """All-reduce (sum) every bucket of the gradient arena; returns when the reduced gradients are usable on
        the current stream."""
        g = self.model.arena().grad
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                works = [dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for lo, hi in self.buckets()]
                for w in works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            for lo, hi in self.buckets():
                dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
