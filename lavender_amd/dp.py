"""Data-parallel gradient exchange over the flat gradient arena (replaces DDP / DeepSpeed ZeRO-1 of the
reference's agent.py:252-265 and utils/deepspeed.py).

One process per GPU; parameters and optimizer state are replicated; the ONLY data-path collective per step is a
sum all-reduce of the fp32 gradient arena (RCCL over xGMI when the process group backend is "nccl"; "gloo" on CPU
for tests).  Because gradients already live in one contiguous buffer in execution order, buckets are plain slices:
no flatten / unflatten copies and no per-parameter hooks.

Overlap with the backward: the arena is laid out [text embeddings | fusion encoder | video encoder | MLM head | ...].
The backward finishes the MLM head and all fusion layers (both the MTM and the VTM pass) BEFORE it enters the video
encoder, so when the first video-side stage starts its backward (engine.VideoEmbedFn) the reducer is notified and
all-reduces those finished ranges (about half of the 886 MB) on a side stream while the Swin backward -- ~40 % of
the backward time -- is still running; each Swin stage's range follows when the backward leaves that stage (stage 2
holds 57 M of the 88 M video-side parameters and finishes with a quarter of the Swin backward still to run).
finish() reduces what is left (stage 0, the patch / video embeddings: ~1 % of the arena).  The division by world size is folded into
the fused AdamW kernel (grad_div).
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
        self._done = []          # [lo, hi) ranges already reduced in this step
        self._works = []
        self.early_ranges = self._fusion_ranges(a)
        self.stage_ranges = self._stage_ranges(a)
        if hasattr(a, "listeners"):
            a.listeners.append(self._on_event)

    @staticmethod
    def _fusion_ranges(a):
        """Arena ranges whose gradients are final once the backward leaves the fusion side: trsfr.*, fc.* and fc_mtm.*"""
        if not hasattr(a, "span") or not hasattr(a, "names"):
            return []
        out = []
        for pre in ("trsfr.", "fc.", "fc_mtm."):
            names = [n for n in a.names if n.startswith(pre)]
            if names:
                out.append(a.span(names))
        return out

    @staticmethod
    def _stage_ranges(a):
        """{s: arena range of enc_img.swin.layers.s.*}: final when the first block of stage s has run its backward (the
        stage's PatchMerging, which follows the blocks in the forward, is differentiated before them)."""
        if not hasattr(a, "span") or not hasattr(a, "names"):
            return {}
        out = {}
        for s in range(8):
            names = [n for n in a.names if n.startswith(f"enc_img.swin.layers.{s}.")]
            if names:
                out[s] = a.span(names)
        return out

    def buckets(self, lo=0, hi=None):
        n = self.model.arena().total if hi is None else hi
        edges = list(range(lo, n, self.bucket_elems)) + [n]
        return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)][::-1]

    def _reduce(self, ranges):
        g = self.model.arena().grad
        if g.is_cuda:
            from .engine import dw_join
            dw_join()                                         # the ranges' weight gradients come from the dW side stream
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for lo, hi in ranges:
                    for b0, b1 in self.buckets(lo, hi):
                        self._works.append(dist.all_reduce(g[b0:b1], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for lo, hi in ranges:
                for b0, b1 in self.buckets(lo, hi):
                    dist.all_reduce(g[b0:b1], op=dist.ReduceOp.SUM, group=self.group)
        self._done.extend(ranges)

    def _on_event(self, name):
        if name == "fusion_grads_final" and self.early_ranges and not any(r in self._done for r in self.early_ranges):
            self._reduce(self.early_ranges)
        elif name.startswith("swin_stage") and name.endswith("_grads_final"):
            rng = self.stage_ranges.get(int(name[len("swin_stage"):-len("_grads_final")]))
            if rng is not None and rng not in self._done:
                self._reduce([rng])

    def finish(self):
        """Reduce every range not yet reduced in this step; returns when all reduced gradients are usable on the
        current stream."""
        total = self.model.arena().total
        rest, pos = [], 0
        for lo, hi in sorted(self._done):
            if lo > pos:
                rest.append((pos, lo))
            pos = max(pos, hi)
        if pos < total:
            rest.append((pos, total))
        self._reduce(rest)
        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                for w in self._works:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self._stream)
        self._works, self._done = [], []
