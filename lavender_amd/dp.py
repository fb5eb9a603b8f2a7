"""Data-parallel gradient exchange over the flat gradient arena (replaces DDP / DeepSpeed ZeRO-1 of the
reference's agent.py:252-265 and utils/deepspeed.py).

One process per GPU; the ONLY data-path collective per step is the gradient exchange below (RCCL over xGMI when the
process group backend is "nccl"; "gloo" on CPU for tests).  Because gradients already live in one contiguous buffer in
execution order, buckets are plain slices: no flatten / unflatten copies and no per-parameter hooks.

Two modes (Agent_Base.prepare_dist_model picks by args.deepspeed, as the reference does):

* ArenaReducer (DDP semantics, agent.py:261-265): replicated parameters and optimizer state; sum all-reduce of the fp32
  gradient arena, overlapped with the backward.  The arena is laid out [text embeddings | fusion encoder | video encoder |
  MLM head | ...]; the backward finishes the MLM head and all fusion layers BEFORE it enters the video encoder, so when the
  first video-side stage starts its backward (engine.VideoEmbedFn) the reducer is notified and all-reduces those finished
  ranges (about half of the 886 MB) on a side stream while the Swin backward -- ~40 % of the backward time -- still runs;
  each Swin stage's range follows when the backward leaves that stage.  finish() reduces what is left (stage 0, the patch /
  video embeddings: ~1 % of the arena).  The division by world size is folded into the fused AdamW kernel (grad_div).
  The comm stream waits for the producers of a range through events on the main stream AND on the weight-gradient side
  stream (engine.dw_stream); the main stream itself is never stalled by an early exchange.

* ZeroOneReducer (DeepSpeed ZeRO-1, utils/deepspeed.py:40-44, agent.py:254-259): the arena is cut into `world` equal
  64-aligned shards; gradients are reduce-scattered (each rank receives the sum of its shard only), the fused AdamW runs
  on the local shard with optimizer state m, v allocated for that shard only (2 x 886 MB / world), and the refreshed bf16
  working copy is all-gathered.  The fp32 masters of the other shards go stale and are re-assembled on demand
  (gather_master(): before a checkpoint).

Contract (both): exactly ONE backward per optimizer step may raise the overlap events; gradient accumulation goes through
begin_step(last_micro_step=False) (events ignored, everything exchanged by finish()).  An event that arrives twice in one
armed step -- the video encoder ran twice, or loss.backward() was called twice -- raises instead of silently summing a
range that later receives more local gradient.
"""
import torch
import torch.distributed as dist


def _backend(group):
    try:
        return dist.get_backend(group)
    except Exception:
        return "gloo"


class ArenaReducer:
    zero_stage = 0

    def __init__(self, model, bucket_mb=64, group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        a = model.arena()
        # identical initial parameters on every rank (DDP broadcasts from rank 0 at wrap time)
        dist.broadcast(a.master, src=0, group=group)
        a.sync_half()
        self._stream = torch.cuda.Stream() if a.master.is_cuda else None
        self._done = []          # [lo, hi) ranges already reduced in this step
        self._works = []
        self._fired = set()
        self._armed = True
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

    # ---- step protocol ---------------------------------------------------------------------------------------------
    def begin_step(self, last_micro_step=True):
        """Called by Agent_Base.backward_step before loss.backward().  last_micro_step=False (gradient accumulation): this
        backward's gradients are not final -- the overlap events are ignored."""
        if self._done or self._works:
            raise RuntimeError("ArenaReducer.begin_step: the previous step's exchange was not finished (finish() not called)")
        self._armed = bool(last_micro_step)
        self._fired = set()

    def _comm_waits_for_producers(self):
        """comm stream <- events on the main stream and on the weight-gradient side stream; nothing waits on the comm stream"""
        self._stream.wait_stream(torch.cuda.current_stream())
        from .engine import _dw_streams
        for dev, st in _dw_streams.items():
            if dev == self.model.arena().grad.device:
                self._stream.wait_stream(st)

    def _reduce(self, ranges):
        g = self.model.arena().grad
        if self._stream is not None:
            self._comm_waits_for_producers()
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
        if not self._armed:
            return
        if name in self._fired:
            raise RuntimeError(f"ArenaReducer: event {name!r} arrived twice in one step -- the video encoder ran more than once or "
                               "loss.backward() was called more than once before backward_step(); use begin_step(last_micro_step=False) "
                               "for the non-final backward passes of a gradient-accumulation step")
        self._fired.add(name)
        if name == "fusion_grads_final":
            if self.early_ranges:
                self._reduce(self.early_ranges)
        elif name.startswith("swin_stage") and name.endswith("_grads_final"):
            rng = self.stage_ranges.get(int(name[len("swin_stage"):-len("_grads_final")]))
            if rng is not None:
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
        self._works, self._done, self._fired, self._armed = [], [], set(), True

    # ---- optimizer hook (identical replicas: plain step) -----------------------------------------------------------------
    def optimizer_step(self, arena, lr4, wd4, step, max_norm, betas, eps):
        arena.adamw_step(lr4, wd4, step, max_norm, float(self.world), betas, eps)

    def gather_master(self):
        """Replicated masters: nothing to do."""


class ZeroOneReducer(ArenaReducer):
    """DeepSpeed ZeRO stage 1 over the arena (see the module docstring)."""
    zero_stage = 1

    def __init__(self, model, group=None):
        super().__init__(model, bucket_mb=64, group=group)
        a = model.arena()
        align = 64
        self.shard = ((a.total + self.world - 1) // self.world + align - 1) // align * align
        self.lo = min(self.rank * self.shard, a.total)
        self.hi = min(self.lo + self.shard, a.total)
        self.padded = self.shard * self.world
        for name in ("grad_full", "half_full"):
            buf = getattr(a, name, None)
            if buf is not None and buf.numel() < self.padded:
                raise RuntimeError(f"ZeroOneReducer: arena.{name} has no room for {self.world} shards of {self.shard} elements")
        self.early_ranges, self.stage_ranges = [], {}          # one reduce-scatter at finish(): no early exchange in this mode
        self._nccl = _backend(group) == "nccl"
        self._master_stale = False

    def _on_event(self, name):
        return

    def finish(self):
        a = self.model.arena()
        g = a.grad_full[:self.padded] if hasattr(a, "grad_full") else a.grad
        from .engine import dw_join
        if g.is_cuda:
            dw_join()
        mine = g[self.rank * self.shard:(self.rank + 1) * self.shard]
        if self._nccl and g.numel() == self.padded:
            dist.reduce_scatter_tensor(mine, g, op=dist.ReduceOp.SUM, group=self.group)       # in place: output = own slice of the input
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)                        # gloo (tests): reduce everything, use the own shard
        self._works, self._done, self._fired, self._armed = [], [], set(), True

    def optimizer_step(self, arena, lr4, wd4, step, max_norm, betas, eps):
        def sum_sq(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        arena.adamw_step(lr4, wd4, step, max_norm, float(self.world), betas, eps, shard=(self.lo, self.hi), sum_gradsq=sum_sq)
        # all-gather the refreshed bf16 working copy; the transposed copy is rebuilt locally from it
        h = arena.half_full[:self.padded]
        mine = h[self.rank * self.shard:(self.rank + 1) * self.shard]
        if self._nccl:
            dist.all_gather_into_tensor(h, mine, group=self.group)
        else:
            keep = mine.clone()
            h.zero_()
            mine.copy_(keep)
            hf = h.view(torch.int32)                           # integer sum of disjoint shards == concatenation, exact for any bit pattern
            dist.all_reduce(hf, op=dist.ReduceOp.SUM, group=self.group)
        arena.sync_transposed()
        self._master_stale = True
        arena.masters_sharded = True

    def gather_master(self):
        """fp32 masters of every shard on every rank (before state_dict() / a checkpoint)."""
        if not self._master_stale:
            return
        a = self.model.arena()
        full = torch.zeros(self.padded, dtype=torch.float32, device=a.master.device)
        full[self.lo:self.hi].copy_(a.master[self.lo:self.hi])
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        a.master.copy_(full[:a.total])
        self._master_stale = False
        a.masters_sharded = False
