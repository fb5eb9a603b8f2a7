"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (lavender_amd.dp.ArenaReducer) -- parameter
broadcast from rank 0, bucketed sum all-reduce of the flat gradient arena, and the folded 1/world division."""
import os
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavender_amd.dp import ArenaReducer
    n = 1000 * 64 + 64
    synced = []
    names = ["enc_txt.a", "trsfr.layer.0.w", "trsfr.layer.1.w", "enc_img.swin.layers.0.w", "enc_img.swin.layers.1.w", "fc_mtm.w", "emb_task"]
    offs = {"enc_txt.a": 0, "trsfr.layer.0.w": 6400, "trsfr.layer.1.w": 19200, "enc_img.swin.layers.0.w": 32000,
            "enc_img.swin.layers.1.w": 38400, "fc_mtm.w": 51200, "emb_task": 60800}
    ends = {"enc_txt.a": 6400, "trsfr.layer.0.w": 19200, "trsfr.layer.1.w": 32000, "enc_img.swin.layers.0.w": 38400,
            "enc_img.swin.layers.1.w": 51200, "fc_mtm.w": 60800, "emb_task": n}
    arena = types.SimpleNamespace(total=n, master=torch.full((n,), float(rank + 1)), grad=torch.zeros(n), names=names, listeners=[],
                                  span=lambda ns: (min(offs[x] for x in ns), max(ends[x] for x in ns)),
                                  sync_half=lambda: synced.append(1))
    model = types.SimpleNamespace(arena=lambda: arena)
    red = ArenaReducer(model, bucket_mb=0.05)                 # ~13k-element buckets -> 5 buckets
    ok = bool((arena.master == 1.0).all()) and len(synced) == 1        # broadcast from rank 0 + bf16 refresh
    bk = red.buckets()
    ok &= bk[0][1] == n and bk[-1][0] == 0 and all(bk[i][0] == bk[i + 1][1] for i in range(len(bk) - 1)) and len(bk) >= 4
    g = torch.Generator().manual_seed(rank)
    local = torch.randn(n, generator=g)
    arena.grad.copy_(local)
    red.finish()
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(r)) for r in range(world))
    ok &= bool(torch.allclose(arena.grad, expect, atol=1e-6))
    # second step with the early (overlapped) exchange of the fusion-side ranges: nothing reduced twice, nothing missed
    ok &= red.early_ranges == [(6400, 32000), (51200, 60800)]
    arena.grad.copy_(local)
    for f in arena.listeners:
        f("fusion_grads_final")
    mid = arena.grad.clone()
    ok &= bool(torch.allclose(mid[6400:32000], expect[6400:32000], atol=1e-6)) and bool(torch.equal(mid[:6400], local[:6400]))
    # the backward then leaves Swin stage 1 (its range becomes final), later stage 0 is left to finish()
    ok &= red.stage_ranges == {0: (32000, 38400), 1: (38400, 51200)}
    for f in arena.listeners:
        f("swin_stage1_grads_final")
        try:
            f("swin_stage1_grads_final")                      # a repeated event in one step must fail loudly (two backwards / two enc_img calls)
            ok = False
        except RuntimeError:
            pass
    mid = arena.grad.clone()
    ok &= bool(torch.allclose(mid[38400:51200], expect[38400:51200], atol=1e-6)) and bool(torch.equal(mid[32000:38400], local[32000:38400]))
    red.finish()
    ok &= bool(torch.allclose(arena.grad, expect, atol=1e-6))
    ok &= red.world == world
    # gradient accumulation: the non-final backward raises the same events, which must be ignored; finish() exchanges everything
    arena.grad.copy_(local)
    red.begin_step(last_micro_step=False)
    for f in arena.listeners:
        f("fusion_grads_final")
        f("fusion_grads_final")
    ok &= bool(torch.equal(arena.grad, local))
    red.finish()
    ok &= bool(torch.allclose(arena.grad, expect, atol=1e-6))
    # begin_step on an unfinished exchange is an error
    red.begin_step()
    for f in arena.listeners:
        f("fusion_grads_final")
    try:
        red.begin_step()
        ok = False
    except RuntimeError:
        pass
    red.finish()
    # half-precision exchange (grad_dtype="bf16"): buckets go through the bf16 staging arena, the sums come back widened; every
    # rank holds the SAME values (replica equality), equal to the bf16-rounded inputs summed in bf16
    arena.listeners.clear()
    red16 = ArenaReducer(model, bucket_mb=0.05, grad_dtype="bf16")
    red16.HALF_MIN_ELEMS = 1024
    arena.grad.copy_(local)
    red16.begin_step()
    for f in arena.listeners:
        f("fusion_grads_final")
    red16.finish()
    exp16 = sum(torch.randn(n, generator=torch.Generator().manual_seed(r)).bfloat16() for r in range(world)).float()
    ok &= bool(torch.equal(arena.grad, exp16)) and bool(torch.allclose(arena.grad, expect, atol=0.05, rtol=0.02))
    q.put((rank, ok, float(arena.grad.double().sum())))
    dist.destroy_process_group()


def test_arena_reducer_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    assert res[0][2] == res[1][2]                               # bf16 exchange: bit-identical gradients on both replicas


# ---- ZeRO-1 (args.deepspeed): reduce-scatter + sharded AdamW + all-gather of the bf16 working copy ---------------------------
class _MockArena:
    """The slice of lavender_amd.arena.ParamArena the reducers use, with the fused AdamW kernel restated on the CPU."""

    def __init__(self, total, rank):
        slack = 4096
        g = torch.Generator().manual_seed(1234)
        self.total = total
        self.master = (torch.randn(total, generator=g) * 0.1 + float(rank))       # rank-dependent: the broadcast must fix it
        self.half_full = torch.zeros(total + slack, dtype=torch.bfloat16)
        self.half = self.half_full[:total]
        self.grad_full = torch.zeros(total + slack)
        self.grad = self.grad_full[:total]
        self.names, self.listeners = [], []
        self.m = self.v = None
        self.transposed_syncs = 0

    def sync_half(self):
        self.half.copy_(self.master.bfloat16())

    def sync_transposed(self):
        self.transposed_syncs += 1

    def adamw_step(self, lr4, wd4, step, max_norm, grad_div, betas, eps, shard=None, sum_gradsq=None):
        from oracle import lavender_ref as R
        lo, hi = (0, self.total) if shard is None else shard
        if self.m is None:
            self.m, self.v = torch.zeros(hi - lo), torch.zeros(hi - lo)
        sq = (self.grad[lo:hi].double() ** 2).sum().float().reshape(1)
        if sum_gradsq is not None:
            sum_gradsq(sq)
        coef = 1.0 / grad_div
        c = max_norm / (sq.sqrt().item() / grad_div + 1e-6)
        if max_norm > 0 and c < 1:
            coef *= c
        p, self.m, self.v = R.adamw_step(self.master[lo:hi], self.grad[lo:hi] * coef, self.m, self.v, step, lr4[0], wd4[0])
        self.master[lo:hi] = p
        self.half[lo:hi] = p.bfloat16()


def _zero_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lavender_amd.dp import ArenaReducer, ZeroOneReducer
    total = 64 * 333                                            # not a multiple of world * 64: the last shard is short
    lr4, wd4 = [1e-2] * 4, [1e-3] * 4
    outs = {}
    for kind in ("ddp", "zero1", "zero1_bf16", "ddp_bf16"):
        arena = _MockArena(total, rank)
        # two named ranges that straddle the shard boundary: the early exchange (reduce to the owning rank per bucket) must cut there
        arena.names = ["trsfr.a", "enc_img.swin.layers.0.w"]
        spans = {"trsfr.a": (64 * 10, 64 * 200), "enc_img.swin.layers.0.w": (64 * 200, 64 * 300)}
        arena.span = lambda ns, sp=spans: (min(sp[x][0] for x in ns), max(sp[x][1] for x in ns))
        model = types.SimpleNamespace(arena=lambda a=arena: a)
        red = (ZeroOneReducer if kind.startswith("zero1") else ArenaReducer)(model, grad_dtype="bf16" if kind.endswith("bf16") else "fp32")
        red.HALF_MIN_ELEMS = 256
        for step in (1, 2):
            arena.grad_full.zero_()
            arena.grad.copy_(torch.randn(total, generator=torch.Generator().manual_seed(100 * step + rank)))
            red.begin_step()
            for f in arena.listeners:
                f("fusion_grads_final")                         # early exchange of the fusion-side range in both modes
                f("swin_stage0_grads_final")
            red.finish()
            red.optimizer_step(arena, lr4, wd4, step, 1.0, (0.9, 0.98), 1e-8)
        red.gather_master()
        outs[kind] = (arena.master.clone(), arena.half.clone(), arena.m.numel(), arena.transposed_syncs)
    (mb0, hb0, _, _), (mb1, hb1, _, _) = outs["ddp_bf16"], outs["zero1_bf16"]
    ok16 = bool(torch.allclose(mb0, mb1, atol=1e-6)) and bool(torch.equal(hb0, hb1))   # bf16 exchange: ZeRO-1 == replicated as well
    ok16 &= bool(torch.allclose(mb0, outs["ddp"][0], atol=2e-2))                        # and close to the fp32 exchange (Adam steps are +-lr)
    (m0, h0, n0, _), (m1, h1, n1, ts) = outs["ddp"], outs["zero1"]
    shard = ((total + world - 1) // world + 63) // 64 * 64
    ok = bool(torch.allclose(m0, m1, atol=1e-6)) and bool(torch.equal(h0, h1))      # ZeRO-1 == replicated AdamW
    ok &= n0 == total and n1 == min(shard, total - min(rank * shard, total)) and ts == 2   # optimizer state for the own shard only
    q.put((rank, ok and ok16, float(m1.double().sum())))
    dist.destroy_process_group()


def test_zero1_matches_replicated_adamw_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    assert res[0][2] == res[1][2]                               # identical parameters on both ranks after gather_master()


def _subgroup_worker(rank, world, port, q):
    """ZeRO-1 inside a process SUB-GROUP {1, 2} of a 3-rank world: shard owners are group-local indices, dist.reduce / broadcast
    take global ranks -- every bucket must still reach the rank that owns it (lavender_amd/dp.py ZeroOneReducer._exchange)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([1, 2])                                # collective over the whole world, members or not
    if rank == 0:
        q.put((rank, True, 0.0))
        dist.barrier()
        dist.destroy_process_group()
        return
    from lavender_amd.dp import ArenaReducer, ZeroOneReducer
    total = 64 * 101
    lr4, wd4 = [1e-2] * 4, [1e-3] * 4
    outs = {}
    for kind in ("ddp", "zero1"):
        arena = _MockArena(total, rank)
        arena.names = ["trsfr.a"]
        arena.span = lambda ns: (64 * 5, 64 * 90)
        model = types.SimpleNamespace(arena=lambda a=arena: a)
        red = (ZeroOneReducer if kind == "zero1" else ArenaReducer)(model, group=grp)
        for step in (1, 2):
            arena.grad_full.zero_()
            arena.grad.copy_(torch.randn(total, generator=torch.Generator().manual_seed(100 * step + rank)))
            red.begin_step()
            for f in arena.listeners:
                f("fusion_grads_final")
            red.finish()
            red.optimizer_step(arena, lr4, wd4, step, 1.0, (0.9, 0.98), 1e-8)
        red.gather_master()
        outs[kind] = (arena.master.clone(), arena.half.clone())
    ok = bool(torch.allclose(outs["ddp"][0], outs["zero1"][0], atol=1e-6)) and bool(torch.equal(outs["ddp"][1], outs["zero1"][1]))
    # the broadcast came from the group's first member (global rank 1), whose masters start at randn * 0.1 + 1.0
    ok &= abs(float(outs["ddp"][0].mean()) - 1.0) < 0.2
    q.put((rank, ok, float(outs["zero1"][0].double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_zero1_in_a_process_subgroup_world3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True), (2, True)]
    assert res[1][2] == res[2][2]
