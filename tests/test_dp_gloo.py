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
        f("swin_stage1_grads_final")                          # a repeated event must not reduce twice
    mid = arena.grad.clone()
    ok &= bool(torch.allclose(mid[38400:51200], expect[38400:51200], atol=1e-6)) and bool(torch.equal(mid[32000:38400], local[32000:38400]))
    red.finish()
    ok &= bool(torch.allclose(arena.grad, expect, atol=1e-6))
    ok &= red.world == world
    q.put((rank, ok))
    dist.destroy_process_group()


def test_arena_reducer_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
