"""GPU: two data-parallel ranks (processes) sharing the one visible MI355X, process group over gloo (RCCL needs one
device per rank; the code path through lavender_amd.dp.ArenaReducer is the same): after backward + finish() every
rank holds the SUM of the per-rank gradients, parameters were broadcast from rank 0, and one optimizer step with
grad_div = world keeps the replicas bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_grads(rank, agent, m):
    from tests.helpers import BERT_CFGS, make_batch
    b = make_batch(2, vocab=BERT_CFGS["micro"]["vocab"], seed=10 + rank)
    torch.manual_seed(5 + rank)
    b.update(agent.masking(b["txt"], b["mask"]))
    batch = agent.prepare_batch(b)
    m.eval()                                            # deterministic (no dropout) so ranks can be replayed
    m.arena().zero_grad()
    np.random.seed(rank)
    out = m(batch)
    ls = agent.loss_func(out["out_mtm"].flatten(0, 1), out["ans_mtm"].flatten()) + \
        agent.loss_func(out["out_vtm"].flatten(0, 1), out["ans_vtm"].flatten())
    ls.backward()
    return m.arena().grad.clone()


def _worker(rank, world, port, q, zero=False, bf16=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lavender_amd as LA
    from tests.helpers import Tok, make_args
    torch.manual_seed(100 + rank)                       # different initial weights: the broadcast must fix that
    m = LA.LAVENDER_Pretrain_MLM(make_args("micro", "micro", 2, lr=1e-3), Tok()).cuda()
    m.arena()
    agent = LA.Agent_Pretrain_MLM(make_args("micro", "micro", 2, lr=1e-3, deepspeed=zero, grad_comm_bf16=bf16), m)
    agent.prepare_dist_model()
    assert agent.dp.grad_dtype == ("bf16" if bf16 else "fp32")
    agent.dp.HALF_MIN_ELEMS = 4096                      # micro model: let its ranges take the half-precision path
    assert agent.dp is not None and agent.dp.world == world and agent.dp.zero_stage == (1 if zero else 0)
    w0 = m.arena().master.clone()
    # reference: both ranks' local gradients computed in this process with the reducer detached
    listeners, m.arena().listeners = m.arena().listeners, []
    g_mine = _rank_grads(rank, agent, m)
    g_other = _rank_grads(1 - rank, agent, m)
    m.arena().listeners = listeners
    # the real thing: backward (fusion-side ranges are all-reduced while the video backward still runs) + finish()
    agent.dp.begin_step()
    _rank_grads(rank, agent, m)
    agent.dp.finish()
    g_sum = m.arena().grad.clone()
    lo, hi = (agent.dp.lo, agent.dp.hi) if zero else (0, g_sum.numel())           # ZeRO-1: only the own shard holds the sum
    rel = ((g_mine + g_other)[lo:hi] - g_sum[lo:hi]).norm() / g_sum[lo:hi].norm()
    agent.optzr.step(max_norm=1.0, dp=agent.dp)
    if zero:
        assert m.arena().m.numel() == hi - lo                                     # optimizer state for the own shard only
        for reader in (m.state_dict, m.arena().sync_half):                        # stale-master readers refuse until the gather
            with pytest.raises(RuntimeError, match="sharded"):
                reader()
    agent.dp.gather_master()
    m.state_dict()
    torch.cuda.synchronize()
    w1 = m.arena().master
    same16 = bool(torch.equal(m.arena().half.float(), w1.bfloat16().float()))     # bf16 working copy == rounded masters everywhere
    q.put((rank, float(rel), float(w0.double().sum()), float(w1.double().sum()), float((w1.double() ** 2).sum()), same16))
    dist.destroy_process_group()


def _run_two_ranks(zero, bf16=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000) + (37 if zero else 0) + (71 if bf16 else 0)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, zero, bf16)) for r in range(2)]
    for p in ps:
        p.start()
    import queue as _queue
    import time as _time
    res, t0 = [], _time.time()
    while len(res) < len(ps):                                    # fail fast when a rank dies instead of waiting out the timeout
        try:
            res.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead or _time.time() - t0 > 300:
                for p in ps:
                    if p.is_alive():
                        p.terminate()
                pytest.fail(f"data-parallel worker failed (exit codes {[p.exitcode for p in ps]})")
    res = sorted(res)
    for p in ps:
        p.join(timeout=60)
    (r0, rel0, w0a, w0b, s0, h0), (r1, rel1, w1a, w1b, s1, h1) = res
    assert rel0 < 2e-2 and rel1 < 2e-2, (rel0, rel1)            # sum of per-rank grads (atomics: order-dependent fp32 rounding)
    assert w0a == w1a                                            # broadcast from rank 0
    assert w0b == w1b and w0b != w0a and s0 == s1                # identical parameters on both replicas after the step
    assert h0 and h1
    return w0b, s0


def test_two_ranks_allreduce_and_replica_consistency():
    _run_two_ranks(False)


def test_zero1_two_ranks_matches_replicated_step():
    """args.deepspeed (ZeRO-1: reduce-scatter, AdamW on the own shard with sharded m / v, all-gather of the bf16 copy) lands on
    the same parameters as the replicated step (the two differ only in the summation order of the clip norm)."""
    w_ddp, s_ddp = _run_two_ranks(False)
    w_z, s_z = _run_two_ranks(True)
    assert abs(w_ddp - w_z) <= 1e-6 * max(1.0, abs(w_ddp)) and abs(s_ddp - s_z) <= 1e-6 * s_ddp, (w_ddp, w_z, s_ddp, s_z)


def test_bf16_gradient_exchange_keeps_replicas_identical():
    """args.grad_comm_bf16: gradient buckets cross the wire as bf16 (lav_cast_f32_to_bf16 / lav_cast_bf16_to_f32 on the comm
    stream); the summed gradient is within bf16 rounding of the fp32 sum and both replicas (DDP and ZeRO-1) stay bit-identical."""
    _run_two_ranks(False, bf16=True)
    _run_two_ranks(True, bf16=True)


def _rccl_worker(port, q):
    """One rank over the REAL backend ("nccl" = RCCL): exercises the reducer's stream / async-work handling against RCCL
    itself (the two-rank test above has to use gloo because RCCL wants one device per rank)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import lavender_amd as LA
    from lavender_amd.dp import ArenaReducer
    from tests.helpers import Tok, make_args
    torch.manual_seed(7)
    m = LA.LAVENDER_Pretrain_MLM(make_args("micro", "micro", 2, lr=1e-3), Tok()).cuda()
    m.arena()
    agent = LA.Agent_Pretrain_MLM(make_args("micro", "micro", 2, lr=1e-3), m)
    g_ref = _rank_grads(0, agent, m)                       # no reducer attached
    agent.dp = ArenaReducer(m)                             # prepare_dist_model() only attaches it for world > 1
    early = list(agent.dp.early_ranges) + list(agent.dp.stage_ranges.values())
    _rank_grads(0, agent, m)
    done_early = sorted(agent.dp._done)
    agent.dp.finish()
    torch.cuda.synchronize()
    g = m.arena().grad.clone()
    ok = bool(((g - g_ref).norm() / g_ref.norm()).item() < 1e-4) and sorted(early) == done_early and agent.dp._done == []
    agent.optzr.step(max_norm=1.0, grad_div=1.0)
    torch.cuda.synchronize()
    q.put(ok)
    dist.destroy_process_group()


def test_reducer_over_rccl_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29700 + (os.getpid() % 1000), q))
    p.start()
    import queue as _queue
    import time as _time
    t0, res = _time.time(), None
    while res is None:
        try:
            res = q.get(timeout=2)
        except _queue.Empty:
            if p.exitcode not in (None, 0) or _time.time() - t0 > 240:
                if p.is_alive():
                    p.terminate()
                pytest.fail(f"RCCL worker failed (exit code {p.exitcode})")
    p.join(timeout=60)
    assert res is True


def _rccl_variant_worker(port, q, zero, bf16):
    """One RCCL rank through the branches that only run on the real backend: ZeroOneReducer's per-bucket dist.reduce(dst = owner)
    over ProcessGroupNCCL, its all_gather_into_tensor of the bf16 working copy (dp.py `self._nccl`), gather_master(), the
    full-load reset of the sharded flag, and the bf16 gradient buckets (cast on the comm stream, widened back at finish()).
    At one rank every collective is a copy, so the result must equal the plain single-process step."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import lavender_amd as LA
    from lavender_amd.dp import ArenaReducer, ZeroOneReducer
    from tests.helpers import Tok, make_args
    ok = True
    finals = {}
    for with_dp in (False, True):
        torch.manual_seed(7)
        m = LA.LAVENDER_Pretrain_MLM(make_args("micro", "micro", 2, lr=1e-3), Tok()).cuda()
        m.arena()
        agent = LA.Agent_Pretrain_MLM(make_args("micro", "micro", 2, lr=1e-3, deepspeed=zero, grad_comm_bf16=bf16), m)
        if with_dp:
            agent.dp = (ZeroOneReducer if zero else ArenaReducer)(m, grad_dtype="bf16" if bf16 else "fp32")
            agent.dp.HALF_MIN_ELEMS = 4096
            ok &= agent.dp._nccl if zero else True
            agent.dp.begin_step()
        g = _rank_grads(0, agent, m)
        if with_dp:
            done_early = sorted(agent.dp._done)
            ok &= len(done_early) > 0                      # the overlapped exchange fired during the backward
            agent.dp.finish()
            if bf16:                                        # the widened bf16 sums are within bf16 rounding of the local gradient
                g2 = m.arena().grad
                ok &= bool(((g2 - g).norm() / g.norm()).item() < 1e-2)
        agent.optzr.step(max_norm=1.0, dp=agent.dp)
        if with_dp and zero:
            ok &= m.arena().masters_sharded
            agent.dp.gather_master()
            ok &= not m.arena().masters_sharded
            # a second sharded step, then a FULL load_state_dict: the masters are whole again without a gather (arena.py masters_sharded)
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            agent.dp.begin_step(); _rank_grads(0, agent, m); agent.dp.finish()
            agent.optzr.step(max_norm=1.0, dp=agent.dp)
            ok &= m.arena().masters_sharded
            m.load_state_dict(sd)
            ok &= not m.arena().masters_sharded
            m.arena().sync_half_if_stale()
            agent.dp.gather_master()                        # nothing to do, must not raise
        torch.cuda.synchronize()
        finals[with_dp] = (m.arena().master.clone(), m.arena().half.clone())
    tol = 2e-3 if bf16 else 1e-6
    ok &= bool(((finals[True][0] - finals[False][0]).abs().max() <= tol).item())
    ok &= bool(torch.equal(finals[True][1].float(), finals[True][0].bfloat16().float()))
    q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("zero,bf16", [(True, False), (False, True), (True, True)])
def test_zero1_and_bf16_buckets_over_rccl_single_rank(zero, bf16):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_variant_worker, args=(29800 + (os.getpid() % 1000) + 3 * zero + 7 * bf16, q, zero, bf16))
    p.start()
    import queue as _queue
    import time as _time
    t0, res = _time.time(), None
    while res is None:
        try:
            res = q.get(timeout=2)
        except _queue.Empty:
            if p.exitcode not in (None, 0) or _time.time() - t0 > 240:
                if p.is_alive():
                    p.terminate()
                pytest.fail(f"RCCL worker failed (exit code {p.exitcode})")
    p.join(timeout=60)
    assert res is True
