"""world_size-2 gloo tests of the multi-GPU path on CPU tensors: the sharding / exchange logic of
genjax_amd.distributed with the compute steps supplied by the CPU oracle (the HIP kernels need a GPU).
Checks that the sharded result is IDENTICAL to the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """Same interface as distributed.HipBackend, on CPU tensors, computed by the C oracle (test-only).
    The plan comes from distributed.HostPlan (the host twin of the plan kernel)."""

    def weight_cumsum(self, x, is_log, lse=None, pairs=None, K_total=None):
        from oracle import cpu
        rec = None
        if pairs is not None:
            rec = lse = self.lse_combine(pairs, K_total)
            is_log = True
        cum, tot = cpu.weight_cumsum(x.numpy(), is_log, None if lse is None else lse.numpy())
        return torch.from_numpy(cum.view(np.int64)), torch.tensor([0, tot], dtype=torch.int64), rec

    def plan(self, totals, rank, u, N_total):
        from genjax_amd.distributed import HostPlan
        return HostPlan([int(t) for t in totals.tolist()], rank, u, N_total)

    def shard_resample(self, cum, plan, u, N_total, rows, own_n):
        from oracle import cpu
        anc = torch.from_numpy(cpu.resample_systematic(cum.numpy().view(np.uint64), u, N_total, base=plan.base, total_all=plan.total,
                                                       out_begin=plan.slot0, n_out=plan.n_valid))
        new_rows = torch.full((rows.shape[0], own_n), float("nan"), dtype=rows.dtype)
        if plan.keep_hi > plan.keep_lo:
            a = anc[plan.keep_lo - plan.slot0: plan.keep_hi - plan.slot0]
            new_rows[:, plan.keep_lo - plan.own_lo: plan.keep_hi - plan.own_lo] = self.gather_rows(rows, a)
        return anc, new_rows

    def pack(self, rows, anc, n_valid, n_pre, n_suf):
        idx = torch.cat([anc[:n_pre], anc[n_valid - n_suf: n_valid]])
        return self.gather_rows(rows, idx).t().contiguous()

    def unpack(self, msg, n_lo, n_hi, dst):
        dst[:, :n_lo] = msg[:n_lo].t()
        if n_hi:
            dst[:, dst.shape[1] - n_hi:] = msg[n_lo:].t()

    def resample_systematic(self, cum, base_total, u, N_total, out_begin, n_out):
        from oracle import cpu
        return torch.from_numpy(cpu.resample_systematic(cum.numpy().view(np.uint64), u, N_total, base=int(base_total[0]),
                                                        total_all=int(base_total[1]), out_begin=out_begin, n_out=n_out))

    def resample_multinomial(self, cum, base_total, key, N_total):
        from oracle import cpu
        return torch.from_numpy(cpu.resample_multinomial(cum.numpy().view(np.uint64), key, N_total, base=int(base_total[0]),
                                                         total_all=int(base_total[1])))

    def gather_rows(self, src, anc):
        from oracle import cpu
        return torch.from_numpy(cpu.gather_rows(src.numpy(), anc.numpy()))

    def lse_combine(self, pairs, K_total):
        m = pairs[:, 0].max()
        s = (pairs[:, 1].double() * torch.exp((pairs[:, 0] - m).double())).sum()
        lse = m.double() + torch.log(s)
        return torch.tensor([m, s, lse, lse - np.log(K_total)], dtype=torch.float32)


def _weights(rs, K, heavy):
    """heavy: False = mild, True = wide spread, "first"/"last" = (nearly) all mass on the first / last few particles"""
    logw = (rs.standard_normal(K) * (5.0 if heavy else 1.0)).astype(np.float32)
    if heavy == "first":
        logw[7:] -= 80.0
    elif heavy == "last":
        logw[:-5] -= 80.0
    return logw


def _worker(rank, world, port, K, R, heavy, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from genjax_amd import distributed as D
    from oracle import cpu
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    rs = np.random.default_rng(0)
    logw = _weights(rs, K, heavy)
    rows = rs.standard_normal((R, K)).astype(np.float32)
    off, k = D.shard(K, rank, world)
    be = OracleBackend()
    local = torch.from_numpy(cpu.logsumexp(logw[off:off + k], K))
    glob = D.global_lse(local, K, backend=be)
    if heavy:        # the gathered-pairs entry: the combine happens inside the prefix-sum step
        new_rows, info = D.resample_exchange(torch.from_numpy(rows[:, off:off + k].copy()), torch.from_numpy(logw[off:off + k].copy()),
                                             None, 0.37, K, backend=be, pairs=D.gather_lse_pairs(local))
        np.testing.assert_array_equal(info["lse"].numpy(), glob.numpy())
    else:
        new_rows, info = D.resample_exchange(torch.from_numpy(rows[:, off:off + k].copy()), torch.from_numpy(logw[off:off + k].copy()),
                                             glob, 0.37, K, backend=be)
    out_q.put((rank, glob.numpy(), new_rows.numpy(), info["sent"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("heavy", [False, True, "first", "last"])
@pytest.mark.parametrize("K,world", [(1000, 2), (4097, 2), (1001, 3)])
def test_sharded_resampling_equals_single_process(K, world, heavy):
    from oracle import cpu
    R = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + [False, True, 'first', 'last'].index(heavy) * 16 + (2 if K > 2000 else 0) + 4 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, R, heavy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process answer
    rs = np.random.default_rng(0)
    logw = _weights(rs, K, heavy)
    rows = rs.standard_normal((R, K)).astype(np.float32)
    lse = cpu.logsumexp(logw, K)
    cum, tot = cpu.weight_cumsum(logw, True, lse)
    want = cpu.gather_rows(rows, cpu.resample_systematic(cum, 0.37, K))
    got = np.concatenate([r[2] for r in res], axis=1)
    np.testing.assert_array_equal(got, want)                       # bit-identical to the unsharded run
    for r in res:
        np.testing.assert_allclose(r[1][2:], lse[2:], rtol=1e-6, atol=1e-6)
    if heavy:
        assert sum(r[3] for r in res) > 0                          # uneven weights force a real exchange


def _multinomial_worker(rank, world, port, K, R, heavy, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from genjax_amd import distributed as D
    from oracle import cpu
    D.init_from_env("gloo")
    rs = np.random.default_rng(0)
    logw = _weights(rs, K, heavy)
    rows = rs.standard_normal((R, K)).astype(np.float32)
    off, k = D.shard(K, rank, world)
    be = OracleBackend()
    local = torch.from_numpy(cpu.logsumexp(logw[off:off + k], K))
    new_rows, info = D.resample_exchange_multinomial(torch.from_numpy(rows[:, off:off + k].copy()), torch.from_numpy(logw[off:off + k].copy()),
                                                     (5, 6), K, D.gather_lse_pairs(local), backend=be)
    out_q.put((rank, new_rows.numpy(), info["sent"], info["received"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("heavy", [False, True, "first"])
@pytest.mark.parametrize("K,world", [(1000, 2), (1001, 3)])
def test_sharded_multinomial_equals_single_process(K, world, heavy):
    """all-to-all multinomial resampling over gloo == the unsharded multinomial draw, bit for bit"""
    from oracle import cpu
    R = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 300) + [False, True, 'first'].index(heavy) * 8 + 4 * world
    procs = [ctx.Process(target=_multinomial_worker, args=(r, world, port, K, R, heavy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rs = np.random.default_rng(0)
    logw = _weights(rs, K, heavy)
    rows = rs.standard_normal((R, K)).astype(np.float32)
    lse = cpu.logsumexp(logw, K)
    cum, tot = cpu.weight_cumsum(logw, True, lse)
    want = cpu.gather_rows(rows, cpu.resample_multinomial(cum, (5, 6), K))
    got = np.concatenate([r[1] for r in res], axis=1)
    np.testing.assert_array_equal(got, want)
    assert sum(r[2] for r in res) == sum(r[3] for r in res) > 0          # children crossed ranks, nothing lost


def _agree_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from genjax_amd import distributed as D
    r, w = D.init_from_env("gloo")
    res = (D.all_agree(True), D.all_agree(rank != 1), D.all_agree(False))
    out_q.put((rank, res))
    dist.destroy_process_group()


def test_all_agree_is_a_logical_and_over_the_ranks():
    """distributed.all_agree: what bench.py and BootstrapFilter use to leave the peer-mapped transport TOGETHER when one rank's
    launch cannot be co-resident or timed out (a rank that went on alone would wait for granules nobody publishes)"""
    from genjax_amd import distributed as D
    assert D.all_agree(True) and not D.all_agree(False)              # no process group: the local answer
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 100
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(v == (True, False, False) for v in res.values()), res
