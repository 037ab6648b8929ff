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
    """Same interface as distributed.HipBackend, on CPU tensors, computed by the C oracle (test-only)."""

    def weight_cumsum(self, x, is_log, lse):
        from oracle import cpu
        cum, tot = cpu.weight_cumsum(x.numpy(), is_log, None if lse is None else lse.numpy())
        return torch.from_numpy(cum.view(np.int64)), torch.tensor([0, tot], dtype=torch.int64)

    def resample_systematic(self, cum, base_total, u, N_total, out_begin, n_out):
        from oracle import cpu
        return torch.from_numpy(cpu.resample_systematic(cum.numpy().view(np.uint64), u, N_total, base=int(base_total[0]),
                                                        total_all=int(base_total[1]), out_begin=out_begin, n_out=n_out))

    def gather_rows(self, src, anc):
        from oracle import cpu
        return torch.from_numpy(cpu.gather_rows(src.numpy(), anc.numpy()))

    def gather_rows_into(self, src, anc, dst, col0):
        dst[:, col0: col0 + anc.numel()] = self.gather_rows(src, anc)

    def lse_combine(self, pairs, K_total):
        m = pairs[:, 0].max()
        s = (pairs[:, 1].double() * torch.exp((pairs[:, 0] - m).double())).sum()
        lse = m.double() + torch.log(s)
        return torch.tensor([m, s, lse, lse - np.log(K_total)], dtype=torch.float32)


def _worker(rank, world, port, K, R, heavy, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from genjax_amd import distributed as D
    from oracle import cpu
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    rs = np.random.default_rng(0)
    logw = (rs.standard_normal(K) * (5.0 if heavy else 1.0)).astype(np.float32)
    rows = rs.standard_normal((R, K)).astype(np.float32)
    off, k = D.shard(K, rank, world)
    be = OracleBackend()
    local = torch.from_numpy(cpu.logsumexp(logw[off:off + k], K))
    glob = D.global_lse(local, K, backend=be)
    new_rows, info = D.resample_exchange(torch.from_numpy(rows[:, off:off + k].copy()), torch.from_numpy(logw[off:off + k].copy()),
                                         glob, 0.37, K, backend=be)
    out_q.put((rank, glob.numpy(), new_rows.numpy(), info["sent"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("heavy", [False, True])
@pytest.mark.parametrize("K", [1000, 4097])
def test_sharded_resampling_equals_single_process(K, heavy):
    from oracle import cpu
    world, R = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300) + (1 if heavy else 0) + (2 if K > 2000 else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, R, heavy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process answer
    rs = np.random.default_rng(0)
    logw = (rs.standard_normal(K) * (5.0 if heavy else 1.0)).astype(np.float32)
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
