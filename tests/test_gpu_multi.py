"""Multi-rank tests of the sharded path on real GPUs.

* test_rccl_two_ranks_* need >= 2 GPUs (they skip on a 1-GPU box): one process per GPU, nccl (= RCCL) process
  group, the one-call RCCL exchange (gjx_shard_resample_step: both all-gathers, device plan, grouped ncclSend /
  ncclRecv with REAL peers) against the unsharded result, bit for bit, with weights uneven enough that children
  cross the rank boundary in both directions; then the sharded bootstrap filter against the one-GPU filter.
* test_bench_launcher_* run everywhere with a GPU: `python bench.py --gpus 2` with no torchrun environment must
  spawn its own ranks and report n_gpus = 2 (on a 1-GPU box both ranks share device 0 and gloo carries the
  collectives, which is what the RCCL-free transport is for).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _weights(K, spread):
    rs = np.random.default_rng(17)
    lw = (rs.standard_normal(K) * (4.0 if spread == "wide" else 1.0)).astype(np.float32)
    if spread == "first":
        lw[K // 5:] -= 50.0          # rank 0 owns nearly all the mass: it sends to everyone, the others only receive
    if spread == "last":
        lw[: K - K // 9] -= 50.0
    if spread == "tilt":
        lw += np.linspace(-3.0, 3.0, K).astype(np.float32)      # mass leans to the high ranks: children move down AND up
    return lw, rs.standard_normal((5, K)).astype(np.float32)


def _rccl_worker(rank, world, port, K, spread, method, q):
    try:
        import torch
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        from genjax_amd import distributed as D
        from genjax_amd import kernels
        torch.cuda.set_device(rank)
        D.init_from_env("nccl")
        lw, rows = _weights(K, spread)
        off, k = D.shard(K, rank, world)
        dev = torch.device("cuda", rank)
        lw_d = torch.as_tensor(lw[off:off + k]).to(dev)
        rows_d = torch.as_tensor(rows[:, off:off + k].copy()).to(dev)
        local = kernels.logsumexp(lw_d, K)
        res = D.ShardedResampler(k, rows.shape[0], K, dev, transport="rccl")
        if method == "systematic":
            out, rec = res.step(rows_d, lw_d, local, 0.37)
        else:
            out, rec = res.step_multinomial(rows_d, lw_d, local, (3, 4))
        torch.cuda.synchronize()
        st = res.stats()
        # unsharded reference on this rank's own GPU (every rank has the full problem)
        lw_f = torch.as_tensor(lw).to(dev)
        lse_f = kernels.logsumexp(lw_f, K)
        if method == "systematic":
            anc = kernels.resample_indices(lw_f, 0.37, K, lse=lse_f)
        else:
            cum, bt = kernels.weight_cumsum(lw_f, True, lse_f)
            anc = kernels.resample_multinomial(cum, bt, (3, 4), K)
        want = kernels.gather_rows(torch.as_tensor(rows).to(dev), anc)
        own_lo, own_n = D.shard(K, rank, world)
        same = bool(torch.equal(out, want[:, own_lo:own_lo + own_n]))
        q.put((rank, same, rec.cpu().numpy(), lse_f.cpu().numpy(), st))
        res.close()
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None))
        raise


@pytest.mark.parametrize("method", ["systematic", "multinomial"])
@pytest.mark.parametrize("spread", ["tilt", "wide", "first", "last"])
def test_rccl_two_ranks_exchange_equals_unsharded(spread, method):
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL send/recv between real peers)")
    import torch.multiprocessing as mp
    world = min(_n_gpus(), 4) if spread in ("first", "last") else 2
    K = 200_003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, K, spread, method, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        assert p.exitcode == 0
    for rank, same, rec, lse_full, st in res:
        assert same, f"rank {rank}: sharded {method} result differs from the unsharded one"
        np.testing.assert_allclose(rec[2:], lse_full[2:], rtol=2e-6, atol=1e-6)
        assert st["transport"] == "rccl" and st["rccl_ranks"] == world
    # children really crossed the fabric
    assert sum(r[4]["sent"] for r in res) > 0 and sum(r[4]["sent"] for r in res) == sum(r[4]["received"] for r in res)


def _run_bench(extra_args, env_extra=None, timeout=900):
    """-> the compact line bench.py prints LAST (what the driver parses: < 4 KB, parseable, roofline inside), with the full record it
    points to (the side file) under "_full" """
    import tempfile
    side = tempfile.NamedTemporaryFile(suffix=".json", delete=False).name
    extra_args = list(extra_args) + ["--extra-file", side]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip().splitlines()[-1] == lines[0], out.stdout[-2000:]
    assert len(lines[0]) < 4096
    rec = json.loads(lines[0])
    assert rec["roofline"]["frac"] > 0 and "exchange_stats" not in rec["config"] and "extra" not in rec
    with open(side) as f:
        rec["_full"] = json.load(f)
    os.unlink(side)
    return rec


def test_bench_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun: two ranks, one JSON line from rank 0, n_gpus = 2, same log-ML as
    one GPU (streams are indexed by the global particle index)."""
    multi = _n_gpus() >= 2
    env = {} if multi else {"GJX_ALL_ON_DEVICE0": "1", "GJX_DIST_BACKEND": "gloo"}
    one = _run_bench(["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--k-per-gpu", str(1 << 17)])
    two = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--k-per-gpu", str(1 << 16)], env)
    assert two["n_gpus"] == 2 and two["config"]["k_particles_total"] == 1 << 17
    # peer-mapped windows when the ranks can map each other's memory (always on one shared GPU), else the collective transport
    assert two["config"]["exchange"] in (("peer", "rccl") if multi else ("peer",))
    # the line carries the summary of the exchange; every rank's view is in the full record
    assert two["config"]["exchange_summary"]["transports_agree"] is True and two["config"]["exchange_summary"]["any_status_bit"] is False
    full = two["_full"]["config"]["exchange_stats"]
    assert len(full["per_rank"]) == 2
    if two["config"]["exchange"] == "rccl":
        assert full["rccl_ranks"] == 2
    else:
        assert full["status"] == 0
    # same global collection, same global stream: the log-ML estimates agree to LSE rounding
    assert abs(two["log_ml"] - one["log_ml"]) <= 2e-5 * abs(one["log_ml"])


def test_bench_launcher_config4_ssm():
    """config 4 shape through the launcher: sharded bootstrap filter, K_total fixed, log-ML equal to the 1-rank run
    of the same K to LSE rounding (small K and T here; the full size is bench.py --workload ssm --gpus 8)."""
    multi = _n_gpus() >= 2
    env = {} if multi else {"GJX_ALL_ON_DEVICE0": "1", "GJX_DIST_BACKEND": "gloo"}
    # default scheme on both sides: tile-scaled weights, on two ranks through the peer-mapped windows (same granules, same
    # integers: the log-ML differs only by the summation order of the LSE records); then the collective transport with
    # global-maximum weights against the one-GPU run under that scheme
    one = _run_bench(["--workload", "ssm", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--ssm-k-total", str(1 << 14)])
    two = _run_bench(["--workload", "ssm", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--ssm-k-total",
                      str(1 << 14)], env)
    assert two["n_gpus"] == 2 and two["config"]["k_particles_total"] == 1 << 14 and two["config"]["k_particles_per_gpu"] == 1 << 13
    assert two["config"]["exchange"] in (("peer", "rccl") if multi else ("peer",))
    assert abs(two["log_ml"] - one["log_ml"]) <= 1e-5 * abs(one["log_ml"])
    one_g = _run_bench(["--workload", "ssm", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--ssm-k-total", str(1 << 14),
                        "--ssm-weights", "global_max"])
    two_g = _run_bench(["--workload", "ssm", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--ssm-k-total",
                        str(1 << 14), "--ssm-weights", "global_max"], env)
    assert two_g["config"]["exchange"] in (("rccl", "torch") if multi else ("torch",))
    assert abs(two_g["log_ml"] - one_g["log_ml"]) <= 1e-5 * abs(one_g["log_ml"])


@pytest.mark.parametrize("move", [None, (1, 0.4)])
def test_peer_filter_on_real_peers_equals_unsharded(move):
    """the peer-mapped sharded filter (gjx_ssm_filter_peer[_move], csrc/gjx_peer.hip) with ONE GPU PER RANK — the windows
    are mapped over xGMI — against the unsharded one-launch filter: particles and weights bit for bit.  Needs >= 2 GPUs
    (tests/test_gpu_peer.py runs the same ranks on one shared GPU)."""
    n = _n_gpus()
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_peer as TP
    from genjax_amd import _abi as A
    from genjax_amd import kernels
    from genjax_amd.inference.pf import LinearGaussianSSM
    world, K_total, T, dx = min(n, 4), 1 << 17, 12, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=TP._filter_worker, args=(r, world, port, K_total, T, dx, A.RNG_FLAT, q, move, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for r in res:
        assert r[1] != "error", r[2]
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s = TP._problem(dx, T)
    ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
    ys = torch.as_tensor(np.asarray(s["y"], np.float32)).cuda()
    for rep in range(3):
        if move is None:
            ref = kernels.ssm_filter(ssm.c_struct("cuda"), (0, 5 + rep), A.RNG_FLAT, ys, K_total, weights=A.WEIGHTS_TILE_SCALED)
        else:
            ref = kernels.ssm_filter_move(ssm.c_struct("cuda"), (0, 5 + rep), A.RNG_FLAT, ys, K_total, move[0], move[1])
        torch.cuda.synchronize()
        np.testing.assert_array_equal(np.concatenate([r[1][rep][0] for r in res], axis=1), ref["x"].cpu().numpy())
        np.testing.assert_array_equal(np.concatenate([r[1][rep][1] for r in res]), ref["logw"].cpu().numpy())
    assert all(r[2] == 0 for r in res), [r[2] for r in res]
    assert all(r[3] == 1 for r in res)                             # every rank had its device to itself
