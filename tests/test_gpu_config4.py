"""BASELINE config 4 at its own size, as a dry run on the ONE GPU of the test box: 8 processes share the device, map each
other's windows through hipIpc and run the sharded kernels concurrently (csrc/gjx_peer.hip, gjx_pfilter.inl) — the same
code path as 8 GPUs over xGMI, with the device's own memory in place of the fabric.

    SSM bootstrap filter, K_total = 2^22 (2^19 per rank), T = 256, d_x = 8      (BASELINE.json configs[3])

with GJX_PEER_VERIFY=1: every pulled particle row is checked against its owner's check word and every re-scanned source
tile against the total in its granule (include/gjx.h), so a stale read cannot go unnoticed.  Required: particles,
log-weights and ancestors bit-identical to gjx_ssm_filter_scheme(tile-scaled) on ONE rank at K = 2^22, log-ML within rtol
1e-4 of the float64 Kalman value, status word 0 on every rank.  The sharded ImportanceK resampling step runs at the largest
size whose grid is co-resident when 8 ranks share one device (8 x 2^17; on 8 devices each rank's 2^20 fit)."""
import os
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _filter_worker(rank, world, port, K_total, T, dx, out_dir, env):
    try:
        import torch
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
        from genjax_amd import _abi as A
        from genjax_amd import distributed as D
        from genjax_amd import kernels, workloads
        from genjax_amd.inference.pf import LinearGaussianSSM
        D.init_from_env("gloo")
        torch.cuda.set_device(0)
        s = workloads.ssm_problem(dx=dx, T=T)
        ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
        ys = torch.as_tensor(np.asarray(s["y"], np.float32)).cuda()
        ctx = kernels.PeerContext(K_total // world, dx, "cuda")
        o = ctx.ssm_filter(ssm.c_struct("cuda"), (0, 7), A.RNG_FLAT, ys, want_ancestors=True)
        torch.cuda.synchronize()
        st = ctx.status()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=o["x"].cpu().numpy(), logw=o["logw"].cpu().numpy(),
                 lse=o["lse_steps"].cpu().numpy(), anc=o["ancestors"].cpu().numpy(), status=st, share=ctx.ranks_on_device)
        ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


def _run_ranks(target, world, args, out_dir, timeout=900, tries=3):
    """the ranks as processes sharing the ONE device.  A run in which a rendezvous timed out (status bit 0: eight processes' grids have
    to be resident together, and the device's scheduler may hold one of them back longer than a lane's poll budget) is what the
    product's host layer repeats (results of such a run are undefined by contract): so does the test, up to `tries` times"""
    for attempt in range(tries):
        try:
            res = _run_ranks_once(target, world, args, out_dir, timeout)
        except AssertionError as e:
            # a rank that dies takes the others' host barriers down with it ("Connection closed by peer"); seen once in a dozen runs
            # of the whole suite, never twice in a row: one more try, then the error stands
            if attempt == tries - 1 or "Connection closed by peer" not in str(e):
                raise
            print("a rank lost its process group (attempt %d): the run is repeated\n%s" % (attempt + 1, str(e)[-600:]))
            res = None
        if res is not None:
            timed_out = any(int(r[k]) & 1 for r in res for k in r.files if k.startswith("status"))
            if not timed_out or attempt == tries - 1:
                return res
            print("a rendezvous timed out (attempt %d): the run is repeated" % (attempt + 1))
        for f in os.listdir(out_dir):
            if f.endswith(".npz") or f.endswith(".err"):
                os.remove(os.path.join(out_dir, f))


def _run_ranks_once(target, world, args, out_dir, timeout=900):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=timeout)
    errs = [f + ": " + open(os.path.join(out_dir, f)).read() for f in sorted(os.listdir(out_dir)) if f.endswith(".err")]
    # (every rank's report: the first rank to fail takes the others' barriers down with it, and it need not be rank 0)
    assert not errs, "\n".join(e[-1200:] for e in errs)
    for p in procs:
        assert p.exitcode == 0, [q.exitcode for q in procs]
    return [np.load(os.path.join(out_dir, "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("data", ["coarse", "fine"])
def test_config4_filter_dry_run(tmp_path, data):
    """8 ranks x 2^19 particles, T = 256, d_x = 8, GJX_PEER_VERIFY=1; DATA window coarse (default) and fine-grained"""
    import torch
    from genjax_amd import _abi as A
    from genjax_amd import kernels, workloads
    from genjax_amd.inference.pf import LinearGaussianSSM
    from oracle import closed_form as cf
    world, K_total, T, dx = 8, 1 << 22, 256, 8
    res = _run_ranks(_filter_worker, world, (K_total, T, dx, str(tmp_path), dict(GJX_PEER_VERIFY="1", GJX_PEER_DATA=data)), str(tmp_path))
    assert all(int(r["status"]) == 0 for r in res), [int(r["status"]) for r in res]      # no timeout, no dead step, no verify mismatch
    assert all(int(r["share"]) == world for r in res)
    s = workloads.ssm_problem(dx=dx, T=T)
    ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
    ys = torch.as_tensor(np.asarray(s["y"], np.float32)).cuda()
    ref = kernels.ssm_filter(ssm.c_struct("cuda"), (0, 7), A.RNG_FLAT, ys, K_total, weights=A.WEIGHTS_TILE_SCALED)
    torch.cuda.synchronize()
    assert kernels.workspace_status(ref["_status_ws"], raise_on_error=False) == 0
    np.testing.assert_array_equal(np.concatenate([r["x"] for r in res], axis=1), ref["x"].cpu().numpy())
    np.testing.assert_array_equal(np.concatenate([r["logw"] for r in res]), ref["logw"].cpu().numpy())
    np.testing.assert_array_equal(np.concatenate([r["anc"] for r in res]), ref["ancestors"].cpu().numpy())
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    for r in res:                                 # every rank holds the global records
        np.testing.assert_allclose(r["lse"][:, 2:], ref["lse_steps"].cpu().numpy()[:, 2:], rtol=2e-6, atol=2e-6)
        log_ml = float(r["lse"][:, 3].astype(np.float64).sum())
        assert abs(log_ml - exact) <= 1e-4 * abs(exact), (log_ml, exact)


def _generic_filter_worker(rank, world, port, K_total, T, dx, out_dir, env):
    try:
        import torch
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
        import genjax_amd as genjax
        from genjax_amd import C, distributed as D, kernels, workloads
        from genjax_amd.inference import BootstrapFilter
        D.init_from_env("gloo")
        torch.cuda.set_device(0)
        scan, carry0, s = workloads.lgssm_scan(dx, T)
        ys = np.asarray(s["y"], np.float32)
        bf = BootstrapFilter(scan, K_total // world, resampler=env.get("GJX_TEST_RESAMPLER", "systematic"))
        rows = max(p.n_slots for p in bf.step_programs(C["y"].set(ys), (carry0, None)))
        ctx = kernels.PeerContext(K_total // world, rows, "cuda")
        o = bf.run_peer(ctx, genjax.key(7), C["y"].set(ys), (carry0, None), want_ancestors=True)
        torch.cuda.synchronize()
        st = ctx.status()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=bf.latent(o, "x").cpu().numpy(), logw=o["logw"].cpu().numpy(),
                 lse=o["lse_steps"].cpu().numpy(), anc=o["ancestors"].cpu().numpy(), status=st, share=ctx.ranks_on_device,
                 grid=o["info"]["grid"], tpb=o["info"]["tiles_per_block"])
        ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


@pytest.mark.parametrize("resampler", ["systematic", "multinomial"])
def test_config4_generic_filter_dry_run(tmp_path, resampler):
    """config 4's workload with the model written as @gen + .scan: 8 ranks x 2^19 particles, T = 256, d_x = 8, GJX_PEER_VERIFY=1 —
    gjx_scan_filter_peer (the filter kernel GENERATED for the step program on the shared skeleton, sharded flavour).  Required:
    states, log-weights and ancestors bit-identical to the one-rank generic filter at K = 2^22, log-ML within rtol 1e-4 of the float64
    Kalman value, status word 0 on every rank.  resampler="multinomial": north_star's "multinomial resampling" over the sharded
    collection by sorted uniforms (gjx_scan_filter_peer_opts) at the full size of config 4"""
    import genjax_amd as genjax
    from genjax_amd import C, workloads
    from genjax_amd.inference import BootstrapFilter
    from oracle import closed_form as cf
    world, K_total, T, dx = 8, 1 << 22, 256, 8
    res = _run_ranks(_generic_filter_worker, world, (K_total, T, dx, str(tmp_path), dict(GJX_PEER_VERIFY="1", GJX_TEST_RESAMPLER=resampler)), str(tmp_path))
    assert all(int(r["status"]) == 0 for r in res), [int(r["status"]) for r in res]
    assert all(int(r["share"]) == world for r in res)
    scan, carry0, s = workloads.lgssm_scan(dx, T)
    ys = np.asarray(s["y"], np.float32)
    bf = BootstrapFilter(scan, K_total, resampler=resampler)
    ref = bf.run(genjax.key(7), C["y"].set(ys), (carry0, None))
    assert not ref["degenerate"]
    np.testing.assert_array_equal(np.concatenate([r["x"] for r in res], axis=1), bf.latent(ref, "x").cpu().numpy())
    np.testing.assert_array_equal(np.concatenate([r["logw"] for r in res]), ref["logw"].cpu().numpy())
    np.testing.assert_array_equal(np.concatenate([r["anc"] for r in res]), ref["ancestors"].cpu().numpy())
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"], q0=float(s["q"]))
    for r in res:
        np.testing.assert_allclose(r["lse"][:, 2:], ref["lse_steps"].cpu().numpy()[:, 2:], rtol=2e-6, atol=2e-6)
        log_ml = float(r["lse"][:, 3].astype(np.float64).sum())
        assert abs(log_ml - exact) <= 1e-4 * abs(exact), (log_ml, exact)
    print("generic filter, config-4 dry run: grid", int(res[0]["grid"]), "x", int(res[0]["tpb"]), "tiles per block; one-rank reference:", ref["info"])


def test_verify_mode_catches_a_bad_row(tmp_path):
    """the check itself: with GJX_PEER_VERIFY_FAULT=<rank> that rank publishes check words that do not belong to its rows —
    what a stale or torn row looks like to a reader — and every rank that pulls from it must raise GJX_STATUS_VERIFY_MISMATCH
    (both kernels: the sharded resampling step and the sharded filter)"""
    res = _run_ranks(_fault_worker, 2, (1 << 15, 4, str(tmp_path)), str(tmp_path), timeout=300)
    assert any(int(r["status_resample"]) & 4 for r in res), [int(r["status_resample"]) for r in res]
    assert any(int(r["status_filter"]) & 4 for r in res), [int(r["status_filter"]) for r in res]


def _fault_worker(rank, world, port, K_total, dx, out_dir):
    try:
        import torch
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0", GJX_PEER_VERIFY="1", GJX_PEER_VERIFY_FAULT="1")
        from genjax_amd import _abi as A
        from genjax_amd import distributed as D
        from genjax_amd import kernels, workloads
        from genjax_amd.inference.pf import LinearGaussianSSM
        D.init_from_env("gloo")
        torch.cuda.set_device(0)
        K = K_total // world
        ctx = kernels.PeerContext(K, dx, "cuda")
        rs = np.random.default_rng(3)
        sl = slice(rank * K, (rank + 1) * K)
        ctx.logw[0].copy_(torch.as_tensor(rs.standard_normal(K_total).astype(np.float32)[sl]))
        ctx.rows[0].copy_(torch.as_tensor(rs.standard_normal((dx, K_total)).astype(np.float32)[:, sl]))
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        ctx.resample_gather(0, 0.3)
        torch.cuda.synchronize()
        st_r = ctx.status()
        dist.barrier()
        s = workloads.ssm_problem(dx=dx, T=6)
        ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
        ys = torch.as_tensor(np.asarray(s["y"], np.float32)).cuda()
        ctx.ssm_filter(ssm.c_struct("cuda"), (0, 7), A.RNG_FLAT, ys)
        torch.cuda.synchronize()
        st_f = ctx.status()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), status_resample=st_r, status_filter=st_f)
        ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


def _resample_worker(rank, world, port, K_total, R, out_dir):
    try:
        import torch
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0", GJX_PEER_VERIFY="1")
        from genjax_amd import distributed as D
        from genjax_amd import kernels
        D.init_from_env("gloo")
        torch.cuda.set_device(0)
        K = K_total // world
        ctx = kernels.PeerContext(K, R, "cuda")
        sl = slice(rank * K, (rank + 1) * K)
        outs = {}
        for call in range(2):
            rs = np.random.default_rng(100 + call)
            lw = (rs.standard_normal(K_total) * 2.0).astype(np.float32)
            rows = rs.standard_normal((R, K_total)).astype(np.float32)
            p = call & 1
            ctx.logw[p].copy_(torch.as_tensor(lw[sl]))
            ctx.rows[p].copy_(torch.as_tensor(rows[:, sl]))
            anc = torch.empty(K, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            if dist.is_initialized():
                dist.barrier()      # the ranks launch together (each spent a different time making its inputs)
            out, _ = ctx.resample_gather(p, 0.41 + 0.2 * call, anc=anc)
            torch.cuda.synchronize()
            outs["rows%d" % call] = out.cpu().numpy()
            outs["anc%d" % call] = anc.cpu().numpy()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), status=ctx.status(), **outs)
        ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


def test_config4_resample_gather_dry_run(tmp_path):
    """gjx_peer_resample_gather on 8 ranks sharing the device, 8 x 2^17 particles x 17 rows (the co-resident limit of ONE
    device: 1024 blocks in total), verify mode on == gjx_resample_indices_tiled + gjx_gather_rows on the whole collection"""
    import torch
    from genjax_amd import kernels
    world, K_total, R = 8, 1 << 20, 17
    res = _run_ranks(_resample_worker, world, (K_total, R, str(tmp_path)), str(tmp_path), timeout=600)
    assert all(int(r["status"]) == 0 for r in res), [int(r["status"]) for r in res]
    for call in range(2):
        rs = np.random.default_rng(100 + call)
        lw = (rs.standard_normal(K_total) * 2.0).astype(np.float32)
        rows = rs.standard_normal((R, K_total)).astype(np.float32)
        anc = kernels.resample_indices_tiled(torch.as_tensor(lw).cuda(), 0.41 + 0.2 * call)
        want = kernels.gather_rows(torch.as_tensor(rows).cuda(), anc).cpu().numpy()
        np.testing.assert_array_equal(np.concatenate([r["anc%d" % call] for r in res]), anc.cpu().numpy())
        np.testing.assert_array_equal(np.concatenate([r["rows%d" % call] for r in res], axis=1), want)
