"""Tile-scaled fixed-point weights (include/gjx.h GJX_WEIGHTS_TILE_SCALED): the multi-launch resampler against the
oracle's restatement (integer logic bit for bit from the device's quantised weights, the exp2 within float32 bounds),
the one-launch filter against the step-by-step loop (bit-identical), and the filter's log-ML against the Kalman filter."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from genjax_amd import _abi as A       # noqa: E402
from genjax_amd import core            # noqa: E402
from oracle import closed_form as cf   # noqa: E402


@pytest.fixture(scope="module")
def K_():
    from genjax_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu
    return cpu


def _np(t):
    return t.detach().cpu().numpy()


def _weight_shapes(K, rs):
    lw = rs.standard_normal(K).astype(np.float32)
    yield "normal", lw
    yield "wide", (lw * 12.0 - 300.0).astype(np.float32)
    spiky = np.full(K, -80.0, np.float32)
    spiky[rs.integers(0, K, max(1, K // 500))] = 0.0
    yield "spiky", spiky
    holes = lw.copy()
    holes[rs.random(K) < 0.3] = -np.inf
    holes[rs.random(K) < 0.01] = np.nan
    if K > 2048:
        holes[1024:2048] = -np.inf                     # a dead tile
        holes[2048:3072] -= 150.0                      # a tile that is shifted out
    yield "holes", holes
    one = np.full(K, -np.inf, np.float32)
    one[K // 3] = -1234.5
    yield "one_live", one
    yield "ramp", np.linspace(-60000.0, 0.0, K).astype(np.float32)   # tile exponents span the whole shift range


@pytest.mark.parametrize("K,N", [(1, 1), (777, 777), (1024, 1024), (1025, 3000), (10_000, 10_000), (1 << 18, 1 << 18), ((1 << 20) + 5, 1 << 19)])
def test_tiled_resampler_against_oracle(K_, oracle, K, N):
    import torch
    rs = np.random.default_rng(K)
    for name, lw in _weight_shapes(K, rs):
        for u in (0.0, 0.3718, 0.999999):
            anc, q, e = K_.resample_indices_tiled(torch.as_tensor(lw).cuda(), u, N, want_q=True)
            anc, q, e = _np(anc), _np(q).view(np.uint32), _np(e)
            # (1) everything but the exp2: the oracle run on the device's quantised weights -> the same ancestors, bit for bit
            anc_o, q_same, e_o, dead = oracle.resample_systematic_tiled(lw, u, N, q=q)
            assert not dead, name
            np.testing.assert_array_equal(e, e_o, err_msg=name)
            np.testing.assert_array_equal(anc, anc_o, err_msg=f"{name} u={u}")
            if u != 0.3718:
                continue
            # (2) the exp2: device v_exp_f32 vs libm exp2f on the same fma argument, a few float32 ulps of the weight
            _, q_o, _, _ = oracle.resample_systematic_tiled(lw, u, N)
            np.testing.assert_allclose(q.astype(np.float64), q_o.astype(np.float64), rtol=4e-7, atol=1.0, err_msg=name)
            assert q.max() <= (1 << 29)


@pytest.mark.parametrize("K,N", [(1, 1), (777, 777), (1023, 1024), (1024, 1023), (1025, 3000), (10_000, 10_000), (1 << 18, 1 << 18), ((1 << 20) + 5, 1 << 19)])
def test_sorted_multinomial_resampler_against_oracle(K_, oracle, K, N):
    """gjx_resample_sorted_multinomial_tiled: multinomial resampling by sorted uniforms (exponential spacings) on the tile-scaled
    weight line.  The spacings and their sums are integers on both sides, the threshold one double multiply: the oracle on the
    device's quantised weights gives the same ancestors BIT FOR BIT; the ancestors are non-decreasing; and they are a multinomial
    draw — the counts of a flat collection have the multinomial's variance 1 - 1/K (a systematic comb would have 0)"""
    import torch
    rs = np.random.default_rng(K + 1)
    for name, lw in _weight_shapes(K, rs):
        for key in ((0, 0), (0x12345678, 0x9ABCDEF0)):
            anc, q, e = K_.resample_sorted_multinomial_tiled(torch.as_tensor(lw).cuda(), key, N, want_q=True)
            anc, q, e = _np(anc), _np(q).view(np.uint32), _np(e)
            anc_o, dead = oracle.resample_sorted_multinomial_tiled(lw, key, N, q=q)
            np.testing.assert_array_equal(anc, anc_o, err_msg=f"{name} key={key}")
            assert dead == (not q.any()), name           # (a 1-particle collection whose particle is a hole: identity on both sides)
            assert (np.diff(anc) >= 0).all() and (dead or (q[anc] > 0).all()), name
    if K >= 10_000:
        flat = torch.zeros(K, device="cuda")
        var = []
        for k in range(6):
            cnt = np.bincount(_np(K_.resample_sorted_multinomial_tiled(flat, (77, k), K)), minlength=K)
            var.append(cnt.var())
        assert abs(np.mean(var) - 1.0) < 6.0 * np.sqrt(3.0 / (6 * K)) + 2.0 / K, np.mean(var)      # (Poisson-like counts: var of a sample variance ~ 3 / K)
        # and the draws follow the weights: the mass of the collection's heavier half
        lw = rs.standard_normal(K).astype(np.float32)
        w = np.exp(lw.astype(np.float64)); w /= w.sum()
        heavy = lw > 0
        got = np.mean([heavy[_np(K_.resample_sorted_multinomial_tiled(torch.as_tensor(lw).cuda(), (5, k), K))].mean() for k in range(8)])
        p = w[heavy].sum()
        assert abs(got - p) < 5.0 * np.sqrt(p * (1 - p) / (8 * K)), (got, p)


def test_sorted_multinomial_dead_collection_and_workspace(K_):
    import torch
    lw = torch.full((5000,), float("-inf"), device="cuda")
    ws = K_.workspace(A.OP_RESAMPLE, 5000, lw.device)
    anc = K_.resample_sorted_multinomial_tiled(lw, (1, 2), ws=ws)
    assert (_np(anc) == np.arange(5000)).all()
    assert K_.workspace_status(ws, raise_on_error=False) & 2
    with pytest.raises(K_.GjxError, match="workspace too small"):        # N far beyond K: the slot-tile sums do not fit
        K_.resample_sorted_multinomial_tiled(torch.zeros(8, device="cuda"), (1, 2), N=1 << 24)


def test_tiled_dead_collection_sets_status(K_):
    import torch
    lw = torch.full((5000,), float("-inf"), device="cuda")
    ws = K_.workspace(A.OP_RESAMPLE, 5000, lw.device)
    anc = K_.resample_indices_tiled(lw, 0.5, ws=ws)
    assert (_np(anc) == np.arange(5000)).all()
    assert K_.workspace_status(ws, raise_on_error=False) & 2
    assert K_.workspace_status(ws, raise_on_error=False) == 0            # read and cleared


@pytest.mark.parametrize("K", [10_000, 1 << 16, (1 << 18) - 77])
def test_tiled_filter_one_launch_equals_step_by_step(K_, K):
    """k_ssm_persistent<TILED> (one rendezvous per step, everything in registers / LDS) == gjx_ssm_step +
    gjx_resample_indices_tiled issued from the host: same ancestors, hence bit-identical particles and weights."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem(T=24)
    for rng in (A.RNG_FLAT, A.RNG_JAX32):
        bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, rng_mode=rng, weights="tile_scaled")
        a = bf.run(core.key(7), s["y"])                    # (the kernel on the shared skeleton, k_pf_persistent, by default ...)
        os.environ["GJX_PF"] = "0"                         # ... and k_ssm_persistent<TILED>, the one-slot-per-lane kernel
        try:
            a0 = bf.run(core.key(7), s["y"])
        finally:
            del os.environ["GJX_PF"]
        np.testing.assert_array_equal(_np(a["x"]), _np(a0["x"]))
        np.testing.assert_array_equal(_np(a["logw"]), _np(a0["logw"]))
        b = bf.run(core.key(7), s["y"], step_by_step=True)
        assert not a["degenerate"]
        np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
        np.testing.assert_array_equal(_np(a["logw"]), _np(b["logw"]))
        np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-6)   # LSE finish order differs
        # the multi-launch fallback of the native loop (no co-resident grid): the same again
        os.environ["GJX_SSM_PERSISTENT"] = "0"
        try:
            c = bf.run(core.key(7), s["y"])
        finally:
            del os.environ["GJX_SSM_PERSISTENT"]
        np.testing.assert_array_equal(_np(a["x"]), _np(c["x"]))
        np.testing.assert_allclose(_np(a["increments"]), _np(c["increments"]), rtol=2e-6, atol=2e-6)
        # and the flag the host layer passes when it repeats a timed-out run (GJX_WEIGHTS_PLAIN_LAUNCHES: per call, no environment)
        import torch
        ys_d = torch.as_tensor(np.asarray(s["y"], np.float32)).cuda()
        for w in (A.WEIGHTS_TILE_SCALED, A.WEIGHTS_GLOBAL_MAX):
            one = K_.ssm_filter(bf.ssm.c_struct("cuda"), core.key(7), rng, ys_d, K, weights=w)
            x1, l1 = _np(one["x"]).copy(), _np(one["lse_steps"]).copy()
            pl = K_.ssm_filter(bf.ssm.c_struct("cuda"), core.key(7), rng, ys_d, K, weights=w | A.WEIGHTS_PLAIN_LAUNCHES)
            np.testing.assert_array_equal(x1, _np(pl["x"]))
            np.testing.assert_allclose(l1[:, 3], _np(pl["lse_steps"])[:, 3], rtol=2e-6, atol=2e-6)
            assert K_.workspace_status(pl["_status_ws"], raise_on_error=False) == 0


@pytest.mark.parametrize("K,force", [((1 << 18) + 1, False), (300_001, False), (1 << 19, False), (1 << 20, False), (1, True), (2049, True), (70_001, True)])
def test_any_size_one_launch_filter_equals_step_by_step(K_, K, force, monkeypatch):
    """k_pf_persistent (the one-launch filter beyond one slot per lane: several quantisation tiles per block, any K; what
    config 4's 2^19 particles per GPU run; GJX_PF=1 takes it at small sizes too) == the host-driven loop, bit for bit,
    on both stream layouts and with an observation matrix."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    if force:
        monkeypatch.setenv("GJX_PF", "1")
    for rng, dx in ((A.RNG_FLAT, 8), (A.RNG_JAX32, 4)):
        s = cf.ssm_problem(dx=dx, T=9)
        rs = np.random.default_rng(K % 97)
        H = None if rng == A.RNG_FLAT else (rs.standard_normal((3, dx)) / np.sqrt(dx)).astype(np.float32)
        y = s["y"] if H is None else (s["y"] @ H.T + 0.3 * rs.standard_normal((9, 3))).astype(np.float32)
        bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"], H=H), K, rng_mode=rng, weights="tile_scaled")
        a = bf.run(core.key(11), y)
        b = bf.run(core.key(11), y, step_by_step=True)
        assert not a["degenerate"]
        np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
        np.testing.assert_array_equal(_np(a["logw"]), _np(b["logw"]))
        np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("weights", ["tile_scaled", "global_max"])
def test_one_launch_filter_tiny_and_ragged_sizes(K_, weights):
    """One particle, one wave, one tile, one particle past a tile; two-step and three-step filters: the one-launch filter
    still equals the host-driven loop bit for bit."""
    import torch
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    for T in (2, 3, 7):
        s = cf.ssm_problem(T=T)
        for K in (1, 2, 63, 64, 65, 1023, 1024, 1025, 2049, 4097):
            bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, weights=weights)
            a = bf.run(core.key(3), s["y"])
            b = bf.run(core.key(3), s["y"], step_by_step=True)
            assert torch.equal(a["x"], b["x"]) and torch.equal(a["logw"], b["logw"]), (T, K)
            np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("weights", ["tile_scaled", "global_max"])
@pytest.mark.parametrize("dx,dy", [(2, 2), (4, 3), (8, 5), (16, 16), (16, 7)])
def test_one_launch_filter_other_shapes(K_, weights, dx, dy):
    """State dimensions 2 .. 16, with and without an observation matrix H (dy != dx): the one-launch filter (model
    constants and per-step values staged in LDS, split draws) equals the host-driven loop bit for bit, and its log-ML
    is that of the float64 Kalman filter within Monte-Carlo error."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem(dx=dx, T=20)
    rs = np.random.default_rng(dx * 100 + dy)
    H = None if dy == dx else (rs.standard_normal((dy, dx)) / np.sqrt(dx)).astype(np.float32)
    y = s["y"] if H is None else (s["y"] @ H.T + 0.3 * rs.standard_normal((20, dy))).astype(np.float32)
    K = 50_000
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"], H=H), K, weights=weights)
    a = bf.run(core.key(5), y)
    b = bf.run(core.key(5), y, step_by_step=True)
    np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
    np.testing.assert_array_equal(_np(a["logw"]), _np(b["logw"]))
    np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-5)
    exact, _, _ = cf.kalman_log_lik(s["A"], y, s["q"], s["r"], H=H)
    assert float(a["log_ml"]) == pytest.approx(exact, abs=0.5 + 0.02 * dx)


@pytest.mark.parametrize("weights", ["tile_scaled", "global_max"])
def test_collapsed_weights_one_launch_equals_step_by_step(K_, weights):
    """A very informative observation (r = 0.02): almost every tile is dead, the few live particles lie far apart, a
    block's source tiles are a long range of mostly empty tiles.  The one-launch filter still equals the host-driven
    loop bit for bit (and does not crawl through the empty tiles round by round)."""
    import time
    import torch
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem(T=10)
    for K in (1 << 16, (1 << 18) - 3):
        bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], 0.02), K, weights=weights)
        a = bf.run(core.key(11), s["y"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = bf.run(core.key(11), s["y"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        b = bf.run(core.key(11), s["y"], step_by_step=True)
        lw = _np(a["logw"])
        assert np.isfinite(lw).all() and (lw.max() - np.median(lw)) > 200.0          # collapsed indeed
        np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
        np.testing.assert_array_equal(lw, _np(b["logw"]))
        np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-5)
        assert dt < 0.02, dt                                                        # 10 steps: well under a millisecond each


def test_tiled_filter_close_to_global_max_filter(K_):
    """Same comb, same streams: the two weight schemes pick the same ancestor except where a threshold falls within the
    quantisation step of a particle boundary — after one resampling almost all particles agree bit for bit."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem(T=2)
    K = 1 << 16
    m = LinearGaussianSSM(s["A"], s["q"], s["r"])
    a = BootstrapFilter(m, K, weights="tile_scaled").run(core.key(3), s["y"])
    b = BootstrapFilter(m, K, weights="global_max").run(core.key(3), s["y"])
    same = (_np(a["x"]) == _np(b["x"])).all(axis=0)
    assert same.mean() > 0.995
    np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=1e-3, atol=1e-3)


def test_default_weights_are_the_tile_scaled_paths(K_):
    """BootstrapFilter() without a scheme runs the tile-scaled one-launch filter (also with resample-move), bit for bit what
    weights="tile_scaled" runs (the sharded side of the default: test_gpu_parity's one-rank RCCL worker)."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem(T=12)
    m = LinearGaussianSSM(s["A"], s["q"], s["r"])
    K = 1 << 15
    for rej in (None, dict(n_moves=1, scale=0.4)):
        a = BootstrapFilter(m, K, rejuvenate=rej).run(core.key(3), s["y"])
        b = BootstrapFilter(m, K, rejuvenate=rej, weights="tile_scaled").run(core.key(3), s["y"])
        c = BootstrapFilter(m, K, rejuvenate=rej, weights="global_max").run(core.key(3), s["y"])
        np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
        np.testing.assert_array_equal(_np(a["increments"]), _np(b["increments"]))
        assert not np.array_equal(_np(a["x"]), _np(c["x"]))


def test_tiled_bootstrap_filter_full_size(K_):
    """BASELINE config 3 (T=256, K=2^18) under the tile-scaled scheme: log-ML vs the float64 Kalman filter, rtol 1e-4."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem()
    exact, incs, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << 18, weights="tile_scaled")
    out = bf.run(core.key(1), s["y"])
    assert not out["degenerate"]
    assert float(out["log_ml"]) == pytest.approx(exact, rel=1e-4)
    np.testing.assert_allclose(_np(out["increments"]), incs, atol=0.2)


@pytest.mark.parametrize("K", [1, 777, 1024, 4096 + 3, 100_000, 1 << 20, (1 << 22) - 5])
def test_plain_launch_resample_gather_equals_the_three_launch_tiled_resampler(K_, K):
    """gjx_resample_gather_tiled (ONE plain launch: tile totals read, no block waits for another) gives the ancestors of
    gjx_resample_indices_tiled — itself compared with the oracle above — bit for bit, for every weight shape (collapsed
    weights take the re-scan loop, balanced ones the window of three tiles), and the children are the gathered rows."""
    import torch
    rs = np.random.default_rng(K + 1)
    rows = torch.as_tensor(rs.standard_normal((3, K)).astype(np.float32)).cuda()
    ws = K_.workspace(A.OP_RESAMPLE, K, "cuda")
    for name, lw in _weight_shapes(K, rs):
        lwd = torch.as_tensor(lw).cuda()
        for u in (0.0, 0.3718, 0.999999):
            want = K_.resample_indices_tiled(lwd, u, K)
            anc = torch.empty(K, dtype=torch.int32, device="cuda")
            out = K_.resample_gather_tiled(lwd, u, rows, anc=anc, ws=ws)
            np.testing.assert_array_equal(_np(anc), _np(want), err_msg=f"{name} u={u}")
            assert torch.equal(out, K_.gather_rows(rows, want)), name
        assert K_.workspace_status(ws, raise_on_error=False) == 0, name
    dead = torch.full((K,), float("-inf"), device="cuda")
    anc = torch.empty(K, dtype=torch.int32, device="cuda")
    K_.resample_gather_tiled(dead, 0.5, rows, anc=anc, ws=ws)
    assert (_np(anc) == np.arange(K)).all() and K_.workspace_status(ws, raise_on_error=False) == 2


@pytest.mark.parametrize("K", [777, 100_000, 1 << 20])
def test_plain_launch_resample_gather_planned_form(K_, K, monkeypatch):
    """the form for many tiles (prefix and shifts computed once by k_tiled_plan and read from memory; what K > 2^20 runs),
    forced at small sizes: the same ancestors"""
    import torch
    monkeypatch.setenv("GJX_TILED_PLANNED", "1")
    rs = np.random.default_rng(K + 2)
    rows = torch.as_tensor(rs.standard_normal((2, K)).astype(np.float32)).cuda()
    for name, lw in _weight_shapes(K, rs):
        lwd = torch.as_tensor(lw).cuda()
        want = K_.resample_indices_tiled(lwd, 0.3718, K)
        anc = torch.empty(K, dtype=torch.int32, device="cuda")
        out = K_.resample_gather_tiled(lwd, 0.3718, rows, anc=anc)
        np.testing.assert_array_equal(_np(anc), _np(want), err_msg=name)
        assert torch.equal(out, K_.gather_rows(rows, want)), name


@pytest.mark.parametrize("K", [1 << 20, 1 << 14, 3 * 1024])
def test_run_program_leaves_the_tile_totals_for_the_resampler(K_, K):
    """the hand-fused mixture kernel, called without an LSE record, leaves {S_b, e_b} of every 1024-particle tile beside its
    block partials: identical to what the resampler computes from the log-weights itself; resampling from them gives the
    same ancestors, the same children and the finished LSE record."""
    import torch
    from genjax_amd import workloads
    prog, _ = workloads.gmm_program()
    out = K_.run_program(prog, (0, 3), K, want_lse=False, want_tiles=True)
    part = out["_partials"]
    assert part.tiles > 0
    assert K_.run_program(prog, (0, 3), K, want_lse=False)["_partials"].tiles == 0          # only on request
    nt = K // 1024
    S = out["_ws"][part.tiles:part.tiles + 8 * nt].view(torch.int64)
    E = out["_ws"][part.tiles + 8 * nt:part.tiles + 12 * nt].view(torch.int32)
    _, q, e = K_.resample_indices_tiled(out["logw"], 0.25, K, want_q=True)
    np.testing.assert_array_equal(_np(E), _np(e))
    np.testing.assert_array_equal(_np(S), _np(q).view(np.uint32).astype(np.int64).reshape(nt, 1024).sum(axis=1))
    a1 = torch.empty(K, dtype=torch.int32, device="cuda")
    a2 = torch.empty(K, dtype=torch.int32, device="cuda")
    lse = torch.empty(4, device="cuda")
    r1 = K_.resample_gather_tiled(out["logw"], 0.25, out["choices"], partials=part.as_arg(), tiles=part.tiles, lse_out=lse, anc=a1)
    r2 = K_.resample_gather_tiled(out["logw"], 0.25, out["choices"], anc=a2)
    assert torch.equal(a1, a2) and torch.equal(r1, r2)
    assert torch.equal(a1, K_.resample_indices_tiled(out["logw"], 0.25, K))
    full = K_.run_program(prog, (0, 3), K)                    # the same run with its own LSE tail
    np.testing.assert_allclose(_np(lse)[2:], _np(full["lse"])[2:], rtol=1e-6, atol=1e-5)
    assert K_.run_program(prog, (0, 3), K + 4, want_lse=False, want_tiles=True)["_partials"].tiles == 0      # K % 1024 != 0: no tiles


def test_api_resample_with_tile_scaled_weights(K_):
    """inference.pf.resample(..., weights="tile_scaled"): one plain launch at a size the co-resident resampler cannot take
    in one launch (K = 2^21); ancestors of gjx_resample_indices_tiled, children gathered, the collection's LSE record
    finished by the same launch when the weights come from an ImportanceK run"""
    import torch
    import genjax_amd as genjax
    from genjax_amd import ChoiceMapBuilder as C
    from genjax_amd.inference import ImportanceK, Target
    from genjax_amd.inference import pf
    K = 1 << 21
    rs = np.random.default_rng(9)
    rows = torch.as_tensor(rs.standard_normal((3, K)).astype(np.float32)).cuda()
    lw = torch.as_tensor((rs.standard_normal(K) * 2.0).astype(np.float32)).cuda()
    key = core.key(12)
    out, anc = pf.resample(rows, lw, key, weights="tile_scaled", check=True)
    want = K_.resample_indices_tiled(lw, pf._unit_from_key(key), K)
    assert torch.equal(anc, want) and torch.equal(out, K_.gather_rows(rows, want))
    with pytest.raises(ValueError):
        pf.resample(rows, lw, key, weights="tile_scaled", n_out=K // 2)
    # without a scheme: tile-scaled above 2^20 (where the co-resident launch does not fit), global maximum up to there
    out_d, anc_d = pf.resample(rows, lw, key)
    assert torch.equal(anc_d, want) and torch.equal(out_d, out)
    Ks = 1 << 16
    out_s, anc_s = pf.resample(rows[:, :Ks].contiguous(), lw[:Ks].contiguous(), key)
    assert torch.equal(anc_s, K_.resample_indices(lw[:Ks].contiguous(), pf._unit_from_key(key), Ks, lse=K_.logsumexp(lw[:Ks].contiguous(), Ks)))

    @genjax.gen
    def model():
        x = genjax.normal(0.0, 1.0) @ "x"
        genjax.normal(x, 0.5) @ "y"

    pc = ImportanceK(Target(model, (), C["y"].set(0.7)), k_particles=1 << 16).run_smc(genjax.key(3))
    ch = pc.particles.choices
    new_rows, a2 = pf.resample(ch, pc.log_weights, core.key(4), collection=pc, weights="tile_scaled")
    assert torch.equal(a2, K_.resample_indices_tiled(pc.log_weights, pf._unit_from_key(core.key(4)), 1 << 16))
    lml = float(pc.get_log_marginal_likelihood_estimate())                      # finished by the resampling launch (or on demand)
    want_lml = float(torch.logsumexp(pc.log_weights.double(), 0) - np.log(1 << 16))
    assert abs(lml - want_lml) < 1e-4


def test_plain_launch_resampler_properties_at_full_size(K_):
    """size-independent properties of systematic resampling at K = 2^22 (the largest single-GPU collection of BASELINE.json's
    configs): ancestors are sorted, every particle's offspring count is within one of K w_i (float64 weights), uniform
    weights reproduce the collection (idempotence), and the children are the ancestors' rows"""
    import torch
    K = 1 << 22
    rs = np.random.default_rng(23)
    lw = (rs.standard_normal(K) * 1.5).astype(np.float32)
    rows = torch.as_tensor(rs.standard_normal((2, K)).astype(np.float32)).cuda()
    anc = torch.empty(K, dtype=torch.int32, device="cuda")
    out = K_.resample_gather_tiled(torch.as_tensor(lw).cuda(), 0.61803, rows, anc=anc)
    a = _np(anc).astype(np.int64)
    assert a.min() >= 0 and a.max() < K and (np.diff(a) >= 0).all()
    w = np.exp(lw.astype(np.float64) - lw.max())
    expect = K * w / w.sum()
    counts = np.bincount(a, minlength=K)
    # the fixed point truncates each weight by < 2^-29 of its tile's largest and each tile by < 2^-28 of the largest weight
    assert np.abs(counts - expect).max() < 1.0 + 2e-3 * expect.max()
    assert torch.equal(out, rows[:, anc.long()])
    flat = torch.zeros(K, device="cuda")
    K_.resample_gather_tiled(flat, 0.5, rows, anc=anc)
    assert torch.equal(anc, torch.arange(K, dtype=torch.int32, device="cuda"))
