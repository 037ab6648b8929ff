"""Particle-filter extensions of SURVEY.md §8 (f-2, f-3): ancestor history with lazy path reconstruction against the
float64 RTS smoother, and the resample-move rejuvenation kernel against its oracle restatement and the Kalman filter."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _np(t):
    return t.detach().cpu().numpy()


def test_ancestor_history_reconstructs_smoothed_trajectories():
    import torch
    from genjax_amd import core, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    from oracle import closed_form as cf
    s = workloads.ssm_problem(dx=4, T=12, r=2.0)
    K = 1 << 18
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, weights="global_max")
    out = bf.run(core.key(3), s["y"], keep_history=True)
    h = out["history"]
    assert len(h) == 12 and len(h.ancestors) == 11
    paths = h.paths()
    assert paths.shape == (12, 4, K)
    # a path is a chain of ancestors: its last point is the particle itself, every earlier point is a particle of that step
    assert torch.equal(paths[-1], h.states[-1])
    idx = torch.tensor([5, 77, K - 1], dtype=torch.int32, device="cuda")
    sub = h.paths(idx)
    cur = idx.long()
    for t in range(11, -1, -1):
        assert torch.equal(sub[t], h.states[t][:, cur])
        if t > 0:
            cur = h.ancestors[t - 1][cur].long()
    # smoothed means from the genealogy vs the float64 Rauch-Tung-Striebel smoother
    ms, Ps = cf.rts_smoother(s["A"], s["y"], s["q"], s["r"])
    got = _np(h.smoothed_means()).astype(np.float64)
    sd = np.sqrt(np.stack([np.diag(P) for P in Ps]))
    # path degeneracy: early steps are represented by few distinct ancestors, so the bound is in posterior std units
    assert np.abs(got - ms).max() < 0.15 * sd.max()
    assert np.abs(got[-4:] - ms[-4:]).max() < 0.03 * sd.max()
    # and the run itself is the ordinary filter
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    assert abs(float(out["log_ml"]) - exact) < 2e-3 * abs(exact)


@pytest.mark.parametrize("rng", [0, 1])
def test_move_step_matches_oracle(rng):
    import torch
    from genjax_amd import kernels, workloads
    from genjax_amd.inference.pf import LinearGaussianSSM
    from oracle import cpu
    s = workloads.ssm_problem(dx=4, T=4, r=1.0)
    K = 5000
    ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
    cs = ssm.c_struct("cuda")
    rs = np.random.default_rng(2)
    x_prev = rs.standard_normal((4, K)).astype(np.float32)
    m_prev = (x_prev * 0.8 + 0.1 * rs.standard_normal((4, K))).astype(np.float32)
    anc = np.sort(rs.integers(0, K, K)).astype(np.int32)
    ys = torch.as_tensor(s["y"]).cuda()
    for t in (1, 2):
        acc = torch.zeros(K, device="cuda")
        x, m, lw, lse = kernels.ssm_step_move(cs, (4, 5), rng, t, K, torch.as_tensor(x_prev).cuda(), torch.as_tensor(m_prev).cuda(),
                                              torch.as_tensor(anc).cuda(), ys[t - 1], ys[t], 3, 0.6, accepted=acc)
        xo, mo, lwo, acco, lseo, margin = cpu.ssm_step_move(s["A"], None, s["q"], s["r"], 1.0, (4, 5), rng, t, K, x_prev, m_prev, anc,
                                                           s["y"][t - 1], s["y"][t], 3, 0.6)
        ok = (np.abs(_np(x) - xo) <= 3e-4 + 3e-4 * np.abs(xo)).all(axis=0)
        assert (margin[~ok] < 3e-4).all() and (~ok).mean() < 0.01          # an accept decided by a near tie
        np.testing.assert_allclose(_np(m)[:, ok], mo[:, ok], rtol=3e-4, atol=3e-4)
        np.testing.assert_allclose(_np(lw)[ok], lwo[ok], rtol=3e-4, atol=3e-4)
        np.testing.assert_array_equal(_np(acc)[ok], acco[ok])
        assert 0.05 < float(acc.mean()) / 3 < 0.95


def test_resample_move_filter_keeps_the_estimate_and_restores_diversity():
    import torch
    from genjax_amd import core, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    from oracle import closed_form as cf
    s = workloads.ssm_problem(dx=4, T=24, r=0.7)         # informative observations: heavy resampling
    K = 1 << 16
    exact, _, means = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    plain = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, weights="global_max")
    moved = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, rejuvenate=dict(n_moves=2, scale=0.35), weights="global_max")
    a = plain.run(core.key(9), s["y"], keep_means=True, keep_history=True)
    b = moved.run(core.key(9), s["y"], keep_means=True, keep_history=True)
    for out in (a, b):
        assert abs(float(out["log_ml"]) - exact) < 3e-3 * abs(exact)
        np.testing.assert_allclose(_np(out["means"]), means, atol=0.05)
    assert 0.1 < moved.last_accept_rate < 0.9
    # the parents actually used at the last step: plain filter duplicates them, the moved filter does not
    pa = a["history"].states[-2][0][a["history"].ancestors[-1].long()]
    assert int(torch.unique(pa).numel()) < K * 0.9
    # the moved filter's history holds the parents that were actually propagated (the MOVED x_{t-1}, with their lineage):
    # along every reconstructed trajectory x_t - A x_{t-1} is the propagation noise (std a little under q: the surviving
    # trajectories are selected by the later observations) exactly as in the plain filter — with the un-moved parents it
    # would carry the move's displacement on top — and the smoothed means agree with the RTS smoother like the plain filter's
    ms, Ps = cf.rts_smoother(s["A"], s["y"], s["q"], s["r"])
    sd = np.sqrt(np.stack([np.diag(P) for P in Ps]))
    Am = torch.as_tensor(np.asarray(s["A"], np.float32)).cuda()
    stds = {}
    for out, name in ((a, "plain"), (b, "moved")):
        h = out["history"]
        paths = h.paths()                                            # [T][dx][K]
        resid = paths[1:] - torch.einsum("de,tek->tdk", Am, paths[:-1])
        stds[name] = float(resid.std())
        assert 0.9 * s["q"] < stds[name] < 1.02 * s["q"], stds
        got = _np(h.smoothed_means()).astype(np.float64)
        assert np.abs(got[-4:] - ms[-4:]).max() < 0.05 * sd.max(), name
        assert np.abs(got - ms).max() < 0.25 * sd.max(), name
    assert abs(stds["moved"] - stds["plain"]) < 0.02 * s["q"], stds


@pytest.mark.parametrize("rng", [0, 1])
@pytest.mark.parametrize("K,dx,useH", [(1 << 16, 8, False), ((1 << 15) - 70, 4, True), (1 << 19, 8, False)])
def test_move_filter_in_one_launch_equals_the_step_by_step_loop(rng, K, dx, useH):
    """Resample-move INSIDE the one-launch filter (gjx_ssm_filter_move, k_pf_persistent<.., MOVE>): the same draws and
    arithmetic as gjx_resample_indices_tiled + gjx_ssm_step_move issued per step from the host — bit-identical particles
    and weights, the same accept count — and the estimate still agrees with the Kalman filter."""
    import torch
    from genjax_amd import core, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    from oracle import closed_form as cf
    if K > (1 << 18) and rng == 1:
        pytest.skip("one size of the large grid is enough")
    s = workloads.ssm_problem(dx=dx, T=20, r=1.0)
    Hm = np.random.default_rng(3).standard_normal((3, dx)).astype(np.float32) if useH else None
    ys = s["y"] if not useH else (s["y"] @ Hm.T).astype(np.float32)
    m = LinearGaussianSSM(s["A"], s["q"], 1.0, Hm)
    bf = BootstrapFilter(m, K, rng_mode=rng, rejuvenate=dict(n_moves=2, scale=0.4), weights="tile_scaled")
    a = bf.run(core.key(21), ys)
    rate_a = bf.last_accept_rate
    assert a.get("history", None) is None and not a["degenerate"]
    b = bf.run(core.key(21), ys, step_by_step=True)
    rate_b = bf.last_accept_rate
    np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
    np.testing.assert_array_equal(_np(a["logw"]), _np(b["logw"]))
    np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-5)
    assert rate_a == pytest.approx(rate_b, rel=1e-6) and 0.05 < rate_a < 0.95
    if not useH:
        exact, _, _ = cf.kalman_log_lik(s["A"], ys, s["q"], 1.0)
        assert abs(float(a["log_ml"]) - exact) < 3e-3 * abs(exact)
