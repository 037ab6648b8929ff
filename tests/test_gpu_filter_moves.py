"""Resample-move with the library's own move requests in the generic filter (genjax_amd/inference/filter_moves.py): HMC over the step's
latents (hmc.py:138-211) and Rejuvenate with an arbitrary proposal (rejuvenate.py:70-94), each with the caller-side accept
(tests/inference/test_requests.py:131-137) on the step-local target, in the step-by-step form of the filter.

  * without moves the step-by-step loop IS gjx_scan_filter: states, log-weights and ancestors bit for bit (both resamplers);
  * every move step-locally against the SAME composition over the oracle (oracle hmc / run_program on the device's rows before the
    move; an accept may go the other way only at a near tie);
  * whole runs: log-ML against the float64 Kalman value / the float64 stochastic-volatility fixture, and the carry diversified.
"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import genjax_amd as genjax               # noqa: E402
from genjax_amd import C, S               # noqa: E402
from genjax_amd import _abi as A          # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def _lgssm(dx, T):
    from genjax_amd import workloads
    scan, carry0, s = workloads.lgssm_scan(dx, T)
    return scan, carry0, s, np.asarray(s["y"], np.float32)


class OracleBackend:
    """the calls a move is made of (filter_moves.DeviceBackend) over the oracle: numpy arrays, oracle/cpu.py"""

    def run(self, prog, key, K, choices, want_site_scores=False):
        from oracle import cpu
        o = cpu.run_program(prog, key, K, choices=np.array(choices, np.float32), want_site_scores=want_site_scores)
        return o

    def hmc(self, prog, key, rows, eps, L):
        from oracle import cpu
        return cpu.hmc(prog, key, rows, eps, L, accept=True)

    def accept(self, log_alpha, key, rows_cur, rows_prop):
        from oracle import cpu
        new, acc, _ = cpu.mh_accept(log_alpha, key, rows_cur, rows_prop)
        rows_cur[...] = new
        return int(acc.sum())

    def empty(self, rows, K):
        return np.zeros((max(rows, 1), K), np.float32)

    def clone(self, x):
        return np.array(x, np.float32)

    def count(self, m):
        return int(np.sum(m))


@pytest.mark.parametrize("resampler", ["systematic", "multinomial"])
def test_the_step_by_step_form_without_moves_is_the_one_launch_filter(resampler):
    from genjax_amd.inference import BootstrapFilter
    from genjax_amd.inference.filter_moves import run_with_moves
    dx, T, K = 4, 12, (1 << 13) + 5
    scan, carry0, s, ys = _lgssm(dx, T)
    bf = BootstrapFilter(scan, K, resampler=resampler)
    ref = bf.run(genjax.key(7), C["y"].set(ys), (carry0, None), keep_ancestors=True)
    ref = {k: (_np(v).copy() if hasattr(v, "cpu") else v) for k, v in ref.items()}
    out = run_with_moves(bf, genjax.key(7), C["y"].set(ys), (carry0, None), [], keep_ancestors=True)
    np.testing.assert_array_equal(_np(out["ancestors"]), ref["ancestors"])
    np.testing.assert_array_equal(_np(out["logw"]), ref["logw"])
    np.testing.assert_array_equal(_np(bf.latent(out, "x")), _np(bf.latent(dict(ref, choices=__import__("torch").as_tensor(ref["choices"])), "x")))
    np.testing.assert_allclose(_np(out["lse_steps"])[:, 2:], ref["lse_steps"][:, 2:], rtol=2e-6, atol=2e-6)


def test_hmc_move_in_the_filter_step_locally_against_the_oracle():
    """HMC over the step's latent x (4 dims) behind every resampling: the rows the device filter moved (recorded before and after the
    move) against oracle.hmc on the same rows, key and program — alpha decides the accept, so a chain may differ only at the oracle's
    own near-tie margin —, then the propagate step against the oracle on the device's moved carry"""
    from genjax_amd.inference import BootstrapFilter, HMC
    from genjax_amd.inference.filter_moves import move_key, run_with_moves, target_program
    from genjax_amd.core import fold_in, split
    from oracle import cpu
    dx, T, K = 4, 4, 1 << 12
    scan, carry0, s, ys = _lgssm(dx, T)
    eps, L = 0.15, 4
    bf = BootstrapFilter(scan, K, moves=[HMC(S["x"], eps, L)])
    rec = []
    out = run_with_moves(bf, genjax.key(11), C["y"].set(ys), (carry0, None), bf.moves, keep_ancestors=True, record=rec)
    assert [r["t"] for r in rec] == [1, 2, 3] and all(a > 0.3 * K * 1 for a in [out["accepted"][0] / 3.0])
    progs = out["programs"]
    k = genjax.key(11)
    keys = []
    for t in range(T):
        k = fold_in(k, t)
        keys.append(split(k)[0])
    for r in rec:
        t = r["t"]
        pp = progs[t - 1]
        sel = [s_.addr for s_ in pp.site_list.sites if pp.modes.get(s_.addr, A.MODE_SAMPLE) == A.MODE_SAMPLE]
        prog_h = target_program(pp, sel)
        assert prog_h.n_slots == r["before"].shape[0]
        o = cpu.hmc(prog_h, move_key(keys[t], 0), _np(r["before"]), eps, L, accept=True)
        dev = _np(r["after"][0])
        bad = np.abs(dev - o["choices"]).max(axis=0) > 3e-4
        assert bad.mean() < 0.01 and (o["margin"][bad] < 2e-3).all(), (bad.mean(), o["margin"][bad][:8])
        np.testing.assert_allclose(dev[:, ~bad], o["choices"][:, ~bad], rtol=3e-4, atol=1e-4)
        assert 0.3 < o["accepted"].mean() < 0.999                      # a real move with a real accept rule
        # the inputs are untouched, the latents of accepted chains moved
        n_in = sum(s_.dim for s_ in pp.site_list.sites if pp.modes.get(s_.addr) == A.MODE_INPUT)
        np.testing.assert_array_equal(dev[:n_in], _np(r["before"])[:n_in])
        assert (np.abs(dev[n_in:] - _np(r["before"])[n_in:]).max(axis=0) > 0).mean() > 0.3
    # the last step against the oracle, fed the device's moved carry
    t = T - 1
    prog = progs[t]
    n_in_prev = sum(s_.dim for s_ in progs[t - 1].site_list.sites if progs[t - 1].modes.get(s_.addr) == A.MODE_INPUT)
    ch_in = np.zeros((prog.n_slots, K), np.float32)
    ch_in[:dx] = _np(rec[-1]["after"][-1])[n_in_prev:]
    ora = cpu.run_program(prog, keys[t], K, choices=ch_in)
    sl = prog.slot_of[("x", t)]
    np.testing.assert_allclose(_np(bf.latent(out, "x")), ora["choices"][sl:sl + dx], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(_np(out["logw"]), ora["weight"], rtol=2e-4, atol=2e-4)


def test_rejuvenate_proposal_move_in_the_filter_step_locally_against_the_oracle():
    """{"x": Rejuvenate(mv_normal_diag, x -> (0.8 x + 0.1, scale))}: an ASYMMETRIC proposal (the backward score differs from the
    forward one) — the device's composition against the same composition over the oracle on the rows before the move"""
    from genjax_amd.inference import BootstrapFilter, Rejuvenate
    from genjax_amd.inference.filter_moves import apply_proposal, move_key, proposal_programs, run_with_moves, target_program
    from genjax_amd.core import fold_in, split
    dx, T, K = 4, 3, 1 << 12
    scan, carry0, s, ys = _lgssm(dx, T)
    sc = np.full(dx, 0.3, np.float32)
    rej = Rejuvenate(genjax.mv_normal_diag, lambda chm: (0.8 * chm.get_value() + 0.1, sc))
    bf = BootstrapFilter(scan, K, moves=[{"x": rej}])
    rec = []
    out = run_with_moves(bf, genjax.key(13), C["y"].set(ys), (carry0, None), bf.moves, record=rec)
    progs = out["programs"]
    k = genjax.key(13)
    keys = []
    for t in range(T):
        k = fold_in(k, t)
        keys.append(split(k)[0])
    tot = 0
    for r in rec:
        t = r["t"]
        pp = progs[t - 1]
        tgt = target_program(pp)
        qs = proposal_programs(rej, dx, pp.rng_mode)
        before = _np(r["before"])
        R_o, na = apply_proposal(OracleBackend(), tgt, ("x", t - 1), qs, move_key(keys[t], 0), before.copy(), K)
        tot += na
        dev = _np(r["after"][0])
        bad = np.abs(dev - R_o).max(axis=0) > 3e-4
        assert bad.mean() < 0.01, bad.mean()                          # an accept decided the other way at a near tie
        np.testing.assert_allclose(dev[:, ~bad], R_o[:, ~bad], rtol=3e-4, atol=1e-4)
        assert 0.1 < na / K < 0.95
    assert abs(out["accepted"][0] - tot) <= 0.02 * K * len(rec)


def test_filter_with_hmc_and_proposal_moves_keeps_the_log_ml_and_diversifies_the_carry():
    """stochastic volatility, T = 256: behind a resampling the carry holds as many distinct values as there are distinct ancestors;
    with an HMC move (and a Rejuvenate move behind it) the copies are distinct again, and the log-ML estimate stays inside the spread
    of the ideal float64 filter (tests/golden/sv_pf_float64.json); the linear-Gaussian model against the float64 Kalman value"""
    from genjax_amd.inference import BootstrapFilter, HMC, Rejuvenate
    from oracle import closed_form as cf
    fx = json.load(open(os.path.join(HERE, "golden", "sv_pf_float64.json")))
    phi, sigma, ys = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)
    T, K = len(ys), 1 << 14

    @genjax.gen
    def step(x_prev, _):
        x = genjax.normal(phi * x_prev, sigma) @ "x"
        genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
        return x, None

    plain = BootstrapFilter(step.scan(n=T), K).run(genjax.key(3), C["y"].set(ys), (0.0, None))
    distinct_ancestors = len(np.unique(_np(plain["ancestors"])))
    mv = BootstrapFilter(step.scan(n=T), K, moves=[HMC(S["x"], 0.25, 3), {"x": Rejuvenate(genjax.normal, lambda chm: (chm.get_value(), 0.2))}])
    om = mv.run(genjax.key(3), C["y"].set(ys), (0.0, None))
    assert om["accepted"][0] > 0.5 * K * (T - 1) and om["accepted"][1] > 0.2 * K * (T - 1), om["accepted"]
    distinct_carry = len(np.unique(_np(om["choices"])[0]))                   # row 0 of the last step: its (moved) input
    assert distinct_carry > 0.95 * K > distinct_ancestors
    mean, std = fx["log_ml_mean"], fx["log_ml_std"]
    assert abs(float(om["log_ml"]) - mean) < 4.5 * std * np.sqrt((1 << 18) / K), (float(om["log_ml"]), mean)
    # linear-Gaussian, dx = 4, T = 48: an HMC move on the state, against Kalman
    from genjax_amd import workloads
    scan, carry0, s = workloads.lgssm_scan(4, 48)
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"], q0=float(s["q"]))
    o = BootstrapFilter(scan, 1 << 15, moves=[HMC(S["x"], 0.2, 3)]).run(genjax.key(5), C["y"].set(np.asarray(s["y"], np.float32)), (carry0, None))
    assert abs(float(o["log_ml"]) - exact) < 2e-3 * abs(exact), (float(o["log_ml"]), exact)


@pytest.mark.parametrize("resampler", ["systematic", "multinomial"])
def test_one_hmc_move_runs_inside_the_library_loop_and_equals_the_step_by_step_form(resampler):
    """moves=[HMC(...)] alone: gjx_scan_filter with gjx_filter_opts::hmc_targets — gather, ONE gjx_hmc launch, propagate per step, issued
    by the library without the host in between — gives the states, log-weights, ancestors and accept count of the step-by-step form
    (the same kernels under the same keys), bit for bit"""
    from genjax_amd.inference import BootstrapFilter, HMC
    from genjax_amd.inference.filter_moves import run_with_moves
    dx, T, K = 4, 9, (1 << 13) + 3
    scan, carry0, s, ys = _lgssm(dx, T)
    mv = [HMC(S["x"], 0.15, 3)]
    bf = BootstrapFilter(scan, K, moves=mv, resampler=resampler)
    a = bf.run(genjax.key(21), C["y"].set(ys), (carry0, None), keep_ancestors=True)
    assert a["info"]["form"] == A.FILTER_FORM_TWO_LAUNCH and a["accepted"][0] == a["accepted_total"] > 0.5 * K * (T - 1)
    a = {k: (_np(v).copy() if hasattr(v, "cpu") else v) for k, v in a.items()}
    b = run_with_moves(bf, genjax.key(21), C["y"].set(ys), (carry0, None), mv, keep_ancestors=True)
    np.testing.assert_array_equal(a["ancestors"], _np(b["ancestors"]))
    np.testing.assert_array_equal(a["logw"], _np(b["logw"]))
    np.testing.assert_array_equal(a["choices"], _np(b["choices"]))
    assert a["accepted"] == b["accepted"]
    np.testing.assert_allclose(a["lse_steps"][:, 2:], _np(b["lse_steps"])[:, 2:], rtol=2e-6, atol=2e-6)
    # a second run with other observations under the same structure: the targets are rebuilt with the tables
    ys2 = ys + 0.25
    a2 = bf.run(genjax.key(21), C["y"].set(ys2), (carry0, None))
    b2 = run_with_moves(bf, genjax.key(21), C["y"].set(ys2), (carry0, None), mv)
    np.testing.assert_array_equal(_np(a2["logw"]), _np(b2["logw"]))
    assert float(a2["log_ml"]) != float(a["log_ml"])


def test_mh_accept_against_the_oracle():
    """gjx_mh_accept — the caller-side accept of the reference's move requests as a device call: the same chains accept as in the
    oracle (a flip only where |log u - alpha| is at the last bit of a float32 log), accepted rows are the proposal's, the others
    untouched, NaN never accepts, the counter counts"""
    import torch
    from genjax_amd import kernels
    from oracle import cpu
    rs = np.random.default_rng(3)
    for K, rows in ((1, 1), (77, 3), (100_003, 5)):
        al = (rs.standard_normal(K) * 2.0 - 0.7).astype(np.float32)
        al[rs.random(K) < 0.01] = np.nan
        al[rs.random(K) < 0.01] = np.inf
        cur = rs.standard_normal((rows, K)).astype(np.float32)
        prop = rs.standard_normal((rows, K)).astype(np.float32)
        want, acc_o, margin = cpu.mh_accept(al, (5, K), cur, prop)
        c = torch.as_tensor(cur).cuda()
        tot = torch.zeros(1, dtype=torch.int64, device="cuda")
        acc = _np(kernels.mh_accept(torch.as_tensor(al).cuda(), (5, K), c, torch.as_tensor(prop).cuda(), total=tot))
        flip = acc != acc_o
        assert flip.sum() <= 2 and (margin[flip] < 1e-5).all(), (flip.sum(), margin[flip])
        got = _np(c)
        np.testing.assert_array_equal(got[:, ~flip], want[:, ~flip])
        np.testing.assert_array_equal(got[:, acc > 0.5], prop[:, acc > 0.5])
        np.testing.assert_array_equal(got[:, acc < 0.5], cur[:, acc < 0.5])
        assert int(tot.item()) == int(acc.sum()) and not acc[np.isnan(al)].any() and acc[np.isposinf(al)].all()
        if K > 1000:
            assert abs(acc.mean() - acc_o.mean()) < 1e-4 and 0.2 < acc.mean() < 0.6


def test_moves_argument_is_checked():
    from genjax_amd.inference import BootstrapFilter, HMC, Regenerate, Rejuvenate
    scan, carry0, s, ys = _lgssm(2, 3)
    with pytest.raises(ValueError, match="address a Rejuvenate"):
        BootstrapFilter(scan, 1024, moves=[Rejuvenate(genjax.normal, lambda chm: (chm.get_value(), 1.0))])
    with pytest.raises(NotImplementedError):
        BootstrapFilter(scan, 1024, moves=[Regenerate(S["x"])])
    with pytest.raises(ValueError, match="not both"):
        BootstrapFilter(scan, 1024, moves=[HMC(S["x"], 0.1, 2)], rejuvenate=dict(n_moves=1, scale=0.3))
    bf = BootstrapFilter(scan, 1024, moves=[HMC(S["nothing"], 0.1, 2)])
    with pytest.raises(ValueError, match="names no continuous latent"):
        bf.run(genjax.key(0), C["y"].set(ys), (carry0, None))
