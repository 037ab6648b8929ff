"""GPU parity tests: every entry point of include/gjx.h, called through the C ABI, against the CPU oracle on the
same seeded inputs, the committed golden fixtures, and size-independent properties at BASELINE.json's full sizes.

Tolerances (north_star: log-ML rtol 1e-4, "matching posterior estimates within stated FP tolerance"):
  * integers — Threefry words, ancestor indices, fixed-point prefix sums, picked indices: BIT-EXACT
  * float32 per-particle values (samples, log-pdfs, scores, weights): rtol 2e-4 / atol 5e-5 — the device uses the
    hardware v_log/v_exp/v_rcp/v_sqrt approximations (≈1 ulp each) where the oracle uses libm
  * reductions (log-sum-exp): rtol 2e-6 against a float64 accumulation
  * discrete draws that hinge on a float comparison (categorical argmax, flip u<p, rejection samplers) may differ
    from the oracle at near-ties: at most 1e-3 of the particles, every other particle within the float tolerance
"""
import math
import os

import numpy as np
import pytest

import helpers as H
from genjax_amd import _abi as A
from genjax_amd import core
from genjax_amd.program import PackedProgram, Param, SiteList
from oracle import closed_form as cf

pytestmark = pytest.mark.gpu
RNGS = [A.RNG_FLAT, A.RNG_JAX32]
RT, AT = 2e-4, 5e-5


@pytest.fixture(scope="module")
def K_():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    from genjax_amd import kernels
    return kernels


def _np(t):
    return t.detach().cpu().numpy()


def _close_cols(a, b, rt=RT, at=AT):
    """per-particle mask: all rows within tolerance"""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    return (np.abs(a - b) <= at + rt * np.abs(b)).all(axis=0) | (np.isnan(a) & np.isnan(b)).all(axis=0)


NEAR_TIE = 3e-4   # relative margin of a discrete decision below which the device (v_exp / v_log / v_rcp, ~1 ulp, and
                  # float32 sums in another order) may land on the other side of the oracle's comparison


def assert_near_ties_only(bad, ora, what="", cap=0.005):
    """Particles that differ from the oracle must be NEAR TIES of a discrete decision the oracle took for them (which
    category, accept / reject, which side of a floor): the oracle reports each particle's smallest decision margin
    (oracle/gjx_oracle.c `decide`), and only margins below NEAR_TIE excuse a mismatch.  `cap` bounds their share."""
    bad = np.asarray(bad, bool)
    if not bad.any():
        return
    m = ora["margin"][bad]
    assert (m < NEAR_TIE).all(), (f"{what}: {int((m >= NEAR_TIE).sum())} of {int(bad.sum())} differing particles are NOT near ties "
                                   f"(largest decision margin among them {float(m.max()):.3g}, bound {NEAR_TIE})")
    assert bad.sum() <= max(1, int(cap * bad.size)), f"{what}: {int(bad.sum())}/{bad.size} near-tie particles (cap {cap})"


def assert_particles_match(gpu, ora, max_bad=None, what=""):
    ok = _close_cols(gpu["choices"], ora["choices"])
    for k in ("score", "weight", "logw"):
        ok &= _close_cols(gpu[k][None], ora[k][None])
    assert_near_ties_only(~ok, ora, what)
    return ok


def _run_both(K_, oracle, prog, key, K, **kw):
    g = K_.run_program(prog, key, K, **kw)
    o = oracle.run_program(prog, key, K, want_margin=True, **kw)
    gg = {k: _np(v) for k, v in g.items() if k in ("choices", "score", "weight", "logw", "lse", "site_scores") and v is not None}
    return gg, o


def test_library_is_the_hip_one(K_):
    from genjax_amd import _lib
    assert os.path.basename(_lib.LIB_PATH) == "libgjx_hip.so" and _lib.load().gjx_version() == A.ABI_VERSION


def test_threefry_bit_exact(K_, oracle, golden):
    for v in golden["threefry_kat"]:
        out = _np(K_.threefry2x32(tuple(v["key"]), 1, ctr_lo0=v["ctr"][1], ctr_hi=v["ctr"][0])).view(np.uint32)
        assert list(out[0]) == v["out"]
    n = 100_003
    out = _np(K_.threefry2x32((0xDEADBEEF, 0x12345678), n, ctr_lo0=0xFFFFFF00, ctr_hi=7)).view(np.uint32)
    for i in (0, 1, 255, 256, 257, n - 1):                 # crosses the 32-bit counter carry
        c = (7 << 32) + 0xFFFFFF00 + i
        assert tuple(out[i]) == oracle.threefry2x32(0xDEADBEEF, 0x12345678, c >> 32, c & 0xFFFFFFFF)


def test_logpdf_table_gpu(K_, golden):
    for row in golden["logpdf_table"]:
        prog = H.one_site(row["kind"], row["a"], row["b"], obs=row["x"], c=row.get("c"), d=row.get("d"))
        got = float(K_.run_program(prog, (0, 1), 1)["score"][0])
        if row["neg_inf"]:
            assert got == -math.inf, row
        else:
            assert got == pytest.approx(row["lp"], rel=1e-4, abs=1e-4), row
    for row in golden["dirichlet_table"]:
        prog = H.one_site("dirichlet", np.asarray(row["alpha"], np.float32), obs=np.asarray(row["x"], np.float32))
        assert float(K_.run_program(prog, (0, 1), 1)["score"][0]) == pytest.approx(row["lp"], rel=2e-4, abs=2e-4), row


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("observed", [(), ("n2", "mv2", "f1", "l0")])
def test_zoo_parity(K_, oracle, rng, observed):
    """every distribution kind and parameter form through the generic interpreter"""
    prog = H.zoo(rng, observed)
    for K in (1, 77, 3000):
        g, o = _run_both(K_, oracle, prog, (11, 22), K, want_site_scores=True)
        ok = assert_particles_match(g, o, max_bad=2e-3 if K > 1000 else 0, what=f"zoo K={K}")
        np.testing.assert_allclose(g["site_scores"][:, ok], o["site_scores"][:, ok], rtol=RT, atol=AT)
        if not observed:
            assert (g["weight"] == 0).all()


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("engine", ["interp", "gen"])
@pytest.mark.parametrize("observed", [(), ("n9", "ku", "mo", "ig")])
def test_zoo3_parity(K_, oracle, rng, observed, engine, monkeypatch):
    """nine more of the reference's TFP wrappers (round 6: chi, exp_gamma, exp_inverse_gamma, half_student_t, kumaraswamy, moyal,
    truncated_cauchy, double_sided_maxwell, inverse_gaussian; the oracle is held to scipy by tests/test_oracle.py) on the site
    interpreter and on the generated kernel: samples, scores, weights; a rejection sampler or a sign may part ways with the oracle only
    at a near tie"""
    monkeypatch.setenv("GJX_ENGINE", engine)
    prog = H.zoo3(rng, observed)
    assert K_.program_engine(prog) == (4 if engine == "gen" else 0)
    for K in (1, 77, 3000):
        g, o = _run_both(K_, oracle, prog, (41, 42), K, want_site_scores=True)
        ok = _close_cols(g["choices"], o["choices"], rt=5e-4, at=2e-4)
        for k in ("score", "weight", "logw"):
            ok &= _close_cols(g[k][None], o[k][None], rt=5e-4, at=5e-4)
        # (a draw at 1e-4 of a pole of the density — |z| -> 0 of the double-sided Maxwell, x -> 1 of kumaraswamy — is float32 ill-conditioned)
        ok |= ~np.isfinite(o["score"]) | (np.abs(o["site_scores"]) > 12.0).any(axis=0)
        assert_near_ties_only(~ok, o, f"zoo3 K={K}", cap=0.01)
        np.testing.assert_allclose(g["site_scores"][:, ok], o["site_scores"][:, ok], rtol=1e-3, atol=5e-4)
        if not observed:
            assert (g["weight"] == 0).all()
        else:
            assert np.isfinite(g["logw"]).all()


def test_zoo3_gradient_parity(K_, oracle):
    """analytic gradients of the nine kinds: device == oracle (the oracle's against finite differences: tests/test_oracle.py), through
    gjx_score_grad; and an HMC move over all of them runs on a generated kernel with the oracle's alpha"""
    import torch
    sl = H.zoo3().site_list
    cont = tuple(s.addr for s in sl.sites if s.kind not in A.NO_GRADIENT_KINDS)
    prog = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=cont)
    base = oracle.run_program(H.zoo3(), (9, 9), 512)["choices"].astype(np.float32)
    sg, gg = K_.score_grad(prog, torch.as_tensor(base).cuda())
    so, go = oracle.score_grad(prog, base)
    fin = np.isfinite(so) & (np.abs(go).max(axis=0) < 1e3)
    np.testing.assert_allclose(_np(sg)[fin], so[fin], rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(_np(gg)[:, fin], go[:, fin], rtol=3e-3, atol=3e-3)
    assert fin.mean() > 0.8
    assert K_.hmc_engine(prog) == 4
    sub_ = np.ascontiguousarray(base[:, fin])
    out = K_.hmc(prog, (1, 2), torch.as_tensor(sub_).cuda(), 0.002, 5, False, False)
    oo = oracle.hmc(prog, (1, 2), sub_, 0.002, 5)
    al, alo = _np(out["alpha"]), oo["alpha"]
    good = np.isfinite(alo) & (np.abs(alo) < 1.0)
    assert good.mean() > 0.7
    np.testing.assert_allclose(al[good], alo[good], rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("observed", [(), ("n9", "di", "po", "tn")])
def test_zoo2_parity(K_, oracle, rng, observed):
    """the wider distribution set (student_t, truncated_normal, poisson, geometric, dirichlet, gumbel, half_cauchy,
    inverse_gamma, weibull, logit_normal, chi2) through the generic interpreter: samples, scores, weights.
    Rejection samplers (gamma family, poisson above rate 10) and floor()-ed draws may part ways with the oracle only
    on a near-tie of the deciding comparison (assert_near_ties_only); the rest must match to the float tolerance."""
    prog = H.zoo2(rng, observed)
    for K in (1, 77, 3000):
        g, o = _run_both(K_, oracle, prog, (31, 32), K, want_site_scores=True)
        ok = _close_cols(g["choices"], o["choices"], rt=5e-4, at=2e-4)
        for k in ("score", "weight", "logw"):
            ok &= _close_cols(g[k][None], o[k][None], rt=5e-4, at=5e-4)
        assert_near_ties_only(~ok, o, f"zoo2 K={K}")
        np.testing.assert_allclose(g["site_scores"][:, ok], o["site_scores"][:, ok], rtol=1e-3, atol=5e-4)
        if not observed:
            assert (g["weight"] == 0).all()
        else:
            assert np.isfinite(g["logw"]).all()


def test_zoo2_gradient_parity(K_, oracle):
    """analytic gradients of the wider set: device == oracle (the oracle's are checked by finite differences in
    tests/test_oracle.py)"""
    import torch
    sl = H.zoo2().site_list
    cont = tuple(s.addr for s in sl.sites if s.kind not in A.NO_GRADIENT_KINDS)
    prog = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=cont)
    base = oracle.run_program(H.zoo2(), (9, 9), 512)["choices"].astype(np.float32)
    sg, gg = K_.score_grad(prog, torch.as_tensor(base).cuda())
    so, go = oracle.score_grad(prog, base)
    np.testing.assert_allclose(_np(sg), so, rtol=2e-4, atol=5e-4)
    big = np.abs(go) > 1e3                                    # near-singular points (x -> 0): compare relatively only
    np.testing.assert_allclose(_np(gg)[~big], go[~big], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(_np(gg)[big], go[big], rtol=1e-2)


def test_shape_parameter_gradient_parity_and_hmc(K_, oracle):
    """digamma-based gradients w.r.t. latent shape parameters: device == oracle, and an HMC move over the
    hyper-parameters of a gamma / beta / student-t hierarchy runs (finite, energy error small at small eps)"""
    import torch
    sl = H.shape_hierarchy()
    base = oracle.run_program(PackedProgram(sl), (3, 4), 2048)["choices"].astype(np.float32)
    prog = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=tuple(s.addr for s in sl.sites))
    sg, gg = K_.score_grad(prog, torch.as_tensor(base).cuda())
    so, go = oracle.score_grad(prog, base)
    assert np.isfinite(_np(gg)).all()
    np.testing.assert_allclose(_np(sg), so, rtol=3e-4, atol=1e-3)
    big = np.abs(go) > 1e3
    np.testing.assert_allclose(_np(gg)[~big], go[~big], rtol=3e-3, atol=3e-3)
    prog_h = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=("la", "lb"))
    out = K_.hmc(prog_h, (1, 2), torch.as_tensor(base).cuda(), 0.002, 20, False, False)
    al = _np(out["alpha"])
    assert np.isfinite(al).all() and np.abs(np.median(al)) < 0.05


@pytest.mark.parametrize("rng", RNGS)
def test_assess_and_reweight_parity(K_, oracle, rng):
    """all sites constrained per particle (assess / ChangeTarget, smc.py:378-391): logw = w + logw_in - sub"""
    sim = H.zoo(rng)
    K = 2000
    tr = oracle.run_program(sim, (1, 2), K)
    sl = sim.site_list
    prog = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, rng_mode=rng)
    rs = np.random.default_rng(0)
    lin, sub = rs.standard_normal(K).astype(np.float32), rs.standard_normal(K).astype(np.float32)
    import torch
    g = K_.run_program(prog, (0, 0), K, choices=torch.as_tensor(tr["choices"]).cuda(), logw_in=torch.as_tensor(lin).cuda(),
                       sub=torch.as_tensor(sub).cuda())
    o = oracle.run_program(prog, (0, 0), K, choices=tr["choices"], logw_in=lin, sub=sub)
    np.testing.assert_array_equal(_np(g["choices"]), tr["choices"])          # values untouched
    np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=RT, atol=AT)
    np.testing.assert_allclose(_np(g["logw"]), o["logw"], rtol=RT, atol=2e-4)
    np.testing.assert_allclose(_np(g["score"]), tr["score"], rtol=RT, atol=2e-4)   # score(simulate) == assess(choices)


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("shape", [(8, 16), (8, 1), (3, 2), (1, 4), (5, 64), (64, 8)])
def test_gmm_fused_generic_oracle(K_, oracle, rng, shape, monkeypatch):
    C, D = shape
    prog, g = H.gmm(D=D, C=C, rng=rng)
    for K in (1, 63, 1000, 4099):
        o = oracle.run_program(prog, (0, 1), K, want_margin=True)
        res = {}
        for force in ("1", "0"):
            monkeypatch.setenv("GJX_FORCE_GENERIC", force)
            assert K_.program_engine(prog) == (0 if force == "1" else 1)
            r = K_.run_program(prog, (0, 1), K)
            res[force] = {k: _np(v) for k, v in r.items() if k in ("choices", "score", "weight", "logw", "lse")}
            assert_particles_match(res[force], o, max_bad=1e-3 if K > 500 else 0, what=f"gmm {shape} K={K} generic={force}")
            np.testing.assert_allclose(res[force]["lse"][[0, 2, 3]], o["lse"][[0, 2, 3]], rtol=1e-5, atol=3e-4)
            assert res[force]["lse"][0] + math.log(res[force]["lse"][1]) == pytest.approx(res[force]["lse"][2], abs=1e-3)
        # fused and generic draw from the same counters: integer choices identical, floats within rounding
        same = res["0"]["choices"][0] == res["1"]["choices"][0]
        assert same.mean() >= 1 - 1e-3
        np.testing.assert_allclose(res["0"]["choices"][:, same], res["1"]["choices"][:, same], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rng", RNGS)
def test_sharding_and_determinism_bitwise(K_, rng, monkeypatch):
    prog, _ = H.gmm(rng=rng)
    K = 12_288
    full = K_.run_program(prog, (4, 2), K)
    again = K_.run_program(prog, (4, 2), K)
    for k in ("choices", "score", "logw"):
        np.testing.assert_array_equal(_np(full[k]), _np(again[k]))
    a = K_.run_program(prog, (4, 2), 4096, offset=0, K_total=K)
    b = K_.run_program(prog, (4, 2), 8192, offset=4096, K_total=K)
    np.testing.assert_array_equal(np.concatenate([_np(a["choices"]), _np(b["choices"])], axis=1), _np(full["choices"]))
    np.testing.assert_array_equal(np.concatenate([_np(a["logw"]), _np(b["logw"])]), _np(full["logw"]))
    # global LSE from the two shards' {max, sumexp} pairs
    import torch
    comb = K_.lse_combine(torch.stack([a["lse"][:2], b["lse"][:2]]).contiguous(), K)
    np.testing.assert_allclose(_np(comb), _np(full["lse"]), rtol=2e-6, atol=2e-6)
    # launch geometry must not change a single per-particle bit
    for ppt in ("1", "2"):
        monkeypatch.setenv("GJX_GMM_PPT", ppt)
        other = K_.run_program(prog, (4, 2), K)
        for k in ("choices", "score", "logw"):
            np.testing.assert_array_equal(_np(full[k]), _np(other[k]))
    # particle indices beyond 2^32 (FLAT folds the high word into the key; JAX32 uses a 64-bit counter)
    big = K_.run_program(prog, (4, 2), 300, offset=(1 << 32) + 5)
    monkeypatch.setenv("GJX_FORCE_GENERIC", "1")
    big_g = K_.run_program(prog, (4, 2), 300, offset=(1 << 32) + 5)
    np.testing.assert_array_equal(_np(big["choices"][0]), _np(big_g["choices"][0]))


def test_gmm_full_size(K_, golden):
    """BASELINE config 2 at K = 2^20: log-ML against the closed form (north_star rtol 1e-4) + invariants."""
    import torch
    prog, g = H.gmm()
    K = 1 << 20
    exact = golden["closed_form"]["gmm_c8_d16_seed0"]
    out = K_.run_program(prog, (0, 1), K)
    lml = float(out["lse"][3])
    assert lml == pytest.approx(exact, rel=1e-4)
    lw = out["logw"].double()
    assert float(out["lse"][2]) == pytest.approx(float(torch.logsumexp(lw, 0)), rel=2e-6)
    l2 = K_.logsumexp(out["logw"])
    np.testing.assert_allclose(_np(l2), _np(out["lse"]), rtol=2e-6, atol=2e-6)
    # weight is the observed site's log-density of the stored x: recompute it in float64 on the host
    x = out["choices"][1:].double()
    y = torch.as_tensor(g["y"], dtype=torch.float64, device=x.device)[:, None]
    want = (-0.5 * ((y - x) / 4.0) ** 2 - math.log(4.0) - 0.5 * math.log(2 * math.pi)).sum(0)
    assert float((out["weight"].double() - want).abs().max()) < 2e-4
    # prior part: score - weight = log p(z) + log N(x; mu_z, sigma_z)
    z = out["choices"][0].long()
    mu = torch.as_tensor(g["mu"], dtype=torch.float64, device=x.device)[z].t()
    lp = torch.log_softmax(torch.as_tensor(g["logits"], dtype=torch.float64, device=x.device), 0)[z]
    prior = lp + (-0.5 * (x - mu) ** 2 - 0.5 * math.log(2 * math.pi)).sum(0)
    assert float(((out["score"] - out["weight"]).double() - prior).abs().max()) < 3e-4
    # component frequencies follow softmax(logits)
    freq = torch.bincount(z, minlength=8).double() / K
    np.testing.assert_allclose(_np(freq), np.exp(g["logits"] - cf.logsumexp(g["logits"])), atol=4e-3)
    # posterior over z from the weights
    w = torch.exp(lw - out["lse"][2].double())
    pz = torch.zeros(8, dtype=torch.float64, device=x.device).index_add_(0, z, w)
    np.testing.assert_allclose(_np(pz), cf.gmm_posterior_z(**g), atol=5e-3)


def test_consumer_side_lse_finish(K_):
    """run_program(want_lse=False) leaves per-block {max, sumexp} pairs; gjx_weight_cumsum (mode 2) finishes the
    reduction in its prologue: same prefix sums bit for bit, same LSE record to rounding."""
    import torch
    for rng in RNGS:
        for K in (1000, 4099, 1 << 18):
            prog, _ = H.gmm(rng=rng)
            full = K_.run_program(prog, (3, 1), K)
            ws = K_.workspace(A.OP_RUN, K)
            part = K_.run_program(prog, (3, 1), K, want_lse=False, ws=ws)
            np.testing.assert_array_equal(_np(part["logw"]), _np(full["logw"]))
            n = K_.run_partials_count(prog, K)
            lse_out = torch.empty(4, device="cuda")
            cum2, bt2 = K_.weight_cumsum(part["logw"], partials=(ws, n), lse_out=lse_out)
            cum1, bt1 = K_.weight_cumsum(full["logw"], True, full["lse"])
            np.testing.assert_allclose(_np(lse_out), _np(full["lse"]), rtol=2e-6, atol=2e-6)
            assert float(lse_out[0]) == float(full["lse"][0])                     # the max is exact
            np.testing.assert_array_equal(_np(cum2), _np(cum1))
            np.testing.assert_array_equal(_np(bt2), _np(bt1))


def test_logsumexp_gpu(K_, oracle):
    import torch
    for K in (1, 2, 255, 256, 257, 100_003, 3_000_001):
        x = (np.random.default_rng(K).standard_normal(K) * 20).astype(np.float32)
        x[:: max(1, K // 7)] = -np.inf
        got = _np(K_.logsumexp(torch.as_tensor(x).cuda(), K_total=2 * K))
        want = oracle.logsumexp(x, 2 * K)
        if np.isneginf(want[2]):
            assert np.isneginf(got[2])
        else:
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    allinf = torch.full((1000,), -float("inf")).cuda()
    assert np.isneginf(_np(K_.logsumexp(allinf))[2])


@pytest.mark.parametrize("rng", RNGS)
def test_categorical_pick_parity(K_, oracle, rng):
    import torch
    for K, off in ((1, 0), (1000, 0), (300_001, 12_345)):
        lw = (np.random.default_rng(K).standard_normal(K) * 3).astype(np.float32)
        l4 = oracle.logsumexp(lw)
        for key in ((5, 6), (7, 8), (9, 10)):
            out = _np(K_.categorical_pick(torch.as_tensor(lw).cuda(), torch.as_tensor(l4).cuda(), key, rng, off))
            bv, bi = oracle.categorical_pick(lw, l4, key, rng, off)
            assert int(out[1]) == bi                                   # index: exact
            assert float(out.view(np.float32)[0]) == pytest.approx(bv, rel=1e-5, abs=1e-5)


def test_resampling_bit_exact(K_, oracle):
    import torch
    for K in (1, 5, 2048, 2049, 70_001, 1 << 20):
        rs = np.random.default_rng(K)
        lw = rs.standard_normal(K) * (4.0 if K % 2 else 1.0)
        w = np.exp(lw - lw.max()).astype(np.float32)
        wd = torch.as_tensor(w).cuda()
        cum, bt = K_.weight_cumsum(wd)
        cum_o, tot_o = oracle.weight_cumsum(w)
        np.testing.assert_array_equal(_np(cum).view(np.uint64), cum_o)
        assert _np(bt).tolist() == [0, tot_o]
        for N, u in ((K, 0.37), (max(1, K // 3), 0.0), (2 * K + 1, 0.999999)):
            anc = _np(K_.resample_systematic(cum, bt, u, N))
            np.testing.assert_array_equal(anc, oracle.resample_systematic(cum_o, u, N))
            assert anc.min() >= 0 and (np.diff(anc) >= 0).all()
        anc_m = _np(K_.resample_multinomial(cum, bt, (7, 8), K))
        np.testing.assert_array_equal(anc_m, oracle.resample_multinomial(cum_o, (7, 8), K))
        # windows of output slots and a second rank's view of the global weight line
        if K > 100:
            np.testing.assert_array_equal(_np(K_.resample_systematic(cum, bt, 0.5, K, out_begin=17, n_out=50)),
                                          oracle.resample_systematic(cum_o, 0.5, K)[17:67])
            bt2 = torch.tensor([tot_o // 3, 2 * tot_o], dtype=torch.int64).cuda()
            np.testing.assert_array_equal(_np(K_.resample_systematic(cum, bt2, 0.25, K)),
                                          oracle.resample_systematic(cum_o, 0.25, K, base=tot_o // 3, total_all=2 * tot_o))
        rows = rs.standard_normal((3, K)).astype(np.float32)
        anc_t = K_.resample_systematic(cum, bt, 0.37, K)
        np.testing.assert_array_equal(_np(K_.gather_rows(torch.as_tensor(rows).cuda(), anc_t)), oracle.gather_rows(rows, _np(anc_t)))
        fused, anc_f = K_.resample_gather_systematic(cum, bt, 0.37, K, torch.as_tensor(rows).cuda(), want_ancestors=True)
        np.testing.assert_array_equal(_np(anc_f), _np(anc_t))
        np.testing.assert_array_equal(_np(fused), oracle.gather_rows(rows, _np(anc_t)))
    # log-weight input: same as normalising on the device first, bit for bit
    lwd = torch.as_tensor(lw.astype(np.float32)).cuda()
    l4 = K_.logsumexp(lwd)
    cum_a, bt_a = K_.weight_cumsum(lwd, True, l4)
    assert int(bt_a[1].item()) == int(cum_a[-1].item()) and int(bt_a[0].item()) == 0
    anc_a = _np(K_.resample_systematic(cum_a, bt_a, 0.1, K))
    counts = np.bincount(anc_a, minlength=K)
    wn = np.exp(lw - cf.logsumexp(lw))
    assert np.abs(counts - K * wn).max() <= 1.01


def test_one_launch_resample_indices(K_, oracle):
    """gjx_resample_indices (prefix sums + ancestors in one co-resident kernel) == the three-launch path, bit for
    bit, for every tile configuration, repeated calls on one workspace (epoch tags) and degenerate weights."""
    import torch
    ws = None
    for K in (1, 300, 1024, 1025, 70_001, 1 << 20, (1 << 20) + 77, 3_000_000, 1 << 22, (1 << 22) + 5):
        rs = np.random.default_rng(K)
        lw = (rs.standard_normal(K) * (5.0 if K % 2 else 1.0)).astype(np.float32)
        lwd = torch.as_tensor(lw).cuda()
        l4 = K_.logsumexp(lwd)
        cum, bt = K_.weight_cumsum(lwd, True, l4)
        ws = K_.workspace(A.OP_RESAMPLE, K)
        for rep, (N, u) in enumerate(((K, 0.37), (K, 0.0), (max(1, K // 3), 0.999999), (2 * K + 1, 0.5))):
            want = K_.resample_systematic(cum, bt, u, N)
            cum2 = torch.empty_like(cum)
            bt2 = torch.empty_like(bt)
            got = K_.resample_indices(lwd, u, N, True, l4, ws=ws, cum=cum2, bt=bt2)
            np.testing.assert_array_equal(_np(got), _np(want))
            np.testing.assert_array_equal(_np(cum2), _np(cum))
            np.testing.assert_array_equal(_np(bt2), _np(bt))
        assert int(ws[40:44].view(torch.int32)) == 0                              # no spin timeout was flagged
    w = torch.zeros(5000).cuda()
    w[4321] = 2.0
    assert (K_.resample_indices(w, 0.7, 5000, is_log=False) == 4321).all()
    # linear weights against the oracle
    wl = np.random.default_rng(1).random(9999).astype(np.float32)
    cum_o, _ = oracle.weight_cumsum(wl)
    np.testing.assert_array_equal(_np(K_.resample_indices(torch.as_tensor(wl).cuda(), 0.123, 9999, is_log=False)),
                                  oracle.resample_systematic(cum_o, 0.123, 9999))


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("n,K", [(1, 1), (3, 777), (1000, 50), (2, 100_003), (64, 256)])
def test_trials_lse_and_pick_against_oracle(K_, oracle, rng, n, K):
    """gjx_trials_lse_pick (one block per trial) == the oracle's logsumexp and categorical_pick of every trial at its offset,
    and == the single-collection kernels; a dead trial reports -inf and its first particle."""
    import torch
    rs = np.random.default_rng(n * 7 + K)
    lw = (rs.standard_normal((n, K)) * 3.0).astype(np.float32)
    lw[rs.random((n, K)) < 0.05] = -np.inf
    if n > 2:
        lw[1] = -np.inf                                            # a dead trial
    off = 12_345
    d = torch.as_tensor(lw.reshape(-1)).cuda()
    lse, pick = K_.trials_lse_pick(d, n, K, (3, 9), rng, offset=off)
    lse, pick = _np(lse), _np(pick)
    for t in sorted(set([0, 1, n // 2, n - 1]) & set(range(n))):
        if not np.isfinite(lw[t]).any():
            assert lse[t][2] == -np.inf and pick[t] == off + t * K
            continue
        want = oracle.logsumexp(lw[t])
        np.testing.assert_allclose(lse[t], want, rtol=3e-6, atol=3e-6)
        _, idx = oracle.categorical_pick(lw[t], lse[t], (3, 9), rng, offset=off + t * K)
        assert pick[t] == idx
        one = K_.categorical_pick(torch.as_tensor(lw[t]).cuda(), torch.as_tensor(lse[t]).cuda(), (3, 9), rng, offset=off + t * K)
        assert int(one[1]) == pick[t]


def test_resampling_fuzz_against_oracle(K_, oracle):
    """Randomised shapes and weight patterns (zeros, a few giants, many ties, subnormal-scale weights, N != K) through
    the one-launch resampler and the gather, against the oracle's integer arithmetic: bit-exact every time."""
    import torch
    rs = np.random.default_rng(int(os.environ.get("GJX_FUZZ_SEED", "2024")))
    ws = {}
    for trial in range(int(os.environ.get("GJX_FUZZ_TRIALS", "70"))):
        K = int(rs.choice([1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097, 33_333, 131_072, 262_145]))
        kind = trial % 7
        if kind == 0:
            w = rs.random(K)
        elif kind == 1:
            w = rs.random(K) * (rs.random(K) < 0.1)                   # mostly zeros
            w[rs.integers(0, K)] += 1e-3
        elif kind == 2:
            w = np.full(K, 0.25)                                      # all ties
        elif kind == 3:
            w = rs.random(K) * 1e-6
            w[rs.integers(0, K, size=max(1, K // 500))] = 1.0         # a few giants
        elif kind == 4:
            w = np.exp(rs.standard_normal(K) * 6.0)
            w /= w.max()
        elif kind == 5:
            w = np.zeros(K); w[rs.integers(0, K)] = 1.0               # one survivor
        else:
            # a contiguous stretch of particles with 9..40 offspring EACH: more "heavy" particles in one tile than
            # threads in the block (found by the 4-process exchange test: the cooperative list used to hold 256)
            w = np.zeros(K); n_live = max(1, K // int(rs.integers(9, 40)))
            a0 = int(rs.integers(0, K - n_live + 1))
            w[a0:a0 + n_live] = 0.5 + rs.random(n_live)
        w = w.astype(np.float32)
        N = int(rs.choice([K, max(1, K // 2), 2 * K + 3, 1]))
        u = float(rs.choice([0.0, 0.5, 0.999999, rs.random()]))
        cum_o, tot = oracle.weight_cumsum(w)
        want = oracle.resample_systematic(cum_o, u, N)
        wd = torch.as_tensor(w).cuda()
        if K not in ws:
            ws[K] = K_.workspace(A.OP_RESAMPLE, K)
        got = K_.resample_indices(wd, u, N, is_log=False, ws=ws[K])
        np.testing.assert_array_equal(_np(got), want, err_msg=f"trial {trial} K={K} N={N} kind={kind} u={u}")
        rows = rs.standard_normal((3, K)).astype(np.float32)
        np.testing.assert_array_equal(_np(K_.gather_rows(torch.as_tensor(rows).cuda(), got)), rows[:, want])
        if N == K:      # resampling + gather in one launch: same ancestors, same children
            anc1 = torch.empty(K, dtype=torch.int32, device="cuda")
            ch = K_.resample_gather(wd, u, torch.as_tensor(rows).cuda(), is_log=False, anc=anc1, ws=ws[K], allow_fallback=False)
            np.testing.assert_array_equal(_np(anc1), want, err_msg=f"one-launch gather: trial {trial} K={K} kind={kind} u={u}")
            np.testing.assert_array_equal(_np(ch), rows[:, want])
        counts = np.bincount(want, minlength=K)
        assert counts[w == 0].sum() == 0                              # weightless particles never survive
    # the same stretch pattern where a lane holds 16 particles (more many-offspring particles per tile than the
    # cooperative list holds: the overflow is written by the owning lanes)
    K = 3_000_000
    w = np.zeros(K, np.float32)
    w[1_000_000:1_000_000 + K // 20] = (0.5 + rs.random(K // 20)).astype(np.float32)
    cum_o, _ = oracle.weight_cumsum(w)
    got = K_.resample_indices(torch.as_tensor(w).cuda(), 0.25, K, is_log=False)
    np.testing.assert_array_equal(_np(got), oracle.resample_systematic(cum_o, 0.25, K))


@pytest.mark.parametrize("K", [1, 255, 4096, 65_537, 1 << 18, (1 << 18) + 1024, 999_999, 1 << 20])
@pytest.mark.parametrize("kind", ["normal", "wide", "two_ends", "one", "stretch"])
def test_one_launch_resample_gather(K_, oracle, K, kind):
    """gjx_resample_gather (weights -> ancestors -> children in one co-resident launch, consumer side: binary search
    in the tile prefix and in the re-scanned source tiles) == oracle ancestors and rows, bit for bit, for log-weights
    with a finished LSE record, odd K, collapsed weights (all slots from one tile / from the two ends: every block's
    thresholds fall into one or two source tiles far away) and 17 rows (the bench shape)."""
    import torch
    rs = np.random.default_rng(K % 1000 + len(kind))
    lw = rs.standard_normal(K).astype(np.float32)
    if kind == "wide":
        lw *= 6.0
    if kind == "two_ends" and K > 8:
        lw[3:K - 3] = -np.inf
    if kind == "one":
        lw[:] = -np.inf; lw[K // 3] = 0.25
    if kind == "stretch":
        lw[: K // 2] -= 60.0; lw[K // 2 + K // 16:] -= 60.0
    rows = rs.standard_normal((17, K)).astype(np.float32)
    lwd, rd = torch.as_tensor(lw).cuda(), torch.as_tensor(rows).cuda()
    lse = K_.logsumexp(lwd, K)
    ws = K_.workspace(A.OP_RESAMPLE, K)
    for u in (0.0, 0.37, 0.999999):
        want = _np(K_.resample_indices(lwd, u, K, lse=lse))             # itself oracle-exact (test_one_launch_resample_indices)
        anc = torch.empty(K, dtype=torch.int32, device="cuda")
        ch = K_.resample_gather(lwd, u, rd, lse=lse, anc=anc, ws=ws, allow_fallback=False)
        np.testing.assert_array_equal(_np(anc), want)
        np.testing.assert_array_equal(_np(ch), rows[:, want])
        ch2 = K_.resample_gather(lwd, u, rd, lse=lse, ws=ws, allow_fallback=False)   # ancestors not asked for
        assert torch.equal(ch, ch2)
    assert K_.workspace_status(ws) == 0
    # plain (non-log) weights straight against the oracle
    w = np.exp(lw - np.max(lw)).astype(np.float32)
    cum_o, _ = oracle.weight_cumsum(w)
    want = oracle.resample_systematic(cum_o, 0.61, K)
    ch = K_.resample_gather(torch.as_tensor(w).cuda(), 0.61, rd, is_log=False, ws=ws, allow_fallback=False)
    np.testing.assert_array_equal(_np(ch), rows[:, want])
    # a dead collection keeps every particle (identity) and says so
    dead = torch.full((K,), -np.inf, device="cuda")
    ch = K_.resample_gather(dead, 0.5, rd, lse=K_.logsumexp(dead, K), ws=ws, allow_fallback=False)
    assert torch.equal(ch, rd) and K_.workspace_status(ws, raise_on_error=False) == 2


def test_degenerate_and_invalid_arguments(K_):
    import torch
    from genjax_amd._lib import GjxError
    w = torch.zeros(1000).cuda()
    w[123] = 1.0
    cum, bt = K_.weight_cumsum(w)
    assert (K_.resample_systematic(cum, bt, 0.9999, 1000) == 123).all()
    assert (K_.resample_multinomial(cum, bt, (1, 2), 1000) == 123).all()
    with pytest.raises(GjxError):
        K_.resample_systematic(cum, bt, 1.5, 1000)                      # u outside [0,1)
    with pytest.raises(GjxError):
        K_.logsumexp(torch.zeros(8).cuda(), ws=torch.empty(8, dtype=torch.uint8).cuda())   # workspace too small


@pytest.mark.parametrize("rng", RNGS)
def test_ssm_step_parity(K_, oracle, rng):
    import torch
    rs = np.random.default_rng(3)
    for dx, dy, useH in ((8, 8, False), (4, 3, True), (2, 2, False), (16, 5, True), (32, 7, True), (1, 1, False)):
        Am = (rs.standard_normal((dx, dx)) * 0.3).astype(np.float32)
        Hm = rs.standard_normal((dy, dx)).astype(np.float32) if useH else None
        y = rs.standard_normal((2, dy)).astype(np.float32)
        from genjax_amd.inference.pf import LinearGaussianSSM
        m = LinearGaussianSSM(Am, 0.5, 2.0, Hm, 1.0)
        cs = m.c_struct("cuda")
        K = 5001
        x0, lw0, l0 = K_.ssm_step(cs, (1, 2), rng, 0, K, None, None, torch.as_tensor(y[0]).cuda())
        xo, lwo, lo = oracle.ssm_step(Am, Hm, 0.5, 2.0, 1.0, (1, 2), rng, 0, K, None, None, y[0])
        np.testing.assert_allclose(_np(x0), xo, rtol=RT, atol=AT)
        np.testing.assert_allclose(_np(lw0), lwo, rtol=RT, atol=2e-4)
        np.testing.assert_allclose(_np(l0), lo, rtol=1e-4, atol=1e-4)
        anc = rs.integers(0, K, K).astype(np.int32)
        x1, lw1, l1 = K_.ssm_step(cs, (3, 4), rng, 1, K, x0, torch.as_tensor(anc).cuda(), torch.as_tensor(y[1]).cuda(), offset=77)
        x1o, lw1o, l1o = oracle.ssm_step(Am, Hm, 0.5, 2.0, 1.0, (3, 4), rng, 1, K, _np(x0), anc, y[1], offset=77)
        np.testing.assert_allclose(_np(x1), x1o, rtol=RT, atol=AT)
        np.testing.assert_allclose(_np(lw1), lw1o, rtol=RT, atol=2e-4)
        np.testing.assert_allclose(_np(l1), l1o, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("weights", ["tile_scaled", "global_max"])
def test_bootstrap_filter_full_size(K_, golden, weights):
    """BASELINE config 3: T=256, K=2^18, systematic resampling every step, on BOTH fixed-point schemes (tile_scaled is what the
    product runs by default).  rtol 1e-4 against the Kalman log-likelihood is not a single-run property at this size — an ideal
    float64 bootstrap filter has an rms relative error of 7.0e-5 over its seeds and 3 of its 16 seeds exceed 1e-4
    (tests/golden/ssm_pf_float64.json) — so a single run is held to that filter's single-run distribution (|z| < 4) and to 2.5e-4
    of the Kalman value; the 32-seed test below holds the rms to the ideal filter's rms and the bias to its standard error."""
    import json
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem()
    exact = golden["closed_form"]["ssm_dx8_T256_seed0"]
    kl, incs, means = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    assert kl == pytest.approx(exact, rel=1e-12)
    ref = np.asarray(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ssm_pf_float64.json")))["log_ml"])
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << 18, weights=weights)
    out = bf.run(core.key(1), s["y"], keep_means=True)
    z = (float(out["log_ml"]) - ref.mean()) / ref.std(ddof=1)
    print(f"{weights}: log-ML {float(out['log_ml']):.4f}, Kalman {exact:.4f}, rel {abs(float(out['log_ml']) - exact) / abs(exact):.3g}, z {z:.2f}")
    assert abs(z) < 4.0
    assert float(out["log_ml"]) == pytest.approx(exact, rel=2.5e-4)
    np.testing.assert_allclose(_np(out["increments"]), incs, atol=0.2)
    # filtered mean vs Kalman mean at full size (posterior sd ~1, min ESS ~0.05 K: sigma_MC ~0.01)
    # (sigma_MC is ~0.01 at the typical step and several times that at the few steps where an outlying observation leaves an ESS of
    # a few hundred: nearly every element within 0.08, all within 0.2, and the rms over all 2048 at the typical level)
    dm = np.abs(_np(out["means"]) - means)
    assert (dm < 0.08).mean() > 0.995 and dm.max() < 0.2, (float((dm < 0.08).mean()), float(dm.max()))
    assert float(np.sqrt(np.mean(dm ** 2))) < 0.015


@pytest.mark.parametrize("weights", ["global_max", "tile_scaled"])
def test_bootstrap_filter_seeds_vs_float64_filter(K_, weights):
    """Is the device filter's log-ML error Monte-Carlo spread or fixed-point quantisation?  32 seeds at config 3 full size
    on the device, against the spread of an IDEAL float64 bootstrap filter (NumPy) over 16 seeds at the same size
    (tests/golden/ssm_pf_float64.json, written by tests/golden/make_ssm_pf_float64.py).  The ideal filter's own rms
    relative error at K = 2^18 is 8e-5 (some of its seeds exceed 1e-4), so single runs at 9e-5 are Monte-Carlo spread;
    asserted: the device rms is within 1.5x the ideal filter's, no device run is beyond 4x that rms, and the mean
    error (bias) is within the spread of a mean of 32."""
    import json
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ssm_pf_float64.json")))
    s = cf.ssm_problem()
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"])
    assert exact == pytest.approx(fx["kalman_log_lik"], rel=1e-12)
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << 18, weights=weights)
    rel = np.array([(float(bf.run(core.key(100 + sd), s["y"])["log_ml"]) - exact) / abs(exact) for sd in range(32)])
    rms, ideal = float(np.sqrt(np.mean(rel ** 2))), fx["rms_rel_err"]
    print(f"{weights}: device rms rel-err {rms:.3g} (mean {rel.mean():.3g}, max |.| {np.abs(rel).max():.3g}, "
          f"{int((np.abs(rel) <= 1e-4).sum())}/32 within 1e-4); ideal float64 filter rms {ideal:.3g} "
          f"({sum(abs(r) <= 1e-4 for r in fx['rel_err'])}/{len(fx['rel_err'])} within 1e-4)")
    assert rms <= 1.5 * ideal
    assert np.abs(rel).max() <= 4.0 * ideal
    assert abs(rel.mean() - np.mean(fx["rel_err"])) <= 3.0 * ideal * math.sqrt(1 / 32 + 1 / len(fx["rel_err"]))


def test_native_filter_loop_equals_step_by_step(K_):
    """gjx_ssm_filter (C++ loop) issues exactly the launches of the host-driven loop: bit-identical results."""
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = cf.ssm_problem(T=40)
    for rng in RNGS:
        bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 10_000, rng_mode=rng, weights="global_max")
        a = bf.run(core.key(7), s["y"])
        b = bf.run(core.key(7), s["y"], step_by_step=True)
        np.testing.assert_allclose(_np(a["increments"]), _np(b["increments"]), rtol=2e-6, atol=2e-6)   # LSE finish order differs
        np.testing.assert_array_equal(_np(a["x"]), _np(b["x"]))
        np.testing.assert_array_equal(_np(a["logw"]), _np(b["logw"]))


def assert_accept_flips_explained(flip, alpha_dev, o):
    """Same uniform on both sides, accept iff log u < alpha: a chain can decide differently from the oracle only if
    log u lies between the device's alpha and the oracle's, i.e. the oracle's decision margin |log u - alpha_o| is at
    most |alpha_dev - alpha_o| (+ the rounding of the device's log)."""
    if flip.any():
        gap = np.abs(alpha_dev[flip] - o["alpha"][flip])
        assert (o["margin"][flip] <= gap + 1e-4 * (1.0 + np.abs(o["alpha"][flip]))).all(), (o["margin"][flip], gap)
        assert (gap <= 5e-2 + 2e-2 * np.abs(o["alpha"][flip])).all(), gap       # and the alphas themselves agree


@pytest.mark.parametrize("rng", RNGS)
def test_score_grad_and_hmc_parity(K_, oracle, rng):
    import torch
    prog, pr = H.logreg(N=64, P=4, rng=rng)
    rs = np.random.default_rng(2)
    n = 777
    ch = (rs.standard_normal((5, n)) * 0.3).astype(np.float32)
    sg, gg = K_.score_grad(prog, torch.as_tensor(ch).cuda())
    so, go = oracle.score_grad(prog, ch)
    np.testing.assert_allclose(_np(sg), so, rtol=RT, atol=2e-4)
    np.testing.assert_allclose(_np(gg), go, rtol=5e-4, atol=5e-4)
    for stale in (False, True):
        g = K_.hmc(prog, (1, 5), torch.as_tensor(ch).cuda(), 0.01, 25, stale, False, offset=3)
        o = oracle.hmc(prog, (1, 5), ch, 0.01, 25, stale, False, offset=3)
        np.testing.assert_allclose(_np(g["choices"]), o["choices"], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=5e-3, atol=5e-3)
        np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=1e-3, atol=5e-3)
        assert (_np(g["accepted"]) == 1).all()
    # fused MH rule: a step size that rejects a fair share of the chains
    g = K_.hmc(prog, (1, 5), torch.as_tensor(ch).cuda(), 0.2, 20, False, True, offset=3)
    o = oracle.hmc(prog, (1, 5), ch, 0.2, 20, False, True, offset=3)
    acc_g, acc_o = _np(g["accepted"]), o["accepted"]
    # a chain may decide differently from the oracle only if log u lies between the two alphas: no blanket allowance
    flip = acc_g != acc_o
    assert_accept_flips_explained(flip, _np(g["alpha"]), o)
    assert flip.mean() < 0.03
    rej = acc_g == 0
    assert 0.05 * n < rej.sum() < 0.6 * n
    np.testing.assert_array_equal(_np(g["choices"])[:, rej], ch[:, rej])          # rejected chains are restored bit for bit
    s_old, _ = K_.score_grad(prog, torch.as_tensor(ch).cuda())
    np.testing.assert_allclose(_np(g["score"])[rej], _np(s_old)[rej], rtol=1e-6, atol=1e-4)
    both = (acc_g == 1) & (acc_o == 1) & (np.abs(o["alpha"]) < 0.5)
    np.testing.assert_allclose(_np(g["choices"])[:, both], o["choices"][:, both], rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("shape", [(64, 4), (200, 8), (1024, 16), (1000, 16), (33, 16), (1024, 16, "GJX_HMC_NO_MFMA2"), (33, 16, "GJX_HMC_NO_MFMA2"),
                                   (1024, 16, "GJX_HMC_NO_BIN"), (1000, 16, "GJX_HMC_NO_BIN")])
def test_hmc_logreg_fused_vs_generic_vs_oracle(K_, oracle, rng, shape, monkeypatch):
    """BASELINE config 5 shape: the fused kernel and the site interpreter run the same program, same streams.  P = 16 runs on
    the matrix cores — by default the two-group kernel for 0/1 observations without a bias (k_hmc_logreg_mfma2); the variables
    step down to the one-group sign-folded kernel and to the general y - sigmoid kernel."""
    import torch
    N, P = shape[:2]
    if len(shape) > 2:
        monkeypatch.setenv(shape[2], "1")
        if shape[2] == "GJX_HMC_NO_BIN":
            monkeypatch.setenv("GJX_HMC_NO_MFMA2", "1")
    prog, pr = H.logreg(N=N, P=P, rng=rng)
    assert K_.hmc_engine(prog) in (2, 3)
    rs = np.random.default_rng(5)
    n = 300
    ch = (rs.standard_normal((P + 1, n)) * 0.2).astype(np.float32)
    eps = 0.01 if N < 1000 else 0.004
    for stale, accept, L in ((False, False, 20), (True, False, 7), (False, True, 20)):
        e = eps * (6 if accept else 1)
        f = K_.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), e, L, stale, accept, offset=11)
        monkeypatch.setenv("GJX_FORCE_GENERIC", "1")
        assert K_.hmc_engine(prog) == 0
        g = K_.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), e, L, stale, accept, offset=11)
        monkeypatch.delenv("GJX_FORCE_GENERIC")
        o = oracle.hmc(prog, (2, 9), ch, e, L, stale, accept, offset=11)
        same = (_np(f["accepted"]) == _np(g["accepted"])) & (_np(f["accepted"]) == o["accepted"])
        # decisions differ from the oracle's only where log u lies between the two alphas
        assert_accept_flips_explained(_np(f["accepted"]) != o["accepted"], _np(f["alpha"]), o)
        assert_accept_flips_explained(_np(g["accepted"]) != o["accepted"], _np(g["alpha"]), o)
        tol = dict(rtol=3e-3, atol=3e-3)
        np.testing.assert_allclose(_np(f["choices"])[:, same], _np(g["choices"])[:, same], **tol)
        np.testing.assert_allclose(_np(f["choices"])[:, same], o["choices"][:, same], **tol)
        np.testing.assert_allclose(_np(f["alpha"])[same], _np(g["alpha"])[same], rtol=1e-2, atol=2e-2)
        np.testing.assert_allclose(_np(f["alpha"])[same], o["alpha"][same], rtol=1e-2, atol=2e-2)
        np.testing.assert_allclose(_np(f["score"])[same], o["score"][same], rtol=1e-4, atol=2e-2)
        if accept:
            rej = _np(f["accepted"]) == 0
            np.testing.assert_array_equal(_np(f["choices"])[:, rej], ch[:, rej])


def _logreg_leapfrog_f64(pr, q0, p0, eps, L):
    """The config-5 trajectory in float64 (NumPy; closed-form gradient of log_tau ~ N(0,1), beta ~ N(0, e^log_tau),
    y ~ bernoulli_logits(X beta)) from the SAME initial state and momenta: hmc.py:170-203 without the stale carry.
    -> (q_L [17, n], alpha [n])."""
    X, y = pr["X"].astype(np.float64), pr["y"].astype(np.float64)[:, None]
    P = X.shape[1]

    def lpg(q):
        lt, b = q[0], q[1:]
        s2, bb, z = np.exp(-2.0 * lt), (b * b).sum(0), X @ b
        lp = -0.5 * lt * lt - P * lt - 0.5 * bb * s2 + (y * z - np.logaddexp(0.0, z)).sum(0)
        g = np.empty_like(q)
        g[0] = -lt - P + bb * s2
        g[1:] = -b * s2 + X.T @ (y - 1.0 / (1.0 + np.exp(-z)))
        return lp, g

    q, p = q0.astype(np.float64), p0.astype(np.float64)
    eps = float(np.float32(eps))
    lp0, g = lpg(q)
    k0 = -0.5 * (p * p).sum(0)
    for _ in range(L):
        p += 0.5 * eps * g
        q += eps * p
        lp, g = lpg(q)
        p += 0.5 * eps * g
    return q, lp - lp0 - 0.5 * (p * p).sum(0) - k0


def test_hmc_logreg_long_trajectory_energy(K_, oracle):
    """config 5 integrator settings (eps 0.01, L 1000, N 1024, P 16).  (1) 256 chains of the fused kernel against the
    SAME trajectories integrated in float64 (same initial states, same momenta), with the tolerance DERIVED from how far
    the float32 oracle itself drifts from float64 over the 1000 steps — no flat number; (2) the energy error alpha is
    the leapfrog's O(eps^2) discretisation error: quartering eps at fixed trajectory length shrinks it ~16x."""
    import torch
    prog, pr = H.logreg(N=1024, P=16)
    rs = np.random.default_rng(6)
    n, nc = 2048, 256
    ch0 = (rs.standard_normal((17, n)) * 0.1).astype(np.float32)
    out = K_.hmc(prog, (3, 3), torch.as_tensor(ch0).cuda(), 0.01, 1000, False, False)
    al = _np(out["alpha"])
    assert np.isfinite(al).all() and np.abs(al).max() < 1.5
    s1, _ = K_.score_grad(prog, out["choices"])
    np.testing.assert_allclose(_np(out["score"]), _np(s1), rtol=1e-4, atol=5e-2)
    assert float((out["choices"].cpu() - torch.as_tensor(ch0)).abs().mean()) > 0.05        # the chains moved
    o = oracle.hmc(prog, (3, 3), ch0[:, :nc].copy(), 0.01, 1000, False, False, want_momenta=True)   # chains 0..255 (same global indices)
    q64, al64 = _logreg_leapfrog_f64(pr, ch0[:, :nc], o["momenta"][:17], 0.01, 1000)
    d_o = np.abs(o["choices"] - q64).max(axis=0)                 # per chain: float32 oracle vs float64
    d_g = np.abs(_np(out["choices"])[:, :nc] - q64).max(axis=0)  # per chain: device vs float64
    a_o, a_g = np.abs(o["alpha"] - al64), np.abs(al[:nc] - al64)
    rms = lambda v: float(np.sqrt(np.mean(np.square(v))))
    print(f"L=1000 drift from float64, 256 chains: oracle f32 rms {rms(d_o):.3g} median {np.median(d_o):.3g} max {d_o.max():.3g}; "
          f"device rms {rms(d_g):.3g} median {np.median(d_g):.3g} max {d_g.max():.3g}; "
          f"alpha: oracle rms {rms(a_o):.3g} max {a_o.max():.3g}, device rms {rms(a_g):.3g} max {a_g.max():.3g}")
    # the device (MFMA contraction order, v_exp/v_rcp) may drift as far as the float32 oracle does, times a small factor
    assert np.median(d_g) <= 3.0 * np.median(d_o)
    assert rms(d_g) <= 3.0 * rms(d_o)
    assert d_g.max() <= 4.0 * d_o.max()
    assert rms(a_g) <= 3.0 * rms(a_o) and a_g.max() <= 4.0 * a_o.max()
    fine = K_.hmc(prog, (3, 3), torch.as_tensor(ch0).cuda(), 0.0025, 4000, False, False)
    ratio = np.abs(al).mean() / np.abs(_np(fine["alpha"])).mean()
    assert 8.0 < ratio < 32.0, ratio                                               # second-order integrator


def test_hmc_logreg_posterior_mean_vs_float64_long_run(K_):
    """posterior mean of (log_tau, beta) of the config-5 model from the device chains against a LONG
    float64 run (tests/golden/logreg_posterior.json, written by tests/golden/make_logreg_posterior.py: an independent NumPy
    sampler), tolerance 3 sigma_MC per coordinate with sigma_MC^2 = posterior variance / chains + (the fixture's own
    standard error)^2.  2^14 chains start far from the posterior (N(0, 0.1^2) around zero; the posterior means reach +-2)
    and take 24 moves of the fused kernel at eps = 0.01 with the Metropolis accept.  The number of leapfrog steps varies
    around the config's 1000 from move to move: with ONE trajectory length every move contracts a near-Gaussian mode of
    frequency w by the same |cos(w tau)|, and the modes for which that is ~1 never forget their start (measured: z-scores
    in the hundreds after 4 moves of exactly 1000 steps)."""
    import json
    import torch
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "logreg_posterior.json")))
    mean, sd, se = (np.array(fx[k]) for k in ("mean", "sd", "mc_se"))
    prog, pr = H.logreg(N=1024, P=16)
    n = 1 << 14
    rs = np.random.default_rng(11)
    ch = torch.as_tensor((rs.standard_normal((17, n)) * 0.1).astype(np.float32)).cuda()
    Ls = [1000, 620, 880, 1140, 760, 1000, 690, 930, 1210, 840, 1000, 570, 1090, 730, 1000, 660, 1170, 890, 1000, 780, 1000, 620, 880, 1140]
    acc = []
    for mv, L in enumerate(Ls):
        out = K_.hmc(prog, (5, 100 + mv), ch, 0.01, L, False, True)
        ch = out["choices"]
        acc.append(float(out["accepted"].mean()))
    got = _np(ch).astype(np.float64)
    sig = np.sqrt(sd ** 2 / n + se ** 2)
    z = (got.mean(axis=1) - mean) / sig
    print("accept rate min", round(min(acc), 3), "z-scores", np.round(z, 2))
    assert min(acc) > 0.9
    # family-wise bound over the 17 coordinates (Bonferroni): P(|z| > 4.0) = 6.3e-5 per coordinate, 1.1e-3 for the maximum of 17 —
    # the test is a 4 sigma test of the worst coordinate, the fixture's Monte-Carlo error of the long run included in sig
    assert np.abs(z).max() < 4.0, z
    np.testing.assert_allclose(got.std(axis=1), sd, rtol=0.04)


def test_hmc_all_kinds_gradient(K_, oracle):
    import torch
    sl = SiteList()
    sl.add("a", A.NORMAL, [0.3, 1.2])
    sl.add("h", A.HALF_NORMAL, [Param.value("a", xf=A.XF_SOFTPLUS)])
    sl.add("l", A.LAPLACE, [Param.value("a"), Param.value("h", xf=A.XF_EXP)])
    sl.add("c", A.CAUCHY, [Param.affine(np.array([[0.5]], np.float32), "l", bias=0.1), 1.3])
    sl.add("ln", A.LOG_NORMAL, [Param.value("c", xf=A.XF_SIGMOID), 0.7])
    sl.add("e", A.EXPONENTIAL, [Param.value("ln")])
    sl.add("g", A.GAMMA, [2.0, Param.value("e", xf=A.XF_SOFTPLUS)])
    sl.add("b", A.BETA, [2.0, 3.0])
    sl.add("y", A.BERNOULLI_LOGITS, [Param.value("b")])
    prog = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=tuple(s.addr for s in sl.sites if s.addr != "y"))
    vals = np.array([[0.4], [0.8], [0.1], [0.9], [1.3], [0.6], [1.7], [0.35], [1.0]], np.float32)
    sg, gg = K_.score_grad(prog, torch.as_tensor(vals).cuda())
    so, go = oracle.score_grad(prog, vals)
    np.testing.assert_allclose(_np(sg), so, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_np(gg), go, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("spread", ["mild", "wide", "first", "last"])
def test_sharded_plan_and_resample_virtual_ranks(K_, oracle, G, spread):
    """The sharded resampling kernels (plan, planned expansion, kept-children gather, pack/unpack), driven for G
    virtual ranks in one process: plan == host twin bit for bit, and the assembled result == the unsharded one."""
    import torch
    from genjax_amd import distributed as D
    K, R, u = 10_007, 3, 0.618
    rs = np.random.default_rng(G)
    lw = (rs.standard_normal(K) * (5.0 if spread == "wide" else 1.0)).astype(np.float32)
    if spread == "first":
        lw[3:] -= 80.0
    if spread == "last":
        lw[:-2] -= 80.0
    rows = rs.standard_normal((R, K)).astype(np.float32)
    w = np.exp(lw - lw.max()).astype(np.float32)       # linear weights: the fixed-point values are bit-exact vs the oracle
    cum_o, _ = oracle.weight_cumsum(w)
    want = oracle.gather_rows(rows, oracle.resample_systematic(cum_o, u, K))
    shards = [D.shard(K, r, G) for r in range(G)]
    cums, tots = [], []
    for off, k in shards:
        cum, bt = K_.weight_cumsum(torch.as_tensor(w[off:off + k]).cuda())
        cums.append(cum)
        tots.append(int(_np(bt)[1]))
    assert sum(tots) == int(cum_o[-1])
    totals = torch.tensor(tots, dtype=torch.int64).cuda()
    got = np.full((R, K), np.nan, np.float32)
    for r, (off, k) in enumerate(shards):
        plan = K_.ShardPlan("cuda").build(totals, r, u, K)
        p = plan.wait()
        hp = D.HostPlan(tots, r, u, K)
        for f in ("base", "total", "slot0", "n_valid", "own_lo", "own_n", "keep_lo", "keep_hi", "status"):
            assert int(getattr(p, f)) == int(getattr(hp, f)), f
        assert list(p.bounds[:G + 1]) == hp.bounds
        src = torch.as_tensor(rows[:, off:off + k].copy()).cuda()
        anc, kept = K_.shard_resample(cums[r], plan, u, K, src, hp.own_n)
        a = _np(anc)[:hp.n_valid]
        np.testing.assert_array_equal(a, oracle.resample_systematic(_np(cums[r]).view(np.uint64), u, K, base=hp.base,
                                                                    total_all=hp.total, out_begin=hp.slot0, n_out=hp.n_valid))
        lo, hi = hp.keep_lo, hp.keep_hi
        np.testing.assert_array_equal(_np(kept)[:, lo - hp.own_lo: hi - hp.own_lo], want[:, lo:hi])
        # everything this rank produces, as [n][R] messages and back
        if hp.n_valid:
            msg = K_.pack_rows(src, anc[:hp.n_valid])
            np.testing.assert_array_equal(_np(msg), rows[:, off:off + k][:, a].T)
            back = torch.full((R, hp.n_valid + 2), float("nan"), device="cuda")
            K_.unpack_rows(msg, back[:, 1:1 + hp.n_valid])
            got[:, hp.slot0: hp.slot0 + hp.n_valid] = _np(back)[:, 1:1 + hp.n_valid]
            assert np.isnan(_np(back)[:, 0]).all() and np.isnan(_np(back)[:, -1]).all()
            # the one-launch forms the exchange uses: surplus = run minus the kept window
            n_pre = max(0, min(hp.slot0 + hp.n_valid, hp.own_lo) - hp.slot0)
            n_suf = max(0, hp.slot0 + hp.n_valid - max(hp.slot0, hp.own_lo + hp.own_n))
            m2 = K_.shard_pack(src, anc, hp.n_valid, n_pre, n_suf)
            np.testing.assert_array_equal(_np(m2), np.concatenate([_np(msg)[:n_pre], _np(msg)[hp.n_valid - n_suf:]]))
            if n_pre + n_suf and n_pre + n_suf <= hp.own_n:
                d2 = torch.full((R, hp.own_n), float("nan"), device="cuda")
                K_.shard_unpack(m2, n_pre, n_suf, d2)
                np.testing.assert_array_equal(_np(d2)[:, :n_pre], _np(m2)[:n_pre].T)
                np.testing.assert_array_equal(_np(d2)[:, hp.own_n - n_suf:], _np(m2)[n_pre:].T)
                assert np.isnan(_np(d2)[:, n_pre: hp.own_n - n_suf]).all()
    np.testing.assert_array_equal(got, want)


def _two_rank_worker(rank, world, port, K, R, spread, q):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GJX_DIST_BACKEND="gloo")
    from genjax_amd import distributed as D
    from genjax_amd import kernels
    D.init_from_env("gloo")
    rs = np.random.default_rng(11)
    lw = (rs.standard_normal(K) * (5.0 if spread == "wide" else 1.0)).astype(np.float32)
    if spread == "first":
        lw[K // 7:] -= 60.0           # nearly all the mass on the first rank: every other rank only receives
    rows = rs.standard_normal((R, K)).astype(np.float32)
    off, k = D.shard(K, rank, world)
    lw_d = torch.as_tensor(lw[off:off + k]).cuda()
    local = kernels.logsumexp(lw_d, K)
    new_rows, info = D.resample_exchange(torch.as_tensor(rows[:, off:off + k].copy()).cuda(), lw_d, None, 0.37, K,
                                         pairs=D.gather_lse_pairs(local))
    q.put((rank, new_rows.cpu().numpy(), info["lse"].cpu().numpy(), info["sent"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spread,world", [("mild", 2), ("wide", 2), ("first", 4)])
def test_sharded_exchange_two_processes_one_gpu(oracle, spread, world):
    """genjax_amd.distributed end to end with the HIP backend: 2 processes share the GPU, gloo carries the
    collectives (device tensors staged through the host).  Result == unsharded oracle, bit for bit."""
    import torch.multiprocessing as mp
    K, R = 50_001, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150) + ["mild", "wide", "first"].index(spread)
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, K, R, spread, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rs = np.random.default_rng(11)
    lw = (rs.standard_normal(K) * (5.0 if spread == "wide" else 1.0)).astype(np.float32)
    if spread == "first":
        lw[K // 7:] -= 60.0
    rows = rs.standard_normal((R, K)).astype(np.float32)
    lse = oracle.logsumexp(lw, K)
    # unsharded run on the device (log weights go through the device exp, so the device is its own reference
    # here; the fixed-point scale is the exact global max in both runs) ...
    import torch
    from genjax_amd import kernels
    lw_d = torch.as_tensor(lw).cuda()
    anc = kernels.resample_indices(lw_d, 0.37, K, lse=kernels.logsumexp(lw_d, K))
    want = kernels.gather_rows(torch.as_tensor(rows).cuda(), anc).cpu().numpy()
    np.testing.assert_array_equal(np.concatenate([r[1] for r in res], axis=1), want)
    # ... which itself differs from the oracle's resampling only at near-ties (device exp vs libm exp)
    cum, _ = oracle.weight_cumsum(lw, True, lse)
    assert (anc.cpu().numpy() != oracle.resample_systematic(cum, 0.37, K)).mean() <= 1e-3
    for r in res:
        np.testing.assert_allclose(r[2][2:], lse[2:], rtol=2e-6)
    if spread != "mild":
        assert sum(r[3] for r in res) > 0
    if spread == "first":
        assert res[0][3] > K // 2 and all(r[3] == 0 for r in res[2:])     # rank 0 feeds everyone; the tail ranks send nothing


def _rccl_one_rank_worker(port, q):
    try:
        _rccl_one_rank_body(port, q)
    except BaseException as e:          # report instead of leaving the parent to time out
        import traceback
        q.put(("error", traceback.format_exc(), None, None, None, None))
        raise


def _rccl_one_rank_body(port, q):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", GJX_FORCE_DIST="1")
    from genjax_amd import distributed as D
    from genjax_amd import kernels
    D.init_from_env("nccl")
    K, R = 70_001, 5
    rs = np.random.default_rng(3)
    lw = torch.as_tensor((rs.standard_normal(K) * 2).astype(np.float32)).cuda()
    rows = torch.as_tensor(rs.standard_normal((R, K)).astype(np.float32)).cuda()
    local = kernels.logsumexp(lw, K)
    res = D.ShardedResampler(K, R, K, "cuda", transport="rccl")
    got, rec = res.step(rows, lw, local, 0.77)
    anc = kernels.resample_indices(lw, 0.77, K, lse=local)
    want = kernels.gather_rows(rows, anc)
    torch.cuda.synchronize()
    # collapsed weights (a stretch of many-offspring particles, the rest dead) through the same context
    lw2 = torch.full((K,), -200.0, device="cuda")
    lw2[20_000:23_000] = torch.as_tensor(rs.standard_normal(3000).astype(np.float32)).cuda()
    local2 = kernels.logsumexp(lw2, K)
    got2, _ = res.step(rows, lw2, local2, 0.31)
    want2 = kernels.gather_rows(rows, kernels.resample_indices(lw2, 0.31, K, lse=local2))
    torch.cuda.synchronize()
    assert torch.equal(got2, want2), "collapsed weights: RCCL path differs from the single-GPU path"
    # multinomial (all-to-all) step through the same context: == unsharded multinomial draw + gather
    got3, rec3 = res.step_multinomial(rows, lw, local, (9, 10))
    cum3, bt3 = kernels.weight_cumsum(lw, True, local)
    want3 = kernels.gather_rows(rows, kernels.resample_multinomial(cum3, bt3, (9, 10), K))
    torch.cuda.synchronize()
    assert torch.equal(got3, want3), "multinomial: RCCL path differs from the single-GPU path"
    assert res.ctx.last_info["sent"] == 0
    got, rec = res.step(rows, lw, local, 0.77)       # last_info below is that of a systematic step
    # the sharded bootstrap filter (per-step exchange through the same transport) against the native one-GPU loop
    from genjax_amd import core, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = workloads.ssm_problem(T=24)
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << 14, weights="global_max")
    os.environ["GJX_FORCE_DIST"] = "0"
    a = bf.run(core.key(5), s["y"])
    os.environ["GJX_FORCE_DIST"] = "1"
    b = bf.run(core.key(5), s["y"])
    c = bf.run(core.key(5), s["y"], step_by_step=True)          # same exchange, one host call per stage
    assert torch.equal(b["x"], c["x"]) and torch.allclose(b["increments"], c["increments"], rtol=2e-6, atol=1e-6)
    # the default scheme, sharded, with an option the peer-mapped filter does not have: global maximum over the collective
    # transport instead of an error
    d = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), 1 << 14).run(core.key(5), s["y"], keep_means=True)
    e = bf.run(core.key(5), s["y"], keep_means=True)
    assert torch.equal(d["x"], e["x"]) and torch.equal(d["means"], e["means"])
    filt = (a["increments"].cpu().numpy(), b["increments"].cpu().numpy(), bool(torch.equal(a["x"], b["x"])), bf._resampler.transport)
    q.put((res.transport, bool(torch.equal(got, want)), rec.cpu().numpy(), local.cpu().numpy(), res.ctx.last_info, filt))
    res.close()
    dist.destroy_process_group()


def test_rccl_transport_single_rank():
    """The one-call RCCL exchange (gjx_shard_resample_step) with a 1-rank communicator: RCCL resolves and
    initialises, the all-gathers run, and the result equals the single-GPU path bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(29950 + os.getpid() % 40, q))
    p.start()
    transport, same, rec, local, info, filt = q.get(timeout=180)
    assert transport != "error", same
    p.join(timeout=60)
    assert p.exitcode == 0
    assert transport == "rccl" and same
    np.testing.assert_allclose(rec, local, rtol=1e-6)
    assert info["sent"] == 0 and info["n_valid"] == 70_001
    inc_native, inc_sharded, same_x, tr2 = filt
    assert tr2 == "rccl" and same_x                       # same ancestors every step -> identical final particles
    np.testing.assert_allclose(inc_sharded, inc_native, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("rng", RNGS)
def test_masked_constraints_parity(K_, oracle, rng):
    """GJX_MODE_OBS_MASK (Mask(value, flag) per particle, distribution.py:129-143): device == oracle; flagged particles
    keep their value and are weighted, the others draw exactly what an unconstrained run draws."""
    import torch
    sl = H.zoo(rng).site_list
    K = 3000
    rs = np.random.default_rng(8)
    masked = {"n0": rs.random(K) < 0.5, "c0": rs.random(K) < 0.3, "mv": rs.random(K) < 0.7, "g0": rs.random(K) < 0.5}
    vals = {"n0": rs.standard_normal((1, K)), "c0": rs.integers(0, 3, (1, K)), "mv": rs.standard_normal((3, K)), "g0": rs.random((1, K)) + 0.5}
    prog = PackedProgram(sl, {a: A.MODE_OBS_MASK for a in masked}, rng_mode=rng)
    ch = np.zeros((prog.n_slots, K), np.float32)
    for a in masked:
        s0 = prog.slot_of[a]
        ch[s0:s0 + sl[a].dim] = vals[a]
        ch[prog.flag_slot_of[a]] = masked[a]
    g = K_.run_program(prog, (6, 7), K, choices=torch.as_tensor(ch).cuda(), want_site_scores=True)
    o = oracle.run_program(prog, (6, 7), K, choices=ch.copy(), want_site_scores=True, want_margin=True)
    gg = {k: _np(v) for k, v in g.items() if k in ("choices", "score", "weight", "logw")}
    assert_particles_match(gg, o, max_bad=2e-3, what="masked zoo")
    free = K_.run_program(PackedProgram(sl, rng_mode=rng), (6, 7), K)
    for a, fl in masked.items():
        s0 = prog.slot_of[a]
        np.testing.assert_array_equal(gg["choices"][s0:s0 + sl[a].dim][:, fl], ch[s0:s0 + sl[a].dim][:, fl])
        if a == "n0":        # first site: nothing upstream differs, so unflagged draws equal the unconstrained run's
            np.testing.assert_array_equal(gg["choices"][s0][~fl], _np(free["choices"])[s0][~fl])
    assert (gg["weight"] != 0).mean() > 0.8


def _random_program(rs, rng_mode):
    """A random valid site program: every kind, parameters drawn from the six expression forms over earlier sites — the five closed
    forms and general expression blocks (GJX_P_EXPR) —
    (positive / probability parameters go through exp / softplus / sigmoid), random constraint modes."""
    POS, PROB, REAL = "pos", "prob", "real"
    spec = {  # kind -> parameter domains
        A.NORMAL: (REAL, POS), A.FLIP: (PROB,), A.BERNOULLI_LOGITS: (REAL,), A.BETA: (POS, POS), A.UNIFORM: None,
        A.EXPONENTIAL: (POS,), A.HALF_NORMAL: (POS,), A.LAPLACE: (REAL, POS), A.LOG_NORMAL: (REAL, POS), A.CAUCHY: (REAL, POS),
        A.GAMMA: (POS, POS), A.STUDENT_T: (POS, REAL, POS), A.POISSON: (POS,), A.GEOMETRIC: (PROB,), A.GUMBEL: (REAL, POS),
        A.HALF_CAUCHY: (REAL, POS), A.INVERSE_GAMMA: (POS, POS), A.WEIBULL: (POS, POS), A.LOGIT_NORMAL: (REAL, POS), A.CHI2: (POS,),
        A.MVNORMAL_DIAG: (REAL, POS),
        A.CHI: (POS,), A.EXP_GAMMA: (POS, POS), A.EXP_INVERSE_GAMMA: (POS, POS), A.HALF_STUDENT_T: (POS, REAL, POS), A.KUMARASWAMY: (POS, POS),
        A.MOYAL: (REAL, POS), A.DOUBLESIDED_MAXWELL: (REAL, POS), A.INVERSE_GAUSSIAN: (POS, POS), A.NEGATIVE_BINOMIAL: (POS, REAL),
        A.VON_MISES: (REAL, POS),
    }
    kinds = [k for k in spec if spec[k] is not None]
    sl = SiteList()
    cont, cats = [], []           # (addr, dim) of continuous sites; (addr, ncat) of categorical sites
    n_sites = int(rs.integers(2, 9))
    for j in range(n_sites):
        addr = f"s{j}"
        if rs.random() < 0.15:
            n = int(rs.integers(2, 6))
            sl.add(addr, A.CATEGORICAL_PROBS if rs.random() < 0.5 else A.CATEGORICAL_LOGITS,
                   [(rs.random(n) + 0.1).astype(np.float32)])
            cats.append((addr, n))
            continue
        kind = A.MVNORMAL_DIAG if rs.random() < 0.12 else int(rs.choice(kinds))      # (vector-valued choices: what a row gather indexes)
        dim = int(rs.integers(2, 5)) if kind == A.MVNORMAL_DIAG else 1

        def param(dom):
            xf = {POS: int(rs.choice([A.XF_EXP, A.XF_SOFTPLUS])), PROB: A.XF_SIGMOID, REAL: A.XF_NONE}[dom]
            form = rs.random()
            base = rs.standard_normal(dim).astype(np.float32) * 0.5
            if form < 0.35 or not cont:
                v = {POS: np.abs(base) + 0.3, PROB: 1 / (1 + np.exp(-base)), REAL: base}[dom]
                return Param.const(v.astype(np.float32))
            vecs = [(a_, d_) for a_, d_ in cont if d_ >= 2]
            if form < 0.5 and cats and vecs:       # a row of an earlier vector-valued choice picked by an earlier categorical (GJX_P_VGATHER)
                a, _ = cats[int(rs.integers(len(cats)))]
                v_, d_ = vecs[int(rs.integers(len(vecs)))]
                return Param.vgather(v_, d_, a, vlen=1, xf=xf)
            if form < 0.6 and cats:
                a, n = cats[int(rs.integers(len(cats)))]
                tab = rs.standard_normal((n, dim)).astype(np.float32) * 0.5
                return Param.gather(tab, a, xf=xf)
            a, d = cont[int(rs.integers(len(cont)))]
            if form < 0.74:
                return Param.value(a, length=1, elem=int(rs.integers(d)), xf=xf)
            if form < 0.87:
                return Param.affine((rs.standard_normal((dim, d)) * 0.3).astype(np.float32), a, bias=base * 0.2, xf=xf)
            # a general expression block over the earlier continuous sites (GJX_P_EXPR), the domain's transform on top of it
            return Param.expr(H.random_expr_outs(rs, cont, dim if rs.random() < 0.5 else 1, "real", depth=2), xf=xf)

        sl.add(addr, kind, [param(dom) for dom in spec[kind]], dim=dim)
        # bounded-magnitude continuous values only feed later parameters (heavy tails would overflow exp())
        if kind in (A.NORMAL, A.MVNORMAL_DIAG, A.LAPLACE, A.GUMBEL, A.LOGIT_NORMAL, A.BETA, A.HALF_NORMAL, A.EXPONENTIAL):
            cont.append((addr, dim))
    return sl


@pytest.mark.parametrize("rng", RNGS)
def test_random_programs_against_oracle(K_, oracle, rng, monkeypatch):
    """Differential test of the engines: 40 random programs, simulate, then re-run with a random subset of sites
    constrained per particle (assess / importance semantics).  Device == oracle up to the stated tolerances.  Even trials
    run on the engine the library picks (a kernel generated for the program, compiled on the spot), odd trials on the site
    interpreter (which also halves the time the test spends in hipRTC)."""
    import torch
    rs = np.random.default_rng(int(os.environ.get("GJX_FUZZ_SEED", "77")) + rng)
    # campaign mode (profiles/fuzz.sh sets the seed): hundreds of programs reach float32 corner cases that no mask on the final
    # score sees — a scale of exp(70), a gradient alpha * (log(alpha) - digamma(alpha) + ...) at alpha = exp(11) whose bracket
    # carries 1e-6 of rounding (seeds 1001-1003: always 1-2 elements in thousands, device and oracle each wrong in their own
    # way).  There a trial may have 2 such particles / 0.1 % such gradient elements; a logic error fails whole columns.
    loose = "GJX_FUZZ_SEED" in os.environ
    K = 600
    for trial in range(int(os.environ.get("GJX_FUZZ_TRIALS", "40"))):
        if trial & 1:
            monkeypatch.setenv("GJX_ENGINE", "interp")
        else:
            monkeypatch.delenv("GJX_ENGINE", raising=False)
        sl = _random_program(rs, rng)
        prog = PackedProgram(sl, rng_mode=rng)
        key = (int(rs.integers(1 << 30)), int(rs.integers(1 << 30)))
        g, o = _run_both(K_, oracle, prog, key, K, want_site_scores=True)
        fin = np.isfinite(o["score"]) & (np.abs(o["score"]) < 1e4)
        # float32 conditioning of the particle: (1) a draw in the denormal range (the device flushes it to zero, the oracle's
        # libm does not); (2) a score that the oracle itself moves by more than the tolerance when its continuous inputs move
        # by a few ulps — a parameter like exp(13) makes a log-density a difference of terms of 1e6 whose last bit is 0.25
        # (found by profiles/fuzz.sh, seeds 1001 / 1002)
        cont_sites = [s_ for s_ in sl.sites if s_.kind not in A.NO_GRADIENT_KINDS and s_.kind not in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS)]
        denorm = ((o["choices"] != 0) & (np.abs(o["choices"]) < 1.2e-38)).any(axis=0)
        # ... or a SCALE in the denormal range — exp(-92) as the scale of a normal whose draw then equals its location: the oracle's
        # libm scores it log(1 / scale) = +91.6, the device flushes the scale to zero (profiles/fuzz.sh, seed 7001 trial 71); a site
        # score above log(1 / FLT_MIN) = 87.3 is the mark
        with np.errstate(invalid="ignore"):
            denorm |= (o["site_scores"] > 87.0).any(axis=0)
        prog_c = PackedProgram(sl, {s_.addr: A.MODE_OBS_SLOT for s_ in sl.sites}, rng_mode=rng)
        ch_p = o["choices"].copy()
        for s_ in cont_sites:
            ch_p[prog_c.slot_of[s_.addr]:prog_c.slot_of[s_.addr] + s_.dim] *= np.float32(1.0 + 2e-6)
        s_a = oracle.run_program(prog_c, key, K, choices=o["choices"].copy())["score"]
        s_b = oracle.run_program(prog_c, key, K, choices=ch_p)["score"]
        with np.errstate(invalid="ignore"):
            fin &= ~denorm & (np.abs(s_b - s_a) <= 5e-4 + 5e-4 * np.abs(s_a))
            if loose:
                # campaign mode keeps away from draws so small that a log() of them loses the value's own float32 rounding.
                # (A cap |v| < 8 used to sit here as well: the oracle's float32 log-densities were piecewise constant at
                # parameters of 1e6 and hid their own ill-conditioning from the probe above.  The oracle now evaluates the closed
                # forms in double, the probe sees every ill-conditioned particle, and the device takes lgamma DIFFERENCES from
                # their asymptotic series: the cap is gone.)
                cs = np.concatenate([np.arange(prog_c.slot_of[s_.addr], prog_c.slot_of[s_.addr] + s_.dim) for s_ in cont_sites]) if cont_sites else np.zeros(0, int)
                if cs.size:
                    v_ = np.abs(o["choices"][cs])
                    fin &= ((v_ > 1e-4) | (v_ == 0)).all(axis=0)
        assert fin.mean() > (0.05 if loose else 0.3), f"trial {trial}: {fin.mean():.2f} of the particles are well conditioned"
        ok = _close_cols(g["choices"], o["choices"], rt=1e-3, at=5e-4) & _close_cols(g["score"][None], o["score"][None], rt=2e-3, at=2e-3)
        miss = ~ok & fin
        if loose and 0 < (miss & (o["margin"] >= NEAR_TIE)).sum() <= 2:
            miss &= o["margin"] < NEAR_TIE
        assert_near_ties_only(miss, o, f"trial {trial} ({[A.KIND_NAMES[s.kind] for s in sl.sites]})", cap=0.02 if loose else 0.005)
        # analytic gradients of the same program at the oracle's draws (every site constrained, float sites selected)
        sel = tuple(s.addr for s in sl.sites if s.kind not in A.NO_GRADIENT_KINDS and s.kind not in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS))
        if sel:
            prog3 = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=sel, rng_mode=rng)
            sg, gg = K_.score_grad(prog3, torch.as_tensor(o["choices"]).cuda())
            so, go = oracle.score_grad(prog3, o["choices"])
            okg = np.isfinite(go) & (np.abs(go) < 1e3) & fin[None, :]
            # conditioning of the GRADIENT, as for the score above: an element the oracle itself moves by more than a fraction of the
            # tolerance when the continuous inputs move by a few float32 ulps (a residual over a scale of softplus(-15) squared, a
            # product of such terms inside an expression block) is not a parity question
            with np.errstate(invalid="ignore"):
                go_p = oracle.score_grad(prog3, ch_p)[1]
                okg &= np.abs(go_p - go) <= 1e-3 + 1e-3 * np.abs(go)
            if loose:
                with np.errstate(invalid="ignore"):
                    off = ~(np.abs(_np(gg) - go) <= 5e-3 + 5e-3 * np.abs(go)) & okg
                # (e.g. d/d(df) of a student-t at df = exp(20): the device takes digamma((df + 1) / 2) - digamma(df / 2) in float32
                # like the reference's autodiff would, the oracle in double)
                assert off.sum() <= max(2, int(1e-2 * okg.sum())), f"trial {trial} gradients: {int(off.sum())} of {int(okg.sum())} elements differ"
            else:
                # (a gradient row is a sum of terms that may cancel: one element in a few thousand whose terms are a hundred times
                # its value carries their float32 rounding — at most 5 in 10^4 elements may miss the tolerance, none by more than 10x)
                with np.errstate(invalid="ignore"):      # (inf - inf where the oracle's gradient is not finite: masked by okg)
                    d_ = np.abs(_np(gg) - go)
                    off = ~(d_ <= 5e-3 + 5e-3 * np.abs(go)) & okg
                assert off.sum() <= max(1, int(5e-4 * okg.sum())) and (d_[off] <= 5e-2 + 5e-2 * np.abs(go[off])).all(), \
                    f"trial {trial} gradients: {int(off.sum())} of {int(okg.sum())} elements differ, worst {float(d_[off].max()) if off.any() else 0.0}"
        # constrain a random subset of sites to the oracle's own draws: values untouched, weights = their log-pdfs
        sub = [s.addr for s in sl.sites if rs.random() < 0.5]
        if not sub:
            continue
        prog2 = PackedProgram(sl, {a: A.MODE_OBS_SLOT for a in sub}, rng_mode=rng)
        ch = o["choices"].copy()
        g2 = K_.run_program(prog2, key, K, choices=torch.as_tensor(ch).cuda(), want_site_scores=True)
        o2 = oracle.run_program(prog2, key, K, choices=ch.copy(), want_site_scores=True)
        idx = [j for j, s in enumerate(sl.sites) if s.addr in sub]
        gs, os_ = _np(g2["site_scores"])[idx], o2["site_scores"][idx]
        good = np.isfinite(os_) & (np.abs(os_) < 1e4) & fin[None, :]
        # conditioning of the SITE score (the probe above holds the particle's total score, which a large neighbour can dominate): an
        # element the oracle itself moves by a good part of the tolerance when the continuous inputs move by a few ulps — exp_gamma at a
        # concentration of 1e5 and a value of 20: a (log b + x) - b e^x - lgamma(a) = 9.35 from terms of 1e6 — is not a parity question
        os_p = oracle.run_program(prog2, key, K, choices=ch_p.copy(), want_site_scores=True)["site_scores"][idx]
        with np.errstate(invalid="ignore"):
            good &= np.abs(os_p - os_) <= 5e-4 + 5e-4 * np.abs(os_)
        if loose:
            with np.errstate(invalid="ignore"):
                off = ~(np.abs(gs - os_) <= 2e-3 + 2e-3 * np.abs(os_)) & good
            assert off.sum() <= max(2, int(5e-3 * good.sum())), f"trial {trial}: {int(off.sum())} of {int(good.sum())} site scores differ"
        else:
            np.testing.assert_allclose(gs[good], os_[good], rtol=2e-3, atol=2e-3, err_msg=f"trial {trial}")
        for a in sub:
            s0 = prog2.slot_of[a]
            np.testing.assert_array_equal(_np(g2["choices"])[s0:s0 + sl[a].dim], ch[s0:s0 + sl[a].dim])
