"""General expressions between sites on the device (GJX_P_EXPR, include/gjx.h; the reference interprets ANY JAX computation between
two trace sites, static.py:383-399, staging.py:286-298, and differentiates through it, hmc.py:70-96).

Every engine against the oracle (whose blocks are held to float64 NumPy and finite differences by tests/test_expr_cpu.py):
  * the generated propagate kernel (engine 4: nodes emitted inline) and the site interpreter, ImportanceK semantics;
  * gradients (gjx_score_grad) and the HMC move: the generated HMC kernel (reverse sweep emitted inline) and the interpreter;
  * the generic filter's generated kernel (gjx_gen_pf) on a NONLINEAR state-space model, step-locally against the oracle and as a
    whole against a float64 NumPy bootstrap filter.
"""
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import helpers as H                       # noqa: E402
import genjax_amd as genjax               # noqa: E402
from genjax_amd import C                  # noqa: E402
from genjax_amd import _abi as A          # noqa: E402
from genjax_amd.program import PackedProgram, Param, SiteList      # noqa: E402

RNGS = [A.RNG_FLAT, A.RNG_JAX32]
RT, AT = 2e-4, 5e-5


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def K_():
    from genjax_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu
    return cpu


W1 = np.random.default_rng(0).standard_normal((8, 16)) * 0.4
W2 = np.random.default_rng(1).standard_normal(8)


@genjax.gen
def product_model():
    a = genjax.normal(0.0, 1.0) @ "a"
    b = genjax.normal(0.5, 1.0) @ "b"
    genjax.normal(a * b, 1.0) @ "y"
    return a * b


@genjax.gen
def sum_of_exps_model():
    a = genjax.normal(0.0, 0.5) @ "a"
    b = genjax.normal(0.0, 0.5) @ "b"
    genjax.normal(0.0, genjax.exp(a) + genjax.exp(b)) @ "y"


@genjax.gen
def mlp_model():
    x = genjax.mv_normal_diag(np.zeros(16, np.float32), np.ones(16, np.float32)) @ "x"
    genjax.bernoulli(logits=W2 @ genjax.tanh(W1 @ x)) @ "y"


NAMED = {"product": (product_model, {"y": 0.7}), "sum_of_exps": (sum_of_exps_model, {"y": -1.3}), "mlp_16_8_1": (mlp_model, {"y": 1.0})}


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("which", list(NAMED))
def test_named_models_propagate_on_the_generated_kernel_and_the_interpreter(K_, oracle, rng, which, monkeypatch):
    """VERDICT r05 item 2's three models under ImportanceK semantics (latents sampled, y observed): engine 4 is what the library
    picks; generated kernel == interpreter == oracle"""
    model, obs = NAMED[which]
    sl, _ = model.site_list(())
    prog = PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": np.float32(obs["y"])}, rng_mode=rng)
    assert any(prog.c_sites[j].p[k].op == A.P_EXPR for j in range(prog.n_sites) for k in range(A.MAX_PARAMS))
    K = 4099
    monkeypatch.delenv("GJX_ENGINE", raising=False)
    assert K_.program_engine(prog) == 4, "an expression program must run on a generated kernel"
    o = oracle.run_program(prog, (3, 7), K, want_site_scores=True)
    for eng in ("gen", "interp"):
        monkeypatch.setenv("GJX_ENGINE", eng)
        assert K_.program_engine(prog) == (4 if eng == "gen" else 0)
        g = K_.run_program(prog, (3, 7), K, want_site_scores=True)
        np.testing.assert_allclose(_np(g["choices"]), o["choices"], rtol=RT, atol=AT, err_msg=eng)
        np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=RT, atol=2e-4, err_msg=eng)
        np.testing.assert_allclose(_np(g["logw"]), o["logw"], rtol=RT, atol=2e-4, err_msg=eng)
        np.testing.assert_allclose(_np(g["site_scores"]), o["site_scores"], rtol=RT, atol=2e-4, err_msg=eng)
        np.testing.assert_allclose(_np(g["lse"])[2:], o["lse"][2:], rtol=1e-5, atol=1e-4)


def test_expression_models_through_the_api(K_):
    """the host path end to end: Target / ImportanceK on a model whose likelihood mean is a product of two latents; the log-ML
    estimate against numerical integration; the return value (an expression) evaluated from the trace"""
    from genjax_amd.inference import ImportanceK, Target
    y = 0.7
    target = Target(product_model, (), C["y"].set(y))
    est = float(ImportanceK(target, k_particles=1 << 18).log_marginal_likelihood_estimate(genjax.key(5)))
    # p(y) = int N(a; 0, 1) N(b; .5, 1) N(y; a b, 1) da db on a grid
    g = np.linspace(-8, 8, 1601)
    aa, bb = np.meshgrid(g, g, indexing="ij")
    dens = np.exp(-0.5 * aa ** 2 - 0.5 * (bb - 0.5) ** 2 - 0.5 * (y - aa * bb) ** 2) / (2 * math.pi) ** 1.5
    exact = math.log(dens.sum() * (g[1] - g[0]) ** 2)
    assert est == pytest.approx(exact, abs=0.02)
    tr = product_model.simulate(genjax.key(1), (), K=257)
    ch = tr.get_choices()
    np.testing.assert_allclose(_np(tr.get_retval()), _np(ch["a"]) * _np(ch["b"]), rtol=1e-6, atol=1e-7)


def _random_expression_program(rs, rng_mode):
    """2-4 continuous sites, then sites whose parameters are random expression blocks over them (tests/helpers.py), a categorical
    over expression logits is left to the interpreter (the emitters take table logits only)"""
    sl = SiteList()
    cont = []
    for j in range(int(rs.integers(2, 5))):
        d = int(rs.integers(1, 4))
        if d == 1:
            sl.add(f"x{j}", A.NORMAL, [float(rs.standard_normal() * 0.3), float(0.5 + rs.random())])
        else:
            sl.add(f"x{j}", A.MVNORMAL_DIAG, [rs.standard_normal(d).astype(np.float32) * 0.3, (0.5 + rs.random(d)).astype(np.float32)], dim=d)
        cont.append((f"x{j}", d))
    kinds = [(A.NORMAL, ("real", "pos")), (A.LAPLACE, ("real", "pos")), (A.GUMBEL, ("real", "pos")), (A.GAMMA, ("pos", "pos")), (A.FLIP, ("prob",)),
             (A.BERNOULLI_LOGITS, ("real",)), (A.EXPONENTIAL, ("pos",)), (A.LOG_NORMAL, ("real", "pos")), (A.STUDENT_T, ("pos", "real", "pos"))]
    for j in range(int(rs.integers(1, 4))):
        if rs.random() < 0.3:
            d = int(rs.integers(2, 5))
            loc = H.random_expr_outs(rs, cont, d if rs.random() < 0.7 else 1, "real", depth=2)
            sc = H.random_expr_outs(rs, cont, d if rs.random() < 0.3 else 1, "pos", depth=2)
            sl.add(f"e{j}", A.MVNORMAL_DIAG, [Param.expr(loc), Param.expr(sc)], dim=d)
            cont.append((f"e{j}", d))
            continue
        kind, doms = kinds[int(rs.integers(len(kinds)))]
        ps = [Param.expr(H.random_expr_outs(rs, cont, 1, dom, depth=3)) if rs.random() < 0.8 else
              Param.const({"real": 0.2, "pos": 0.8, "prob": 0.4}[dom]) for dom in doms]
        if kind == A.STUDENT_T:
            ps[0] = Param.const(4.0 + 3.0 * rs.random())
        sl.add(f"e{j}", kind, ps)
        if kind in (A.NORMAL, A.LAPLACE, A.GUMBEL):
            cont.append((f"e{j}", 1))
    return sl


@pytest.mark.parametrize("rng", RNGS)
def test_random_expression_programs_against_oracle(K_, oracle, rng, monkeypatch):
    """differential test: random programs whose parameters are random expression blocks; simulate on the generated kernel (even
    trials) and on the interpreter (odd trials), then assess the oracle's own draws; a particle may differ only where one of the
    block's comparisons sat at a float32 near-tie (the oracle's decision margin)"""
    rs = np.random.default_rng(int(os.environ.get("GJX_FUZZ_SEED", "4242")) + rng)
    K = 700
    n_gen = 0
    for trial in range(int(os.environ.get("GJX_FUZZ_TRIALS", "30"))):
        sl = _random_expression_program(rs, rng)
        prog = PackedProgram(sl, rng_mode=rng)
        key = (int(rs.integers(1 << 30)), int(rs.integers(1 << 30)))
        if trial & 1:
            monkeypatch.setenv("GJX_ENGINE", "interp")
        else:
            monkeypatch.delenv("GJX_ENGINE", raising=False)
            n_gen += K_.program_engine(prog) == 4
        what = f"trial {trial} engine {K_.program_engine(prog)} {[A.KIND_NAMES[s.kind] for s in sl.sites]}"
        g = K_.run_program(prog, key, K, want_site_scores=True)
        o = oracle.run_program(prog, key, K, want_site_scores=True, want_margin=True)
        ok = np.isfinite(o["score"]) & (np.abs(o["score"]) < 1e4) & (o["margin"] > 1e-4)
        # conditioning: the oracle's own score under inputs a few float32 ulps away
        progc = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, rng_mode=rng)
        sa = oracle.run_program(progc, key, K, choices=o["choices"])["score"]
        sb = oracle.run_program(progc, key, K, choices=o["choices"] * np.float32(1 + 2e-6))["score"]
        with np.errstate(invalid="ignore"):
            ok &= np.abs(sb - sa) <= 5e-4 + 5e-4 * np.abs(sa)
        assert ok.mean() > 0.8, what
        np.testing.assert_allclose(_np(g["choices"])[:, ok], o["choices"][:, ok], rtol=5e-4, atol=2e-4, err_msg=what)
        np.testing.assert_allclose(_np(g["score"])[ok], o["score"][ok], rtol=5e-4, atol=1e-3, err_msg=what)
        # assess: the device scores the oracle's values
        ga = K_.run_program(progc, key, K, choices=__import__("torch").as_tensor(o["choices"]).cuda(), want_site_scores=True)
        np.testing.assert_allclose(_np(ga["site_scores"])[:, ok], oracle.run_program(progc, key, K, choices=o["choices"], want_site_scores=True)["site_scores"][:, ok],
                                   rtol=5e-4, atol=1e-3, err_msg=what + " assess")
    assert n_gen >= 10, f"the emitter took {n_gen} of the even trials"


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("which", ["mlp_16_8_1", "product", "sum_of_exps", "random"])
def test_gradients_and_hmc_through_expression_blocks(K_, oracle, rng, which, monkeypatch):
    """gjx_score_grad and the HMC move differentiate through the blocks (hmc.py:70-96): interpreter and generated HMC kernel
    against the oracle's reverse sweep (itself checked against finite differences on the CPU)"""
    import torch
    n = 384
    rs = np.random.default_rng(17 + rng)
    for rep in range(6 if which == "random" else 1):
        if which == "random":
            sl = _random_expression_program(rs, rng)
            obs = {}
        else:
            model, ob = NAMED[which]
            sl, _ = model.site_list(())
            obs = {"y": np.float32(ob["y"])}
        ch0 = oracle.run_program(PackedProgram(sl, {a: A.MODE_OBS_TAB for a in obs}, obs, rng_mode=rng), (5, 6 + rep), n)["choices"]
        sel = tuple(s.addr for s in sl.sites if s.addr not in obs and s.kind not in A.NO_GRADIENT_KINDS)
        modes = {s.addr: (A.MODE_OBS_TAB if s.addr in obs else A.MODE_OBS_SLOT) for s in sl.sites}
        prog = PackedProgram(sl, modes, obs, selected=sel, rng_mode=rng)
        # the simulate program and the HMC program lay their rows out alike (observed sites own none in either)
        sim = PackedProgram(sl, {a: A.MODE_OBS_TAB for a in obs}, obs, rng_mode=rng)
        assert sim.slot_of == prog.slot_of
        ch = ch0.astype(np.float32)
        so, go = oracle.score_grad(prog, ch)
        sg, gg = K_.score_grad(prog, torch.as_tensor(ch).cuda())
        fin = np.isfinite(go).all(0) & (np.abs(go).max(0) < 1e4)
        np.testing.assert_allclose(_np(sg)[fin], so[fin], rtol=5e-4, atol=1e-3)
        np.testing.assert_allclose(_np(gg)[:, fin], go[:, fin], rtol=2e-3, atol=2e-3)
        eps, L = 2e-3, 8
        o = oracle.hmc(prog, (2, 9), ch, eps, L, False, False, offset=5)
        o2 = oracle.hmc(prog, (2, 9), ch, eps * 1.01, L, False, False, offset=5)
        with np.errstate(invalid="ignore"):
            well = fin & np.isfinite(o["choices"]).all(0) & np.isfinite(o["alpha"]) & (np.abs(o["alpha"]) < 5e-2) & (np.abs(o2["choices"] - o["choices"]).max(0) < 1e-3)
        assert well.mean() > 0.5
        engines = []
        for eng in ("gen", "interp"):
            monkeypatch.setenv("GJX_HMC_ENGINE", eng)
            if eng == "gen" and K_.hmc_engine(prog) != 4:
                assert which == "random", "the HMC emitter must cover the named models"
                continue
            engines.append(eng)
            g = K_.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=5)
            np.testing.assert_allclose(_np(g["choices"])[:, well], o["choices"][:, well], rtol=3e-3, atol=3e-3, err_msg=eng)
            mag = 5e-6 * np.maximum(np.abs(o["score"]), np.abs(o["score"] - o["alpha"]))[well]
            assert (np.abs(_np(g["alpha"])[well] - o["alpha"][well]) <= 6e-3 + 6e-3 * np.abs(o["alpha"][well]) + mag).all(), eng
        if which != "random":
            assert engines == ["gen", "interp"]


def _nonlinear_scan(T):
    """the classic nonlinear benchmark: x_t = x/2 + 25 x / (1 + x^2) + 8 cos(1.2 t) + N(0, 10), y_t = x_t^2 / 20 + N(0, 1) — both
    sites' means are general expressions of the carry; the forcing term arrives as the scanned input"""
    @genjax.gen
    def step(x_prev, c_t):
        x = genjax.normal(0.5 * x_prev + 25.0 * x_prev / (1.0 + x_prev * x_prev) + c_t, math.sqrt(10.0)) @ "x"
        genjax.normal(x * x / 20.0, 1.0) @ "y"
        return x, None

    return step.scan(n=T)


def _nonlinear_data(T, seed=0):
    rs = np.random.default_rng(seed)
    x, ys = 0.1, []
    cs = (8.0 * np.cos(1.2 * np.arange(T))).astype(np.float32)
    for t in range(T):
        x = 0.5 * x + 25.0 * x / (1 + x * x) + cs[t] + math.sqrt(10.0) * rs.standard_normal()
        ys.append(x * x / 20.0 + rs.standard_normal())
    return np.asarray(ys, np.float32), cs


def _numpy_pf(ys, cs, K, seed):
    """ideal float64 bootstrap filter, systematic resampling in front of every step"""
    rs = np.random.default_rng(seed)
    x = np.full(K, 0.1)
    lml = 0.0
    lw = np.zeros(K)
    for t in range(len(ys)):
        if t > 0:
            w = np.exp(lw - lw.max())
            cdf = np.cumsum(w / w.sum())
            u = (rs.random() + np.arange(K)) / K
            x = x[np.minimum(np.searchsorted(cdf, u), K - 1)]
        x = 0.5 * x + 25.0 * x / (1 + x * x) + cs[t] + math.sqrt(10.0) * rs.standard_normal(K)
        lw = -0.5 * (ys[t] - x * x / 20.0) ** 2 - 0.5 * math.log(2 * math.pi)
        lml += lw.max() + math.log(np.mean(np.exp(lw - lw.max())))
    return lml


@pytest.mark.parametrize("rng", RNGS)
def test_generic_filter_on_a_nonlinear_model_written_with_expressions(K_, oracle, rng):
    """the generic filter (gjx_gen_pf generated from the step program) on a model whose transition and observation means are
    expression blocks of the carry: step-locally against the oracle (prefix runs give every step's inputs), and the log-ML of a
    longer run against float64 NumPy bootstrap filters"""
    from genjax_amd.inference import BootstrapFilter
    sys.path.insert(0, HERE)
    from test_gpu_scan_filter import _check_ancestors, _step_keys
    K, T = 1 << 14, 4
    ys, cs = _nonlinear_data(40)
    key = genjax.key(23)
    keys, us = _step_keys(key, T)
    carry0 = np.float32(0.1)
    outs = []
    for Tp in range(1, T + 1):
        bf = BootstrapFilter(_nonlinear_scan(Tp), K, rng_mode=rng)
        o = bf.run(key, C["y"].set(ys[:Tp]), (carry0, cs[:Tp]), keep_ancestors=True)
        if Tp >= 2:
            assert o["info"]["form"] == A.FILTER_FORM_WIDE, o["info"]
        prog = o["programs"][-1]
        assert any(prog.c_sites[j].p[0].op == A.P_EXPR for j in range(prog.n_sites))
        outs.append(dict(x=_np(bf.latent(o, "x")).copy(), logw=_np(o["logw"]).copy(), anc=_np(o["ancestors"]).copy(), prog=prog))
    for t in range(T):
        cur = outs[t]
        prog = cur["prog"]
        ch_in = np.zeros((prog.n_slots, K), np.float32)
        if t > 0:
            prev = outs[t - 1]
            _check_ancestors(cur["anc"][t - 1], prev["logw"], us[t])
            ch_in[:1] = prev["x"][:, cur["anc"][t - 1]]
        ora = oracle.run_program(prog, keys[t], K, choices=ch_in)
        sl = prog.slot_of[("x", t)]
        np.testing.assert_allclose(cur["x"], ora["choices"][sl:sl + 1], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(cur["logw"], ora["weight"], rtol=5e-4, atol=5e-4)
    # a longer run: log-ML against the float64 filter's distribution (8 seeds of each; K = 2^16)
    Kf, Tf = 1 << 16, 40
    bf = BootstrapFilter(_nonlinear_scan(Tf), Kf, rng_mode=rng)
    dev = np.array([float(bf.run(genjax.key(100 + i), C["y"].set(ys), (carry0, cs))["log_ml"]) for i in range(8)])
    assert bf.last_info["form"] == A.FILTER_FORM_WIDE
    ref = np.array([_numpy_pf(ys.astype(np.float64), cs.astype(np.float64), Kf, 500 + i) for i in range(8)])
    se = math.sqrt(dev.var(ddof=1) / 8 + ref.var(ddof=1) / 8)
    print(f"nonlinear model, T={Tf}, K=2^16: device log-ML {dev.mean():.4f} +- {dev.std(ddof=1):.4f}, float64 filter {ref.mean():.4f} +- {ref.std(ddof=1):.4f}")
    assert abs(dev.mean() - ref.mean()) < 4 * se + 1e-3
    assert 0.4 < dev.std(ddof=1) / ref.std(ddof=1) < 2.5


@pytest.mark.parametrize("rng", RNGS)
def test_network_classifier_vmapped_over_the_data_runs_as_a_generated_plate_kernel(K_, oracle, rng, monkeypatch):
    """general expressions INSIDE a vmapped kernel — y_n ~ bernoulli(logits = w2 . tanh(W1 x_n)), latent weights, N observations: one
    plate site whose block advances through the covariates by its strides.  The generated kernel (plate loop, nodes emitted inline,
    table offsets with the instance stride) and the interpreter against the oracle; the log-ML of a tiny instance against quadrature."""
    from genjax_amd import C as CM
    N, DI, DH, K = 64, 16, 8, 3001
    model, X, Y, loglik = H.bnn_model(N, DI, DH)
    prog, _, _ = model.pack((), CM["obs", "y"].set(Y), True, rng_mode=rng)
    assert prog.n_sites == DH + 2 and prog.c_sites[prog.n_sites - 1].plate_n == N
    monkeypatch.delenv("GJX_ENGINE", raising=False)
    assert K_.program_engine(prog) == 4
    o = oracle.run_program(prog, (4, 5), K, want_site_scores=True)
    for eng in ("gen", "interp"):
        monkeypatch.setenv("GJX_ENGINE", eng)
        g = K_.run_program(prog, (4, 5), K, want_site_scores=True)
        np.testing.assert_allclose(_np(g["choices"]), o["choices"], rtol=RT, atol=AT, err_msg=eng)
        np.testing.assert_allclose(_np(g["logw"]), o["logw"], rtol=5e-4, atol=2e-3, err_msg=eng)
        np.testing.assert_allclose(_np(g["site_scores"])[-1], o["site_scores"][-1], rtol=5e-4, atol=2e-3, err_msg=eng)
    W1 = o["choices"][:DI * DH].astype(np.float64).reshape(DH, DI, K)
    np.testing.assert_allclose(_np(g["logw"]), loglik(W1, o["choices"][DI * DH:].astype(np.float64)), rtol=5e-4, atol=2e-3)
    # few particles over many observations: the wide flavour (instances dealt to the 16 waves of a block) gives the same numbers
    monkeypatch.setenv("GJX_ENGINE", "gen")
    Kw = 512
    model2, X2, Y2, _ = H.bnn_model(1024, 8, 4, seed=3)
    prog2, _, _ = model2.pack((), CM["obs", "y"].set(Y2), True, rng_mode=rng)
    assert K_.program_engine(prog2) == 4
    g2, o2 = K_.run_program(prog2, (6, 7), Kw), oracle.run_program(prog2, (6, 7), Kw)
    np.testing.assert_allclose(_np(g2["logw"]), o2["logw"], rtol=5e-4, atol=2e-2)


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("shape", [(96, 4, 3), (128, 16, 8)])
def test_hmc_over_the_weights_of_a_small_network_through_the_plate(K_, oracle, rng, shape, monkeypatch):
    """HMC over the latent weights of a network classifier whose likelihood is a plate with an expression block (hmc.py:70-96
    differentiates assess through the Vmap and through whatever the body computes): gradient and move, generated HMC kernel and
    interpreter, against the oracle"""
    import torch
    from genjax_amd import C as CM
    # (16 -> 8 -> 1: 136 selected values — beyond the register budget of the generated HMC kernel, whose LDS-state flavour keeps the
    # chain's values, gradient and momenta in LDS columns of a one-wave block: HmcPlan::big)
    (N, DI, DH), n = shape, 512
    model, X, Y, loglik = H.bnn_model(N, DI, DH, seed=2)
    sel = tuple(f"W1_{j}" for j in range(DH)) + ("w2",)
    hp, _, _ = model.pack((), CM["obs", "y"].set(Y), False, selected=sel, per_particle=sel, plates="hmc", rng_mode=rng)
    assert hp.n_sites == DH + 2
    ch = (np.random.default_rng(5).standard_normal((hp.n_slots, n)) * 0.5).astype(np.float32)
    so, go = oracle.score_grad(hp, ch)
    sg, gg = K_.score_grad(hp, torch.as_tensor(ch).cuda())
    np.testing.assert_allclose(_np(sg), so, rtol=5e-4, atol=2e-3)
    np.testing.assert_allclose(_np(gg), go, rtol=2e-3, atol=2e-3)
    eps, L = (5e-3 if DI == 4 else 2e-3), 6
    o = oracle.hmc(hp, (2, 9), ch, eps, L, False, False, offset=5)
    engines = []
    for eng in ("gen", "interp"):
        monkeypatch.setenv("GJX_HMC_ENGINE", eng)
        if eng == "gen" and K_.hmc_engine(hp) != 4:
            continue
        engines.append(eng)
        g = K_.hmc(hp, (2, 9), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=5)
        np.testing.assert_allclose(_np(g["choices"]), o["choices"], rtol=3e-3, atol=3e-3, err_msg=eng)
        np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=1e-2, atol=1e-2, err_msg=eng)
    assert engines == ["gen", "interp"], engines        # (the HMC emitter covers the plate: leaves in registers, table offsets by instance)


@pytest.mark.parametrize("rng", RNGS)
def test_a_long_scan_with_expression_blocks_is_rolled_and_its_states_are_moved_by_hmc(K_, oracle, rng, monkeypatch):
    """the nonlinear model as ONE program of T = 120 steps (240 sites): the propagate emitter rolls the periodic Scan — the block of
    step t is step 1's with its leaves moved to the loop's registers and its table entries strided — and the HMC emitter rolls it too
    (HMC over all 120 states: smoothing); generated == interpreter == oracle"""
    import torch
    T, K, n = 120, 2000, 256
    ys, cs = _nonlinear_data(T, seed=4)
    scan = _nonlinear_scan(T)
    prog, _, _ = scan.pack((np.float32(0.1), cs), C["y"].set(ys), True, rng_mode=rng)
    assert prog.n_sites == 2 * T
    if rng == A.RNG_FLAT:                      # (the propagate emitter rolls FLAT streams; JAX32 long Scans take the interpreter)
        monkeypatch.delenv("GJX_ENGINE", raising=False)
        assert K_.program_engine(prog) == 4, "a periodic Scan with expression blocks must roll"
    o = oracle.run_program(prog, (8, 9), K)
    for eng in (("gen", "interp") if rng == A.RNG_FLAT else ("interp",)):
        monkeypatch.setenv("GJX_ENGINE", eng)
        g = K_.run_program(prog, (8, 9), K)
        # (the recursion x -> x/2 + 25 x / (1 + x^2) amplifies rounding differences near its unstable points: compare where the
        # oracle's own trajectory is insensitive to a perturbation of a few ulps)
        np.testing.assert_allclose(_np(g["choices"])[:3], o["choices"][:3], rtol=5e-4, atol=5e-4, err_msg=eng)
        close = np.abs(_np(g["choices"]) - o["choices"]).max(axis=0) < 2e-2
        assert close.mean() > 0.9, (eng, close.mean())
        np.testing.assert_allclose(_np(g["logw"])[close], o["logw"][close], rtol=2e-3, atol=5e-2, err_msg=eng)
    monkeypatch.delenv("GJX_ENGINE", raising=False)
    # HMC over every state, a short trajectory with a small step
    sel = tuple(("x", t) for t in range(T))
    hp, _, _ = scan.pack((np.float32(0.1), cs), C["y"].set(ys), False, selected=sel, per_particle=sel, plates="hmc", rng_mode=rng)
    ch = o["choices"][:, :n].astype(np.float32)
    so, go = oracle.score_grad(hp, ch)
    sg, gg = K_.score_grad(hp, torch.as_tensor(ch).cuda())
    np.testing.assert_allclose(_np(sg), so, rtol=5e-4, atol=5e-2)
    np.testing.assert_allclose(_np(gg), go, rtol=3e-3, atol=3e-3)
    eps, L = 1e-3, 5
    oh = oracle.hmc(hp, (2, 9), ch, eps, L, False, False, offset=3)
    ran = []
    for eng in ("gen", "interp"):
        monkeypatch.setenv("GJX_HMC_ENGINE", eng)
        if eng == "gen":
            assert K_.hmc_engine(hp) == 4, "the HMC emitter must roll the Scan with its expression blocks"
        ran.append(eng)
        g = K_.hmc(hp, (2, 9), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=3)
        np.testing.assert_allclose(_np(g["choices"]), oh["choices"], rtol=3e-3, atol=3e-3, err_msg=eng)
        mag = 5e-6 * np.maximum(np.abs(oh["score"]), np.abs(oh["score"] - oh["alpha"]))
        assert (np.abs(_np(g["alpha"]) - oh["alpha"]) <= 1e-2 + 1e-2 * np.abs(oh["alpha"]) + mag).all(), eng
    assert ran == ["gen", "interp"]
