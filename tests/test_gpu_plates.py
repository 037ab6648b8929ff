"""Vmap as a device construct (include/gjx.h "Plates"; reference combinators/vmap.py:180-218): the m sites of a vmapped kernel
are m plate-tagged device sites and the engines run ONE instance loop over them — per-instance rows, observations, masks,
tables and sources; categorical sites and gathers on an index of the same instance included.  Every test runs the HIP path
through the C ABI and checks it against the ORACLE (oracle/gjx_oracle.c) on the very program the device ran."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import genjax_amd as genjax
from genjax_amd import C
from genjax_amd import _abi as A

NEAR_TIE = 3e-4


def _np(t):
    return t.detach().cpu().numpy()


def _mixture(N, seed=0, mu=(-2.0, 0.5, 3.0)):
    """the textbook mixture: z_i ~ categorical(logits), x_i ~ normal(mu[z_i], s), x observed — vmapped over N data"""
    rs = np.random.default_rng(seed)
    mu = np.array(mu, np.float32)
    ys = (mu[rs.integers(0, 3, N)] + 0.7 * rs.standard_normal(N)).astype(np.float32)
    logits = np.array([0.2, -0.3, 0.1], np.float32)

    @genjax.gen
    def kernel(lg):
        z = genjax.categorical(logits=lg) @ "z"
        return genjax.normal(genjax.take(mu, z), 0.7) @ "x"

    @genjax.gen
    def model():
        kernel.repeat(n=N)(logits) @ "k"

    return model, C["k", "x"].set(ys), ys, mu, logits


def _check_against_oracle(prog, out, ora, discrete_rows=()):
    """values / scores against the oracle; a particle may differ only where one of the oracle's discrete decisions for it
    was a near tie (oracle `decide` margins: the smallest margin over ALL of the particle's decisions — thousands, in a plate) — so
    the fraction that differs is bounded by the fraction of near-tie particles the oracle itself reports, and is expected well below
    it (a near tie flips only when the two sides' rounding errors straddle it: at most half of them); discrete ELEMENTS may differ
    in at most 1e-4 of all elements"""
    ch, oc = _np(out["choices"]), ora["choices"]
    close = (np.abs(ch - oc) <= 5e-5 + 2e-4 * np.abs(oc)).all(axis=0)
    for k in ("score", "weight"):
        g, o = _np(out[k]), ora[k]
        close &= np.abs(g - o) <= 2e-3 + 3e-4 * np.abs(o)
    bad = ~close
    if bad.any():
        m = ora["margin"][bad]
        assert (m < NEAR_TIE).all(), f"{int((m >= NEAR_TIE).sum())} differing particles are not near ties (max margin {float(m.max()):.3g})"
        near = float((ora["margin"] < NEAR_TIE).mean())
        assert bad.mean() <= 0.6 * near + 3.0 / bad.size, f"{bad.mean():.4f} of the particles differ; the oracle has {near:.4f} near-tie particles"
    if len(discrete_rows):
        diff = ch[list(discrete_rows)] != oc[list(discrete_rows)]
        assert diff.mean() <= 1e-4, diff.mean()


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
@pytest.mark.parametrize("engine", ["gen", "interp"])
def test_vmapped_mixture_is_two_device_sites_and_matches_the_oracle(rng, engine, monkeypatch):
    """N = 4096 data: 2 device sites (8192 logical ones: far beyond the 1023 site numbers of the FLAT layout), a GENERATED
    kernel with one instance loop (and the site interpreter), both stream layouts — JAX32 under the reference's instance-key
    rule split(key, n)[i] (vmap.py:186, 201)"""
    from genjax_amd import kernels
    from oracle import cpu
    N, K = 4096, 2048
    model, chm, ys, mu, logits = _mixture(N)
    prog, _, _ = model.pack((), chm, True, rng_mode=rng)
    assert prog.n_sites == 2 <= 4 and len(prog.site_list.sites) == 2 * N and prog.n_slots == N
    assert all(prog.c_sites[j].plate == 1 and prog.c_sites[j].plate_n == N for j in range(2))
    if engine == "gen":
        assert kernels.program_engine(prog) == 4
        src = kernels.program_source(prog, 2)
        assert "for (int i_ = 0; i_ < %d; ++i_)" % N in src
    else:
        monkeypatch.setenv("GJX_ENGINE", "interp")
    out = kernels.run_program(prog, (0, 11), K, want_site_scores=True)
    ora = cpu.run_program(prog, (0, 11), K, want_margin=True, want_site_scores=True)
    _check_against_oracle(prog, out, ora, discrete_rows=range(N))
    ss, so = _np(out["site_scores"]), ora["site_scores"]
    same = (_np(out["choices"]) == ora["choices"]).all(axis=0)
    np.testing.assert_allclose(ss[:, same], so[:, same], rtol=3e-4, atol=2e-3)          # one row per BODY site: sums over the instances
    # the weights are the mixture's: log w = sum_i log N(y_i; mu[z_i], s) — recomputed in float64 from the device's own z
    z = _np(out["choices"])[:N].astype(int)
    lw = (-0.5 * ((ys[:, None] - mu[z]) / 0.7) ** 2 - np.log(0.7) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(_np(out["weight"]), lw, rtol=2e-5, atol=0.02)
    # and z follows softmax(logits)
    p = np.exp(logits - logits.max())
    p /= p.sum()
    assert np.abs(np.bincount(z.ravel(), minlength=3) / z.size - p).max() < 2e-3


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
@pytest.mark.parametrize("lpp", [4, 16])
def test_few_particles_over_very_many_instances_take_several_lanes_per_particle(rng, lpp, monkeypatch):
    """K = 512 particles over N = 8192 data: with one lane per particle the launch has 8 blocks.  The wide flavour deals a particle's
    instances to 4 or 16 LANES as well as to the 16 waves of its block (a wave holds 64 / lpp particles; 16 x lpp chunks of instances
    whose partial sums meet in LDS in chunk order): same values as the oracle, every instance's row written by the lane that drew it"""
    from genjax_amd import kernels
    from oracle import cpu
    N, K = 8192, 512
    model, chm, ys, mu, logits = _mixture(N, seed=2)
    prog, _, _ = model.pack((), chm, True, rng_mode=rng)
    monkeypatch.setenv("GJX_ENGINE", "gen")
    monkeypatch.setenv("GJX_GEN_WIDE", "1")
    monkeypatch.setenv("GJX_GEN_LPP", str(lpp))
    out = kernels.run_program(prog, (0, 13), K, want_site_scores=True)
    assert out["_engine"] == 4
    ora = cpu.run_program(prog, (0, 13), K, want_margin=True, want_site_scores=True)
    _check_against_oracle(prog, out, ora, discrete_rows=range(N))
    z = _np(out["choices"])[:N].astype(int)
    lw = (-0.5 * ((ys[:, None] - mu[z]) / 0.7) ** 2 - np.log(0.7) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(_np(out["weight"]), lw, rtol=2e-5, atol=0.03)
    same = (_np(out["choices"]) == ora["choices"]).all(axis=0)
    np.testing.assert_allclose(_np(out["site_scores"])[:, same], ora["site_scores"][:, same], rtol=3e-4, atol=4e-3)
    np.testing.assert_allclose(float(out["lse"][2]), float(np.log(np.exp(ora["weight"].astype(np.float64) - ora["weight"].max()).sum()) + ora["weight"].max()), rtol=1e-5, atol=5e-3)
    monkeypatch.delenv("GJX_GEN_LPP")
    monkeypatch.delenv("GJX_GEN_WIDE")
    assert "#define LPP_ %d" % lpp in kernels.program_source(prog, 1 | 512 | (1024 if lpp == 4 else 2048))


def test_plate_through_the_inference_layer():
    """ImportanceK on the vmapped mixture (smc.py:298-315): log-ML against the closed form prod_i sum_c pi_c N(y_i; mu_c, s)"""
    from genjax_amd.inference import ImportanceK, Target
    N = 512
    model, chm, ys, mu, logits = _mixture(N, seed=3)
    p = np.exp(logits - logits.max())
    p /= p.sum()
    dens = p[None, :] * np.exp(-0.5 * ((ys[:, None] - mu[None, :]) / 0.7) ** 2) / (0.7 * np.sqrt(2 * np.pi))
    exact = float(np.log(dens.sum(axis=1)).sum())
    # the prior proposal is hopeless at N = 512 (the estimator's relative variance grows like 3^N for separated components), so
    # the estimate itself is checked on a 12-datum model with overlapping components (relative variance ~ 1.1^12) to rtol 1e-3
    N2 = 12
    model2, chm2, ys2, mu2, _ = _mixture(N2, seed=4, mu=(-0.3, 0.0, 0.3))
    dens2 = p[None, :] * np.exp(-0.5 * ((ys2[:, None] - mu2[None, :]) / 0.7) ** 2) / (0.7 * np.sqrt(2 * np.pi))
    exact2 = float(np.log(dens2.sum(axis=1)).sum())
    est = float(ImportanceK(Target(model2, (), chm2), k_particles=1 << 20).log_marginal_likelihood_estimate(genjax.key(5)))
    assert est == pytest.approx(exact2, rel=1e-3)
    # and the big one runs (no refusal for > 1023 unrolled sites) and is a lower bound in expectation
    tr, w = model.importance(genjax.key(1), chm, (), K=4096)
    assert tr.prog.n_sites == 2 and np.isfinite(_np(w)).all() and float(_np(w).max()) < exact + 50.0
    got = tr.get_choices()
    assert got["k", 7, "z"].shape == (4096,) and tuple(got["k", :, "z"].shape) == (4096, N)


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
@pytest.mark.parametrize("engine", ["gen", "interp"])
def test_downstream_reader_and_chained_vmaps_match_the_oracle(rng, engine, monkeypatch):
    """a site outside the plate reads ONE instance (zs[3]); a second vmap reads instance i of the first: both used to fail to
    pack with plates on; now the reads go to the plate's rows.  (JAX32: the site behind the two-site plate takes the caller's
    NEXT counter — a Vmap call is one traced site —, in both engines as in the oracle)"""
    from genjax_amd import kernels
    monkeypatch.setenv("GJX_ENGINE", engine)
    from oracle import cpu
    n = 24
    xs = np.linspace(-1.0, 1.0, n).astype(np.float32)

    @genjax.gen
    def k1(x):
        return genjax.normal(x, 1.0) @ "z"

    @genjax.gen
    def k2(z):
        c = genjax.categorical(logits=np.array([0.0, 0.5], np.float32)) @ "c"
        return genjax.normal(z, genjax.take(np.array([0.5, 1.5], np.float32), c)) @ "w"

    @genjax.gen
    def model():
        zs = k1.vmap()(xs) @ "zs"
        ws = k2.vmap()(zs) @ "ws"
        y = genjax.normal(zs[3], 1.0) @ "y"
        return y

    K = 3000
    prog, _, _ = model.pack((), C.n(), True, rng_mode=rng)
    assert prog.n_sites == 4 and len(prog.site_list.sites) == 3 * n + 1
    out = kernels.run_program(prog, (0, 4), K)
    ora = cpu.run_program(prog, (0, 4), K, want_margin=True)
    _check_against_oracle(prog, out, ora)
    ch = _np(out["choices"])
    z3, y = ch[prog.slot_of[(("zs", "z"), 3)]], ch[prog.slot_of["y"]]
    assert abs(np.corrcoef(z3, y)[0, 1] - 1.0 / np.sqrt(2.0)) < 0.05        # y ~ N(z_3, 1), z_3 ~ N(x_3, 1)
    tr = model.simulate(genjax.key(2), (), K=100)
    assert tr.prog.n_sites <= 4


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_plate_with_per_instance_masks_tables_and_affine_rows(rng):
    """per-instance DATA of every kind: Mask(value, flag) constraints per instance and particle, a gather table per instance,
    an affine row per instance over TWO latent sites outside the plate (hierarchical regression with an intercept)"""
    from genjax_amd import kernels
    from genjax_amd.core import Mask
    import torch
    from oracle import cpu
    n, P, K = 40, 3, 2500
    rs = np.random.default_rng(5)
    X = rs.standard_normal((n, P)).astype(np.float32)
    tabs = rs.standard_normal((n, 2)).astype(np.float32)

    @genjax.gen
    def kernel(x_row, tab, beta, b0):
        c = genjax.flip(0.4) @ "c"
        m = genjax.normal(genjax.take(tab, c), 1.0) @ "m"
        return genjax.normal(x_row @ beta + b0, 0.8) @ "y"

    @genjax.gen
    def model():
        beta = genjax.normal(np.zeros(P, np.float32), 1.0) @ "beta"
        b0 = genjax.normal(0.0, 2.0) @ "b0"
        kernel.vmap(in_axes=(0, 0, None, None))(X, tabs, beta, b0) @ "k"

    yobs = rs.standard_normal(n).astype(np.float32)
    mvals = rs.standard_normal((K, n)).astype(np.float32)
    flags = rs.uniform(size=(K, n)) < 0.5
    chm = C["k", "y"].set(yobs)
    for i in range(n):                       # Mask(value, flag) per instance AND particle (distribution.py:129-143)
        chm = chm | C["k", i, "m"].set(Mask(mvals[:, i], flags[:, i]))
    tr, w = model.importance(genjax.key(9), chm, (), K=K)
    prog = tr.prog
    assert prog.n_sites == 5 and sum(1 for j in range(5) if prog.c_sites[j].plate) == 3
    # the oracle on the same inputs: rebuild the input rows (constraint values and flags) exactly as the host laid them out
    ch_in = np.zeros((prog.n_slots, K), np.float32)
    for i in range(n):
        ch_in[prog.slot_of[(("k", "m"), i)]] = mvals[:, i]
        ch_in[prog.flag_slot_of[(("k", "m"), i)]] = flags[:, i]
    ora = cpu.run_program(prog, genjax.key(9), K, choices=ch_in, want_margin=True)
    out = dict(choices=tr.choices, score=tr.get_score(), weight=w)
    _check_against_oracle(prog, out, ora)
    m_dev = _np(tr.get_choices()["k", :, "m"])
    assert (m_dev[flags] == mvals[flags]).all() and not (m_dev[~flags] == mvals[~flags]).any()      # constrained exactly where the flag is set


def test_flat_plate_draws_equal_the_vector_form():
    """the plate form draws at the elements a vector site of n * dim elements would use: a plate that ALSO has a vector form
    (normals only) gives the same particles either way, bit for bit on the integer side of the streams"""
    from genjax_amd import kernels
    n = 32

    @genjax.gen
    def kernel(mu):
        z = genjax.normal(mu, 1.0) @ "z"
        return genjax.normal(z, 0.5) @ "y"

    @genjax.gen
    def model():
        mu = genjax.normal(0.0, 2.0) @ "mu"
        kernel.repeat(n=n)(mu) @ "k"

    ys = np.linspace(-1, 1, n).astype(np.float32)
    chm = C["k", "y"].set(ys)
    prog_v, _, _ = model.pack((), chm, True)                      # vector form (3 plain sites)
    assert prog_v.n_sites == 3 and all(prog_v.c_sites[j].plate == 0 for j in range(3))
    import genjax_amd.program as P
    orig = P._try_compact
    try:
        P._try_compact = lambda *a, **k: None                      # force the plate form
        model._pack_cache = {}
        prog_p, _, _ = model.pack((), chm, True)
    finally:
        P._try_compact = orig
        model._pack_cache = {}
    assert prog_p.n_sites == 3 and prog_p.c_sites[1].plate == 1
    a = kernels.run_program(prog_v, (0, 3), 4096)
    b = kernels.run_program(prog_p, (0, 3), 4096)
    np.testing.assert_allclose(_np(a["choices"]), _np(b["choices"]), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(_np(a["weight"]), _np(b["weight"]), rtol=1e-5, atol=1e-4)


def _regression_with_a_latent_per_datum(N, P, rng_mode, select_eta=False):
    """ls ~ N(0,1); beta ~ N(0, I_P); per datum (a vmapped kernel of TWO sites): eta_i ~ N(x_i . beta, exp(ls)), y_i ~ bernoulli(logits = eta_i);
    y observed, everything else constrained per chain; HMC moves (ls, beta) — or, `select_eta`, the N latents as well"""
    import genjax_amd as genjax
    rs = np.random.default_rng(0)
    X = (0.5 * rs.standard_normal((N, P))).astype(np.float32)

    @genjax.gen
    def kern(x_row, beta, ls):
        eta = genjax.normal(x_row @ beta, genjax.exp(ls)) @ "eta"
        return genjax.bernoulli(logits=eta) @ "y"

    @genjax.gen
    def model():
        ls = genjax.normal(0.0, 1.0) @ "ls"
        beta = genjax.normal(np.zeros(P, np.float32), 1.0) @ "beta"
        kern.vmap(in_axes=(0, None, None))(X, beta, ls) @ "k"

    y = (rs.uniform(size=N) < 0.5).astype(np.float32)
    lat = ["ls", "beta"] + [(("k", "eta"), i) for i in range(N)]
    sel = ["ls", "beta"] + ([(("k", "eta"), i) for i in range(N)] if select_eta else [])
    prog, _, _ = model.pack((), C["k", "y"].set(y), False, selected=tuple(sel), per_particle=tuple(lat), plates="hmc", rng_mode=rng_mode)
    return prog


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_score_grad_through_a_plate_matches_the_oracle(rng):
    """selection_gradient over a vmapped kernel (hmc.py:70-96 differentiates any assess; vmap.py:363-376): gjx_score_grad on a
    PLATE-TAGGED program — a two-site body, the instance's latent constrained per chain — against the oracle: the score and the
    gradient rows of the selected sites outside the plate AND of the per-instance latents inside it"""
    import torch
    from genjax_amd import kernels
    from oracle import cpu
    N, P, n = 96, 4, 300
    prog = _regression_with_a_latent_per_datum(N, P, rng, select_eta=True)
    assert prog.n_sites == 4 and prog.c_sites[2].plate == 1 and prog.c_sites[3].plate == 1 and prog.n_slots == 1 + P + N
    ch = (np.random.default_rng(3).standard_normal((prog.n_slots, n)) * 0.4).astype(np.float32)
    gs, gg = kernels.score_grad(prog, torch.as_tensor(ch).cuda())
    os_, og = cpu.score_grad(prog, ch)
    np.testing.assert_allclose(_np(gs), os_, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(_np(gg), og, rtol=2e-3, atol=2e-3)
    assert np.abs(og[1 + P:]).max() > 0.1 and np.abs(og[:1 + P]).max() > 0.1


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_hmc_over_a_plate_tagged_program_generated_kernel_interpreter_and_oracle(rng, monkeypatch):
    """HMC.edit over (ls, beta) of a regression with a latent per datum — a vmapped kernel with a TWO-site body, 256 instances —
    on the kernel GENERATED for the plate-tagged program (the instances dealt to the four lanes of a chain, per-chain rows of the
    body read instance by instance), on the site interpreter and in the oracle: same streams, trajectories by the tolerances of
    tests/test_gpu_hmcgen.py.  No HMC program is refused for being plate-tagged."""
    import torch
    from genjax_amd import kernels
    from oracle import cpu
    N, P, n = 256, 4, 400
    prog = _regression_with_a_latent_per_datum(N, P, rng)
    assert prog.c_sites[2].plate == 1 and prog.c_sites[2].mode == A.MODE_OBS_SLOT and prog.c_sites[3].mode == A.MODE_OBS_TAB
    ch = (np.random.default_rng(5).standard_normal((prog.n_slots, n)) * 0.3).astype(np.float32)
    src = kernels.program_hmc_source(prog)
    assert "plate of 256 instances x 2 sites" in src and "for (int i_ = q_; i_ < 256; i_ += CPL)" in src
    for stale, accept in ((False, False), (True, False), (False, True)):
        e, L = 0.004 * (4 if accept else 1), 15
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        assert kernels.hmc_engine(prog) == 4
        g = kernels.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), e, L, stale, accept, offset=11)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        assert kernels.hmc_engine(prog) == 0
        it = kernels.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), e, L, stale, accept, offset=11)
        o = cpu.hmc(prog, (2, 9), ch, e, L, stale, accept, offset=11)
        gc, ic = _np(g["choices"]), _np(it["choices"])
        assert np.isfinite(gc).all() and np.isfinite(_np(g["alpha"])).all()
        np.testing.assert_array_equal(gc[1 + P:], ch[1 + P:])                      # the per-datum latents are not moved
        if not accept:
            np.testing.assert_allclose(gc, ic, rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(_np(g["alpha"]), _np(it["alpha"]), rtol=5e-3, atol=5e-3)
            np.testing.assert_allclose(gc, o["choices"], rtol=3e-3, atol=3e-3)
            np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=6e-3, atol=6e-3)
            np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=1e-3, atol=6e-3)
            assert np.abs(gc[:1 + P] - ch[:1 + P]).max() > 1e-3
        else:
            acc_g, acc_o = _np(g["accepted"]) > 0.5, o["accepted"] > 0.5
            assert (acc_g != acc_o).mean() < 0.02
            assert 0.3 < acc_g.mean() <= 1.0
    monkeypatch.delenv("GJX_HMC_ENGINE")


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_hmc_moves_the_latents_inside_a_plate(rng, monkeypatch):
    """the N per-datum latents SELECTED as well (one momentum leaf for the whole vmapped address, hmc.py:120-130): chain state of
    1 + P + N values.  The generated kernel keeps the plate's rows — positions, momenta, gradients — in the caller's workspace,
    read and written by the lane that owns the instance; the site interpreter keeps all state in memory; both against the oracle,
    with the stale-gradient compatibility mode and with the accept step"""
    import torch
    from genjax_amd import kernels
    from oracle import cpu
    N, P, n = 64, 3, 300                                                         # (300 chains: the last block has idle quads)
    prog = _regression_with_a_latent_per_datum(N, P, rng, select_eta=True)
    ch = (np.random.default_rng(8).standard_normal((prog.n_slots, n)) * 0.3).astype(np.float32)
    src = kernels.program_hmc_source(prog)
    assert "// PROWS 64" in src and "wg_[(int64_t)(0 + i_ * 1 + 0) * n_ + ic_] = ga[" in src
    for stale, accept in ((False, False), (True, False), (False, True)):
        eps = 0.03 if accept else 0.01
        o = cpu.hmc(prog, (4, 1), ch, eps, 10, stale, accept, offset=3)
        for engine, code in (("gen", 4), ("interp", 0)):
            monkeypatch.setenv("GJX_HMC_ENGINE", engine)
            assert kernels.hmc_engine(prog) == code
            g = kernels.hmc(prog, (4, 1), torch.as_tensor(ch).cuda(), eps, 10, stale, accept, offset=3)
            gc = _np(g["choices"])
            if not accept:
                np.testing.assert_allclose(gc, o["choices"], rtol=3e-3, atol=3e-3)
                np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=6e-3, atol=6e-3)
                np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=1e-3, atol=6e-3)
                assert np.abs(gc[1 + P:] - ch[1 + P:]).max() > 1e-3             # the latents inside the plate moved
            else:
                acc_g, acc_o = _np(g["accepted"]) > 0.5, o["accepted"] > 0.5
                assert (acc_g != acc_o).mean() < 0.02 and 0.2 < acc_g.mean() < 1.0
                same = acc_g == acc_o
                np.testing.assert_allclose(gc[:, same], o["choices"][:, same], rtol=3e-3, atol=3e-3)
                np.testing.assert_array_equal(gc[:, ~acc_g], ch[:, ~acc_g])      # a rejected chain keeps every row, the plate's included
    monkeypatch.delenv("GJX_HMC_ENGINE")


def _mixture_with_latent_means(N, seed=0):
    """the Bayesian mixture: mu ~ N(0, 3^2 I_3), ls ~ N(0, 1); per datum z_i ~ categorical(logits), x_i ~ normal(mu[z_i], exp(ls)) — the
    component mean is a row of a LATENT choice picked by a discrete choice (GJX_P_VGATHER)"""
    rs = np.random.default_rng(seed)
    true_mu = np.array([-2.5, 0.0, 3.0], np.float32)
    ztrue = rs.integers(0, 3, N)
    ys = (true_mu[ztrue] + 0.6 * rs.standard_normal(N)).astype(np.float32)
    logits = np.array([0.2, -0.3, 0.1], np.float32)

    @genjax.gen
    def kern(mu, ls, lg):
        z = genjax.categorical(logits=lg) @ "z"
        return genjax.normal(mu[z], genjax.exp(ls)) @ "x"

    @genjax.gen
    def model():
        mu = genjax.normal(np.zeros(3, np.float32), 3.0) @ "mu"
        ls = genjax.normal(0.0, 1.0) @ "ls"
        kern.repeat(n=N)(mu, ls, logits) @ "k"

    return model, ys, logits, ztrue


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
@pytest.mark.parametrize("engine", ["gen", "interp"])
def test_mixture_with_latent_means_matches_the_oracle(rng, engine, monkeypatch):
    """importance over (mu, ls, z) with x observed: the x site reads mu[z_i] — one of the rows of an earlier latent choice — in the
    generated kernel (registers picked by a select chain) and in the interpreter, against the oracle and against numpy"""
    from genjax_amd import kernels
    from oracle import cpu
    N, K = 300, 3000
    model, ys, logits, _ = _mixture_with_latent_means(N)
    prog, _, _ = model.pack((), C["k", "x"].set(ys), True, rng_mode=rng)
    assert prog.n_sites == 4 and prog.c_sites[3].p[0].op == A.P_VGATHER and (prog.c_sites[3].p[0].moff, prog.c_sites[3].p[0].n, prog.c_sites[3].p[0].d_slot) == (0, 3, 1)
    monkeypatch.setenv("GJX_ENGINE", engine)
    assert kernels.program_engine(prog) == (4 if engine == "gen" else 0)
    out = kernels.run_program(prog, (0, 4), K)
    ora = cpu.run_program(prog, (0, 4), K, want_margin=True)
    _check_against_oracle(prog, out, ora)
    ch = _np(out["choices"])
    mu, ls, z = ch[:3], ch[3], ch[4:4 + N].astype(int)
    lx = (-0.5 * ((ys[:, None] - np.take_along_axis(mu, z, axis=0)) / np.exp(ls)) ** 2 - ls - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(_np(out["weight"]), lx, rtol=3e-4, atol=3e-2)


@pytest.mark.parametrize("engine", ["gen", "interp"])
def test_mixture_with_latent_means_and_one_assignment_for_every_particle(engine, monkeypatch):
    """importance over (mu, ls) with x AND z observed (one assignment vector for every particle: z owns no storage, the index of the
    row gather comes from the table, GJX_P_VGATHER with slot = -1) — both engines against the oracle and against numpy"""
    from genjax_amd import kernels
    from oracle import cpu
    N, K = 500, 4096
    model, ys, logits, ztrue = _mixture_with_latent_means(N, seed=5)
    prog, _, _ = model.pack((), C["k", "x"].set(ys) | C["k", "z"].set(ztrue.astype(np.float32)), True)
    q = prog.c_sites[3].p[0]
    assert prog.n_slots == 4 and (q.op, q.slot, q.d_off) == (A.P_VGATHER, -1, 1)
    monkeypatch.setenv("GJX_ENGINE", engine)
    assert kernels.program_engine(prog) == (4 if engine == "gen" else 0)
    out = kernels.run_program(prog, (0, 6), K)
    ora = cpu.run_program(prog, (0, 6), K)
    np.testing.assert_allclose(_np(out["choices"]), ora["choices"], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(_np(out["weight"]), ora["weight"], rtol=3e-4, atol=3e-2)
    ch = _np(out["choices"])
    mu, ls = ch[:3], ch[3]
    lz = (logits - np.log(np.exp(logits).sum()))[ztrue].sum()
    lx = (-0.5 * ((ys[:, None] - mu[ztrue]) / np.exp(ls)) ** 2 - ls - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(_np(out["weight"]), lx + lz, rtol=3e-4, atol=5e-2)


@pytest.mark.parametrize("z_shared", [False, True])
def test_hmc_over_the_means_and_log_sigma_of_the_4096_datum_mixture(z_shared, monkeypatch):
    """HMC.edit over (mu, ls) of the 4096-datum vmapped mixture with the assignments z fixed — per chain (rows of the chain), or one
    assignment for every chain (the index of the row gather then comes from the table) —: a generated kernel (the plate loop dealt to
    the lanes of a chain, select chains forwards and for the gradient), the interpreter and the oracle follow the same trajectories;
    gjx_score_grad's gradient rows against the oracle"""
    import torch
    from genjax_amd import kernels
    from oracle import cpu
    N, n = 4096, 256
    model, ys, logits, ztrue = _mixture_with_latent_means(N, seed=3)
    rs = np.random.default_rng(7)
    if z_shared:
        zfix = ztrue.astype(np.float32)
        prog, _, _ = model.pack((), C["k", "x"].set(ys) | C["k", "z"].set(zfix), False, selected=("mu", "ls"), per_particle=("mu", "ls"), plates="hmc")
        assert prog.n_slots == 4 and prog.c_sites[3].p[0].op == A.P_VGATHER and prog.c_sites[3].p[0].slot == -1
    else:
        lat = ["mu", "ls"] + [(("k", "z"), i) for i in range(N)]
        prog, _, _ = model.pack((), C["k", "x"].set(ys), False, selected=("mu", "ls"), per_particle=tuple(lat), plates="hmc")
        assert prog.n_slots == 4 + N and prog.c_sites[3].p[0].slot == 4
    assert prog.n_sites == 4 and prog.c_sites[2].plate == 1 and prog.c_sites[3].plate == 1
    ch = np.zeros((prog.n_slots, n), np.float32)
    # chains start near the mode (the posterior of 4096 data is narrow: sd of a mean about 0.016), a few assignments wrong per chain
    ch[:3] = np.array([-2.5, 0.0, 3.0], np.float32)[:, None] + 0.03 * rs.standard_normal((3, n))
    ch[3] = np.log(0.6) + 0.02 * rs.standard_normal(n)
    if not z_shared:
        ch[4:] = np.where(rs.uniform(size=(N, n)) < 0.01, rs.integers(0, 3, (N, n)), ztrue[:, None])
    gs, gg = kernels.score_grad(prog, torch.as_tensor(ch).cuda())
    os_, og = cpu.score_grad(prog, ch)
    np.testing.assert_allclose(_np(gs), os_, rtol=3e-4, atol=5e-2)
    np.testing.assert_allclose(_np(gg)[:4], og[:4], rtol=3e-3, atol=5e-2)
    assert np.abs(og[:4]).max() > 10.0
    src = kernels.program_hmc_source(prog)
    assert "plate of 4096 instances x 2 sites" in src and "gi_3_0 == 2 ? w_ : 0.0f" in src
    eps, L = 1e-3, 12
    outs = {}
    for engine in ("gen", "interp"):
        monkeypatch.setenv("GJX_HMC_ENGINE", engine)
        assert kernels.hmc_engine(prog) == (4 if engine == "gen" else 0)
        outs[engine] = kernels.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=5)
    monkeypatch.delenv("GJX_HMC_ENGINE")
    o = cpu.hmc(prog, (2, 9), ch, eps, L, False, False, offset=5)
    g, it = _np(outs["gen"]["choices"]), _np(outs["interp"]["choices"])
    np.testing.assert_allclose(g[:4], it[:4], rtol=1e-3, atol=3e-4)
    np.testing.assert_allclose(g[:4], o["choices"][:4], rtol=1e-3, atol=3e-4)
    # (alpha is a difference of two scores of size 4e3: a position error of 3e-4 under a gradient of 1e3 per unit moves it by 0.3)
    np.testing.assert_allclose(_np(outs["gen"]["alpha"]), o["alpha"], rtol=2e-2, atol=0.3)
    np.testing.assert_allclose(_np(outs["gen"]["alpha"]), _np(outs["interp"]["alpha"]), rtol=2e-2, atol=0.3)
    np.testing.assert_array_equal(g[4:], ch[4:])
    assert np.abs(g[:4] - ch[:4]).max() > 3e-3


def test_config5_written_with_vmap_and_a_two_site_body_runs_on_a_generated_hmc_kernel(monkeypatch):
    """config 5 as a user writes it — `kernel.vmap()(X)` — with TWO observed sites per datum (a binary and a real-valued response
    of the same linear predictor): the HMC packing keeps table-valued plates in the vector form, so both likelihood sites are big
    affine sites of one generated kernel (engine 4, the matrix-core flavour for the contraction); against the oracle"""
    import torch
    import genjax_amd as genjax
    from genjax_amd import kernels
    from oracle import cpu
    N, P, n = 1024, 16, 512
    rs = np.random.default_rng(0)
    X = rs.standard_normal((N, P)).astype(np.float32)

    @genjax.gen
    def kern(x_row, beta):
        genjax.bernoulli(logits=x_row @ beta) @ "y"
        genjax.normal(x_row @ beta, 2.0) @ "w"

    @genjax.gen
    def model():
        lt = genjax.normal(0.0, 1.0) @ "log_tau"
        beta = genjax.normal(np.zeros(P, np.float32), genjax.exp(lt)) @ "beta"
        kern.vmap(in_axes=(0, None))(X, beta) @ "k"

    y = (rs.uniform(size=N) < 0.5).astype(np.float32)
    w = rs.standard_normal(N).astype(np.float32)
    prog, _, _ = model.pack((), C["k", "y"].set(y) | C["k", "w"].set(w), False, selected=("log_tau", "beta"), per_particle=("log_tau", "beta"), plates="hmc")
    assert prog.n_sites == 4 and all(prog.c_sites[j].plate == 0 for j in range(4)) and prog.c_sites[2].dim == N and prog.c_sites[3].dim == N
    monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
    assert kernels.hmc_engine(prog) == 4
    ch = (rs.standard_normal((1 + P, n)) * 0.1).astype(np.float32)
    g = kernels.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), 0.002, 20, False, False)
    o = cpu.hmc(prog, (2, 9), ch, 0.002, 20, False, False)
    np.testing.assert_allclose(_np(g["choices"]), o["choices"], rtol=3e-3, atol=3e-3)
    np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=6e-3, atol=2e-2)


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_hmc_and_score_grad_read_input_sites_as_values(rng, monkeypatch):
    """a kernel's argument / a Scan step's carry as a GJX_MODE_INPUT site in an HMC target: x | x_prev ~ N(A x_prev, q) moved given
    the per-chain x_prev, y | x observed — the INPUT rows are read like any per-chain value, carry no density and no gradient row.
    Generated kernel, interpreter and oracle; the same target with x_prev constrained per chain (OBS_SLOT under a flat prior of
    its own) gives the same gradient of x"""
    import torch
    from genjax_amd import kernels
    from genjax_amd.program import PackedProgram, Param, SiteList
    from oracle import cpu
    dx, n = 4, 300
    rs = np.random.default_rng(2)
    Am = (0.5 * rs.standard_normal((dx, dx))).astype(np.float32)
    y = rs.standard_normal(dx).astype(np.float32)
    sl = SiteList()
    sl.add("xp", A.MVNORMAL_DIAG, [np.zeros(dx, np.float32), np.ones(dx, np.float32)])
    sl.add("x", A.MVNORMAL_DIAG, [Param.affine(Am, "xp"), np.full(dx, 0.7, np.float32)])
    sl.add("y", A.MVNORMAL_DIAG, [Param.value("x", dx), np.full(dx, 0.5, np.float32)])
    prog = PackedProgram(sl, {"xp": A.MODE_INPUT, "x": A.MODE_OBS_SLOT, "y": A.MODE_OBS_TAB}, {"y": y}, selected=("x",), rng_mode=rng)
    assert prog.c_sites[0].mode == A.MODE_INPUT and prog.n_slots == 2 * dx
    ch = (0.5 * rs.standard_normal((prog.n_slots, n))).astype(np.float32)
    gs, gg = kernels.score_grad(prog, torch.as_tensor(ch).cuda())
    os_, og = cpu.score_grad(prog, ch)
    np.testing.assert_allclose(_np(gs), os_, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(_np(gg), og, rtol=2e-3, atol=2e-3)
    assert not og[:dx].any() and np.abs(og[dx:]).max() > 0.1                  # the input rows have no gradient row
    xp, x = ch[:dx].astype(np.float64), ch[dx:].astype(np.float64)
    want = (-0.5 * ((x - Am.astype(np.float64) @ xp) / 0.7) ** 2 - np.log(0.7) - 0.5 * np.log(2 * np.pi)).sum(axis=0) \
        + (-0.5 * ((y[:, None] - x) / 0.5) ** 2 - np.log(0.5) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(os_, want, rtol=2e-5, atol=2e-3)               # the input's own "density" is no part of the score
    for engine, code in (("gen", 4), ("interp", 0)):
        monkeypatch.setenv("GJX_HMC_ENGINE", engine)
        assert kernels.hmc_engine(prog) == code
        g = kernels.hmc(prog, (3, 8), torch.as_tensor(ch).cuda(), 0.05, 10, False, False, offset=2)
        o = cpu.hmc(prog, (3, 8), ch, 0.05, 10, False, False, offset=2)
        gc = _np(g["choices"])
        np.testing.assert_array_equal(gc[:dx], ch[:dx])
        np.testing.assert_allclose(gc, o["choices"], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=5e-3, atol=5e-3)
        assert np.abs(gc[dx:] - ch[dx:]).max() > 1e-2
    monkeypatch.delenv("GJX_HMC_ENGINE")


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_random_vmapped_models_hmc_generated_interpreter_oracle(rng, monkeypatch):
    """differential test of the plate paths of the HMC emitter: random two- and three-site vmapped kernels — the per-datum latent's
    family, what its location and scale read (an affine form of the outside coefficients and the datum's row, a row of a latent vector
    picked by a per-datum categorical, a shared log-scale), the observation's family, the number of data (never a multiple of the
    lanes), whether the per-datum latents are moved too, the lanes per chain — generated kernel == site interpreter == oracle"""
    import torch
    from genjax_amd import kernels
    from oracle import cpu
    rs = np.random.default_rng(int(os.environ.get("GJX_FUZZ_SEED", "77")) + rng)
    trials, covered = int(os.environ.get("GJX_FUZZ_TRIALS", "10")), 0
    for trial in range(trials):
        N = int(rs.choice([13, 37, 70, 131, 300]))
        P = int(rs.choice([2, 3, 5]))
        lat_kind = str(rs.choice(["normal", "laplace", "cauchy", "gumbel", "student_t"]))
        obs_kind = str(rs.choice(["bernoulli", "normal", "poisson", "laplace"]))
        use_mix = bool(rs.integers(2))                    # location = a row of a latent 3-vector picked by a per-datum categorical
        move_eta = bool(rs.integers(2))
        cpl = str(rs.choice(["4", "16", "64"]))
        X = (0.4 * rs.standard_normal((N, P))).astype(np.float32)
        lg = np.array([0.3, -0.2, 0.1], np.float32)

        @genjax.gen
        def kern(x_row, beta, mu, ls):
            if use_mix:
                z = genjax.categorical(logits=lg) @ "z"
                loc = mu[z]
            else:
                loc = x_row @ beta
            sc = genjax.exp(ls)
            if lat_kind == "normal":
                eta = genjax.normal(loc, sc) @ "eta"
            elif lat_kind == "laplace":
                eta = genjax.laplace(loc, sc) @ "eta"
            elif lat_kind == "cauchy":
                eta = genjax.cauchy(loc, sc) @ "eta"
            elif lat_kind == "gumbel":
                eta = genjax.gumbel(loc, sc) @ "eta"
            else:
                eta = genjax.student_t(4.0, loc, sc) @ "eta"
            if obs_kind == "bernoulli":
                genjax.bernoulli(logits=eta) @ "y"
            elif obs_kind == "normal":
                genjax.normal(eta, 0.8) @ "y"
            elif obs_kind == "poisson":
                genjax.poisson(genjax.exp(0.3 * eta)) @ "y"
            else:
                genjax.laplace(eta, 0.7) @ "y"

        @genjax.gen
        def model():
            beta = genjax.normal(np.zeros(P, np.float32), 1.0) @ "beta"
            mu = genjax.normal(np.zeros(3, np.float32), 2.0) @ "mu"
            ls = genjax.normal(-0.5, 0.3) @ "ls"
            kern.vmap(in_axes=(0, None, None, None))(X, beta, mu, ls) @ "k"

        y = (rs.poisson(1.0, N) if obs_kind == "poisson" else (rs.uniform(size=N) < 0.5) if obs_kind == "bernoulli" else rs.standard_normal(N)).astype(np.float32)
        etas = [(("k", "eta"), i) for i in range(N)]
        zs = [(("k", "z"), i) for i in range(N)] if use_mix else []
        sel = ["beta", "mu", "ls"] + (etas if move_eta else [])
        try:
            prog, _, _ = model.pack((), C["k", "y"].set(y), False, selected=tuple(sel), per_particle=tuple(["beta", "mu", "ls"] + etas + zs), plates="hmc", rng_mode=rng)
        except Exception as e:                     # (a shape the host lowers differently: not this test's subject)
            print("trial", trial, "not packed:", type(e).__name__, e)
            continue
        if not any(prog.c_sites[j].plate for j in range(prog.n_sites)):
            continue
        n = 200
        ch = np.zeros((prog.n_slots, n), np.float32)
        ch[:] = 0.3 * rs.standard_normal((prog.n_slots, n))
        ch[prog.slot_of["ls"]] = -0.5 + 0.1 * rs.standard_normal(n)
        if use_mix:
            z0 = prog.slot_of[(("k", "z"), 0)]
            ch[z0:z0 + N] = rs.integers(0, 3, (N, n))
        what = f"trial {trial}: N={N} P={P} latent {lat_kind} obs {obs_kind} mix {use_mix} move_eta {move_eta} lanes {cpl}"
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        monkeypatch.setenv("GJX_HMC_GEN_CPL", cpl)
        if kernels.hmc_engine(prog) != 4:
            print(what, "-> not on a generated kernel")
            continue
        covered += 1
        eps, L = 2e-3, 6
        g = kernels.hmc(prog, (5, trial), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=1)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        it = kernels.hmc(prog, (5, trial), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=1)
        o = cpu.hmc(prog, (5, trial), ch, eps, L, False, False, offset=1)
        o2 = cpu.hmc(prog, (5, trial), ch, eps * 1.01, L, False, False, offset=1)
        with np.errstate(invalid="ignore"):
            well = np.isfinite(o["choices"]).all(0) & np.isfinite(o["alpha"]) & (np.abs(o2["choices"] - o["choices"]).max(0) < 1e-3) & (np.abs(o["alpha"]) < 0.5)
        assert well.mean() > 0.5, what
        gc, ic = _np(g["choices"]), _np(it["choices"])
        np.testing.assert_allclose(gc[:, well], ic[:, well], rtol=3e-3, atol=3e-3, err_msg=what + " generated vs interpreter")
        np.testing.assert_allclose(gc[:, well], o["choices"][:, well], rtol=3e-3, atol=3e-3, err_msg=what + " generated vs oracle")
        mag = 2e-5 * np.abs(o["score"])[well]
        assert (np.abs(_np(g["alpha"])[well] - o["alpha"][well]) <= 2e-2 + mag).all(), what + " alpha"
        if not move_eta:
            e0 = prog.slot_of[(("k", "eta"), 0)]
            np.testing.assert_array_equal(gc[e0:e0 + N], ch[e0:e0 + N])
    monkeypatch.delenv("GJX_HMC_ENGINE", raising=False)
    monkeypatch.delenv("GJX_HMC_GEN_CPL", raising=False)
    assert covered >= trials // 2, covered
