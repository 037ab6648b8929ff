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


def _check_against_oracle(prog, out, ora, discrete_rows=(), cap_particles=0.08):
    """values / scores against the oracle; a particle may differ only where one of the oracle's discrete decisions for it
    was a near tie (oracle `decide` margins), and discrete ELEMENTS may differ in at most 1e-4 of all elements"""
    ch, oc = _np(out["choices"]), ora["choices"]
    close = (np.abs(ch - oc) <= 5e-5 + 2e-4 * np.abs(oc)).all(axis=0)
    for k in ("score", "weight"):
        g, o = _np(out[k]), ora[k]
        close &= np.abs(g - o) <= 2e-3 + 3e-4 * np.abs(o)
    bad = ~close
    if bad.any():
        m = ora["margin"][bad]
        assert (m < NEAR_TIE).all(), f"{int((m >= NEAR_TIE).sum())} differing particles are not near ties (max margin {float(m.max()):.3g})"
        assert bad.mean() <= cap_particles, f"{bad.mean():.4f} of the particles differ"
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
def test_downstream_reader_and_chained_vmaps_match_the_oracle(rng):
    """a site outside the plate reads ONE instance (zs[3]); a second vmap reads instance i of the first: both used to fail to
    pack with plates on; now the reads go to the plate's rows"""
    from genjax_amd import kernels
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


def test_hmc_entry_points_refuse_plate_tagged_and_input_sites():
    """gjx_hmc / gjx_score_grad walk plain site lists: a plate-tagged program (or one with INPUT sites) is refused with
    GJX_EUNSUPPORTED instead of being read with the wrong rows (the host's HMC packs vmapped kernels in the vector form)"""
    import torch
    from genjax_amd import kernels
    from genjax_amd._lib import GjxError
    model, chm, ys, mu, logits = _mixture(64)
    prog, _, _ = model.pack((), chm, True)
    assert prog.c_sites[0].plate == 1
    ch = torch.zeros((max(prog.n_slots, 1), 32), device="cuda")
    with pytest.raises(GjxError, match="plate-tagged"):
        kernels.score_grad(prog, ch)
    # every site constrained (what gjx_hmc asks for first), still plate-tagged
    p2, _, _ = model.pack((), chm ^ C["k", "z"].set(np.zeros(64, np.float32)), True)
    assert p2.c_sites[0].plate == 1 and all(p2.c_sites[j].mode == A.MODE_OBS_TAB for j in range(p2.n_sites))
    with pytest.raises(GjxError, match="plate-tagged"):
        kernels.hmc(p2, (1, 2), torch.zeros((max(p2.n_slots, 1), 32), device="cuda"), 0.01, 2, False, False)
