"""Generated HMC kernels (csrc/gjx_codegen.hip `gjx_hmc_gen`, gjx_hmc engine 4): the kernel emitted from a program's site
list against the oracle's HMC.edit restatement (oracle/gjx_oracle.c, hmc.py:156-211) and against the site interpreter
(k_hmc_generic) on the same streams — same tolerances as the hand-written kernels' tests in test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H  # noqa: E402
from genjax_amd import _abi as A  # noqa: E402
from genjax_amd.program import PackedProgram, Param, SiteList  # noqa: E402

pytestmark = pytest.mark.gpu
RNGS = [A.RNG_FLAT, A.RNG_JAX32]


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


def _scan_target(T, rng):
    """an observed random-walk Scan of T steps with every x_t selected: x_t ~ N(x_{t-1}, 0.3), y_t ~ N(x_t, 0.5)"""
    prog, ys = H.scan_chain(T, rng=rng, carry=True, observe=True, sigma=0.3, r=0.5)
    sl = prog.site_list
    modes = {s.addr: (A.MODE_OBS_TAB if s.addr[0] == "y" else A.MODE_OBS_SLOT) for s in sl.sites}
    obs = {("y", t): ys[t] for t in range(T)}
    return PackedProgram(sl, modes, obs, selected=tuple(("x", t) for t in range(T)), rng_mode=rng), T


def _hierarchy_target(rng):
    sl = H.shape_hierarchy(rng)
    return PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=("la", "lb"), rng_mode=rng)


def _mixed_target(rng):
    """gather through a discrete (unselected) site, an affine over two selected vectors, transforms on value parameters"""
    rs = np.random.default_rng(4)
    sl = SiteList()
    sl.add("z", A.CATEGORICAL_LOGITS, [np.array([0.2, -0.3, 0.5], np.float32)])
    sl.add("mu", A.MVNORMAL_DIAG, [Param.gather(rs.standard_normal((3, 4)).astype(np.float32), "z"), Param.const(np.full(4, 0.8, np.float32))], dim=4)
    sl.add("ls", A.NORMAL, [-0.5, 0.3])
    sl.add("w", A.MVNORMAL_DIAG, [Param.value("mu", 4), Param.value("ls", xf=A.XF_EXP)], dim=4)
    sl.add("y", A.NORMAL, [Param.affine(rs.standard_normal((6, 4)).astype(np.float32), "w", bias=rs.standard_normal(6).astype(np.float32)),
                           Param.value("ls", xf=A.XF_SOFTPLUS)], dim=6)
    sl.add("k", A.POISSON, [Param.affine(np.abs(rs.standard_normal((1, 4))).astype(np.float32) * 0.2, "mu", bias=np.float32(1.0), xf=A.XF_SOFTPLUS)])
    modes = {"z": A.MODE_OBS_SLOT, "mu": A.MODE_OBS_SLOT, "ls": A.MODE_OBS_SLOT, "w": A.MODE_OBS_SLOT, "y": A.MODE_OBS_TAB, "k": A.MODE_OBS_TAB}
    obs = {"y": rs.standard_normal(6).astype(np.float32), "k": np.float32(2.0)}
    # (second program: the same sites with the latents sampled — a start state in which the discrete site holds valid values)
    sim = PackedProgram(sl, {"y": A.MODE_OBS_TAB, "k": A.MODE_OBS_TAB}, obs, rng_mode=rng)
    return PackedProgram(sl, modes, obs, selected=("mu", "ls", "w"), rng_mode=rng), sim


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("which", ["logreg_small", "logreg_cfg5", "hierarchy", "scan16", "mixed"])
def test_generated_hmc_kernel_against_oracle_and_interpreter(K_, oracle, rng, which, monkeypatch):
    import torch
    n = 500
    if which == "logreg_small":
        prog, _ = H.logreg(N=64, P=4, rng=rng)
        ch = (np.random.default_rng(2).standard_normal((5, n)) * 0.3).astype(np.float32)
        eps, L = 0.01, 25
    elif which == "logreg_cfg5":
        prog, _ = H.logreg(N=1024, P=16, rng=rng)
        ch = (np.random.default_rng(5).standard_normal((17, n)) * 0.2).astype(np.float32)
        eps, L = 0.004, 20
    elif which == "hierarchy":
        prog = _hierarchy_target(rng)
        ch = oracle.run_program(PackedProgram(H.shape_hierarchy(rng), rng_mode=rng), (3, 4), n)["choices"].astype(np.float32)
        eps, L = 0.002, 20
    elif which == "scan16":
        prog, T = _scan_target(16, rng)
        ch = (np.random.default_rng(6).standard_normal((16, n)) * 0.3).astype(np.float32)
        eps, L = 0.02, 15
    else:
        prog, sim = _mixed_target(rng)
        ch = oracle.run_program(sim, (3, 1), n)["choices"].astype(np.float32)
        eps, L = 0.01, 12
    monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
    assert K_.hmc_engine(prog) == 4, "the emitter does not cover this program"
    for stale, accept in ((False, False), (True, False), (False, True)):
        e = eps * (4 if accept else 1)
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        g = K_.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), e, L, stale, accept, offset=11)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        assert K_.hmc_engine(prog) == 0
        it = K_.hmc(prog, (2, 9), torch.as_tensor(ch).cuda(), e, L, stale, accept, offset=11)
        o = oracle.hmc(prog, (2, 9), ch, e, L, stale, accept, offset=11)
        gc, ic = _np(g["choices"]), _np(it["choices"])
        assert np.isfinite(gc).all() and np.isfinite(_np(g["alpha"])).all()
        if not accept:
            # generated vs interpreter: same streams, same arithmetic up to the order of the sums
            np.testing.assert_allclose(gc, ic, rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(_np(g["alpha"]), _np(it["alpha"]), rtol=5e-3, atol=5e-3)
            np.testing.assert_allclose(gc, o["choices"], rtol=3e-3, atol=3e-3)
            np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=6e-3, atol=6e-3)
            np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=1e-3, atol=6e-3)
            assert (_np(g["accepted"]) == 1).all()
        else:
            acc_g, acc_o = _np(g["accepted"]), o["accepted"]
            flip = acc_g != acc_o
            # a chain may decide differently from the oracle only if log u lies within the alpha tolerance of the boundary
            assert (o["margin"][flip] < 6e-3 + 6e-3 * np.abs(o["alpha"][flip])).all()
            assert flip.mean() < 0.03
            rej = acc_g == 0
            np.testing.assert_array_equal(gc[:, rej], ch[:, rej])              # rejected chains are untouched, bit for bit
            both = (acc_g == 1) & (acc_o == 1) & (np.abs(o["alpha"]) < 0.5)
            np.testing.assert_allclose(gc[:, both], o["choices"][:, both], rtol=3e-2, atol=3e-2)


def _regression_target(which, rng):
    """big likelihood sites whose AFFINE parameter the emitter puts on the matrix cores (hmc_emit_mfma_site): other row / input
    counts than config 5's, a vector bias, a transform on the contraction, a second (value) parameter with a gradient of its
    own, two such sites over the same coefficients"""
    rs = np.random.default_rng(17)
    sl = SiteList()
    if which == "normal_96x32":
        N, P = 96, 32
        sl.add("ls", A.NORMAL, [-0.3, 0.4])
        sl.add("w", A.MVNORMAL_DIAG, [Param.const(np.zeros(P, np.float32)), Param.const(np.full(P, 0.7, np.float32))], dim=P)
        X = (rs.standard_normal((N, P)) * 0.4).astype(np.float32)
        sl.add("y", A.NORMAL, [Param.affine(X, "w", bias=rs.standard_normal(N).astype(np.float32)), Param.value("ls", xf=A.XF_EXP)], dim=N)
        obs = {"y": rs.standard_normal(N).astype(np.float32)}
        sel = ("ls", "w")
    elif which == "poisson_48x32":
        N, P = 48, 32
        sl.add("w", A.MVNORMAL_DIAG, [Param.const(np.zeros(P, np.float32)), Param.const(np.full(P, 0.5, np.float32))], dim=P)
        X = (rs.standard_normal((N, P)) * 0.2).astype(np.float32)
        sl.add("k", A.POISSON, [Param.affine(X, "w", bias=np.float32(1.0), xf=A.XF_SOFTPLUS)], dim=N)
        obs = {"k": rs.poisson(1.5, N).astype(np.float32)}
        sel = ("w",)
    else:
        N, P = 128, 16
        sl.add("w", A.MVNORMAL_DIAG, [Param.const(np.zeros(P, np.float32)), Param.const(np.full(P, 1.0, np.float32))], dim=P)
        X1, X2 = (rs.standard_normal((N, P)) * 0.5).astype(np.float32), (rs.standard_normal((64, P)) * 0.5).astype(np.float32)
        sl.add("b", A.BERNOULLI_LOGITS, [Param.affine(X1, "w", bias=np.float32(0.2))], dim=N)
        sl.add("y", A.NORMAL, [Param.affine(X2, "w", bias=np.float32(-0.1)), Param.const(np.float32(0.8))], dim=64)
        obs = {"b": (rs.random(N) < 0.5).astype(np.float32), "y": rs.standard_normal(64).astype(np.float32)}
        sel = ("w",)
    modes = {s.addr: (A.MODE_OBS_TAB if s.addr in obs else A.MODE_OBS_SLOT) for s in sl.sites}
    return PackedProgram(sl, modes, obs, selected=sel, rng_mode=rng)


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("which", ["normal_96x32", "poisson_48x32", "two_sites_16"])
def test_generated_hmc_matrix_core_sites(K_, oracle, rng, which, monkeypatch):
    """the matrix-core flavour of the generated HMC kernel (16 chains per wave, forward and backward contraction on
    v_mfma_f32_16x16x4_f32) == the same program with the flavour switched off (scalar rolled loop) == interpreter == oracle"""
    import torch
    prog = _regression_target(which, rng)
    n = 1000                                                              # (not a multiple of 16: shadow chains in the last wave)
    ch = (np.random.default_rng(8).standard_normal((prog.n_slots, n)) * 0.2).astype(np.float32)
    eps, L = 0.004, 12
    monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
    assert K_.hmc_engine(prog) == 4
    src = K_.program_hmc_source(prog)
    assert src.count("on the matrix cores") == (2 if which == "two_sites_16" else 1) and "mfma_f32_16x16x4f32" in src
    for stale in (False, True):
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        monkeypatch.delenv("GJX_HMC_GEN_NO_MFMA", raising=False)
        g = K_.hmc(prog, (4, 1), torch.as_tensor(ch).cuda(), eps, L, stale, False, offset=5)
        monkeypatch.setenv("GJX_HMC_GEN_NO_MFMA", "1")
        assert "on the matrix cores" not in K_.program_hmc_source(prog)
        gs = K_.hmc(prog, (4, 1), torch.as_tensor(ch).cuda(), eps, L, stale, False, offset=5)
        monkeypatch.delenv("GJX_HMC_GEN_NO_MFMA")
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        it = K_.hmc(prog, (4, 1), torch.as_tensor(ch).cuda(), eps, L, stale, False, offset=5)
        o = oracle.hmc(prog, (4, 1), ch, eps, L, stale, False, offset=5)
        gc = _np(g["choices"])
        assert np.isfinite(gc).all()
        np.testing.assert_allclose(gc, _np(gs["choices"]), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(gc, _np(it["choices"]), rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(gc, o["choices"], rtol=3e-3, atol=3e-3)
        np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=6e-3, atol=6e-3)
        np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=1e-3, atol=6e-3)


def test_generated_hmc_is_the_default_engine_for_unmatched_programs(K_, oracle):
    """engine selection: the hand-written kernels for the config-5 shape, the generated kernel for everything else the
    emitter covers, the interpreter for the rest (a dirichlet site)"""
    prog, _ = H.logreg(N=64, P=4)
    assert K_.hmc_engine(prog) in (2, 3)
    assert K_.hmc_engine(_hierarchy_target(A.RNG_FLAT)) == 4
    assert K_.hmc_engine(_scan_target(16, A.RNG_FLAT)[0]) == 4
    sl = SiteList()
    sl.add("th", A.DIRICHLET, [np.array([1.0, 2.0, 3.0], np.float32)], dim=3)
    sl.add("x", A.NORMAL, [0.0, 1.0])
    pd = PackedProgram(sl, {"th": A.MODE_OBS_SLOT, "x": A.MODE_OBS_SLOT}, selected=("x",))
    assert K_.hmc_engine(pd) == 0
    src = K_.program_hmc_source(_scan_target(16, A.RNG_FLAT)[0])
    assert "gjx_hmc_gen" in src and "sweep<false>" in src


@pytest.mark.parametrize("rng", RNGS)
def test_random_programs_generated_hmc_vs_interpreter_and_oracle(K_, oracle, rng, monkeypatch):
    """Differential test of the HMC emitter: random site programs (test_gpu_parity._random_program: every kind, the four
    parameter forms, transforms), every site constrained to the oracle's own draws, every differentiable site selected,
    a short trajectory without the accept step — generated kernel == site interpreter == oracle on the chains whose
    oracle result is well conditioned.  GJX_FUZZ_TRIALS / GJX_FUZZ_SEED widen the campaign (profiles/gputests.sh)."""
    import torch
    from test_gpu_parity import _random_program
    trials = int(os.environ.get("GJX_FUZZ_TRIALS", "24"))
    rs = np.random.default_rng(int(os.environ.get("GJX_FUZZ_SEED", "311")) + rng)
    n, covered = 256, 0
    for trial in range(trials):
        sl = _random_program(rs, rng)
        sel = tuple(s.addr for s in sl.sites if s.kind not in A.NO_GRADIENT_KINDS and s.kind not in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS))
        key = (int(rs.integers(1 << 30)), int(rs.integers(1 << 30)))
        if not sel:
            continue
        ch = oracle.run_program(PackedProgram(sl, rng_mode=rng), key, n)["choices"].astype(np.float32)
        prog = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=sel, rng_mode=rng)
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        if K_.hmc_engine(prog) != 4:
            continue
        covered += 1
        what = f"trial {trial} ({[A.KIND_NAMES[s.kind] for s in sl.sites]})"
        eps, L = 1e-3, 6
        o = oracle.hmc(prog, key, ch, eps, L, False, False, offset=3)
        o2 = oracle.hmc(prog, key, ch, eps * 1.01, L, False, False, offset=3)
        ch3 = ch.copy()                                       # the selected values moved by a few float32 ulps
        for a_ in sel:
            ch3[prog.slot_of[a_]:prog.slot_of[a_] + sl[a_].dim] *= np.float32(1.0 + 4e-6)
        o3 = oracle.hmc(prog, key, ch3, eps, L, False, False, offset=3)
        g = K_.hmc(prog, key, torch.as_tensor(ch).cuda(), eps, L, False, False, offset=3)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        it = K_.hmc(prog, key, torch.as_tensor(ch).cuda(), eps, L, False, False, offset=3)
        gc, ic, ga, ia = _np(g["choices"]), _np(it["choices"]), _np(g["alpha"]), _np(it["alpha"])
        # well-conditioned chains: the oracle's own result is finite, moderate, moves by less than 1e-3 when the step size
        # changes by 1 %, and does not notice a start that is a few float32 ulps away (a chain that sits next to a pole of
        # its density amplifies every rounding difference: device and oracle round differently)
        with np.errstate(invalid="ignore"):
            # (|alpha| < 1e-3: with eps = 1e-3 and 6 steps the energy error of a smooth chain is ~1e-6; an alpha of 1e-2 that
            # does not shrink with eps is the float32 rounding of the score itself, e.g. lgamma((nu+1)/2) - lgamma(nu/2) of a
            # student-t with nu = exp(10): one ulp of lgamma(16000) is 0.016)
            well = (np.isfinite(o["choices"]).all(0) & np.isfinite(o["alpha"]) & (np.abs(o["choices"]).max(0) < 1e3) & (np.abs(o["alpha"]) < 1e-3)
                    & (np.abs(o2["choices"] - o["choices"]).max(0) < 1e-3) & (np.abs(o2["alpha"] - o["alpha"]) < 5e-4)
                    & (np.abs(o3["choices"] - o["choices"]).max(0) < 2e-4) & (np.abs(o3["alpha"] - o["alpha"]) < 5e-4))
        assert well.mean() > 0.2, what
        np.testing.assert_allclose(gc[:, well], ic[:, well], rtol=2e-3, atol=2e-3, err_msg=what + " generated vs interpreter")
        np.testing.assert_allclose(gc[:, well], o["choices"][:, well], rtol=3e-3, atol=3e-3, err_msg=what + " generated vs oracle")
        # alpha is a difference of two scores: its float32 rounding error grows with their magnitude (summation order and the
        # hardware log / exp differ between the device and the oracle's libm)
        mag = 5e-6 * np.maximum(np.abs(o["score"]), np.abs(o["score"] - o["alpha"]))[well]
        assert (np.abs(ga[well] - ia[well]) <= 5e-3 + 5e-3 * np.abs(ia[well]) + mag).all(), what + " alpha generated vs interpreter"
        err = np.abs(ga[well] - o["alpha"][well]) - (6e-3 + 6e-3 * np.abs(o["alpha"][well]) + mag)
        w = int(np.argmax(err))
        assert err[w] <= 0, (f"{what} alpha generated vs oracle: chain {np.flatnonzero(well)[w]} generated {ga[well][w]} interpreter {ia[well][w]} "
                             f"oracle {o['alpha'][well][w]} score {o['score'][well][w]} start {ch[:, np.flatnonzero(well)[w]]}")
    assert covered >= trials // 3, f"the emitter covered {covered} of {trials} random programs"


def _long_scan_model(T, rng, with_global):
    """stochastic volatility over T steps: x_t ~ normal(phi x_{t-1}, sigma), y_t ~ normal(0, exp(x_t / 2)) observed, every x_t selected —
    `with_global`: the transition's scale is exp(ls) with ls ~ normal(-1.2, 0.3) a selected choice in FRONT of the Scan"""
    import genjax_amd as genjax
    from genjax_amd import C
    phi = 0.95
    ys = np.random.default_rng(1).standard_normal(T).astype(np.float32)

    @genjax.gen
    def model():
        ls = (genjax.normal(-1.2, 0.3) @ "ls") if with_global else None

        @genjax.gen
        def step(x_prev, _):          # (a closure over the choice in front of the Scan: gen.py's closures with keyword arguments)
            x = genjax.normal(phi * x_prev, genjax.exp(ls) if with_global else 0.3) @ "x"
            genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
            return x, None

        step.scan(n=T)(0.0, None) @ "s"

    xs = [(("s", "x"), t) for t in range(T)]
    sel = (["ls"] if with_global else []) + xs
    prog, _, _ = model.pack((), C["s", "y"].set(ys), False, selected=tuple(sel), per_particle=tuple(sel), rng_mode=rng)
    return prog


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("T,with_global", [(128, False), (203, False), (96, True)])
def test_hmc_over_a_long_scan_runs_on_a_rolled_generated_kernel(K_, oracle, rng, T, with_global, monkeypatch):
    """HMC.edit over every state of a T-step Scan (hmc.py:70-96 differentiates any assess; scan.py:237-294) — 2 T sites, T selected
    values: more than the straight-line kernel's registers hold.  The generated kernel rolls the Scan: the steps are dealt to the
    lanes of a chain in contiguous chunks, the trajectory state of the steps' values lives in workspace rows, a lane carries a
    row's own gradient part into the next step's iteration.  Against the site interpreter and the oracle, with 4, 16 and 64 lanes
    per chain, the stale-gradient compatibility mode and the accept step; no HMC program goes to the interpreter for its length."""
    import torch
    n, L = 300, 8
    prog = _long_scan_model(T, rng, with_global)
    assert prog.n_sites == 2 * T + (1 if with_global else 0)
    src = K_.program_hmc_source(prog)
    assert "rolled Scan: %d steps x 2 sites, 1 selected values per step" % T in src and "// PROWS %d" % T in src
    ch = (np.random.default_rng(4).standard_normal((prog.n_slots, n)) * 0.3).astype(np.float32)
    if with_global:
        ch[0] = -1.2 + 0.1 * ch[0]
    for stale, accept, cpl in ((False, False, "4"), (True, False, "16"), (False, True, "64"), (False, False, "64")):
        eps = 0.02 if accept else 0.005
        o = oracle.hmc(prog, (6, 2), ch, eps, L, stale, accept, offset=7)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        assert K_.hmc_engine(prog) == 0
        it = K_.hmc(prog, (6, 2), torch.as_tensor(ch).cuda(), eps, L, stale, accept, offset=7)
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        monkeypatch.setenv("GJX_HMC_GEN_CPL", cpl)
        assert K_.hmc_engine(prog) == 4
        g = K_.hmc(prog, (6, 2), torch.as_tensor(ch).cuda(), eps, L, stale, accept, offset=7)
        gc = _np(g["choices"])
        assert np.isfinite(gc).all()
        if not accept:
            np.testing.assert_allclose(gc, o["choices"], rtol=3e-3, atol=3e-3)
            np.testing.assert_allclose(gc, _np(it["choices"]), rtol=3e-3, atol=3e-3)
            np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=6e-3, atol=2e-2)
            np.testing.assert_allclose(_np(g["score"]), o["score"], rtol=1e-3, atol=2e-2)
            assert np.abs(gc - ch).max() > 1e-3
        else:
            acc_g, acc_o = _np(g["accepted"]) > 0.5, o["accepted"] > 0.5
            assert (acc_g != acc_o).mean() < 0.02 and 0.1 < acc_g.mean() <= 1.0
            same = acc_g == acc_o
            np.testing.assert_allclose(gc[:, same], o["choices"][:, same], rtol=3e-3, atol=3e-3)
            np.testing.assert_array_equal(gc[:, ~acc_g], ch[:, ~acc_g])
    monkeypatch.delenv("GJX_HMC_ENGINE")
    monkeypatch.delenv("GJX_HMC_GEN_CPL")


@pytest.mark.parametrize("rng", RNGS)
def test_random_long_scans_hmc_generated_interpreter_oracle(K_, oracle, rng, monkeypatch):
    """differential test of the rolled-Scan path of the HMC emitter: random step kernels — the transition's family, a two-dimensional
    or scalar state, an optional selected choice in front of the Scan that the steps read, the observation's family, a number of steps
    that no chunking divides, the lanes per chain — generated kernel == site interpreter == oracle"""
    import torch
    import genjax_amd as genjax
    from genjax_amd import C
    rs = np.random.default_rng(int(os.environ.get("GJX_FUZZ_SEED", "911")) + rng)
    trials, covered = int(os.environ.get("GJX_FUZZ_TRIALS", "8")), 0
    for trial in range(trials):
        T = int(rs.choice([67, 101, 150, 259]))
        dim = int(rs.choice([1, 2]))
        trans = str(rs.choice(["normal", "laplace", "student_t", "gumbel"]))
        obs_kind = str(rs.choice(["normal", "poisson", "bernoulli", "sv"]))
        with_global = bool(rs.integers(2))
        cpl = str(rs.choice(["4", "16", "64"]))
        Am = (0.9 * np.eye(dim) + 0.05 * rs.standard_normal((dim, dim))).astype(np.float32)

        @genjax.gen
        def model():
            ls = (genjax.normal(-1.0, 0.3) @ "ls") if with_global else None

            @genjax.gen
            def step(x_prev, _):
                sc = genjax.exp(ls) if with_global else 0.4
                loc = Am @ x_prev if dim > 1 else 0.9 * x_prev
                if dim > 1:
                    x = genjax.mv_normal_diag(loc, np.full(dim, 0.4, np.float32)) @ "x"
                    h = x[0]
                elif trans == "normal":
                    x = genjax.normal(loc, sc) @ "x"
                    h = x
                elif trans == "laplace":
                    x = genjax.laplace(loc, sc) @ "x"
                    h = x
                elif trans == "student_t":
                    x = genjax.student_t(5.0, loc, sc) @ "x"
                    h = x
                else:
                    x = genjax.gumbel(loc, sc) @ "x"
                    h = x
                if obs_kind == "normal":
                    genjax.normal(h, 0.6) @ "y"
                elif obs_kind == "poisson":
                    genjax.poisson(genjax.exp(0.4 * h)) @ "y"
                elif obs_kind == "bernoulli":
                    genjax.bernoulli(logits=h) @ "y"
                else:
                    genjax.normal(0.0, genjax.exp(0.5 * h)) @ "y"
                return x, None

            step.scan(n=T)(np.zeros(dim, np.float32) if dim > 1 else 0.0, None) @ "s"

        ys = (rs.poisson(1.0, T) if obs_kind == "poisson" else (rs.uniform(size=T) < 0.5) if obs_kind == "bernoulli" else rs.standard_normal(T)).astype(np.float32)
        xs = [(("s", "x"), t) for t in range(T)]
        sel = (["ls"] if with_global else []) + xs
        prog, _, _ = model.pack((), C["s", "y"].set(ys), False, selected=tuple(sel), per_particle=tuple(sel), rng_mode=rng)
        n = 150
        ch = (0.3 * rs.standard_normal((prog.n_slots, n))).astype(np.float32)
        if with_global:
            ch[prog.slot_of["ls"]] = -1.0 + 0.1 * rs.standard_normal(n)
        what = f"trial {trial}: T={T} dim={dim} transition {trans} obs {obs_kind} global {with_global} lanes {cpl}"
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        monkeypatch.setenv("GJX_HMC_GEN_CPL", cpl)
        if K_.hmc_engine(prog) != 4:
            print(what, "-> not on a generated kernel")
            continue
        assert "rolled Scan" in K_.program_hmc_source(prog), what
        covered += 1
        eps, L = 2e-3, 5
        g = K_.hmc(prog, (8, trial), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=4)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        it = K_.hmc(prog, (8, trial), torch.as_tensor(ch).cuda(), eps, L, False, False, offset=4)
        o = oracle.hmc(prog, (8, trial), ch, eps, L, False, False, offset=4)
        o2 = oracle.hmc(prog, (8, trial), ch, eps * 1.01, L, False, False, offset=4)
        with np.errstate(invalid="ignore"):
            well = np.isfinite(o["choices"]).all(0) & np.isfinite(o["alpha"]) & (np.abs(o2["choices"] - o["choices"]).max(0) < 1e-3) & (np.abs(o["alpha"]) < 0.5)
        assert well.mean() > 0.5, what
        gc = _np(g["choices"])
        np.testing.assert_allclose(gc[:, well], _np(it["choices"])[:, well], rtol=3e-3, atol=3e-3, err_msg=what + " generated vs interpreter")
        np.testing.assert_allclose(gc[:, well], o["choices"][:, well], rtol=3e-3, atol=3e-3, err_msg=what + " generated vs oracle")
        mag = 2e-5 * np.abs(o["score"])[well]
        assert (np.abs(_np(g["alpha"])[well] - o["alpha"][well]) <= 2e-2 + mag).all(), what + " alpha"
    monkeypatch.delenv("GJX_HMC_ENGINE", raising=False)
    monkeypatch.delenv("GJX_HMC_GEN_CPL", raising=False)
    assert covered >= trials // 2, covered


@pytest.mark.parametrize("rng", RNGS)
@pytest.mark.parametrize("groups,dim", [(10, 12), (12, 16)])
def test_generated_hmc_with_the_chain_state_in_lds(K_, oracle, rng, groups, dim, monkeypatch):
    """straight-line programs beyond the register budget of the generated HMC kernel (more than 96 values / 48 selected ones: here 120
    and 192 selected scalars in 10 / 12 vector sites under a common scale) run on its LDS-state flavour (HmcPlan::big: values,
    gradient and momenta as LDS columns of a one-wave block) instead of the site interpreter.  Generated ==
    interpreter == oracle, with and without the accept; the stale-carry compatibility mode where the first gradient still fits the
    LDS, and through the interpreter (silently, same results) where it does not."""
    import torch
    rs = np.random.default_rng(3)
    sl = SiteList()
    sl.add("ls", A.NORMAL, [0.0, 0.5])
    for j in range(groups):
        sl.add(f"m{j}", A.MVNORMAL_DIAG, [np.zeros(dim, np.float32), Param.value("ls", xf=A.XF_EXP)], dim=dim)
        sl.add(f"y{j}", A.MVNORMAL_DIAG, [Param.value(f"m{j}", dim), np.full(dim, 0.7, np.float32)], dim=dim)
    modes = {s.addr: (A.MODE_OBS_TAB if s.addr.startswith("y") else A.MODE_OBS_SLOT) for s in sl.sites}
    obs = {f"y{j}": rs.standard_normal(dim).astype(np.float32) for j in range(groups)}
    sel = ("ls",) + tuple(f"m{j}" for j in range(groups))
    prog = PackedProgram(sl, modes, obs, selected=sel, rng_mode=rng)
    assert prog.n_slots == 1 + groups * dim
    n = 300
    ch = (rs.standard_normal((prog.n_slots, n)) * 0.4).astype(np.float32)
    monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
    assert K_.hmc_engine(prog) == 4, "the LDS-state flavour must take this program"
    assert "LdsCol v{" in K_.program_hmc_source(prog)
    nostale = "// NOSTALE 1" in K_.program_hmc_source(prog)
    assert nostale == (groups * dim + 1 > 150)
    for stale, accept in ((False, False), (True, False), (False, True)):
        eps, L = (0.02 if accept else 0.005), 8
        o = oracle.hmc(prog, (4, 5), ch, eps, L, stale, accept, offset=2)
        monkeypatch.setenv("GJX_HMC_ENGINE", "gen")
        g = K_.hmc(prog, (4, 5), torch.as_tensor(ch).cuda(), eps, L, stale, accept, offset=2)
        monkeypatch.setenv("GJX_HMC_ENGINE", "interp")
        it = K_.hmc(prog, (4, 5), torch.as_tensor(ch).cuda(), eps, L, stale, accept, offset=2)
        if not accept:
            np.testing.assert_allclose(_np(g["choices"]), o["choices"], rtol=3e-3, atol=3e-3)
            np.testing.assert_allclose(_np(g["choices"]), _np(it["choices"]), rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(_np(g["alpha"]), o["alpha"], rtol=1e-2, atol=1e-2)
        else:
            flip = _np(g["accepted"]) != o["accepted"]
            assert flip.mean() < 0.03 and (o["margin"][flip] < 1e-2 + 1e-2 * np.abs(o["alpha"][flip])).all()
            rej = _np(g["accepted"]) == 0
            np.testing.assert_array_equal(_np(g["choices"])[:, rej], ch[:, rej])
