"""Scan steps on the device: the site loop is a device loop over the steps' descriptors, step keys chain as in
combinators/scan.py:268 (include/gjx.h "Scan steps"), and a Scan may be longer than the 1023 site numbers of one FLAT
stream.  Checked against the oracle, against single-site programs run under the host-computed chained keys, and
between the interpreter and the generated kernels."""
import os

import numpy as np
import pytest

import helpers as H
from genjax_amd import _abi as A
from genjax_amd import core
from test_scan_keys import chain_keys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K_():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from genjax_amd import kernels
    return kernels


def _np(t):
    return t.detach().cpu().numpy()


class engine:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.old = os.environ.get("GJX_ENGINE")
        os.environ["GJX_ENGINE"] = self.name

    def __exit__(self, *exc):
        if self.old is None:
            del os.environ["GJX_ENGINE"]
        else:
            os.environ["GJX_ENGINE"] = self.old


@pytest.mark.parametrize("eng", ["interp", "gen"])
def test_step_streams_follow_the_chained_key_rule(K_, eng):
    T, K, key = 7, 4096, (1234, 5678)
    one = H.one_site("normal", 0.0, 1.0)
    for scan_id in (0, 5):
        prog, _ = H.scan_chain(T, carry=False, scan_id=scan_id)
        with engine(eng):
            assert K_.program_engine(prog) == (4 if eng == "gen" else 0)
            got = _np(K_.run_program(prog, key, K)["choices"])
            for t, kt in enumerate(chain_keys(key, T, scan_id)):
                want = _np(K_.run_program(one, kt, K)["choices"][0])
                np.testing.assert_array_equal(got[t], want)


def test_long_scan_on_the_device_equals_the_oracle(K_, oracle):
    """1500 steps = 1500 sites: more than one FLAT stream has site numbers; the interpreter's site loop walks the step
    descriptors on the device, chaining the step keys."""
    T, K = 1500, 2048
    prog, _ = H.scan_chain(T, carry=True, sigma=0.1)
    out = K_.run_program(prog, (7, 9), K)
    ora = oracle.run_program(prog, (7, 9), K)
    np.testing.assert_allclose(_np(out["choices"]), ora["choices"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(_np(out["score"]), ora["score"], rtol=2e-4)
    # an observed chain: weights accumulate over the steps
    prog, _ = H.scan_chain(600, carry=True, observe=True, sigma=0.3, r=0.7)
    out = K_.run_program(prog, (3, 4), K)
    ora = oracle.run_program(prog, (3, 4), K)
    np.testing.assert_allclose(_np(out["logw"]), ora["logw"], rtol=3e-4)
    np.testing.assert_allclose(_np(out["lse"])[2:], ora["lse"][2:], rtol=1e-4)


def test_generated_kernel_of_a_scan_equals_the_interpreter(K_, oracle):
    T, K = 20, 1 << 14
    prog, _ = H.scan_chain(T, carry=True, observe=True, sigma=0.3, r=0.7)
    with engine("gen"):
        assert K_.program_engine(prog) == 4
        g = K_.run_program(prog, (5, 6), K)
    with engine("interp"):
        i = K_.run_program(prog, (5, 6), K)
    np.testing.assert_array_equal(_np(g["choices"]), _np(i["choices"]))     # same streams, same sampler arithmetic
    np.testing.assert_allclose(_np(g["logw"]), _np(i["logw"]), rtol=1e-4, atol=1e-4)
    ora = oracle.run_program(prog, (5, 6), K)
    np.testing.assert_allclose(_np(g["choices"]), ora["choices"], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(_np(g["logw"]), ora["logw"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("T,observe", [(20, True), (7, False), (300, True), (1500, True)])
def test_rolled_scan_kernel_equals_the_interpreter_and_the_oracle(K_, oracle, T, observe):
    """A periodic Scan is emitted as ONE loop over its steps (gjx_codegen.hip "Rolled Scans": previous / current step in
    registers, rows and table offsets advancing with t, chained step key): same values as the interpreter's walk over
    the T step descriptors, bit for bit (same streams, same sampler arithmetic), same weights to float rounding."""
    K = 1 << 13
    prog, _ = H.scan_chain(T, carry=True, observe=observe, sigma=0.3, r=0.7)
    old = os.environ.get("GJX_GEN_ROLL")
    os.environ["GJX_GEN_ROLL"] = "1"
    try:
        with engine("gen"):
            assert K_.program_engine(prog) == 4
            src = K_.program_source(prog)
            assert "for (int t_ = 1; t_ < %d; ++t_)" % T in src and "skt = fold_in(skt, (uint32_t)t_)" in src
            g = K_.run_program(prog, (5, 6), K, want_site_scores=True)
    finally:
        if old is None:
            del os.environ["GJX_GEN_ROLL"]
        else:
            os.environ["GJX_GEN_ROLL"] = old
    with engine("interp"):
        i = K_.run_program(prog, (5, 6), K, want_site_scores=True)
    np.testing.assert_array_equal(_np(g["choices"]), _np(i["choices"]))
    np.testing.assert_allclose(_np(g["site_scores"]), _np(i["site_scores"]), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_np(g["logw"]), _np(i["logw"]), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(_np(g["lse"])[2:], _np(i["lse"])[2:], rtol=1e-4)
    if T <= 300:
        ora = oracle.run_program(prog, (5, 6), K)
        np.testing.assert_allclose(_np(g["choices"]), ora["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(g["logw"]), ora["logw"], rtol=3e-4, atol=3e-4)


def test_long_scan_picks_the_rolled_kernel_by_itself(K_):
    """Too long to unroll (the generator takes 48 sites): the engine choice rolls the Scan instead of falling back to the
    interpreter."""
    prog, _ = H.scan_chain(400, carry=True, observe=True, sigma=0.3, r=0.7)
    assert K_.program_engine(prog) == 4
    short, _ = H.scan_chain(3, carry=True, observe=True)          # fewer than 4 steps: plain unrolled kernel
    assert "for (int t_ = 1;" not in K_.program_source(short)


def test_gen_scan_api_with_more_steps_than_site_numbers(K_):
    """`kernel.scan(n=T)` through the API with T > 1023: simulate, then the score is the sum of the step log-densities of
    the returned choices (scan.py:283-294)."""
    import genjax_amd as genjax

    @genjax.gen
    def step(carry, _):
        x = genjax.normal(carry, 0.25) @ "x"
        return x, x

    T, K = 1200, 256
    tr = step.scan(n=T).simulate(genjax.key(11), (0.0, None), K=K)
    xs = np.stack([_np(tr.get_choices()[t, "x"]) for t in (0, 1, 599, T - 1)])
    assert np.isfinite(xs).all() and xs.shape == (4, K)
    allx = _np(tr.get_choices()[:, "x"]) if hasattr(tr.get_choices(), "__getitem__") else None
    x = np.asarray(allx, np.float64)
    if x.shape[0] != T:
        x = x.T
    inc = np.diff(np.vstack([np.zeros((1, K)), x]), axis=0) / 0.25
    want = (-0.5 * inc ** 2 - 0.5 * np.log(2 * np.pi) - np.log(0.25)).sum(axis=0)
    np.testing.assert_allclose(_np(tr.get_score()), want, rtol=1e-4)


def test_observed_scan_through_the_api_runs_the_rolled_kernel(K_):
    """`@gen` step kernel -> `.scan(n=T)` -> Target with a whole-sequence constraint -> ImportanceK: the packed program
    is periodic, so the engine is the rolled generated kernel; the importance weights are the observation log-densities
    of the sampled trajectories (static.py:377 summed over the steps, scan.py:283-294)."""
    import genjax_amd as genjax
    from genjax_amd import C
    from genjax_amd.inference import ImportanceK, Target

    @genjax.gen
    def step(carry, _):
        x = genjax.normal(carry, 0.25) @ "x"
        genjax.normal(x, 0.5) @ "y"
        return x, x

    T, K = 300, 1 << 12
    ys = np.random.default_rng(0).standard_normal(T).astype(np.float32) * 0.3
    model = step.scan(n=T)
    prog, _, _ = model.pack((0.0, None), C["y"].set(ys), True)
    assert K_.program_engine(prog) == 4 and "for (int t_ = 1;" in K_.program_source(prog)
    pc = ImportanceK(Target(model, (0.0, None), C["y"].set(ys)), k_particles=K).run_smc(genjax.key(5))
    x = np.asarray(_np(pc.get_particles().get_choices()[:, "x"]), np.float64)
    if x.shape[0] != T:
        x = x.T
    want = (-0.5 * ((ys[:, None] - x) / 0.5) ** 2 - 0.5 * np.log(2 * np.pi) - np.log(0.5)).sum(axis=0)
    np.testing.assert_allclose(_np(pc.get_log_weights()), want, rtol=2e-4)


def test_rolled_scan_with_a_hyper_site_vector_state_and_affine_transition(K_, oracle):
    """The general shapes: a site BEFORE the Scan that every step reads (scale of the transition noise), a vector state
    with an affine transition x_t ~ N(A x_{t-1}, s), a categorical regime per step that gathers the observation scale,
    and vector observations — rolled kernel vs interpreter (bitwise values) vs oracle."""
    from genjax_amd.program import PackedProgram, Param, SiteList
    rs = np.random.default_rng(3)
    T, D, K = 60, 3, 1 << 12
    Amat = (0.9 * np.eye(D) + 0.05 * rs.standard_normal((D, D))).astype(np.float32)
    ys = rs.standard_normal((T, D)).astype(np.float32)
    sl = SiteList()
    modes, obs = {}, {}
    sl.add("s", A.HALF_NORMAL, [Param.const(0.5)])
    for t in range(T):
        tag = (0 << 20) | (t + 1)
        loc = Param.affine(Amat, ("x", t - 1)) if t > 0 else Param.const(np.zeros(D, np.float32))
        sx = sl.add(("x", t), A.MVNORMAL_DIAG, [loc, Param.value("s", 1)], dim=D)
        sz = sl.add(("z", t), A.CATEGORICAL_LOGITS, [np.array([0.2, -0.1, 0.4], np.float32)])
        sy = sl.add(("y", t), A.MVNORMAL_DIAG, [Param.value(("x", t), D), Param.gather(np.array([[0.5], [1.0], [2.0]], np.float32), ("z", t))], dim=D)
        for s_ in (sx, sz, sy):
            s_.scan = tag
        modes[("y", t)] = A.MODE_OBS_TAB
        obs[("y", t)] = ys[t]
    prog = PackedProgram(sl, modes, obs, rng_mode=A.RNG_FLAT)
    with engine("gen"):
        assert K_.program_engine(prog) == 4                      # 181 sites: only the rolled form fits the generator
        assert "for (int t_ = 1;" in K_.program_source(prog)
        g = K_.run_program(prog, (8, 9), K, want_site_scores=True)
    with engine("interp"):
        i = K_.run_program(prog, (8, 9), K, want_site_scores=True)
    gc, ic = _np(g["choices"]), _np(i["choices"])
    np.testing.assert_array_equal(gc, ic)
    np.testing.assert_allclose(_np(g["site_scores"]), _np(i["site_scores"]), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(_np(g["logw"]), _np(i["logw"]), rtol=3e-4, atol=3e-4)
    ora = oracle.run_program(prog, (8, 9), K, want_margin=True)
    bad = (np.abs(gc - ora["choices"]) > 5e-5 + 2e-4 * np.abs(ora["choices"])).any(axis=0)
    assert bad.mean() < 0.02                                   # regime flips at near ties propagate down the chain
    good = ~bad
    np.testing.assert_allclose(_np(g["logw"])[good], ora["logw"][good], rtol=5e-4, atol=5e-4)


def _scan_with_extras(T, x_mode, post=True, pre_masked=False, seed=0):
    """mu ~ N(0, 1) in front of a Scan of T steps {x_t ~ N(x_{t-1}, 0.3) (x_{-1} = mu), y_t ~ N(x_t, 0.7) observed}, and behind
    it z ~ N(x_{T-1}, 1), v ~ N(mu, 2), w ~ N(z, 0.5) observed.  x_mode: the constraint mode of every x_t (per-particle values /
    per-particle masks live in rows of choices[][])."""
    from genjax_amd import _abi as A
    from genjax_amd.program import PackedProgram, Param, SiteList
    rs = np.random.default_rng(seed)
    sl = SiteList()
    modes, obs = {}, {}
    sl.add("mu", A.NORMAL, [0.0, 1.0])
    if pre_masked:
        modes["mu"] = A.MODE_OBS_MASK
    for t in range(T):
        loc = Param.value(("x", t - 1)) if t else Param.value("mu")          # step 0 starts at the pre-Scan site
        sx = sl.add(("x", t), A.NORMAL, [loc, Param.const(0.3)])
        sy = sl.add(("y", t), A.NORMAL, [Param.value(("x", t)), Param.const(0.7)])
        sx.scan = sy.scan = (3 << 20) | (t + 1)
        modes[("y", t)] = A.MODE_OBS_TAB
        obs[("y", t)] = np.float32(rs.standard_normal())
        if x_mode != A.MODE_SAMPLE:
            modes[("x", t)] = x_mode
    if post:
        sl.add("z", A.NORMAL, [Param.value(("x", T - 1)), Param.const(1.0)])
        sl.add("v", A.NORMAL, [Param.value("mu"), Param.const(2.0)])
        sl.add("w", A.NORMAL, [Param.value("z"), Param.const(0.5)])
        modes["w"] = A.MODE_OBS_TAB
        obs["w"] = np.float32(0.3)
    return PackedProgram(sl, modes, obs)


def _rolled_vs_interp_vs_oracle(K_, oracle, prog, K, T, ch=None):
    import torch
    old = os.environ.get("GJX_GEN_ROLL")
    os.environ["GJX_GEN_ROLL"] = "1"
    try:
        with engine("gen"):
            assert K_.program_engine(prog) == 4
            src = K_.program_source(prog)
            assert "for (int t_ = 1; t_ < %d; ++t_)" % T in src
            g = K_.run_program(prog, (5, 6), K, choices=None if ch is None else torch.as_tensor(ch).cuda(), want_site_scores=True)
    finally:
        if old is None:
            del os.environ["GJX_GEN_ROLL"]
        else:
            os.environ["GJX_GEN_ROLL"] = old
    with engine("interp"):
        i = K_.run_program(prog, (5, 6), K, choices=None if ch is None else torch.as_tensor(ch).cuda(), want_site_scores=True)
    np.testing.assert_array_equal(_np(g["choices"]), _np(i["choices"]))
    np.testing.assert_allclose(_np(g["site_scores"]), _np(i["site_scores"]), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_np(g["logw"]), _np(i["logw"]), rtol=2e-4, atol=3e-4)
    np.testing.assert_allclose(_np(g["score"]), _np(i["score"]), rtol=2e-4, atol=3e-4)
    ora = oracle.run_program(prog, (5, 6), K, choices=None if ch is None else ch.copy())
    np.testing.assert_allclose(_np(g["choices"]), ora["choices"], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(_np(g["logw"]), ora["logw"], rtol=3e-4, atol=5e-4)
    return g


@pytest.mark.parametrize("T", [12, 300])
def test_rolled_scan_with_sites_behind_the_scan(K_, oracle, T):
    """sites after the Scan (a final latent and an observation) are emitted behind the loop and read the LAST step from
    the registers the final carry left it in"""
    from genjax_amd import _abi as A
    prog = _scan_with_extras(T, A.MODE_SAMPLE, post=True)
    g = _rolled_vs_interp_vs_oracle(K_, oracle, prog, 4096, T)
    z = _np(g["choices"])[prog.slot_of["z"]]
    xl = _np(g["choices"])[prog.slot_of[("x", T - 1)]]
    assert 0.8 < np.std(z - xl) < 1.2                                  # z ~ N(x_{T-1}, 1)


@pytest.mark.parametrize("T,post", [(10, False), (200, True)])
def test_rolled_scan_with_per_particle_constraints(K_, oracle, T, post):
    """every x_t constrained to the particle's OWN value (GJX_MODE_OBS_SLOT: what assess / ChangeTarget / Update run over a
    Scan trace): the rolled loop loads the step's rows as t advances; the weight is the whole score"""
    from genjax_amd import _abi as A
    prog = _scan_with_extras(T, A.MODE_OBS_SLOT, post=post)
    K = 2048
    rs = np.random.default_rng(4)
    ch = (rs.standard_normal((prog.n_slots, K)) * 0.5).astype(np.float32)
    g = _rolled_vs_interp_vs_oracle(K_, oracle, prog, K, T, ch)
    for t in (0, T // 2, T - 1):
        np.testing.assert_array_equal(_np(g["choices"])[prog.slot_of[("x", t)]], ch[prog.slot_of[("x", t)]])   # untouched


@pytest.mark.parametrize("T,post,pre_masked", [(9, True, True), (150, False, False)])
def test_rolled_scan_with_per_particle_masks(K_, oracle, T, post, pre_masked):
    """Mask(value, flag) per particle and step (GJX_MODE_OBS_MASK): flag rows advance with t like the value rows"""
    from genjax_amd import _abi as A
    prog = _scan_with_extras(T, A.MODE_OBS_MASK, post=post, pre_masked=pre_masked)
    K = 2048
    rs = np.random.default_rng(5)
    ch = (rs.standard_normal((prog.n_slots, K)) * 0.5).astype(np.float32)
    for a, fs in prog.flag_slot_of.items():
        ch[fs] = rs.random(K) < 0.4
    g = _rolled_vs_interp_vs_oracle(K_, oracle, prog, K, T, ch)
    a = ("x", T // 2)
    fl = ch[prog.flag_slot_of[a]] != 0
    np.testing.assert_array_equal(_np(g["choices"])[prog.slot_of[a]][fl], ch[prog.slot_of[a]][fl])
    assert (_np(g["choices"])[prog.slot_of[a]][~fl] != ch[prog.slot_of[a]][~fl]).mean() > 0.99


@pytest.mark.parametrize("T", [120, 259])
def test_hidden_markov_model_with_latent_emission_means_rolled(T, monkeypatch):
    """mu ~ N(0, 4 I_3) in front of a T-step Scan; c_t ~ categorical(P[c_{t-1}]), y_t ~ N(mu[c_t], 0.5) observed: the emission reads a row
    of a choice in FRONT of the Scan picked by the step's discrete choice (GJX_P_VGATHER in a rolled loop).  ImportanceK on the generated
    (rolled) kernel and on the interpreter against the oracle; HMC over mu with the states fixed per chain on the rolled generated
    kernel (nothing of the Scan is selected: the loop only feeds mu's gradient) against interpreter and oracle"""
    import torch
    import genjax_amd as genjax
    from genjax_amd import C, kernels
    from oracle import cpu
    rs = np.random.default_rng(1)
    ys = rs.standard_normal(T).astype(np.float32)
    P = np.array([[1.0, -1.0, 0.0], [0.0, 1.0, -1.0], [-1.0, 0.0, 1.0]], np.float32)

    @genjax.gen
    def model():
        mu = genjax.normal(np.zeros(3, np.float32), 2.0) @ "mu"

        @genjax.gen
        def step(c_prev, _):
            c = genjax.categorical(logits=genjax.take(P, c_prev) if not isinstance(c_prev, (int, np.integer)) else P[int(c_prev)]) @ "c"
            genjax.normal(mu[c], 0.5) @ "y"
            return c, None

        step.scan(n=T)(0, None) @ "s"

    K = 2000
    prog, _, _ = model.pack((), C["s", "y"].set(ys), True)
    assert prog.n_sites == 1 + 2 * T
    src = kernels.program_source(prog, 1)
    assert "for (int t_ = 1" in src and "vg_" in src
    ora = cpu.run_program(prog, (3, 4), K, want_margin=True)
    for engine in ("gen", "interp"):
        monkeypatch.setenv("GJX_ENGINE", engine)
        out = kernels.run_program(prog, (3, 4), K)
        assert out["_engine"] == (4 if engine == "gen" else 0)
        ch, oc = out["choices"].cpu().numpy(), ora["choices"]
        same = (ch == oc).all(axis=0) | (np.abs(ch - oc) <= 5e-5 + 2e-4 * np.abs(oc)).all(axis=0)
        assert (ora["margin"][~same] < 3e-4).all() and same.mean() > 0.9          # a particle differs only behind a near-tie of a categorical draw
        np.testing.assert_allclose(out["weight"].cpu().numpy()[same], ora["weight"][same], rtol=3e-4, atol=3e-2)
    monkeypatch.delenv("GJX_ENGINE")
    cs = [(("s", "c"), t) for t in range(T)]
    hp, _, _ = model.pack((), C["s", "y"].set(ys), False, selected=("mu",), per_particle=tuple(["mu"] + cs))
    n = 200
    ch = np.zeros((hp.n_slots, n), np.float32)
    ch[:3] = 0.5 * rs.standard_normal((3, n))
    ch[3:] = rs.integers(0, 3, (T, n))
    hs = kernels.program_hmc_source(hp)
    assert "rolled Scan: %d steps x 2 sites, 0 selected values per step" % T in hs
    o = cpu.hmc(hp, (5, 6), ch, 0.01, 8, False, False, offset=2)
    for engine, code, cpl in (("gen", 4, "4"), ("gen", 4, "64"), ("interp", 0, "4")):
        monkeypatch.setenv("GJX_HMC_ENGINE", engine)
        monkeypatch.setenv("GJX_HMC_GEN_CPL", cpl)
        assert kernels.hmc_engine(hp) == code
        g = kernels.hmc(hp, (5, 6), torch.as_tensor(ch).cuda(), 0.01, 8, False, False, offset=2)
        gc = g["choices"].cpu().numpy()
        np.testing.assert_allclose(gc[:3], o["choices"][:3], rtol=3e-3, atol=3e-3)
        np.testing.assert_array_equal(gc[3:], ch[3:])
        np.testing.assert_allclose(g["alpha"].cpu().numpy(), o["alpha"], rtol=1e-2, atol=5e-2)
    monkeypatch.delenv("GJX_HMC_ENGINE")
    monkeypatch.delenv("GJX_HMC_GEN_CPL")
