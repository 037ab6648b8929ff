"""CPU checks (no GPU, no compute call into the product library) of round 4's program constructs: plate-tagged sites
(include/gjx.h "Plates"; reference combinators/vmap.py:180-218), GJX_MODE_INPUT sites and the one-step programs of the
bootstrap filter for any Scan kernel (scan.py:237-294), the matrix-core flavour of the generated kernels.  The ORACLE runs the
packed programs (it is the checker); the product side is exercised as far as it goes without a device: tracing, packing, the
emitted HIP source and its hipRTC cross-compilation for gfx950."""
import ctypes

import numpy as np
import pytest

import genjax_amd as genjax
from genjax_amd import C
from genjax_amd import _abi as A


def _mixture(N):
    rs = np.random.default_rng(0)
    mu = np.array([-2.0, 0.5, 3.0], np.float32)
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


def test_abi7_structs():
    assert ctypes.sizeof(A.GjxParam) == 48 and ctypes.sizeof(A.GjxSite) == 240
    assert A.GjxSite.plate.offset == 32 and A.GjxSite.p.offset == 48 and A.GjxParam.d_off.offset == 28
    assert ctypes.sizeof(A.GjxRunOpts) == 56 and ctypes.sizeof(A.GjxRunInfo) == 16 and ctypes.sizeof(A.GjxRunResample) == 72
    assert A.GjxRunResample.u.offset == 48 and A.GjxRunOpts.resample.offset == 48
    hdr = open(__file__.rsplit("/tests/", 1)[0] + "/include/gjx.h").read()
    assert "#define GJX_ABI_VERSION %d" % A.ABI_VERSION in hdr and A.ABI_VERSION >= 9 and "GJX_P_EXPR = 5" in hdr and "GJX_E_LINN = 24" in hdr and "GJX_MODE_INPUT = 4" in hdr and "GJX_STATUS_VERIFY_MISMATCH = 4" in hdr
    # ABI 8: the generic filter takes its form as arguments and reports the form that ran (no environment, no per-thread state)
    assert ctypes.sizeof(A.GjxFilterOpts) == 88 and A.GjxFilterOpts.hmc_targets.offset == 40 and A.GjxFilterOpts.hmc_L.offset == 52 and A.GjxFilterOpts.hmc_workspace_bytes.offset == 80 and ctypes.sizeof(A.GjxFilterInfo) == 16 and A.GjxFilterOpts.timeline.offset == 8 and A.GjxFilterOpts.n_moves.offset == 24 and A.GjxFilterOpts.accepted_total.offset == 32
    assert "GJX_FILTER_FORM_WIDE = 3" in hdr and "GJX_FILTER_NO_ONE_LAUNCH = 3" in hdr


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_vmapped_mixture_lowers_to_two_plate_sites_and_the_oracle_scores_it(rng):
    from oracle import cpu
    N, K = 600, 400                                  # 1200 logical sites: more than the 1023 site numbers of the FLAT layout
    model, chm, ys, mu, logits = _mixture(N)
    prog, _, _ = model.pack((), chm, True, rng_mode=rng)
    assert prog.n_sites == 2 and prog.n_slots == N and len(prog.site_list.sites) == 2 * N
    z_site, x_site = prog.c_sites[0], prog.c_sites[1]
    assert (z_site.plate, z_site.plate_n, z_site.dim, z_site.slot) == (1, N, 1, 0)
    assert (x_site.plate, x_site.plate_n, x_site.mode, x_site.d_obs, x_site.slot) == (1, N, A.MODE_OBS_TAB, 1, -1)
    assert (x_site.p[0].op, x_site.p[0].slot, x_site.p[0].d_slot, x_site.p[0].d_off) == (A.P_GATHER, 0, 1, 0)
    assert prog.slot_of[(("k", "z"), 17)] == 17 and prog.obs_off[(("k", "x"), 17)] == prog.obs_off[("@plate", ("k", "x"))] + 17
    o = cpu.run_program(prog, (0, 5), K, want_site_scores=True)
    z = o["choices"][:N].astype(int)
    p = np.exp(logits - logits.max())
    p /= p.sum()
    lz = np.log(p)[z].sum(axis=0)
    lx = (-0.5 * ((ys[:, None] - mu[z]) / 0.7) ** 2 - np.log(0.7) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(o["weight"], lx, rtol=2e-5, atol=5e-3)
    np.testing.assert_allclose(o["score"], lx + lz, rtol=2e-5, atol=5e-3)
    np.testing.assert_allclose(o["site_scores"][0], lz, rtol=2e-5, atol=5e-3)       # one row per BODY site: the sum over its instances
    assert np.abs(np.bincount(z.ravel(), minlength=3) / z.size - p).max() < 5e-3


def test_jax32_a_plate_advances_the_callers_site_counter_by_one():
    """static.py:349-352: the Vmap call is ONE traced site of its caller — the site behind a plate has counter J + 1 whatever
    the number of sites in the vmapped kernel's body (the counter used to advance by the body's length)"""
    from oracle import cpu
    n, K = 8, 64

    @genjax.gen
    def one(x):
        return genjax.normal(x, 1.0) @ "m"

    @genjax.gen
    def two(x):
        c = genjax.flip(0.4) @ "c"
        return genjax.normal(x, 1.0) @ "m"

    def model_of(kernel):
        @genjax.gen
        def model():
            a = genjax.normal(0.0, 1.0) @ "a"
            kernel.vmap()(np.zeros(n, np.float32)) @ "k"
            return genjax.normal(0.0, 1.0) @ "t"
        return model

    draws = []
    for kernel in (one, two):
        prog, _, _ = model_of(kernel).pack((), C.n(), True, rng_mode=A.RNG_JAX32)
        o = cpu.run_program(prog, (3, 4), K)
        draws.append((o["choices"][prog.slot_of["a"]].copy(), o["choices"][prog.slot_of["t"]].copy()))
    np.testing.assert_array_equal(draws[0][0], draws[1][0])
    np.testing.assert_array_equal(draws[0][1], draws[1][1])       # "t" is traced site 3 of its caller in both programs


def test_a_row_of_a_latent_choice_picked_by_a_discrete_choice():
    """GJX_P_VGATHER (`mu[z]`, the component mean of a mixture with LATENT means): the host lowering in its four storage cases, the
    oracle's values against numpy, its gradient against central differences, and the emitted kernels (select chains: registers
    cannot be indexed)"""
    from genjax_amd import kernels
    from oracle import cpu
    N, K, n = 40, 500, 64
    rs = np.random.default_rng(0)
    ys = (2.0 * rs.standard_normal(N)).astype(np.float32)
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

    def score(c):
        mu, ls, z = c[:3], c[3], c[4:].astype(int)
        lx = (-0.5 * ((ys[:, None] - np.take_along_axis(mu, z, axis=0)) / np.exp(ls)) ** 2 - ls - 0.5 * np.log(2 * np.pi)).sum(axis=0)
        lz = (logits - np.log(np.exp(logits).sum()))[z].sum(axis=0)
        lp = (-0.5 * (mu / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)).sum(axis=0) + (-0.5 * ls ** 2 - 0.5 * np.log(2 * np.pi))
        return lx, lx + lz + lp

    for rng in (A.RNG_FLAT, A.RNG_JAX32):
        prog, _, _ = model.pack((), C["k", "x"].set(ys), True, rng_mode=rng)
        q = prog.c_sites[3].p[0]
        assert prog.n_sites == 4 and (q.op, q.slot, q.moff, q.n, q.len, q.d_slot, q.d_moff) == (A.P_VGATHER, 4, 0, 3, 1, 1, 0)
        o = cpu.run_program(prog, (0, 5), K)
        lx, sc = score(o["choices"].astype(np.float64))
        np.testing.assert_allclose(o["weight"], lx, rtol=2e-5, atol=2e-3)
        np.testing.assert_allclose(o["score"], sc, rtol=2e-5, atol=2e-3)
    assert "vg_3_0 = gi_3_0[p] == 2 ? v[2][p] : vg_3_0" in kernels.program_source(prog, 1)
    kernels.program_precompile(prog, 1)
    # the HMC program: z fixed per chain
    lat = ["mu", "ls"] + [(("k", "z"), i) for i in range(N)]
    hp, _, _ = model.pack((), C["k", "x"].set(ys), False, selected=("mu", "ls"), per_particle=tuple(lat), plates="hmc")
    ch = np.zeros((hp.n_slots, n), np.float32)
    ch[:3], ch[3], ch[4:] = rs.standard_normal((3, n)), 0.1 * rs.standard_normal(n), rs.integers(0, 3, (N, n))
    sc, g = cpu.score_grad(hp, ch)
    c64 = ch.astype(np.float64)
    np.testing.assert_allclose(sc, score(c64)[1], rtol=2e-5, atol=2e-3)
    for r in range(4):
        e = np.zeros_like(c64)
        e[r] = 1e-5
        num = (score(c64 + e)[1] - score(c64 - e)[1]) / 2e-5
        np.testing.assert_allclose(g[r], num, rtol=2e-4, atol=2e-4 * np.abs(num).max())
    assert not g[4:].any()
    src = kernels.program_hmc_source(hp)
    assert "ga[2] += gi_3_0 == 2 ? w_ : 0.0f" in src and "pre_0 = gi_3_0 == 1 ? v[1] : pre_0" in src
    assert "// PROWS 0" in src and "// CPLMAX 4" in src          # (40 instances: four lanes per chain is all the loop can use)
    kernels.program_hmc_precompile(hp)
    # z constrained to ONE assignment for every chain: it owns no storage, the index of the row comes from the table
    zfix = rs.integers(0, 3, N).astype(np.float32)
    hp2, _, _ = model.pack((), C["k", "x"].set(ys) | C["k", "z"].set(zfix), False, selected=("mu", "ls"), per_particle=("mu", "ls"), plates="hmc")
    q = hp2.c_sites[3].p[0]
    assert (q.op, q.slot, q.d_off, hp2.n_slots) == (A.P_VGATHER, -1, 1, 4) and np.array_equal(hp2.tab[q.off:q.off + N], zfix)
    sc2, g2 = cpu.score_grad(hp2, ch[:4].copy())
    full = np.concatenate([ch[:4], np.repeat(zfix[:, None], n, axis=1)]).astype(np.float64)
    np.testing.assert_allclose(sc2, score(full)[1], rtol=2e-5, atol=2e-3)
    assert "(int)TAB((" in kernels.program_hmc_source(hp2)
    # mu constrained for everybody, z latent: an ordinary row gather whose table IS the observed value (set_obs keeps working)
    p3, _, _ = model.pack((), C["k", "x"].set(ys) | C["mu"].set(np.array([-1.0, 0.5, 2.0], np.float32)), True)
    q = p3.c_sites[3].p[0]
    assert q.op == A.P_GATHER and np.array_equal(p3.tab[q.off:q.off + 3], [-1.0, 0.5, 2.0])
    o3 = cpu.run_program(p3, (0, 5), K)
    p3.set_obs("mu", np.array([0.0, 0.0, 0.0], np.float32))
    o4 = cpu.run_program(p3, (0, 5), K)
    assert np.abs(o3["weight"] - o4["weight"]).max() > 1.0


def test_plate_assess_equals_the_unrolled_program():
    """no randomness: a trace's score on the plate lowering == on the unrolled one (per-instance tables, observations, masks,
    an affine row per instance over two latent sites outside the plate)"""
    from oracle import cpu
    from genjax_amd.core import Mask
    n, P, K = 24, 3, 300
    rs = np.random.default_rng(5)
    X = rs.standard_normal((n, P)).astype(np.float32)
    tabs = rs.standard_normal((n, 2)).astype(np.float32)

    @genjax.gen
    def kernel(x_row, tab, beta, b0):
        c = genjax.flip(0.4) @ "c"
        genjax.normal(genjax.take(tab, c), 1.0) @ "m"
        return genjax.normal(x_row @ beta + b0, 0.8) @ "y"

    @genjax.gen
    def model():
        beta = genjax.normal(np.zeros(P, np.float32), 1.0) @ "beta"
        b0 = genjax.normal(0.0, 2.0) @ "b0"
        kernel.vmap(in_axes=(0, 0, None, None))(X, tabs, beta, b0) @ "k"

    chm = C["k", "y"].set(rs.standard_normal(n).astype(np.float32))
    prog_p, _, _ = model.pack((), chm, True)
    prog_u, _, _ = model.pack((), chm, True, plates=False)
    assert prog_p.n_sites == 5 and prog_u.n_sites == 2 + 3 * n and [prog_p.c_sites[j].plate for j in range(5)] == [0, 0, 1, 1, 1]
    y_site = prog_p.c_sites[4]
    assert (y_site.p[0].op, y_site.p[0].d_moff, y_site.p[0].d_slot, y_site.p[0].n) == (A.P_AFFINE, P + 1, 0, P + 1)   # [beta, b0] span, a row per instance
    drawn = cpu.run_program(prog_p, (0, 9), K)
    # the same values on the unrolled program, every latent constrained per particle
    lat = [s.addr for s in prog_u.site_list.sites if prog_u.slot_of[s.addr] >= 0]
    prog_a, _, _ = model.pack((), chm, False, per_particle=tuple(lat), plates=False)
    ch = np.zeros((prog_a.n_slots, K), np.float32)
    for a_ in lat:
        d = prog_a.site_list[a_].dim
        ch[prog_a.slot_of[a_]:prog_a.slot_of[a_] + d] = drawn["choices"][prog_p.slot_of[a_]:prog_p.slot_of[a_] + d]
    again = cpu.run_program(prog_a, (0, 0), K, choices=ch)
    np.testing.assert_allclose(again["score"], drawn["score"], rtol=3e-6, atol=2e-4)


def test_reads_of_plate_instances_from_outside_and_from_a_later_plate_pack():
    """`zs = k.vmap()(xs); y ~ normal(zs[3], 1)` and `ws = k2.vmap()(zs)` crashed at pack time with plates on"""
    from oracle import cpu
    n = 16
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
        k2.vmap()(zs) @ "ws"
        return genjax.normal(zs[3], 1.0) @ "y"

    for rng in (A.RNG_FLAT, A.RNG_JAX32):
        prog, _, _ = model.pack((), C.n(), True, rng_mode=rng)
        assert prog.n_sites == 4
        o = cpu.run_program(prog, (0, 4), 4000)
        ch = o["choices"]
        z, w = ch[prog.slot_of[(("zs", "z"), 5)]], ch[prog.slot_of[(("ws", "w"), 5)]]
        assert abs(np.corrcoef(z, w)[0, 1]) > 0.5                                   # w_5 ~ N(z_5, .) reads instance 5 of the first plate
        assert abs(np.corrcoef(ch[prog.slot_of[(("zs", "z"), 3)]], ch[prog.slot_of["y"]])[0, 1] - 2 ** -0.5) < 0.05


def test_generated_sources_compile_for_gfx950():
    """hipRTC cross-compiles without a GPU: the plate loop, the INPUT / ancestor-gather read side with tile totals, and the
    matrix-core flavour of a big affine site"""
    from genjax_amd import kernels, workloads
    from genjax_amd.inference.scan_filter import ScanBootstrapFilter
    model, chm, *_ = _mixture(64)
    for rng in (A.RNG_FLAT, A.RNG_JAX32):
        prog, _, _ = model.pack((), chm, True, rng_mode=rng)
        src = kernels.program_source(prog, 2)
        assert "for (int i_ = 0; i_ < 64; ++i_)" in src and ("ik_[p]" in src) == (rng == A.RNG_JAX32)
        kernels.program_precompile(prog, 2)
    s = workloads.ssm_problem(dx=4, T=3)
    Am = np.asarray(s["A"], np.float32)

    @genjax.gen
    def step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(4, 0.5, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(4, 2.0, np.float32)) @ "y"
        return x, None

    progs = ScanBootstrapFilter(step.scan(n=3), 1024).step_programs(C["y"].set(np.asarray(s["y"], np.float32)), (np.zeros(4, np.float32), None))
    assert [p.n_input_rows for p in progs] == [0, 4, 4] and progs[1].c_sites[0].mode == A.MODE_INPUT
    assert bytes(progs[1].c_sites)[:3 * 240] == bytes(progs[2].c_sites)[:3 * 240]      # periodic steps: ONE generated kernel
    src = kernels.program_source(progs[1], 4)
    assert "a.in_rows ? LDIN(a.in_rows + " in src and "a.tile_S[tix]" in src and "gjx_gen_steps" in src and "tiled_search_tile<false, true>" in src
    kernels.program_precompile(progs[1], 4)
    lr, _ = workloads.logreg_importance_program()
    src = kernels.program_source(lr, 257)
    assert "__builtin_amdgcn_mfma_f32_16x16x4f32" in src and "mfma_s" in src
    kernels.program_precompile(lr, 257)


def test_scan_step_program_on_the_oracle():
    """the oracle treats INPUT sites as rows that are already there (no draw, no score, no site number): x_t - A x_in has the
    transition's spread, the weight is the observation's log-density"""
    from genjax_amd import workloads
    from genjax_amd.inference.scan_filter import ScanBootstrapFilter
    from oracle import cpu
    s = workloads.ssm_problem(dx=8, T=3)
    Am, q, r = np.asarray(s["A"], np.float32), float(s["q"]), float(s["r"])

    @genjax.gen
    def step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(8, q, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(8, r, np.float32)) @ "y"
        return x, None

    ys = np.asarray(s["y"], np.float32)
    progs = ScanBootstrapFilter(step.scan(n=3), 64).step_programs(C["y"].set(ys), (np.zeros(8, np.float32), None))
    K = 20000
    ch = np.random.default_rng(0).standard_normal((progs[1].n_slots, K)).astype(np.float32)
    o = cpu.run_program(progs[1], (1, 2), K, choices=ch, want_site_scores=True)
    x_in, x = ch[:8], o["choices"][8:16]
    np.testing.assert_array_equal(o["choices"][:8], x_in)
    assert abs(np.std(x - Am @ x_in) - q) < 0.01 and (o["site_scores"][0] == 0).all()
    lw = (-0.5 * ((ys[1][:, None] - x) / r) ** 2 - np.log(r) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(o["weight"], lw, rtol=2e-5, atol=2e-4)
    # the same draws as a program WITHOUT the input site would make at site number 1 (INPUT sites take no number)
    o0 = cpu.run_program(progs[0], (1, 2), K)
    np.testing.assert_allclose(o0["choices"][:8], x - Am @ x_in, rtol=0, atol=2e-6)


def test_generic_filter_on_the_oracle_alone_matches_kalman():
    """the whole generic filter restated on the CPU — the host's step programs, the oracle's program runner and its tile-scaled
    systematic resampler, the filter's key discipline — against the float64 Kalman log-likelihood: checks the lowering of a Scan
    into step programs (carry as INPUT sites, observations per step) and the key chain without a GPU"""
    from genjax_amd import workloads
    from genjax_amd.core import fold_in, split
    from genjax_amd.inference.pf import _unit_from_key
    from genjax_amd.inference.scan_filter import ScanBootstrapFilter
    from oracle import closed_form as cf
    from oracle import cpu
    T, K = 12, 8192
    s = workloads.ssm_problem(dx=4, T=T)
    Am, q, r = np.asarray(s["A"], np.float32), float(s["q"]), float(s["r"])

    @genjax.gen
    def step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(4, q, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(4, r, np.float32)) @ "y"
        return x, None

    ys = np.asarray(s["y"], np.float32)
    progs = ScanBootstrapFilter(step.scan(n=T), K).step_programs(C["y"].set(ys), (np.zeros(4, np.float32), None))
    assert len(progs) == T and [p.n_input_rows for p in progs] == [0] + [4] * (T - 1)
    ests = []
    for seed in (1, 2, 3):
        k, log_ml, x_prev, lw_prev = genjax.key(seed), 0.0, None, None
        for t in range(T):
            k = fold_in(k, t)
            kp, kr = split(k)
            ch = np.zeros((progs[t].n_slots, K), np.float32)
            if t > 0:
                anc, _, _, dead = cpu.resample_systematic_tiled(lw_prev, _unit_from_key(kr))
                assert not dead
                ch[:4] = x_prev[:, anc]
            o = cpu.run_program(progs[t], kp, K, choices=ch)
            sl = progs[t].slot_of[("x", t)]
            x_prev, lw_prev = o["choices"][sl:sl + 4], o["weight"]
            m = lw_prev.astype(np.float64).max()
            log_ml += m + np.log(np.exp(lw_prev.astype(np.float64) - m).mean())
        ests.append(log_ml)
    exact, _, _ = cf.kalman_log_lik(s["A"], s["y"], s["q"], s["r"], q0=q)
    assert abs(np.mean(ests) - exact) < 0.01 * abs(exact) and np.std(ests) < 0.01 * abs(exact), (ests, exact)


@pytest.mark.parametrize("rng", [A.RNG_FLAT, A.RNG_JAX32])
def test_scan_inside_a_scan_step_lowers_to_runs_with_their_own_keys(rng):
    """host lowering of a Scan nested in a Scan step (gen.py ScanCombinator._unroll; the reference nests freely, scan.py:237-294):
    a fresh scan id per instantiation of the inner scan and for the rest of the enclosing step behind it — on the oracle every
    site's innovation is an independent unit normal (a repeated (key, site number) pair would show as a correlation of 1)"""
    from genjax_amd.program import PackedProgram
    from oracle import cpu

    @genjax.gen
    def inner(x, _):
        z = genjax.normal(x, 1.0) @ "z"
        return z, z

    @genjax.gen
    def outer(x, _):
        a = genjax.normal(x, 0.5) @ "a"
        c, _zs = inner.scan(n=3)(a, None) @ "in"
        b = genjax.normal(c, 0.25) @ "b"
        return b, b

    sl, _ = outer.scan(n=2).site_list((0.0, None))
    tags = [s.scan for s in sl.sites]
    assert len(sl.sites) == 10 and len({t >> 20 for t in tags}) == 5          # outer, 2 x inner, 2 x continuation
    assert [s.addr for s in sl.sites][1:4] == [(("in", "z"), (0, i)) for i in range(3)]
    p = PackedProgram(sl, rng_mode=rng)
    ch = cpu.run_program(p, (3, 4), 100000)["choices"]
    names, sd = [s.addr for s in sl.sites], [0.5, 1, 1, 1, 0.25] * 2
    inn = np.stack([(ch[p.slot_of[names[i]]] - (ch[p.slot_of[names[i - 1]]] if i else 0.0)) / sd[i] for i in range(10)])
    assert np.abs(inn.std(axis=1) - 1.0).max() < 0.02 and np.abs(np.corrcoef(inn) - np.eye(10)).max() < 0.02


def test_set_obs_on_a_plate_program_equals_a_fresh_pack():
    """PackedProgram.set_obs restacks every instance of a folded parameter of a plate site — (a) an observed
    site OUTSIDE the plate read through a gather / an affine form by the body (shared table, d_elem == 0), (b) an observed BODY
    site read by a later body site (derived source named by the device site list only)"""
    n = 12
    rs = np.random.default_rng(2)
    tab = rs.standard_normal(3).astype(np.float32)
    xs = rs.standard_normal(n).astype(np.float32)

    @genjax.gen
    def kernel(x, c, w):
        m = genjax.normal(genjax.take(tab, c), 1.0) @ "m"
        v = genjax.normal(2.0 * w + x, 0.5) @ "v"
        return genjax.normal(3.0 * v + x, 0.3) @ "y"

    @genjax.gen
    def model():
        c = genjax.categorical(logits=np.zeros(3, np.float32)) @ "c"
        w = genjax.normal(0.0, 1.0) @ "w"
        kernel.vmap(in_axes=(0, None, None))(xs, c, w) @ "k"

    def chm(cv, wv, vs):
        return C["c"].set(cv) | C["w"].set(np.float32(wv)) | C["k", "v"].set(vs) | C["k", "y"].set(np.zeros(n, np.float32))

    v0, v1 = rs.standard_normal(n).astype(np.float32), rs.standard_normal(n).astype(np.float32)
    prog, _, _ = model.pack((), chm(0, 0.5, v0), True)
    fresh, _, _ = model.pack((), chm(2, -1.5, v1), True)
    assert any(prog.c_sites[j].plate for j in range(prog.n_sites))
    assert prog.n_sites == fresh.n_sites and prog.tab.size == fresh.tab.size and not np.array_equal(prog.tab, fresh.tab)
    prog.set_obs("c", 2)
    prog.set_obs("w", np.float32(-1.5))
    prog.set_obs(("@plate", ("k", "v")) if ("@plate", ("k", "v")) in prog.obs_off else (("k", "v"), 0), v1)
    np.testing.assert_array_equal(prog.tab, fresh.tab)


def test_filter_kernel_of_a_step_program_is_emitted_and_cross_compiles():
    """GJX_FILTER_FORM_WIDE (include/gjx.h): the step program of a Scan kernel as the MODEL of the filter skeleton the hand-written
    linear-Gaussian filter runs on (csrc/gjx_pfcore.h; reference recursion: scan.py:237-294).  On the CPU: the emitted source has the
    model's pieces — table staged per step, the normal draws of the sampled site hoisted into the granule wait, the carry read
    through the ancestor with scoped loads, rows stored with scoped stores — and hipRTC compiles it for gfx950."""
    from genjax_amd import kernels
    from genjax_amd.inference.scan_filter import ScanBootstrapFilter
    dx, T = 4, 5
    rs = np.random.default_rng(1)
    Am = (0.5 * rs.standard_normal((dx, dx))).astype(np.float32)

    @genjax.gen
    def step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(dx, 0.5, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(dx, 2.0, np.float32)) @ "y"
        return x, None

    bf = ScanBootstrapFilter(step.scan(n=T), 4096)
    progs = bf.step_programs(C["y"].set(rs.standard_normal((T, dx)).astype(np.float32)), (np.zeros(dx, np.float32), None))
    src = kernels.program_filter_source(progs[1], 2)
    assert "pf_core<GenPfModel, SPL, 0, 0, false>" in src and "#define SPL 2" in src and "#define NHOIST 4" in src      # one rank: agent scope, no verify mode
    assert "pf_core<GenPfModel, SPL, 2, 2, false>" in kernels.program_filter_source(progs[1], 2 | 256)              # sharded flavour: decided at run time
    assert "pf_core<GenPfModel, SPL, 0, 0, true>" in kernels.program_filter_source(progs[1], 2 | 1024)              # multinomial resampling by sorted uniforms
    assert "pf_core<GenPfModel, SPL, 2, 2, true>" in kernels.program_filter_source(progs[1], 1 | 256 | 1024)        # ... on a sharded collection
    # the step's draws: the stream's words are taken ahead (behind the publish), Box-Muller is finished inside the slot
    assert "d.nz[3] = __uint_as_float(bs.get(3u));" in src and "box_muller(__float_as_uint(dr_->nz[0]), __float_as_uint(dr_->nz[1]), hn0_, hn1_);" in src
    assert "LDIN(a.in_rows + (int64_t)3 * a.in_stride + src_[p])" in src and "tab_s[e] = tb_[e]" in src
    assert "score[p] +=" not in src and "q2[p]" in src           # weights only: the sampled site's log-density is not evaluated (the observed one's is)
    assert src.count("stream_normal<RNG>(") == 0 and src.count("bs.get(") == dx     # the site itself hashes nothing any more
    assert "const float* __restrict__ tb_ = tbn_;" in src and "tbn_ = f.tabs[t + 1];" in src     # the table's address is read a step ahead
    with pytest.raises(Exception):
        kernels.program_filter_source(progs[0], 1)                 # step 0 reads no carry: not a filter step
    kernels.program_filter_precompile(progs[1], 2)                 # hipRTC cross-compiles without a GPU
    kernels.program_filter_precompile(progs[2], 2)                 # same structure, other table: the same kernel (cache hit)

    # a scalar model with a discrete site in the carry: nothing hoisted for the categorical site, the normal observation is scored
    P = np.array([[2.0, -1.0], [-1.5, 1.5]], np.float32)
    sig = np.array([0.3, 1.2], np.float32)

    @genjax.gen
    def step2(z_prev, _):
        z = genjax.categorical(logits=genjax.take(P, z_prev) if not isinstance(z_prev, (int, float, np.integer)) else P[int(z_prev)]) @ "z"
        genjax.normal(0.0, genjax.take(sig, z)) @ "y"
        return z, None

    bf2 = ScanBootstrapFilter(step2.scan(n=3), 2048)
    p2 = bf2.step_programs(C["y"].set(np.array([0.1, -0.7, 1.9], np.float32)), (0, None))
    src2 = kernels.program_filter_source(p2[1], 1)
    assert "#define NHOIST 0" in src2 and "gjx_gen_pf" in src2
    kernels.program_filter_precompile(p2[1], 1)


def test_wide_plate_kernel_deals_the_instances_to_the_waves_of_a_block():
    """the 2-D form of a plate program (gjx_program_precompile ppt | 512; reference: Vmap.generate sums the instances' weights,
    vmap.py:180-218): a block of 16 waves shares 64 x PPT particles, every wave runs a contiguous chunk of the instances, partial
    sums meet in LDS in wave order.  On the CPU: the emitted source and its hipRTC compilation"""
    from genjax_amd import kernels
    model, chm, ys, mu, logits = _mixture(600)
    prog, _, _ = model.pack((), chm, True)
    src = kernels.program_source(prog, 2 | 512)
    assert "__launch_bounds__(1024)" in src and "#define LPP_ 1" in src and "#define OWN_ (threadIdx.x < 64u && (threadIdx.x % LPP_) == 0u)" in src
    assert "for (int i_ = plo_; i_ < phi_; ++i_)" in src and "PRED_(score, psc_);" in src and "PRED_(weight, pwt_);" in src
    assert "const int64_t tile = 64 * (int64_t)PPT;" in src and "if ((threadIdx.x >> 6) >= 4u) return;" in src
    # few particles over very many instances: 4 / 16 lanes per particle as well (| 1024, | 2048): 16 / 4 particles per wave
    src4 = kernels.program_source(prog, 1 | 512 | 1024)
    assert "#define LPP_ 4" in src4 and "const int64_t tile = 16 * (int64_t)PPT;" in src4
    kernels.program_precompile(prog, 1 | 512 | 2048)
    plain = kernels.program_source(prog, 2)
    assert "__launch_bounds__(256)" in plain and "for (int i_ = 0; i_ < 600; ++i_)" in plain and "PRED_(" not in plain
    kernels.program_precompile(prog, 2 | 512)
    kernels.program_precompile(prog, 1 | 512)


def test_hmc_emitter_forms_compile_without_a_gpu():
    """the forms of the generated HMC kernel that round 5 added — selected sites inside a plate (trajectory rows in the workspace), a
    long periodic Scan (rolled: steps dealt to the lanes of a chain), INPUT sites read as values — are emitted and cross-compiled here
    (hipRTC for gfx950); the oracle's score and gradient of the same programs against central differences in float64"""
    from genjax_amd import kernels
    from genjax_amd.program import PackedProgram, Param, SiteList
    from oracle import cpu
    rs = np.random.default_rng(3)
    # (1) a regression with a latent per datum, the latents moved too
    N, P = 40, 3
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
    prog, _, _ = model.pack((), C["k", "y"].set(y), False, selected=tuple(lat), per_particle=tuple(lat), plates="hmc")
    src = kernels.program_hmc_source(prog)
    assert "// PROWS %d" % N in src and "wq_[(int64_t)(0 + i_ * 1 + 0) * n_ + ic_]" in src and "// CPLMAX 4" in src
    kernels.program_hmc_precompile(prog)
    # (2) every state of a 150-step stochastic-volatility Scan
    T = 150
    ys = rs.standard_normal(T).astype(np.float32)

    @genjax.gen
    def sv():
        @genjax.gen
        def step(x_prev, _):
            x = genjax.normal(0.95 * x_prev, 0.3) @ "x"
            genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
            return x, None
        step.scan(n=T)(0.0, None) @ "s"

    xs = [(("s", "x"), t) for t in range(T)]
    ps, _, _ = sv.pack((), C["s", "y"].set(ys), False, selected=tuple(xs), per_particle=tuple(xs))
    assert ps.n_sites == 2 * T and kernels.hmc_engine(ps) == 4
    src = kernels.program_hmc_source(ps)
    assert "rolled Scan: %d steps x 2 sites, 1 selected values per step" % T in src and "// PROWS %d" % T in src and "// CPLMAX 16" in src
    kernels.program_hmc_precompile(ps)
    ch = (0.3 * rs.standard_normal((ps.n_slots, 5))).astype(np.float32)
    sc, g = cpu.score_grad(ps, ch)

    def score(x):                      # x [T][n] float64
        prev = np.vstack([np.zeros((1, x.shape[1])), x[:-1]])
        lt = -0.5 * ((x - 0.95 * prev) / 0.3) ** 2 - np.log(0.3) - 0.5 * np.log(2 * np.pi)
        lo = -0.5 * (ys[:, None] / np.exp(0.5 * x)) ** 2 - 0.5 * x - 0.5 * np.log(2 * np.pi)
        return (lt + lo).sum(axis=0)

    x64 = ch.astype(np.float64)
    np.testing.assert_allclose(sc, score(x64), rtol=2e-5, atol=2e-3)
    for t in (0, 1, 77, T - 1):
        e = np.zeros_like(x64)
        e[t] = 1e-5
        np.testing.assert_allclose(g[t], (score(x64 + e) - score(x64 - e)) / 2e-5, rtol=2e-3, atol=2e-3)
    # (3) an INPUT site: a value the other sites read, no density, never selected
    sl = SiteList()
    sl.add("xp", A.MVNORMAL_DIAG, [np.zeros(2, np.float32), np.ones(2, np.float32)])
    sl.add("x", A.MVNORMAL_DIAG, [Param.affine(np.array([[0.5, 0.1], [0.0, 0.7]], np.float32), "xp"), np.full(2, 0.7, np.float32)])
    pi_ = PackedProgram(sl, {"xp": A.MODE_INPUT, "x": A.MODE_OBS_SLOT}, selected=("x",))
    assert "INPUT, 2 rows from slot 0" in kernels.program_hmc_source(pi_)
    kernels.program_hmc_precompile(pi_)
    ci = rs.standard_normal((4, 6)).astype(np.float32)
    si, gi = cpu.score_grad(pi_, ci)
    want = (-0.5 * ((ci[2:] - np.array([[0.5, 0.1], [0.0, 0.7]]) @ ci[:2]) / 0.7) ** 2 - np.log(0.7) - 0.5 * np.log(2 * np.pi)).sum(axis=0)
    np.testing.assert_allclose(si, want, rtol=2e-5, atol=2e-4)
    assert not gi[:2].any()


def test_step_programs_fold_row_gathers_of_known_values():
    """the one-step programs of a filter over a kernel with `vec[idx]`: the row gather of the step's own latent is a GJX_P_VGATHER in every
    step program (vector and index slots of THAT step); the filter kernel is emitted for it (select chain over the step's registers)
    and compiles"""
    from genjax_amd import kernels
    from genjax_amd.inference.scan_filter import ScanBootstrapFilter
    Am = np.array([[0.9, 0.1], [-0.1, 0.8]], np.float32)

    @genjax.gen
    def step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(2, 0.5, np.float32)) @ "x"
        c = genjax.flip(0.4) @ "c"
        genjax.normal(x[c], 0.6) @ "y"
        return x, None

    bf = ScanBootstrapFilter(step.scan(n=3), 2048)
    ps = bf.step_programs(C["y"].set(np.array([0.1, -0.7, 1.9], np.float32)), (np.zeros(2, np.float32), None))
    for t, p in enumerate(ps):
        ysite = [j for j in range(p.n_sites) if p.c_sites[j].mode == A.MODE_OBS_TAB]
        assert len(ysite) == 1
        q = p.c_sites[ysite[0]].p[0]
        assert q.op == A.P_VGATHER and q.n == 2 and q.len == 1 and q.moff == p.slot_of[("x", t)] and q.slot == p.slot_of[("c", t)]
    src = kernels.program_filter_source(ps[1], 1)
    assert "gjx_gen_pf" in src and "vg_" in src
    kernels.program_filter_precompile(ps[1], 1)
