"""The reference's combinator tests (tests/generative_functions/test_vmap_combinator.py, test_repeat_combinator.py,
test_scan_combinator.py) restated against genjax_amd, for the forms the site-program lowering supports (mapped
arguments are host data; nesting up to three index levels, TestNestedCombinators).  Line numbers of the originals are cited."""
import numpy as np
import pytest

import genjax_amd as genjax
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd import IndexRequest, Regenerate, Selection, StaticRequest, Update
from genjax_amd import Selection as S

pytestmark = pytest.mark.gpu


def f(t):
    return float(t.detach().cpu()) if hasattr(t, "detach") else float(t)


def lp(v, m):
    return f(genjax.normal.assess(C.v(v), (m, 1.0))[0])


class TestVmap:
    def test_vmap_combinator_simple_normal(self):                            # test_vmap_combinator.py:29-40
        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def model(x):
            z = genjax.normal(x, 1.0) @ "z"
            return z

        map_over = np.arange(0, 50, dtype=np.float32)
        tr = model.simulate(genjax.key(314159), (map_over,))
        z = tr.get_choices()[:, "z"]
        assert z.shape == (50,)
        inner = sum(lp(f(z[i]), float(map_over[i])) for i in range(50))
        assert f(tr.get_score()) == pytest.approx(inner, rel=1e-5)

    def test_vmap_simple_normal_project(self):                               # :42-58
        @genjax.gen
        def model(x):
            z = genjax.normal(x, 1.0) @ "z"
            return z

        vmapped = model.vmap(in_axes=(0,))
        key = genjax.key(314159)
        tr = vmapped.simulate(key, (np.arange(0, 10, dtype=np.float32),))
        assert f(tr.project(key, Selection.all())) == pytest.approx(f(tr.get_score()), rel=1e-6)
        assert f(tr.project(key, Selection.none())) == 0.0

    def test_vmap_combinator_vector_choice_map_importance(self):             # :60-79
        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def kernel(x):
            z = genjax.normal(x, 1.0) @ "z"
            return z

        map_over = np.arange(0, 3, dtype=np.float32)
        chm = C.n()
        for idx, v in zip(range(3), [3.0, 2.0, 3.0]):
            chm = chm | C[idx, "z"].set(v)
        (_, w) = kernel.importance(genjax.key(314159), chm, (map_over,))
        assert f(w) == pytest.approx(lp(3.0, 0.0) + lp(2.0, 1.0) + lp(3.0, 2.0), rel=1e-6)
        (_, w2) = kernel.importance(genjax.key(314159), C[:, "z"].set(np.array([3.0, 2.0, 3.0], np.float32)), (map_over,))
        assert f(w2) == f(w)

    def test_vmap_combinator_indexed_choice_map_importance(self):            # :81-101
        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def kernel(x):
            z = genjax.normal(x, 1.0) @ "z"
            return z

        key, sub_key = genjax.split(genjax.key(314159))
        map_over = np.arange(0, 3, dtype=np.float32)
        (_, w) = kernel.importance(sub_key, C[0, "z"].set(3.0), (map_over,))
        assert f(w) == pytest.approx(lp(3.0, 0.0), rel=1e-6)
        zv = [3.0, -1.0, 2.0]
        chm = C.n()
        for idx, v in enumerate(zv):
            chm = chm | C[idx, "z"].set(v)
        (tr, _) = kernel.importance(sub_key, chm, (map_over,))
        for i in range(3):
            assert f(tr.get_choices()[i, "z"]) == zv[i]

    def test_vmap_combinator_assess(self):                                   # :158-170
        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def model(x):
            z = genjax.normal(x, 1.0) @ "z"
            return z

        map_over = np.arange(0, 50, dtype=np.float32)
        tr = model.simulate(genjax.key(314159), (map_over,))
        assert f(model.assess(tr.get_choices(), (map_over,))[0]) == pytest.approx(f(tr.get_score()), rel=1e-6)

    def test_vmap_validation(self):                                          # :172-206 (length mismatch is an error)
        @genjax.gen
        def foo(loc, scale):
            return genjax.normal(loc, scale) @ "x"

        with pytest.raises(ValueError):
            foo.vmap(in_axes=(0, 0)).simulate(genjax.key(1), (np.zeros(3, np.float32), np.ones(4, np.float32)))
        with pytest.raises(ValueError):
            foo.vmap(in_axes=(0, None, None)).simulate(genjax.key(1), (np.zeros(3, np.float32), 1.0))

    def test_vmap_regenerate_and_update(self):                               # :278-326 (n 1000 -> 200: one site per instance)
        n = 200

        @genjax.gen
        def model():
            x = genjax.normal(0.0, 1.0) @ "x"
            _ = genjax.normal.vmap()(np.zeros(n, np.float32), np.ones(n, np.float32)) @ "a"
            return x

        key, sub_key = genjax.split(genjax.key(314159))
        tr = model.simulate(sub_key, ())
        for idx in range(5):
            old_a = f(tr.get_choices()["a", idx])
            request = StaticRequest({"a": IndexRequest(idx, Regenerate(S.all()))})
            new_tr, fwd_w, _, _ = request.edit(key, tr, ())
            new_a = f(new_tr.get_choices()["a", idx])
            assert new_a != old_a
            assert f(fwd_w) == pytest.approx(lp(new_a, 0.0) - lp(old_a, 0.0), rel=1e-4, abs=1e-5)
            assert f(new_tr.get_choices()["a", idx + 1]) == f(tr.get_choices()["a", idx + 1])
            request = StaticRequest({"a": IndexRequest(idx, Update(C.v(idx + 7.0)))})
            new_tr, fwd_w, _, _ = request.edit(key, tr, ())
            assert f(new_tr.get_choices()["a", idx]) == idx + 7.0
            assert f(fwd_w) == pytest.approx(lp(idx + 7.0, 0.0) - lp(old_a, 0.0), rel=1e-4, abs=1e-5)


class TestRepeatCombinator:
    def test_repeat_combinator_importance(self):                             # test_repeat_combinator.py:23-30
        @genjax.gen
        def model():
            return genjax.normal(0.0, 1.0) @ "x"

        tr, w = model.repeat(n=10).importance(genjax.key(314), C[1, "x"].set(3.0), ())
        assert f(tr.get_choices()[1, "x"]) == 3.0
        assert f(w) == pytest.approx(lp(3.0, 0.0), rel=1e-6)
        assert tr.get_choices()[:, "x"].shape == (10,)


@genjax.iterate(n=10)
@genjax.gen
def scanner(x):
    z = genjax.normal(x, 1.0) @ "z"
    return z


class TestIterateSimpleNormal:
    def test_iterate_simple_normal(self):                                    # test_scan_combinator.py:41-52
        key, sub_key = genjax.split(genjax.key(314159))
        tr = scanner.simulate(sub_key, (0.01,))
        assert f(tr.project(key, genjax.Selection.all())) == pytest.approx(f(tr.get_score()), rel=1e-6)
        rv = tr.get_retval()
        assert len(rv) == 11 and f(rv[0]) == pytest.approx(0.01) and f(rv[3]) == f(tr.get_choices()[2, "z"])

    def test_iterate_simple_normal_importance(self):                         # :54-61
        key, sub_key = genjax.split(genjax.key(314159))
        for i in range(1, 5):
            tr, w = scanner.importance(sub_key, C[i, "z"].set(0.5), (0.01,))
            value = f(tr.get_choices()[i, "z"])
            assert value == 0.5
            prev = f(tr.get_choices()[i - 1, "z"])
            assert f(w) == pytest.approx(lp(value, prev), rel=1e-5, abs=1e-6)

    def test_iterate_simple_normal_update(self):                             # :63-82
        key, sub_key = genjax.split(genjax.key(314159))
        for i in range(1, 5):
            tr, _w = scanner.importance(sub_key, C[i, "z"].set(0.5), (0.01,))
            new_tr, w, _rd, discard = scanner.update(sub_key, tr, C[i, "z"].set(1.0), None)
            ch = new_tr.get_choices()
            assert f(ch[i, "z"]) == 1.0 and f(discard[i, "z"]) == 0.5
            prev, nxt = f(ch[i - 1, "z"]), f(ch[i + 1, "z"])
            want = (lp(1.0, prev) + lp(nxt, 1.0)) - (lp(0.5, prev) + lp(nxt, 0.5))
            assert f(w) == pytest.approx(want, rel=1e-4, abs=1e-5)

    def test_scan_regenerate(self):                                          # :468-495 shape: one step regenerated
        key, sub_key = genjax.split(genjax.key(3))
        tr = scanner.simulate(sub_key, (0.0,))
        old = tr.get_choices()
        new_tr, w, _, _ = Regenerate(Selection.at[4, "z"]).edit(key, tr, None)
        new = new_tr.get_choices()
        assert f(new[4, "z"]) != f(old[4, "z"]) and f(new[3, "z"]) == f(old[3, "z"]) and f(new[5, "z"]) == f(old[5, "z"])
        z3, z5 = f(old[3, "z"]), f(old[5, "z"])
        want = (lp(f(new[4, "z"]), z3) + lp(z5, f(new[4, "z"]))) - (lp(f(old[4, "z"]), z3) + lp(z5, f(old[4, "z"])))
        assert f(w) == pytest.approx(want, rel=1e-4, abs=1e-5)
        assert f(new_tr.get_score()) == pytest.approx(f(tr.get_score()) + want, rel=1e-4, abs=1e-4)


# ---------------------------------------------------------------------------------------------------------------
# Device plates (program.py compact_plates): the instances of a vmapped kernel as ONE vector site per kernel site, against
# the ORACLE (oracle/gjx_oracle.c restates the site semantics of distribution.py:117-147 / static.py:340-399) on the very
# program the device ran, and against the unrolled lowering of the same model.
# ---------------------------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def _hier_logreg(N=1024, P=16, seed=0):
    rs = np.random.default_rng(seed)
    X = rs.standard_normal((N, P)).astype(np.float32)
    beta = rs.standard_normal(P)
    y = (rs.uniform(size=N) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(np.float32)

    @genjax.gen
    def obs(x_row, b):
        return genjax.bernoulli(x_row @ b) @ "y"

    @genjax.gen
    def model(Xd):
        lt = genjax.normal(0.0, 1.0) @ "log_tau"
        b = genjax.normal(np.zeros(P, np.float32), genjax.exp(lt)) @ "beta"
        obs.vmap(in_axes=(0, None))(Xd, b) @ "ys"
        return b

    return model, X, y


class TestVmapPlates:
    def test_vmapped_hierarchical_logreg_is_three_sites_and_matches_the_oracle(self):
        """vmap.py:193-218 with N = 1024 observed instances + 2 latents: the device program has 3 sites (the plate is one
        bernoulli site with a [1024 x 16] affine parameter), runs on a GENERATED kernel, and its particles, scores and
        weights equal the oracle's on that program and the unrolled lowering's on the same key."""
        from genjax_amd import kernels
        from oracle import cpu
        model, X, y = _hier_logreg()
        chm = C["ys", "y"].set(y)
        prog, _, _ = model.pack((X,), chm, True)
        assert len(prog.site_list.sites) == 1026 and prog.n_sites == 3 and len(prog.plate_of) == 1024
        assert kernels.program_engine(prog) == 4                                  # a kernel generated from the 3-site list
        K = 4096
        tr, w = model.importance(genjax.key(7), chm, (X,), K=K)
        assert tr.prog.n_sites == 3
        o = cpu.run_program(tr.prog, genjax.key(7), K)
        ch = _np(tr.choices)
        np.testing.assert_allclose(ch, o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(w), o["weight"], rtol=3e-4, atol=2e-2)        # a sum of 1024 log-densities around -700
        np.testing.assert_allclose(_np(tr.get_score()), o["score"], rtol=3e-4, atol=2e-2)
        # the unrolled lowering of the same model under the same key (at N = 256: 1026 sites are more than the FLAT stream's
        # 1023 site numbers, which is one of the things the plate removes): the latents come from the same streams (they
        # precede the plate), the weight is the same sum
        model_s, Xs, ys_ = _hier_logreg(N=256)
        chm_s = C["ys", "y"].set(ys_)
        prog_p, _, _ = model_s.pack((Xs,), chm_s, True)
        prog_u, _, _ = model_s.pack((Xs,), chm_s, True, plates=False)
        assert prog_p.n_sites == 3 and prog_u.n_sites == 258
        op, ou = kernels.run_program(prog_p, genjax.key(7), 512), kernels.run_program(prog_u, genjax.key(7), 512)
        np.testing.assert_allclose(_np(ou["choices"]), _np(op["choices"]), rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(ou["weight"]), _np(op["weight"]), rtol=3e-4, atol=5e-3)
        # logical addressing is unchanged: every instance's value is there
        got = tr.get_choices()
        assert got["ys", 5, "y"].shape == (K,) and bool((got["ys", 5, "y"] == bool(y[5])).all())

    def test_plate_with_a_latent_per_instance_matches_the_oracle(self):
        """a kernel with two sites, the second reading the first of the SAME instance, and a latent outside the plate:
        z_i ~ normal(mu, 1), y_i ~ normal(z_i, 0.5) observed -> vector sites z[n], y[n] with an elementwise VALUE link"""
        from oracle import cpu
        n = 64
        ys = np.random.default_rng(1).standard_normal(n).astype(np.float32) + 1.0

        @genjax.gen
        def kernel(mu):
            z = genjax.normal(mu, 1.0) @ "z"
            return genjax.normal(z, 0.5) @ "y"

        @genjax.gen
        def model():
            mu = genjax.normal(0.0, 2.0) @ "mu"
            kernel.repeat(n=n)(mu) @ "k"
            return mu

        chm = C["k", "y"].set(ys)
        K = 3000
        tr, w = model.importance(genjax.key(11), chm, (), K=K)
        assert tr.prog.n_sites == 3 and len(tr.prog.site_list.sites) == 1 + 2 * n
        o = cpu.run_program(tr.prog, genjax.key(11), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(w), o["weight"], rtol=2e-4, atol=2e-3)
        z = tr.get_choices()["k", :, "z"]                                          # [K][n]
        assert tuple(z.shape) == (K, n)
        lw = sum(genjax.normal.assess(C.v(float(ys[i])), (float(z[0, i]), 0.5))[0] for i in range(n))
        assert f(w[0]) == pytest.approx(f(lw), rel=1e-4)
        # through the inference layer: ImportanceK on the plate program == logsumexp of these weights - log K
        from genjax_amd.inference import ImportanceK, Target
        pc = ImportanceK(Target(model, (), chm), k_particles=K).run_smc(genjax.key(3))
        lw = _np(pc.get_log_weights()).astype(np.float64)
        assert f(pc.get_log_marginal_likelihood_estimate()) == pytest.approx(float(np.log(np.exp(lw - lw.max()).sum()) + lw.max() - np.log(K)), rel=1e-5)

    def test_hmc_on_a_vmapped_model_uses_the_fused_kernel_and_matches_the_oracle(self):
        """HMC.edit (hmc.py:156-211) on the .vmap-written config-5 model: the plate lowering gives the hand-written
        matrix-core kernel its shape back; alpha and the moved values equal the oracle's HMC on the same program"""
        from genjax_amd import HMC, kernels
        from oracle import cpu
        model, X, y = _hier_logreg()
        chm = C["ys", "y"].set(y)
        K = 256
        tr, _ = model.importance(genjax.key(2), chm, (X,), K=K)
        req = HMC(Selection.at["log_tau"] | Selection.at["beta"], 0.004, 20)
        new_tr, alpha, _, _ = req.edit(genjax.key(9), tr, None)
        shared_prog = req.last_program
        assert shared_prog.n_sites == 3 and kernels.hmc_engine(shared_prog) in (2, 3)     # the hand-written config-5 kernels
        o = cpu.hmc(shared_prog, genjax.key(9), _np(tr.rows_for(shared_prog)), 0.004, 20, False, False)
        # alpha is a difference of two scores of size |score| (a sum of 1024 log-densities; chains started from the prior reach
        # several thousand): the tolerance carries the float32 resolution of those sums
        tol = 6e-3 + 6e-3 * np.abs(o["alpha"]) + 2e-5 * np.abs(o["score"])
        assert (np.abs(_np(alpha) - o["alpha"]) <= tol).all(), float(np.abs(_np(alpha) - o["alpha"]).max())
        np.testing.assert_allclose(_np(new_tr.choices), o["choices"], rtol=3e-3, atol=3e-3)

    def test_simple_vmap_simulate_and_assess_match_the_oracle(self):
        """test_vmap_combinator.py:29-40, 158-170 with the oracle as the checker instead of the device itself"""
        from oracle import cpu

        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def model(x):
            return genjax.normal(x, 1.0) @ "z"

        map_over = np.arange(0, 50, dtype=np.float32)
        tr = model.simulate(genjax.key(314159), (map_over,), K=1000)
        assert tr.prog.n_sites == 1 and len(tr.prog.site_list.sites) == 50
        o = cpu.run_program(tr.prog, genjax.key(314159), 1000)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(tr.get_score()), o["score"], rtol=2e-4, atol=2e-3)
        z = tr.get_choices()[:, "z"]
        want = (-0.5 * (z.double().cpu() - np.arange(50)) ** 2 - 0.5 * np.log(2 * np.pi)).sum(dim=1)
        np.testing.assert_allclose(_np(tr.get_score()), want.numpy(), rtol=1e-4, atol=1e-3)
        sc, _ = model.assess(tr.get_particle(3).get_choices(), (map_over,))
        assert f(sc) == pytest.approx(f(tr.get_score()[3]), rel=1e-5)


class TestRepeatAndIterateAgainstTheOracle:
    """the repeat / iterate (Scan) lowerings checked by the ORACLE run on the very program the API traced, not by the
    device itself: values of every step, per-particle weight and score"""

    def test_repeat_importance_matches_the_oracle(self):                     # test_repeat_combinator.py:23-30
        from oracle import cpu

        @genjax.gen
        def model():
            return genjax.normal(0.0, 1.0) @ "x"

        K = 2048
        tr, w = model.repeat(n=10).importance(genjax.key(314), C[1, "x"].set(3.0), (), K=K)
        o = cpu.run_program(tr.prog, genjax.key(314), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(w), o["weight"], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(_np(tr.get_score()), o["score"], rtol=2e-4, atol=2e-4)
        assert np.allclose(o["weight"], lp(3.0, 0.0), rtol=1e-6)              # and the oracle agrees with the closed form

    def test_iterate_simulate_and_importance_match_the_oracle(self):         # test_scan_combinator.py:41-61
        from oracle import cpu
        K = 2048
        tr = scanner.simulate(genjax.key(314159), (0.01,), K=K)
        o = cpu.run_program(tr.prog, genjax.key(314159), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(tr.get_score()), o["score"], rtol=2e-4, atol=3e-4)
        tr2, w2 = scanner.importance(genjax.key(5), C[3, "z"].set(0.5), (0.01,), K=K)
        o2 = cpu.run_program(tr2.prog, genjax.key(5), K)
        np.testing.assert_allclose(_np(tr2.choices), o2["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(w2), o2["weight"], rtol=2e-4, atol=3e-4)
        # the chained step keys (scan.py:268): step t's draw differs from a run in which the steps shared one key
        z = _np(tr.choices)
        assert np.abs(np.diff(z, axis=0)).mean() > 0.5


class TestNestedCombinators:
    """combinators nested in each other (combinators/vmap.py:193-218 and scan.py:237-294 compose freely in the reference):
    a vmap inside a vmap, a scan inside every vmap instance, a vmap inside every scan step.  Addresses carry the indices
    outermost first (``chm[i, j, "x"]``); values, weights and scores are checked by the ORACLE on the traced program."""

    def test_vmap_in_vmap(self):
        from oracle import cpu

        @genjax.gen
        def cell(mu):
            return genjax.normal(mu, 0.5) @ "x"

        @genjax.gen
        def row(mus):
            return cell.vmap(in_axes=0)(mus) @ "cells"

        mus = np.arange(12, dtype=np.float32).reshape(3, 4)
        grid = row.vmap(in_axes=0)
        K = 1024
        tr = grid.simulate(genjax.key(5), (mus,), K=K)
        o = cpu.run_program(tr.prog, genjax.key(5), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(tr.get_score()), o["score"], rtol=2e-4, atol=3e-4)
        ch = tr.get_choices()
        whole = ch[:, :, "cells", "x"]
        assert whole.shape == (K, 3, 4)
        assert np.array_equal(_np(ch[2, 1, "cells", "x"]), _np(whole[:, 2, 1]))
        assert np.array_equal(_np(ch[1, :, "cells", "x"]), _np(whole[:, 1, :]))
        assert abs(float(whole[:, 2, 3].mean()) - 11.0) < 0.1                         # instance (2, 3) has mean mus[2, 3]
        # one instance constrained: the weight is its log-density, every other instance is sampled
        tr2, w = grid.importance(genjax.key(6), C[1, 2, "cells", "x"].set(6.5), (mus,), K=K)
        o2 = cpu.run_program(tr2.prog, genjax.key(6), K)
        np.testing.assert_allclose(_np(w), o2["weight"], rtol=2e-4, atol=1e-5)
        want = -0.5 * ((6.5 - 6.0) / 0.5) ** 2 - np.log(0.5) - 0.5 * np.log(2 * np.pi)
        assert np.allclose(_np(w), want, rtol=1e-5)
        assert float(tr2.get_choices()[1, 2, "cells", "x"][0]) == 6.5
        # the whole grid constrained at once: assess == the sum of the 12 log-densities
        vals = (mus + 0.25).astype(np.float32)
        sc, _ = grid.assess(C["cells", "x"].set(vals), (mus,))
        assert f(sc) == pytest.approx(12 * (-0.5 * 0.25 - np.log(0.5) - 0.5 * np.log(2 * np.pi)), rel=1e-5)

    def test_scan_in_vmap(self):
        from oracle import cpu

        @genjax.gen
        def step(x, _):
            z = genjax.normal(x, 0.3) @ "z"
            return z, z

        @genjax.gen
        def chain(x0):
            return step.scan(n=6)(x0, None) @ "walk"

        x0s = np.array([0.0, 10.0, -5.0], np.float32)
        K = 2048
        tr = chain.vmap(in_axes=0).simulate(genjax.key(8), (x0s,), K=K)
        o = cpu.run_program(tr.prog, genjax.key(8), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        z = tr.get_choices()[:, :, "walk", "z"]                                       # [K][instance][step]
        assert z.shape == (K, 3, 6)
        zz = _np(z)
        assert np.allclose(zz[:, :, 0].mean(axis=0), x0s, atol=0.05)                  # every chain starts at its own x0
        inc = np.diff(zz, axis=2)
        assert 0.27 < inc.std() < 0.33
        # the instances are separate scans with their own chained keys: their increments are uncorrelated
        c = np.corrcoef(inc[:, 0, :].ravel(), inc[:, 1, :].ravel())[0, 1]
        assert abs(c) < 0.05
        # a constraint on one step of one instance
        tr2, w = chain.vmap(in_axes=0).importance(genjax.key(9), C[2, 3, "walk", "z"].set(-4.0), (x0s,), K=K)
        o2 = cpu.run_program(tr2.prog, genjax.key(9), K)
        np.testing.assert_allclose(_np(w), o2["weight"], rtol=3e-4, atol=3e-4)

    def test_vmap_in_scan(self):
        from oracle import cpu

        @genjax.gen
        def obs(x, off):
            return genjax.normal(x + off, 1.0) @ "y"

        @genjax.gen
        def step(x, _):
            z = genjax.normal(x, 0.5) @ "z"
            ys = obs.vmap(in_axes=(None, 0))(z, np.array([0.0, 1.0, 2.0], np.float32)) @ "sensors"
            return z, z

        T, K = 8, 2048
        model = step.scan(n=T)
        yobs = np.linspace(-1.0, 1.0, T * 3).astype(np.float32).reshape(T, 3)
        tr, w = model.importance(genjax.key(4), C["sensors", "y"].set(yobs), (0.0, None), K=K)
        o = cpu.run_program(tr.prog, genjax.key(4), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(w), o["weight"], rtol=3e-4, atol=2e-3)
        z = _np(tr.get_choices()[:, "z"])                                             # [K][T]
        want = sum((-0.5 * (yobs[t, i] - (z[:, t] + i)) ** 2 - 0.5 * np.log(2 * np.pi)) for t in range(T) for i in range(3))
        np.testing.assert_allclose(_np(w), want, rtol=3e-4, atol=2e-3)
        assert float(tr.get_choices()[5, 1, "sensors", "y"][0]) == pytest.approx(float(yobs[5, 1]))

    @pytest.mark.parametrize("rng", [0, 1])
    def test_scan_in_scan(self, rng, monkeypatch):
        """a Scan inside a Scan step (the reference nests freely, scan.py:237-294): every instantiation of the inner scan and the
        rest of the enclosing step behind it are runs of sites with their own chained keys (gen.py ScanCombinator._unroll), so no
        stream repeats.  HIP == oracle on the packed program; innovations of all sites are independent unit normals; constraints
        on one inner step of one outer step weigh as its log-density."""
        from oracle import cpu
        monkeypatch.setenv("GJX_RNG", "jax32" if rng else "flat")

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

        K, T = 1 << 15, 4
        model = outer.scan(n=T)
        tr = model.simulate(genjax.key(1), (0.0, None), K=K)
        assert tr.prog.rng_mode == rng
        o = cpu.run_program(tr.prog, genjax.key(1), K)
        np.testing.assert_allclose(_np(tr.choices), o["choices"], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(_np(tr.score), o["score"], rtol=3e-4, atol=2e-3)
        chm = tr.get_choices()
        z = _np(chm[:, :, "in", "z"])                                                # [K][outer step][inner step]
        a, b = _np(chm[:, "a"]), _np(chm[:, "b"])                                    # [K][T]
        assert z.shape == (K, T, 3) and a.shape == (K, T)
        inn = [a[:, 0] / 0.5]
        for t in range(T):
            if t:
                inn.append((a[:, t] - b[:, t - 1]) / 0.5)
            inn.append(z[:, t, 0] - a[:, t])
            inn += [z[:, t, i] - z[:, t, i - 1] for i in (1, 2)]
            inn.append((b[:, t] - z[:, t, 2]) / 0.25)
        inn = np.stack(inn)
        assert np.abs(inn.std(axis=1) - 1.0).max() < 0.02
        assert np.abs(np.corrcoef(inn) - np.eye(len(inn))).max() < 0.03               # (K = 2^15: 1 / sqrt(K) = 0.0055)
        tr2, w = model.importance(genjax.key(2), C[2, 1, "in", "z"].set(0.75), (0.0, None), K=K)
        o2 = cpu.run_program(tr2.prog, genjax.key(2), K)
        np.testing.assert_allclose(_np(w), o2["weight"], rtol=3e-4, atol=3e-4)
        z2 = _np(tr2.get_choices()[:, :, "in", "z"])
        assert (z2[:, 2, 1] == 0.75).all()
        np.testing.assert_allclose(_np(w), -0.5 * (0.75 - z2[:, 2, 0]) ** 2 - 0.5 * np.log(2 * np.pi), rtol=3e-4, atol=3e-4)


class TestIterateAccumulateReduceMethods:
    """reference tests/generative_functions/test_scan_combinator.py:95-123, 212-247 (the method forms) and :324-438 (scan with
    parameters, inferred length, zero length, validation)"""

    def test_iterate_and_iterate_final(self):                                # :99-123
        @genjax.gen
        def inc(x):
            return x + 1

        assert f(inc.simulate(genjax.key(314159), (0,)).get_retval()) == 1
        rv = inc.iterate(n=4).simulate(genjax.key(314159), (0,)).get_retval()
        assert [f(v) for v in rv] == [0, 1, 2, 3, 4]
        assert f(inc.iterate_final(n=10).simulate(genjax.key(314159), (0,)).get_retval()) == 10

    def test_accumulate_and_reduce(self):                                    # :217-247
        @genjax.gen
        def add(x, y):
            return x + y

        assert f(add.simulate(genjax.key(314159), (0, 2)).get_retval()) == 2
        rv = add.accumulate().simulate(genjax.key(314159), (0, np.ones(4, np.float32))).get_retval()
        assert [f(v) for v in rv] == [0, 1, 2, 3, 4]
        assert f(add.reduce().simulate(genjax.key(314159), (0, np.ones(10, np.float32))).get_retval()) == 10

    def test_scan_update_below_an_address(self):                             # :324-349 (the Pytree argument as a dict)
        @genjax.gen
        def step(b, a):
            return genjax.normal(b + a["x"], 1e-6) @ "b", None

        @genjax.gen
        def model(k):
            return step.scan(n=3)(k, {"x": np.array([1.0, 2.0, 3.0], np.float32)}) @ "steps"

        tr = model.simulate(genjax.key(1), (1.0,))
        u, w, _, _ = tr.update(genjax.key(2), C["steps", 1, "b"].set(99.0))
        got = u.get_choices()["steps", :, "b"].cpu().numpy()
        np.testing.assert_allclose(got, [2.0, 99.0, 7.0], atol=0.1)
        assert f(w) < -100.0

    def test_scan_with_parameters(self):                                     # :357-378
        @genjax.gen
        def step(data, state, update):
            new_state = state + genjax.normal(update, data["noise"]) @ "state"
            return new_state, new_state

        @genjax.gen
        def model(data):
            stepper = step.partial_apply(data)
            return stepper.scan(n=3)(data["initial"], data["updates"]) @ "s"

        tr = model.simulate(genjax.key(314159), ({"initial": 3.0, "updates": np.array([5.0, 6.0, 7.0], np.float32), "noise": 1e-6},))
        end, steps = tr.get_retval()
        np.testing.assert_allclose([f(v) for v in steps], [8.0, 14.0, 21.0], atol=0.1)
        assert f(end) == pytest.approx(21.0, abs=0.1)

    def test_scan_length_inferred_zero_length_and_validation(self):          # :380-438
        @genjax.gen
        def walk_step(x, std):
            new_x = genjax.normal(x, std) @ "x"
            return new_x, new_x

        args = (0.0, np.array([2.0, 4.0, 3.0, 5.0, 1.0], np.float32))
        tr = walk_step.scan(n=5).simulate(genjax.key(314159), args)
        _, expected = tr.get_retval()
        np.testing.assert_allclose(tr.get_choices()[:, "x"].cpu().numpy(), [f(v) for v in expected], rtol=1e-6)
        tr2 = walk_step.scan().simulate(genjax.key(314159), args)
        np.testing.assert_array_equal(tr2.get_choices()[:, "x"].cpu().numpy(), tr.get_choices()[:, "x"].cpu().numpy())

        @genjax.gen
        def step(state, sigma):
            new_x = genjax.normal(state, sigma) @ "x"
            return (new_x, new_x + 1)

        empty = step.scan(n=0).simulate(genjax.key(1), (2.0, np.zeros(0, np.float32)))
        assert empty.get_choices().static_is_empty()
        step.scan().importance(genjax.key(2), empty.get_choices(), (2.0, np.zeros(0, np.float32)))

        @genjax.gen
        def foo(shift, d):
            x = genjax.normal(d["loc"], d["scale"]) @ "x"
            return x + shift, None

        with pytest.raises(ValueError, match="scan got values with different leading axis sizes: 1, 2."):
            foo.scan().simulate(genjax.key(3), (1.0, {"loc": np.array([10.0, 12.0], np.float32), "scale": np.array([1.0], np.float32)}))


class TestVmapRemaining:
    """reference tests/generative_functions/test_vmap_combinator.py:103-119, 230-243"""

    def test_nested_indexed_choice_map_importance(self):                     # :103-119
        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def model(x):
            z = genjax.normal(x, 1.0) @ "z"
            return z

        @genjax.vmap(in_axes=(0,))
        @genjax.gen
        def higher_model(x):
            return model(x) @ "outer"

        chm = C[0, "outer", 1, "z"].set(1.0)
        _, w = higher_model.importance(genjax.key(314159), chm, (np.ones((3, 3), np.float32),))
        assert f(w) == pytest.approx(f(genjax.normal.assess(C.v(1.0), (1.0, 1.0))[0]), abs=1e-6)

    def test_zero_length_vmap(self):                                         # :230-243
        @genjax.gen
        def step(state, sigma):
            new_x = genjax.normal(state, sigma) @ "x"
            return (new_x, new_x + 1)

        tr = step.vmap(in_axes=(None, 0)).simulate(genjax.key(20), (2.0, np.zeros(0, np.float32)))
        assert tr.get_choices().static_is_empty()
