"""The reference's combinator tests (tests/generative_functions/test_vmap_combinator.py, test_repeat_combinator.py,
test_scan_combinator.py) restated against genjax_amd, for the forms the site-program lowering supports (one index
level; mapped arguments are host data).  Line numbers of the originals are cited."""
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
