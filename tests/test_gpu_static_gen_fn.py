"""The reference's tests of StaticGenerativeFunction (tests/generative_functions/test_static_gen_fn.py), restated
against genjax_amd: simulate / assess / importance / update on @gen functions, nested calls, weight identities.
Same structure and assertions (line numbers cited); jax.jit wrappers dropped, jnp -> python/numpy scalars."""
import math

import numpy as np
import pytest

import genjax_amd as genjax
from genjax_amd import ChoiceMap, MissingAddress, Selection
from genjax_amd import ChoiceMapBuilder as C
from genjax_amd.inference import Regenerate, StaticRequest, Update

pytestmark = pytest.mark.gpu


def f(t):
    return float(t.detach().cpu()) if hasattr(t, "detach") else float(t)


def approx(x, rel=None):
    return pytest.approx(f(x), rel=rel, abs=1e-5)


class TestStaticGenFnSimulate:
    def test_simple_normal_simulate(self):                                   # :208-223
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_normal.simulate(sub_key, ())
        choice = tr.get_choices()
        (_, score1) = genjax.normal.importance(key, choice.get_submap("y1"), (0.0, 1.0))
        (_, score2) = genjax.normal.importance(key, choice.get_submap("y2"), (0.0, 1.0))
        assert f(tr.get_score()) == approx(f(score1) + f(score2), 0.01)
        assert f(tr.get_retval()) == approx(f(choice["y1"]) + f(choice["y2"]), 1e-6)

    def test_simple_normal_multiple_returns(self):                           # :225-244
        @genjax.gen
        def simple_normal_multiple_returns():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1, y2

        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_normal_multiple_returns.simulate(sub_key, ())
        y1_, y2_ = tr.get_choices()["y1"], tr.get_choices()["y2"]
        y1, y2 = tr.get_retval()
        assert f(y1) == f(y1_) and f(y2) == f(y2_)
        (score1, _) = genjax.normal.assess(C.v(y1), (0.0, 1.0))
        (score2, _) = genjax.normal.assess(C.v(y2), (0.0, 1.0))
        assert f(tr.get_score()) == approx(f(score1) + f(score2), 0.01)

    def test_hierarchical_simple_normal_multiple_returns(self):              # :246-271
        @genjax.gen
        def _submodel():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1, y2

        @genjax.gen
        def hierarchical_simple_normal_multiple_returns():
            y1, y2 = _submodel() @ "y1"
            return y1, y2

        key, sub_key = genjax.split(genjax.key(314159))
        tr = hierarchical_simple_normal_multiple_returns.simulate(sub_key, ())
        y1_, y2_ = tr.get_choices()["y1", "y1"], tr.get_choices()["y1", "y2"]
        y1, y2 = tr.get_retval()
        assert f(y1) == f(y1_) and f(y2) == f(y2_)
        (score1, _) = genjax.normal.assess(C.v(y1), (0.0, 1.0))
        (score2, _) = genjax.normal.assess(C.v(y2), (0.0, 1.0))
        assert f(tr.get_score()) == approx(f(score1) + f(score2), 0.01)


class TestStaticGenFnAssess:
    def test_simple_normal_assess(self):                                     # :287-300
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_normal.simulate(sub_key, ())
        (score, _retval) = simple_normal.assess(tr.get_choices(), ())
        assert f(score) == approx(tr.get_score(), 1e-6)

    def test_assess_missing_address(self):                                   # :302-320
        @genjax.gen
        def model():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        with pytest.raises(MissingAddress) as exc:
            _ = model.assess(C["y1"].set(1.0), ())
        assert exc.value.args == ("y2",)
        with pytest.raises(MissingAddress) as exc:
            _ = model.assess(C["y2"].set(1.0), ())
        assert exc.value.args == ("y1",)
        score, retval = model.assess(C["y1"].set(1.0).at["y2"].set(-1.0), ())
        assert f(score) == pytest.approx(-2.837877, rel=1e-5) and f(retval) == 0.0


class TestStaticGenFnImportance:
    def test_simple_normal_importance_with_args(self):                       # :383-397 (CustomTree -> two scalars)
        @genjax.gen
        def simple_normal(x, y):
            y1 = genjax.normal(x, 1.0) @ "y1"
            y2 = genjax.normal(y, 1.0) @ "y2"
            return y1 + y2

        key = genjax.key(314159)
        (tr, w) = simple_normal.importance(key, C["y1"].set(5.0), (3.0, 5.0))
        choice = tr.get_choices()
        (_, score1) = genjax.normal.importance(key, choice.get_submap("y1"), (3.0, 1.0))
        (_, score2) = genjax.normal.importance(key, choice.get_submap("y2"), (5.0, 1.0))
        assert f(tr.get_score()) == approx(f(score1) + f(score2), 0.01)
        assert f(w) == approx(score1, 0.01)

    def test_importance_weight_correctness(self):                            # :441-490
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        key = genjax.key(314159)
        # Full constraints.
        choice = C["y1"].set(0.5).at["y2"].set(0.5)
        (tr, w) = simple_normal.importance(key, choice, ())
        assert f(tr.get_choices()["y1"]) == 0.5 and f(tr.get_choices()["y2"]) == 0.5
        (_, score_1) = genjax.normal.importance(key, choice.get_submap("y1"), (0.0, 1.0))
        (_, score_2) = genjax.normal.importance(key, choice.get_submap("y2"), (0.0, 1.0))
        test_score = f(score_1) + f(score_2)
        assert f(tr.get_score()) == approx(test_score, 0.0001) and f(w) == approx(test_score, 0.0001)
        # Partial constraints.
        (tr, w) = simple_normal.importance(key, C["y2"].set(0.5), ())
        tr_chm = tr.get_choices()
        assert f(tr_chm["y2"]) == 0.5
        score_1, _ = genjax.normal.assess(tr_chm.get_submap("y1"), (0.0, 1.0))
        score_2, _ = genjax.normal.assess(tr_chm.get_submap("y2"), (0.0, 1.0))
        assert f(tr.get_score()) == approx(f(score_1) + f(score_2), 0.0001) and f(w) == approx(score_2, 0.0001)
        # No constraints.
        (tr, w) = simple_normal.importance(key, C.n(), ())
        tr_chm = tr.get_choices()
        score_1, _ = genjax.normal.assess(tr_chm.get_submap("y1"), (0.0, 1.0))
        score_2, _ = genjax.normal.assess(tr_chm.get_submap("y2"), (0.0, 1.0))
        assert f(tr.get_score()) == approx(f(score_1) + f(score_2), 0.0001) and f(w) == 0.0


class TestStaticGenFnUpdate:
    def test_simple_normal_update(self):                                     # :502-550
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_normal.simulate(sub_key, ())
        new = C["y1"].set(2.0)
        original_choice, original_score = tr.get_choices(), tr.get_score()
        key, sub_key = genjax.split(key)
        (updated, w, _, discard) = simple_normal.update(sub_key, tr, new, ())
        updated_choice = updated.get_choices()
        (_, score1) = genjax.normal.importance(key, updated_choice.get_submap("y1"), (0.0, 1.0))
        (_, score2) = genjax.normal.importance(key, updated_choice.get_submap("y2"), (0.0, 1.0))
        assert f(original_choice["y1",]) == f(discard["y1",])
        assert f(updated.get_score()) == approx(f(original_score) + f(w), 1e-5)
        assert f(updated.get_score()) == approx(f(score1) + f(score2), 0.01)
        new = C["y1"].set(2.0).at["y2"].set(3.0)
        key, sub_key = genjax.split(key)
        (updated, w, _, discard) = simple_normal.update(sub_key, tr, new, ())
        uc = updated.get_choices()
        (_, score1) = genjax.normal.importance(key, uc.get_submap("y1"), (0.0, 1.0))
        (_, score2) = genjax.normal.importance(key, uc.get_submap("y2"), (0.0, 1.0))
        assert f(updated.get_score()) == approx(f(original_score) + f(w), 1e-5)
        assert f(updated.get_score()) == approx(f(score1) + f(score2), 0.01)

    def test_simple_linked_normal_update(self):                              # :552-581
        @genjax.gen
        def simple_linked_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(y1, 1.0) @ "y2"
            y3 = genjax.normal(y1 + y2, 1.0) @ "y3"
            return y1 + y2 + y3

        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_linked_normal.simulate(sub_key, ())
        new = C["y1"].set(2.0)
        original_choice, original_score = tr.get_choices(), tr.get_score()
        key, sub_key = genjax.split(key)
        (updated, w, _, discard) = simple_linked_normal.update(sub_key, tr, new, ())
        uc = updated.get_choices()
        y1, y2, y3 = uc["y1"], uc["y2"], uc.get_submap("y3")
        score1, _ = genjax.normal.assess(C.v(y1), (0.0, 1.0))
        score2, _ = genjax.normal.assess(C.v(y2), (f(y1), 1.0))
        score3, _ = genjax.normal.assess(y3, (f(y1) + f(y2), 1.0))
        assert f(original_choice["y1"]) == f(discard["y1"])
        assert f(updated.get_score()) == approx(f(original_score) + f(w), 0.01)
        assert f(updated.get_score()) == approx(f(score1) + f(score2) + f(score3), 0.01)

    def test_simple_hierarchical_normal(self):                               # :583-620
        @genjax.gen
        def _inner(x):
            y1 = genjax.normal(x, 1.0) @ "y1"
            return y1

        @genjax.gen
        def simple_hierarchical_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = _inner(y1) @ "y2"
            y3 = _inner(y1 + y2) @ "y3"
            return y1 + y2 + y3

        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_hierarchical_normal.simulate(sub_key, ())
        new = C["y1"].set(2.0)
        original_choice, original_score = tr.get_choices(), tr.get_score()
        key, sub_key = genjax.split(key)
        (updated, w, _, discard) = simple_hierarchical_normal.update(sub_key, tr, new, ())
        uc = updated.get_choices()
        y1, y2, y3 = uc["y1"], uc["y2", "y1"], uc["y3", "y1"]
        assert f(y1) == f(new["y1"]) and f(y2) == f(original_choice["y2", "y1"]) and f(y3) == f(original_choice["y3", "y1"])
        score1, _ = genjax.normal.assess(C.v(y1), (0.0, 1.0))
        score2, _ = genjax.normal.assess(C.v(y2), (f(y1), 1.0))
        score3, _ = genjax.normal.assess(C.v(y3), (f(y1) + f(y2), 1.0))
        assert f(original_choice["y1"]) == f(discard["y1"])
        assert f(updated.get_score()) == approx(f(original_score) + f(w), 1e-4)
        assert f(updated.get_score()) == approx(f(score1) + f(score2) + f(score3), 0.01)

    @staticmethod
    def update_weight_correctness_general_assertions(simple_linked_normal):  # :622-667
        key, sub_key = genjax.split(genjax.key(314159))
        tr = simple_linked_normal.simulate(sub_key, ())
        old_y1, old_y2, old_y3 = (f(tr.get_choices()[a]) for a in ("y1", "y2", "y3"))
        new_y1 = 2.0
        new = C["y1"].set(new_y1)
        key, sub_key = genjax.split(key)
        (updated, w, _, _) = simple_linked_normal.update(sub_key, tr, new, ())
        (_, w_edit, _, _) = tr.edit(sub_key, Update(new))
        assert f(w_edit) == f(w)
        assert f(updated.get_choices()["y1"]) == new_y1
        lp = lambda v, m: f(genjax.normal.assess(C.v(v), (m, 1.0))[0])
        d_y3 = lp(old_y3, new_y1 + old_y2) - lp(old_y3, old_y1 + old_y2)
        d_y2 = lp(old_y2, new_y1) - lp(old_y2, old_y1)
        d_y1 = lp(new_y1, 0.0) - lp(old_y1, 0.0)
        assert f(w) == pytest.approx(d_y3 + d_y2 + d_y1, rel=1e-4, abs=1e-4)
        # composition of update calls
        new_y3 = 2.0
        key, sub_key = genjax.split(key)
        (updated, w, _, _) = simple_linked_normal.update(sub_key, updated, C["y3"].set(new_y3), ())
        assert f(updated.get_choices()["y3"]) == 2.0
        correct_w = lp(new_y3, new_y1 + old_y2) - lp(old_y3, new_y1 + old_y2)
        assert f(w) == pytest.approx(correct_w, rel=1e-4, abs=1e-4)

    def test_update_weight_correctness(self):                                # :669-731
        @genjax.gen
        def simple_linked_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(y1, 1.0) @ "y2"
            y3 = genjax.normal(y1 + y2, 1.0) @ "y3"
            return y1 + y2 + y3

        self.update_weight_correctness_general_assertions(simple_linked_normal)

        @genjax.gen
        def curried_linked_normal(v1, v2, v3):
            y1 = genjax.normal(0.0, v1) @ "y1"
            y2 = genjax.normal(y1, v2) @ "y2"
            y3 = genjax.normal(y1 + y2, v3) @ "y3"
            return y1 + y2 + y3

        self.update_weight_correctness_general_assertions(curried_linked_normal.partial_apply(1.0, 1.0, 1.0))
        self.update_weight_correctness_general_assertions(curried_linked_normal.partial_apply(1.0).partial_apply(1.0, 1.0))


class TestStaticGenFnAddresses:
    def test_simple_normal_addr_dup(self):                                   # :778-788
        @genjax.gen
        def simple_normal_addr_dup():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y1"
            return y1 + y2

        with pytest.raises(genjax.AddressReuse) as exc_info:
            _ = simple_normal_addr_dup.simulate(genjax.key(314159), ())
        assert exc_info.value.args == ("y1",)

    def test_static_edit_request_hierarchical(self):                         # :934-961 shape: requests addressed through callees
        @genjax.gen
        def inner():
            a = genjax.normal(0.0, 1.0) @ "a"
            b = genjax.normal(a, 1.0) @ "b"
            return b

        @genjax.gen
        def outer():
            x = inner() @ "x"
            y = genjax.normal(x, 1.0) @ "y"
            return y

        key, sub_key = genjax.split(genjax.key(1))
        tr = outer.simulate(sub_key, ())
        req = StaticRequest({"x": StaticRequest({"a": Update(C.choice(0.25))}), "y": Regenerate(Selection.all())})
        key, sub_key = genjax.split(key)
        new_tr, w, _, bwd = req.edit(sub_key, tr, ())
        old, new = tr.get_choices(), new_tr.get_choices()
        assert f(new["x", "a"]) == 0.25 and f(new["x", "b"]) == f(old["x", "b"]) and f(new["y"]) != f(old["y"])
        lp = lambda v, m: -0.5 * (v - m) ** 2 - 0.5 * math.log(2 * math.pi)
        # Update part: new score - old score of the touched densities; a regenerated leaf contributes its own
        # score change (distribution.py:258-272: incremental_w = w - trace.get_score())
        want = (lp(0.25, 0.0) + lp(f(old["x", "b"]), 0.25)) - (lp(f(old["x", "a"]), 0.0) + lp(f(old["x", "b"]), f(old["x", "a"])))
        want += lp(f(new["y"]), f(old["x", "b"])) - lp(f(old["y"]), f(old["x", "b"]))
        assert f(w) == pytest.approx(want, rel=1e-4, abs=1e-4)
        back, w2, _, _ = bwd.edit(genjax.key(7), new_tr, ())
        assert f(back.get_choices()["x", "a"]) == pytest.approx(f(old["x", "a"]))


class TestDistributions:
    """tests/generative_functions/test_distributions.py:22-190 (per-site semantics of distribution.py:117-244);
    masks with host booleans only."""

    def test_simulate(self):                                                 # :23-26
        tr = genjax.normal(0.0, 1.0).simulate(genjax.key(314159), ())
        assert f(tr.get_score()) == f(genjax.normal(0.0, 1.0).assess(tr.get_choices(), ())[0])

    def test_importance(self):                                               # :28-58
        key = genjax.key(314159)
        (tr, w) = genjax.normal.importance(key, C.n(), (0.0, 1.0))
        assert f(w) == 0.0
        (tr, w) = genjax.normal.importance(key, C.v(1.0), (0.0, 1.0))
        assert f(w) == f(genjax.normal(0.0, 1.0).assess(tr.get_choices(), ())[0])
        (tr, w) = genjax.normal.importance(key, C.v(1.0).mask(np.array(True)), (0.0, 1.0))
        v = f(tr.get_choices().get_value())
        assert v == 1.0 and f(w) == f(genjax.normal.assess(C.v(v), (0.0, 1.0))[0])
        (tr, w) = genjax.normal.importance(key, C.v(1.0).mask(np.array(False)), (0.0, 1.0))
        assert f(tr.get_choices().get_value()) != 1.0 and f(w) == 0.0

    def test_update(self):                                                   # :60-190
        from genjax_amd import Diff, NoChange, UnknownChange
        key, sub_key = genjax.split(genjax.key(314159))
        tr = genjax.normal.simulate(sub_key, (0.0, 1.0))
        old = tr.get_choices()
        a = lambda chm, args: f(genjax.normal.assess(chm, args)[0])
        same = (Diff(0.0, NoChange), Diff(1.0, NoChange))
        # no constraint, no change to arguments
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.n(), same)
        assert f(new_tr.get_choices().get_value()) == f(old.get_value())
        assert f(new_tr.get_score()) == pytest.approx(a(old, (0.0, 1.0)), rel=1e-6) and f(w) == 0.0
        # constraint, no change to arguments
        (new_tr, w, _, discard) = genjax.normal.update(sub_key, tr, C.v(1.0), same)
        assert f(new_tr.get_choices().get_value()) == 1.0 and f(discard.get_value()) == f(old.get_value())
        assert f(new_tr.get_score()) == pytest.approx(a(C.v(1.0), (0.0, 1.0)), rel=1e-6)
        assert f(w) == pytest.approx(a(C.v(1.0), (0.0, 1.0)) - a(old, (0.0, 1.0)), rel=1e-5, abs=1e-6)
        # no constraint, change to arguments
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.n(), (Diff(1.0, UnknownChange), Diff(1.0, NoChange)))
        assert f(new_tr.get_choices().get_value()) == f(old.get_value())
        assert f(new_tr.get_score()) == pytest.approx(a(old, (1.0, 1.0)), rel=1e-6)
        assert f(w) == pytest.approx(a(old, (1.0, 1.0)) - a(old, (0.0, 1.0)), rel=1e-5, abs=1e-6)
        # constraint, change to arguments
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.v(1.0), (Diff(1.0, UnknownChange), Diff(2.0, UnknownChange)))
        assert f(new_tr.get_choices().get_value()) == 1.0
        assert f(new_tr.get_score()) == pytest.approx(a(C.v(1.0), (1.0, 2.0)), rel=1e-6)
        assert f(w) == pytest.approx(a(C.v(1.0), (1.0, 2.0)) - a(old, (0.0, 1.0)), rel=1e-5, abs=1e-6)
        # masked constraints (True / False), with and without argument changes
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.v(1.0).mask(np.array(True)), same)
        assert f(new_tr.get_choices().get_value()) == 1.0
        assert f(w) == pytest.approx(a(C.v(1.0), (0.0, 1.0)) - a(old, (0.0, 1.0)), rel=1e-5, abs=1e-6)
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.v(1.0).mask(True), (Diff(1.0, UnknownChange), Diff(1.0, NoChange)))
        assert f(new_tr.get_choices().get_value()) == 1.0
        assert f(w) == pytest.approx(a(C.v(1.0), (1.0, 1.0)) - a(old, (0.0, 1.0)), rel=1e-5, abs=1e-6)
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.v(1.0).mask(False), same)
        assert f(new_tr.get_choices().get_value()) == f(old.get_value()) and f(w) == 0.0
        (new_tr, w, _, _) = genjax.normal.update(sub_key, tr, C.v(1.0).mask(False), (Diff(1.0, UnknownChange), Diff(1.0, NoChange)))
        assert f(new_tr.get_choices().get_value()) == f(old.get_value())
        assert f(w) == pytest.approx(a(old, (1.0, 1.0)) - a(old, (0.0, 1.0)), rel=1e-5, abs=1e-6)

    def test_update_with_changed_model_arguments(self):
        """static.py:827-865 with argdiffs: an @gen function re-scored under new arguments, choices kept."""
        from genjax_amd import Diff

        @genjax.gen
        def model(mu, sd):
            x = genjax.normal(mu, sd) @ "x"
            y = genjax.normal(x, 0.5) @ "y"
            return y

        tr = model.simulate(genjax.key(2), (0.0, 1.0))
        ch = tr.get_choices()
        new_tr, w, _, _ = model.update(genjax.key(3), tr, C.n(), Diff.unknown_change((1.0, 2.0)))
        assert f(new_tr.get_choices()["x"]) == f(ch["x"]) and new_tr.get_args() == (1.0, 2.0)
        lpn = lambda v, m, s: -0.5 * ((v - m) / s) ** 2 - math.log(s) - 0.5 * math.log(2 * math.pi)
        assert f(w) == pytest.approx(lpn(f(ch["x"]), 1.0, 2.0) - lpn(f(ch["x"]), 0.0, 1.0), rel=1e-4, abs=1e-5)


class TestNoChoicesForwardRefClosure:
    """reference test_static_gen_fn.py:197-206, 274-285 (no choices), :803-821 (forward reference), :825-885 (closures)"""

    def test_simulate_and_assess_with_no_choices(self):
        @genjax.gen
        def empty(x):
            return (x - 3.0) * (x - 3.0)

        tr = empty.simulate(genjax.key(314159), (1.0,))
        assert f(tr.get_score()) == 0.0 and f(tr.get_retval()) == 4.0
        score, _ = empty.assess(tr.get_choices(), (1.0,))
        assert f(score) == f(tr.get_score())

    def test_forward_ref(self):
        def make_gen_fn():
            @genjax.gen
            def proposal(x):
                x = outlier(x) @ "x"
                return x

            @genjax.gen
            def outlier(prob):
                is_outlier = genjax.bernoulli(probs=prob) @ "is_outlier"
                return is_outlier

            return proposal

        tr = make_gen_fn().simulate(genjax.key(314159), (0.3,))
        want = math.log(0.3) if bool(tr.get_retval()) else math.log(0.7)
        assert f(tr.get_score()) == pytest.approx(want, abs=1e-6)

    def test_gen_fn_closure(self):
        @genjax.gen
        def model():
            return genjax.normal(1.0, 0.001) @ "x"

        gfc = model()
        tr = gfc.simulate(genjax.key(0), ())
        lp = lambda v: f(genjax.normal.assess(C.v(v), (1.0, 0.001))[0])
        assert f(tr.get_score()) == pytest.approx(lp(f(tr.get_retval())), rel=1e-5, abs=1e-4)
        tr_u, w = gfc.importance(genjax.key(1), C.kw(x=1.1), ())
        assert f(tr_u.get_score()) == pytest.approx(lp(1.1), rel=1e-5) and f(w) == f(tr_u.get_score())

    def test_gen_fn_closure_with_kwargs(self):
        @genjax.gen
        def model(x, y, z=None):
            if z is None:
                raise ValueError("z must be provided")
            _sampled = genjax.normal(x + y, z) @ "sampled"
            return z

        key = genjax.key(0)
        with pytest.raises(ValueError, match="z must be provided"):
            model(1.0, 2.0)(key)
        gfc = model(1.0, 2.0, z=3.0)
        assert f(gfc(key)) == 3.0 and f(gfc(key, z=10.0)) == 10.0
        arg_tuple = (1.0, 2.0, 3.0)
        assert gfc.simulate(key, ()).get_choices() == model.simulate(key, arg_tuple).get_choices()
        chm = C.kw(sampled=3.5)
        assert f(gfc.assess(chm, ())[0]) == f(model.assess(chm, arg_tuple)[0])
        assert f(gfc.importance(key, C.kw(sampled=3.0), ())[1]) == f(model.importance(key, C.kw(sampled=3.0), arg_tuple)[1])


class TestInlineMethodsPartialApply:
    """reference test_static_gen_fn.py:987-1114 (inline), :1116-1145 (gen on a method), :1147-1165 (partial_apply)"""

    @staticmethod
    def _models():
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        @genjax.gen
        def higher_model():
            return simple_normal.inline()

        @genjax.gen
        def higher_higher_model():
            return higher_model.inline()

        return higher_model, higher_higher_model

    def test_inline_simulate_importance_update_assess(self):
        lp = lambda v: f(genjax.normal.assess(C.v(v), (0.0, 1.0))[0])
        for m in self._models():
            tr = m.simulate(genjax.key(314159), ())
            ch = tr.get_choices()
            assert "y1" in ch and "y2" in ch
            tr_i, w = m.importance(genjax.key(1), C["y1"].set(3.0), ())
            assert f(w) == pytest.approx(lp(3.0), rel=1e-6)
            old = f(ch["y1"])
            tr_u, w_u, _, _ = m.update(genjax.key(2), tr, C["y1"].set(3.0), ())
            assert f(w_u) == pytest.approx(lp(3.0) - lp(old), rel=1e-4, abs=1e-5)
            score, _ = m.assess(C["y1"].set(3.0).at["y2"].set(3.0), ())
            assert f(score) == pytest.approx(2 * lp(3.0), rel=1e-6)

    def test_gen_method(self):
        class Model:
            def __init__(self, foo, bar):
                self.foo, self.bar = foo, bar

            @genjax.gen
            def run(self, x):
                y = genjax.normal(self.foo, self.bar) @ "y"
                z = genjax.normal(x, 1.0) @ "z"
                return y + z

        m = Model(4.0, 6.0)
        tr = m.run.simulate(genjax.key(0), (1.0,))
        chm = tr.get_choices()
        assert tr.get_args() == (1.0,)                     # the curried `self` is not among the arguments
        assert tr.gen_fn.partial_args[0] is m
        assert "y" in chm and "z" in chm and "q" not in chm

    def test_partial_apply(self):
        @genjax.gen
        def model(x, y, z):
            return genjax.normal(x, y + z) @ "x"

        double_curry = model.partial_apply(1.0).partial_apply(1.0)
        tr = double_curry.simulate(genjax.key(0), (2.0,))
        assert tr.get_args() == (2.0,) and tr.gen_fn.partial_args == (1.0, 1.0)
        assert f(tr.get_score()) == pytest.approx(f(genjax.normal.assess(C.v(tr.get_retval()), (1.0, 3.0))[0]), rel=1e-5)


class TestStaticRetvalAndEditRequests:
    """reference test_static_gen_fn.py:139-151 (literal return value), :401-415 (assess == score), :889-932 (StaticRequest
    composition, also at a tupled address)"""

    def test_static_retval(self):
        @genjax.gen
        def fn():
            return 1

        tr = fn.simulate(genjax.key(0), ())
        tr.update(genjax.key(0), C.n(), ())
        assert tr.get_retval() == 1

    def test_assess_of_own_choices_is_the_score(self):
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ "y1"
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        tr = simple_normal.simulate(genjax.key(314159), ())
        score, _ = simple_normal.assess(tr.get_choices(), ())
        assert f(score) == pytest.approx(f(tr.get_score()), rel=1e-6)

    @pytest.mark.parametrize("first", ["y1", ("y1", "y3")])
    def test_static_edit_request_composition(self, first):
        @genjax.gen
        def simple_normal():
            y1 = genjax.normal(0.0, 1.0) @ first
            y2 = genjax.normal(0.0, 1.0) @ "y2"
            return y1 + y2

        tr = simple_normal.simulate(genjax.key(0), ())
        request = StaticRequest({first: Regenerate(Selection.all()), "y2": Update(C.v(3.0))})
        new_tr, w, _, bwd_request = request.edit(genjax.key(1), tr, ())
        assert f(new_tr.get_choices()["y2"]) == 3.0 and f(w) != 0.0
        old_tr, w_, _, _ = bwd_request.edit(genjax.key(2), new_tr, ())
        assert f(old_tr.get_choices()["y2"]) == f(tr.get_choices()["y2"]) and f(w_) != 0.0
        assert f(w) + f(w_) == pytest.approx(0.0, abs=1e-5)
