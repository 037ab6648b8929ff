"""Edit requests and conditional SMC on the device against the oracle's restatement of the reference's weight algebra
(oracle/edits.py: distribution.py:179-300, rejuvenate.py:70-94, smc.py:317-351) — weights AND values, not
self-consistency between two device paths."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

RT, AT = 3e-4, 3e-4
NEAR_TIE = 3e-4


def _np(t):
    return t.detach().cpu().numpy()


def _model():
    import genjax_amd as genjax

    @genjax.gen
    def model(s):
        x = genjax.normal(0.0, 1.0) @ "x"
        k = genjax.flip(0.3) @ "k"
        y = genjax.normal(x, s) @ "y"
        z = genjax.gamma(2.0, genjax.exp(0.3 * x)) @ "z"
        return y

    return model


def _assess_prog(gf, args):
    """every site constrained per particle (values come from the rows)"""
    from genjax_amd import ChoiceMap
    sl, _ = gf.site_list(args)
    prog, _, _ = gf.pack(args, ChoiceMap.empty(), False, per_particle=tuple(s.addr for s in sl.sites))
    return prog


def test_update_weights_and_discard_against_oracle():
    import torch
    import genjax_amd as genjax
    from genjax_amd import C, Update
    from oracle import edits
    model = _model()
    K = 4096
    tr = model.simulate(genjax.key(3), (0.5,), K)
    rs = np.random.default_rng(1)
    new_x = rs.standard_normal(K).astype(np.float32)
    new_tr, w, _, bwd = Update(C["x"].set(torch.as_tensor(new_x).cuda())).edit(genjax.key(4), tr, None)
    old = _np(tr.choices)
    new = old.copy()
    new[tr.prog.slot_of["x"]] = new_x
    prog = _assess_prog(model, (0.5,))
    w_o, ss_o, score_o = edits.update(prog, old, prog, new, K)
    np.testing.assert_allclose(_np(w), w_o, rtol=RT, atol=AT)
    np.testing.assert_allclose(_np(new_tr.score), score_o, rtol=RT, atol=AT)
    np.testing.assert_array_equal(_np(new_tr.choices), new)                       # only the constrained address moved
    np.testing.assert_array_equal(_np(bwd.constraint["x"]), old[tr.prog.slot_of["x"]])   # discard = old value (distribution.py:241)
    # new ARGUMENTS, no new values (":case None", distribution.py:226-235): every site re-assessed under s = 0.9
    new_tr2, w2, _, bwd2 = Update().edit(genjax.key(5), tr, (0.9,))
    w2_o, _, score2_o = edits.update(prog, old, _assess_prog(model, (0.9,)), old, K)
    np.testing.assert_allclose(_np(w2), w2_o, rtol=RT, atol=AT)
    np.testing.assert_allclose(_np(new_tr2.score), score2_o, rtol=RT, atol=AT)
    assert len(bwd2.constraint) == 0


def test_regenerate_values_and_weights_against_oracle():
    import genjax_amd as genjax
    from genjax_amd import ChoiceMap, Regenerate, S
    from oracle import edits
    model = _model()
    K = 4096
    tr = model.simulate(genjax.key(3), (0.5,), K)
    old = _np(tr.choices)
    for sel, addrs in ((S["x"], ("x",)), (S["z"] | S["k"], ("z", "k"))):
        key = genjax.key(11)
        new_tr, w, _, bwd = Regenerate(sel).edit(key, tr, None)
        rest = tuple(a for a in ("x", "k", "y", "z") if a not in addrs)
        prog_regen, _, _ = model.pack((0.5,), ChoiceMap.empty(), True, per_particle=rest)
        w_o, ch_o, _, margin = edits.regenerate(_assess_prog(model, (0.5,)), old, prog_regen, key, K)
        got = _np(new_tr.choices)
        ok = (np.abs(got - ch_o) <= AT + RT * np.abs(ch_o)).all(axis=0)
        assert (margin[~ok] < NEAR_TIE).all() and (~ok).mean() < 0.01              # gamma accept / flip near ties only
        np.testing.assert_allclose(_np(w)[ok], w_o[ok], rtol=RT, atol=AT)
        for a in rest:                                                            # unselected values untouched
            np.testing.assert_array_equal(got[tr.prog.slot_of[a]], old[tr.prog.slot_of[a]])
        for a in addrs:                                                           # backward request restores the old values
            np.testing.assert_array_equal(_np(bwd.constraint[a]), old[tr.prog.slot_of[a]])


def test_rejuvenate_against_oracle():
    import genjax_amd as genjax
    from genjax_amd import ChoiceMap, Rejuvenate, StaticRequest
    from genjax_amd.core import fold_in, split
    from oracle import cpu, edits
    model = _model()
    K = 2048
    tr = model.simulate(genjax.key(3), (0.5,), K)
    old = _np(tr.choices)
    key = genjax.key(21)
    req = StaticRequest({"x": Rejuvenate(genjax.normal, lambda chm: (chm.get_value(), 0.25))})
    new_tr, w, _, _ = req.edit(key, tr, None)

    # the reference's steps (rejuvenate.py:76-88) with the oracle's arithmetic
    @genjax.gen
    def prop():
        cur = genjax.normal(0.0, 1.0) @ "cur"
        genjax.normal(cur, 0.25) @ "new"

    k = fold_in(key, 1)                      # StaticRequest: request n gets fold_in(key, n + 1)
    k, sub_key = split(k)
    xs = tr.prog.slot_of["x"]
    p_fwd, _, _ = prop.pack((), ChoiceMap.empty(), True, per_particle=("cur",))
    ch = np.zeros((2, K), np.float32)
    ch[p_fwd.slot_of["cur"]] = old[xs]
    fwd = cpu.run_program(p_fwd, sub_key, K, choices=ch, want_site_scores=True)
    z_new = fwd["choices"][p_fwd.slot_of["new"]]
    fwd_score = fwd["site_scores"][1]
    new = old.copy()
    new[xs] = z_new
    prog = _assess_prog(model, (0.5,))
    w_upd, _, score_o = edits.update(prog, old, prog, new, K)
    p_bwd, _, _ = prop.pack((), ChoiceMap.empty(), False, per_particle=("cur", "new"))
    ch2 = np.zeros((2, K), np.float32)
    ch2[p_bwd.slot_of["cur"]] = z_new
    ch2[p_bwd.slot_of["new"]] = old[xs]
    bwd_score = cpu.run_program(p_bwd, (0, 0), K, choices=ch2, want_site_scores=True)["site_scores"][1]
    w_o = edits.rejuvenate(w_upd, fwd_score, bwd_score)
    np.testing.assert_allclose(_np(new_tr.choices)[xs], z_new, rtol=RT, atol=AT)
    np.testing.assert_allclose(_np(w), w_o, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(_np(new_tr.score), score_o, rtol=1e-3, atol=1e-3)


def test_rejuvenate_a_vector_valued_address_against_oracle():
    """Rejuvenate over an arbitrary proposal (rejuvenate.py:70-94) at a VECTOR-valued address: a random-walk proposal
    mv_normal_diag(current, 0.3) on a 3-vector — proposal draw, Update weight, reverse proposal score, all against the oracle"""
    import genjax_amd as genjax
    from genjax_amd import ChoiceMap, Rejuvenate, StaticRequest
    from genjax_amd.core import fold_in, split
    from oracle import cpu, edits
    d, K = 3, 2048
    sc = np.array([0.3, 0.2, 0.4], np.float32)

    @genjax.gen
    def model():
        v = genjax.mv_normal_diag(np.zeros(d, np.float32), np.ones(d, np.float32)) @ "v"
        genjax.mv_normal_diag(v, np.full(d, 0.5, np.float32)) @ "y"
        genjax.normal(v[1], 1.0) @ "t"

    tr = model.simulate(genjax.key(5), (), K)
    old = _np(tr.choices)
    key = genjax.key(22)
    req = StaticRequest({"v": Rejuvenate(genjax.mv_normal_diag, lambda chm: (chm.get_value(), sc))})
    new_tr, w, _, _ = req.edit(key, tr, None)

    @genjax.gen
    def prop():
        cur = genjax.mv_normal_diag(np.zeros(d, np.float32), np.ones(d, np.float32)) @ "cur"
        genjax.mv_normal_diag(cur, sc) @ "new"

    k = fold_in(key, 1)
    k, sub_key = split(k)
    vs = tr.prog.slot_of["v"]
    p_fwd, _, _ = prop.pack((), ChoiceMap.empty(), True, per_particle=("cur",))
    ch = np.zeros((2 * d, K), np.float32)
    ch[p_fwd.slot_of["cur"]:p_fwd.slot_of["cur"] + d] = old[vs:vs + d]
    fwd = cpu.run_program(p_fwd, sub_key, K, choices=ch, want_site_scores=True)
    z_new = fwd["choices"][p_fwd.slot_of["new"]:p_fwd.slot_of["new"] + d]
    new = old.copy()
    new[vs:vs + d] = z_new
    prog = _assess_prog(model, ())
    w_upd, _, score_o = edits.update(prog, old, prog, new, K)
    p_bwd, _, _ = prop.pack((), ChoiceMap.empty(), False, per_particle=("cur", "new"))
    ch2 = np.zeros((2 * d, K), np.float32)
    ch2[p_bwd.slot_of["cur"]:p_bwd.slot_of["cur"] + d] = z_new
    ch2[p_bwd.slot_of["new"]:p_bwd.slot_of["new"] + d] = old[vs:vs + d]
    bwd_score = cpu.run_program(p_bwd, (0, 0), K, choices=ch2, want_site_scores=True)["site_scores"][1]
    w_o = edits.rejuvenate(w_upd, fwd["site_scores"][1], bwd_score)
    np.testing.assert_allclose(_np(new_tr.choices)[vs:vs + d], z_new, rtol=RT, atol=AT)
    np.testing.assert_allclose(_np(w), w_o, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(_np(new_tr.score), score_o, rtol=1e-3, atol=1e-3)
    assert np.abs(_np(new_tr.choices)[vs:vs + d] - old[vs:vs + d]).max() > 0.1


def test_run_csmc_weights_and_values_against_oracle():
    import genjax_amd as genjax
    from genjax_amd import C, ChoiceMap, ImportanceK, Target
    from genjax_amd.core import split
    from oracle import edits
    model = _model()
    K = 1000
    target = Target(model, (0.5,), C["y"].set(0.7))
    retained = C["x"].set(0.3) | C["k"].set(1.0) | C["z"].set(1.7)
    key = genjax.key(31)
    pc = ImportanceK(target, k_particles=K).run_csmc(key, retained)
    k2, sub_key = split(key)
    prog_f, _, _ = model.pack((0.5,), target.constraint, True)
    prog_r, _, _ = model.pack((0.5,), target.constraint, True, per_particle=("x", "k", "z"))
    rows = np.zeros((prog_r.n_slots, 1), np.float32)
    for a, v in (("x", 0.3), ("k", 1.0), ("z", 1.7)):
        rows[prog_r.slot_of[a]] = v
    lw_o, ch_o = edits.run_csmc(prog_f, prog_r, k2, sub_key, K, rows)
    got_lw, got_ch = _np(pc.get_log_weights()), _np(pc.get_particles().choices)
    assert got_lw.shape == (K,)
    # the retained particle is stacked LAST with its importance weight (smc.py:337-349)
    np.testing.assert_allclose(got_ch[:, -1], ch_o[:, -1], rtol=1e-6)
    np.testing.assert_allclose(got_lw[-1], lw_o[-1], rtol=RT, atol=AT)
    ok = (np.abs(got_ch - ch_o) <= AT + RT * np.abs(ch_o)).all(axis=0)
    assert ok.mean() > 0.99                                                        # gamma / flip near ties
    np.testing.assert_allclose(got_lw[ok], lw_o[ok], rtol=RT, atol=AT)
