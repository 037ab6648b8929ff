"""CPU restatement of the reference's MCMC edit requests and conditional SMC — TEST INFRASTRUCTURE (parity oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(genjax_amd/) never does.  Everything here is the reference's weight algebra, site by site, on top of the C oracle's
densities and samplers (oracle/gjx_oracle.c); citations are file:line under /root/reference/src/genjax/_src/.

    Update      generative_functions/distributions/distribution.py:179-244 (per site), static.py:443-466 (sum)
    Regenerate  distribution.py:258-300 (per site), static.py:906-946
    Rejuvenate  inference/requests/rejuvenate.py:70-94
    run_csmc    inference/smc.py:317-351

A "program" is a genjax_amd.program.PackedProgram (the ABI-level site list; the oracle shares the struct layout, not
code).  Values are SoA numpy arrays [n_slots][K].  Parity with the reference's own sample streams is unpinned (the
reference cannot run here), as for the rest of the oracle.
"""
from __future__ import annotations

import numpy as np

from . import cpu


def site_scores(prog_all_constrained, choices, K):
    """per-site log-densities of the given values: what DistributionTrace.get_score() holds (distribution.py:59-82)"""
    out = cpu.run_program(prog_all_constrained, (0, 0), K, choices=choices, want_site_scores=True)
    return out["site_scores"][: prog_all_constrained.n_sites].copy(), out["score"].copy()


def update(prog_old, old_choices, prog_new, new_choices, K):
    """Update.edit on a static trace.  Every site is revisited (static.py:443-466); at a primitive site
    (distribution.py:179-244) the forward term is the log-density of the NEW value if the constraint has one
    (":case v", 237-244) or of the OLD value otherwise (":case None", 226-235), under the (possibly new) arguments; the
    backward term is the old trace's score; w = fwd - bwd; the handler adds the site weights up (static.py:463).
    prog_old / prog_new: every site constrained (OBS_SLOT / OBS_TAB) to old_choices / new_choices.
    -> (weight f32[K], new per-site scores, new score)"""
    old_ss, _ = site_scores(prog_old, old_choices, K)
    new_ss, new_score = site_scores(prog_new, new_choices, K)
    w = np.zeros(K, np.float32)
    for j in range(new_ss.shape[0]):                      # weight += w_j in site order, float32 as the handler does
        w = (w + (new_ss[j] - old_ss[j]).astype(np.float32)).astype(np.float32)
    return w, new_ss, new_score


def regenerate(prog_old, old_choices, prog_regen, key, K):
    """Regenerate(selection).edit.  Selected sites draw a fresh value from their distribution under the current
    arguments and contribute new_score - old_score (distribution.py:264-276); the others are re-assessed when their
    arguments changed (new_score - old_score, 286-298) and contribute 0 otherwise (278-285 — which is also
    new - old).  prog_regen: selected sites in SAMPLE mode, the rest constrained per particle; same key and particle
    indices as the device run.  -> (weight, new choices, new per-site scores, margin)"""
    old_ss, _ = site_scores(prog_old, old_choices, K)
    out = cpu.run_program(prog_regen, key, K, choices=old_choices.copy(), want_site_scores=True, want_margin=True)
    new_ss = out["site_scores"][: prog_regen.n_sites]
    w = np.zeros(K, np.float32)
    for j in range(new_ss.shape[0]):
        w = (w + (new_ss[j] - old_ss[j]).astype(np.float32)).astype(np.float32)
    return w, out["choices"], new_ss, out["margin"]


def rejuvenate(update_weight, fwd_proposal_score, bwd_proposal_score):
    """Rejuvenate.edit (rejuvenate.py:76-88): propose z' ~ q(. | args(z)) with score fwd; apply Update(z') (weight w);
    score the reverse proposal q(z | args(z')) = bwd; final_weight = w + bwd - fwd."""
    return (update_weight + bwd_proposal_score - fwd_proposal_score).astype(np.float32)


def run_csmc(prog_fresh, prog_retained, key, sub_key, K, retained_rows, log_q=None, retained_q_score=None):
    """ImportanceK.run_csmc (smc.py:317-351): K - 1 fresh particles from target.importance (keys split from sub_key:
    particle i <-> stream i of sub_key) plus the retained choice map run through target.importance with `key`, stacked
    LAST (stack_to_first_dim, smc.py:56-68); log-weights = target weights - proposal scores (0 without q, 343).
    prog_fresh: the target's program (constraint observed, rest sampled); prog_retained: the same with the retained
    addresses constrained per particle; retained_rows: [n_slots][1] values.  -> (log_weights f32[K], choices [n][K])"""
    n = max(prog_fresh.n_slots, 1)
    ret = cpu.run_program(prog_retained, key, 1, choices=np.asarray(retained_rows, np.float32).reshape(n, 1))
    if K == 1:
        lw, ch = ret["weight"], ret["choices"]
    else:
        fr = cpu.run_program(prog_fresh, sub_key, K - 1)
        lw = np.concatenate([fr["weight"], ret["weight"]]).astype(np.float32)
        ch = np.concatenate([fr["choices"], ret["choices"]], axis=1)
    if log_q is not None:
        lw = (lw - np.concatenate([np.asarray(log_q, np.float32), np.asarray(retained_q_score, np.float32).reshape(1)])).astype(np.float32)
    return lw, ch
