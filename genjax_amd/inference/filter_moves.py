"""Resample-move with ARBITRARY moves in the generic particle filter — the step-by-step form.

The reference composes a rejuvenation move out of edit requests: ``Rejuvenate`` takes any proposal generative function and returns
the Metropolis-Hastings ratio as the move's weight (inference/requests/rejuvenate.py:70-94), ``HMC`` is an EditRequest like any
other (inference/requests/hmc.py:138-211), and the caller applies the accept rule (tests/inference/test_requests.py:131-137).  The
one-launch filter kernel carries ONE move, generated from the step program: random-walk Metropolis on the carry
(``rejuvenate=dict(n_moves=, scale=)``, include/gjx.h gjx_filter_opts::n_moves).  This module is the general form: the filter as a
loop of device calls per step — resample, gather, the moves, propagate — in which a move is one of the library's own requests
applied to the STEP-LOCAL target of the gathered carry,

    pi_{t-1}(x) = p(x | the ancestor's own inputs) * p(y_{t-1} | x)        (the step program t-1 in assess form),

which the filter's particles are distributed as at that point, so any pi-invariant kernel leaves the filter proper:

  * ``HMC(selection, eps, L)`` — gjx_hmc over the selected latents of the step with the accept rule fused (one launch for all K
    particles: the generated HMC kernel where the emitter covers the step program, hmc.py:138-211 + test_requests.py:134-137);
  * ``{addr: Rejuvenate(dist, argument_mapping)}`` (or the ``StaticRequest`` of it) — forward draw from the proposal at the current
    value, model ratio by two assess runs of the step program, backward score, accept with log u < w + bwd - fwd
    (rejuvenate.py:70-94).

Everything that computes is a call of the HIP library (gjx_run_program_ex, gjx_hmc, gjx_gather_rows, the tile-scaled resamplers,
gjx_mh_accept: the reference's caller-side accept as a device call); what the host adds are the row copies between the calls and the
four-term sum of the acceptance ratio.  Without moves the loop reproduces gjx_scan_filter bit
for bit (same step keys, comb offsets / resampling keys, resamplers): tests/test_gpu_filter_moves.py holds it to that, and holds every
move step-locally to the same composition over the oracle.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _abi as A
from ..core import ChoiceMap, Key, fold_in, split
from ..program import PackedProgram

MOVE_KEY_TAG = 0x6D6F7665            # "move": the key of move m of step t = fold_in(fold_in(step key t, MOVE_KEY_TAG), m)


def move_key(step_key: Key, m: int) -> Key:
    return fold_in(fold_in(step_key, MOVE_KEY_TAG), m)


def normalise_moves(moves) -> list:
    """-> [("hmc", request) | ("proposal", addr, Rejuvenate)]"""
    from .requests import HMC, Rejuvenate, StaticRequest
    out = []
    for mv in moves or ():
        if isinstance(mv, HMC):
            out.append(("hmc", mv))
        elif isinstance(mv, StaticRequest) or isinstance(mv, dict):
            addressed = mv.addressed if isinstance(mv, StaticRequest) else mv
            for addr, req in addressed.items():
                if not isinstance(req, Rejuvenate):
                    raise NotImplementedError(f"filter moves: {type(req).__name__} at {addr!r} (HMC, or Rejuvenate under an address)")
                out.append(("proposal", addr, req))
        elif isinstance(mv, Rejuvenate):
            raise ValueError("filter moves: address a Rejuvenate move — {addr: Rejuvenate(dist, argument_mapping)}")
        else:
            raise NotImplementedError(f"filter moves: {type(mv).__name__} (HMC, or Rejuvenate under an address)")
    return out


class DeviceBackend:
    """the calls a move is made of, on the HIP library"""

    def __init__(self, dev):
        self.dev = dev

    def run(self, prog: PackedProgram, key: Key, K: int, choices, want_site_scores=False) -> dict:
        from .. import kernels
        return kernels.run_program(prog, key, K, choices=choices, want_lse=False, want_weight=False, want_site_scores=want_site_scores,
                                   device=self.dev, ws=kernels.shared_workspace(A.OP_RUN, K, self.dev))

    def hmc(self, prog: PackedProgram, key: Key, rows, eps: float, L: int) -> dict:
        from .. import kernels
        return kernels.hmc(prog, key, rows, eps, L, stale=False, accept=True)

    def accept(self, log_alpha, key: Key, rows_cur, rows_prop):
        """gjx_mh_accept: the caller-side accept (log u < alpha) in place on rows_cur; -> accepted chains (a device scalar: the run
        reads its counters once, at its end)"""
        from .. import kernels
        return kernels.mh_accept(log_alpha, key, rows_cur, rows_prop).sum()

    def empty(self, rows: int, K: int):
        return torch.empty((max(rows, 1), K), dtype=torch.float32, device=self.dev)

    def clone(self, x):
        return x.clone()

    def count(self, mask):
        return mask.sum()


def target_program(step_prog: PackedProgram, selected=(), rng_mode=None) -> PackedProgram:
    """the step program in ASSESS form: INPUT sites stay inputs (the ancestor's own inputs), the step's latent choices are given per
    particle (OBS_SLOT), its observations stay in the table; ``selected``: the latents an HMC move changes.  Same rows as the step."""
    sl = step_prog.site_list
    modes, obs = {}, {}
    for s in sl.sites:
        md = step_prog.modes.get(s.addr, A.MODE_SAMPLE)
        if md == A.MODE_INPUT:
            modes[s.addr] = A.MODE_INPUT
        elif md == A.MODE_OBS_TAB:
            modes[s.addr] = A.MODE_OBS_TAB
            obs[s.addr] = step_prog.filter_obs[s.addr]
        elif md == A.MODE_SAMPLE:
            modes[s.addr] = A.MODE_OBS_SLOT
        else:
            raise NotImplementedError("filter moves: a step with a custom proposal (the move's target is the step's prior-proposal form)")
    p = PackedProgram(sl, modes, obs, selected=tuple(selected), rng_mode=step_prog.rng_mode if rng_mode is None else rng_mode, plates=False)
    for s in sl.sites:
        if p.slot_of[s.addr] != step_prog.slot_of[s.addr]:
            raise AssertionError(f"filter moves: the assess form lays {s.addr!r} out differently from the step program")
    return p


def hmc_selection(step_prog: PackedProgram, req) -> list:
    """the latent choices of a step that an HMC request's selection names (float leaves only, as hmc.py:49-65)"""
    sel = [s.addr for s in step_prog.site_list.sites if step_prog.modes.get(s.addr, A.MODE_SAMPLE) == A.MODE_SAMPLE
           and req.selection.check(_bare(s.addr)) and s.kind not in A.NO_GRADIENT_KINDS]
    if not sel:
        raise ValueError("filter moves: the HMC selection names no continuous latent of the step")
    return sel


def hmc_targets(progs, req) -> list:
    """gjx_filter_opts::hmc_targets of a run: the assess forms of steps 0 .. T-2 with the request's selection flagged"""
    return [target_program(p, hmc_selection(p, req)) for p in progs[:-1]]


def proposal_programs(rej, dim: int, rng_mode: int):
    """(forward, backward) programs of a Rejuvenate proposal on a value of event size ``dim``: sites "cur" (the current value, given)
    and "new" (drawn from / scored under proposal(*argument_mapping(cur)))"""
    gf = rej._proposal_gf(dim)
    fwd = gf.pack((), ChoiceMap.empty(), True, rng_mode=rng_mode, per_particle=("cur",))[0]
    bwd = gf.pack((), ChoiceMap.empty(), False, rng_mode=rng_mode, per_particle=("cur", "new"))[0]
    return fwd, bwd


def apply_hmc(b, prog_h: PackedProgram, key: Key, R, eps: float, L: int):
    """one HMC move with the accept fused, in place on the rows R of the step-local target; -> (R, accepted count)"""
    out = b.hmc(prog_h, key, R, eps, L)
    return out["choices"], b.count(out["accepted"] > 0.5)


def apply_proposal(b, target: PackedProgram, addr, progs_q, key: Key, R, K: int):
    """one Metropolis-Hastings move of the choice at ``addr`` with a Rejuvenate proposal (rejuvenate.py:70-94 + the caller's accept):
    -> (R with the accepted proposals in the rows of addr, accepted count).  key -> (k_accept, k_draw)."""
    fwd_p, bwd_p = progs_q
    d = int(target.site_list[addr].dim)
    s0 = target.slot_of[addr]
    k_acc, k_draw = split(key)
    old = b.clone(R[s0:s0 + d])
    cf = b.empty(fwd_p.n_slots, K)
    cf[fwd_p.slot_of["cur"]: fwd_p.slot_of["cur"] + d] = old
    of = b.run(fwd_p, k_draw, K, cf, want_site_scores=True)
    new = b.clone(of["choices"][fwd_p.slot_of["new"]: fwd_p.slot_of["new"] + d])
    fwd = of["site_scores"][1]
    cb = b.empty(bwd_p.n_slots, K)
    cb[bwd_p.slot_of["cur"]: bwd_p.slot_of["cur"] + d] = new
    cb[bwd_p.slot_of["new"]: bwd_p.slot_of["new"] + d] = old
    bwd = b.run(bwd_p, k_draw, K, cb, want_site_scores=True)["site_scores"][1]
    lp_old = b.clone(b.run(target, k_draw, K, R)["score"])
    Rn = b.clone(R)
    Rn[s0:s0 + d] = new
    lp_new = b.run(target, k_draw, K, Rn)["score"]
    alpha = (lp_new - lp_old) + (bwd - fwd)
    na = b.accept(alpha, k_acc, R[s0:s0 + d], new)
    return R, na


def step_rows(prog: PackedProgram):
    """(number of INPUT rows, rows of the step's own latent choices in row order) of a step program without proposal sites"""
    n_in = sum(s.dim for s in prog.site_list.sites if prog.modes.get(s.addr) == A.MODE_INPUT)
    own = sorted((prog.slot_of[s.addr], s.dim) for s in prog.site_list.sites
                 if prog.modes.get(s.addr, A.MODE_SAMPLE) == A.MODE_SAMPLE and prog.slot_of[s.addr] >= 0)
    return n_in, own


def run_with_moves(bf, key: Key, constraint: ChoiceMap, args, moves, device=None, keep_ancestors: bool = False, backend=None,
                   record: list | None = None) -> dict:
    """the filter of ``bf`` (a ScanBootstrapFilter without custom proposal and without sites in front of the Scan) step by step, with
    ``moves`` applied to the gathered carry behind every resampling.  Same result dict as ``bf.run``; ``accepted`` = accepted
    proposals per move over the run.  ``record``: a list that receives, per step with moves, the rows before and after every move
    (tests)."""
    from .. import kernels
    from .pf import _unit_from_key
    dev = kernels._dev(device)
    b = backend or DeviceBackend(dev)
    # (the step programs, the moves' target and proposal programs and their device tables are kept per structure + data of the run,
    # as ScanBootstrapFilter.run keeps its own: a second run with the same observations packs and uploads nothing)
    from .scan_filter import _run_keys
    sk, dk = _run_keys(constraint, args, dev)
    mc = bf.__dict__.get("_move_cache")
    if mc is None or mc["sk"] != sk or mc["dk"] != dk or mc["moves"] is not moves:
        mc = bf._move_cache = dict(sk=sk, dk=dk, moves=moves, progs=bf.step_programs(constraint, args), cache={})
        for p_ in mc["progs"]:
            _bind(p_, dev)
    progs = mc["progs"]
    if bf.proposal is not None or getattr(bf, "_has_statics", False):
        raise NotImplementedError("filter moves: the step-by-step form runs the prior proposal on a model that is the Scan")
    T, K = len(progs), bf.K
    specs = normalise_moves(moves)
    f32 = torch.float32
    rows = [torch.empty((max(p.n_slots, 1), K), dtype=f32, device=dev) for p in progs[:2]]
    lse = torch.empty((T, 4), dtype=f32, device=dev)
    logw = [torch.empty(K, dtype=f32, device=dev) for _ in range(2)]
    anc_all = torch.empty((max(T - 1, 1), K), dtype=torch.int32, device=dev) if keep_ancestors else None
    anc = torch.empty(K, dtype=torch.int32, device=dev)
    ws_run, ws_res = kernels.workspace(A.OP_RUN, K, dev), kernels.workspace(A.OP_RESAMPLE, K, dev)
    cum = torch.empty(K, dtype=torch.int64, device=dev)
    accepted = [0] * len(specs)
    cache: dict = mc["cache"]
    k = key
    for t in range(T):
        k = fold_in(k, t)
        k_prop, k_res = split(k)
        p = progs[t]
        cur = rows[t & 1] if rows[t & 1].shape[0] >= max(p.n_slots, 1) else torch.empty((max(p.n_slots, 1), K), dtype=f32, device=dev)
        rows[t & 1] = cur
        out = dict(logw=logw[t & 1], lse=lse[t])
        if t == 0:
            kernels.run_program(p, k_prop, K, choices=cur, ws=ws_run, out=out, want_weight=False, device=dev)
            continue
        pp = progs[t - 1]
        prev = rows[(t - 1) & 1]
        n_in_prev, own_prev = step_rows(pp)
        if bf.resampler == "multinomial":
            kernels.resample_sorted_multinomial_tiled(logw[(t - 1) & 1], k_res, K, anc=anc, cum=cum, ws=ws_res)
        else:
            kernels.resample_indices_tiled(logw[(t - 1) & 1], _unit_from_key(k_res), K, anc=anc, cum=cum, ws=ws_res)
        if anc_all is not None:
            anc_all[t - 1] = anc
        lo = own_prev[0][0] if own_prev else n_in_prev
        n_own = sum(d for _, d in own_prev)
        if own_prev and (lo != n_in_prev or own_prev[-1][0] + own_prev[-1][1] != lo + n_own):
            raise NotImplementedError("filter moves: the step's latent rows are not one block behind its inputs")
        if specs:
            # the gathered particle of step t-1: [its own inputs | its latent choices], then the moves on it
            R = kernels.gather_rows(prev[: n_in_prev + n_own], anc)
            rec = dict(t=t, before=R.clone(), after=[]) if record is not None else None
            for m, spec in enumerate(specs):
                km = move_key(k_prop, m)
                if spec[0] == "hmc":
                    req = spec[1]
                    sel = hmc_selection(pp, req)
                    tk = ("target", t - 1, tuple(sel))
                    if tk not in cache:
                        cache[tk] = _bind(target_program(pp, sel), dev)
                    R, na = apply_hmc(b, cache[tk], km, R, req.eps, req.L)
                else:
                    _, name, rej = spec
                    addr = _find(pp, name)
                    d = int(pp.site_list[addr].dim)
                    qk = ("q", m, d)
                    if qk not in cache:
                        cache[qk] = tuple(_bind(q, dev) for q in proposal_programs(rej, d, pp.rng_mode))
                    tk = ("target", t - 1, ())
                    if tk not in cache:
                        cache[tk] = _bind(target_program(pp), dev)
                    R, na = apply_proposal(b, cache[tk], addr, cache[qk], km, R, K)
                accepted[m] = accepted[m] + na
                if rec is not None:
                    rec["after"].append(R.clone())
            if rec is not None:
                record.append(rec)
            kernels.run_program(p, k_prop, K, choices=cur, ws=ws_run, out=out, want_weight=False, device=dev, in_rows=R[n_in_prev:],
                                store_inputs=True)
        else:
            kernels.run_program(p, k_prop, K, choices=cur, ws=ws_run, out=out, want_weight=False, device=dev, in_rows=prev[lo:],
                                ancestors=anc, store_inputs=True)
    incs = lse[:, 3]
    last = progs[-1]
    # (the resamplers flag a collection without weight — identity ancestors — in their workspace's status word: read once, here)
    st = kernels.workspace_status(ws_res, raise_on_error=False) if T > 1 else 0
    return dict(log_ml=incs.sum(), increments=incs, lse_steps=lse, choices=rows[(T - 1) & 1][: max(last.n_slots, 1)], logw=logw[(T - 1) & 1],
                programs=progs, ancestors=anc_all if keep_ancestors else anc, degenerate=bool(st & 2), accepted=[int(a) for a in accepted],
                info=dict(form=A.FILTER_FORM_TWO_LAUNCH, form_name="step by step with moves (device calls per step)", launches=None, grid=0,
                          tiles_per_block=0))


def _bare(addr):
    """the address of a step's site without its step index: ("x", 3) -> "x" (what a user's selection names)"""
    from .scan_filter import _name
    return _name(addr)


def _find(prog: PackedProgram, name):
    for s in prog.site_list.sites:
        if (s.addr == name or _bare(s.addr) == name) and prog.modes.get(s.addr, A.MODE_SAMPLE) == A.MODE_SAMPLE:
            return s.addr
    raise KeyError(f"filter moves: the step has no latent choice {name!r}")


def _bind(prog: PackedProgram, dev):
    """table on the device (PackedProgram.c_program uploads on first use per device)"""
    prog.c_program(dev)
    return prog
