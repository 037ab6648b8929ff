"""Edit requests: Update, Regenerate, HMC (reference: core/generative/requests.py, concepts.py:95-168,
inference/requests/hmc.py).  Each ``edit`` is one kernel launch over all chains of a batched Trace;
``edit(key, tr, argdiffs) -> (new_trace, weight, retdiff, backward_request)`` as in concepts.py:95-109.
"""
from __future__ import annotations

import numpy as np

from .. import _abi as A
from ..core import ChoiceMap, Key, Selection
from ..gen import Trace, _value_rows


class EditRequest:
    def edit(self, key: Key, tr: Trace, argdiffs=None):
        raise NotImplementedError


def _rows_and_shared(tr: Trace):
    """Current values of a trace split the way its program stores them."""
    shared = ChoiceMap({a: v for a, v in tr.shared.items()})
    rows = {s.addr: tr.choices[tr.prog.slot_of[s.addr]: tr.prog.slot_of[s.addr] + s.dim]
            for s in tr.prog.site_list.sites if tr.prog.slot_of[s.addr] >= 0}
    return shared, rows


class Update(EditRequest):
    """Replace the values at the constrained addresses, keep the rest; weight = new score - old score
    (generative_function.py:1687-1689, distribution.py:179-244, static.py:827-865)."""

    def __init__(self, constraint: ChoiceMap | None = None):
        self.constraint = constraint or ChoiceMap.empty()

    def edit(self, key: Key, tr: Trace, argdiffs=None):
        shared, rows = _rows_and_shared(tr)
        discard = {}
        for addr, v in self.constraint.items():
            s = tr.prog.site_list[addr]
            discard[addr] = tr._site_value(addr)
            sv, r = _value_rows(v, s.dim)
            if addr in rows:
                import torch
                rows[addr] = r if r is not None else torch.as_tensor(sv, device=tr.score.device).reshape(s.dim, 1)
            else:
                if r is not None:
                    raise ValueError(f"{addr!r} is constrained to one shared value in this trace")
                shared = ChoiceMap({**dict(shared.items()), addr: sv})
        new_tr, out = tr.gen_fn._run(key, tr.K, tr.args, shared, False, tr.batched, prev_rows=rows)
        w = out["score"] - tr.score
        return new_tr, (w if tr.batched else w[0]), None, Update(ChoiceMap(discard))


class Regenerate(EditRequest):
    """Resample the selected addresses from their (new-argument) prior, keep the rest; the weight is
    new score - old score summed over every visited site (distribution.py:258-300 line 269,
    static.py:906-946; pinned by tests/inference/test_requests.py:52-57)."""

    def __init__(self, selection: Selection):
        self.selection = selection

    def edit(self, key: Key, tr: Trace, argdiffs=None):
        shared, rows = _rows_and_shared(tr)
        old = {}
        for s in tr.prog.site_list.sites:
            if self.selection.check(s.addr):
                old[s.addr] = tr._site_value(s.addr)
                if s.addr in rows:
                    del rows[s.addr]
                else:
                    shared = ChoiceMap({a: v for a, v in shared.items() if a != s.addr})
        new_tr, out = tr.gen_fn._run(key, tr.K, tr.args, shared, True, tr.batched, prev_rows=rows)
        w = out["score"] - tr.score
        return new_tr, (w if tr.batched else w[0]), None, Update(ChoiceMap(old))


class HMC(EditRequest):
    """Hamiltonian Monte Carlo move on the selected float sites (hmc.py:138-211).  Returns the new
    trace and alpha; the caller accepts with ``log(uniform) < alpha`` (test_requests.py:134-137) or
    passes ``accept=True`` to fuse that rule into the kernel.

    ``stale_gradient_compat=True`` reproduces hmc.py:186, where the scan carry returns the gradient it
    RECEIVED, so every step's first half-kick uses the gradient at the initial position."""

    def __init__(self, selection: Selection, eps, L: int = 10, stale_gradient_compat: bool = False,
                 accept: bool = False):
        self.selection, self.eps, self.L = selection, float(np.asarray(eps)), int(L)
        self.stale_gradient_compat, self.accept = stale_gradient_compat, accept
        self.last_accepted = None

    def edit(self, key: Key, tr: Trace, argdiffs=None):
        from .. import kernels
        shared, rows = _rows_and_shared(tr)
        sel = [s.addr for s in tr.prog.site_list.sites if self.selection.check(s.addr)]
        prog, _, _ = tr.gen_fn.pack(tr.args, shared, False, selected=sel, rng_mode=tr.prog.rng_mode,
                                    per_particle=tuple(rows))
        assert prog.slot_of == tr.prog.slot_of
        out = kernels.hmc(prog, key, tr.choices.clone(), self.eps, self.L, self.stale_gradient_compat, self.accept)
        new_tr = Trace(tr.gen_fn, tr.args, tr.prog, out["choices"], out["score"], tr.shared, tr.batched, tr.retval_sym)
        alpha = out["alpha"]
        bwd = HMC(self.selection, self.eps, self.L, self.stale_gradient_compat, self.accept)
        bwd.last_accepted = self.last_accepted = out["accepted"] if self.accept else None
        return new_tr, (alpha if tr.batched else alpha[0]), None, bwd


def SafeHMC(selection: Selection, eps, L: int = 10) -> HMC:
    """hmc.py:214-223 — the retdiff assertion is vacuous here (argdiffs are always no-change)."""
    return HMC(selection, eps, L)
