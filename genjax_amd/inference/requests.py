"""Edit requests: Update, Regenerate, HMC (reference: core/generative/requests.py, concepts.py:95-168,
inference/requests/hmc.py).  Each ``edit`` is one kernel launch over all chains of a batched Trace;
``edit(key, tr, argdiffs) -> (new_trace, weight, retdiff, backward_request)`` as in concepts.py:95-109.
"""
from __future__ import annotations

import numpy as np

from .. import _abi as A
from ..core import ChoiceMap, Key, Selection, key_of
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


def _unrolled_score(tr: Trace):
    """The trace's score with the SAME order of summation as an edit's re-run: edits run on the unrolled lowering (one site
    per plate instance — a move may treat instances differently), and their weight is a difference of two float32 sums of
    all site scores, exact only if the unchanged terms are added in the same order on both sides.  A trace that came from
    a program with plates (vector sites: per-plate partial sums) is therefore re-assessed once on the unrolled program."""
    if not getattr(tr.prog, "plate_of", None):
        return tr.score
    shared, rows = _rows_and_shared(tr)
    _, out = tr.gen_fn._run((0, 0), tr.K, tr.args, shared, False, tr.batched, prev_rows=rows, plates=False)
    return out["score"]


def _new_args(tr: Trace, argdiffs):
    """arguments of the edited trace: the primals of ``argdiffs`` (Diff-tagged or plain), or the old ones"""
    from ..core import Diff
    if argdiffs is None or (isinstance(argdiffs, tuple) and len(argdiffs) == 0 and len(tr.args) != 0):
        return tr.args
    args = Diff.tree_primal(argdiffs)
    args = tuple(args) if isinstance(args, (tuple, list)) else (args,)
    if len(args) != len(tr.args):
        raise ValueError(f"argdiffs has {len(args)} entries, the trace's generative function takes {len(tr.args)}")
    return args


def _per_site(constraint: ChoiceMap, site_list) -> dict:
    """the constraint keyed by site address: a whole-sequence entry (``C[:].set({"x": xs})``, ``C["x"].set(xs)`` for the sites
    ("x", i) of a Scan / Vmap / repeat) is dealt to its instances, leading axis = instance index"""
    from ..core import norm_addr
    addrs = {st.addr for st in site_list.sites}
    out = {}
    for addr, v in constraint._d.items():
        if addr in addrs or addr == ():
            out[addr] = v
            continue
        members = [a for a in addrs if isinstance(a, tuple) and len(a) == 2 and a[0] == addr and norm_addr(a)[1] is not None]
        if not members:
            out[addr] = v                                   # unknown address: reported by the lookup below
            continue
        arr = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        for a in members:
            out[a] = arr[a[1]]
    return out


class Update(EditRequest):
    """Replace the values at the constrained addresses, keep the rest; weight = new score - old score
    (generative_function.py:1687-1689, distribution.py:179-244, static.py:827-865)."""

    def __init__(self, constraint: ChoiceMap | None = None):
        self.constraint = constraint or ChoiceMap.empty()

    def edit(self, key: Key, tr: Trace, argdiffs=None):
        shared, rows = _rows_and_shared(tr)
        args = _new_args(tr, argdiffs)
        discard = {}
        for addr, v in _per_site(self.constraint, tr.prog.site_list).items():   # including the bare-value address () of a distribution trace
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
        old_score = _unrolled_score(tr)
        new_tr, out = tr.gen_fn._run(key, tr.K, args, shared, False, tr.batched, prev_rows=rows, plates=False)
        w = out["score"] - old_score
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
        old_score = _unrolled_score(tr)
        new_tr, out = tr.gen_fn._run(key, tr.K, _new_args(tr, argdiffs), shared, True, tr.batched, prev_rows=rows, plates=False)
        w = out["score"] - old_score
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
        # float leaves only, as the reference's selection_gradient (hmc.py:49-65, 90-96)
        sel = [s.addr for s in tr.prog.site_list.sites if self.selection.check(s.addr) and s.kind not in A.NO_GRADIENT_KINDS]
        prog, _, _ = tr.gen_fn.pack(tr.args, shared, False, selected=sel, rng_mode=tr.prog.rng_mode,
                                    per_particle=tuple(rows), plates="hmc")         # (vector form for table-valued plates, else plate-tagged)
        # (the move's program may lay its rows out differently from the trace's: other modes, plates; the new trace is bound
        # to the program its rows follow)
        self.last_program = prog             # the move's program (every site constrained, the selected ones flagged): tests, engine queries
        out = kernels.hmc(prog, key, tr.rows_for(prog), self.eps, self.L, self.stale_gradient_compat, self.accept)
        new_tr = Trace(tr.gen_fn, tr.args, prog if prog.slot_of != tr.prog.slot_of else tr.prog, out["choices"], out["score"], tr.shared,
                       tr.batched, tr.retval_sym)
        alpha = out["alpha"]
        bwd = HMC(self.selection, self.eps, self.L, self.stale_gradient_compat, self.accept)
        bwd.last_accepted = self.last_accepted = out["accepted"] if self.accept else None
        return new_tr, (alpha if tr.batched else alpha[0]), None, bwd


def SafeHMC(selection: Selection, eps, L: int = 10) -> HMC:
    """hmc.py:214-223 — the retdiff assertion is vacuous here (argdiffs are always no-change)."""
    return HMC(selection, eps, L)


class Rejuvenate(EditRequest):
    """Metropolis-Hastings proposal move without the accept step (inference/requests/rejuvenate.py:70-94):
    propose z' ~ q(. | argument_mapping(old choices)), apply it as an Update (weight w), score the reverse
    proposal, return w + log q(z_old | args(z')) - log q(z' | args(z_old)).  Used through
    ``StaticRequest({addr: Rejuvenate(dist, argument_mapping)})``; ``argument_mapping`` receives a value
    ChoiceMap (``chm.get_value()`` is the current choice, scalar or vector-valued) and is traced symbolically once per event size."""

    def __init__(self, proposal, argument_mapping):
        self.proposal, self.argument_mapping = proposal, argument_mapping
        self._gf = {}

    def _proposal_gf(self, dim: int = 1):
        """the proposal as a two-site generative function: a carrier of the current value (event size ``dim``; its score is not
        used) and the proposal distribution on the arguments ``argument_mapping`` makes of it"""
        if dim not in self._gf:
            from ..gen import StaticGenerativeFunction, mv_normal_diag, normal
            prop, amap = self.proposal, self.argument_mapping

            def body():
                cur = (normal(0.0, 1.0) if dim == 1 else mv_normal_diag(np.zeros(dim, np.float32), np.ones(dim, np.float32))) @ "cur"
                args = amap(ChoiceMap.v(cur))
                _ = prop(*args) @ "new"
            self._gf[dim] = StaticGenerativeFunction(body)
        return self._gf[dim]

    def edit_at(self, key: Key, tr: Trace, addr):
        from ..core import split
        site = tr.prog.site_list[addr]
        d = int(site.dim)
        if tr.prog.slot_of[addr] < 0:
            raise NotImplementedError("Rejuvenate: the address is constrained to one value for every particle (nothing to move)")
        gf = self._proposal_gf(d)
        old = tr.choices[tr.prog.slot_of[addr]: tr.prog.slot_of[addr] + d]          # [d][K]
        key, sub_key = split(key)
        fwd_tr, fwd_out = gf._run(sub_key, tr.K, (), ChoiceMap.empty(), True, True, prev_rows={"cur": old},
                                  want_site_scores=True)
        if fwd_tr.prog.site_list["new"].dim != d:
            raise ValueError(f"Rejuvenate: the proposal's event size {fwd_tr.prog.site_list['new'].dim} differs from the address's {d}")
        new = fwd_tr.choices[fwd_tr.prog.slot_of["new"]: fwd_tr.prog.slot_of["new"] + d]
        fwd = fwd_out["site_scores"][1]
        new_val = (new[0] if tr.batched else new[0, 0]) if d == 1 else (new.T if tr.batched else new[:, 0])
        new_tr, w, _, bwd_req = Update(ChoiceMap({addr: new_val})).edit(key, tr, None)
        _, bwd_out = gf._run(sub_key, tr.K, (), ChoiceMap.empty(), False, True, prev_rows={"cur": new, "new": old},
                             want_site_scores=True)
        bwd = bwd_out["site_scores"][1]
        final = (w if tr.batched else w.reshape(1)) + bwd - fwd
        return new_tr, (final if tr.batched else final[0]), None, Rejuvenate(self.proposal, self.argument_mapping)

    def edit(self, key: Key, tr: Trace, argdiffs=None):
        raise NotImplementedError("address a Rejuvenate move through StaticRequest({addr: Rejuvenate(...)})")


class IndexRequest(EditRequest):
    """A request for ONE instance of a vmap / scan level (concepts.py:154-165); used inside a StaticRequest as
    ``{"a": IndexRequest(idx, Regenerate(S.all()))}``."""

    def __init__(self, idx, request: EditRequest):
        self.idx, self.request = int(np.asarray(idx)), request


class StaticRequest(EditRequest):
    """Address-wise composition of requests (generative_functions/static.py:130-131): ``{addr: request}``."""

    def __init__(self, addressed: dict, absolute=()):
        self.addressed = dict(addressed)
        self.absolute = list(absolute)      # requests already expressed on the enclosing trace (backward of re-rooted ones)

    @staticmethod
    def _join(prefix, addr):
        pre = tuple(prefix) if isinstance(prefix, tuple) else (prefix,)
        return pre + (tuple(addr) if isinstance(addr, tuple) else (addr,))

    @classmethod
    def _reroot(cls, req, prefix, sites=()):
        """the request ``req`` addressed to the callee at ``prefix``, as a request on the enclosing trace whose site
        keys are ``sites``"""
        if isinstance(req, HMC):
            return HMC(req.selection.prefixed(prefix), req.eps, req.L, req.stale_gradient_compat, req.accept)
        if isinstance(req, Regenerate):
            return Regenerate(req.selection.prefixed(prefix))
        if isinstance(req, Update):
            return Update(ChoiceMap({key_of(cls._join(prefix, a)): v for a, v in req.constraint.items()}))
        if isinstance(req, StaticRequest):
            return StaticRequest({key_of(cls._join(prefix, a)): r for a, r in req.addressed.items()})
        if isinstance(req, IndexRequest):
            return cls._reroot_indexed(req, prefix, sites)
        raise NotImplementedError(f"{type(req).__name__} addressed to a sub-generative-function")

    @classmethod
    def _reroot_indexed(cls, req: "IndexRequest", prefix, sites):
        """instance ``idx`` of the vmap / scan level at ``prefix``: its sites are the keys (name, idx) whose name path
        starts with the prefix; the sub-request's addresses are relative to the instance"""
        from ..core import norm_addr
        pre = tuple(prefix) if isinstance(prefix, tuple) else (prefix,)
        inst = []
        for k in sites:
            name, idx = norm_addr(k)
            path = name if isinstance(name, tuple) else (name,)
            if idx == req.idx and path[: len(pre)] == pre:
                rel = path[len(pre):]
                inst.append((k, rel[0] if len(rel) == 1 else rel))      # rel == () for a vmapped distribution
        if not inst:
            raise KeyError(f"IndexRequest({req.idx}) addressed to {prefix!r}: no such instance")
        sub = req.request
        if isinstance(sub, Regenerate):
            return Regenerate(Selection(k for k, rel in inst if sub.selection.check(rel)))
        if isinstance(sub, Update):
            if sub.constraint.has_value():
                (k, _), = [x for x in inst if x[1] == ()]
                return Update(ChoiceMap({k: sub.constraint.get_value()}))
            return Update(ChoiceMap({k: sub.constraint[rel] for k, rel in inst if rel in sub.constraint}))
        if isinstance(sub, HMC):
            return HMC(Selection(k for k, rel in inst if sub.selection.check(rel)), sub.eps, sub.L, sub.stale_gradient_compat, sub.accept)
        raise NotImplementedError(f"IndexRequest around {type(sub).__name__}")

    def edit(self, key: Key, tr: Trace, argdiffs=None):
        from ..core import fold_in
        sites = [s.addr for s in tr.prog.site_list.sites]
        total, bwd, bwd_abs = None, {}, []
        for n, req in enumerate(self.absolute):
            tr, w, _, b = req.edit(fold_in(key, 1000 + n), tr, argdiffs)
            total = w if total is None else total + w
            bwd_abs.append(b)
        for n, (addr, req) in enumerate(self.addressed.items()):
            k = fold_in(key, n + 1)
            if key_of(addr) not in tr.prog.site_list:          # the address of a callee: recurse with re-rooted request
                if isinstance(req, Update) and req.constraint.has_value():
                    raise KeyError(f"Update(C.choice(v)) addressed to {addr!r}, which is not a leaf choice")
                tr, w, _, b = self._reroot(req, addr, sites).edit(k, tr, argdiffs)
                total = w if total is None else total + w
                bwd_abs.append(b)                              # its addresses are already those of this trace
                continue
            addr = key_of(addr)
            if isinstance(req, Rejuvenate):
                tr, w, _, b = req.edit_at(k, tr, addr)
            elif isinstance(req, Regenerate):
                tr, w, _, b = Regenerate(Selection((addr,))).edit(k, tr, argdiffs)
            elif isinstance(req, Update):
                v = req.constraint.get_value() if req.constraint.has_value() else req.constraint[addr]
                tr, w, _, b = Update(ChoiceMap({addr: v})).edit(k, tr, argdiffs)
            elif isinstance(req, HMC):
                tr, w, _, b = HMC(Selection((addr,)), req.eps, req.L, req.stale_gradient_compat, req.accept).edit(k, tr, argdiffs)
            else:
                raise NotImplementedError(type(req).__name__)
            total = w if total is None else total + w
            bwd[addr] = b
        return tr, total, None, StaticRequest(bwd, bwd_abs)
