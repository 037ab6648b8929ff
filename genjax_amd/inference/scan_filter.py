"""Bootstrap particle filter for ANY Scan kernel (SURVEY.md §8 R-2 beyond the one hand-coded linear-Gaussian model).

The reference supplies the ingredients and no filter: ``Scan.generate`` runs a kernel ``(carry, x) ->
(carry, out)`` T times, step t receiving the carry of step t-1, keys chained ``key_t = fold_in(key_{t-1}, t)``, weights added
over steps (combinators/scan.py:237-294); ``Scan.edit_index`` / ``IndexRequest`` extend a trace by one step (scan.py:325-416).
Here the unrolled trace of ``kernel.scan(n=T)`` is cut into T one-step programs — the sites of step t plus GJX_MODE_INPUT
sites standing for the choices of step t-1 it reads — and the C loop ``gjx_scan_filter`` runs, per step, the step's generated
propagate + reweight kernel with the tile-scaled systematic resampler's search in its prologue (one launch per step; two where
the kernel's shape does not allow it), reading its carry through the ancestors (include/gjx.h).  Periodic Scans give T-1 programs of one structure: one kernel is generated, only the tables
(the step's observation) differ.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _abi as A
from .. import config
from .._lib import check, load
from ..core import ChoiceMap, Key
from ..gen import MissingAddress, ScanCombinator, _constraint_value, _value_rows
from ..program import PackedProgram, Site, SiteList, fold_known


def _step_of(site) -> int:
    return (int(site.scan) & 0xFFFFF) - 1


class ScanBootstrapFilter:
    """``ScanBootstrapFilter(kernel.scan(n=T), k_particles).run(key, constraint, (carry0, xs))``: SMC with the prior as proposal
    and systematic resampling in front of every step.  ``constraint`` holds the observations of every step (``C["y"].set(ys)``
    with a leading step axis, or ``C[t, "y"]`` per step); the sites it names are the observed ones (the SAME sites in every
    step), the others are propagated."""

    def __init__(self, scan: ScanCombinator, k_particles: int, rng_mode: int | None = None, proposal: ScanCombinator | None = None,
                 proposal_args=None, rejuvenate: dict | None = None, resampler: str = "systematic", moves=None):
        """``proposal``: ``q_step.scan(n=T)`` — a kernel ``(carry, x) -> (carry, out)`` like the model's whose sites PROPOSE the model's
        latent choices of the same names (the importance step with a custom proposal, inference/smc.py:302-313, applied per Scan step:
        scan.py:325-416 extends a trace by one step): step t draws from q_t(. | carry, x_t) and weights by
        log p(latents, observations | carry) - log q(latents) (proper weighting: SURVEY.md §9 H2).  ``proposal_args``: its (carry0, xs)
        — e.g. the observations as xs for a look-ahead proposal; latents it does not name are drawn from the model's prior.
        ``rejuvenate=dict(n_moves=n, scale=s)``: resample-move — behind every resampling from the second on each particle's carry takes n
        random-walk Metropolis steps that leave the previous step's posterior invariant (the reference's Rejuvenate with a symmetric
        proposal and the caller-side accept, requests/rejuvenate.py:70-94; generated from the step program, include/gjx.h
        gjx_filter_opts::n_moves); ``out["accepted_total"]`` counts the accepted moves of a run.
        ``moves=[...]``: ANY of the library's move requests behind every resampling — ``HMC(S["x"], eps, L)`` over the step's continuous
        latents (hmc.py:138-211), ``{"x": Rejuvenate(dist, argument_mapping)}`` with an arbitrary proposal (rejuvenate.py:70-94) — each
        with the caller-side accept (test_requests.py:131-137) on the step-local target; the filter then runs step by step as device
        calls (inference/filter_moves.py) instead of in the one-launch kernel; ``out["accepted"]`` counts per move."""
        from ..gen import StaticGenerativeFunction
        if not isinstance(scan, (ScanCombinator, StaticGenerativeFunction)):
            raise TypeError("ScanBootstrapFilter needs kernel.scan(n=T), or a @gen model whose body is sites in front of one kernel.scan(n=T)(...) call")
        if proposal is not None and not isinstance(proposal, ScanCombinator):
            raise TypeError("proposal must be q_step.scan(n=T)")
        self.scan, self.K = scan, int(k_particles)
        self.proposal, self.proposal_args = proposal, proposal_args
        self.rejuvenate = dict(rejuvenate) if rejuvenate else None
        self.moves = list(moves) if moves else None
        if self.moves and self.rejuvenate:
            raise ValueError("give the random-walk move (rejuvenate=...: inside the one-launch kernel) or moves=[...] (step by step), not both")
        if self.moves:
            from .filter_moves import normalise_moves
            normalise_moves(self.moves)       # (raises on what is not a move)
        if resampler not in ("systematic", "multinomial"):
            raise ValueError("resampler must be 'systematic' or 'multinomial'")
        self.resampler = resampler            # multinomial: every slot draws its own uniform (GJX_FILTER_MULTINOMIAL; three plain launches per step)
        self.rng_mode = config.rng_mode() if rng_mode is None else rng_mode
        self._cache: dict = {}

    # -- the T one-step programs -------------------------------------------------------------------------------
    def step_programs(self, constraint: ChoiceMap, args) -> list[PackedProgram]:
        sl, _ = self.scan.site_list(tuple(args))
        if len({(int(s.scan) & 0xFFFFFFFF) >> 20 for s in sl.sites}) > 1:
            raise NotImplementedError("ScanBootstrapFilter: the kernel contains a Scan of its own (its steps cannot be told from the filter's)")
        steps: list[list[Site]] = []
        pre: list[Site] = []                  # sites in FRONT of the Scan: static parameters (Scan.generate is a callee of any @gen body, scan.py:237-294)
        for s in sl.sites:
            t = _step_of(s)
            if t < 0:
                if steps:
                    raise NotImplementedError("ScanBootstrapFilter: sites BEHIND the Scan are not filtered (sites in front of it are)")
                pre.append(s)
                continue
            while len(steps) <= t:
                steps.append([])
            steps[t].append(s)
        q_steps: list[list[Site]] = [[] for _ in steps]
        if self.proposal is not None:
            if self.proposal_args is None:
                raise ValueError("a proposal needs proposal_args=(carry0, xs)")
            qsl, _ = self.proposal.site_list(tuple(self.proposal_args))
            for s in qsl.sites:
                t = _step_of(s)
                if t < 0 or t >= len(steps):
                    raise NotImplementedError("ScanBootstrapFilter: the proposal must be a Scan of the model's length")
                q_steps[t].append(s)
        progs = []
        prev_latent: list[Site] = []
        known: dict = {}                      # observed sites of the previous step: address -> value
        # the sites in front of the Scan: observed ones are constants of every step; latent ones are drawn by step 0 and travel with
        # the particle (GJX_SITE_CARRIED inputs of every later step, GJX_FILTER_ABSOLUTE_INPUTS)
        known_static: dict = {}
        statics: list[Site] = []
        for s in pre:
            found, cval = _constraint_value(constraint, s.addr)
            if found:
                sv, per = _value_rows(cval, s.dim)
                if per is not None:
                    raise NotImplementedError("ScanBootstrapFilter: observations are shared by the particles")
                known_static[s.addr] = np.broadcast_to(sv, (s.dim,)).astype(np.float32)
            else:
                statics.append(s)
        self._has_statics = bool(statics)
        if statics and (self.proposal is not None or self.rejuvenate):
            raise NotImplementedError("ScanBootstrapFilter: a model with latent sites in front of the Scan runs with the prior proposal and without moves")
        n_lat0 = sum(s.dim for s in steps[0] if not _constraint_value(constraint, s.addr)[0]) if steps else 0
        for t, cur in enumerate(steps):
            step_sl = SiteList()
            modes, obs = {}, {}
            if t == 0:
                # step 0 runs the sites in front of the Scan as they are; with statics its rows are laid out like a later step's —
                # [statics | as many unused rows as a step has carry rows | own] — so that ONE kernel serves steps 1 .. T-1
                for s in pre:
                    step_sl.sites.append(Site(s.addr, s.kind, list(s.params), s.dim, s.ncat, step_sl.n_slots, 0))
                    step_sl.n_slots += s.dim
                    if s.addr in known_static:
                        modes[s.addr] = A.MODE_OBS_TAB
                        obs[s.addr] = known_static[s.addr]
                if statics and n_lat0:
                    step_sl.sites.append(Site("@pad", A.NORMAL, [], n_lat0, 0, step_sl.n_slots, 0))
                    step_sl.n_slots += n_lat0
                    modes["@pad"] = A.MODE_INPUT
            else:
                for ps in statics:            # the statics first, then the carry: what this step may read, in ROW order of the step before
                    step_sl.sites.append(Site(ps.addr, ps.kind, [], ps.dim, 0, step_sl.n_slots, 0))
                    step_sl.n_slots += ps.dim
                    modes[ps.addr] = A.MODE_INPUT
            for ps in prev_latent:            # the carry: what this step may read of step t-1, in that step's ROW order
                w = ps.dim
                step_sl.sites.append(Site(ps.addr, ps.kind, [], w, 0, step_sl.n_slots, 0))
                step_sl.n_slots += w
                modes[ps.addr] = A.MODE_INPUT
            # the proposal's sites of this step: drawn, their log-density leaves the weight (GJX_SITE_PROPOSAL)
            q_here = {s.addr for s in q_steps[t]}
            q_names, proposed_by = [], {}
            kn = {**known_static, **known}

            def q_param(p):
                """a parameter of a proposal site: sources among the proposal's own sites of this step are renamed, the carry keeps
                the model's addresses (the INPUT sites)"""
                import dataclasses
                if p.op == A.P_CONST:
                    return p
                if p.op == A.P_EXPR:
                    from .. import expr as E
                    return dataclasses.replace(p, outs=tuple(E.rewrite_leaves(p.outs, lambda a_, e_: E.value(("@q", a_), e_) if a_ in q_here else None)))
                if p.terms:
                    return dataclasses.replace(p, terms=[((("@q", a_) if a_ in q_here else a_), m) for a_, m in p.terms],
                                               src=("@q", p.src) if p.src in q_here else p.src)
                if p.op == A.P_VGATHER and p.vsrc in q_here:
                    p = dataclasses.replace(p, vsrc=("@q", p.vsrc))
                return dataclasses.replace(p, src=("@q", p.src)) if p.src in q_here else p
            for s in q_steps[t]:
                rows = s.ncat if s.ncat else s.dim
                qa = ("@q", s.addr)
                ns = Site(qa, s.kind, [q_param(fold_known(p, kn, rows)) for p in s.params], s.dim, s.ncat, step_sl.n_slots, 0)
                for p in ns.params:
                    for a_ in p.sources():
                        if a_ not in step_sl:
                            raise NotImplementedError(f"ScanBootstrapFilter: proposal site {s.addr!r} reads {a_!r}, which is neither of this step nor of the one before")
                step_sl.sites.append(ns)
                step_sl.n_slots += s.dim
                q_names.append(qa)
            now_known, latent = {}, []
            for s in cur:
                rows = s.ncat if s.ncat else s.dim
                ns = Site(s.addr, s.kind, [fold_known(p, kn, rows) for p in s.params], s.dim, s.ncat, step_sl.n_slots, 0)
                for p in ns.params:
                    for a_ in p.sources():
                        if a_ not in step_sl:
                            raise NotImplementedError(f"ScanBootstrapFilter: site {s.addr!r} reads {a_!r}, which is neither of this step nor of the one before")
                step_sl.sites.append(ns)
                step_sl.n_slots += s.dim
                found, cval = _constraint_value(constraint, s.addr)
                if found:
                    if s.addr in q_here:
                        raise ValueError(f"ScanBootstrapFilter: {s.addr!r} is observed AND proposed")
                    sv, per = _value_rows(cval, s.dim)
                    if per is not None:
                        raise NotImplementedError("ScanBootstrapFilter: observations are shared by the particles")
                    modes[s.addr] = A.MODE_OBS_TAB
                    obs[s.addr] = now_known[s.addr] = np.broadcast_to(sv, (s.dim,)).astype(np.float32)
                else:
                    if s.addr in q_here:      # scored at the proposal's draw (its rows ARE the proposal site's)
                        modes[s.addr] = A.MODE_OBS_PROPOSED
                        proposed_by[s.addr] = ("@q", s.addr)
                    latent.append(s)
            missing = q_here - {s.addr for s in cur}
            if missing:
                raise ValueError(f"ScanBootstrapFilter: the proposal names {sorted(map(repr, missing))}, which the model's step does not have")
            extra = {}
            if statics and t > 0:
                # absolute rows of the step before: a static sits where THAT step keeps it (its own rows in step 0, its stored INPUT
                # rows later), the carry in that step's own rows
                prev_prog = progs[-1]
                extra = dict(carried=[ps.addr for ps in statics],
                             input_row_of={ps.addr: prev_prog.slot_of[ps.addr] for ps in list(statics) + list(prev_latent)})
            prog = PackedProgram(step_sl, modes, obs, rng_mode=self.rng_mode, plates=False, proposal=q_names, proposed_by=proposed_by, **extra)
            prog.filter_obs = dict(obs)       # (inference/filter_moves.py builds the step's assess form from them)
            progs.append(prog)
            # the next step reads this step's latent rows in ROW order (a proposed latent sits where its proposal site drew it)
            prev_latent = sorted(latent, key=lambda s_: prog.slot_of[s_.addr])
            known = now_known
        return progs

    def run(self, key: Key, constraint: ChoiceMap, args=(None, None), device=None, keep_ancestors: bool = False, keep_history: bool = False):
        """-> dict(log_ml, increments f32[T], choices f32[n_slots][K] of the last step (its INPUT rows unused), logw,
        programs, ancestors (the last resampling's, or int32[T-1][K] with keep_ancestors), degenerate).
        ``keep_history``: every step's choices and every resampling's ancestors are kept (gjx_scan_filter_history) and
        ``out["history"]`` (a ScanHistory) reconstructs trajectories from them."""
        from .. import kernels
        dev = kernels._dev(device)
        hmc_move = None
        if self.moves:
            from .filter_moves import normalise_moves, run_with_moves
            specs = normalise_moves(self.moves)
            if keep_history and not (len(specs) == 1 and specs[0][0] == "hmc"):
                raise NotImplementedError("filter moves: keep_history is kept by the library's own loops (no moves, the random-walk move, ONE HMC move)")
            # ONE HMC move: inside the library's own step loop (gjx_filter_opts::hmc_targets: gather, gjx_hmc, propagate per step, no
            # host between the launches); anything else — proposals, several moves — step by step from here
            if len(specs) == 1 and specs[0][0] == "hmc" and not getattr(self, "_moves_step_by_step", False):
                hmc_move = specs[0][1]
            else:
                return run_with_moves(self, key, constraint, args, self.moves, device=dev, keep_ancestors=keep_ancestors)
        # two keys: the STRUCTURE of the run (addresses, shapes, dtypes: what decides the site lists, hence the kernels) and its DATA
        # (every byte of the observations and of the kernel's arguments — both are folded into the step programs' tables; repr()
        # elides the middle of long arrays and must not be used here).  New data under an old structure re-fills the tables of the
        # cached programs (one upload); the programs, their ids and the library's per-program caches stay.
        sk, dk = _run_keys(constraint, args, dev)
        c = self._cache
        if c.get("sk") == sk and c.get("dk") != dk:
            fresh = self.step_programs(constraint, args)
            old = c["progs"]
            if len(fresh) == len(old) and all(a.sites_bytes() == b_.sites_bytes() and a.tab.size == b_.tab.size and a.n_slots == b_.n_slots
                                              for a, b_ in zip(fresh, old)):
                for a, b_ in zip(fresh, old):
                    b_.tab[:] = a.tab
                    b_.site_list, b_.modes, b_.filter_obs = a.site_list, a.modes, a.filter_obs
                _upload_tables(old, c["tabs_dev"])
                c["dk"] = dk
                c.pop("hmc", None)            # (the move's targets hold the observations too: rebuilt)
            else:
                c = self._cache = {}
        if c.get("sk") != sk or c.get("dk") != dk:
            progs = self.step_programs(constraint, args)
            tabs_dev = _bind_device(progs, dev)
            cps = (A.GjxProgram * len(progs))()
            for t, p in enumerate(progs):
                cps[t] = p.c_program(tabs_dev[0].device)
            bufs = c.get("bufs") if c.get("sk") == sk else None
            c = self._cache = dict(sk=sk, dk=dk, progs=progs, cps=cps, tabs_dev=tabs_dev)
            if bufs is not None:
                c["bufs"] = bufs
        progs, cps = self._cache["progs"], self._cache["cps"]
        T, K = len(progs), self.K
        n_rows = max(max(p.n_slots for p in progs), 1)
        f32 = torch.float32
        self._hmc_state = None
        if hmc_move is not None and T >= 2:
            hs = self._cache.get("hmc")
            if hs is None or hs["move"] is not hmc_move:
                from .filter_moves import hmc_targets
                tg = hmc_targets(progs, hmc_move)
                tdev = _bind_device(tg, dev)
                tcps = (A.GjxProgram * len(tg))()
                for t, p in enumerate(tg):
                    tcps[t] = p.c_program(tdev[0].device)
                wsb = max(int(load().gjx_hmc_workspace_bytes(C.byref(tcps[t]), K)) for t in range(len(tg)))
                hs = self._cache["hmc"] = dict(move=hmc_move, progs=tg, cps=tcps, tabs_dev=tdev, rows=torch.empty((n_rows, K), dtype=f32, device=dev),
                                               out=torch.empty((3, K), dtype=f32, device=dev), ws=torch.zeros(max(wsb, 256), dtype=torch.uint8, device=dev))
            self._hmc_state = hs
        b = self._cache.get("bufs")
        if b is None or b["rows_a"].shape != (n_rows, K):
            # two run workspaces and a second log-weight buffer: the one-launch step (resampling in the generated kernel's
            # prologue, include/gjx.h gjx_run_resample) alternates between them; OP_RUN + OP_RESAMPLE is the minimum
            need = 2 * load().gjx_workspace_bytes(A.OP_RUN, K) + load().gjx_workspace_bytes(A.OP_RESAMPLE, K) + 4 * K + 512
            # ... and behind them the area of the steps kernel (every step from the third in ONE launch: granules, pair arrays, the
            # per-step tables / keys / comb offsets); the size of the workspace handed over selects the form (include/gjx.h)
            self._ws_one_launch_per_step = need
            need += 192 * ((K + 1023) // 1024) + 32 * 4096 + 2048 + (8 * K + 512 if self.resampler == "multinomial" else 0)
            if getattr(self, "_minimal_workspace", False):        # (tests: the library then runs the two-launch step by itself)
                need = load().gjx_workspace_bytes(A.OP_RUN, K) + load().gjx_workspace_bytes(A.OP_RESAMPLE, K)
                self._ws_one_launch_per_step = need
            b = self._cache["bufs"] = dict(rows_a=torch.empty((n_rows, K), dtype=f32, device=dev), rows_b=torch.empty((n_rows, K), dtype=f32, device=dev),
                                           logw=torch.empty(K, dtype=f32, device=dev), anc=torch.empty(K, dtype=torch.int32, device=dev),
                                           ws=torch.zeros(need, dtype=torch.uint8, device=dev))
        lse = torch.empty((T, 4), dtype=f32, device=dev)
        ws_bytes = b["ws"].numel() if T <= 4096 else min(b["ws"].numel(), self._ws_one_launch_per_step)
        opts, info = self._opts(dev), A.GjxFilterInfo()
        if keep_history:
            rows_all = torch.empty((T, n_rows, K), dtype=f32, device=dev)
            anc_all = torch.empty((max(T - 1, 1), K), dtype=torch.int32, device=dev)
            check(load().gjx_scan_filter_history(C.cast(cps, C.c_void_p), T, key[0], key[1], K, kernels._ptr(rows_all), n_rows, kernels._ptr(b["logw"]),
                                                 kernels._ptr(anc_all), kernels._ptr(lse), kernels._ptr(b["ws"]), ws_bytes, kernels._stream(),
                                                 C.byref(opts), C.byref(info)), "gjx_scan_filter_history")
            self.last_info = dict(form=int(info.form), form_name=A.FILTER_FORM_NAMES[int(info.form)], launches=int(info.launches), grid=int(info.grid),
                                  tiles_per_block=int(info.tiles_per_block))
            st = self._status(b)
            if st & 1:
                return self._repeat_after_timeout(key, constraint, args, device, keep_ancestors, keep_history)
            incs = lse[:, 3]
            logw = self._out(b["logw"])
            moved = bool(self.rejuvenate and int(self.rejuvenate.get("n_moves", 0)) > 0) or self._hmc_state is not None
            hist = ScanHistory(progs, rows_all, anc_all[: T - 1], logw, moved=moved)
            return dict(log_ml=incs.sum(), increments=incs, lse_steps=lse, choices=rows_all[T - 1][: max(progs[-1].n_slots, 1)], logw=logw,
                        programs=progs, ancestors=anc_all[: T - 1], history=hist, degenerate=bool(st & 2), info=self.last_info,
                        accepted_total=(int(self._acc.item()) if moved and getattr(self, "_acc", None) is not None else None))
        anc_all = torch.empty((max(T - 1, 1), K), dtype=torch.int32, device=dev) if keep_ancestors else None
        check(load().gjx_scan_filter(C.cast(cps, C.c_void_p), T, key[0], key[1], K, kernels._ptr(b["rows_a"]), kernels._ptr(b["rows_b"]),
                                     kernels._ptr(b["logw"]), kernels._ptr(b["anc"]), kernels._ptr(anc_all), kernels._ptr(lse),
                                     kernels._ptr(b["ws"]), ws_bytes, kernels._stream(), C.byref(opts), C.byref(info)), "gjx_scan_filter")
        self.last_info = dict(form=int(info.form), form_name=A.FILTER_FORM_NAMES[int(info.form)], launches=int(info.launches), grid=int(info.grid),
                              tiles_per_block=int(info.tiles_per_block))
        st = self._status(b)
        if st & 1:
            return self._repeat_after_timeout(key, constraint, args, device, keep_ancestors, keep_history)
        incs = lse[:, 3]
        last = progs[-1]
        ch = (b["rows_b"] if (T - 1) & 1 else b["rows_a"])[: max(last.n_slots, 1)]
        return dict(log_ml=incs.sum(), increments=incs, lse_steps=lse, choices=self._out(ch), logw=self._out(b["logw"]), programs=progs,
                    ancestors=anc_all if keep_ancestors else self._out(b["anc"]), degenerate=bool(st & 2), info=self.last_info,
                    accepted_total=(int(self._acc.item()) if (self.rejuvenate or self._hmc_state) and getattr(self, "_acc", None) is not None else None),
                    accepted=([int(self._acc.item())] if self._hmc_state else None))

    def _repeat_after_timeout(self, key, constraint, args, device, keep_ancestors, keep_history):
        """GJX_STATUS_POLL_TIMEOUT: the one-launch kernel needs its whole grid resident and something else held compute units.  The
        same run, same keys, again: with one launch per step (bit-identical results) — or, with resample-move (which exists in the
        one-launch form only), the one-launch form once more after the device has drained; a second time-out is an error, never a
        silently different algorithm."""
        import warnings
        moved = bool(self.rejuvenate and int(self.rejuvenate.get("n_moves", 0)) > 0)
        if getattr(self, "_repeating", False):
            raise RuntimeError("generic filter: the one-launch filter kernel timed out twice waiting for its peer blocks; resample-move "
                               "(rejuvenate=...) needs the one-launch form — free the device of other kernels, or run without rejuvenation")
        if moved:
            warnings.warn("generic filter: the one-launch kernel timed out waiting for its peer blocks; the run is repeated once the device "
                          "is idle (resample-move has no per-step form)")
            torch.cuda.synchronize()
            self._repeating = True
            try:
                return self.run(key, constraint, args, device, keep_ancestors, keep_history)
            finally:
                self._repeating = False
        warnings.warn("generic filter: the steps kernel timed out waiting for its peer blocks (another kernel holds compute units); "
                      "the run was repeated with one launch per step")
        self._no_steps_kernel = True
        try:
            return self.run(key, constraint, args, device, keep_ancestors, keep_history)
        finally:
            self._no_steps_kernel = False

    def run_peer(self, ctx, key: Key, constraint: ChoiceMap, args=(None, None), want_ancestors: bool = False):
        """the same filter on a collection SHARDED over the ranks of a ``kernels.PeerContext`` (one process per GPU; ``self.K`` is the
        number of particles of THIS rank, the context's K_local; its rows >= the step programs' rows): gjx_scan_filter_peer —
        two launches per rank whatever T is, granules pushed and carry rows pulled through the peer-mapped windows, no host in the
        loop; ``resampler="multinomial"``: sorted uniforms over the whole sharded collection, no exchange beyond the systematic filter's.  Every rank calls it with the same key, observations and arguments.  -> dict(log_ml (global), increments, lse_steps,
        choices (this rank's part of the last step: a view of the window), logw, ancestors? (global indices), programs, info)"""
        if ctx.K != self.K:
            raise ValueError("run_peer: the filter's particle count must be the context's K_local")
        if self.moves or (self.rejuvenate and int(self.rejuvenate.get("n_moves", 0)) > 0):
            raise NotImplementedError("run_peer: the sharded filter kernel has no resample-move yet (rejuvenate=..., moves=...); run() on one GPU has")
        if getattr(self, "_has_statics", False):
            raise NotImplementedError("run_peer: a model with latent sites in front of the Scan runs on one GPU")
        dev = ctx.device
        sk, dk = _run_keys(constraint, args, dev)
        c = self._cache
        if c.get("sk") != sk or c.get("dk") != dk:
            progs = self.step_programs(constraint, args)
            tabs_dev = _bind_device(progs, dev)
            cps = (A.GjxProgram * len(progs))()
            for t, p in enumerate(progs):
                cps[t] = p.c_program(tabs_dev[0].device)
            c = self._cache = dict(sk=sk, dk=dk, progs=progs, cps=cps, tabs_dev=tabs_dev)
        progs, cps = c["progs"], c["cps"]
        if max(p.n_slots for p in progs) > ctx.nrows:
            raise ValueError(f"run_peer: the context has {ctx.nrows} rows, the step programs need {max(p.n_slots for p in progs)}")
        if c.get("prepared_for") is not ctx or c.get("prepared_sk") != sk:
            # first run of this structure on this context: kernels generated, compiled and loaded on EVERY rank, then a host barrier —
            # the ranks enter the filter together (compile times differ by seconds, a rank's poll budget is shorter)
            ctx.scan_filter_prepare(cps, len(progs), opts=self._peer_opts())
            c["prepared_for"], c["prepared_sk"] = ctx, sk
        o = ctx.scan_filter(cps, len(progs), key, want_ancestors=want_ancestors, opts=self._peer_opts())
        incs = o["lse_steps"][:, 3]
        return dict(log_ml=incs.sum(), increments=incs, lse_steps=o["lse_steps"], choices=o["rows"][: max(progs[-1].n_slots, 1)], logw=o["logw"],
                    ancestors=o["ancestors"], programs=progs, info=o["info"])

    def _peer_opts(self):
        """of the run's options, what the sharded kernel carries (gjx_scan_filter_peer_opts): the multinomial resampler"""
        if self.resampler != "multinomial":
            return None
        o = A.GjxFilterOpts()
        o.flags = A.FILTER_MULTINOMIAL
        return o

    def _opts(self, dev=None) -> "A.GjxFilterOpts":
        """the form of the run as ARGUMENTS of the call (the library reads no environment variable).  Attributes of the filter, or —
        for scripts and tests — these variables, read HERE: GJX_SCAN_FILTER_TWO_LAUNCH=1 (search launch + step launch),
        GJX_SCAN_FILTER_PERSISTENT=0 (no one-launch form), GJX_SCAN_FILTER_WIDE=0 (not the 16-wave filter kernel),
        GJX_CORESIDENT_BLOCKS=n (assume n resident blocks: the time-out path)"""
        import os
        o = A.GjxFilterOpts()
        fl = int(getattr(self, "flags", 0))
        if os.environ.get("GJX_SCAN_FILTER_TWO_LAUNCH", "0") not in ("", "0"):
            fl |= A.FILTER_TWO_LAUNCH
        if os.environ.get("GJX_SCAN_FILTER_PERSISTENT", "1") == "0" or getattr(self, "_no_steps_kernel", False):
            fl |= A.FILTER_NO_ONE_LAUNCH
        if os.environ.get("GJX_SCAN_FILTER_WIDE", "1") == "0":
            fl |= A.FILTER_NO_WIDE
        if self.resampler == "multinomial":
            fl |= A.FILTER_MULTINOMIAL
        if getattr(self, "_has_statics", False):
            fl |= A.FILTER_ABSOLUTE_INPUTS
        o.flags = fl
        o.coresident_blocks = int(os.environ.get("GJX_CORESIDENT_BLOCKS", "0") or 0)
        hs = getattr(self, "_hmc_state", None)
        if (self.rejuvenate and int(self.rejuvenate.get("n_moves", 0)) > 0) or hs is not None:
            if hs is None:
                o.n_moves, o.move_scale = int(self.rejuvenate["n_moves"]), float(self.rejuvenate.get("scale", 0.5))
            if dev is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            if getattr(self, "_acc", None) is None or self._acc.device != dev:       # the counter lives where the filter runs
                self._acc = torch.zeros(1, dtype=torch.int64, device=dev)
            o.accepted_total = self._acc.data_ptr()
        if hs is not None:
            self._acc.zero_()
            o.hmc_targets = C.cast(hs["cps"], C.c_void_p)
            o.hmc_eps, o.hmc_L = float(hs["move"].eps), int(hs["move"].L)
            o.hmc_rows, o.hmc_out = hs["rows"].data_ptr(), hs["out"].data_ptr()
            o.hmc_workspace, o.hmc_workspace_bytes = hs["ws"].data_ptr(), hs["ws"].numel()
        tl = getattr(self, "timeline", None)
        if tl is not None:
            o.timeline, o.timeline_bytes = tl.data_ptr(), tl.numel() * tl.element_size()
        return o

    def _out(self, t):
        """results leave as copies: the buffers of a filter are reused by its next run (``alias_outputs = True`` hands out the
        buffers themselves — valid until the next run — for callers that time the loop)"""
        return t if getattr(self, "alias_outputs", False) else t.clone()

    def _status(self, b) -> int:
        """status word of the run (one stream synchronisation; read and cleared): bit 0 GJX_STATUS_POLL_TIMEOUT — the steps kernel
        gave up waiting for its peer blocks —, bit 1 GJX_STATUS_ZERO_TOTAL — a collection without weight was resampled with
        identity ancestors (``degenerate``).  ``check_status = False`` skips the synchronisation (the library clears the word at the
        start of every run, so nothing stale can leak into the next one) and reports 0."""
        from .. import kernels
        if not getattr(self, "check_status", True):
            return 0
        off = load().gjx_workspace_bytes(A.OP_RUN, self.K)
        return int(kernels.workspace_status(b["ws"][off:], raise_on_error=False))

    def latent(self, out: dict, name) -> torch.Tensor:
        """rows of the last step's choice ``name`` — or of the latent site ``name`` in FRONT of the Scan as the last step's particles
        carry it (their weights are out["logw"]): f32[dim][K]"""
        p = out["programs"][-1]
        for own in (True, False):
            for s in p.site_list.sites:
                is_in = p.modes.get(s.addr) == A.MODE_INPUT
                carried = is_in and bool(p.c_sites[p.site_list._pos(s.addr)].flags & A.SITE_CARRIED)
                if _name(s.addr) == name and p.slot_of[s.addr] >= 0 and ((own and not is_in) or (not own and carried)):
                    return out["choices"][p.slot_of[s.addr]: p.slot_of[s.addr] + s.dim]
        raise KeyError(name)


class ScanHistory:
    """Record of a filter run over a Scan: the choices of every step (SoA rows of the step's program) and the ancestors of every
    resampling.  The reference's ScanTrace stacks the whole trace per particle and a resampling gathers whole traces
    (scan.py:56-97); here the trajectory that ends in particle i of the last step is read back lazily: i_{t-1} =
    ancestors[t-1][i_t], one row gather per step."""

    def __init__(self, programs, rows_all, ancestors, logw, moved: bool = False):
        self.programs, self.rows_all, self.ancestors, self.logw = programs, rows_all, ancestors, logw
        # resample-move (gjx_filter_opts.n_moves > 0): the parent a particle of step t+1 was propagated from is the MOVED carry the
        # kernel stored in step t+1's INPUT rows (at the child's index), not the own rows of step t at the ancestor's index
        self.moved = bool(moved)

    def __len__(self):
        return len(self.programs)

    def _rows_of(self, t, name):
        p = self.programs[t]
        for s in p.site_list.sites:
            if _name(s.addr) == name and p.modes.get(s.addr) != A.MODE_INPUT and p.slot_of[s.addr] >= 0:
                return p.slot_of[s.addr], s.dim
        for s in p.site_list.sites:      # a static in front of the Scan: the step's stored (GJX_SITE_CARRIED) input rows
            if _name(s.addr) == name and p.slot_of[s.addr] >= 0 and (p.c_sites[p.site_list._pos(s.addr)].flags & A.SITE_CARRIED):
                return p.slot_of[s.addr], s.dim
        raise KeyError(name)

    def _input_rows_of(self, t, name):
        """rows of step t's INPUT site that stands for choice ``name`` of step t-1 (the carry as the step received it), or None"""
        p = self.programs[t]
        for s in p.site_list.sites:
            if _name(s.addr) == name and p.modes.get(s.addr) == A.MODE_INPUT and p.slot_of[s.addr] >= 0:
                return p.slot_of[s.addr], s.dim
        return None

    def step(self, t, name) -> torch.Tensor:
        """the particles of step t as propagated (before the next resampling and its move): f32[dim][K]"""
        r0, d = self._rows_of(t, name)
        return self.rows_all[t, r0:r0 + d]

    def paths(self, name, idx=None) -> torch.Tensor:
        """-> f32[T][dim][n]: the trajectories of choice ``name`` that end in particles ``idx`` (default: all) of the last step.
        With resample-move, x_t (t < T-1) of a trajectory is the value its child at step t+1 was propagated FROM — the moved carry in
        step t+1's INPUT rows at the child's index — so that every (x_t, x_{t+1}) pair of a path is a transition the filter made."""
        T = len(self.programs)
        K = self.rows_all.shape[2]
        cur = torch.arange(K, device=self.rows_all.device) if idx is None else idx.to(torch.int64)
        out = [None] * T
        child = None
        for t in range(T - 1, -1, -1):
            rin = self._input_rows_of(t + 1, name) if (self.moved and child is not None) else None
            if rin is not None:
                out[t] = self.rows_all[t + 1, rin[0]:rin[0] + rin[1]][:, child]
            else:
                if self.moved and child is not None:
                    raise NotImplementedError(f"ScanHistory.paths({name!r}) with resample-move: step {t + 1} does not receive {name!r} as carry")
                out[t] = self.step(t, name)[:, cur]
            if t > 0:
                child = cur
                cur = self.ancestors[t - 1][cur].to(torch.int64)
        return torch.stack(out)

    def smoothed_means(self, name) -> torch.Tensor:
        """E[x_t | y_{1:T}] from the reconstructed trajectories, weighted by the last step's weights: f32[T][dim]"""
        w = torch.softmax(self.logw.double(), dim=0)
        return (self.paths(name).double() * w).sum(dim=2).float()


def _name(addr):
    return addr[0] if isinstance(addr, tuple) and len(addr) == 2 and isinstance(addr[1], (int, np.integer)) else addr


def _leaves(x):
    if isinstance(x, (tuple, list)):
        for e in x:
            yield from _leaves(e)
    elif isinstance(x, dict):
        for k in sorted(x, key=repr):
            yield from _leaves(x[k])
    else:
        yield x


def _run_keys(constraint, args, dev):
    """-> (structure key, data key) of a filter run: see ScanBootstrapFilter.run"""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    shape = []
    for a, v in sorted(constraint._d.items(), key=lambda kv: repr(kv[0])):
        arr = np.ascontiguousarray(kernels_np(v))
        shape.append((repr(a), arr.shape, str(arr.dtype)))
        h.update(arr.tobytes())
    for leaf in _leaves(args):
        if leaf is None:
            shape.append(None)
            continue
        arr = np.ascontiguousarray(kernels_np(leaf))
        shape.append((arr.shape, str(arr.dtype)))
        h.update(arr.tobytes())
    return (tuple(shape), str(dev)), h.digest()


def _bind_device(progs, dev):
    """the site lists and the float tables of all step programs in TWO device tensors (two uploads instead of two per step); every
    program's device pointers are views (16-byte aligned).  -> the table tensor and the offsets, for later re-fills"""
    so, to, ns, nt_ = [], [], 0, 0
    for p in progs:
        so.append(ns)
        ns += (len(p.sites_bytes() or b"\0" * 96) + 15) & ~15
        to.append(nt_)
        nt_ += (p.tab.size + 3) & ~3
    sb = np.zeros(max(ns, 16), np.uint8)
    tb = np.zeros(max(nt_, 4), np.float32)
    for p, o1, o2 in zip(progs, so, to):
        b = p.sites_bytes() or b"\0" * 96
        sb[o1:o1 + len(b)] = np.frombuffer(b, np.uint8)
        tb[o2:o2 + p.tab.size] = p.tab
    sd, td = torch.from_numpy(sb).to(dev), torch.from_numpy(tb).to(dev)
    for p, o1, o2 in zip(progs, so, to):
        p._dev = (sd[o1:o1 + max(len(p.sites_bytes()), 96)], td[o2:o2 + p.tab.size])
        p._aux = None
    return td, to


def _upload_tables(progs, tabs_dev):
    td, to = tabs_dev
    tb = np.zeros(td.numel(), np.float32)
    for p, o2 in zip(progs, to):
        tb[o2:o2 + p.tab.size] = p.tab
    td.copy_(torch.from_numpy(tb))
    for p in progs:
        p._aux = None                                # derived constants follow the table


def kernels_np(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
