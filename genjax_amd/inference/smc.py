"""Importance sampling / SMC drivers with the reference's class names (inference/smc.py, sp.py).

Host code here only does what the reference does at trace time: key splits, constraint merging,
choosing site modes.  Every per-particle operation is one launch of the HIP kernels:
  run_smc            -> gjx_run_program (propagate + reweight + fused LSE partials)   smc.py:298-315
  log-ML estimate    -> 4-float device result of the LSE finish                       smc.py:96-97
  sample_particle    -> gjx_categorical_pick (Gumbel-max argmax reduce) + column read smc.py:102-109
  ChangeTarget       -> gjx_run_program with every site constrained                   smc.py:370-396
"""
from __future__ import annotations

from typing import Any

import numpy as np

from ..core import ChoiceMap, Key, split
from ..gen import GenerativeFunction, Marginal, Trace


class Target:
    """Unnormalised posterior: generative function + args + constraint (sp.py:52-94)."""

    def __init__(self, p: GenerativeFunction, args: tuple, constraint: ChoiceMap):
        if isinstance(p, Marginal):
            raise TypeError("Target does not support Marginal generative functions.")  # sp.py:46-49
        if not isinstance(p, GenerativeFunction):
            raise TypeError("Target needs a generative function")
        self.p, self.args, self.constraint = p, tuple(args), constraint

    def importance(self, key: Key, constraint: ChoiceMap, K: int | None = None):
        merged = self.constraint.merge(constraint)
        return self.p.importance(key, merged, self.args, K)

    def filter_to_unconstrained(self, choice_map: ChoiceMap) -> ChoiceMap:
        return choice_map.filter(~self.constraint.get_selection())

    def __getitem__(self, addr):
        return self.constraint[addr]

    def same_as(self, other: "Target") -> bool:
        from ..gen import _args_key
        return self.p is other.p and _args_key(self.args) == _args_key(other.args) and self.constraint == other.constraint

    def _cache_key(self):
        """Content identity for trace caches (a proposal ``q(target)`` is traced once per distinct target)."""
        from ..gen import _IdKey, _args_key, _value_key
        items = []
        from ..core import Masked
        for addr, v in self.constraint._d.items():
            items.append((repr(addr), (_value_key(v.value), _value_key(v.flag)) if isinstance(v, Masked) else _value_key(v)))
        return ("Target", _IdKey(self.p), _args_key(self.args), tuple(items))


class ParticleCollection:
    """Weighted particles (smc.py:76-109): a batched Trace + log-weights, all on the device."""

    def __init__(self, particles: Trace, log_weights, is_valid=True, lse=None, offset: int = 0, K_total=None, partials=None):
        self.particles, self.log_weights, self.is_valid = particles, log_weights, is_valid
        self._lse = lse            # device f32[4] = {max, sumexp, lse, lse - log K} when already reduced
        self._partials = partials  # kernels.RunPartials of the producing run (block {max, sumexp} pairs), or None
        self.offset = offset
        self.K_total = K_total or int(log_weights.shape[0])

    def get_particles(self) -> Trace:
        return self.particles

    def get_particle(self, idx) -> Trace:
        return self.particles.get_particle(idx)

    def get_log_weights(self):
        return self.log_weights

    def __len__(self):
        return int(self.log_weights.shape[0])

    def __getitem__(self, idx):
        return self.get_particle(idx), self.log_weights[idx]

    def lse(self):
        """The LSE record, reduced on first use: from the producing run's block partials while they are still in its
        workspace (one small launch), otherwise from the log-weights."""
        if self._lse is None:
            from .. import kernels
            p = self._partials
            if p is not None and p.valid():
                self._lse = p.finish(self.K_total)
            else:
                self._lse = kernels.logsumexp(self.log_weights, self.K_total)
        return self._lse

    def lse_partials(self):
        """kernels.RunPartials of the producing run if still valid (for ``inference.pf.resample(..., collection=)``)"""
        p = self._partials
        return p if (self._lse is None and p is not None and p.valid()) else None

    def get_log_marginal_likelihood_estimate(self):
        """logsumexp(log_weights) - log K  (smc.py:96-97), a 0-d device tensor."""
        return self.lse()[3]

    def sample_particle(self, key: Key) -> Trace:
        """1-of-K categorical draw over the weights + gather (smc.py:102-109)."""
        from .. import kernels
        out = kernels.categorical_pick(self.log_weights, self.lse(), key, self.particles.prog.rng_mode, self.offset)
        idx = int(out[1].item()) - self.offset
        return self.get_particle(idx)


class TrialCollections:
    """n independent particle collections of K particles each, produced by ONE launch over n K particles: what the reference
    gets from ``jax.vmap(alg.run_smc)(jax.random.split(key, n))`` (README.md:108-113).  Trial t owns the global particle
    indices [t K, (t + 1) K); its weights are those of ``alg.run_smc(key, offset=t K, K_local=K)`` of an n K-particle run,
    normalised on their own."""

    def __init__(self, particles: Trace, log_weights, n_trials: int, K: int):
        self.particles, self.log_weights, self.n_trials, self.K = particles, log_weights, int(n_trials), int(K)
        self._lse = None

    def __len__(self):
        return self.n_trials

    def lse(self):
        """f32[n_trials][4] = {max, sumexp, lse, lse - log K} per trial (one launch, gjx_trials_lse_pick)"""
        if self._lse is None:
            from .. import kernels
            self._lse, _ = kernels.trials_lse_pick(self.log_weights, self.n_trials, self.K, None, self.particles.prog.rng_mode)
        return self._lse

    def get_log_weights(self):
        return self.log_weights.view(self.n_trials, self.K)

    def get_log_marginal_likelihood_estimates(self):
        """f32[n_trials]: logsumexp(log_weights of the trial) - log K  (smc.py:96-97 per trial)"""
        return self.lse()[:, 3]

    def trial(self, t: int) -> ParticleCollection:
        """trial t as a ParticleCollection (a copy of its columns)"""
        a, b = t * self.K, (t + 1) * self.K
        tr = self.particles
        sub = Trace(tr.gen_fn, tr.args, tr.prog, tr.choices[:, a:b].contiguous(), tr.score[a:b].contiguous(), tr.shared, True, tr.retval_sym)
        return ParticleCollection(sub, self.log_weights[a:b].contiguous(), True, self.lse()[t], a, self.K)

    def sample_particles(self, key: Key) -> Trace:
        """one particle per trial, drawn 1-of-K over the trial's weights (smc.py:102-109 per trial): a batched Trace of
        n_trials particles"""
        from .. import kernels
        tr = self.particles
        self._lse, pick = kernels.trials_lse_pick(self.log_weights, self.n_trials, self.K, key, tr.prog.rng_mode)
        ch = kernels.gather_rows(tr.choices, pick)
        sc = kernels.gather_rows(tr.score.reshape(1, -1), pick).reshape(-1)
        return Trace(tr.gen_fn, tr.args, tr.prog, ch, sc, tr.shared, True, tr.retval_sym)


class SMCAlgorithm:
    """smc.py:117-225"""

    def get_num_particles(self) -> int:
        raise NotImplementedError

    def get_final_target(self) -> Target:
        raise NotImplementedError

    def run_smc(self, key: Key) -> ParticleCollection:
        raise NotImplementedError

    def run_csmc(self, key: Key, retained: ChoiceMap) -> ParticleCollection:
        raise NotImplementedError

    def log_marginal_likelihood_estimate(self, key: Key, target: Target | None = None):
        algorithm = ChangeTarget(self, target) if target else self
        key, sub_key = split(key)
        return algorithm.run_smc(sub_key).get_log_marginal_likelihood_estimate()

    # GenSP interface
    def random_weighted(self, key: Key, *args: Any):
        target = args[0]
        assert isinstance(target, Target)
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        pc = algorithm.run_smc(key)
        particle = pc.sample_particle(sub_key)
        log_density_estimate = particle.get_score() - pc.get_log_marginal_likelihood_estimate()
        chm = target.filter_to_unconstrained(particle.get_choices())
        return log_density_estimate, chm

    def estimate_logpdf(self, key: Key, v: ChoiceMap, *args: Any):
        target = args[0]
        assert isinstance(target, Target)
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        pc = algorithm.run_csmc(key, v)
        particle = pc.sample_particle(sub_key)
        return particle.get_score() - pc.get_log_marginal_likelihood_estimate()

    def estimate_normalizing_constant(self, key: Key, target: Target):
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        return algorithm.run_smc(sub_key).get_log_marginal_likelihood_estimate()

    def estimate_reciprocal_normalizing_constant(self, key: Key, target: Target, latent_choices: ChoiceMap, w):
        """smc.py:213-225: ``w`` with ``latent_choices`` is already properly weighted for ``target``, so the retained
        particle skips the reweighting step (ChangeTarget.run_csmc_for_normalizing_constant)."""
        return ChangeTarget(self, target).run_csmc_for_normalizing_constant(key, latent_choices, w)

    def simulate(self, key: Key, args: tuple):
        """Distribution.simulate of an Algorithm (distribution.py:108-115): (score, choices) as a pair."""
        w, chm = self.random_weighted(key, *args)
        return w, chm


def _propose(q, key: Key, target: Target, K: int, offset: int = 0, K_total=None):
    """q.random_weighted vmapped over particles: -> (log_q f32[K], per-particle rows addr -> [dim][K]).
    Properly weighted (log_q = the proposal's full density), unlike the reference's
    Marginal.random_weighted path (sp.py:226-230).  ``offset`` is the global index of this
    rank's first particle: a shard proposes from the same streams the unsharded run would use."""
    gf = q.gen_fn if isinstance(q, Marginal) else q
    tr, out = gf._run(key, K, (target,), ChoiceMap.empty(), True, True, offset=offset, K_total=K_total)
    return out["score"], {a: r for a, r in tr.full_choice_rows().items()}


class ImportanceK(SMCAlgorithm):
    """K-particle importance sampling (smc.py:282-351)."""

    def __init__(self, target: Target, q=None, k_particles: int = 2):
        self.target, self.q, self.k_particles = target, q, int(k_particles)

    def get_num_particles(self):
        return self.k_particles

    def get_final_target(self):
        return self.target

    def run_smc(self, key: Key, offset: int = 0, K_local: int | None = None) -> ParticleCollection:
        """``offset``/``K_local`` select this rank's shard of the K particles (global indices
        [offset, offset + K_local)); the default is the whole collection."""
        K = self.k_particles if K_local is None else int(K_local)
        key, sub_key = split(key)                    # smc.py:299
        # sub_keys = split(sub_key, K) happens on the device: particle i <-> Threefry(sub_key, (0, i))
        if self.q is not None:
            log_q, rows = _propose(self.q, sub_key, self.target, K, offset, self.k_particles)     # smc.py:302-304
            tr, out = self.target.p._run(sub_key, K, self.target.args, self.target.constraint, True, True,
                                         prev_rows=rows, sub=log_q, want_lse=True, offset=offset,
                                         K_total=self.k_particles)
        else:
            # no LSE tail in the producing kernel: its block partials stay in the workspace and are reduced by whoever
            # needs the record first (the resampler's prologue, or one small launch)
            tr, out = self.target.p._run(sub_key, K, self.target.args, self.target.constraint, True, True,
                                         want_lse=False, offset=offset, K_total=self.k_particles)
        return ParticleCollection(tr, out["logw"], True, out["lse"], offset, self.k_particles, partials=out.get("_partials"))

    def run_smc_trials(self, key: Key, n_trials: int) -> TrialCollections:
        """``n_trials`` independent runs of this algorithm in one launch — the device form of the reference idiom
        ``jax.vmap(alg.run_smc)(jax.random.split(key, n_trials))`` (README.md:108-113).  The trials are the consecutive
        K-particle shards of ONE n_trials * K-particle run under ``key`` (counter-based streams: disjoint particle
        indices are independent), each normalised on its own."""
        n, K = int(n_trials), self.k_particles
        key, sub_key = split(key)
        if self.q is not None:
            log_q, rows = _propose(self.q, sub_key, self.target, n * K, 0, n * K)
            tr, out = self.target.p._run(sub_key, n * K, self.target.args, self.target.constraint, True, True,
                                         prev_rows=rows, sub=log_q, want_lse=False, K_total=n * K)
        else:
            tr, out = self.target.p._run(sub_key, n * K, self.target.args, self.target.constraint, True, True,
                                         want_lse=False, K_total=n * K)
        return TrialCollections(tr, out["logw"], n, K)

    def random_weighted_trials(self, key: Key, n_trials: int, *args: Any):
        """``jax.vmap(alg.random_weighted, in_axes=(0, None))(jax.random.split(key, n_trials), target)`` (README.md:110-113)
        in three launches: -> (log-density estimates f32[n_trials], ChoiceMap of the unconstrained choices with a leading
        trial axis)."""
        target = args[0]
        assert isinstance(target, Target)
        key, sub_key = split(key)
        tc = self.run_smc_trials(key, n_trials)
        if not target.same_as(self.target):
            pc = ChangeTarget(self, target)._reweight(key, ParticleCollection(tc.particles, tc.log_weights, True))
            tc = TrialCollections(pc.get_particles(), pc.get_log_weights(), tc.n_trials, tc.K)
        picked = tc.sample_particles(sub_key)
        est = picked.get_score() - tc.get_log_marginal_likelihood_estimates()
        return est, target.filter_to_unconstrained(picked.get_choices())

    def run_csmc(self, key: Key, retained: ChoiceMap) -> ParticleCollection:
        """K-1 fresh particles plus the retained choice map stacked last (smc.py:317-351)."""
        import torch
        K = self.k_particles
        key, sub_key = split(key)
        tgt = self.target
        if self.q is not None:
            log_q, rows = _propose(self.q, sub_key, tgt, K - 1) if K > 1 else (None, {})
            gf = self.q.gen_fn if isinstance(self.q, Marginal) else self.q
            ret_score, _ = gf.assess(retained, (tgt,))
            sl, _ = gf.site_list((tgt,))
            stacked = {}
            for s in sl.sites:
                rv = torch.as_tensor(np.asarray(_host(retained[s.addr]), np.float32).reshape(s.dim, 1), device=ret_score.device)
                stacked[s.addr] = torch.cat([rows[s.addr], rv], dim=1) if K > 1 else rv
            sub = torch.cat([log_q, ret_score.reshape(1)]) if K > 1 else ret_score.reshape(1)
            tr, out = tgt.p._run(key, K, tgt.args, tgt.constraint, True, True, prev_rows=stacked, sub=sub, want_lse=True)
            return ParticleCollection(tr, out["logw"], True, out["lse"])
        # no proposal: K-1 fresh particles from the prior; the retained choices enter as per-particle
        # constraints of a 1-particle run, so both runs share one slot layout and are stacked column-wise
        ret_rows = {}
        sl, _ = tgt.p.site_list(tgt.args)
        for s in sl.sites:
            if s.addr in retained:
                ret_rows[s.addr] = torch.as_tensor(np.asarray(_host(retained[s.addr]), np.float32).reshape(s.dim, 1))
        tr_r, out_r = tgt.p._run(key, 1, tgt.args, tgt.constraint, True, True, prev_rows=ret_rows)
        if K == 1:
            return ParticleCollection(tr_r, out_r["weight"], True)
        tr_f, out_f = tgt.p._run(sub_key, K - 1, tgt.args, tgt.constraint, True, True)
        pf = {a: r for a, r in tr_f.full_choice_rows().items() if tr_r.prog.slot_of[a] >= 0}
        ch = torch.cat([torch.cat([pf[s.addr] for s in sl.sites if tr_r.prog.slot_of[s.addr] >= 0], dim=0), tr_r.choices], dim=1)
        tr = Trace(tgt.p, tgt.args, tr_r.prog, ch.contiguous(), torch.cat([tr_f.score, tr_r.score]), tr_r.shared, True,
                   tr_r.retval_sym)
        return ParticleCollection(tr, torch.cat([out_f["weight"], out_r["weight"]]), True)


class Importance(ImportanceK):
    """1-particle special case (smc.py:233-279)."""

    def __init__(self, target: Target, q=None):
        super().__init__(target, q, 1)


class ChangeTarget(SMCAlgorithm):
    """Reweight an existing collection to a new target (smc.py:359-465)."""

    def __init__(self, prev: SMCAlgorithm, target: Target):
        self.prev, self.target = prev, target

    def get_num_particles(self):
        return self.prev.get_num_particles()

    def get_final_target(self):
        return self.target

    def _reweight(self, key: Key, collection: ParticleCollection) -> ParticleCollection:
        prev_target = self.prev.get_final_target()
        particles = collection.get_particles()
        if self.target.same_as(prev_target):
            # every site would be re-assessed to the same score: this_weight == weight
            return collection
        rows = particles.full_choice_rows()
        latents = {a: r for a, r in rows.items() if a not in prev_target.constraint}   # sp.py:89-91
        K = particles.K
        tr, out = self.target.p._run(key, K, self.target.args, self.target.constraint, True, True, prev_rows=latents,
                                     logw_in=collection.get_log_weights(), sub=particles.score, want_lse=True)
        return ParticleCollection(tr, out["logw"], True, out["lse"])          # smc.py:383: w' - score + w

    def run_smc(self, key: Key) -> ParticleCollection:
        return self._reweight(key, self.prev.run_smc(key))

    def run_csmc(self, key: Key, retained: ChoiceMap) -> ParticleCollection:
        return self._reweight(key, self.prev.run_csmc(key, retained))

    def run_csmc_for_normalizing_constant(self, key: Key, latent_choices: ChoiceMap, w):
        """smc.py:432-465: reciprocal-normalising-constant estimate used by the variational interface — the K-1 fresh
        particles are reweighted to this target, the retained one keeps the caller's weight ``w``."""
        import math
        import torch
        from .. import kernels
        from ..core import split
        key, sub_key = split(key)
        coll = self.prev.run_csmc(sub_key, latent_choices)
        lw = self._reweight(key, coll).get_log_weights().clone()
        retained_score = coll.get_particles().score[-1]
        retained_weight = coll.get_log_weights()[-1]
        lw[-1] = torch.as_tensor(w, dtype=lw.dtype, device=lw.device) - retained_score + retained_weight
        total = kernels.logsumexp(lw.contiguous())[2]
        return retained_score - (total - math.log(self.get_num_particles()))


def _host(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else v
