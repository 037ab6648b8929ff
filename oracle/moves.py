"""Restatement of the generic filter's resample-move (include/gjx.h gjx_filter_opts::n_moves).  TEST INFRASTRUCTURE ONLY.

The reference's ingredients: Rejuvenate (inference/requests/rejuvenate.py:70-94: propose, Update, weight the reverse proposal) with a
symmetric random-walk proposal — whose two proposal densities cancel — and the caller-side accept log u < w of
tests/inference/test_requests.py:131-137.  Here, as in the device kernel: behind the resampling in front of step t >= 2, particle i's
gathered carry x (the latent choices of step t-1) takes n_moves Metropolis steps; the target is the joint density of step t-1's model
sites given ITS inputs (what the ancestor was propagated from) — evaluated by the C oracle's assess of the step program with every
latent constrained per particle.  Streams: site 1022 of the step's propagation key; with R' = R (continuous carry rows) rounded up to
even, move n draws elements n (R' + 2) + c for the c-th continuous row and n (R' + 2) + R' for the accept's uniform
(uniform_from_bits(bits, tiny, 1)) — the accept never shares an element with the Box-Muller partner of a normal."""
from __future__ import annotations

import ctypes as C

import numpy as np

from genjax_amd import _abi as A
from genjax_amd.program import PackedProgram

from . import cpu

DISCRETE = (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS, A.FLIP, A.BERNOULLI_LOGITS, A.POISSON, A.GEOMETRIC)
MOVE_SITE = 1022


def _stream(rng_mode, key, gidx0, n_particles, e0, n, bits=False):
    L = cpu.lib()
    out = np.zeros((n_particles, n), np.uint32 if bits else np.float32)
    f = L.gjxo_stream_bits if bits else L.gjxo_stream_normals
    f.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p]
    for i in range(n_particles):
        row = np.zeros(n, out.dtype)
        f(int(rng_mode), int(key[0]), int(key[1]), int(gidx0 + i), MOVE_SITE, int(e0), int(n), row.ctypes.data_as(C.c_void_p))
        out[i] = row
    return out


def assess_program(step: PackedProgram) -> PackedProgram:
    """the step program with every latent choice constrained per particle and its proposal sites dropped: its `weight` is the joint
    log-density of the model's sites at the given values"""
    sl = step.site_list
    from genjax_amd.program import SiteList
    keep = SiteList()
    modes, obs = {}, {}
    for s in sl.sites:
        if isinstance(s.addr, tuple) and len(s.addr) == 2 and s.addr[0] == "@q":
            continue
        keep.sites.append(s)
        m = step.modes.get(s.addr, A.MODE_SAMPLE)
        if m == A.MODE_INPUT:
            modes[s.addr] = A.MODE_INPUT
        elif m == A.MODE_OBS_TAB:
            modes[s.addr] = A.MODE_OBS_TAB
            obs[s.addr] = step._obs_value(s.addr).copy()
        else:
            modes[s.addr] = A.MODE_OBS_SLOT
    keep.n_slots = sl.n_slots
    return PackedProgram(keep, modes, obs, rng_mode=step.rng_mode, plates=False)


def rw_metropolis_move(prev_step: PackedProgram, key_t, x, pin, n_moves: int, scale: float, gidx0: int = 0):
    """x f32[R][K]: the gathered carry (latent rows of step t-1, in INPUT-row order of step t); pin f32[R_in][K]: the inputs step t-1
    itself was propagated from (gathered through the same ancestors); key_t: the propagation key of step t.
    -> (moved x, accepted moves per particle)"""
    ap = assess_program(prev_step)
    K = x.shape[1]
    in_sites = [s for s in ap.site_list.sites if ap.modes.get(s.addr) == A.MODE_INPUT]
    lat_sites = [s for s in ap.site_list.sites if ap.modes.get(s.addr) == A.MODE_OBS_SLOT]
    lat_sites.sort(key=lambda s: ap.slot_of[s.addr])
    cont = np.concatenate([np.full(s.dim, s.kind not in DISCRETE) for s in lat_sites]) if lat_sites else np.zeros(0, bool)
    R = int(cont.sum())
    Rp = R + (R & 1)

    def logpi(xv):
        ch = np.zeros((max(ap.n_slots, 1), K), np.float32)
        r = 0
        for s in in_sites:
            ch[ap.slot_of[s.addr]: ap.slot_of[s.addr] + s.dim] = pin[r:r + s.dim]
            r += s.dim
        r = 0
        for s in lat_sites:
            ch[ap.slot_of[s.addr]: ap.slot_of[s.addr] + s.dim] = xv[r:r + s.dim]
            r += s.dim
        return cpu.run_program(ap, (0, 0), K, choices=ch)["weight"].astype(np.float32)

    x = x.astype(np.float32).copy()
    cur = logpi(x)
    nacc = np.zeros(K, np.int64)
    for n in range(n_moves):
        z = _stream(prev_step.rng_mode, key_t, gidx0, K, n * (Rp + 2), R)            # [K][R]
        ub = _stream(prev_step.rng_mode, key_t, gidx0, K, n * (Rp + 2) + Rp, 1, bits=True)[:, 0]
        xq = x.copy()
        xq[cont] = (np.float32(scale) * z.T + x[cont]).astype(np.float32)
        prop = logpi(xq)
        tiny = np.float32(1.17549435e-38)
        unit = (((ub >> 9) | 0x3F800000).astype(np.uint32).view(np.float32) - np.float32(1.0))
        u = np.maximum(tiny, unit * (np.float32(1.0) - tiny) + tiny)
        acc = np.log(u.astype(np.float64)) < (prop.astype(np.float64) - cur.astype(np.float64))
        x[:, acc] = xq[:, acc]
        cur = np.where(acc, prop, cur)
        nacc += acc
    return x, nacc
