"""ctypes wrapper of the C oracle (oracle/gjx_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package ``genjax_amd`` never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from genjax_amd import _abi as A
from genjax_amd.program import PackedProgram

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgjx_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gjx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgjx_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_checker_lib = None
FAST_FLAGS = ["-O3", "-march=native", "-fPIC", "-std=c11", "-fopenmp", "-fno-math-errno"]
build_flags = "-O2 -ffp-contract=off (the checker build, oracle/Makefile)"


def use_fast_build() -> str:
    """bench.py's cpu_baseline leg ONLY: time the oracle compiled for speed on THIS host (-O3 -march=native: SURVEY.md §8(d)) instead
    of the checker build (-O2 -ffp-contract=off, which the parity tests use).  Compiled into a temporary directory at run time — a
    -march=native object built elsewhere could not be trusted to run here.  Falls back to the checker build if no compiler is
    there.  -> the flags of the build now in use"""
    global _lib, _checker_lib, build_flags
    import tempfile
    if _checker_lib is None:
        _checker_lib = lib()
    src = os.path.join(_HERE, "gjx_oracle.c")
    out = os.path.join(tempfile.mkdtemp(prefix="gjx_oracle_fast_"), "libgjx_oracle_fast.so")
    try:
        subprocess.check_call([os.environ.get("CC", "gcc")] + FAST_FLAGS + ["-I", os.path.join(_HERE, "..", "include"), "-shared", "-o", out, src, "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = _bind(C.CDLL(out))
        build_flags = " ".join(FAST_FLAGS) + " (compiled on this host for the timed baseline)"
    except Exception:
        _lib = _checker_lib
        build_flags = "-O2 -ffp-contract=off (the checker build: the fast build could not be compiled here)"
    return build_flags


def use_checker_build() -> None:
    global _lib, build_flags
    if _checker_lib is not None:
        _lib = _checker_lib
        build_flags = "-O2 -ffp-contract=off (the checker build, oracle/Makefile)"


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    vp, i32, i64, u32, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double
    L.gjxo_threefry2x32.argtypes = [u32, u32, u32, u32, vp]
    L.gjxo_run_program.argtypes = [A.PP, u32, u32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64]
    L.gjxo_logsumexp.argtypes = [vp, i64, i64, vp]
    L.gjxo_categorical_pick.argtypes = [vp, i64, i64, vp, u32, u32, i32, vp, vp]
    L.gjxo_weight_cumsum.argtypes = [vp, i64, i32, vp, vp, vp]
    L.gjxo_resample_systematic.argtypes = [vp, i64, u64, u64, f64, i64, i64, i64, vp]
    L.gjxo_resample_multinomial.argtypes = [vp, i64, u64, u64, u32, u32, i64, i64, i64, vp]
    L.gjxo_resample_systematic_tiled.argtypes = [vp, i64, f64, i64, vp, vp, vp, vp]
    L.gjxo_resample_sorted_multinomial_tiled.argtypes = [vp, i64, u32, u32, i64, vp, vp]
    L.gjxo_mh_accept.argtypes = [vp, i64, u32, u32, vp, vp, i64, i32, vp]
    L.gjxo_mh_accept.restype = C.c_int64
    L.gjxo_exp_spacing.argtypes = [u32]
    L.gjxo_exp_spacing.restype = C.c_uint64
    L.gjxo_gather_rows.argtypes = [vp, i64, vp, i64, i32, vp, i64]
    L.gjxo_ssm_step.argtypes = [i32, i32, vp, vp, f32, f32, f32, u32, u32, i32, i32, i64, i64, vp,
                                i64, vp, vp, vp, vp, vp, i64]
    L.gjxo_ssm_step_move.argtypes = [i32, i32, vp, vp, f32, f32, f32, u32, u32, i32, i32, i64, i64, vp, vp, i64, vp, vp, vp, i32, f32,
                                     vp, vp, vp, vp, vp, i64]
    L.gjxo_score_grad.argtypes = [A.PP, i64, vp, vp, vp]
    L.gjxo_hmc.argtypes = [A.PP, u32, u32, i64, i64, f32, i32, i32, i32, vp, vp, vp, vp]
    for n in ("gjxo_erfinv",):
        getattr(L, n).argtypes = [f32]
        getattr(L, n).restype = f32
    for n in ("gjxo_normal_from_bits", "gjxo_gumbel_from_bits", "gjxo_unit_from_bits"):
        getattr(L, n).argtypes = [u32]
        getattr(L, n).restype = f32
    L.gjxo_set_margin_buffer.argtypes = [vp, i64]
    L.gjxo_set_momenta_buffer.argtypes = [vp, i64]
    L.gjxo_num_threads.restype = C.c_int
    L.gjxo_set_num_threads.argtypes = [C.c_int]
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def threefry2x32(k0, k1, c0, c1):
    out = np.zeros(2, np.uint32)
    lib().gjxo_threefry2x32(k0, k1, c0, c1, _p(out))
    return int(out[0]), int(out[1])


def fold_in(key, i):
    return threefry2x32(key[0], key[1], (i >> 32) & 0xFFFFFFFF, i & 0xFFFFFFFF)


def split(key, n=2):
    return [fold_in(key, i) for i in range(n)]


def set_threads(n: int):
    lib().gjxo_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().gjxo_num_threads())


def run_program(prog: PackedProgram, key, K, offset=0, choices=None, logw_in=None, sub=None,
                want_site_scores=False, K_total=None, want_margin=False):
    """Returns dict(choices [n_slots,K], score, weight, logw, lse[4], site_scores?, margin?).
    margin[i]: the smallest relative distance between the two sides of any float comparison that decided a DISCRETE
    outcome for particle i (category, accept / reject, floor); 3e38 when the particle took no such decision."""
    K = int(K)
    ns = max(prog.n_slots, 1)
    ch = np.zeros((ns, K), np.float32) if choices is None else np.ascontiguousarray(choices, np.float32).copy()
    score = np.zeros(K, np.float32)
    weight = np.zeros(K, np.float32)
    logw = np.zeros(K, np.float32)
    lse = np.zeros(4, np.float32)
    ss = np.zeros((max(prog.n_sites, 1), K), np.float32) if want_site_scores else None
    cp = prog.c_program(None)
    li = None if logw_in is None else np.ascontiguousarray(logw_in, np.float32)
    sb = None if sub is None else np.ascontiguousarray(sub, np.float32)
    margin = np.full(K, 3.0e38, np.float32) if want_margin else None
    if want_margin:
        lib().gjxo_set_margin_buffer(_p(margin), K)
    try:
        rc = lib().gjxo_run_program(C.byref(cp), key[0], key[1], K, int(offset), _p(ch), _p(score), _p(weight),
                                    _p(logw), _p(li), _p(sb), _p(ss), _p(lse), int(K_total or K))
    finally:
        if want_margin:
            lib().gjxo_set_margin_buffer(None, 0)
    assert rc == 0, rc
    out = dict(choices=ch, score=score, weight=weight, logw=logw, lse=lse)
    if margin is not None:
        out["margin"] = margin
    if ss is not None:
        out["site_scores"] = ss
    return out


def logsumexp(x, K_total=None):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(4, np.float32)
    lib().gjxo_logsumexp(_p(x), x.size, int(K_total or x.size), _p(out))
    return out


def categorical_pick(logw, lse, key, rng_mode=A.RNG_FLAT, offset=0):
    logw = np.ascontiguousarray(logw, np.float32)
    lse = np.ascontiguousarray(lse, np.float32)
    bv = C.c_float()
    bi = C.c_int64()
    lib().gjxo_categorical_pick(_p(logw), logw.size, int(offset), _p(lse), key[0], key[1], rng_mode,
                                C.byref(bv), C.byref(bi))
    return float(bv.value), int(bi.value)


def weight_cumsum(x, is_log=False, lse=None):
    x = np.ascontiguousarray(x, np.float32)
    cum = np.zeros(x.size, np.uint64)
    tot = C.c_uint64()
    l = None if lse is None else np.ascontiguousarray(lse, np.float32)
    lib().gjxo_weight_cumsum(_p(x), x.size, int(is_log), _p(l), _p(cum), C.byref(tot))
    return cum, int(tot.value)


def resample_systematic(cum, u, N_total, base=0, total_all=None, out_begin=0, n_out=None):
    cum = np.ascontiguousarray(cum, np.uint64)
    n_out = int(N_total if n_out is None else n_out)
    anc = np.zeros(n_out, np.int32)
    total_all = int(cum[-1]) if total_all is None else int(total_all)
    lib().gjxo_resample_systematic(_p(cum), cum.size, int(base), total_all, float(u), int(N_total),
                                   int(out_begin), n_out, _p(anc))
    return anc


def resample_systematic_tiled(logw, u, N=None, q=None):
    """-> (ancestors int32[N], q uint32[K], e int32[ceil(K/1024)], dead bool): the tile-scaled fixed-point scheme;
    ``q`` = quantised weights to use instead of the oracle's own exp2f (the device's)."""
    logw = np.ascontiguousarray(logw, np.float32)
    K = logw.size
    N = int(K if N is None else N)
    anc = np.zeros(N, np.int32)
    q_out = np.zeros(K, np.uint32)
    e_out = np.zeros((K + 1023) // 1024, np.int32)
    qi = None if q is None else np.ascontiguousarray(q, np.uint32)
    rc = lib().gjxo_resample_systematic_tiled(_p(logw), K, float(u), N, _p(qi), _p(anc), _p(q_out), _p(e_out))
    return anc, q_out, e_out, bool(rc)


def resample_sorted_multinomial_tiled(logw, key, N=None, q=None):
    """-> (ancestors int32[N], dead): multinomial resampling by SORTED uniforms (exponential spacings of the slots' words) under the
    tile-scaled fixed point; ``q``: quantised weights to use instead of the oracle's own exp2f (the device's)"""
    logw = np.ascontiguousarray(logw, np.float32)
    K = logw.size
    N = int(K if N is None else N)
    anc = np.zeros(N, np.int32)
    qi = None if q is None else np.ascontiguousarray(q, np.uint32)
    rc = lib().gjxo_resample_sorted_multinomial_tiled(_p(logw), K, key[0], key[1], N, _p(qi), _p(anc))
    return anc, bool(rc)


def mh_accept(log_alpha, key, rows_cur, rows_prop):
    """gjx_mh_accept: -> (rows after the accept f32[rows][K], accepted f32[K], margin f32[K] = |log u - alpha| of every chain)"""
    la = np.ascontiguousarray(log_alpha, np.float32)
    K = la.size
    cur = np.ascontiguousarray(rows_cur, np.float32).copy().reshape(-1, K)
    prop = np.ascontiguousarray(rows_prop, np.float32).reshape(-1, K)
    acc = np.zeros(K, np.float32)
    margin = np.full(K, 3.0e38, np.float32)
    lib().gjxo_set_margin_buffer(_p(margin), K)
    try:
        lib().gjxo_mh_accept(_p(la), K, key[0], key[1], _p(cur), _p(prop), K, cur.shape[0], _p(acc))
    finally:
        lib().gjxo_set_margin_buffer(None, 0)
    return cur, acc, margin


def exp_spacing(word: int) -> int:
    return int(lib().gjxo_exp_spacing(int(word) & 0xFFFFFFFF))


def resample_multinomial(cum, key, N_total, base=0, total_all=None, out_begin=0, n_out=None):
    cum = np.ascontiguousarray(cum, np.uint64)
    n_out = int(N_total if n_out is None else n_out)
    anc = np.zeros(n_out, np.int32)
    total_all = int(cum[-1]) if total_all is None else int(total_all)
    lib().gjxo_resample_multinomial(_p(cum), cum.size, int(base), total_all, key[0], key[1], int(N_total),
                                    int(out_begin), n_out, _p(anc))
    return anc


def gather_rows(src, anc):
    src = np.ascontiguousarray(src, np.float32)
    anc = np.ascontiguousarray(anc, np.int32)
    dst = np.zeros((src.shape[0], anc.size), np.float32)
    lib().gjxo_gather_rows(_p(src), src.shape[1], _p(anc), anc.size, src.shape[0], _p(dst), anc.size)
    return dst


def ssm_step(A_, H, q, r, q0, key, rng_mode, t, K, x_prev, anc, y, offset=0, K_total=None):
    A_ = np.ascontiguousarray(A_, np.float32)
    dx = A_.shape[0]
    Hc = None if H is None else np.ascontiguousarray(H, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    dy = y.size
    xp = None if x_prev is None else np.ascontiguousarray(x_prev, np.float32)
    an = None if anc is None else np.ascontiguousarray(anc, np.int32)
    xo = np.zeros((dx, K), np.float32)
    lw = np.zeros(K, np.float32)
    lse = np.zeros(4, np.float32)
    lib().gjxo_ssm_step(dx, dy, _p(A_), _p(Hc), q, r, q0, key[0], key[1], rng_mode, t, K, int(offset),
                        _p(xp), 0 if xp is None else xp.shape[1], _p(an), _p(y), _p(xo), _p(lw), _p(lse),
                        int(K_total or K))
    return xo, lw, lse


def ssm_step_move(A_, H, q, r, q0, key, rng_mode, t, K, x_prev, m_prev, anc, y_prev, y, n_moves, move_scale, offset=0):
    """-> (x_out, m_out, logw, accepted, lse, margin): the step with the resample-move rejuvenation in front"""
    A_ = np.ascontiguousarray(A_, np.float32)
    dx = A_.shape[0]
    Hc = None if H is None else np.ascontiguousarray(H, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    yp = None if y_prev is None else np.ascontiguousarray(y_prev, np.float32)
    xp = None if x_prev is None else np.ascontiguousarray(x_prev, np.float32)
    mp = None if m_prev is None else np.ascontiguousarray(m_prev, np.float32)
    an = None if anc is None else np.ascontiguousarray(anc, np.int32)
    x_out, m_out = np.zeros((dx, K), np.float32), np.zeros((dx, K), np.float32)
    logw, acc, lse = np.zeros(K, np.float32), np.zeros(K, np.float32), np.zeros(4, np.float32)
    margin = np.full(K, 3.0e38, np.float32)
    lib().gjxo_set_margin_buffer(_p(margin), K)
    try:
        rc = lib().gjxo_ssm_step_move(dx, y.size, _p(A_), _p(Hc), q, r, q0, key[0], key[1], rng_mode, int(t), int(K), int(offset), _p(xp),
                                      _p(mp), 0 if xp is None else xp.shape[1], _p(an), _p(yp), _p(y), int(n_moves), float(move_scale),
                                      _p(x_out), _p(m_out), _p(logw), _p(acc), _p(lse), int(K))
    finally:
        lib().gjxo_set_margin_buffer(None, 0)
    assert rc == 0
    return x_out, m_out, logw, acc, lse, margin


def score_grad(prog: PackedProgram, choices):
    ch = np.ascontiguousarray(choices, np.float32)
    n = ch.shape[1]
    score = np.zeros(n, np.float32)
    grad = np.zeros_like(ch)
    cp = prog.c_program(None)
    lib().gjxo_score_grad(C.byref(cp), n, _p(ch), _p(score), _p(grad))
    return score, grad


def hmc(prog: PackedProgram, key, choices, eps, L, stale=False, accept=False, offset=0, want_momenta=False):
    """-> dict(choices, score, alpha, accepted, margin): margin[i] = |log u - alpha| of chain i's accept decision (with
    ``accept``; 3e38 otherwise) — a device chain may take the other branch only where this is below the tolerance to
    which its alpha matches the oracle's.  ``want_momenta``: also "momenta" [selected scalars, n], the initial draw."""
    ch = np.ascontiguousarray(choices, np.float32).copy()
    n = ch.shape[1]
    mom = np.zeros((prog.n_slots, n), np.float32) if want_momenta else None
    if want_momenta:
        lib().gjxo_set_momenta_buffer(_p(mom), n)
    score = np.zeros(n, np.float32)
    alpha = np.zeros(n, np.float32)
    acc = np.zeros(n, np.float32)
    margin = np.full(n, 3.0e38, np.float32)
    cp = prog.c_program(None)
    lib().gjxo_set_margin_buffer(_p(margin), n)
    try:
        lib().gjxo_hmc(C.byref(cp), key[0], key[1], n, int(offset), float(eps), int(L), int(stale), int(accept),
                       _p(ch), _p(score), _p(alpha), _p(acc))
    finally:
        lib().gjxo_set_margin_buffer(None, 0)
        lib().gjxo_set_momenta_buffer(None, 0)
    out = dict(choices=ch, score=score, alpha=alpha, accepted=acc, margin=margin)
    if want_momenta:
        out["momenta"] = mom
    return out
