"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/gjx.h).

Every function launches asynchronously on torch's current HIP stream and returns device tensors.
"""
from __future__ import annotations

import ctypes as C
import time

import torch

from . import _abi as A
from ._lib import GjxError, check, load
from .program import PackedProgram


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle(device_index=None) -> int:
    """torch's current HIP stream of the device as a raw handle (the C-level query when torch has it: building a
    torch.cuda.Stream object per kernel call was a fifth of the API step's host time)"""
    if _raw_stream is not None:
        return int(_raw_stream(torch.cuda.current_device() if device_index is None else device_index))
    return torch.cuda.current_stream(device_index).cuda_stream


def _stream():
    return C.c_void_p(_stream_handle())


def _dev(device=None):
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def workspace(op: int, K: int, device=None) -> torch.Tensor:
    n = load().gjx_workspace_bytes(op, int(K))
    return torch.zeros(n, dtype=torch.uint8, device=_dev(device))   # control block must start zeroed (gjx.h)


_shared_ws: dict = {}


def shared_workspace(op: int, K: int, device=None) -> torch.Tensor:
    """A zero-initialised workspace kept per (op, size, device, stream) for callers that hand nothing of it on (every entry
    point leaves the control block zeroed, so one allocation + one memset serves all later calls on that stream)."""
    dev = _dev(device)
    n = load().gjx_workspace_bytes(op, int(K))
    k = (int(op), n, dev, _stream_handle(dev.index))   # per op: a run's block partials must survive the resampler's granules
    ws = _shared_ws.get(k)
    if ws is None:
        if len(_shared_ws) >= 16:
            _shared_ws.pop(next(iter(_shared_ws)))
        ws = _shared_ws[k] = torch.zeros(n, dtype=torch.uint8, device=dev)
    return ws


def workspace_status(ws: torch.Tensor, raise_on_error: bool = True) -> int:
    """Read and clear the status word of a workspace (synchronises the stream).  Bit 0: a co-resident kernel ran out of
    its poll budget (its output is undefined); bit 1: a resampling call saw a zero total weight (identity ancestors)."""
    st = C.c_int32(0)
    check(load().gjx_workspace_status(_ptr(ws), C.byref(st), _stream()), "gjx_workspace_status")
    if raise_on_error and st.value:
        what = [m for b, m in ((1, "a co-resident kernel timed out waiting for its peers (grid not co-resident?)"),
                               (2, "all weights are zero / -inf / NaN: nothing to resample from")) if st.value & b]
        raise GjxError("workspace status %d: %s" % (st.value, "; ".join(what)))
    return int(st.value)


def threefry2x32(key, n: int, ctr_lo0: int = 0, ctr_hi: int = 0, device=None) -> torch.Tensor:
    out = torch.empty((n, 2), dtype=torch.int32, device=_dev(device))
    check(load().gjx_threefry2x32(key[0], key[1], ctr_hi, ctr_lo0, n, _ptr(out), _stream()), "gjx_threefry2x32")
    return out


class DispatchTimer:
    """A pair of HIP events a call attaches to the dispatch of its kernel (run_program(timer=...), importance_step(timer=...))."""

    def __init__(self):
        self.a, self.b = C.c_void_p(), C.c_void_p()
        check(load().gjx_event_create(C.byref(self.a)), "gjx_event_create")
        check(load().gjx_event_create(C.byref(self.b)), "gjx_event_create")

    def elapsed_us(self) -> float:
        us = C.c_float()
        check(load().gjx_event_elapsed_us(self.a, self.b, C.byref(us)), "gjx_event_elapsed_us")
        return float(us.value)

    def close(self) -> None:
        load().gjx_event_destroy(self.a)
        load().gjx_event_destroy(self.b)


def _cp_for_query(prog: PackedProgram, device=None):
    """Program struct for engine / grid queries: the device form (with the prepared constants the fused engines need)
    whenever a GPU is there, the host form otherwise (struct checks on a CPU-only box)."""
    return prog.c_program(_dev(device) if (device is not None or torch.cuda.is_available()) else None)


def program_engine(prog: PackedProgram, device=None) -> int:
    cp = _cp_for_query(prog, device)
    return int(load().gjx_program_engine(C.byref(cp)))


def program_source(prog: PackedProgram, ppt: int = 0) -> str:
    """HIP source of the kernel gjx_codegen generates for this program (raises GjxError if the emitter does not cover it)."""
    cp = prog.c_program(None)
    n = load().gjx_program_source(C.byref(cp), int(ppt), None, 0)
    if n < 0:
        check(int(n), "gjx_program_source")
    buf = C.create_string_buffer(int(n) + 1)
    load().gjx_program_source(C.byref(cp), int(ppt), buf, int(n) + 1)
    return buf.value.decode()


def jit_stats() -> dict:
    """what the generated-kernel caches did in this process so far (gjx_jit_stats)"""
    out = (C.c_int64 * 4)()
    check(load().gjx_jit_stats(out), "gjx_jit_stats")
    return dict(hiprtc_compiles=int(out[0]), disk_hits=int(out[1]), hiprtc_ms=int(out[2]) / 1e3, structures=int(out[3]))


def program_precompile(prog: PackedProgram, ppt: int) -> None:
    """Compile the program's generated kernel with hipRTC (works without a GPU) into the in-memory and on-disk caches."""
    cp = prog.c_program(None)
    check(load().gjx_program_precompile(C.byref(cp), int(ppt)), "gjx_program_precompile")


def program_hmc_source(prog: PackedProgram) -> str:
    """HIP source of the HMC kernel gjx_codegen generates for this program (raises GjxError if the emitter does not cover it)."""
    cp = prog.c_program(None)
    n = load().gjx_program_hmc_source(C.byref(cp), None, 0)
    if n < 0:
        check(int(n), "gjx_program_hmc_source")
    buf = C.create_string_buffer(int(n) + 1)
    load().gjx_program_hmc_source(C.byref(cp), buf, int(n) + 1)
    return buf.value.decode()


def program_hmc_precompile(prog: PackedProgram) -> None:
    """Compile the program's generated HMC kernel with hipRTC (works without a GPU) into the in-memory and on-disk caches."""
    cp = prog.c_program(None)
    check(load().gjx_program_hmc_precompile(C.byref(cp)), "gjx_program_hmc_precompile")


def program_filter_source(step: PackedProgram, tiles_per_block: int = 1) -> str:
    """HIP source of the FILTER kernel gjx_codegen generates for a step program (gjx_gen_pf: the step's sites as the model of the
    shared filter skeleton, csrc/gjx_pfcore.h); raises GjxError if the emitter does not cover it"""
    cp = step.c_program(None)
    n = load().gjx_program_filter_source(C.byref(cp), int(tiles_per_block), None, 0)
    if n < 0:
        check(int(n), "gjx_program_filter_source")
    buf = C.create_string_buffer(int(n) + 1)
    load().gjx_program_filter_source(C.byref(cp), int(tiles_per_block), buf, int(n) + 1)
    return buf.value.decode()


def program_filter_precompile(step: PackedProgram, tiles_per_block: int = 1) -> None:
    """Compile the step program's filter kernel with hipRTC (works without a GPU) into the in-memory and on-disk caches."""
    cp = step.c_program(None)
    check(load().gjx_program_filter_precompile(C.byref(cp), int(tiles_per_block)), "gjx_program_filter_precompile")


class RunPartials:
    """The per-block {max, sumexp} pairs a run with ``want_lse=False`` left in its workspace: the consumers that can
    reduce them in their own prologue (resample_gather / resample_indices, ``partials=``) save the producer's serial
    LSE tail.  Valid only until the same workspace is used by another run."""

    def __init__(self, ws, gen, prog, K, offset, n=None, tiles=0):
        self.ws, self.gen, self.prog, self.K, self.offset = ws, gen, prog, K, offset
        self._n = n                     # the grid the run actually launched (recorded right after the call)
        self.tiles = int(tiles)         # byte offset of the tile totals the run left for resample_gather_tiled (0: none)

    def valid(self) -> bool:
        return getattr(self.ws, "_gjx_gen", None) == self.gen

    def count(self) -> int:
        if self._n is None:
            self._n = run_partials_count(self.prog, self.K, self.offset, self.ws.device)
        return self._n

    def as_arg(self):
        """(workspace, n) for the ``partials=`` argument of the resampling wrappers"""
        return (self.ws, self.count())

    def finish(self, K_total) -> torch.Tensor:
        """-> the finished LSE record f32[4] (one small launch over the pairs)"""
        n = self.count()
        pairs = self.ws[256:256 + 8 * n].view(torch.float32)
        return lse_combine(pairs, K_total)


def run_program(prog: PackedProgram, key, K: int, offset: int = 0, choices=None, logw_in=None, sub=None,
                want_site_scores=False, want_lse=True, K_total=None, device=None, ws=None, out=None,
                want_weight=True, want_tiles=False, in_rows=None, ancestors=None, store_inputs=False, timer=None):
    """gjx_run_program.  Returns dict(choices, score, weight, logw, lse[, site_scores]).  ``want_tiles`` (with
    ``want_lse=False``): ask the kernel to leave the tile totals of the tile-scaled resampler beside its block partials
    (RunPartials.tiles; 0 when this engine / size cannot).  ``in_rows`` f32[rows][stride] (+ ``ancestors`` int32[K]): the
    program's INPUT sites read in_rows[obs_off + d][ancestors[i]] — the particle gather of a resampling step fused into this
    propagate step (gjx_run_program_ex)."""
    dev = _dev(device)
    K = int(K)
    f32 = torch.float32
    if out is None:
        out = {}
    ch = choices if choices is not None else out.get("choices")
    if ch is None:
        ch = torch.empty((max(prog.n_slots, 1), K), dtype=f32, device=dev)
    score = out.get("score") if out.get("score") is not None else torch.empty(K, dtype=f32, device=dev)
    weight = (out.get("weight") if out.get("weight") is not None else torch.empty(K, dtype=f32, device=dev)) if want_weight else None
    logw = out.get("logw") if out.get("logw") is not None else torch.empty(K, dtype=f32, device=dev)
    lse = (out.get("lse") if out.get("lse") is not None else torch.empty(4, dtype=f32, device=dev)) if want_lse else None
    ss = torch.empty((max(prog.n_sites, 1), K), dtype=f32, device=dev) if want_site_scores else None
    if ws is None:
        ws = workspace(A.OP_RUN, K, dev)
    cp = prog.c_program(dev)
    opts = A.GjxRunOpts()
    opts.flags = (A.RUN_LEAVE_TILES if want_tiles else 0) | (A.RUN_STORE_INPUTS if store_inputs else 0)
    if timer is not None:                     # DispatchTimer: HIP events attached to the propagate kernel's dispatch
        opts.flags |= A.RUN_TIME_DISPATCH
        opts.start_event, opts.stop_event = timer.a, timer.b
    if in_rows is not None:
        opts.in_rows, opts.in_stride = in_rows.data_ptr(), int(in_rows.stride(0))
        opts.in_ancestors = None if ancestors is None else ancestors.data_ptr()
    info = A.GjxRunInfo()
    rc = load().gjx_run_program_ex(C.byref(cp), key[0], key[1], K, int(offset), _ptr(ch), _ptr(score), _ptr(weight),
                                   _ptr(logw), _ptr(logw_in), _ptr(sub), _ptr(ss), _ptr(lse), int(K_total or K),
                                   _ptr(ws), ws.numel(), _stream(), C.byref(opts), C.byref(info))
    check(rc, "gjx_run_program_ex")
    # generation stamp of the workspace: with lse == None the kernel leaves per-block {max, sumexp} partials at ws + 256,
    # valid until the next run through the same workspace (RunPartials.valid)
    ws._gjx_gen = getattr(ws, "_gjx_gen", 0) + 1
    res = dict(choices=ch, score=score, weight=weight, logw=logw, lse=lse, _ws=ws, _engine=int(info.engine))
    if lse is None:
        res["_partials"] = RunPartials(ws, ws._gjx_gen, prog, K, int(offset), int(info.n_partials), int(info.tiles_offset))
    if ss is not None:
        res["site_scores"] = ss
    return res


def importance_step(prog: PackedProgram, key, K: int, u: float, offset: int = 0, out=None, ws=None, device=None,
                    allow_fallback: bool = True, timer=None):
    """One importance step (propagate + reweight + LSE + systematic resample + gather).  One launch when the program has
    a fused engine and the grid is co-resident (gjx_importance_step), otherwise the three calls it replaces.
    -> dict(choices, score, logw, lse, rows (resampled), ancestors, fused: bool)"""
    dev = _dev(device)
    K = int(K)
    f32 = torch.float32
    out = {} if out is None else out
    n = max(prog.n_slots, 1)
    for name, shape, dt in (("choices", (n, K), f32), ("score", (K,), f32), ("logw", (K,), f32), ("lse", (4,), f32),
                            ("rows", (n, K), f32), ("ancestors", (K,), torch.int32)):
        if out.get(name) is None:
            out[name] = torch.empty(shape, dtype=dt, device=dev)
    if ws is None:
        ws = out.get("_ws")
    if ws is None:
        ws = workspace(A.OP_RUN, K, dev)
    out["_ws"] = ws
    cp = prog.c_program(dev)
    rc = load().gjx_importance_step_ex(C.byref(cp), key[0], key[1], K, int(offset), _ptr(out["choices"]), _ptr(out["score"]),
                                       _ptr(out["logw"]), _ptr(out["lse"]), float(u), _ptr(out["rows"]), _ptr(out["ancestors"]),
                                       _ptr(ws), ws.numel(), _stream(), timer.a if timer is not None else None,
                                       timer.b if timer is not None else None)
    if rc == A.EUNSUPPORTED and allow_fallback:
        run_program(prog, key, K, offset=offset, ws=ws, out=out, want_weight=False, want_lse=False)
        if out.get("_ws2") is None:
            out["_ws2"] = workspace(A.OP_RESAMPLE, K, dev)
        resample_indices(out["logw"], u, K, partials=(ws, run_partials_count(prog, K, offset)), lse_out=out["lse"], K_total=K,
                         anc=out["ancestors"], ws=out["_ws2"])
        gather_rows(out["choices"], out["ancestors"], out["rows"])
        out["fused"] = False
        return out
    check(rc, "gjx_importance_step")
    out["fused"] = True
    return out


def logsumexp(x: torch.Tensor, K_total=None, ws=None) -> torch.Tensor:
    K = x.numel()
    out = torch.empty(4, dtype=torch.float32, device=x.device)
    if ws is None:
        ws = workspace(A.OP_LSE, K, x.device)
    check(load().gjx_logsumexp(_ptr(x), K, int(K_total or K), _ptr(out), _ptr(ws), ws.numel(), _stream()), "gjx_logsumexp")
    return out


def lse_combine(pairs: torch.Tensor, K_total: int) -> torch.Tensor:
    out = torch.empty(4, dtype=torch.float32, device=pairs.device)
    G = pairs.numel() // 2
    check(load().gjx_lse_combine(_ptr(pairs), G, int(K_total), _ptr(out), _stream()), "gjx_lse_combine")
    return out


def categorical_pick(logw: torch.Tensor, lse: torch.Tensor, key, rng_mode=A.RNG_FLAT, offset=0, ws=None) -> torch.Tensor:
    """Returns an int32[2] device tensor: [bitcast(best value), index]."""
    K = logw.numel()
    out = torch.empty(2, dtype=torch.int32, device=logw.device)
    if ws is None:
        ws = workspace(A.OP_PICK, K, logw.device)
    check(load().gjx_categorical_pick(_ptr(logw), K, int(offset), _ptr(lse), key[0], key[1], rng_mode, _ptr(out),
                                      _ptr(ws), ws.numel(), _stream()), "gjx_categorical_pick")
    return out


def trials_lse_pick(logw: torch.Tensor, n_trials: int, K: int, key=None, rng_mode=A.RNG_FLAT, offset=0):
    """gjx_trials_lse_pick: per-trial LSE records f32[n_trials][4] and (with ``key``) per-trial 1-of-K draws int32[n_trials]
    (global indices) of n_trials * K log-weights laid out trial after trial."""
    assert logw.numel() == n_trials * K
    lse = torch.empty((n_trials, 4), dtype=torch.float32, device=logw.device)
    pick = torch.empty(n_trials, dtype=torch.int32, device=logw.device) if key is not None else None
    k = key if key is not None else (0, 0)
    check(load().gjx_trials_lse_pick(_ptr(logw), int(n_trials), int(K), int(offset), k[0], k[1], rng_mode, _ptr(lse),
                                     _ptr(pick) if pick is not None else None, _stream()), "gjx_trials_lse_pick")
    return lse, pick


def weight_cumsum(x: torch.Tensor, is_log=False, lse=None, ws=None, out=None, partials=None, lse_out=None, K_total=None,
                  pairs=None):
    """-> (cum uint64[K] as int64 payload, base_total int64[2] = {0, total}).
    ``partials=(run_workspace, n)`` selects mode 2: the max comes from the per-block pairs a preceding
    run_program(want_lse=False) left in its workspace, and ``lse_out`` (f32[4]) receives the finished record."""
    K = x.numel()
    cum = torch.empty(K, dtype=torch.int64, device=x.device) if out is None else out[0]
    bt = torch.empty(2, dtype=torch.int64, device=x.device) if out is None else out[1]
    if ws is None:
        ws = workspace(A.OP_RESAMPLE, K, x.device)
    if pairs is not None:                                  # raw {max, sumexp} pairs, e.g. one per rank
        mode, lp, npart = 2, _ptr(pairs), pairs.numel() // 2
    elif partials is not None:
        run_ws, n = partials
        mode, lp, npart = 2, C.c_void_p(run_ws.data_ptr() + 256), int(n)
    else:
        mode, lp, npart = int(bool(is_log)), _ptr(lse), 0
    check(load().gjx_weight_cumsum(_ptr(x), K, mode, lp, npart, _ptr(cum), _ptr(bt), _ptr(lse_out), int(K_total or K),
                                   _ptr(ws), ws.numel(), _stream()), "gjx_weight_cumsum")
    return cum, bt


def resample_indices(x: torch.Tensor, u: float, N: int | None = None, is_log=True, lse=None, partials=None, lse_out=None,
                     K_total=None, anc=None, cum=None, bt=None, ws=None) -> torch.Tensor:
    """gjx_resample_indices: weights -> systematic ancestors in one launch (single GPU).  -> ancestors int32[N]"""
    K = x.numel()
    N = int(N or K)
    if anc is None:
        anc = torch.empty(N, dtype=torch.int32, device=x.device)
    if ws is None:
        ws = workspace(A.OP_RESAMPLE, K, x.device)
    if partials is not None:
        run_ws, n = partials
        mode, lp, npart = 2, C.c_void_p(run_ws.data_ptr() + 256), int(n)
    else:
        mode, lp, npart = int(bool(is_log)), _ptr(lse), 0
    def call(cum_, bt_):
        return load().gjx_resample_indices(_ptr(x), K, mode, lp, npart, float(u), N, _ptr(anc), _ptr(cum_), _ptr(bt_), _ptr(lse_out),
                                           int(K_total or K), _ptr(ws), ws.numel(), _stream())

    rc = call(cum, bt)
    if rc == A.EUNSUPPORTED and (cum is None or bt is None):
        # the grid would not be co-resident on THIS device (its capacity is the library's to know: a partitioned or smaller
        # GPU holds fewer blocks than a full MI355X): the three-launch path needs the prefix-sum buffers
        rc = call(torch.empty(K, dtype=torch.int64, device=x.device) if cum is None else cum,
                  torch.empty(2, dtype=torch.int64, device=x.device) if bt is None else bt)
    check(rc, "gjx_resample_indices")
    return anc


def resample_indices_tiled(logw: torch.Tensor, u: float, N: int | None = None, anc=None, cum=None, ws=None, want_q=False):
    """gjx_resample_indices_tiled: log-weights -> systematic ancestors under the tile-scaled fixed point (gjx.h,
    GJX_WEIGHTS_TILE_SCALED).  -> ancestors int32[N], or (ancestors, q int32[K], e int32[ceil(K/1024)]) with want_q"""
    K = logw.numel()
    N = int(N or K)
    dev = logw.device
    if anc is None:
        anc = torch.empty(N, dtype=torch.int32, device=dev)
    if cum is None:
        cum = torch.empty(K, dtype=torch.int64, device=dev)
    if ws is None:
        ws = workspace(A.OP_RESAMPLE, K, dev)
    q = torch.empty(K, dtype=torch.int32, device=dev) if want_q else None
    e = torch.empty((K + 1023) // 1024, dtype=torch.int32, device=dev) if want_q else None
    check(load().gjx_resample_indices_tiled(_ptr(logw), K, float(u), N, _ptr(anc), _ptr(cum), _ptr(q), _ptr(e), _ptr(ws), ws.numel(),
                                            _stream()), "gjx_resample_indices_tiled")
    return (anc, q, e) if want_q else anc


def resample_sorted_multinomial_tiled(logw: torch.Tensor, key, N: int | None = None, anc=None, cum=None, ws=None, want_q=False):
    """gjx_resample_sorted_multinomial_tiled: log-weights -> MULTINOMIAL ancestors (sorted uniforms from exponential spacings under
    `key`, tile-scaled fixed point), non-decreasing.  -> ancestors int32[N], or (ancestors, q, e) with want_q"""
    K = logw.numel()
    N = int(N or K)
    dev = logw.device
    if anc is None:
        anc = torch.empty(N, dtype=torch.int32, device=dev)
    if cum is None:
        cum = torch.empty(K, dtype=torch.int64, device=dev)
    if ws is None:
        ws = workspace(A.OP_RESAMPLE, K, dev)
    q = torch.empty(K, dtype=torch.int32, device=dev) if want_q else None
    e = torch.empty((K + 1023) // 1024, dtype=torch.int32, device=dev) if want_q else None
    check(load().gjx_resample_sorted_multinomial_tiled(_ptr(logw), K, key[0], key[1], N, _ptr(anc), _ptr(cum), _ptr(q), _ptr(e), _ptr(ws),
                                                       ws.numel(), _stream()), "gjx_resample_sorted_multinomial_tiled")
    return (anc, q, e) if want_q else anc


def resample_gather(x: torch.Tensor, u: float, rows: torch.Tensor, is_log=True, lse=None, partials=None, lse_out=None,
                    K_total=None, out=None, anc=None, ws=None, allow_fallback=True) -> torch.Tensor:
    """gjx_resample_gather: weights -> systematic ancestors -> out[r, j] = rows[r, ancestor(j)] in ONE launch (N = K);
    `anc` (int32[K]) receives the ancestors when given.  Falls back to resample_indices + gather_rows when the grid
    would not be co-resident (K > 2^20 on a full MI355X) unless allow_fallback is False."""
    K = x.numel()
    assert rows.dim() == 2 and rows.shape[1] == K and rows.stride(1) == 1
    if out is None:
        out = torch.empty_like(rows)
    if ws is None:
        ws = workspace(A.OP_RESAMPLE, K, x.device)
    if partials is not None:
        run_ws, n = partials
        mode, lp, npart = 2, C.c_void_p(run_ws.data_ptr() + 256), int(n)
    else:
        mode, lp, npart = int(bool(is_log)), _ptr(lse), 0
    rc = load().gjx_resample_gather(_ptr(x), K, mode, lp, npart, float(u), _ptr(rows), rows.stride(0), rows.shape[0], _ptr(out),
                                    out.stride(0), _ptr(anc), _ptr(lse_out), int(K_total or K), _ptr(ws), ws.numel(), _stream())
    if rc == A.EUNSUPPORTED and allow_fallback:
        a = resample_indices(x, u, K, is_log=is_log, lse=lse, partials=partials, lse_out=lse_out, K_total=K_total, anc=anc, ws=ws)
        return gather_rows(rows, a, dst=out)
    check(rc, "gjx_resample_gather")
    return out


def resample_gather_tiled(logw: torch.Tensor, u: float, rows: torch.Tensor, partials=None, tiles: int = 0, lse_out=None,
                          K_total=None, out=None, anc=None, ws=None) -> torch.Tensor:
    """gjx_resample_gather_tiled: the tile-scaled systematic resampler + row gather (N = K) as a PLAIN launch (no
    co-resident grid).  ``partials=(run_workspace, n)``: the block pairs of the producing run (lse_out receives the finished
    record); ``tiles``: byte offset of the tile totals that run left in the same workspace (RunPartials.tiles; 0: they are
    computed by one extra small launch).  Ancestors == resample_indices_tiled's, bit for bit."""
    K = logw.numel()
    assert rows.dim() == 2 and rows.shape[1] == K and rows.stride(1) == 1
    if out is None:
        out = torch.empty_like(rows)
    if ws is None:
        ws = workspace(A.OP_RESAMPLE, K, logw.device)
    mode, lp, npart, S, E = 0, None, 0, None, None
    if partials is not None:
        run_ws, n = partials
        mode, lp, npart = 2, C.c_void_p(run_ws.data_ptr() + 256), int(n)
        if tiles:
            S = C.c_void_p(run_ws.data_ptr() + int(tiles))
            E = C.c_void_p(run_ws.data_ptr() + int(tiles) + 8 * ((K + 1023) // 1024))
    check(load().gjx_resample_gather_tiled(_ptr(logw), K, S, E, mode, lp, npart, float(u), _ptr(rows), rows.stride(0), rows.shape[0],
                                           _ptr(out), out.stride(0), _ptr(anc), _ptr(lse_out), int(K_total or K), _ptr(ws),
                                           ws.numel(), _stream()), "gjx_resample_gather_tiled")
    return out


def run_partials_count(prog: PackedProgram, K: int, offset: int = 0, device=None) -> int:
    cp = _cp_for_query(prog, device)
    return int(load().gjx_run_partials_count(C.byref(cp), int(K), int(offset)))


def resample_systematic(cum, base_total, u: float, N_total: int, out_begin=0, n_out=None, prefill=True) -> torch.Tensor:
    """Slots that fall on another rank's particles are left untouched by the kernel; ``prefill`` marks them -1."""
    n_out = int(N_total if n_out is None else n_out)
    anc = (torch.full((n_out,), -1, dtype=torch.int32, device=cum.device) if prefill
           else torch.empty(n_out, dtype=torch.int32, device=cum.device))
    check(load().gjx_resample_systematic(_ptr(cum), cum.numel(), _ptr(base_total), float(u), int(N_total),
                                         int(out_begin), n_out, _ptr(anc), _stream()), "gjx_resample_systematic")
    return anc


def resample_gather_systematic(cum, base_total, u: float, N_total: int, src: torch.Tensor, dst=None, out_begin=0,
                               n_out=None, want_ancestors=False, anc=None):
    """Systematic ancestors + SoA row gather in one call.  -> (dst [rows][n_out], ancestors)"""
    n_out = int(N_total if n_out is None else n_out)
    rows, stride = src.shape
    if dst is None:
        dst = torch.empty((rows, n_out), dtype=torch.float32, device=src.device)
    if anc is None:
        anc = torch.empty(n_out, dtype=torch.int32, device=src.device)
    check(load().gjx_resample_gather_systematic(_ptr(cum), cum.numel(), _ptr(base_total), float(u), int(N_total),
                                                int(out_begin), n_out, _ptr(src), stride, rows, _ptr(dst), dst.shape[1],
                                                _ptr(anc), _stream()), "gjx_resample_gather_systematic")
    return dst, anc


def resample_multinomial(cum, base_total, key, N_total: int, out_begin=0, n_out=None) -> torch.Tensor:
    n_out = int(N_total if n_out is None else n_out)
    anc = torch.empty(n_out, dtype=torch.int32, device=cum.device)
    check(load().gjx_resample_multinomial(_ptr(cum), cum.numel(), _ptr(base_total), key[0], key[1], int(N_total),
                                          int(out_begin), n_out, _ptr(anc), _stream()), "gjx_resample_multinomial")
    return anc


def gather_rows(src: torch.Tensor, anc: torch.Tensor, dst: torch.Tensor | None = None) -> torch.Tensor:
    rows = src.shape[0]
    n = anc.numel()
    if dst is None:
        dst = torch.empty((rows, n), dtype=torch.float32, device=src.device)
    # dst may be a column window of a wider SoA buffer: its row stride is what the kernel needs
    check(load().gjx_gather_rows(_ptr(src), src.stride(0), _ptr(anc), n, rows, _ptr(dst), dst.stride(0), _stream()),
          "gjx_gather_rows")
    return dst


class ShardPlan:
    """Device plan of one rank's part of a sharded resampling + its pinned host mirror (gjx_shard_plan)."""
    _SEQ = A.GjxShardPlan.seq.offset // 8

    def __init__(self, device):
        n = C.sizeof(A.GjxShardPlan)
        self.dev = torch.empty(n, dtype=torch.uint8, device=device)
        self.host = torch.zeros(n, dtype=torch.uint8).pin_memory()
        self._words = self.host.numpy().view("int64")
        self._struct = A.GjxShardPlan.from_buffer(self.host.numpy())     # live view of the pinned bytes
        self.seq = 0

    def build(self, totals: torch.Tensor, rank: int, u: float, N_total: int) -> "ShardPlan":
        self.seq += 1
        check(load().gjx_shard_plan_build(_ptr(totals), totals.numel(), int(rank), float(u), int(N_total), self.seq,
                                          _ptr(self.dev), C.c_void_p(self.host.data_ptr()), _stream()), "gjx_shard_plan_build")
        return self

    def wait(self, timeout_s: float = 60.0) -> A.GjxShardPlan:
        """Spin until the plan kernel of the last build() has published its sequence number in the pinned mirror
        (no stream or event synchronisation: kernels queued behind the plan keep running).  The returned struct
        is a live view, valid until the next build()."""
        w, k, s = self._words, self._SEQ, self.seq
        if w[k] != s:
            t0 = time.perf_counter()
            while w[k] != s:
                if time.perf_counter() - t0 > timeout_s:
                    raise GjxError("gjx_shard_plan: the plan kernel did not complete within %.0f s" % timeout_s)
        return self._struct


def shard_resample(cum, plan: ShardPlan, u: float, N_total: int, src: torch.Tensor, own_n: int, anc=None, dst=None):
    """gjx_shard_resample: this rank's ancestors (device-planned slot run) + in-place gather of the kept children.
    -> (ancestors int32[N_total] scratch, dst f32[R][own_n])"""
    R = src.shape[0]
    if anc is None:
        anc = torch.empty(int(N_total), dtype=torch.int32, device=src.device)
    if dst is None:
        dst = torch.empty((R, own_n), dtype=src.dtype, device=src.device)
    check(load().gjx_shard_resample(_ptr(cum), cum.numel(), _ptr(plan.dev), float(u), int(N_total), _ptr(anc), anc.numel(),
                                    _ptr(src), src.stride(0), R, _ptr(dst), dst.stride(0) if own_n else 0, int(own_n),
                                    _stream()), "gjx_shard_resample")
    return anc, dst


def rccl_library_path() -> str:
    """The RCCL build this process already uses (torch's), so there is a single RCCL instance per process."""
    import os
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


def rccl_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    check(load().gjx_rccl_unique_id(rccl_library_path().encode(), C.cast(buf, C.c_void_p)), "gjx_rccl_unique_id")
    return bytes(buf)


class ShardContext:
    """gjx_shard_ctx: RCCL communicator + scratch for the one-call sharded resampling of a fixed shape."""

    def __init__(self, unique_id: bytes, world: int, rank: int, K_local: int, rows: int, N_total: int):
        self._h = C.c_void_p()
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(load().gjx_shard_ctx_create(rccl_library_path().encode(), C.cast(idb, C.c_void_p), world, rank, int(K_local),
                                          int(rows), int(N_total), C.byref(self._h)), "gjx_shard_ctx_create")
        self.world, self.rank, self.K, self.rows, self.N_total = world, rank, int(K_local), int(rows), int(N_total)
        q, rem = divmod(self.N_total, world)
        self.own_n = q + (1 if rank < rem else 0)
        self._info = (C.c_int64 * 4)()

    def step(self, rows: torch.Tensor, logw: torch.Tensor, local_lse: torch.Tensor, u: float, out=None, lse_out=None):
        """-> (new_rows f32[R][own_n], global LSE record f32[4]); info in .last_info after the call"""
        assert rows.shape == (self.rows, self.K) and logw.numel() == self.K and rows.stride(1) == 1
        if out is None:
            out = torch.empty((self.rows, self.own_n), dtype=rows.dtype, device=rows.device)
        if lse_out is None:
            lse_out = torch.empty(4, dtype=torch.float32, device=rows.device)
        check(load().gjx_shard_resample_step(self._h, _ptr(logw), _ptr(local_lse), _ptr(rows), rows.stride(0), _ptr(out),
                                             out.stride(0), float(u), _ptr(lse_out), C.cast(self._info, C.c_void_p), _stream()),
              "gjx_shard_resample_step")
        return out, lse_out

    def step_multinomial(self, rows: torch.Tensor, logw: torch.Tensor, local_lse: torch.Tensor, key, out=None, lse_out=None):
        """gjx_shard_resample_multinomial_step -> (new_rows f32[R][own_n], global LSE record f32[4])"""
        assert rows.shape == (self.rows, self.K) and logw.numel() == self.K and rows.stride(1) == 1
        if out is None:
            out = torch.empty((self.rows, self.own_n), dtype=rows.dtype, device=rows.device)
        if lse_out is None:
            lse_out = torch.empty(4, dtype=torch.float32, device=rows.device)
        check(load().gjx_shard_resample_multinomial_step(self._h, _ptr(logw), _ptr(local_lse), _ptr(rows), rows.stride(0), _ptr(out),
                                                         out.stride(0), key[0], key[1], _ptr(lse_out),
                                                         C.cast(self._info, C.c_void_p), _stream()),
              "gjx_shard_resample_multinomial_step")
        return out, lse_out

    def stats(self) -> dict:
        """counters since creation: resampling steps, children sent / received by this rank, communicator size"""
        o = (C.c_int64 * 4)()
        check(load().gjx_shard_ctx_stats(self._h, C.cast(o, C.c_void_p)), "gjx_shard_ctx_stats")
        return dict(steps=int(o[0]), sent=int(o[1]), received=int(o[2]), rccl_ranks=int(o[3]))

    @property
    def last_info(self) -> dict:
        i = self._info
        return dict(sent=int(i[0]), received=int(i[1]), slot0=int(i[2]), n_valid=int(i[3]))

    def close(self):
        if self._h:
            load().gjx_shard_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RawDevice:
    """__cuda_array_interface__ view of library-owned device memory (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=3, strides=None)
        self._owner = owner


class PeerContext:
    """gjx_peer_ctx: this rank's peer-mapped exchange windows of a collection sharded over the ranks of a process group
    (one process per GPU).  ``rows[p]`` f32[rows][K_local] and ``logw[p]`` f32[K_local] (p = 0, 1) are torch views of the
    DATA window: kernels write their particles there, other ranks read them through hipIpc mappings.  The handles travel
    through one all-gather on ``group`` (any backend) at construction; nothing else ever goes through the host."""

    def __init__(self, K_local: int, rows: int, device=None, group=None, ranks_on_device: int | None = None, flags: int | None = None):
        """``flags`` (A.PEER_VERIFY_ON | A.PEER_DATA_FINE | ...): the context's switches as an argument; None: from the environment
        (GJX_PEER_VERIFY, GJX_PEER_DATA), read by the library at creation"""
        import torch.distributed as dist
        self.group = group
        self.device = _dev(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.K, self.nrows = int(K_local), int(rows)
        self._h = C.c_void_p()
        host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        if ranks_on_device is None:
            ranks_on_device = 1
            if self.world > 1:      # ranks that share this physical device (dry runs on one GPU) split its co-resident capacity
                props = torch.cuda.get_device_properties(self.device)
                ident = str(getattr(props, "uuid", "")) or "index-%d" % self.device.index
                mine = torch.tensor(list(ident.encode()[:64].ljust(64, b" ")), dtype=torch.uint8)
                allv = torch.empty(self.world * 64, dtype=torch.uint8)
                if host_staged:
                    dist.all_gather_into_tensor(allv, mine, group=group)
                else:
                    allv_d = allv.to(self.device)
                    dist.all_gather_into_tensor(allv_d, mine.to(self.device), group=group)
                    allv = allv_d.cpu()
                ranks_on_device = sum(1 for g in range(self.world) if bytes(allv[64 * g:64 * g + 64].tolist()) == bytes(mine.tolist()))
        self.ranks_on_device = int(ranks_on_device)
        def agree(ok: bool) -> bool:
            """every rank learns whether ALL ranks succeeded (a rank that failed alone must not leave the others in a collective)"""
            if self.world == 1:
                return ok
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if host_staged else self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        def gather_bytes(mine: bytes, n: int) -> bytes:
            m = torch.tensor(list(mine), dtype=torch.uint8)
            allv = torch.empty(self.world * n, dtype=torch.uint8)
            if host_staged:
                dist.all_gather_into_tensor(allv, m, group=group)
            else:
                d = allv.to(self.device)
                dist.all_gather_into_tensor(d, m.to(self.device), group=group)
                allv = d.cpu()
            return bytes(allv.tolist())

        err = None
        with torch.cuda.device(self.device):
            if flags is None:
                rc = load().gjx_peer_ctx_create(self.world, self.rank, self.K, self.nrows, self.ranks_on_device, C.byref(self._h))
            else:
                rc = load().gjx_peer_ctx_create_ex(self.world, self.rank, self.K, self.nrows, self.ranks_on_device, int(flags), C.byref(self._h))
            buf = (C.c_uint8 * 128)()
            if rc == 0 and self.world > 1:
                rc = load().gjx_peer_ctx_export(self._h, C.cast(buf, C.c_void_p))
            if rc != 0:
                err = load().gjx_last_error().decode("utf-8", "replace")
            if not agree(rc == 0):
                self._destroy_now()
                raise GjxError("PeerContext: windows could not be created / exported on every rank: %s" % (err or "another rank failed"))
            if self.world > 1:
                allh = gather_bytes(bytes(buf), 128)
                hb = (C.c_uint8 * (128 * self.world)).from_buffer_copy(allh)
                rc = load().gjx_peer_ctx_connect(self._h, C.cast(hb, C.c_void_p))
                if rc != 0:
                    err = load().gjx_last_error().decode("utf-8", "replace")
                if not agree(rc == 0):      # (no peer access between two of the devices, IPC disabled, ...)
                    # agree() is a collective every rank has just left: nobody maps anything new from here on, every rank
                    # (failed or not) tears its context down and raises — no further collective that some ranks would skip
                    self._destroy_now()
                    raise GjxError("PeerContext: the windows could not be mapped on every rank: %s" % (err or "another rank failed"))
            o = (C.c_uint64 * 6)()
            check(load().gjx_peer_ctx_buffers(self._h, C.cast(o, C.c_void_p)), "gjx_peer_ctx_buffers")
        self.rows = [torch.as_tensor(_RawDevice(o[p], (self.nrows, self.K), "<f4", self), device=self.device) for p in (0, 1)]
        self.logw = [torch.as_tensor(_RawDevice(o[2 + p], (self.K,), "<f4", self), device=self.device) for p in (0, 1)]
        self.window_bytes = (int(o[4]), int(o[5]))
        if self.world > 1:
            dist.barrier(group=group)      # every rank has mapped every window before anyone launches into them

    def _destroy_now(self) -> None:
        if self._h:
            load().gjx_peer_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def ssm_filter(self, ssm: A.GjxSsm, key, rng_mode, ys: torch.Tensor, want_ancestors: bool = False, move=None):
        """gjx_ssm_filter_peer -> dict(lse_steps [T][4] global records, x (particles of the last step: a view of the window),
        logw, ancestors?).  ``move=(n_moves, scale)``: gjx_ssm_filter_peer_move (resample-move rejuvenation inside the
        launch), adds accepted_total int64[1] (this rank's accepted moves)."""
        T = ys.shape[0]
        lse = torch.empty((T, 4), dtype=torch.float32, device=self.device)
        anc = torch.empty(self.K, dtype=torch.int32, device=self.device) if want_ancestors else None
        if move is not None:
            acc = torch.zeros(1, dtype=torch.int64, device=self.device)
            check(load().gjx_ssm_filter_peer_move(C.byref(ssm), key[0], key[1], rng_mode, int(T), self._h, _ptr(ys), _ptr(lse), _ptr(anc),
                                                  int(move[0]), float(move[1]), _ptr(acc), _stream()), "gjx_ssm_filter_peer_move")
            return dict(lse_steps=lse, x=self.rows[(T - 1) & 1], logw=self.logw[0], ancestors=anc, accepted_total=acc)
        check(load().gjx_ssm_filter_peer(C.byref(ssm), key[0], key[1], rng_mode, int(T), self._h, _ptr(ys), _ptr(lse), _ptr(anc), _stream()),
              "gjx_ssm_filter_peer")
        return dict(lse_steps=lse, x=self.rows[(T - 1) & 1], logw=self.logw[0], ancestors=anc)

    def scan_filter_prepare(self, cps, T: int, opts=None) -> None:
        """gjx_scan_filter_peer_prepare + a barrier of the context's group: every rank has generated, compiled and loaded the kernels
        of the run before any rank launches into the others' windows (call it once per program structure, on every rank)"""
        info = A.GjxFilterInfo()
        check(load().gjx_scan_filter_peer_prepare_opts(self._h, C.cast(cps, C.c_void_p), int(T), C.byref(opts) if opts is not None else None, C.byref(info)),
              "gjx_scan_filter_peer_prepare")
        self.barrier()

    def barrier(self) -> None:
        """host barrier over the context's ranks (a no-op for one rank)"""
        import torch.distributed as dist
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)

    def scan_filter(self, cps, T: int, key, want_ancestors: bool = False, opts=None):
        """gjx_scan_filter_peer: the bootstrap filter for ANY Scan kernel on this sharded collection — ``cps``: the ctypes array of
        this rank's T step programs (inference/scan_filter.py).  -> dict(lse_steps [T][4] global records, rows (the last step's
        choices: a view of the window), logw, ancestors?, info)"""
        lse = torch.empty((T, 4), dtype=torch.float32, device=self.device)
        anc = torch.empty(self.K, dtype=torch.int32, device=self.device) if want_ancestors else None
        need = load().gjx_workspace_bytes(A.OP_RUN, self.K) + 8 * int(T) + 512
        if getattr(self, "_sf_ws", None) is None or self._sf_ws.numel() < need:
            self._sf_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        info = A.GjxFilterInfo()
        check(load().gjx_scan_filter_peer_opts(self._h, C.cast(cps, C.c_void_p), int(T), key[0], key[1], _ptr(lse), _ptr(anc), _ptr(self._sf_ws),
                                               self._sf_ws.numel(), _stream(), C.byref(opts) if opts is not None else None, C.byref(info)),
              "gjx_scan_filter_peer")
        return dict(lse_steps=lse, rows=self.rows[(T - 1) & 1], logw=self.logw[0], ancestors=anc,
                    info=dict(form=int(info.form), launches=int(info.launches), grid=int(info.grid), tiles_per_block=int(info.tiles_per_block)))

    def resample_gather(self, parity: int, u: float, partials=None, out=None, anc=None, lse_out=None):
        """gjx_peer_resample_gather: logw[parity], rows[parity] -> the children of this rank's slots (one launch).
        ``partials=(run_workspace, n)``: the block pairs of the producing run -> lse_out receives the global record."""
        if out is None:
            out = torch.empty((self.nrows, self.K), dtype=torch.float32, device=self.device)
        lp, npart = (None, 0)
        if partials is not None:
            run_ws, n = partials
            lp, npart = C.c_void_p(run_ws.data_ptr() + 256), int(n)
            if lse_out is None:
                lse_out = torch.empty(4, dtype=torch.float32, device=self.device)
        check(load().gjx_peer_resample_gather(self._h, int(parity), lp, npart, float(u), _ptr(out), out.stride(0), _ptr(anc),
                                              _ptr(lse_out), _stream()), "gjx_peer_resample_gather")
        return out, lse_out

    def status(self) -> int:
        """bit 0: a rendezvous timed out (results undefined), bit 1: a step had zero total weight, bit 2 (GJX_PEER_VERIFY=1): a
        pulled row or source tile did not match its owner's check word / granule total; read and cleared"""
        st = C.c_int32(0)
        check(load().gjx_peer_ctx_status(self._h, C.byref(st), _stream()), "gjx_peer_ctx_status")
        return int(st.value)

    def close(self) -> None:
        if self._h:
            import torch.distributed as dist
            torch.cuda.synchronize(self.device)
            if self.world > 1 and dist.is_initialized():
                dist.barrier(group=self.group)      # nobody unmaps a window a peer's kernel may still read
            self.rows, self.logw = [], []
            load().gjx_peer_ctx_destroy(self._h)
            self._h = C.c_void_p()


def shard_pack(src: torch.Tensor, anc: torch.Tensor, n_valid: int, n_pre: int, n_suf: int) -> torch.Tensor:
    """gjx_shard_pack: the surplus children of this rank's slot run as [n_pre + n_suf][R] messages (one launch)"""
    R = src.shape[0]
    msg = torch.empty((n_pre + n_suf, R), dtype=src.dtype, device=src.device)
    check(load().gjx_shard_pack(_ptr(src), src.stride(0), R, _ptr(anc), int(n_valid), int(n_pre), int(n_suf), _ptr(msg),
                                _stream()), "gjx_shard_pack")
    return msg


def shard_unpack(msg: torch.Tensor, n_lo: int, n_hi: int, dst: torch.Tensor) -> None:
    """gjx_shard_unpack: received [n_lo + n_hi][R] messages into the head / tail columns of dst f32[R][own_n]"""
    R, own_n = dst.shape
    check(load().gjx_shard_unpack(_ptr(msg), int(n_lo), int(n_hi), R, _ptr(dst), dst.stride(0), own_n, _stream()),
          "gjx_shard_unpack")


def pack_rows(src: torch.Tensor, anc: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """out[j][r] = src[r][anc[j]] (gjx_gather_rows_strided)"""
    R, n = src.shape[0], anc.numel()
    if out is None:
        out = torch.empty((n, R), dtype=src.dtype, device=src.device)
    if n:
        check(load().gjx_gather_rows_strided(_ptr(src), src.stride(0), 1, _ptr(anc), n, R, _ptr(out), 1, R, _stream()),
              "gjx_gather_rows_strided")
    return out


def unpack_rows(msg: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[r][j] = msg[j][r] for a [n][R] block; dst is an SoA column window (gjx_gather_rows_strided)"""
    n, R = msg.shape
    if n:
        check(load().gjx_gather_rows_strided(_ptr(msg), 1, R, None, n, R, _ptr(dst), dst.stride(0), 1, _stream()),
              "gjx_gather_rows_strided")


def ssm_step(ssm: A.GjxSsm, key, rng_mode, t, K, x_prev, anc, y, x_out=None, logw=None, lse=None, offset=0,
             K_total=None, ws=None, device=None):
    dev = _dev(device) if x_prev is None else x_prev.device
    if x_out is None:
        x_out = torch.empty((ssm.dx, K), dtype=torch.float32, device=dev)
    if logw is None:
        logw = torch.empty(K, dtype=torch.float32, device=dev)
    if lse is None:
        lse = torch.empty(4, dtype=torch.float32, device=dev)
    if ws is None:
        ws = workspace(A.OP_SSM, K, dev)
    stride = 0 if x_prev is None else x_prev.shape[1]
    check(load().gjx_ssm_step(C.byref(ssm), key[0], key[1], rng_mode, int(t), int(K), int(offset), _ptr(x_prev),
                              stride, _ptr(anc), _ptr(y), _ptr(x_out), _ptr(logw), _ptr(lse), int(K_total or K),
                              _ptr(ws), ws.numel(), _stream()), "gjx_ssm_step")
    return x_out, logw, lse


def ssm_step_move(ssm: A.GjxSsm, key, rng_mode, t, K, x_prev, m_prev, anc, y_prev, y, n_moves, move_scale, x_out=None,
                  m_out=None, logw=None, accepted=None, lse=None, offset=0, K_total=None, ws=None, device=None, x_moved=None):
    """gjx_ssm_step_move: resample-move rejuvenation of x_{t-1} fused in front of the propagate + reweight step"""
    dev = _dev(device) if x_prev is None else x_prev.device
    f32 = torch.float32
    x_out = torch.empty((ssm.dx, K), dtype=f32, device=dev) if x_out is None else x_out
    m_out = torch.empty((ssm.dx, K), dtype=f32, device=dev) if m_out is None else m_out
    logw = torch.empty(K, dtype=f32, device=dev) if logw is None else logw
    lse = torch.empty(4, dtype=f32, device=dev) if lse is None else lse
    if ws is None:
        ws = workspace(A.OP_SSM, K, dev)
    stride = 0 if x_prev is None else x_prev.shape[1]
    check(load().gjx_ssm_step_move(C.byref(ssm), key[0], key[1], rng_mode, int(t), int(K), int(offset), _ptr(x_prev), _ptr(m_prev),
                                   stride, _ptr(anc), _ptr(y_prev), _ptr(y), int(n_moves), float(move_scale), _ptr(x_out),
                                   _ptr(m_out), _ptr(logw), _ptr(accepted), _ptr(x_moved), _ptr(lse), int(K_total or K), _ptr(ws), ws.numel(),
                                   _stream()), "gjx_ssm_step_move")
    return x_out, m_out, logw, lse


def ssm_filter_sharded(ssm: A.GjxSsm, key, rng_mode, ys: torch.Tensor, ctx: "ShardContext", offset: int, bufs=None):
    """gjx_ssm_filter_sharded: this rank's part of the T-step bootstrap filter over a gjx_shard_ctx, looped in C++.
    -> dict(lse_steps [T][4] global records, x (propagated particles of the last step), logw)"""
    dev = ys.device
    T, K = ys.shape[0], ctx.K
    if bufs is None:
        need = load().gjx_workspace_bytes(A.OP_SSM, K)
        bufs = dict(xa=torch.empty((ssm.dx, K), dtype=torch.float32, device=dev),
                    xb=torch.empty((ssm.dx, K), dtype=torch.float32, device=dev),
                    logw=torch.empty(K, dtype=torch.float32, device=dev), lse=torch.empty((T, 4), dtype=torch.float32, device=dev),
                    ws=torch.zeros(need + 64, dtype=torch.uint8, device=dev))
    check(load().gjx_ssm_filter_sharded(C.byref(ssm), key[0], key[1], rng_mode, T, ctx._h, int(offset), _ptr(ys), _ptr(bufs["xa"]),
                                        _ptr(bufs["xb"]), _ptr(bufs["logw"]), _ptr(bufs["lse"]), _ptr(bufs["ws"]),
                                        bufs["ws"].numel(), _stream()), "gjx_ssm_filter_sharded")
    return dict(lse_steps=bufs["lse"], x=bufs["xa"], logw=bufs["logw"], _bufs=bufs)


def hmc(prog: PackedProgram, key, choices: torch.Tensor, eps: float, L: int, stale=False, accept=False, offset=0,
        ws=None):
    """gjx_hmc: in-place HMC move of every chain column.  Returns dict(choices, score, alpha, accepted)."""
    if not choices.is_contiguous():
        raise ValueError("hmc: choices must be a contiguous f32[n_slots][n] tensor (it is moved in place)")
    n = choices.shape[1]
    dev = choices.device
    cp = prog.c_program(dev)
    score = torch.empty(n, dtype=torch.float32, device=dev)
    alpha = torch.empty(n, dtype=torch.float32, device=dev)
    acc = torch.empty(n, dtype=torch.float32, device=dev)
    need = load().gjx_hmc_workspace_bytes(C.byref(cp), n)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    check(load().gjx_hmc(C.byref(cp), key[0], key[1], n, int(offset), float(eps), int(L), int(bool(stale)),
                         int(bool(accept)), _ptr(choices), _ptr(score), _ptr(alpha), _ptr(acc), _ptr(ws), ws.numel(),
                         _stream()), "gjx_hmc")
    return dict(choices=choices, score=score, alpha=alpha, accepted=acc, _ws=ws)


def mh_accept(log_alpha: torch.Tensor, key, rows_cur: torch.Tensor, rows_prop: torch.Tensor, accepted=None, total=None):
    """gjx_mh_accept: the caller-side accept of the reference's move requests (log u < alpha) in place on rows_cur f32[rows][K];
    ``total``: an int64[1] device counter that receives the number of accepted chains.  -> accepted f32[K] (1 / 0)"""
    K = log_alpha.numel()
    if rows_cur.shape != rows_prop.shape or rows_cur.shape[-1] != K or rows_cur.stride(-1) != 1 or rows_prop.stride(-1) != 1 or rows_cur.stride(0) != rows_prop.stride(0):
        raise ValueError("mh_accept: rows_cur / rows_prop must be f32[rows][K] of one layout")
    if accepted is None:
        accepted = torch.empty(K, dtype=torch.float32, device=log_alpha.device)
    check(load().gjx_mh_accept(_ptr(log_alpha), K, key[0], key[1], _ptr(rows_cur), _ptr(rows_prop), rows_cur.stride(0), rows_cur.shape[0],
                               _ptr(accepted), _ptr(total), _stream()), "gjx_mh_accept")
    return accepted


def hmc_engine(prog: PackedProgram) -> int:
    cp = prog.c_program(None)
    return int(load().gjx_hmc_engine(C.byref(cp)))


def score_grad(prog: PackedProgram, choices: torch.Tensor):
    n = choices.shape[1]
    dev = choices.device
    cp = prog.c_program(dev)
    score = torch.empty(n, dtype=torch.float32, device=dev)
    grad = torch.empty_like(choices)
    check(load().gjx_score_grad(C.byref(cp), n, _ptr(choices), _ptr(score), _ptr(grad), _stream()), "gjx_score_grad")
    return score, grad


def ssm_filter(ssm: A.GjxSsm, key, rng_mode, ys: torch.Tensor, K: int, bufs=None, weights: int = A.WEIGHTS_GLOBAL_MAX):
    """gjx_ssm_filter_scheme: the whole T-step bootstrap filter in one native call; ``weights`` picks the fixed-point
    scheme of its resampler (A.WEIGHTS_GLOBAL_MAX | A.WEIGHTS_TILE_SCALED).
    -> dict(lse_steps [T][4], x (final state rows), logw, ancestors)"""
    dev = ys.device
    T = ys.shape[0]
    if bufs is None:
        need = load().gjx_workspace_bytes(A.OP_SSM, K)
        bufs = dict(xa=torch.empty((ssm.dx, K), dtype=torch.float32, device=dev),
                    xb=torch.empty((ssm.dx, K), dtype=torch.float32, device=dev),
                    logw=torch.empty(K, dtype=torch.float32, device=dev), cum=torch.empty(K, dtype=torch.int64, device=dev),
                    anc=torch.empty(K, dtype=torch.int32, device=dev), lse=torch.empty((T, 4), dtype=torch.float32, device=dev),
                    ws=torch.zeros(2 * need + 64, dtype=torch.uint8, device=dev))
    check(load().gjx_ssm_filter_scheme(C.byref(ssm), key[0], key[1], rng_mode, T, int(K), _ptr(ys), _ptr(bufs["xa"]),
                                       _ptr(bufs["xb"]), _ptr(bufs["logw"]), _ptr(bufs["cum"]), _ptr(bufs["anc"]), _ptr(bufs["lse"]),
                                       int(weights), _ptr(bufs["ws"]), bufs["ws"].numel(), _stream()), "gjx_ssm_filter_scheme")
    # the resampling half of the workspace (status word of the co-resident step kernels): second OP_SSM-sized block
    need = load().gjx_workspace_bytes(A.OP_SSM, K)
    return dict(lse_steps=bufs["lse"], x=bufs["xa"] if (T - 1) % 2 == 0 else bufs["xb"], logw=bufs["logw"],
                ancestors=bufs["anc"], _bufs=bufs, _status_ws=bufs["ws"][need:])


def ssm_filter_move(ssm: A.GjxSsm, key, rng_mode, ys: torch.Tensor, K: int, n_moves: int, move_scale: float):
    """gjx_ssm_filter_move: the bootstrap filter WITH resample-move rejuvenation (n_moves random-walk Metropolis steps on
    every resampled particle, tile-scaled resampler) — step 0 and then steps 1 .. T-1 in ONE launch.
    -> dict(lse_steps [T][4], x, logw, ancestors, accepted_total int64[1] (accepted moves over the run), _status_ws), or None
    when the shape / size is outside the one-launch kernel (the caller runs the step-by-step loop)."""
    dev = ys.device
    T = ys.shape[0]
    need = load().gjx_workspace_bytes(A.OP_SSM, K)
    f32 = torch.float32
    b = dict(xa=torch.empty((ssm.dx, K), dtype=f32, device=dev), xb=torch.empty((ssm.dx, K), dtype=f32, device=dev),
             ma=torch.empty((ssm.dx, K), dtype=f32, device=dev), mb=torch.empty((ssm.dx, K), dtype=f32, device=dev),
             logw=torch.empty(K, dtype=f32, device=dev), lw2=torch.empty(K, dtype=f32, device=dev),
             anc=torch.empty(K, dtype=torch.int32, device=dev), lse=torch.empty((T, 4), dtype=f32, device=dev),
             acc=torch.zeros(1, dtype=torch.int64, device=dev), ws=torch.zeros(2 * need + 64, dtype=torch.uint8, device=dev))
    rc = load().gjx_ssm_filter_move(C.byref(ssm), key[0], key[1], rng_mode, T, int(K), _ptr(ys), _ptr(b["xa"]), _ptr(b["xb"]),
                                    _ptr(b["ma"]), _ptr(b["mb"]), _ptr(b["logw"]), _ptr(b["lw2"]), _ptr(b["anc"]), _ptr(b["lse"]),
                                    int(n_moves), float(move_scale), _ptr(b["acc"]), _ptr(b["ws"]), b["ws"].numel(), _stream())
    if rc == A.EUNSUPPORTED:
        return None
    check(rc, "gjx_ssm_filter_move")
    return dict(lse_steps=b["lse"], x=b["xa"] if (T - 1) % 2 == 0 else b["xb"], logw=b["logw"], ancestors=b["anc"],
                accepted_total=b["acc"], _bufs=b, _status_ws=b["ws"][need:])
