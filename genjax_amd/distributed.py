"""Particle sharding over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-device code; this is the build's own design:
  * rank r owns the contiguous global particle range [r*K/G, (r+1)*K/G); random streams are indexed
    by the GLOBAL particle index, so every per-particle value is independent of G;
  * propagate + reweight needs no communication;
  * global log-sum-exp: all-gather of one {max, sumexp} pair per rank (8 bytes) + local combine;
  * resampling: all-gather of one uint64 weight total per rank places every rank on the global
    fixed-point weight line; each rank finds the ancestors of the output slots that land on ITS
    particles (contiguous in slot order), and an all-to-all-v moves those rows to the slot owners.
Compute steps are taken from a ``backend`` (the HIP kernels by default) so the exchange logic can be
exercised on CPU tensors in the gloo tests.
"""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist


def _forced() -> bool:
    """GJX_FORCE_DIST=1: take the sharded code path (process group, collectives, plan, all-to-all) even with a
    single rank — how the RCCL plumbing is exercised on a 1-GPU box."""
    return os.environ.get("GJX_FORCE_DIST", "0") == "1"


def init_from_env(backend: str | None = None) -> tuple[int, int]:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract). -> (rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or _forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # RCCL needs a device per local rank: with fewer visible devices (a dry run of the multi-rank path on one GPU) the
            # ranks share devices and gloo carries the host-side collectives
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            enough = torch.cuda.is_available() and torch.cuda.device_count() >= local_world and not os.environ.get("GJX_ALL_ON_DEVICE0")
            backend = os.environ.get("GJX_DIST_BACKEND") or ("nccl" if enough else "gloo")
        if backend == "nccl":
            # (more ranks than visible devices never reaches this branch: RCCL needs one device per rank)
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def all_agree(ok: bool, device=None, group=None) -> bool:
    """True when `ok` is True on EVERY rank (one small all-reduce; True at once without a process group)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    staged = dist.get_backend(group) == "gloo"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if staged else (device or torch.device("cuda", torch.cuda.current_device())))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t[0]))


def _host_staged(group=None) -> bool:
    """gloo moves host memory: device tensors are staged through the CPU (debug / single-GPU dry runs only)."""
    return dist.get_backend(group) == "gloo"


def _all_gather(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    if inp.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None) -> None:
    if inp.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def shard(K_total: int, rank: int, world: int) -> tuple[int, int]:
    """-> (offset, K_local) of the contiguous shard of rank ``rank`` (remainder to the low ranks)."""
    base, rem = divmod(int(K_total), world)
    k = base + (1 if rank < rem else 0)
    off = rank * base + min(rank, rem)
    return off, k


def comb_threshold(j: int, u: float, step: float, total: int) -> int:
    """floor((j + u) * step) clamped below total — the kernels' IEEE-double expression, bit for bit."""
    t = int((float(j) + u) * step)
    return min(t, total - 1) if total > 0 else t


def slots_below(c: int, u: float, total: int, N: int) -> int:
    """#{j in [0, N): comb_threshold(j) < c}  (host twin of slots_below() in csrc/gjx_resample.hip)."""
    if c <= 0 or total <= 0:
        return 0
    if c >= total:
        return N
    step = float(total) / float(N)
    g = int(min(max(math.ceil(c * (float(N) / float(total)) - u), 0), N))
    while g > 0 and comb_threshold(g - 1, u, step, total) >= c:
        g -= 1
    while g < N and comb_threshold(g, u, step, total) < c:
        g += 1
    return g


class HostPlan:
    """Host-computed gjx_shard_plan (same fields), for backends without the plan kernel (CPU dry runs)."""

    def __init__(self, totals: list[int], rank: int, u: float, N_total: int):
        world = len(totals)
        self.total = sum(totals)
        self.base = sum(totals[:rank])
        self.bounds = [slots_below(sum(totals[:r]), u, self.total, N_total) for r in range(world + 1)]
        self.slot0, self.n_valid = self.bounds[rank], self.bounds[rank + 1] - self.bounds[rank]
        self.own_lo, self.own_n = shard(N_total, rank, world)
        self.keep_lo = max(self.slot0, self.own_lo)
        self.keep_hi = max(self.keep_lo, min(self.slot0 + self.n_valid, self.own_lo + self.own_n))
        self.status = 0 if self.total > 0 else 1

    def wait(self):
        return self


class HipBackend:
    """Compute steps on the HIP kernels (device tensors).  Internal temporaries (zeroed workspace, prefix sums,
    plan) are kept per (K, device); everything handed back to the caller is freshly allocated."""

    def __init__(self):
        self._plan = None
        self._tmp = {}

    def _buffers(self, K, dev):
        from . import _abi as A
        from . import kernels
        b = self._tmp.get((K, dev))
        if b is None:
            b = (kernels.workspace(A.OP_RESAMPLE, K, dev), torch.empty(K, dtype=torch.int64, device=dev),
                 torch.empty(2, dtype=torch.int64, device=dev))
            self._tmp = {(K, dev): b}
        return b

    def weight_cumsum(self, x, is_log, lse=None, pairs=None, K_total=None):
        """-> (cum, {0, local total}, global LSE record or None).  ``pairs`` = the all-gathered per-rank
        {max, sumexp}: the kernels' prologue reduces them, so no separate combine launch."""
        from . import kernels
        ws, cum, bt = self._buffers(x.numel(), x.device)
        rec = None
        if pairs is not None:
            rec = torch.empty(4, dtype=torch.float32, device=x.device)
            kernels.weight_cumsum(x, True, pairs=pairs, lse_out=rec, K_total=K_total, ws=ws, out=(cum, bt))
        else:
            kernels.weight_cumsum(x, is_log, lse, ws=ws, out=(cum, bt))
        return cum, bt, rec

    def plan(self, totals, rank, u, N_total):
        from . import kernels
        if self._plan is None or self._plan.dev.device != totals.device:
            self._plan = kernels.ShardPlan(totals.device)
        return self._plan.build(totals, rank, u, N_total)

    def shard_resample(self, cum, plan, u, N_total, rows, own_n):
        from . import kernels
        return kernels.shard_resample(cum, plan, u, N_total, rows, own_n)

    def pack(self, rows, anc, n_valid, n_pre, n_suf):
        from . import kernels
        return kernels.shard_pack(rows, anc, n_valid, n_pre, n_suf)

    def unpack(self, msg, n_lo, n_hi, dst):
        from . import kernels
        kernels.shard_unpack(msg, n_lo, n_hi, dst)

    def resample_systematic(self, cum, base_total, u, N_total, out_begin, n_out):
        from . import kernels
        return kernels.resample_systematic(cum, base_total, u, N_total, out_begin, n_out, prefill=False)

    def resample_multinomial(self, cum, base_total, key, N_total):
        from . import kernels
        return kernels.resample_multinomial(cum, base_total, key, N_total)

    def gather_rows(self, src, anc):
        from . import kernels
        return kernels.gather_rows(src, anc)

    def lse_combine(self, pairs, K_total):
        from . import kernels
        return kernels.lse_combine(pairs, K_total)


_default_backend = None


def _backend(backend):
    global _default_backend
    if backend is not None:
        return backend
    if _default_backend is None:
        _default_backend = HipBackend()
    return _default_backend


def gather_lse_pairs(local_lse: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of the per-rank {max, sumexp} (first two floats of an LSE record): f32[world][2]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not _forced():
        return local_lse[:2].reshape(1, 2)
    pairs = torch.empty(world * 2, dtype=torch.float32, device=local_lse.device)
    _all_gather(pairs, local_lse[:2].contiguous(), group)
    return pairs.view(world, 2)


def global_lse(local_lse: torch.Tensor, K_total: int, backend=None, group=None) -> torch.Tensor:
    """Combine per-rank {max, sumexp} (first two entries of the 4-float LSE record) into the global
    record {max, sumexp, lse, lse - log K_total}.  One 8-byte all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_lse
    return _backend(backend).lse_combine(gather_lse_pairs(local_lse, group), K_total)


def resample_exchange(rows: torch.Tensor, logw: torch.Tensor, lse_global, u: float, N_total: int,
                      backend=None, group=None, is_log: bool = True, pairs=None):
    """Systematic resampling of a sharded collection.

    rows f32[R][K_local] (SoA), logw f32[K_local].  The fixed-point weight scale is the GLOBAL maximum, given
    either as ``lse_global`` (the combined record) or as ``pairs`` (gather_lse_pairs: the combine then happens
    in the prefix-sum kernels' prologue and the record comes back in info["lse"]).
    Returns (new_rows f32[R][own_n], info): this rank ends up with the particles of its output-slot range.

    Stream order per rank (2 collectives + 1 all-to-all-v, one host wait that does not drain the GPU):
      prefix sums -> all-gather totals -> plan (device + pinned host) -> expand ancestors -> gather kept children
      [host reads the plan while those run] -> pack surplus -> all-to-all-v -> unpack.
    """
    backend = _backend(backend)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = rows.device
    R = rows.shape[0]
    cum, bt_local, rec = backend.weight_cumsum(logw, is_log, lse_global, pairs=pairs, K_total=N_total)
    if world == 1 and not _forced():
        anc = backend.resample_systematic(cum, bt_local, u, N_total, 0, N_total)
        return backend.gather_rows(rows, anc), dict(sent=0, ancestors=anc, lse=rec)
    totals = torch.empty(world, dtype=torch.int64, device=dev)
    _all_gather(totals, bt_local[1:2].contiguous(), group)
    own_lo, own_n = shard(N_total, rank, world)
    plan = backend.plan(totals, rank, u, N_total)
    anc, new_rows = backend.shard_resample(cum, plan, u, N_total, rows, own_n)
    p = plan.wait()                                   # the expansion and the local gather are already queued
    if p.status:
        raise ValueError("resample_exchange: all weights are zero")
    b = list(p.bounds[: world + 1])
    slot0, n_valid = int(p.slot0), int(p.n_valid)

    def overlap(a0, a1, b0, b1):
        return max(0, min(a1, b1) - max(a0, b0))

    owners = [shard(N_total, d, world) for d in range(world)]
    send_counts = [0 if d == rank else overlap(slot0, slot0 + n_valid, lo, lo + k) for d, (lo, k) in enumerate(owners)]
    recv_counts = [0 if s_ == rank else overlap(b[s_], b[s_ + 1], own_lo, own_lo + own_n) for s_ in range(world)]
    # My slot run minus the window I keep is a prefix piece (slots owned by lower ranks) and a suffix piece
    # (higher ranks); in slot order both are already sorted by destination, so the message buffer is two packs.
    own_hi, run_hi = own_lo + own_n, slot0 + n_valid
    n_pre = max(0, min(run_hi, own_lo) - slot0)
    n_suf = max(0, run_hi - max(slot0, own_hi))
    assert n_pre == sum(send_counts[:rank]) and n_suf == sum(send_counts[rank + 1:]), (n_pre, n_suf, send_counts)
    send = backend.pack(rows, anc, n_valid, n_pre, n_suf)
    recv = torch.empty((sum(recv_counts), R), dtype=rows.dtype, device=dev)
    _all_to_all(recv, send, recv_counts, send_counts, group)
    # received blocks arrive in source-rank order = slot order: lower ranks fill the head of my slot range,
    # higher ranks its tail
    n_lo = max(0, min(own_hi, slot0) - own_lo)
    n_hi = max(0, own_hi - max(own_lo, run_hi))
    assert n_lo == sum(recv_counts[:rank]) and n_hi == sum(recv_counts[rank + 1:]), (n_lo, n_hi, recv_counts)
    if n_lo + n_hi:
        backend.unpack(recv, n_lo, n_hi, new_rows)
    return new_rows, dict(sent=int(send.shape[0]), slot_off=own_lo, ancestors=anc, lse=rec, bounds=b)


def resample_exchange_multinomial(rows: torch.Tensor, logw: torch.Tensor, key, N_total: int, pairs, backend=None, group=None):
    """Multinomial resampling of a sharded collection over torch.distributed (the reference transport of
    gjx_shard_resample_multinomial_step; gloo dry runs and CPU tests).  Slot j draws from the hash of its global index;
    every rank finds the slots that land on its particles and an all-to-all-v carries each child to the slot's owner.
    -> (new_rows f32[R][own_n], info)"""
    backend = _backend(backend)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = rows.device
    R = rows.shape[0]
    cum, bt_local, rec = backend.weight_cumsum(logw, True, None, pairs=pairs, K_total=N_total)
    totals = torch.empty(world, dtype=torch.int64, device=dev)
    if world > 1 or _forced():
        _all_gather(totals, bt_local[1:2].contiguous(), group)
    else:
        totals.copy_(bt_local[1:2])
    tl = [int(t) for t in totals.tolist()]
    if sum(tl) <= 0:
        raise ValueError("resample_exchange_multinomial: all weights are zero")
    bt = torch.tensor([sum(tl[:rank]), sum(tl)], dtype=torch.int64, device=dev)
    anc = backend.resample_multinomial(cum, bt, key, N_total)            # int32[N_total], -1 where the slot is not mine
    sel = torch.nonzero(anc >= 0).flatten()                              # slots, ascending = sorted by owner
    lows = torch.tensor([shard(N_total, d, world)[0] for d in range(world)] + [N_total], dtype=torch.int64, device=dev)
    owner = torch.bucketize(sel, lows[1:], right=True)
    counts = torch.bincount(owner, minlength=world).to(torch.int64)
    matrix = torch.empty(world * world, dtype=torch.int64, device=dev)
    if world > 1 or _forced():
        _all_gather(matrix, counts, group)
    else:
        matrix.copy_(counts)
    M = matrix.view(world, world).tolist()
    own_lo, own_n = shard(N_total, rank, world)
    new_rows = torch.empty((R, own_n), dtype=rows.dtype, device=dev)
    picked = backend.gather_rows(rows, anc[sel].contiguous()) if sel.numel() else torch.empty((R, 0), dtype=rows.dtype, device=dev)
    mine = owner == rank
    new_rows[:, (sel[mine] - own_lo)] = picked[:, mine]
    send_counts = [0 if d == rank else int(M[rank][d]) for d in range(world)]
    recv_counts = [0 if s_ == rank else int(M[s_][rank]) for s_ in range(world)]
    if world > 1:
        away = ~mine
        slot_local = (sel[away] - lows[owner[away]]).to(torch.int32)
        send = torch.cat([picked[:, away].t().contiguous(), slot_local.view(torch.float32).unsqueeze(1)], dim=1).contiguous()
        recv = torch.empty((sum(recv_counts), R + 1), dtype=rows.dtype, device=dev)
        _all_to_all(recv, send, recv_counts, send_counts, group)
        if recv.shape[0]:
            slots = recv[:, R].contiguous().view(torch.int32).to(torch.int64)
            new_rows[:, slots] = recv[:, :R].t()
    return new_rows, dict(sent=sum(send_counts), received=sum(recv_counts), lse=rec, ancestors=anc)


class ShardedResampler:
    """Systematic resampling of a collection sharded over the ranks of a group, for a fixed shape
    (K_local particles x ``rows`` SoA rows per rank, N_total output slots).

    transport "rccl": the whole exchange is one host call (gjx_shard_resample_step: RCCL all-gathers and grouped
    send/recv issued from C on the caller's stream, local gather overlapped on a second stream).
    transport "torch": the same kernels driven from Python with torch.distributed collectives
    (resample_exchange) — the only choice for gloo dry runs and CPU tests.
    "auto" (default, or GJX_SHARD_TRANSPORT) takes "rccl" when the group's backend is nccl, after a one-off
    self-check that both transports return identical bits on a synthetic uneven collection; otherwise "torch".
    """

    def __init__(self, K_local: int, rows: int, N_total: int, device, transport: str | None = None, backend=None,
                 group=None):
        self.K, self.rows, self.N_total, self.group, self.backend = int(K_local), int(rows), int(N_total), group, backend
        self.device = torch.device(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        transport = transport or os.environ.get("GJX_SHARD_TRANSPORT", "auto")
        if transport not in ("auto", "rccl", "torch"):
            raise ValueError(f"unknown transport {transport!r}")
        self.ctx = None
        nccl = dist.is_initialized() and dist.get_backend(group) == "nccl" and self.device.type == "cuda"
        if transport == "rccl" and not nccl:
            raise ValueError('transport "rccl" needs an initialised nccl (RCCL) process group')
        if transport != "torch" and nccl:
            self._open_rccl(strict=(transport == "rccl"))
        self.transport = "rccl" if self.ctx is not None else "torch"

    def _open_rccl(self, strict: bool) -> None:
        from . import kernels
        from ._lib import GjxError
        uid = torch.zeros(128, dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            uid.copy_(torch.tensor(list(kernels.rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0, group=self.group)
        err = None
        try:
            ctx = kernels.ShardContext(bytes(uid.cpu().tolist()), self.world, self.rank, self.K, self.rows, self.N_total)
        except GjxError as e:        # symmetric failures only (missing library / symbols); comm init itself is collective
            ctx, err = None, e
        ok = torch.tensor([1 if ctx is not None else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 1:
            self.ctx = ctx
            try:
                verdict = 1 if self._self_check() else 0
            except Exception as e:           # the reference transport itself failed: nothing to compare against
                import warnings
                warnings.warn(f"ShardedResampler: self-check of the RCCL transport could not run ({e!r}); keeping RCCL")
                verdict = 2
            ok = torch.tensor([verdict], dtype=torch.int32, device=self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) >= 1:
                return
            err = "the RCCL transport and the torch.distributed transport disagree on the self-check collection"
        if ctx is not None:
            ctx.close()
        self.ctx = None
        if strict:
            raise RuntimeError(f"ShardedResampler: RCCL transport unavailable: {err}")
        import warnings
        warnings.warn(f"ShardedResampler: falling back to the torch.distributed transport ({err})")

    def _self_check(self) -> bool:
        """Both transports on a deterministic collection whose weights lean towards the high ranks (so children
        cross rank boundaries in both directions of the slot order)."""
        off, k = shard(self.K * self.world, self.rank, self.world)
        i = torch.arange(off, off + self.K, device=self.device, dtype=torch.float32)
        logw = 2.0 * torch.sin(i * 0.37) + 3.0 * (i / float(self.K * self.world)) ** 2 - (3.0 if self.rank % 2 else 0.0)
        rows = (i[None, :] + torch.arange(self.rows, device=self.device, dtype=torch.float32)[:, None] * 0.25).contiguous()
        from . import kernels
        local = kernels.logsumexp(logw, self.K * self.world)
        a, rec_a = self.ctx.step(rows, logw, local, 0.4321)
        b, info = resample_exchange(rows, logw, None, 0.4321, self.N_total, self.backend, self.group,
                                    pairs=gather_lse_pairs(local, self.group))
        torch.cuda.synchronize()
        return bool(torch.equal(a, b)) and bool(torch.equal(rec_a, info["lse"]))

    def step(self, rows: torch.Tensor, logw: torch.Tensor, local_lse: torch.Tensor, u: float):
        """-> (new_rows f32[rows][own_n], global LSE record f32[4])"""
        if self.ctx is not None:
            return self.ctx.step(rows, logw, local_lse, u)
        new_rows, info = resample_exchange(rows, logw, None, u, self.N_total, self.backend, self.group,
                                           pairs=gather_lse_pairs(local_lse, self.group))
        return new_rows, info["lse"]

    def step_multinomial(self, rows: torch.Tensor, logw: torch.Tensor, local_lse: torch.Tensor, key):
        """multinomial resampling with the slots' uniforms from ``key`` -> (new_rows f32[rows][own_n], global LSE record)"""
        if self.ctx is not None:
            return self.ctx.step_multinomial(rows, logw, local_lse, key)
        new_rows, info = resample_exchange_multinomial(rows, logw, key, self.N_total, gather_lse_pairs(local_lse, self.group),
                                                       self.backend, self.group)
        return new_rows, info["lse"]

    def stats(self) -> dict:
        """transport, communicator size and (RCCL transport) children sent / received by this rank since creation"""
        d = dict(transport=self.transport, ranks=self.world)
        if self.ctx is not None:
            d.update(self.ctx.stats())
        return d

    def close(self) -> None:
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None


# ---------------------------------------------------------------------------------------------------------------
# Peer-mapped exchange (csrc/gjx_peer.hip): the sharded kernels talk to each other through hipIpc-mapped windows, no
# host in the loop.  Opened once per process after a self-check; the collective transports above stay as the fallback.
# ---------------------------------------------------------------------------------------------------------------
_peer_verdict: dict = {}
_peer_report: dict = {}          # what the self-check saw, per (device, group): bench.py prints it for every rank

SELF_CHECK_TILES = 64            # tiles of 1024 particles per rank in the self-check (enough for rows to cross every pair of ranks)


def _peer_self_check(device, group) -> bool:
    """First contact with the fabric: a sharded tile-scaled filter through the peer windows — 64 tiles per rank, 6 steps, CHECK WORDS
    ON (A.PEER_VERIFY_ON: every pulled row against its owner's word, every re-scanned tile against its granule) — against the same
    filter run unsharded on this rank: this rank's slice of the particles and log-weights must match bit for bit, no rendezvous may
    time out and no check may fail.  Collective: every rank gets the same verdict; the details stay in ``peer_report``."""
    import numpy as np
    from . import _abi as A
    from . import kernels
    from .inference.pf import LinearGaussianSSM
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    host = _host_staged(group) if dist.is_initialized() else True
    Kl, T = SELF_CHECK_TILES * 1024, 6
    rep = dict(rank=rank, world=world, verify=True, tiles_per_rank=SELF_CHECK_TILES, steps=T, status=None, bit_identical=None, error=None)
    _peer_report[(str(torch.device(device)), id(group))] = rep
    ok = 0
    pc = None
    try:
        pc = kernels.PeerContext(Kl, 4, device, group, flags=A.PEER_VERIFY_ON)
    except Exception as e:                 # symmetric: PeerContext raises on every rank or on none
        import warnings
        warnings.warn(f"peer-mapped exchange unavailable: {e}")
        rep["error"] = repr(e)
        return False
    try:
        rep["ranks_on_this_device"] = pc.ranks_on_device
        th = 0.3 + 0.1 * np.arange(2)
        Am = np.zeros((4, 4), np.float32)
        for i, t in enumerate(th):
            c, s_ = 0.9 * np.cos(t), 0.9 * np.sin(t)
            Am[2 * i:2 * i + 2, 2 * i:2 * i + 2] = [[c, -s_], [s_, c]]
        ssm = LinearGaussianSSM(Am, 0.5, 2.0)
        ys = torch.as_tensor(np.random.default_rng(5).standard_normal((T, 4)).astype(np.float32), device=device)
        cs = ssm.c_struct(device)
        got = pc.ssm_filter(cs, (0, 77), A.RNG_FLAT, ys)
        torch.cuda.synchronize(device)
        st = pc.status()
        ref = kernels.ssm_filter(cs, (0, 77), A.RNG_FLAT, ys, Kl * world, weights=A.WEIGHTS_TILE_SCALED)
        sl = slice(rank * Kl, (rank + 1) * Kl)
        same = torch.equal(got["x"], ref["x"][:, sl]) and torch.equal(got["logw"], ref["logw"][sl])
        rep["status"], rep["bit_identical"] = int(st), bool(same)
        ok = 1 if (same and st == 0) else 0
    except Exception as e:
        import warnings
        warnings.warn(f"peer-mapped exchange: self-check raised {e!r}")
        rep["error"] = repr(e)
        ok = 0
    if world > 1:
        t = torch.tensor([ok], dtype=torch.int32, device="cpu" if host else device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        ok = int(t.item())
    rep["verdict_all_ranks"] = bool(ok)
    try:
        pc.close()
    except Exception:
        pass
    return bool(ok)


def peer_report(device, group=None) -> dict:
    """what this rank's self-check of the peer-mapped exchange saw ({} when it never ran: GJX_PEER=0 / 1)"""
    return dict(_peer_report.get((str(torch.device(device)), id(group)), {}))


def gather_objects(obj, group=None) -> list:
    """every rank's (picklable) object on every rank, in rank order — the per-rank diagnostics of a multi-GPU bench line"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def peer_available(device, group=None) -> bool:
    """True when the peer-mapped exchange works between the ranks of ``group`` (checked once per process and device;
    GJX_PEER=0 switches it off, GJX_PEER=1 skips the check)."""
    mode = os.environ.get("GJX_PEER", "auto")
    if mode == "0":
        return False
    key = (str(torch.device(device)), id(group))
    if key not in _peer_verdict:
        _peer_verdict[key] = True if mode == "1" else _peer_self_check(torch.device(device), group)
        if not _peer_verdict[key]:
            import warnings
            warnings.warn("peer-mapped exchange failed its self-check: sharded collections fall back to the collective transports")
    return _peer_verdict[key]
