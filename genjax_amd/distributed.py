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


def init_from_env(backend: str | None = None) -> tuple[int, int]:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract). -> (rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GJX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


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


class HipBackend:
    """Compute steps on the HIP kernels (device tensors)."""

    def weight_cumsum(self, x, is_log, lse):
        from . import kernels
        return kernels.weight_cumsum(x, is_log, lse)

    def resample_systematic(self, cum, base_total, u, N_total, out_begin, n_out):
        from . import kernels
        return kernels.resample_systematic(cum, base_total, u, N_total, out_begin, n_out, prefill=False)

    def gather_rows(self, src, anc):
        from . import kernels
        return kernels.gather_rows(src, anc)

    def gather_rows_into(self, src, anc, dst, col0):
        """dst[:, col0 : col0 + len(anc)] = src[:, anc]"""
        from . import kernels
        kernels.gather_rows(src, anc, dst[:, col0: col0 + anc.numel()])

    def lse_combine(self, pairs, K_total):
        from . import kernels
        return kernels.lse_combine(pairs, K_total)


def global_lse(local_lse: torch.Tensor, K_total: int, backend=None, group=None) -> torch.Tensor:
    """Combine per-rank {max, sumexp} (first two entries of the 4-float LSE record) into the global
    record {max, sumexp, lse, lse - log K_total}.  One 8-byte all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_lse
    backend = backend or HipBackend()
    pairs = torch.empty(world * 2, dtype=torch.float32, device=local_lse.device)
    _all_gather(pairs, local_lse[:2].contiguous(), group)
    return backend.lse_combine(pairs.view(world, 2), K_total)


def resample_exchange(rows: torch.Tensor, logw: torch.Tensor, lse_global: torch.Tensor, u: float, N_total: int,
                      backend=None, group=None, is_log: bool = True):
    """Systematic resampling of a sharded collection.

    rows f32[R][K_local] (SoA), logw f32[K_local], lse_global the GLOBAL record (its max scales the
    fixed-point weights identically on every rank).  Returns (new_rows f32[R][n_mine], info) where this
    rank ends up with the particles of its output-slot range [slot_off, slot_off + n_mine).
    """
    backend = backend or HipBackend()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = rows.device
    cum, bt_local = backend.weight_cumsum(logw, is_log, lse_global)      # bt_local = {0, local total}
    if world == 1:
        anc = backend.resample_systematic(cum, bt_local, u, N_total, 0, N_total)
        return backend.gather_rows(rows, anc), dict(sent=0, ancestors=anc)
    totals = torch.empty(world, dtype=torch.int64, device=dev)
    _all_gather(totals, bt_local[1:2].contiguous(), group)
    tot_host = [int(t) for t in totals.cpu().tolist()]           # G integers: the ONE host sync of the step
    total_all = sum(tot_host)
    base = sum(tot_host[:rank])
    bt = torch.tensor([base, total_all], dtype=torch.int64, device=dev)
    # Every rank's run of output slots follows from the totals alone (same IEEE-double comb arithmetic as the
    # kernels), so all send/receive counts are known on the host without another round trip.
    bounds = [0]
    for r in range(world):
        bounds.append(slots_below(sum(tot_host[:r + 1]), u, total_all, N_total))
    slot0, n_valid = bounds[rank], bounds[rank + 1] - bounds[rank]
    anc = backend.resample_systematic(cum, bt, u, N_total, slot0, n_valid)

    def overlap(a0, a1, b0, b1):
        return max(0, min(a1, b1) - max(a0, b0))

    owners = [shard(N_total, d, world) for d in range(world)]
    send_counts = [overlap(slot0, slot0 + n_valid, lo, lo + k) for lo, k in owners]
    my_lo, my_k = owners[rank]
    recv_counts = [overlap(bounds[s_], bounds[s_ + 1], my_lo, my_lo + my_k) for s_ in range(world)]
    # Children whose output slot this rank owns are gathered straight into place (SoA, one pass); only the
    # surplus rows that belong to other ranks' slots travel, as small [n][R] blocks in one all-to-all-v.
    R = rows.shape[0]
    new_rows = torch.empty((R, my_k), dtype=rows.dtype, device=dev)
    lo_l, hi_l = max(slot0, my_lo), min(slot0 + n_valid, my_lo + my_k)
    if hi_l > lo_l:
        backend.gather_rows_into(rows, anc[lo_l - slot0: hi_l - slot0], new_rows, lo_l - my_lo)
    send_counts[rank] = 0
    recv_counts[rank] = 0
    parts, pos = [], 0
    for d, (lo, k) in enumerate(owners):
        a0, a1 = max(slot0, lo), min(slot0 + n_valid, lo + k)
        if d != rank and a1 > a0:
            parts.append(backend.gather_rows(rows, anc[a0 - slot0: a1 - slot0]).t())
    send = torch.cat(parts, dim=0).contiguous() if parts else torch.empty((0, R), dtype=rows.dtype, device=dev)
    recv = torch.empty((sum(recv_counts), R), dtype=rows.dtype, device=dev)
    _all_to_all(recv, send, recv_counts, send_counts, group)
    for s_ in range(world):
        if recv_counts[s_]:
            a0 = max(bounds[s_], my_lo) - my_lo
            new_rows[:, a0: a0 + recv_counts[s_]] = recv[pos: pos + recv_counts[s_]].t()
            pos += recv_counts[s_]
    return new_rows, dict(sent=int(send.shape[0]), slot_off=my_lo, ancestors=anc)

