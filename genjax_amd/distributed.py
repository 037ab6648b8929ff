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
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard(K_total: int, rank: int, world: int) -> tuple[int, int]:
    """-> (offset, K_local) of the contiguous shard of rank ``rank`` (remainder to the low ranks)."""
    base, rem = divmod(int(K_total), world)
    k = base + (1 if rank < rem else 0)
    off = rank * base + min(rank, rem)
    return off, k


class HipBackend:
    """Compute steps on the HIP kernels (device tensors)."""

    def weight_cumsum(self, x, is_log, lse):
        from . import kernels
        return kernels.weight_cumsum(x, is_log, lse)

    def resample_systematic(self, cum, base_total, u, N_total, out_begin, n_out):
        from . import kernels
        return kernels.resample_systematic(cum, base_total, u, N_total, out_begin, n_out)

    def gather_rows(self, src, anc):
        from . import kernels
        return kernels.gather_rows(src, anc)

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
    dist.all_gather_into_tensor(pairs, local_lse[:2].contiguous(), group=group)
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
    dist.all_gather_into_tensor(totals, bt_local[1:2].contiguous(), group=group)
    tot_host = [int(t) for t in totals.cpu().tolist()]           # G integers: the only host sync per step
    total_all = sum(tot_host)
    base = sum(tot_host[:rank])
    bt = torch.tensor([base, total_all], dtype=torch.int64, device=dev)
    # output slots whose comb threshold lands on my particles form one contiguous run; bracket it with the
    # same double arithmetic the kernel uses, padded by one slot, and let the -1 markers trim the ends.
    step = float(total_all) / float(N_total)
    if tot_host[rank] == 0 or step == 0.0:
        j0, j1 = 0, 0
    else:
        j0 = max(0, int(base / step - u) - 1)
        j1 = min(N_total, int((base + tot_host[rank]) / step - u) + 2)
    anc = backend.resample_systematic(cum, bt, u, N_total, j0, j1 - j0)
    mine = anc >= 0
    n_valid = int(mine.sum())
    first = int(torch.nonzero(mine)[0]) if n_valid else 0
    anc = anc[first:first + n_valid]
    slot0 = j0 + first                                            # my children occupy slots [slot0, slot0+n_valid)
    # split my run by destination rank (slot owner)
    send_counts = []
    for d in range(world):
        lo, k = shard(N_total, d, world)
        a, b = max(slot0, lo), min(slot0 + n_valid, lo + k)
        send_counts.append(max(0, b - a))
    children = backend.gather_rows(rows, anc)                     # [R][n_valid], slot order
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(c) for c in rc.cpu().tolist()]
    R = rows.shape[0]
    send = children.t().contiguous()                              # [n_valid][R]: splits run along dim 0
    recv = torch.empty((sum(recv_counts), R), dtype=rows.dtype, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
    off_mine, n_mine = shard(N_total, rank, world)
    assert recv.shape[0] == n_mine, (recv.shape, n_mine)
    return recv.t().contiguous(), dict(sent=n_valid - send_counts[rank], slot_off=off_mine, ancestors=anc)
