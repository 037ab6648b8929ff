#!/usr/bin/env python3
"""bench.py — whole-job throughput of the ImportanceK hot path on N MI355X GPUs of one node.

A "step" is one pass of the path over one batch of synthetic input, per GPU:
    propagate + reweight + per-block log-sum-exp partials (ONE kernel, gjx_run_program)
      -> [N > 1: 8-byte all-gather of {max, sumexp} + combine]
      -> fixed-point prefix sum of the weights (2 kernels)
      -> systematic ancestors by per-particle slot-range expansion (no search), row gather by ancestor
         [N > 1: search, all-to-all-v of the rows whose slot another rank owns]
on BASELINE.json configs[1]: the Gaussian-mixture Target (C = 8 components, D = 16 latent dims),
ImportanceK with k_particles = 2^20 PER GPU (weak scaling: the collection grows with N and is
sharded by particle index; results are independent of N because streams are indexed globally).

Prints ONE JSON line (rank 0).  value = K_total * steps / wall time of the timed region (max over
ranks), inputs resident in HBM.  roofline.achieved = algorithmic bytes of the propagate+reweight
kernel (4*D + 12 = 76 B per particle, 0 read) / its average duration measured with
HIP events on the launch stream inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

K_PER_GPU = 1 << 20
D, C = 16, 8
ALGO_BYTES_PER_PARTICLE = 4 * D + 12          # z + x[D] + score + log_weight, written once; nothing read
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(prog, K, budget_s=12.0):
    """The CPU restatement (oracle/, OpenMP over all host cores) on a bounded sample of the same step."""
    from oracle import cpu
    threads = cpu.num_threads()
    t0 = time.perf_counter()
    reps = 0
    while True:
        o = cpu.run_program(prog, (0, 1 + reps), K)
        cum, _ = cpu.weight_cumsum(o["logw"], True, o["lse"])
        anc = cpu.resample_systematic(cum, 0.5, K)
        cpu.gather_rows(o["choices"], anc)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or reps >= 64:
            break
    return dict(value=K * reps / dt, unit="particle-steps/s", cores=threads, kind="port",
                sample=f"{reps} steps of K=2^{int(math.log2(K))} particles of the same workload, "
                       f"propagate+reweight+LSE on {threads} OpenMP threads, resample+gather single-threaded")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--k-per-gpu", type=int, default=K_PER_GPU)
    ap.add_argument("--event-samples", type=int, default=16,
                    help="number of timed steps whose propagate+reweight kernel is bracketed with HIP events "
                         "(timing events are not free on ROCm — hundreds of live ones slow every launch — so the "
                         "kernel duration is sampled at evenly spaced steps INSIDE the timed region)")
    args = ap.parse_args()

    from genjax_amd import _abi as A
    from genjax_amd import distributed as DD
    from genjax_amd import kernels
    import helpers as H
    from oracle import closed_form as cf

    rank, world = DD.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = 0 if os.environ.get("GJX_ALL_ON_DEVICE0") else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    K = args.k_per_gpu
    K_total = K * world
    off = rank * K
    prog, g = H.gmm(D=D, C=C)
    assert kernels.program_engine(prog) == 1, "fused mixture kernel not selected"
    ws = kernels.workspace(A.OP_RUN, K, dev)
    ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
    out = kernels.run_program(prog, (0, 1), K, offset=off, K_total=K_total, ws=ws, want_weight=False)
    rows = torch.empty_like(out["choices"])
    cum = torch.empty(K, dtype=torch.int64, device=dev)
    bt = torch.empty(2, dtype=torch.int64, device=dev)
    anc = torch.empty(K, dtype=torch.int32, device=dev)
    lse_rec = torch.empty(4, dtype=torch.float32, device=dev)
    n_part = kernels.run_partials_count(prog, K, off)
    n_samp = max(1, min(args.event_samples, args.steps))
    sample_at = {int(round(j * (args.steps - 1) / max(n_samp - 1, 1))): j for j in range(n_samp)}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_samp)]

    def step(i, timed):
        key = (0, 1 + i)
        j = sample_at.get(i) if timed else None
        if j is not None:
            ev[j][0].record()
        kernels.run_program(prog, key, K, offset=off, K_total=K_total, ws=ws, out=out, want_weight=False,
                            want_lse=(world > 1))
        if j is not None:
            ev[j][1].record()
        u = ((i * 2654435761) % (1 << 23)) / float(1 << 23)
        if world == 1:
            # single GPU: the LSE reduction is finished by the prefix-sum kernels' prologue (no serial tail)
            kernels.weight_cumsum(out["logw"], ws=ws2, out=(cum, bt), partials=(ws, n_part), lse_out=lse_rec, K_total=K_total)
            kernels.resample_gather_systematic(cum, bt, u, K_total, out["choices"], rows, anc=anc)
            return lse_rec
        lse = DD.global_lse(out["lse"], K_total)
        DD.resample_exchange(out["choices"], out["logw"], lse, u, K_total)
        return lse

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i % max(args.steps, 1), False)
    barrier()
    t0 = time.perf_counter()
    lse = None
    for i in range(args.steps):
        lse = step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        achieved = ALGO_BYTES_PER_PARTICLE * K / (kern_ms * 1e-3) / 1e9
        exact = cf.gmm_log_ml(**g)
        lml = float(lse[3])
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r01_pmc_run_gmm.json")
        if os.path.exists(tp):
            traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
        res = dict(
            metric="particle_steps_per_sec", value=K_total * args.steps / dt, unit="particle-steps/s",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload="gmm_c8_d16 ImportanceK: propagate+reweight+LSE, systematic resample, gather "
                                 "(BASELINE.json configs[1])",
                        k_particles_per_gpu=K, k_particles_total=K_total, rng_stream="flat", sharding=f"particles x{world}"),
            roofline=dict(bound="hbm", kernel="gjx::k_run_gmm<FLAT,16,4,256>", achieved=achieved, peak=HBM_PEAK_GBS,
                          unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                          kernel_us=kern_ms * 1e3, algorithmic_bytes_per_launch=ALGO_BYTES_PER_PARTICLE * K,
                          note="kernel is VALU-bound by Threefry-2x32-20 integer work (DESIGN.md); frac is vs HBM"),
            log_ml=lml, log_ml_exact=exact, log_ml_rel_err=abs(lml - exact) / abs(exact),
        )
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(prog, min(K, 1 << 20))
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
