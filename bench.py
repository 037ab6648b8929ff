#!/usr/bin/env python3
"""bench.py — whole-job throughput of the inference hot path on N MI355X GPUs of one node.

Default workload (the one BASELINE.json's metric is quoted on, configs[1]): the Gaussian-mixture
Target (C = 8 components, D = 16 latent dims), ImportanceK with k_particles = 2^20 PER GPU (weak
scaling: the collection grows with N and is sharded by particle index; results are independent of N
because random streams are indexed by the global particle index).  A "step" is one pass of the path
over one batch of synthetic input, per GPU:
    propagate + reweight + per-block log-sum-exp partials (ONE kernel, gjx_run_program)
      -> [N > 1: 8-byte all-gather of {max, sumexp} + combine]
      -> resampling indices: fixed-point prefix sum of the weights + systematic ancestors by per-particle
         slot-range expansion (no search) — ONE co-resident kernel on one GPU (its prologue also finishes the
         LSE); with N > 1 the prefix sum, an 8-byte all-gather of totals and the expansion are separate launches
      -> row gather by ancestor [N > 1: all-to-all-v of the rows whose output slot another rank owns]

Prints ONE JSON line (rank 0).  value = K_total * steps / wall time of the timed region (max over
ranks), inputs resident in HBM.  roofline.achieved = algorithmic bytes of the propagate+reweight
kernel (4*D + 12 = 76 B per particle, 0 read) / its average duration measured with
HIP events on the launch stream inside the timed region.

Other §8 rows can be measured with --workload ssm (config 3: bootstrap filter, T = 256, K = 2^18) and
--workload hmc (config 5: 2^16 chains x L = 1000 leapfrog steps); they print the same JSON shape.
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

import numpy as np
import torch
import torch.distributed as dist

K_PER_GPU = 1 << 20
D, C = 16, 8
ALGO_BYTES_PER_PARTICLE = 4 * D + 12          # z + x[D] + score + log_weight, written once; nothing read
MIN_TIMED_S = 0.05                             # the timed region is repeated until this much has been timed (median reported)
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3                       # vector / f32-MFMA peak


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", "closed_form.json")) as f:
        return json.load(f)[name]


# ---------------------------------------------------------------------------------------------
# CPU baselines: the oracle (oracle/, a C restatement with OpenMP) on a bounded sample of the same step.
# This is the only place bench.py touches oracle/.
# ---------------------------------------------------------------------------------------------
def cpu_baseline_gmm(prog, K, budget_s=12.0):
    """the same step (propagate + reweight + LSE, prefix sum, systematic ancestors, gather) through the C oracle: on all
    host cores (every stage OpenMP-parallel) and on ONE thread"""
    from oracle import cpu
    flags = cpu.use_fast_build()                # the timed baseline runs the oracle compiled -O3 -march=native on this host

    def run(budget):
        t0 = time.perf_counter()
        reps = 0
        while True:
            o = cpu.run_program(prog, (0, 1 + reps), K)
            cum, _ = cpu.weight_cumsum(o["logw"], True, o["lse"])
            anc = cpu.resample_systematic(cum, 0.5, K)
            cpu.gather_rows(o["choices"], anc)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or reps >= 64:
                return reps, dt

    threads = baseline_threads(cpu)
    reps, dt = run(budget_s)
    cpu.set_threads(1)
    K1 = min(K, 1 << 17)
    t0 = time.perf_counter()
    o = cpu.run_program(prog, (0, 1), K1)
    cum, _ = cpu.weight_cumsum(o["logw"], True, o["lse"])
    cpu.gather_rows(o["choices"], cpu.resample_systematic(cum, 0.5, K1))
    dt1 = time.perf_counter() - t0
    cpu.set_threads(threads)
    cpu.use_checker_build()
    return dict(value=K * reps / dt, unit="particle-steps/s", cores=threads, kind="port", build=flags,
                sample=f"{reps} steps of K=2^{int(math.log2(K))} particles of the same workload on {threads} OpenMP threads "
                       f"(all stages parallel; {usable_cpus()} usable CPUs: affinity mask capped by the cgroup quota)",
                single_thread=dict(value=K1 / dt1, unit="particle-steps/s", cores=1, sample=f"1 step of K=2^{int(math.log2(K1))}"))


def usable_cpus() -> int:
    """CPUs this process can actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows 256 logical
    CPUs and grants 16 CPUs of time: 128 OpenMP threads there run 2.5x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                 # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def baseline_threads(cpu) -> int:
    """OpenMP threads of the CPU baseline = usable CPUs (never more than the runtime's own default)"""
    n = max(1, min(usable_cpus(), cpu.num_threads()))
    cpu.set_threads(n)
    return n


def cpu_baseline_ssm(s, K, T, budget_s=12.0):
    from genjax_amd import core
    from oracle import cpu
    flags = cpu.use_fast_build()
    threads = baseline_threads(cpu)
    t0 = time.perf_counter()
    key = core.key(1)
    x = lw = lse = None
    steps = 0
    for t in range(T):
        key = core.fold_in(key, t)
        kp, _ = core.split(key)
        anc = None
        if t > 0:
            cum, _ = cpu.weight_cumsum(lw, True, lse)
            anc = cpu.resample_systematic(cum, 0.5, K)
        x, lw, lse = cpu.ssm_step(s["A"], None, s["q"], s["r"], 1.0, kp, 0, t, K, x, anc, s["y"][t])
        steps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cpu.use_checker_build()
    return dict(value=K * steps / dt, unit="particle-steps/s", cores=threads, kind="port", build=flags,
                sample=f"first {steps} of {T} filter steps at K=2^{int(math.log2(K))} on {threads} OpenMP threads (all stages parallel)")


def cpu_baseline_hmc(prog, P, L, budget_s=12.0):
    from oracle import cpu
    flags = cpu.use_fast_build()
    threads = baseline_threads(cpu)
    n = max(threads * 4, 256)
    ch = (np.random.default_rng(0).standard_normal((P + 1, n)) * 0.1).astype(np.float32)
    Lc = min(L, 50)
    t0 = time.perf_counter()
    reps = 0
    while True:
        cpu.hmc(prog, (1, 2 + reps), ch, 0.01, Lc, False, True)
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cpu.use_checker_build()
    return dict(value=n * Lc * reps / dt, unit="chain-leapfrogs/s", cores=threads, kind="port", build=flags,
                sample=f"{reps} moves of {n} chains x {Lc} leapfrog steps on {threads} OpenMP threads")


# ---------------------------------------------------------------------------------------------

def _exchange_record(exch, dev, status_word=None):
    """The first multi-GPU line must diagnose itself: EVERY rank's view — transport chosen, verdict and details of the peer self-check
    (64 tiles per rank, check words on: genjax_amd/distributed.py), status words, ranks of the communicator — gathered to all ranks
    (collective: every rank calls it) and put under config.exchange_stats.per_rank by rank 0."""
    import torch.distributed as dist
    from genjax_amd import distributed as DD
    mine = dict(exch)
    mine["rank"] = dist.get_rank() if dist.is_initialized() else 0
    mine["device"] = str(dev)
    mine["status_word"] = status_word
    mine["peer_self_check"] = DD.peer_report(dev) or "not run (one rank, or GJX_PEER=0 / 1)"
    if dist.is_initialized():
        mine["process_group"] = dict(backend=dist.get_backend(), ranks=dist.get_world_size())
    per_rank = DD.gather_objects(mine)
    out = dict(exch)
    out["per_rank"] = per_rank
    out["transports_agree"] = len({r.get("transport") for r in per_rank}) == 1
    out["any_status_bit"] = any(bool(r.get("status_word")) for r in per_rank)
    return out


def run_gmm(args, rank, world, dev):
    from genjax_amd import _abi as A
    from genjax_amd import distributed as DD
    from genjax_amd import kernels, workloads

    K = args.k_per_gpu
    K_total = K * world
    off = rank * K
    prog, g = workloads.gmm_program(D=D, C=C)
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
    # per sampled step: HIP events attached to the kernel's own dispatch (gjx_run_opts.start_event / stop_event: begin / end of the
    # kernel, what rocprofv3's kernel trace reports) on even samples, and a plain event pair recorded around the call
    # (which also times the dispatch hand-offs on both sides) on odd ones
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_samp)]
    timers = [kernels.DispatchTimer() for _ in range(n_samp)]

    sharded = world > 1 or os.environ.get("GJX_FORCE_DIST", "0") == "1"
    # one GPU: the whole step is one launch when the grid is co-resident (K % 1024 == 0, K / 1024 blocks fit the device)
    step_out = dict(choices=out["choices"], score=out["score"], logw=out["logw"], rows=rows, ancestors=anc, lse=lse_rec, _ws=ws)
    fused_step = False
    # (measured slower than the three launches on MI355X — 77-90 us vs 70 us per step, DESIGN.md §5 — so it is opt-in)
    if not sharded and os.environ.get("GJX_FUSED_STEP", "0") == "1":
        try:
            kernels.importance_step(prog, (0, 1), K, 0.5, out=step_out, allow_fallback=False)
            fused_step = True
        except Exception:
            fused_step = False
    # two launches per step: propagate+reweight, then resampling + gather in one co-resident launch (K <= 2^20 on a full
    # MI355X; GJX_THREE_LAUNCH=1 keeps resample_indices + gather_rows)
    two_launch = False
    if not sharded and not fused_step and os.environ.get("GJX_THREE_LAUNCH", "0") != "1":
        try:
            kernels.run_program(prog, (0, 1), K, ws=ws, out=out, want_weight=False, want_lse=False)
            kernels.resample_gather(out["logw"], 0.5, out["choices"], partials=(ws, n_part), lse_out=lse_rec, K_total=K_total,
                                    out=rows, ws=ws2, allow_fallback=False)
            two_launch = True
        except Exception:
            two_launch = False
    # sharded: peer-mapped windows (the producing kernel writes its particles into the context's buffers, alternating
    # from step to step; one resampling launch per rank, no collective call) — or, if the ranks cannot map each other's
    # memory, the collective transport (RCCL all-gathers + host-sized all-to-all-v)
    peer = None
    if sharded and K % 1024 == 0 and os.environ.get("GJX_SHARD_TRANSPORT", "auto") in ("auto", "peer") and DD.peer_available(dev):
        peer = kernels.PeerContext(K, out["choices"].shape[0], dev)
        peer_out = [dict(choices=peer.rows[p], score=out["score"], logw=peer.logw[p], _ws=ws) for p in (0, 1)]
        peer_calls = [0]                         # buffer parity: strictly alternating from call to call on every rank
        # one trial step: the one-launch resampler needs K / 1024 blocks co-resident per rank (ranks that SHARE a device —
        # a dry run of the multi-rank path on one GPU — split its capacity); if any rank cannot, all take the collective transport
        from genjax_amd._lib import GjxError
        # (the ranks enter their first rendezvous together: kernels are loaded by the warm-up above, the barrier takes the process
        # start-up skew out — a rank that waits longer than its poll budget for a late peer would report a time-out, and the run
        # would fall back to the collective transport for no reason)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        try:
            for _ in range(4):
                par = peer_calls[0] & 1
                peer_calls[0] += 1
                kernels.run_program(prog, (0, 1), K, offset=off, K_total=K_total, ws=ws, out=peer_out[par], want_weight=False, want_lse=False)
                peer.resample_gather(par, 0.5, partials=(ws, n_part), out=rows, lse_out=lse_rec)
            torch.cuda.synchronize()
            ok = not (peer.status() & 1)         # a rendezvous between the ranks timed out: their grids do not run side by side
        except GjxError:
            ok = False
        if not DD.all_agree(ok, dev):
            peer.close()
            peer = None
    resampler = DD.ShardedResampler(K, out["choices"].shape[0], K_total, dev) if (sharded and peer is None) else None

    def step(i, timed):
        key = (0, 1 + i)
        j = sample_at.get(i) if timed else None
        tmr = None
        if j is not None:
            if j % 2 == 0:
                tmr = timers[j]             # events attached to the kernel's dispatch (gjx_run_program_ex / gjx_importance_step_ex)
            else:
                ev[j][0].record()
        if peer is not None:
            par = peer_calls[0] & 1
            peer_calls[0] += 1
            kernels.run_program(prog, key, K, offset=off, K_total=K_total, ws=ws, out=peer_out[par], want_weight=False, want_lse=False, timer=tmr)
            if j is not None and j % 2 == 1:
                ev[j][1].record()
        elif not fused_step:
            kernels.run_program(prog, key, K, offset=off, K_total=K_total, ws=ws, out=out, want_weight=False,
                                want_lse=sharded, timer=tmr)
            if j is not None and j % 2 == 1:
                ev[j][1].record()
        u = ((i * 2654435761) % (1 << 23)) / float(1 << 23)
        if fused_step:
            # one launch: propagate + reweight + LSE + prefix sums + systematic ancestors + gather (gjx_importance_step)
            kernels.importance_step(prog, key, K, u, out=step_out, allow_fallback=False, timer=tmr)
            if j is not None and j % 2 == 1:
                ev[j][1].record()
            return step_out["lse"]
        if world == 1 and not sharded:
            # single GPU: the LSE reduction is finished by the prefix-sum kernels' prologue (no serial tail)
            if two_launch:      # weights -> ancestors -> children in one launch (ancestors stay on chip)
                kernels.resample_gather(out["logw"], u, out["choices"], partials=(ws, n_part), lse_out=lse_rec, K_total=K_total,
                                        out=rows, ws=ws2, allow_fallback=False)
            else:
                kernels.resample_indices(out["logw"], u, K_total, partials=(ws, n_part), lse_out=lse_rec, K_total=K_total, anc=anc, ws=ws2)
                kernels.gather_rows(out["choices"], anc, rows)
            return lse_rec
        if peer is not None:
            # one launch: tile granules, two G-word hops between the ranks, ancestors, children pulled from their owners
            peer.resample_gather(par, u, partials=(ws, n_part), out=rows, lse_out=lse_rec)
            return lse_rec
        # collective transport: 8-byte all-gather of per-rank {max, sumexp} (reduced in the prefix-sum prologue), 8-byte
        # all-gather of weight totals, device-side plan, local gather + all-to-all-v of the surplus children
        _, lse_g = resampler.step(out["choices"], out["logw"], out["lse"], u)
        return lse_g

    # One-off costs that belong to neither W nor the timed region: RCCL sets up channels lazily per collective, and
    # the HIP runtime grows its signal / kernel-argument pools the first time the launch queue gets deep — a single
    # 35-45 ms host stall inside one launch call (seen with a per-call timer around the launches), 2-3x the whole default timed region.  A deep
    # un-synchronised burst here triggers it before the clock starts.
    for i in range(300):
        step(i, False)
    torch.cuda.synchronize()
    dt, lse = timed_loop(args, world, dev, step)
    exch = dict(transport="none")
    status_word = None
    if peer is not None:
        lse = lse.clone()
        torch.cuda.synchronize()
        status_word = peer.status()
        exch = dict(transport="peer", ranks=world, ranks_on_this_device=peer.ranks_on_device, status=status_word,
                    note="peer-mapped windows (hipIpc): one resampling launch per rank, two G-word hops, children pulled from their owners")
        peer.close()
    if resampler is not None:
        lse = lse.clone()
        torch.cuda.synchronize()
        exch = resampler.stats()
        resampler.close()            # communicator torn down on every rank while the process group is still up
    if sharded:
        exch = _exchange_record(exch, dev, status_word)      # (collective: every rank's view travels to rank 0)
    if rank != 0:
        return None
    disp = [timers[j].elapsed_us() for j in range(n_samp) if j % 2 == 0]
    brk = [ev[j][0].elapsed_time(ev[j][1]) * 1e3 for j in range(n_samp) if j % 2 == 1]
    kern_ms = sum(disp) / len(disp) * 1e-3
    bracket_us = sum(brk) / len(brk) if brk else None
    for t in timers:
        t.close()
    # algorithmic bytes per particle: propagate+reweight 4 D + 12 = 76 written; resampling 4 + 4
    # (ancestor written, ancestor read; the log-weight is not re-read when the step is one launch); gather 8 (D + 1) = 136
    step_bytes = ALGO_BYTES_PER_PARTICLE + 8 + 8 * (D + 1)
    algo_bytes = (step_bytes if fused_step else ALGO_BYTES_PER_PARTICLE) * K
    kernel_name = "gjx::k_run_gmm_flat<16,4,256,STEP>" if fused_step else "gjx::k_run_gmm_flat<16,4,256>"
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    exact = golden("gmm_c8_d16_seed0")
    lml = float(lse[3])
    # HBM bytes per launch from the PMC counters are NOT measured in this run (counters need their own rocprofv3 passes):
    # `traffic` stays null and the figure of the committed counter run is reported beside it, labelled as such
    traffic_prof = None
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        tp = os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")   # FETCH_SIZE x 2 + WRITE_SIZE per launch (profiles/README.md)
        if os.path.exists(tp):                                           # keys: rocprofv3 kernel names without "void " and blanks
            want = "gjx::k_run_gmm_flat<%d,4,256," % D
            # template arguments: <D, particles per lane, threads, STEP (the one-launch importance step), TILES (tile totals left for the tile-scaled resampler)>
            v = next((v for k, v in json.load(open(tp)).items()
                      if k.startswith(want) and k[len(want):].startswith("true") == bool(fused_step)), None)
            if v is not None:
                traffic_prof = dict(bytes_per_launch=v, file="profiles/%s_pmc_traffic.json" % tag,
                                    note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run")
                break
    res = dict(
        metric="particle_steps_per_sec", value=K_total * args.steps / dt, unit="particle-steps/s",
        n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload="gmm_c8_d16 ImportanceK: propagate+reweight+LSE, systematic resample, gather "
                             "(BASELINE.json configs[1])",
                    k_particles_per_gpu=K, k_particles_total=K_total, rng_stream="flat", sharding=f"particles x{world}",
                    exchange=exch["transport"], exchange_stats=exch),
        roofline=dict(bound="hbm", kernel=kernel_name, achieved=achieved, peak=HBM_PEAK_GBS,
                      unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=None, traffic_from_profiles=traffic_prof,
                      stream="flat (the build's own stream layout and samplers, not the reference's key structure: "
                             "roofline_jax32_stream is the same kernel on the reference's structure)",
                      kernel_us=kern_ms * 1e3, kernel_us_event_pair_around_call=bracket_us,
                      timing="HIP events attached to the kernel dispatch on %d steps spread over the timed region" % len(disp),
                      algorithmic_bytes_per_launch=algo_bytes,
                      launches_per_step=1 if fused_step else (2 if (two_launch or peer is not None) else 3),
                      note=("the whole importance step is this one launch: propagate+reweight (76 B/particle), resampling (8), "
                            "gather of 17 rows (136); " if fused_step else "") +
                           "the propagate+reweight phase is bound by integer VALU issue (Threefry-2x32-20), see DESIGN.md §5; frac is vs HBM"),
        log_ml=lml, log_ml_exact=exact, log_ml_rel_err=abs(lml - exact) / abs(exact),
        timed_regions=len(timed_loop.last_regions), timed_units_per_region=timed_loop.last_reps,
        timed_total_ms=sum(timed_loop.last_regions) * timed_loop.last_reps * 1e3,
        timing_note="one unit = exactly `steps` steps; `timed_units_per_region` units are enqueued back to back between one pair "
                    "of barrier + synchronize and the bracket is divided by that count; median of `timed_regions` brackets",
        timed_region_ms_min_median_max=[min(timed_loop.last_regions) * 1e3, dt * 1e3, max(timed_loop.last_regions) * 1e3],
    )
    # the same kernel launched back to back with itself (a train of VALU-bound launches runs at a lower shader clock than
    # the kernel does behind the memory-bound gather inside the step: DESIGN.md section 7)
    if not fused_step:
        tm = [kernels.DispatchTimer() for _ in range(8)]
        for i in range(64):
            kernels.run_program(prog, (0, 1 + i), K, offset=off, K_total=K_total, ws=ws, out=out, want_weight=False, want_lse=False,
                                timer=tm[i // 8] if i % 8 == 4 else None)
        torch.cuda.synchronize()
        us = sorted(t.elapsed_us() for t in tm)[len(tm) // 2]
        for t in tm:
            t.close()
        res["roofline"]["back_to_back"] = dict(kernel_us=us, achieved=algo_bytes / (us * 1e-6) / 1e9, frac=algo_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                               note="median of 8 dispatch-timed launches inside a train of 64 identical launches")
    if fused_step:
        # the propagate+reweight kernel on its own (what the step's first phase costs as a separate launch), measured
        # with dispatch events outside the timed region, alternating with the other two kernels of the three-launch step
        tm = [kernels.DispatchTimer() for _ in range(8)]
        for i in range(40):
            kernels.run_program(prog, (0, 1 + i), K, offset=off, K_total=K_total, ws=ws, out=out, want_weight=False, want_lse=False,
                                timer=tm[i // 5] if i % 5 == 0 else None)
            kernels.resample_indices(out["logw"], 0.5, K_total, partials=(ws, n_part), lse_out=lse_rec, K_total=K_total, anc=anc, ws=ws2)
            kernels.gather_rows(out["choices"], anc, rows)
        torch.cuda.synchronize()
        us = sum(t.elapsed_us() for t in tm) / len(tm)
        for t in tm:
            t.close()
        res["roofline_propagate_kernel"] = dict(kernel="gjx::k_run_gmm_flat<16,4,256>", kernel_us=us, achieved=ALGO_BYTES_PER_PARTICLE * K / (us * 1e-6) / 1e9,
                                                unit="GB/s", frac=ALGO_BYTES_PER_PARTICLE * K / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                algorithmic_bytes_per_launch=ALGO_BYTES_PER_PARTICLE * K,
                                                timing="dispatch events on 8 of 40 three-launch steps run after the timed region")
    if world == 1:
        # the same propagate+reweight kernel on the reference's stream layout and samplers (GJX_RNG_JAX32:
        # key per particle, key per site, one 32-bit word per draw, erfinv normals, Gumbel-max categorical)
        prog_j, _ = workloads.gmm_program(D=D, C=C, rng=A.RNG_JAX32)
        oj = kernels.run_program(prog_j, (0, 1), K, ws=ws, want_weight=False, want_lse=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            kernels.run_program(prog_j, (0, 1 + i), K, ws=ws, out=oj, want_weight=False, want_lse=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res["roofline_jax32_stream"] = dict(kernel="gjx::k_run_gmm<JAX32,16,4,256>", kernel_us=us,
                                            achieved=ALGO_BYTES_PER_PARTICLE * K / (us * 1e-6) / 1e9, unit="GB/s",
                                            frac=ALGO_BYTES_PER_PARTICLE * K / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_gmm(prog, min(K, 1 << 20))
    return res


def run_ssm(args, rank, world, dev):
    """config 3: linear-Gaussian SSM d=8, T=256, bootstrap filter K=2^18, systematic resampling every step."""
    from genjax_amd import core, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    s = workloads.ssm_problem()
    T = 256
    # N = 1: config 3 (K = 2^18).  N > 1: config 4, K = 2^22 in total however many ranks share it (strong scaling;
    # 2^19 per GPU at N = 8); --weak keeps 2^19 per GPU instead (K = N * 2^19).
    if args.ssm_k_total:
        K = int(args.ssm_k_total)
    elif world == 1:
        K = 1 << 18
    else:
        K = (world << 19) if args.weak else (1 << 22)
    K_local = K // world
    scaling = "weak" if (args.weak or world == 1) else "strong"
    # fixed-point scheme of the resampler (include/gjx.h): tile-scaled = one rendezvous per step (among the blocks of one
    # GPU, or among the ranks through peer-mapped windows); global_max on a sharded collection = the collective transport
    scheme = args.ssm_weights
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, weights=scheme)
    ys = torch.as_tensor(s["y"], device=dev)
    last = {}

    def step(i, timed):                      # one "step" here = one whole T-step filter run (T*K particle-steps)
        last["out"] = bf.run(core.key(1 + i), ys, device=dev, rank=rank, world=world)
        return last["out"]["log_ml"]

    # see run_gmm: the HIP runtime's one-off pool-growth stalls (35-45 ms each, two or three of them over the first
    # ~10^4 launches when the queue is kept this deep) must not land between the barriers of a 60 ms timed region
    for i in range(20 if world == 1 else 2):
        step(i, False)
    torch.cuda.synchronize()
    dt, lml = timed_loop(args, world, dev, step)
    lml = float(lml)
    other = None
    if world == 1:                           # the other weight scheme beside it, same keys, untimed warm-up then 3 runs
        o_name = "global_max" if scheme == "tile_scaled" else "tile_scaled"
        bo = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, weights=o_name)
        for i in range(3):
            bo.run(core.key(1 + i), ys, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            lo = bo.run(core.key(args.warmup + args.steps + i - 2), ys, device=dev)["log_ml"]
        torch.cuda.synchronize()
        other = dict(weights=o_name, us_per_filter_step=(time.perf_counter() - t0) / 3 / T * 1e6, log_ml=float(lo))
    exch = dict(transport="none")
    status_word = None
    torch.cuda.synchronize()
    if getattr(bf, "_peer", None) is not None:
        status_word = bf._peer.status()
        exch = dict(transport="peer", ranks=world, ranks_on_this_device=bf._peer.ranks_on_device, status=status_word,
                    note="peer-mapped windows (hipIpc): granules pushed, source tiles and ancestors pulled inside one launch; no collective calls")
    elif getattr(bf, "_resampler", None) is not None:
        exch = bf._resampler.stats()
    bf.close()
    if world > 1:
        exch = _exchange_record(exch, dev, status_word)      # (collective: every rank's view travels to rank 0)
    if rank != 0:
        return None
    exact = golden("ssm_dx8_T256_seed0")
    per_step_us = dt / args.steps / T * 1e6
    algo = (8 * 8 + 24) * K_local            # 8*d_x + 16..24 B per particle-step
    res = dict(
        metric="particle_steps_per_sec", value=K * T * args.steps / dt, unit="particle-steps/s", n_gpus=world,
        steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling=scaling,
        vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload="lgssm_d8_T256 bootstrap filter, systematic resampling every step (BASELINE.json configs[%d]); "
                             "one bench step = one T=256 filter run" % (2 if world == 1 else 3), k_particles_per_gpu=K_local,
                    k_particles_total=K, T=T, rng_stream="flat", sharding=f"particles x{world}",
                    resampler_weights=scheme, exchange=exch["transport"], exchange_stats=exch),
        roofline=dict(bound="hbm", kernel=(("gjx::k_ssm_persistent<FLAT,8,1024,%s> (steps 1..T-1 of the filter in ONE launch: %s grid "
                                            "rendezvous per step, resample + propagate + reweight)"
                                            % (("true", "one") if scheme == "tile_scaled" else ("false", "two")))
                                           if exch["transport"] == "none" and K_local <= (1 << 18) and scheme != "tile_scaled" else
                                           "gjx::k_pf_persistent<FLAT,8,SPL> (steps 1..T-1 in ONE launch, SPL 1024-slot tiles per block, one rendezvous per step"
                                           + (" among all ranks through peer-mapped windows)" if exch["transport"] == "peer" else ")")
                                           if (exch["transport"] == "peer" or (exch["transport"] == "none" and scheme == "tile_scaled")) else
                                           "sharded filter step: k_ssm_step + collective exchange"),
                      achieved=algo / (per_step_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      frac=algo / (per_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, traffic=None, kernel_us=per_step_us,
                      algorithmic_bytes_per_launch=algo,
                      note="latency-bound at K=2^18: grid rendezvous (~3.5 us each) + the ancestor search + the propagate phase per "
                           "step, no kernel boundary (phase timeline: profiles/, DESIGN.md section 5); kernel_us = wall time per filter step"),
        log_ml=float(lml), log_ml_exact=exact, log_ml_rel_err=abs(float(lml) - exact) / abs(exact),
    )
    # rtol 1e-4 is NOT a single-run property at K = 2^18: an ideal float64 bootstrap filter of this size has an rms relative error of
    # 7.0e-5 over its seeds (tests/golden/ssm_pf_float64.json, make_ssm_pf_float64.py).  The honest single-run figure is the z-score of
    # this run's log-ML against that filter's single-run distribution (|z| < 3 expected; the tests hold the 32-seed rms to its rms).
    if K == (1 << 18):
        with open(os.path.join(ROOT, "tests", "golden", "ssm_pf_float64.json")) as f:
            ref = np.asarray(json.load(f)["log_ml"], np.float64)
        res["log_ml_z"] = (float(lml) - float(ref.mean())) / float(ref.std(ddof=1))
        res["log_ml_z_note"] = "z of this run against the single-run spread of an ideal float64 filter (16 seeds, tests/golden/ssm_pf_float64.json)"
    if other is not None:
        other["log_ml_rel_err"] = abs(other["log_ml"] - exact) / abs(exact)
        if "log_ml_z" in res:
            other["log_ml_z"] = (other["log_ml"] - float(ref.mean())) / float(ref.std(ddof=1))
        res["other_weight_scheme"] = other
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_ssm(s, K, T)
    return res


def run_hmc(args, rank, world, dev):
    """config 5: hierarchical logistic regression N=1024, P=16; HMC 2^16 chains x L=1000, eps 0.01 (chains shard by rank)."""
    from genjax_amd import kernels, workloads
    N, P, L = 1024, 16, args.leapfrog
    n = (1 << 16)
    prog, pr = workloads.logreg_program(N=N, P=P)
    engine = kernels.hmc_engine(prog)
    assert engine in (2, 3), "fused logistic-regression HMC kernel not selected"
    ch0 = torch.as_tensor((np.random.default_rng(rank).standard_normal((P + 1, n)) * 0.1).astype(np.float32), device=dev)
    state = {"ch": ch0.clone(), "ws": None}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 8))]

    def step(i, timed):
        j = i if (timed and i < len(ev)) else None
        if j is not None:
            ev[j][0].record()
        out = kernels.hmc(prog, (1, 2 + i), state["ch"], 0.01, L, False, True, offset=rank * n, ws=state["ws"])
        if j is not None:
            ev[j][1].record()
        state["ws"] = out["_ws"]
        return out["accepted"]

    dt, acc = timed_loop(args, world, dev, step)
    if rank != 0:
        return None
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    flops = n * L * (2 * 2 * N * P + 10 * N)            # X beta and X^T r (2 flop per FMA) + sigmoid
    tf = flops / (kern_ms * 1e-3) / 1e12
    res = dict(
        metric="chain_leapfrogs_per_sec", value=n * world * L * args.steps / dt, unit="chain-leapfrogs/s", n_gpus=world,
        steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype="f32", data="synthetic",
        config=dict(workload="hier_logreg_N1024_P16 HMC with fused MH accept (BASELINE.json configs[4]); one bench step = "
                             f"one move of L={L} leapfrog steps", chains_per_gpu=n, leapfrog=L, eps=0.01, rng_stream="flat"),
        roofline=dict(bound="mfma", kernel=("gjx::k_hmc_logreg_mfma2<FLAT,false,1024>" if engine == 3 else "gjx::k_hmc_logreg<FLAT,16,false>"), achieved=tf, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                      frac=tf / FP32_PEAK_TFLOPS, traffic=None, kernel_us=kern_ms * 1e3,
                      note="both contractions on v_mfma_f32_16x16x4_f32 (exact f32).  The f32 MFMA occupies the SIMD's own f32 lanes: vector "
                           "instructions beside it cost their full issue time on top of its 32 cycles (profiles/r03_mfma_valu_overlap_microbench.txt), "
                           "so per 32 observations a wave needs 32 MFMAs (1024 cycles) + 48 sigmoid instructions (~330) + ~80 of LDS reads and loop: "
                           "MFMA busy 0.72 by SQ_VALU_MFMA_BUSY_CYCLES; ~0 HBM bytes"),
        accept_rate=float(acc.mean()),
    )
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_hmc(prog, P, L)
    return res


def run_hmc_generic(dev):
    """The same config-5 model through the GENERIC per-chain HMC kernel (k_hmc_generic: forward sweep over the site list
    with analytic d logpdf / d(value, params), any program) instead of the fused logistic-regression kernel: what a model
    without a dedicated kernel gets.  Same 2^16 chains, L = 5, same eps."""
    from genjax_amd import kernels, workloads
    N, P, L, n = 1024, 16, 5, 1 << 16
    prog, _ = workloads.logreg_program(N=N, P=P)
    old = os.environ.get("GJX_FORCE_GENERIC")
    os.environ["GJX_FORCE_GENERIC"] = "1"
    try:
        eng = kernels.hmc_engine(prog)
        ch = torch.as_tensor((np.random.default_rng(0).standard_normal((P + 1, n)) * 0.1).astype(np.float32), device=dev)
        out = kernels.hmc(prog, (1, 2), ch, 0.01, L, False, True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(3):
            out = kernels.hmc(prog, (1, 3 + i), ch, 0.01, L, False, True, ws=out["_ws"])
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
    finally:
        if old is None:
            del os.environ["GJX_FORCE_GENERIC"]
        else:
            os.environ["GJX_FORCE_GENERIC"] = old
    flops = n * L * (2 * 2 * N * P + 10 * N)
    return dict(engine=eng, chains=n, leapfrog=L, ms_per_move=ms, chain_leapfrogs_per_sec=n * L / (ms * 1e-3),
                tflops=flops / (ms * 1e-3) / 1e12, accept_rate=float(out["accepted"].mean()),
                note="generic site-list HMC kernel on the config-5 model (the fused MFMA kernel is extra.hmc)")


def run_hmc_generated(dev):
    """HMC kernels GENERATED from the site list (gjx_hmc engine 4) at 2^16 chains: the config-5 model through the generated
    kernel next to the hand-written matrix-core kernel, and two targets that have no hand-written kernel — a hierarchy with
    latent shape parameters (digamma gradients) and a 16-step observed random-walk Scan — generated vs site interpreter."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from genjax_amd import _abi as A
    from genjax_amd import kernels, workloads
    from genjax_amd.program import PackedProgram
    n = 1 << 16

    def timed(prog, ch0, eps, L, engine, reps=3):
        old = os.environ.get("GJX_HMC_ENGINE")
        os.environ["GJX_HMC_ENGINE"] = engine
        try:
            eng = kernels.hmc_engine(prog)
            ch = ch0.clone()
            out = kernels.hmc(prog, (1, 2), ch, eps, L, False, True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(reps):
                out = kernels.hmc(prog, (1, 3 + i), ch, eps, L, False, True, ws=out["_ws"])
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
        finally:
            if old is None:
                del os.environ["GJX_HMC_ENGINE"]
            else:
                os.environ["GJX_HMC_ENGINE"] = old
        return dict(engine=eng, ms_per_move=ms, chain_leapfrogs_per_sec=n * L / (ms * 1e-3), accept_rate=float(out["accepted"].mean()))

    res = {}
    N, P = 1024, 16
    prog, _ = workloads.logreg_program(N=N, P=P)
    ch = torch.as_tensor((np.random.default_rng(0).standard_normal((P + 1, n)) * 0.1).astype(np.float32), device=dev)
    g = timed(prog, ch, 0.01, 100, "gen")
    f = timed(prog, ch, 0.01, 100, "fused")
    flops = n * 100 * (2 * 2 * N * P + 10 * N)
    g["tflops"] = flops / (g["ms_per_move"] * 1e-3) / 1e12
    g["vs_hand_written_mfma_kernel"] = g["chain_leapfrogs_per_sec"] / f["chain_leapfrogs_per_sec"]
    res["hier_logreg_N1024_P16_L100"] = dict(generated=g, hand_written=f)
    sl = H.shape_hierarchy()
    hp = PackedProgram(sl, {s.addr: A.MODE_OBS_SLOT for s in sl.sites}, selected=("la", "lb"))
    # start states: a draw from the prior by the DEVICE's own simulate (the CPU checker is not touched outside cpu_baseline_*)
    base = kernels.run_program(PackedProgram(sl), (3, 4), 4096)["choices"].float()
    ch = base.repeat(1, n // 4096).contiguous()
    gi, ii = timed(hp, ch, 0.002, 200, "gen"), timed(hp, ch, 0.002, 200, "interp", reps=1)
    res["shape_hierarchy_L200"] = dict(generated=gi, interpreter=ii, speedup=ii["ms_per_move"] / gi["ms_per_move"])
    sp, ys = H.scan_chain(16, carry=True, observe=True, sigma=0.3, r=0.5)
    ssl = sp.site_list
    modes = {s.addr: (A.MODE_OBS_TAB if s.addr[0] == "y" else A.MODE_OBS_SLOT) for s in ssl.sites}
    scp = PackedProgram(ssl, modes, {("y", t): ys[t] for t in range(16)}, selected=tuple(("x", t) for t in range(16)))
    ch = torch.as_tensor((np.random.default_rng(6).standard_normal((16, n)) * 0.3).astype(np.float32), device=dev)
    gi, ii = timed(scp, ch, 0.02, 200, "gen"), timed(scp, ch, 0.02, 200, "interp", reps=1)
    res["scan_T16_L200"] = dict(generated=gi, interpreter=ii, speedup=ii["ms_per_move"] / gi["ms_per_move"])
    return res


def run_api(dev, K, steps=100):
    """The gmm step through the public API instead of a hand-built program: @gen body -> Target -> ImportanceK.run_smc
    -> N-of-K systematic resampling (inference.pf.resample).  The traced site list and the packed program (table on the
    device, engine chosen) are cached per (gen fn, arguments, constraint content), so a step costs the same launches
    as the kernel-level loop plus the Python of the API objects.  -> dict(ms_per_step, particle_steps_per_sec, engine)"""
    import genjax_amd as genjax
    from genjax_amd import C, kernels, workloads
    from genjax_amd.inference import ImportanceK, Target
    from genjax_amd.inference.pf import resample
    g = workloads.gmm_problem(C=8, D=D)

    @genjax.gen
    def model():
        z = genjax.categorical(g["logits"]) @ "z"
        mu, sig = genjax.const(g["mu"]), genjax.const(g["sigma"])
        x = genjax.mv_normal_diag(mu[z], sig[z]) @ "x"
        genjax.mv_normal_diag(x, g["r"]) @ "y"
        return x

    target = Target(model, (), C["y"].set(g["y"]))
    alg = ImportanceK(target, k_particles=K)
    keys = genjax.split(genjax.key(7), steps + 10)

    def step(k):
        pc = alg.run_smc(k)
        tr = pc.get_particles()
        rows, anc = resample(tr.choices, pc.get_log_weights(), k, collection=pc)
        return pc, rows

    for k in keys[:10]:
        pc, rows = step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in keys[10:]:
        pc, rows = step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(ms_per_step=dt / steps * 1e3, particle_steps_per_sec=K * steps / dt, engine=kernels.program_engine(pc.get_particles().prog),
                log_ml=float(pc.get_log_marginal_likelihood_estimate()), steps=steps)


def run_codegen(dev):
    """Kernels generated from the site list (hipRTC, csrc/gjx_codegen.hip) at K = 2^20: the mixture program through its
    generated kernel next to the hand-fused one, and two programs that have no hand-written kernel.  Kernel time from
    HIP events attached to the dispatch; `frac` = algorithmic bytes (4 B per stored scalar + score + log-weight) / time
    against 8 TB/s; `bound` says what actually limits each."""
    from genjax_amd import _abi as A
    from genjax_amd import kernels, workloads
    from genjax_amd.program import PackedProgram, Param, SiteList
    K = 1 << 20

    def timed(prog, engine):
        old = os.environ.get("GJX_ENGINE")
        os.environ["GJX_ENGINE"] = engine
        try:
            eng = kernels.program_engine(prog)
            ws = kernels.workspace(A.OP_RUN, K, dev)
            out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False)
            # (a kernel timed cold runs up to 12 % slower than inside a train: clocks.  A generated kernel is compiled right before its
            # first launch — seconds of an idle GPU — so the train is long enough to bring the clocks back for every engine alike:
            # 400 launches, about 12 ms; with 40 the hand-fused kernel, measured first, came out 6 % faster than in a warm A/B)
            for i in range(400 if eng != 0 else 0):
                kernels.run_program(prog, (0, 2 + i), K, ws=ws, out=out, want_weight=False)
            tm = [kernels.DispatchTimer() for _ in range(5)]
            for i, t in enumerate(tm):
                kernels.run_program(prog, (0, 2 + i), K, ws=ws, out=out, want_weight=False, timer=t if eng != 0 else None)
            torch.cuda.synchronize()
            us = sorted(t.elapsed_us() for t in tm)[2] if eng != 0 else None
            for t in tm:
                t.close()
        finally:
            if old is None:
                del os.environ["GJX_ENGINE"]
            else:
                os.environ["GJX_ENGINE"] = old
        b = (4 * prog.n_slots + 8) * K
        return dict(engine=eng, kernel_us=us, algorithmic_bytes=b, frac=(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if us else None,
                    roofline=(dict(bound="hbm", algorithmic_bytes_per_launch=b, kernel_us=us, achieved=b / (us * 1e-6) / 1e9, peak=HBM_PEAK_GBS,
                                   unit="GB/s", frac=b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, timing="HIP events attached to the dispatch, median of 5")
                              if us else None),
                    log_ml=float(out["lse"][3]))

    res = {}
    gmm, _ = workloads.gmm_program(D=D, C=C)
    first = (timed(gmm, "auto"), timed(gmm, "gen"))
    # the pair is timed a second time once both kernels exist, and the second pass is the line: in the first pass the generated kernel is
    # compiled (seconds, GPU idle) between the two trains, and on a box with a cold kernel cache that alone read 1.24 against 1.15
    res["gmm_hand_fused"] = dict(timed(gmm, "auto"), first_pass_kernel_us=first[0]["kernel_us"])
    res["gmm_generated"] = dict(timed(gmm, "gen"), first_pass_kernel_us=first[1]["kernel_us"])
    res["gmm_generated"]["vs_hand_fused"] = res["gmm_generated"]["kernel_us"] / res["gmm_hand_fused"]["kernel_us"]
    sl = SiteList()
    sl.add("p", A.BETA, [np.float32(2.0), np.float32(2.0)])
    sl.add("v", A.FLIP, [Param.value("p", 1)])
    bb = PackedProgram(sl, {"v": A.MODE_OBS_TAB}, {"v": np.float32(1.0)})
    res["beta_bernoulli"] = dict(timed(bb, "auto"), log_ml_exact=math.log(0.5),
                                 bound="VALU: two Marsaglia-Tsang log-gamma variates per particle for 12 B of output")
    # two programs no hand-written kernel or matcher knows, HBM-shaped like the headline model (cheap samplers, 16 stored
    # scalars per particle): a conjugate normal-normal model and an observed random-walk chain (every site depends on the
    # previous one); exact log-ML of both from the Gaussian marginal
    g = workloads.gmm_problem(C=C, D=D)
    sl = SiteList()
    sl.add("x", A.MVNORMAL_DIAG, [Param.const(np.zeros(D, np.float32)), Param.const(np.full(D, 1.5, np.float32))], dim=D)
    sl.add("y", A.MVNORMAL_DIAG, [Param.value("x", D), Param.const(g["r"])], dim=D)
    nn = PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": g["y"]})
    var = 1.5 ** 2 + np.asarray(g["r"], np.float64) ** 2
    nn_exact = float(np.sum(-0.5 * np.asarray(g["y"], np.float64) ** 2 / var - 0.5 * np.log(2 * np.pi * var)))
    res["normal_normal_d16"] = dict(timed(nn, "auto"), log_ml_exact=nn_exact)
    T = 16
    ys = (np.cumsum(np.random.default_rng(3).standard_normal(T)) * 0.7).astype(np.float32)
    sl = SiteList()
    for t in range(T):
        sl.add(f"x{t}", A.NORMAL, [Param.const(0.0) if t == 0 else Param.value(f"x{t - 1}", 1), Param.const(1.0)])
        sl.add(f"y{t}", A.NORMAL, [Param.value(f"x{t}", 1), Param.const(2.0)])
    rw = PackedProgram(sl, {f"y{t}": A.MODE_OBS_TAB for t in range(T)}, {f"y{t}": ys[t] for t in range(T)})
    cov = np.minimum.outer(np.arange(1, T + 1), np.arange(1, T + 1)).astype(np.float64) + 4.0 * np.eye(T)
    rw_exact = float(-0.5 * ys.astype(np.float64) @ np.linalg.solve(cov, ys.astype(np.float64)) - 0.5 * np.linalg.slogdet(2 * np.pi * cov)[1])
    res["random_walk_T16"] = dict(timed(rw, "auto"), log_ml_exact=rw_exact)
    lr, _ = workloads.logreg_importance_program(N=1024, P=16)
    r = timed(lr, "auto")
    r["flops"] = K * (2 * 1024 * 16 + 10 * 1024)
    r["tflops"] = r["flops"] / (r["kernel_us"] * 1e-6) / 1e12 if r["kernel_us"] else None
    r["bound"] = ("f32 matrix cores + vector ALU, which do not overlap on a SIMD: the [1024 x 16] x [16 x particles] contraction on "
                  "v_mfma_f32_16x16x4_f32 (16 K FMA per particle) and 1 K softplus per particle, for 76 B of output")
    r["frac_of_f32_peak"] = r["tflops"] / FP32_PEAK_TFLOPS if r["tflops"] else None
    res["hier_logreg_prior_likelihood"] = r
    return res


def run_sharded_one_rank(dev):
    """The SHARDED code paths with one rank, next to the unsharded ones (same process, same sizes, back to back): what the
    sharding machinery itself costs before any fabric latency.  gmm: propagate+reweight into the peer context's window +
    gjx_peer_resample_gather, vs propagate+reweight + gjx_resample_gather.  ssm: gjx_ssm_filter_peer vs
    gjx_ssm_filter_scheme at K = 2^19 (BASELINE configs[3]'s per-GPU size)."""
    from genjax_amd import _abi as A
    from genjax_amd import core, kernels, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    K = K_PER_GPU
    prog, _ = workloads.gmm_program(D=D, C=C)
    ws = kernels.workspace(A.OP_RUN, K, dev)
    ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
    n_part = kernels.run_partials_count(prog, K, 0)
    out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False, want_lse=False)
    rows = torch.empty_like(out["choices"])
    rec = torch.empty(4, dtype=torch.float32, device=dev)
    peer = kernels.PeerContext(K, out["choices"].shape[0], dev)
    pout = [dict(choices=peer.rows[p], score=out["score"], logw=peer.logw[p]) for p in (0, 1)]

    def plain(i):
        kernels.run_program(prog, (0, 1 + i), K, ws=ws, out=out, want_weight=False, want_lse=False)
        kernels.resample_gather(out["logw"], 0.5, out["choices"], partials=(ws, n_part), lse_out=rec, K_total=K, out=rows, ws=ws2, allow_fallback=False)

    def sharded(i):
        kernels.run_program(prog, (0, 1 + i), K, ws=ws, out=pout[i & 1], want_weight=False, want_lse=False)
        peer.resample_gather(i & 1, 0.5, partials=(ws, n_part), out=rows, lse_out=rec)

    def timed(fn, n=400):
        for i in range(50):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n * 1e6)
        return sorted(ts)[1]

    g_plain, g_shard = timed(plain), timed(sharded)
    lml = float(rec[3])
    st = peer.status()
    peer.close()
    s = workloads.ssm_problem()
    ys = torch.as_tensor(s["y"], device=dev)
    Ks = 1 << 19
    bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), Ks, weights="tile_scaled")
    T = ys.shape[0]

    def t_ssm(fn):
        fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            o = fn(1 + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 / T * 1e6, float(o["log_ml"])

    s_plain, l_plain = t_ssm(lambda i: bf.run(core.key(1 + i), ys, device=dev))
    s_shard, l_shard = t_ssm(lambda i: bf._run_peer(core.key(1 + i), ys, dev, 1, True))
    bf.close()
    return dict(note="sharded code paths (peer-mapped windows, csrc/gjx_peer.hip) run with ONE rank beside the unsharded kernels: the "
                     "cost of the sharding machinery itself; with more ranks add two G-word hops (gmm) / one granule all-gather (ssm) per step",
                gmm=dict(k_particles=K, us_per_step_unsharded=g_plain, us_per_step_sharded_1rank=g_shard, ratio=g_shard / g_plain,
                         log_ml=lml, status=st),
                ssm=dict(k_particles=Ks, T=T, us_per_filter_step_unsharded=s_plain, us_per_filter_step_sharded_1rank=s_shard,
                         ratio=s_shard / s_plain, log_ml_unsharded=l_plain, log_ml_sharded=l_shard))


def run_round3(dev):
    """Round-3 paths, short runs: (1) the ImportanceK step with the PLAIN-launch tile-scaled resampler
    (gjx_resample_gather_tiled: no co-resident grid) next to the one-launch global-maximum resampler, at K = 2^20 and at
    the sizes the latter cannot run in one launch; (2) the bootstrap filter with resample-move rejuvenation inside the
    one-launch filter vs the step-by-step loop; (3) assess of a T = 256 Scan trace (every latent constrained to the
    particle's own value) on the rolled generated kernel vs the site interpreter."""
    from genjax_amd import _abi as A
    from genjax_amd import core, kernels, workloads
    from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
    from genjax_amd.program import PackedProgram, Param, SiteList
    res = {}
    prog, _ = workloads.gmm_program(D=D, C=C)

    def timed(fn, n=300):
        for i in range(30):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n * 1e6)
        return sorted(ts)[1]

    steps = {}
    for K in (1 << 20, 1 << 21, 1 << 22):
        ws = kernels.workspace(A.OP_RUN, K, dev)
        ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
        out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False, want_lse=False, want_tiles=True)
        part = out["_partials"]
        rows = torch.empty_like(out["choices"])
        rec = torch.empty(4, dtype=torch.float32, device=dev)

        def tiled(i):
            kernels.run_program(prog, (0, 1 + i), K, ws=ws, out=out, want_weight=False, want_lse=False, want_tiles=True)
            kernels.resample_gather_tiled(out["logw"], 0.5, out["choices"], partials=(ws, part.count()), tiles=part.tiles, lse_out=rec,
                                          K_total=K, out=rows, ws=ws2)

        def gmax(i):
            kernels.run_program(prog, (0, 1 + i), K, ws=ws, out=out, want_weight=False, want_lse=False)
            kernels.resample_gather(out["logw"], 0.5, out["choices"], partials=(ws, part.count()), lse_out=rec, K_total=K, out=rows, ws=ws2)

        n = 300 if K == 1 << 20 else 100
        t_t, t_g = timed(tiled, n), timed(gmax, n)
        steps[str(K)] = dict(us_per_step_tile_scaled_plain_launch=t_t, us_per_step_global_max=t_g,
                             particle_steps_per_sec_tile_scaled=K / t_t * 1e6, tiles_from_the_propagate_kernel=bool(part.tiles),
                             log_ml=float(rec[3]))
        del out, rows
    res["importance_step_tile_scaled_resampler"] = dict(
        note="global_max = gjx_resample_gather (one co-resident launch up to K = 2^20 on a full MI355X, beyond that "
             "gjx_resample_indices + gjx_gather_rows); tile_scaled = gjx_resample_gather_tiled, a plain launch at every size", **steps)
    s = workloads.ssm_problem()
    ys = torch.as_tensor(s["y"], device=dev)
    T = ys.shape[0]
    mv = {}
    for K in (1 << 18, 1 << 19):
        bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), K, rejuvenate=dict(n_moves=1, scale=0.4), weights="tile_scaled")
        row = {}
        for name, sbs in (("one_launch", False), ("step_by_step", True)):
            bf.run(core.key(1), ys, device=dev, step_by_step=sbs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(2):
                o = bf.run(core.key(2 + i), ys, device=dev, step_by_step=sbs)
            torch.cuda.synchronize()
            row["us_per_filter_step_" + name] = (time.perf_counter() - t0) / 2 / T * 1e6
            row["log_ml_" + name] = float(o["log_ml"])
        row["accept_rate"] = bf.last_accept_rate
        mv[str(K)] = row
    res["resample_move_filter"] = dict(note="lgssm_d8_T256, one random-walk Metropolis move per particle and step, tile-scaled resampler", **mv)
    # assess of a Scan trace: x_t ~ N(x_{t-1}, 0.3) constrained per particle, y_t observed
    Ts, Ka = 256, 1 << 16
    sl = SiteList()
    modes, obs = {}, {}
    for t in range(Ts):
        sx = sl.add(("x", t), A.NORMAL, [Param.value(("x", t - 1)) if t else Param.const(0.0), Param.const(0.3)])
        sy = sl.add(("y", t), A.NORMAL, [Param.value(("x", t)), Param.const(0.7)])
        sx.scan = sy.scan = (1 << 20) | (t + 1)
        modes[("x", t)], modes[("y", t)], obs[("y", t)] = A.MODE_OBS_SLOT, A.MODE_OBS_TAB, np.float32(0.1)
    pa = PackedProgram(sl, modes, obs)
    ch = torch.randn((pa.n_slots, Ka), device=dev) * 0.5
    row = {}
    for name in ("gen", "interp"):
        os.environ["GJX_ENGINE"] = name
        try:
            row["engine_" + name] = kernels.program_engine(pa)
            row["us_" + name] = timed(lambda i: kernels.run_program(pa, (0, 1), Ka, choices=ch, want_weight=False), 20)
        finally:
            del os.environ["GJX_ENGINE"]
    row["speedup"] = row["us_interp"] / row["us_gen"]
    res["scan_trace_assess_T256"] = dict(k_particles=Ka, note="every x_t constrained to the particle's own value (OBS_SLOT): rolled generated kernel vs site interpreter", **row)
    # (4) BASELINE configs[0] (README quick example: 50 SIR trials x K = 50 through random_weighted): one call per trial vs
    # all trials as the shards of one launch (ImportanceK.random_weighted_trials), and the same at 4096 trials
    import genjax_amd as genjax
    from genjax_amd import ChoiceMap as Cm

    @genjax.gen
    def beta_bernoulli():
        p = genjax.beta(2.0, 2.0) @ "p"
        v = genjax.flip(p) @ "v"
        return v

    target = genjax.Target(beta_bernoulli, (), Cm.d({"v": True}))
    alg = genjax.ImportanceK(target, k_particles=50)
    k0 = genjax.key(314159)

    def loop50(i):
        return [alg.random_weighted(sk, target) for sk in genjax.split(genjax.fold_in(k0, i), 50)]

    tr = dict(note="beta_bernoulli, K = 50 particles per trial, wall time of the Python call incl. trace bookkeeping")
    tr["us_50_trials_one_call_each"] = timed(loop50, n=3)
    tr["us_50_trials_one_launch"] = timed(lambda i: alg.random_weighted_trials(genjax.fold_in(k0, i), 50, target), n=30)
    tr["us_4096_trials_one_launch"] = timed(lambda i: alg.random_weighted_trials(genjax.fold_in(k0, i), 4096, target), n=30)
    _, pch = alg.random_weighted_trials(k0, 4096, target)
    tr["posterior_mean_p_4096_trials"] = float(pch["p"].mean())
    res["readme_trials"] = tr
    return res


def _kalman_log_lik(A_, y, q, r, q0):
    """float64 Kalman log-likelihood of x_0 ~ N(0, q0^2 I), x_t ~ N(A x_{t-1}, q^2 I), y_t ~ N(x_t, r^2 I)"""
    A_ = np.asarray(A_, np.float64)
    y = np.asarray(y, np.float64)
    dx = A_.shape[0]
    m, P, ll = np.zeros(dx), q0 * q0 * np.eye(dx), 0.0
    for t in range(y.shape[0]):
        if t > 0:
            m, P = A_ @ m, A_ @ P @ A_.T + q * q * np.eye(dx)
        S = P + r * r * np.eye(dx)
        v = y[t] - m
        Si = np.linalg.inv(S)
        ll += -0.5 * (v @ Si @ v + np.linalg.slogdet(S)[1] + dx * math.log(2 * math.pi))
        Kg = P @ Si
        m, P = m + Kg @ v, (np.eye(dx) - Kg) @ P
    return float(ll)


def run_round4(dev):
    """Round-4 paths, short runs: (1) the bootstrap filter for ANY Scan kernel (gjx_scan_filter: one plain launch per step —
    resampling search, ancestor gather, propagate, reweight in the step's generated kernel) on config 3's model written as
    @gen + .scan, next to the hand-written one-launch filter at 2^18 and 2^20 particles, and on a stochastic-volatility model; (2) a vmapped mixture (N = 4096 data: two plate-tagged
    device sites, one instance loop) on its generated kernel and on the site interpreter."""
    import genjax_amd as genjax
    from genjax_amd import C as CM
    from genjax_amd import _abi as A
    from genjax_amd import kernels, workloads
    from genjax_amd.inference import BootstrapFilter, LinearGaussianSSM
    res = {}
    s = workloads.ssm_problem()
    T, K, q, r = 256, 1 << 18, float(s["q"]), float(s["r"])
    Am = np.asarray(s["A"], np.float32)
    dx = Am.shape[0]

    @genjax.gen
    def lg_step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(dx, q, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(dx, r, np.float32)) @ "y"
        return x, None

    def time_runs(run, n):
        # every run timed on its own (host call → device idle) and the MEDIAN reported, all runs listed beside it: a run is 3–10 ms, one
        # descheduled host thread or a clock step inside a mean of five moved the line by 10 % from box to box
        ts = []
        for i in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run(i)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts, out

    def time_filter(bf, chm, args, n=5):
        for i in range(2):
            bf.run(genjax.key(i), chm, args)
        dt, ts, out = time_runs(lambda i: bf.run(genjax.key(10 + i), chm, args), n)
        return dt, float(out["log_ml"]), dict(out.get("info", {}), runs=ts)

    exact = _kalman_log_lik(s["A"], s["y"], q, r, q)          # float64 closed form (x_0 ~ N(0, q^2 I): the Scan's step 0)
    ys_d = torch.as_tensor(s["y"], device=dev)
    for Kf, tag in ((K, "2e18"), (1 << 20, "2e20")):
        bf = BootstrapFilter(lg_step.scan(n=T), Kf)
        bf.alias_outputs = True                      # (the timed loop hands out the filter's own buffers: no copies inside the timing)
        dt, lml, finfo = time_filter(bf, CM["y"].set(np.asarray(s["y"], np.float32)), (np.zeros(dx, np.float32), None), n=9 if Kf == K else 5)
        hand = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"], q0=q), Kf, weights="tile_scaled")
        for i in range(3):
            hand.run(genjax.key(i), ys_d, device=dev)
        dth, tsh, _ = time_runs(lambda i: hand.run(genjax.key(10 + i), ys_d, device=dev), 9 if Kf == K else 5)
        res[f"scan_filter_lgssm_d8_T256_K{tag}"] = dict(
            us_per_step=dt / T * 1e6, particle_steps_per_sec=Kf * T / dt, log_ml=lml, log_ml_rel_err=abs(lml - exact) / abs(exact),
            # the form the LIBRARY reports for this run (gjx_filter_info), not what the environment asked for
            form=finfo.get("form_name"), form_id=finfo.get("form"), launches_per_run=finfo.get("launches"), grid=finfo.get("grid"),
            tiles_per_block=finfo.get("tiles_per_block"),
            hand_written_one_launch_filter_us_per_step=dth / T * 1e6, ratio=dt / dth,
            runs_us_per_step=[round(x / T * 1e6, 2) for x in finfo.get("runs", [])], hand_written_runs_us_per_step=[round(x / T * 1e6, 2) for x in tsh],
            timing="median of the listed whole runs, each from the host's call to an idle device",
            roofline=dict(bound="hbm", algorithmic_bytes_per_particle_step=8 * dx + 24, bytes_per_step=(8 * dx + 24) * Kf, us_per_step=dt / T * 1e6,
                          achieved=(8 * dx + 24) * Kf / (dt / T) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                          frac=(8 * dx + 24) * Kf / (dt / T) / 1e9 / HBM_PEAK_GBS, timing="wall clock over whole runs incl. step 0 and the host's calls"),
            engine=("gjx_gen_pf: the step program's sites as the model of the shared filter skeleton (csrc/gjx_pfcore.h), 16 waves per tile"
                    if finfo.get("form") == 3 else "gjx_gen / gjx_gen_steps: the tile-scaled resampler's search in the prologue of the step's generated code"))
    with open(os.path.join(ROOT, "tests", "golden", "sv_pf_float64.json")) as f:
        fx = json.load(f)
    phi, sigma, ysv = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)

    @genjax.gen
    def sv_step(x_prev, _):
        x = genjax.normal(phi * x_prev, sigma) @ "x"
        genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
        return x, None

    bsv = BootstrapFilter(sv_step.scan(n=len(ysv)), K)
    bsv.alias_outputs = True
    dts, lsv, sinfo = time_filter(bsv, CM["y"].set(ysv), (0.0, None))
    res["scan_filter_stochastic_volatility_T256_K2e18"] = dict(us_per_step=dts / len(ysv) * 1e6, log_ml=lsv, float64_filter_mean=fx["log_ml_mean"],
                                                               form=sinfo.get("form_name"), launches_per_run=sinfo.get("launches"),
                                                               float64_filter_std=fx["log_ml_std"], z=(lsv - fx["log_ml_mean"]) / fx["log_ml_std"])
    # a single run's z says little (rounds 4 / 5 timed ONE seed twice: +2.8, +3.7 against the 16-seed fixture's sd): 16 seeds against the
    # 256-seed fixture — mean difference in standard errors (tests/test_gpu_scan_filter.py holds 32 seeds to 3 SE; profiles/r06_sv_bias.txt)
    ests = np.array([float(bsv.run(genjax.key(300 + i), CM["y"].set(ysv), (0.0, None))["log_ml"]) for i in range(16)])
    ref64 = np.asarray(fx["log_ml"], np.float64)
    se_ = float(np.sqrt(ests.var(ddof=1) / ests.size + ref64.var(ddof=1) / ref64.size))
    res["scan_filter_stochastic_volatility_T256_K2e18"].update(seeds=16, mean_log_ml=float(ests.mean()), bias=float(ests.mean() - ref64.mean()),
                                                               bias_over_se=float((ests.mean() - ref64.mean()) / se_), spread_ratio=float(ests.std(ddof=1) / ref64.std(ddof=1)))
    # (2) the vmapped mixture: many instances x moderately many particles, and the few-particles / very-many-instances corner
    mu = np.array([-2.0, 0.5, 3.0], np.float32)

    def plate_row(N, Kp, engines):
        rs = np.random.default_rng(0)
        yv = (mu[rs.integers(0, 3, N)] + 0.7 * rs.standard_normal(N)).astype(np.float32)

        @genjax.gen
        def mk(lg):
            z = genjax.categorical(logits=lg) @ "z"
            return genjax.normal(genjax.take(mu, z), 0.7) @ "x"

        @genjax.gen
        def mix():
            mk.repeat(n=N)(np.array([0.2, -0.3, 0.1], np.float32)) @ "k"

        prog, _, _ = mix.pack((), CM["k", "x"].set(yv), True)
        row = dict(device_sites=prog.n_sites, logical_sites=len(prog.site_list.sites), K=Kp, N=N)
        for name, env in engines:
            old = {k: os.environ.get(k) for k in ("GJX_ENGINE", "GJX_GEN_WIDE")}
            os.environ.update(env)
            try:
                ws = kernels.workspace(A.OP_RUN, Kp, dev)
                o = kernels.run_program(prog, (0, 1), Kp, ws=ws, want_weight=False)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(5):
                    kernels.run_program(prog, (0, 2 + i), Kp, ws=ws, out=o, want_weight=False)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 5 * 1e3
                row[name] = dict(engine=o["_engine"], kernel_us=us, particle_instances_per_sec=Kp * N / (us * 1e-6), bytes=4.0 * N * Kp,
                                 achieved_GBs=4.0 * N * Kp / (us * 1e-6) / 1e9,
                                 roofline=dict(bound="hbm", algorithmic_bytes_per_launch=4.0 * N * Kp, kernel_us=us, achieved=4.0 * N * Kp / (us * 1e-6) / 1e9,
                                               peak=HBM_PEAK_GBS, unit="GB/s", frac=4.0 * N * Kp / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                               timing="event pair around 5 back-to-back launches"))
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        return row

    engines = (("gen", dict(GJX_ENGINE="gen")),                                  # the library's own choice: the wide form here
               ("gen_one_lane_per_particle", dict(GJX_ENGINE="gen", GJX_GEN_WIDE="0")),   # round 4's form: a lane walks all instances
               ("interp", dict(GJX_ENGINE="interp")))
    row = plate_row(4096, 1 << 17, engines)
    row["generated_vs_interpreter"] = row["interp"]["kernel_us"] / row["gen"]["kernel_us"]
    row["wide_vs_one_lane_per_particle"] = row["gen_one_lane_per_particle"]["kernel_us"] / row["gen"]["kernel_us"]
    row["form"] = ("gjx_gen, ppt | 512: a block of 16 waves shares 64 x PPT particles, the instances of the plate are dealt to the waves in "
                   "contiguous chunks, partial sums joined in LDS in wave order; few particles: | 1024 / | 2048 = 4 / 16 lanes per particle as well")
    res["vmapped_mixture_plate"] = row
    res["vmapped_mixture_plate_K2e12_N2e16"] = plate_row(65536, 1 << 12, engines[:2])
    return res


def run_round5(dev):
    """Round-5 paths, short runs: the generic filter (gjx_scan_filter) with a custom proposal per step, with Metropolis moves behind
    the resampling and with the multinomial resampler; HMC over a plate-tagged program (a regression with a latent per datum: the
    vmapped kernel's two-site body as a plate loop of the generated kernel) next to the site interpreter."""
    import genjax_amd as genjax
    from genjax_amd import C as CM
    from genjax_amd import kernels, workloads
    from genjax_amd.inference import BootstrapFilter
    res = {}
    T, K, dx = 256, 1 << 18, 8
    s = workloads.ssm_problem()
    q, r = float(s["q"]), float(s["r"])
    Am = np.asarray(s["A"], np.float32)
    ys = np.asarray(s["y"], np.float32)
    exact = _kalman_log_lik(s["A"], s["y"], q, r, q)
    s2 = 1.0 / (1.0 / q ** 2 + 1.0 / r ** 2)

    @genjax.gen
    def lg_step(x_prev, _):
        x = genjax.mv_normal_diag(Am @ x_prev, np.full(dx, q, np.float32)) @ "x"
        genjax.mv_normal_diag(x, np.full(dx, r, np.float32)) @ "y"
        return x, None

    @genjax.gen
    def q_step(x_prev, y_t):          # the locally optimal proposal of the model (H = I)
        x = genjax.mv_normal_diag((s2 / q ** 2) * (Am @ x_prev) + (s2 / r ** 2) * y_t, np.full(dx, np.sqrt(s2), np.float32)) @ "x"
        return x, None

    def time_filter(bf, chm, args, n=4):
        bf.alias_outputs = True
        for i in range(2):
            bf.run(genjax.key(i), chm, args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            out = bf.run(genjax.key(10 + i), chm, args)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out

    def row(dt, out, Kf, Tn, bytes_per):
        info = out.get("info", {})
        return dict(us_per_step=dt / Tn * 1e6, particle_steps_per_sec=Kf * Tn / dt, log_ml=float(out["log_ml"]), form=info.get("form_name"),
                    launches_per_run=info.get("launches"), grid=info.get("grid"), tiles_per_block=info.get("tiles_per_block"),
                    roofline=dict(bound="hbm", algorithmic_bytes_per_particle_step=bytes_per, bytes_per_step=bytes_per * Kf, us_per_step=dt / Tn * 1e6,
                                  achieved=bytes_per * Kf / (dt / Tn) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=bytes_per * Kf / (dt / Tn) / 1e9 / HBM_PEAK_GBS,
                                  timing="wall clock over whole runs incl. step 0 and the host's calls"))

    carry0 = np.zeros(dx, np.float32)
    chm = CM["y"].set(ys)
    dtb, ob = time_filter(BootstrapFilter(lg_step.scan(n=T), K), chm, (carry0, None))
    dtp, op = time_filter(BootstrapFilter(lg_step.scan(n=T), K, proposal=q_step.scan(n=T), proposal_args=(carry0, ys)), chm, (carry0, None))
    rp = row(dtp, op, K, T, 8 * dx + 24)
    rp.update(log_ml_rel_err=abs(float(op["log_ml"]) - exact) / abs(exact), bootstrap_proposal_us_per_step=dtb / T * 1e6, ratio_to_bootstrap=dtp / dtb,
              note="log w = log p(x_t | x_{t-1}) + log p(y_t | x_t) - log q(x_t | x_{t-1}, y_t): proposal sites (GJX_SITE_PROPOSAL) and the model's latent "
                   "scored at the proposal's draw (GJX_MODE_OBS_PROPOSED) in ONE step program")
    res["scan_filter_lgssm_optimal_proposal_T256_K2e18"] = rp
    dtm, om = time_filter(BootstrapFilter(lg_step.scan(n=T), K, resampler="multinomial"), chm, (carry0, None))
    rm = row(dtm, om, K, T, 8 * dx + 24)
    rm.update(log_ml_rel_err=abs(float(om["log_ml"]) - exact) / abs(exact), systematic_us_per_step=dtb / T * 1e6, ratio_to_systematic=dtm / dtb,
              note="GJX_FILTER_MULTINOMIAL: multinomial resampling by sorted uniforms (exponential spacings) INSIDE the one-launch filter kernel: "
                   "the spacing sums ride the tile granules of the step's one rendezvous (pf_core's MULTI flavour)")
    res["scan_filter_lgssm_multinomial_T256_K2e18"] = rm
    with open(os.path.join(ROOT, "tests", "golden", "sv_pf_float64.json")) as f:
        fx = json.load(f)
    phi, sigma, ysv = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)

    @genjax.gen
    def sv_step(x_prev, _):
        x = genjax.normal(phi * x_prev, sigma) @ "x"
        genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
        return x, None

    Tn = len(ysv)
    dt0, o0 = time_filter(BootstrapFilter(sv_step.scan(n=Tn), K), CM["y"].set(ysv), (0.0, None))
    dt2, o2 = time_filter(BootstrapFilter(sv_step.scan(n=Tn), K, rejuvenate=dict(n_moves=2, scale=0.3)), CM["y"].set(ysv), (0.0, None))
    r2 = row(dt2, o2, K, Tn, 8 * 1 + 24 + 8)
    r2.update(without_moves_us_per_step=dt0 / Tn * 1e6, accept_rate=float(o2["accepted_total"]) / (2.0 * K * (Tn - 2)),
              float64_filter_mean=fx["log_ml_mean"], float64_filter_std=fx["log_ml_std"], z=(float(o2["log_ml"]) - fx["log_ml_mean"]) / fx["log_ml_std"],
              note="two random-walk Metropolis moves per particle behind every resampling, target = the previous step's density re-scored by code "
                   "generated from the step program, inside the filter kernel")
    res["scan_filter_stochastic_volatility_2_moves_T256_K2e18"] = r2

    # HMC over a plate-tagged program: ls ~ N(0,1), beta ~ N(0, I_P); per datum eta_i ~ N(x_i . beta, exp(ls)), y_i ~ bernoulli(logits = eta_i)
    N, P, n, L = 256, 4, 1 << 14, 20
    rs = np.random.default_rng(0)
    X = (0.5 * rs.standard_normal((N, P))).astype(np.float32)

    @genjax.gen
    def kern(x_row, beta, ls):
        eta = genjax.normal(x_row @ beta, genjax.exp(ls)) @ "eta"
        return genjax.bernoulli(logits=eta) @ "y"

    @genjax.gen
    def model():
        ls = genjax.normal(0.0, 1.0) @ "ls"
        beta = genjax.normal(np.zeros(P, np.float32), 1.0) @ "beta"
        kern.vmap(in_axes=(0, None, None))(X, beta, ls) @ "k"

    y = (rs.uniform(size=N) < 0.5).astype(np.float32)
    lat = ["ls", "beta"] + [(("k", "eta"), i) for i in range(N)]
    prog, _, _ = model.pack((), CM["k", "y"].set(y), False, selected=("ls", "beta"), per_particle=tuple(lat), plates="hmc")
    prog_all, _, _ = model.pack((), CM["k", "y"].set(y), False, selected=tuple(lat), per_particle=tuple(lat), plates="hmc")
    ch0 = (torch.randn((prog.n_slots, n), device=dev) * 0.3).contiguous()
    hm = {}
    for engine in ("gen", "interp", "gen_latents_moved_too", "interp_latents_moved_too"):
        moved = engine.endswith("_too")
        prog_ = prog_all if moved else prog
        prog_run, engine_name = prog_, engine
        engine = engine.split("_")[0]
        old = os.environ.get("GJX_HMC_ENGINE")
        os.environ["GJX_HMC_ENGINE"] = engine
        try:
            eng = kernels.hmc_engine(prog_run)
            out = kernels.hmc(prog_run, (1, 2), ch0.clone(), 0.01, L, False, True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(3):
                out = kernels.hmc(prog_run, (1, 3 + i), ch0.clone(), 0.01, L, False, True, ws=out["_ws"])
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 3
        finally:
            if old is None:
                del os.environ["GJX_HMC_ENGINE"]
            else:
                os.environ["GJX_HMC_ENGINE"] = old
        # per chain and gradient sweep: the state rows read once per instance (eta_i: 4 B) — the rest lives in registers / LDS; with the
        # latents moved too their position, momentum and gradient rows go through the workspace: 10 accesses per row and leapfrog step
        hm[engine_name] = dict(engine=eng, ms_per_move=ms, chain_leapfrogs_per_sec=n * L / (ms * 1e-3), accept_rate=float(out["accepted"].mean()),
                          roofline=dict(bound="hbm", algorithmic_bytes_per_launch=4.0 * N * n * (L + 1), kernel_us=ms * 1e3,
                                        achieved=4.0 * N * n * (L + 1) / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=4.0 * N * n * (L + 1) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, timing="event pair around 3 back-to-back moves (incl. a clone of the state)"))
    hm["speedup"] = hm["interp"]["ms_per_move"] / hm["gen"]["ms_per_move"]
    hm["speedup_latents_moved_too"] = hm["interp_latents_moved_too"]["ms_per_move"] / hm["gen_latents_moved_too"]["ms_per_move"]
    hm["device_sites"], hm["plate_instances"], hm["chains"], hm["leapfrog"] = prog.n_sites, N, n, L
    res["hmc_plate_regression_latent_per_datum_N256"] = hm

    # HMC over (mu, log sigma) of the 4096-datum vmapped mixture, assignments z fixed per chain: x_i ~ normal(mu[z_i], exp(ls)) reads a
    # row of a LATENT choice (GJX_P_VGATHER)
    Nm, nm, Lm = 4096, 1 << 13, 10
    true_mu = np.array([-2.5, 0.0, 3.0], np.float32)
    zt = rs.integers(0, 3, Nm)
    ysm = (true_mu[zt] + 0.6 * rs.standard_normal(Nm)).astype(np.float32)
    lg = np.array([0.2, -0.3, 0.1], np.float32)

    @genjax.gen
    def mk(mu, ls, lg_):
        z = genjax.categorical(logits=lg_) @ "z"
        return genjax.normal(mu[z], genjax.exp(ls)) @ "x"

    @genjax.gen
    def mix():
        mu = genjax.normal(np.zeros(3, np.float32), 3.0) @ "mu"
        ls = genjax.normal(0.0, 1.0) @ "ls"
        mk.repeat(n=Nm)(mu, ls, lg) @ "k"

    latm = ["mu", "ls"] + [(("k", "z"), i) for i in range(Nm)]
    pm, _, _ = mix.pack((), CM["k", "x"].set(ysm), False, selected=("mu", "ls"), per_particle=tuple(latm), plates="hmc")
    chm0 = torch.empty((pm.n_slots, nm), device=dev)
    chm0[:3] = torch.as_tensor(true_mu, device=dev)[:, None] + 0.03 * torch.randn((3, nm), device=dev)
    chm0[3] = math.log(0.6) + 0.02 * torch.randn(nm, device=dev)
    chm0[4:] = torch.as_tensor(zt, device=dev, dtype=torch.float32)[:, None]
    mm = {}
    for engine in ("gen", "interp"):
        old = os.environ.get("GJX_HMC_ENGINE")
        os.environ["GJX_HMC_ENGINE"] = engine
        try:
            eng = kernels.hmc_engine(pm)
            out = kernels.hmc(pm, (1, 2), chm0, 1e-3, Lm, False, True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 if engine == "gen" else 1
            a.record()
            for i in range(reps):
                out = kernels.hmc(pm, (1, 3 + i), chm0, 1e-3, Lm, False, True, ws=out["_ws"])
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
        finally:
            if old is None:
                del os.environ["GJX_HMC_ENGINE"]
            else:
                os.environ["GJX_HMC_ENGINE"] = old
        byt = 4.0 * Nm * nm * (Lm + 1)            # the chain's assignment rows, read once per gradient sweep
        mm[engine] = dict(engine=eng, ms_per_move=ms, chain_leapfrogs_per_sec=nm * Lm / (ms * 1e-3), accept_rate=float(out["accepted"].mean()),
                          roofline=dict(bound="hbm", algorithmic_bytes_per_launch=byt, kernel_us=ms * 1e3, achieved=byt / (ms * 1e-3) / 1e9,
                                        peak=HBM_PEAK_GBS, unit="GB/s", frac=byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        timing="event pair around back-to-back moves"))
    mm["speedup"] = mm["interp"]["ms_per_move"] / mm["gen"]["ms_per_move"]
    mm["device_sites"], mm["plate_instances"], mm["chains"], mm["leapfrog"] = pm.n_sites, Nm, nm, Lm
    res["hmc_mixture_latent_means_N4096"] = mm

    # HMC over every state of a T = 256 stochastic-volatility Scan (512 sites, 256 selected values): the generated kernel rolls the Scan
    Ts, ns_, Ls = len(ysv), 1 << 13, 10

    @genjax.gen
    def sv_model():
        sv_step.scan(n=Ts)(0.0, None) @ "s"

    xs_ = [(("s", "x"), t) for t in range(Ts)]
    ps, _, _ = sv_model.pack((), CM["s", "y"].set(ysv), False, selected=tuple(xs_), per_particle=tuple(xs_))
    chs = (0.3 * torch.randn((ps.n_slots, ns_), device=dev)).contiguous()
    sm = {}
    for engine in ("gen", "interp"):
        old = os.environ.get("GJX_HMC_ENGINE")
        os.environ["GJX_HMC_ENGINE"] = engine
        try:
            eng = kernels.hmc_engine(ps)
            out = kernels.hmc(ps, (1, 2), chs.clone(), 0.01, Ls, False, True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 if engine == "gen" else 1
            a.record()
            for i in range(reps):
                out = kernels.hmc(ps, (1, 3 + i), chs.clone(), 0.01, Ls, False, True, ws=out["_ws"])
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
        finally:
            if old is None:
                del os.environ["GJX_HMC_ENGINE"]
            else:
                os.environ["GJX_HMC_ENGINE"] = old
        byt = 4.0 * Ts * ns_ * (10 * Ls + 2)       # position, momentum and gradient rows: 10 accesses per row and leapfrog step
        sm[engine] = dict(engine=eng, ms_per_move=ms, chain_leapfrogs_per_sec=ns_ * Ls / (ms * 1e-3), accept_rate=float(out["accepted"].mean()),
                          roofline=dict(bound="hbm", algorithmic_bytes_per_launch=byt, kernel_us=ms * 1e3, achieved=byt / (ms * 1e-3) / 1e9,
                                        peak=HBM_PEAK_GBS, unit="GB/s", frac=byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        timing="event pair around back-to-back moves (incl. a clone of the state)"))
    sm["speedup"] = sm["interp"]["ms_per_move"] / sm["gen"]["ms_per_move"]
    sm["device_sites"], sm["steps"], sm["chains"], sm["leapfrog"] = ps.n_sites, Ts, ns_, Ls
    res["hmc_stochastic_volatility_scan_T256"] = sm
    return res


def run_round6(dev):
    """Round-6 paths, short runs: general expressions (GJX_P_EXPR) — a 16 -> 8 -> 1 network likelihood under ImportanceK on the
    generated kernel and on the interpreter; a network classifier vmapped over the data as ONE plate (generated plate kernel vs
    interpreter; HMC over its weights, generated vs interpreter); the generic filter on a nonlinear model whose step means are
    expression blocks; the generic filter over a model with latent parameters in front of the Scan."""
    import math as m_
    import genjax_amd as genjax
    from genjax_amd import C as CM
    from genjax_amd import _abi as A
    from genjax_amd import kernels
    from genjax_amd.inference import BootstrapFilter
    from genjax_amd.program import PackedProgram
    res = {}

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def with_engine(var, val, fn):
        old = os.environ.get(var)
        os.environ[var] = val
        try:
            return fn()
        finally:
            if old is None:
                del os.environ[var]
            else:
                os.environ[var] = old

    # (1) bernoulli(logits = w2 . tanh(W1 x)), x latent (16): ImportanceK at K = 2^20
    W1 = np.random.default_rng(0).standard_normal((8, 16)) * 0.4
    w2 = np.random.default_rng(1).standard_normal(8)

    @genjax.gen
    def mlp():
        x = genjax.mv_normal_diag(np.zeros(16, np.float32), np.ones(16, np.float32)) @ "x"
        genjax.bernoulli(logits=w2 @ genjax.tanh(W1 @ x)) @ "y"

    sl, _ = mlp.site_list(())
    prog = PackedProgram(sl, {"y": A.MODE_OBS_TAB}, {"y": np.float32(1.0)})
    K = 1 << 20
    row = {}
    for eng in ("gen", "interp"):
        out = with_engine("GJX_ENGINE", eng, lambda: kernels.run_program(prog, (0, 1), K, want_weight=False, want_lse=False))
        us = with_engine("GJX_ENGINE", eng, lambda: timed(lambda: kernels.run_program(prog, (0, 1), K, out=out, want_weight=False, want_lse=False), 20 if eng == "gen" else 5))
        row[eng] = dict(kernel_us=us, engine=with_engine("GJX_ENGINE", eng, lambda: kernels.program_engine(prog)))
    algo = (4 * 16 + 8) * K
    row["gen"].update(achieved=algo / (row["gen"]["kernel_us"] * 1e-6) / 1e9, unit="GB/s", frac=algo / (row["gen"]["kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                      algorithmic_bytes_per_launch=algo)
    row["generated_vs_interpreter"] = row["interp"]["kernel_us"] / row["gen"]["kernel_us"]
    row["note"] = "the 17-node block (8 LINV rows over the 16 latent values, 8 tanh, one LINN row) emitted inline; 72 B written per particle"
    res["mlp_16_8_1_importance_K2e20"] = row

    # (2) a network classifier vmapped over N = 1024 observations (one plate site, strided block), K = 2^14 particles
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    N, DI, DH, Kp = 1024, 16, 8, 1 << 14
    model, X, Y, _ = H.bnn_model(N, DI, DH)
    bprog, _, _ = model.pack((), CM["obs", "y"].set(Y), True)
    row = dict(device_sites=bprog.n_sites, instances=N, particles=Kp)
    for eng in ("gen", "interp"):
        out = with_engine("GJX_ENGINE", eng, lambda: kernels.run_program(bprog, (0, 1), Kp, want_weight=False, want_lse=False))
        us = with_engine("GJX_ENGINE", eng, lambda: timed(lambda: kernels.run_program(bprog, (0, 1), Kp, out=out, want_weight=False, want_lse=False), 10 if eng == "gen" else 3))
        row[eng] = dict(kernel_us=us)
    flops = Kp * N * (2 * DI * DH + 2 * DH + 12 * DH)
    row["gen"]["tflops"] = flops / (row["gen"]["kernel_us"] * 1e-6) / 1e12
    row["generated_vs_interpreter"] = row["interp"]["kernel_us"] / row["gen"]["kernel_us"]
    res["network_classifier_plate_N1024_K2e14"] = row
    # HMC over the weights of a smaller network through the plate (generated HMC kernel vs interpreter)
    N2, n_ch = 512, 1 << 13
    model2, X2, Y2, _ = H.bnn_model(N2, 4, 3, seed=2)
    sel = tuple(f"W1_{j}" for j in range(3)) + ("w2",)
    hp, _, _ = model2.pack((), CM["obs", "y"].set(Y2), False, selected=sel, per_particle=sel, plates="hmc")
    ch0 = torch.as_tensor((np.random.default_rng(5).standard_normal((hp.n_slots, n_ch)) * 0.3).astype(np.float32), device=dev)
    row = dict(observations=N2, chains=n_ch, leapfrog=10)
    for eng in ("gen", "interp"):
        e_ = with_engine("GJX_HMC_ENGINE", eng, lambda: kernels.hmc_engine(hp))
        ms = with_engine("GJX_HMC_ENGINE", eng, lambda: timed(lambda: kernels.hmc(hp, (1, 2), ch0, 0.01, 10, False, True), 5 if eng == "gen" else 2)) * 1e-3
        row[eng] = dict(ms_per_move=ms, engine=e_)
    row["generated_vs_interpreter"] = row["interp"]["ms_per_move"] / row["gen"]["ms_per_move"]
    res["hmc_network_weights_through_plate"] = row
    # ... and the 16 -> 8 -> 1 network's 136 weights over 1024 observations: beyond the register budget, the generated kernel keeps the
    # chain state in LDS columns (HmcPlan::big)
    model3, X3, Y3, _ = H.bnn_model(1024, 16, 8, seed=2)
    sel3 = tuple(f"W1_{j}" for j in range(8)) + ("w2",)
    hp3, _, _ = model3.pack((), CM["obs", "y"].set(Y3), False, selected=sel3, per_particle=sel3, plates="hmc")
    n3 = 1 << 12
    ch3 = torch.as_tensor((np.random.default_rng(6).standard_normal((hp3.n_slots, n3)) * 0.3).astype(np.float32), device=dev)
    row = dict(observations=1024, weights=136, chains=n3, leapfrog=10)
    for eng in ("gen", "interp"):
        e_ = with_engine("GJX_HMC_ENGINE", eng, lambda: kernels.hmc_engine(hp3))
        ms = with_engine("GJX_HMC_ENGINE", eng, lambda: timed(lambda: kernels.hmc(hp3, (1, 2), ch3, 0.005, 10, False, True), 3 if eng == "gen" else 1)) * 1e-3
        row[eng] = dict(ms_per_move=ms, engine=e_)
    fl = n3 * 11 * 1024 * (4 * 16 * 8 + 4 * 8 + 30 * 8)            # forward + reverse sweep of the block per observation, 11 sweeps
    row["gen"]["tflops"] = fl / (row["gen"]["ms_per_move"] * 1e-3) / 1e12
    row["generated_vs_interpreter"] = row["interp"]["ms_per_move"] / row["gen"]["ms_per_move"]
    res["hmc_network_16_8_1_weights_lds_state"] = row

    # (2b) two INDEPENDENT ImportanceK steps in flight on two HIP streams (config 2's steps are independent runs: different keys): the
    #      VALU-bound propagate kernel of one step overlaps the HBM-bound gather of the other.  Plain launches only (the tile-scaled
    #      resampler: no co-resident grid beside another kernel).  NOT the headline: a kernel's own duration stretches when it shares the
    #      device, so roofline.frac is quoted on the unpipelined step; this is what a caller with independent batches gets.
    from genjax_amd import workloads
    gprog, _ = workloads.gmm_program(D=D, C=C)
    Kg = 1 << 20
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = []
    for _ in range(2):
        ws_a, ws_b = kernels.workspace(A.OP_RUN, Kg, dev), kernels.workspace(A.OP_RESAMPLE, Kg, dev)
        o_ = kernels.run_program(gprog, (0, 1), Kg, ws=ws_a, want_weight=False, want_lse=False, want_tiles=True)
        bufs.append(dict(ws=ws_a, ws2=ws_b, out=o_, rows=torch.empty_like(o_["choices"]), lse=torch.empty(4, dtype=torch.float32, device=dev)))
    torch.cuda.synchronize()

    def one_step(i, b):
        kernels.run_program(gprog, (0, 1 + i), Kg, ws=b["ws"], out=b["out"], want_weight=False, want_lse=False, want_tiles=True)
        pr = b["out"]["_partials"]
        kernels.resample_gather_tiled(b["out"]["logw"], 0.37, b["out"]["choices"], partials=(b["ws"], pr.count()), tiles=pr.tiles, lse_out=b["lse"],
                                      K_total=Kg, out=b["rows"], ws=b["ws2"])

    def run_steps(n, pipelined):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if pipelined:
                with torch.cuda.stream(streams[i & 1]):
                    one_step(i, bufs[i & 1])
            else:
                one_step(i, bufs[0])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    for pip in (False, True):
        run_steps(50, pip)
    serial = sorted(run_steps(400, False) for _ in range(3))[1]
    piped = sorted(run_steps(400, True) for _ in range(3))[1]
    res["two_independent_steps_in_flight"] = dict(
        us_per_step_one_stream=serial * 1e6, us_per_step_two_streams=piped * 1e6, particle_steps_per_sec_two_streams=Kg / piped, speedup=serial / piped,
        log_ml=float(bufs[1]["lse"][3]),
        note="gjx_run_program_ex (tile totals left) + gjx_resample_gather_tiled per step; independent steps alternate between two streams and two "
             "sets of buffers; not the headline (see the comment in bench.py)")

    # (3) the generic filter on the nonlinear benchmark model (both step means are expression blocks of the carry)
    def time_filter(bf, chm, args, n=5):
        bf.alias_outputs = True
        for i in range(2):
            bf.run(genjax.key(i), chm, args)
        runs = []
        for i in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = bf.run(genjax.key(10 + i), chm, args)
            torch.cuda.synchronize()
            runs.append(time.perf_counter() - t0)
        return sorted(runs)[len(runs) // 2], out

    T, Kf = 256, 1 << 18

    @genjax.gen
    def nl_step(x_prev, c_t):
        x = genjax.normal(0.5 * x_prev + 25.0 * x_prev / (1.0 + x_prev * x_prev) + c_t, m_.sqrt(10.0)) @ "x"
        genjax.normal(x * x / 20.0, 1.0) @ "y"
        return x, None

    rs = np.random.default_rng(0)
    cs = (8.0 * np.cos(1.2 * np.arange(T))).astype(np.float32)
    xv, ysl = 0.1, []
    for t in range(T):
        xv = 0.5 * xv + 25.0 * xv / (1 + xv * xv) + cs[t] + m_.sqrt(10.0) * rs.standard_normal()
        ysl.append(xv * xv / 20.0 + rs.standard_normal())
    ys = np.asarray(ysl, np.float32)
    bf = BootstrapFilter(nl_step.scan(n=T), Kf)
    dt, out = time_filter(bf, CM["y"].set(ys), (np.float32(0.1), cs))
    res["scan_filter_nonlinear_expressions_T256_K2e18"] = dict(us_per_step=dt / T * 1e6, form=bf.last_info["form_name"], launches=bf.last_info["launches"],
                                                              log_ml=float(out["log_ml"]), particle_steps_per_sec=Kf * T / dt,
                                                              note="x_t ~ normal(x/2 + 25 x / (1 + x^2) + 8 cos(1.2 t), sqrt 10), y_t ~ normal(x^2 / 20, 1): GJX_P_EXPR blocks inside gjx_gen_pf")
    # (4) latent parameters in front of the Scan (stochastic volatility with latent phi and log sigma)
    from genjax_amd.inference import BootstrapFilter as BF

    @genjax.gen
    def sv_step(carry, _):
        x_prev, phi, ls = carry
        x = genjax.normal(phi * x_prev, genjax.exp(ls)) @ "x"
        genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
        return (x, phi, ls), None

    @genjax.gen
    def sv_model():
        phi = genjax.uniform(0.8, 0.99) @ "phi"
        ls = genjax.normal(m_.log(0.3), 0.3) @ "log_sigma"
        sv_step.scan(n=T)((0.0, phi, ls), None) @ "chain"

    ysv = (np.random.default_rng(3).standard_normal(T) * 1.2).astype(np.float32)
    bf2 = BF(sv_model, Kf)
    dt2, out2 = time_filter(bf2, CM["chain", "y"].set(ysv), ())
    lw = out2["logw"].double()
    w = torch.exp(lw - lw.max())
    res["scan_filter_latent_parameters_T256_K2e18"] = dict(us_per_step=dt2 / T * 1e6, form=bf2.last_info["form_name"], launches=bf2.last_info["launches"],
                                                          log_ml=float(out2["log_ml"]),
                                                          posterior_mean_phi=float((w * bf2.latent(out2, "phi")[0].double()).sum() / w.sum()),
                                                          note="phi and log sigma drawn in front of the Scan travel with the particles (GJX_SITE_CARRIED inputs)")
    # (5) moves that are the library's own requests (inference/filter_moves.py): the filter step by step as device calls, an HMC move
    # (gjx_hmc on the step-local target, accept fused) and a Rejuvenate proposal move behind every resampling; stochastic volatility
    from genjax_amd import S
    from genjax_amd.inference import HMC, Rejuvenate
    with open(os.path.join(ROOT, "tests", "golden", "sv_pf_float64.json")) as f:
        fx = json.load(f)
    phi_, sig_, yfx = fx["phi"], fx["sigma"], np.asarray(fx["y"], np.float32)

    @genjax.gen
    def sv1(x_prev, _):
        x = genjax.normal(phi_ * x_prev, sig_) @ "x"
        genjax.normal(0.0, genjax.exp(0.5 * x)) @ "y"
        return x, None

    Km = 1 << 16
    for name, mv in (("hmc_L3", [HMC(S["x"], 0.25, 3)]), ("rejuvenate_proposal", [{"x": Rejuvenate(genjax.normal, lambda chm: (chm.get_value(), 0.2))}]),
                     ("no_move_step_by_step", None)):
        bfm = BootstrapFilter(sv1.scan(n=len(yfx)), Km, moves=mv)
        if mv is None:
            from genjax_amd.inference.filter_moves import run_with_moves
            bfm.run = lambda k_, c_, a_, _b=bfm, _none=[]: run_with_moves(_b, k_, c_, a_, _none)
        dtm, om = time_filter(bfm, CM["y"].set(yfx), (0.0, None), n=3)
        acc = om.get("accepted") or []
        res[f"scan_filter_sv_moves_{name}_T256_K2e16"] = dict(
            us_per_step=dtm / len(yfx) * 1e6, log_ml=float(om["log_ml"]), z=(float(om["log_ml"]) - fx["log_ml_mean"]) / (fx["log_ml_std"] * 2.0),
            accept_rate=[a / (Km * (len(yfx) - 1)) for a in acc], distinct_carry=int(torch.unique(om["choices"][0]).numel()),
            form=(om.get("info") or {}).get("form_name"),
            note="ONE HMC move: inside the library's step loop (gjx_filter_opts::hmc_targets: gather, gjx_hmc, propagate per step); proposal moves "
                 "and the no-move reference: the same calls composed from the host (inference/filter_moves.py); z against the float64 fixture's spread scaled to K = 2^16")
    return res


def _config4_worker(rank, world, port, K_total, T, dx, out_dir, verify):
    """one rank of the config-4 dry run (processes sharing ONE device): the sharded filter over peer-mapped windows"""
    import hashlib
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)        # (gloo announces its ranks on stdout: the parent's stdout carries ONE JSON line)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", GJX_PEER_VERIFY="1" if verify else "0")
    from genjax_amd import _abi as A
    from genjax_amd import distributed as DD
    from genjax_amd import kernels, workloads
    from genjax_amd.inference.pf import LinearGaussianSSM
    DD.init_from_env("gloo")
    torch.cuda.set_device(0)
    s = workloads.ssm_problem(dx=dx, T=T)
    ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
    ys = torch.as_tensor(np.asarray(s["y"], np.float32)).cuda()
    ctx = kernels.PeerContext(K_total // world, dx, "cuda")
    cs = ssm.c_struct("cuda")
    o = ctx.ssm_filter(cs, (0, 7), A.RNG_FLAT, ys, want_ancestors=True)
    torch.cuda.synchronize()
    st = ctx.status()
    sha = {k: hashlib.sha256(o[k].cpu().numpy().tobytes()).hexdigest() for k in ("x", "logw", "ancestors")}
    log_ml = float(o["lse_steps"][:, 3].double().sum())
    times = []
    for rep in range(3):
        dist.barrier()
        t0 = time.perf_counter()
        ctx.ssm_filter(cs, (0, 8 + rep), A.RNG_FLAT, ys)
        torch.cuda.synchronize()
        dist.barrier()
        times.append(time.perf_counter() - t0)
    st |= ctx.status()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(dict(status=st, sha=sha, log_ml=log_ml, times=times, share=ctx.ranks_on_device), f)
    ctx.close()
    dist.destroy_process_group()


def run_config4_dry_run(dev, world=8, K_total=1 << 22, T=256, dx=8):
    """BASELINE config 4 at its own size on the ONE device of this box: `world` processes share the GPU, map each other's
    windows (hipIpc) and run gjx_ssm_filter_peer concurrently — correctness of the 8-rank path at full size (bit-identical
    to the one-rank filter, log-ML vs float64 Kalman, verify mode on), NOT a scaling number."""
    import hashlib
    import tempfile
    import torch.multiprocessing as mp
    from genjax_amd import _abi as A
    from genjax_amd import kernels, workloads
    from genjax_amd.inference.pf import LinearGaussianSSM
    out = {}
    for verify in (True, False):
        with tempfile.TemporaryDirectory() as td:
            ctx = mp.get_context("spawn")
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            procs = [ctx.Process(target=_config4_worker, args=(r, world, port, K_total, T, dx, td, verify)) for r in range(world)]
            for p in procs:
                p.start()
            for p in procs:
                p.join(timeout=600)
            if any(p.exitcode != 0 for p in procs):
                return dict(error="a rank failed", exitcodes=[p.exitcode for p in procs])
            res = [json.load(open(os.path.join(td, "rank%d.json" % r))) for r in range(world)]
        out["verify_on" if verify else "verify_off"] = res
    s = workloads.ssm_problem(dx=dx, T=T)
    ssm = LinearGaussianSSM(s["A"], s["q"], s["r"])
    ys = torch.as_tensor(np.asarray(s["y"], np.float32)).to(dev)
    ref = kernels.ssm_filter(ssm.c_struct(dev), (0, 7), A.RNG_FLAT, ys, K_total, weights=A.WEIGHTS_TILE_SCALED)
    torch.cuda.synchronize()
    Kl = K_total // world
    same = True
    for r in range(world):
        want = dict(x=ref["x"][:, r * Kl:(r + 1) * Kl].contiguous(), logw=ref["logw"][r * Kl:(r + 1) * Kl].contiguous(),
                    ancestors=ref["ancestors"][r * Kl:(r + 1) * Kl].contiguous())
        for k, v in want.items():
            h = hashlib.sha256(v.cpu().numpy().tobytes()).hexdigest()
            same = same and all(out[m][r]["sha"][k] == h for m in out)
    exact = golden("ssm_dx8_T256_seed0")
    # one-rank reference timing at the same total size, for the ratio
    t0 = time.perf_counter()
    kernels.ssm_filter(ssm.c_struct(dev), (0, 8), A.RNG_FLAT, ys, K_total, weights=A.WEIGHTS_TILE_SCALED, bufs=ref["_bufs"])
    torch.cuda.synchronize()
    one_rank = time.perf_counter() - t0
    def summary(res):
        tm = sorted(max(r["times"][i] for r in res) for i in range(3))[1]
        return dict(status_words=[r["status"] for r in res], us_per_filter_step=tm / T * 1e6, particle_steps_per_sec=K_total * T / tm,
                    log_ml=res[0]["log_ml"], log_ml_rel_err=abs(res[0]["log_ml"] - exact) / abs(exact))
    return dict(what="config 4 (SSM bootstrap filter, K_total=2^22, T=256, d_x=8) as %d processes sharing ONE MI355X: gjx_ssm_filter_peer over "
                     "hipIpc-mapped windows; the device's memory stands in for xGMI" % world,
                ranks=world, ranks_on_this_device=out["verify_on"][0]["share"], k_particles_total=K_total, T=T,
                bit_identical_to_one_rank_filter=bool(same), verify_on=summary(out["verify_on"]), verify_off=summary(out["verify_off"]),
                one_rank_same_size_us_per_filter_step=one_rank / T * 1e6,
                note="all ranks time-share one device: the step time is a correctness-run figure, not a scaling measurement")


HEADLINE_MAX_BYTES = 4096


def _short(sv, n=140):
    return sv if not isinstance(sv, str) or len(sv) <= n else sv[:n - 1] + "~"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def headline_record(res: dict, extra_path=None) -> dict:
    """The compact object of the driver's line: the contract's keys + roofline + cpu_baseline + parity figures, no prose, no nested
    `extra` (a 25 KB line could not be parsed).  Everything dropped here is in the side file."""
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data"))
    cfg = res.get("config", {})
    out["config"] = {k: _short(v, 200) for k, v in cfg.items() if k != "exchange_stats"}
    ex = cfg.get("exchange_stats")
    if isinstance(ex, dict) and ex.get("transport", "none") != "none":
        out["config"]["exchange_summary"] = _pick(ex, ("transport", "ranks", "status", "transports_agree", "any_status_bit"))
    rf = res.get("roofline", {})
    out["roofline"] = {k: _short(v, 100) for k, v in _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_us",
                                                                "algorithmic_bytes_per_launch", "launches_per_step")).items()}
    tp = rf.get("traffic_from_profiles")
    if isinstance(tp, dict):
        out["roofline"]["traffic_from_profiles"] = _pick(tp, ("bytes_per_launch", "file"))
    if isinstance(rf.get("back_to_back"), dict):
        out["roofline"]["back_to_back_frac"] = rf["back_to_back"].get("frac")
    if "cpu_baseline" in res:
        cb = res["cpu_baseline"]
        out["cpu_baseline"] = {k: _short(v, 180) for k, v in _pick(cb, ("value", "unit", "cores", "kind", "sample")).items()}
    for k in ("log_ml", "log_ml_exact", "log_ml_rel_err", "log_ml_z", "accept_rate", "energy_error_rms"):
        if k in res:
            out[k] = res[k]
    if isinstance(res.get("roofline_jax32_stream"), dict):
        out["roofline_jax32_stream_frac"] = res["roofline_jax32_stream"].get("frac")
    other = {}
    for name, r in (res.get("extra") or {}).items():
        if name in ("ssm", "hmc") and isinstance(r, dict):
            o = _pick(r, ("value", "unit", "ms_per_step", "log_ml_rel_err", "log_ml_z", "accept_rate"))
            o["workload"] = _short(r.get("config", {}).get("workload", name), 120)
            o["roofline"] = {k: _short(v, 60) for k, v in _pick(r.get("roofline", {}), ("bound", "kernel", "frac", "achieved", "unit", "kernel_us")).items()}
            other[name] = o
    if other:
        out["other_configs"] = other
    if "jit" in res:
        out["jit_compiles_at_runtime"] = res["jit"]["hiprtc_compiles"]
    if extra_path:
        out["full_record"] = os.path.relpath(extra_path, ROOT) if extra_path.startswith(ROOT) else extra_path
    return out


def headline_line(res: dict, extra_path=None) -> str:
    rec = headline_record(res, extra_path)
    line = json.dumps(rec, separators=(",", ":"))
    if len(line) > HEADLINE_MAX_BYTES:          # never let a long string cost the round's measurement: drop the optional blocks
        for k in ("other_configs", "roofline_jax32_stream_frac", "full_record"):
            rec.pop(k, None)
        rec["config"] = _pick(rec["config"], ("workload", "k_particles_per_gpu", "k_particles_total", "sharding", "exchange"))
        line = json.dumps(rec, separators=(",", ":"))
    assert len(line) <= 2 * HEADLINE_MAX_BYTES and "\n" not in line
    return line


def write_full_record(res: dict, args) -> str:
    """everything measured (incl. `extra` and the long notes) as indented JSON beside bench.py — and under gpurun_out/ when that
    exists, so a GPU-box run brings it home"""
    path = args.extra_file
    try:
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
    except OSError:
        path = None
    gp = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(gp) and os.access(gp, os.W_OK):
        try:
            with open(os.path.join(gp, "bench_extra_%s.json" % args.workload), "w") as f:
                json.dump(res, f, indent=1)
        except OSError:
            pass
    return path


def respawn(n: int) -> None:
    """Replace this process by `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>`
    (one rank per GPU over RCCL; rendezvous on 127.0.0.1 and a free port)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this stack
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def timed_loop(args, world, dev, step):
    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def region(reps):
        """reps x args.steps steps enqueued back to back between ONE pair of barriers -> seconds per args.steps steps (max over ranks)"""
        last = None
        barrier()
        t0 = time.perf_counter()
        for r in range(reps):
            for i in range(args.steps):
                last = step(i, r == 0)              # the kernel's HIP-event samples ride in the first unit of every bracket (events are not free)
        barrier()
        dt = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / reps, last

    for i in range(args.warmup):
        step(i % max(args.steps, 1), False)
    # The timed unit is EXACTLY args.steps steps.  A short unit (20 steps = 1.2 ms) between two host barriers is mostly barrier
    # and queue-drain edge, so `reps` units are enqueued back to back between ONE pair of barriers (reps sized from a first,
    # calibrating unit so that a bracket spans >= MIN_TIMED_S) and the bracket's time is divided by reps; that bracket is
    # repeated and the MEDIAN reported.  Every rank takes the same decisions (they see the same max-reduced times).
    est, last = region(1)
    reps = max(1, min(4096, int(math.ceil(MIN_TIMED_S / max(est, 1e-7)))))
    dts = []
    for _ in range(5 if reps > 1 else 1):
        dt, last = region(reps)
        dts.append(dt)
    if reps == 1:
        dts.append(est)
    timed_loop.last_regions = dts
    timed_loop.last_reps = reps
    return sorted(dts)[len(dts) // 2], last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["gmm", "ssm", "hmc"], default="gmm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--k-per-gpu", type=int, default=K_PER_GPU)
    ap.add_argument("--leapfrog", type=int, default=1000)
    ap.add_argument("--weak", action="store_true", help="ssm with --gpus N > 1: 2^19 particles per GPU instead of 2^22 in total")
    ap.add_argument("--ssm-weights", choices=["tile_scaled", "global_max"], default="tile_scaled",
                    help="ssm on one GPU: fixed-point scheme of the filter's systematic resampler (include/gjx.h)")
    ap.add_argument("--ssm-k-total", type=int, default=0, help="ssm: total number of particles (overrides the config-3/4 sizes)")
    ap.add_argument("--api", action="store_true", help="gmm on one GPU: print only the API-level measurement (extra.api)")
    ap.add_argument("--no-extra", action="store_true", help="gmm on one GPU: skip the short ssm / hmc runs summarised under other_configs")
    ap.add_argument("--extra", action="store_true",
                    help="gmm on one GPU: also run the round's other measurements (generated kernels, generic filter, plates, HMC engines, "
                         "config-4 dry run, API step); they are written to the side file (--extra-file), not to the line")
    ap.add_argument("--extra-file", default=os.environ.get("GJX_BENCH_EXTRA", os.path.join(ROOT, "bench_extra.json")),
                    help="where the full record (everything measured, incl. long notes) is written; the stdout line stays compact")
    ap.add_argument("--event-samples", type=int, default=16,
                    help="number of timed steps whose propagate+reweight kernel is bracketed with HIP events "
                         "(timing events are not free on ROCm — hundreds of live ones slow every launch — so the "
                         "kernel duration is sampled at evenly spaced steps INSIDE the timed region)")
    args = ap.parse_args()
    dflt = {"gmm": (200, 20), "ssm": (10, 2), "hmc": (5, 1)}[args.workload]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        respawn(args.gpus)                   # plain `python bench.py --gpus N`: become N ranks, one per GPU
    from genjax_amd import distributed as DD
    rank, world = DD.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one GPU per rank; with fewer visible devices than ranks (a dry run of the multi-rank path on one GPU) the ranks share them
    local = 0 if os.environ.get("GJX_ALL_ON_DEVICE0") else int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.api:
        print(json.dumps(dict(metric="particle_steps_per_sec_api", **run_api(dev, args.k_per_gpu))))
        return
    res = {"gmm": run_gmm, "ssm": run_ssm, "hmc": run_hmc}[args.workload](args, rank, world, dev)
    if args.workload == "gmm" and world == 1 and res is not None and not args.no_extra:
        # the other single-GPU rows of BASELINE.json, short runs: their summaries ride in the line under other_configs
        extra = {}
        for name, fn, st in (("ssm", run_ssm, 3), ("hmc", run_hmc, 2)):
            a2 = argparse.Namespace(**vars(args))
            a2.workload, a2.steps, a2.warmup, a2.no_cpu_baseline = name, st, 1, True
            r2 = fn(a2, rank, world, dev)
            extra[name] = {k: r2[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline") if k in r2}
            for k in ("log_ml_rel_err", "log_ml_z", "accept_rate", "other_weight_scheme"):
                if k in r2:
                    extra[name][k] = r2[k]
        if args.extra:
            # everything else the round measured (generated kernels, the generic filter, plates, HMC engines, the config-4 dry run,
            # the API-level step): minutes of work and tens of KB of JSON — they go to the side file, never onto the driver's line
            for name, fn in (("sharded_one_rank", run_sharded_one_rank), ("codegen", run_codegen), ("hmc_generic", run_hmc_generic),
                             ("hmc_generated", run_hmc_generated), ("round4", run_round4), ("round5", run_round5), ("round6", run_round6),
                             ("config4_dry_run", run_config4_dry_run), ("round3", run_round3)):
                try:
                    extra[name] = fn(dev)
                except Exception as e:                  # (reported, never fatal for the headline line)
                    extra[name] = dict(error=repr(e))
            api = run_api(dev, args.k_per_gpu)
            api["vs_kernel_level_step"] = api["ms_per_step"] / res["ms_per_step"]
            extra["api"] = api
        res["extra"] = extra
    if rank == 0 and res is not None:
        from genjax_amd import kernels
        res["jit"] = kernels.jit_stats()
        path = write_full_record(res, args)
        print(headline_line(res, path))            # the LAST stdout line: compact, what the driver parses
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
