"""launch shapes of the generated mixture kernel next to the hand-fused one (K = 2^20 by default, KLOG=..): grids (GJX_GEN_GRID) and particles
per lane (GJX_GEN_PPT); each variant behind a train of 400 launches, median and minimum of 15 dispatch-timed launches, two passes.  DESIGN.md §9."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from genjax_amd import _abi as A, kernels, workloads
dev = torch.device("cuda:0")
K = 1 << int(os.environ.get("KLOG", "20"))
prog, _ = workloads.gmm_program(D=16, C=8)
ws = kernels.workspace(A.OP_RUN, K, dev)
def timed(engine, env, warm=400):
    old = {k: os.environ.get(k) for k in list(env) + ["GJX_ENGINE"]}
    os.environ["GJX_ENGINE"] = engine
    os.environ.update(env)
    try:
        out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False)
        for i in range(warm):
            kernels.run_program(prog, (0, 2 + i), K, ws=ws, out=out, want_weight=False)
        tm = [kernels.DispatchTimer() for _ in range(15)]
        for i, t in enumerate(tm):
            kernels.run_program(prog, (0, 2 + i), K, ws=ws, out=out, want_weight=False, timer=t)
        torch.cuda.synchronize()
        us = sorted(t.elapsed_us() for t in tm)
        lml = float(out["lse"][3])
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return round(us[7], 2), round(us[0], 2), lml
variants = [("hand", "auto", {}), ("gen", "gen", {})]
for g in (512, 640, 768, 896):
    variants.append((f"gen_grid{g}", "gen", {"GJX_GEN_GRID": str(g)}))
for p in (2, 4):
    for g in (512, 1024, 2048):
        variants.append((f"gen_ppt{p}_grid{g}", "gen", {"GJX_GEN_PPT": str(p), "GJX_GEN_GRID": str(g)}))
res = {}
for rep in range(2):
    for name, eng, env in variants:
        res.setdefault(name, []).append(timed(eng, env))
for k, v in res.items(): print(k, v)
