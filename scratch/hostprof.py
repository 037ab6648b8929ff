import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from genjax_amd import _abi as A, kernels
import helpers as H
prog, g = H.gmm()
K = 1 << 20
dev = torch.device("cuda", 0)
ws = kernels.workspace(A.OP_RUN, K, dev); ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False)
rows = torch.empty_like(out["choices"]); zero = torch.zeros(1, dtype=torch.int64, device=dev)
T = {k: 0.0 for k in ("run", "cumsum", "cat", "sys", "gather")}
def step(i, sync):
    t = time.perf_counter()
    kernels.run_program(prog, (0, 1 + i), K, ws=ws, out=out, want_weight=False)
    if sync: torch.cuda.synchronize()
    t1 = time.perf_counter(); T["run"] += t1 - t
    cum, total = kernels.weight_cumsum(out["logw"], True, out["lse"], ws=ws2)
    if sync: torch.cuda.synchronize()
    t2 = time.perf_counter(); T["cumsum"] += t2 - t1
    bt = torch.cat([zero, total])
    if sync: torch.cuda.synchronize()
    t3 = time.perf_counter(); T["cat"] += t3 - t2
    anc = kernels.resample_systematic(cum, bt, 0.3, K)
    if sync: torch.cuda.synchronize()
    t4 = time.perf_counter(); T["sys"] += t4 - t3
    kernels.gather_rows(out["choices"], anc, rows)
    if sync: torch.cuda.synchronize()
    t5 = time.perf_counter(); T["gather"] += t5 - t4
for sync in (True, False):
    for k in T: T[k] = 0.0
    for i in range(20): step(i, sync)
    torch.cuda.synchronize()
    for k in T: T[k] = 0.0
    t0 = time.perf_counter()
    for i in range(100): step(i, sync)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("sync" if sync else "async", {k: round(v / 100 * 1e6, 1) for k, v in T.items()}, "total us/step", round(tot / 100 * 1e6, 1))
