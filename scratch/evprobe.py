"""How long does the slow-launch period after the first timing event last?  Single-GPU step, 100-step reps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genjax_amd import kernels, workloads, _abi as A
dev = torch.device("cuda", 0)
K = 1 << 20
prog, g = workloads.gmm_program(D=16, C=8)
ws = kernels.workspace(A.OP_RUN, K, dev); ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
out = kernels.run_program(prog, (0, 1), K, K_total=K, ws=ws, want_weight=False)
rows = torch.empty_like(out["choices"]); anc = torch.empty(K, dtype=torch.int32, device=dev); rec = torch.empty(4, device=dev)
npart = kernels.run_partials_count(prog, K, 0)
def step(i):
    kernels.run_program(prog, (0, 1 + i), K, K_total=K, ws=ws, out=out, want_weight=False, want_lse=False)
    kernels.resample_indices(out["logw"], 0.3, K, partials=(ws, npart), lse_out=rec, K_total=K, anc=anc, ws=ws2)
    kernels.gather_rows(out["choices"], anc, rows)
def rep(n, tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize(); print("%-28s %.1f us/step" % (tag, (time.perf_counter() - t0) / n * 1e6), flush=True)
mode = sys.argv[1]
for r in range(3): rep(100, "before any timing event")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); step(0); e1.record()
if mode == "elapsed":
    torch.cuda.synchronize(); print("elapsed", e0.elapsed_time(e1))
if mode == "del":
    torch.cuda.synchronize(); del e0, e1
for r in range(10): rep(100, "after first timing event")
