"""Locate the one-off ~36 ms stall: per-step host enqueue time over many steps (no sync inside)."""
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
torch.cuda.synchronize()
T0 = time.perf_counter()
ts = []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for i in range(N):
    t = time.perf_counter()
    kernels.run_program(prog, (0, 1 + i), K, K_total=K, ws=ws, out=out, want_weight=False, want_lse=False)
    t1 = time.perf_counter()
    kernels.resample_indices(out["logw"], 0.3, K, partials=(ws, npart), lse_out=rec, K_total=K, anc=anc, ws=ws2)
    t2 = time.perf_counter()
    kernels.gather_rows(out["choices"], anc, rows)
    t3 = time.perf_counter()
    ts.append((t - T0, t1 - t, t2 - t1, t3 - t2))
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print("total %.1f ms for %d steps = %.1f us/step" % (tot * 1e3, N, tot / N * 1e6))
for i, (a, b, c, d) in enumerate(ts):
    if max(b, c, d) > 2e-3:
        print("step %d at %.1f ms: run %.2f ms, resample %.2f ms, gather %.2f ms" % (i, a * 1e3, b * 1e3, c * 1e3, d * 1e3))
