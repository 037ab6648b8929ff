import os, sys, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from genjax_amd import _abi as A, kernels
import helpers as H
mode = sys.argv[1]
if mode == "dist":
    import torch.distributed as dist
prog, g = H.gmm()
K = 1 << 20
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ws = kernels.workspace(A.OP_RUN, K, dev); ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False)
rows = torch.empty_like(out["choices"]); zero = torch.zeros(1, dtype=torch.int64, device=dev)
if mode == "events":
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
def step(i):
    kernels.run_program(prog, (0, 1 + i), K, ws=ws, out=out, want_weight=False, K_total=K, offset=0)
    u = 0.3 if mode != "u" else ((i * 2654435761) % (1 << 23)) / float(1 << 23)
    cum, total = kernels.weight_cumsum(out["logw"], True, out["lse"], ws=ws2)
    anc = kernels.resample_systematic(cum, torch.cat([zero, total]), u, K)
    kernels.gather_rows(out["choices"], anc, rows)
for i in range(20): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): step(i)
torch.cuda.synchronize()
print(mode, "us/step", round((time.perf_counter() - t0) / 200 * 1e6, 1))
