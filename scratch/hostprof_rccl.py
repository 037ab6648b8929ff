"""Per-step time of the one-call RCCL exchange with one rank (GJX_FORCE_DIST=1)."""
import os, sys, time
os.environ["GJX_FORCE_DIST"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from genjax_amd import distributed as DD, kernels, workloads, _abi as A

DD.init_from_env()
dev = torch.device("cuda", 0)
K = 1 << 20
prog, g = workloads.gmm_program(D=16, C=8)
ws = kernels.workspace(A.OP_RUN, K, dev)
out = kernels.run_program(prog, (0, 1), K, K_total=K, ws=ws, want_weight=False)
res = DD.ShardedResampler(K, out["choices"].shape[0], K, dev, transport=os.environ.get("TRANSPORT", "rccl"))
print("transport", res.transport)
N = 300
MODE = os.environ.get("EVMODE", "none")
evs = []
for rep in range(3):
    t_run = t_res = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        if MODE == "ev" and i % 20 == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
        t = time.perf_counter()
        kernels.run_program(prog, (0, 1 + i), K, K_total=K, ws=ws, out=out, want_weight=False, want_lse=True)
        t_run += time.perf_counter() - t
        if MODE == "ev" and i % 20 == 0:
            b.record(); evs.append((a, b))
        t = time.perf_counter()
        res.step(out["choices"], out["logw"], out["lse"], 0.3)
        t_res += time.perf_counter() - t
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("per step us: host %.1f total %.1f  (run_program %.1f, resample step %.1f)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, t_run / N * 1e6, t_res / N * 1e6))
res.close()
dist.destroy_process_group()
