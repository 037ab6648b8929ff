"""Host-time breakdown of the sharded step with one rank (GJX_FORCE_DIST=1): where do the microseconds go?"""
import os, sys, time, collections
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
T = collections.defaultdict(float)

def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t; return r
    setattr(mod, name, g)

for n in ("_all_gather", "_all_to_all"):
    wrap(DD, n)
for n in ("weight_cumsum", "shard_resample", "pack_rows", "unpack_rows"):
    wrap(kernels, n)
wrap(kernels.ShardPlan, "build"); wrap(kernels.ShardPlan, "wait")
N = 300
MODE = os.environ.get("EVMODE", "none")
if MODE in ("live", "dead"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize(); print("elapsed", e0.elapsed_time(e1))
    if MODE == "dead":
        del e0, e1
for rep in range(2):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        if MODE == "each" and i % 20 == 0:
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ea.record()
        t = time.perf_counter()
        kernels.run_program(prog, (0, 1 + i), K, K_total=K, ws=ws, out=out, want_weight=False, want_lse=True)
        T["run_program"] += time.perf_counter() - t
        if MODE == "each" and i % 20 == 0:
            eb.record()
        t = time.perf_counter()
        pairs = DD.gather_lse_pairs(out["lse"])
        T["gather_lse_pairs"] += time.perf_counter() - t
        t = time.perf_counter()
        DD.resample_exchange(out["choices"], out["logw"], None, 0.3, K, pairs=pairs)
        T["resample_exchange"] += time.perf_counter() - t
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("per step us: host %.1f  total %.1f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-20s %7.1f us" % (k, v / N * 1e6))
dist.destroy_process_group()
