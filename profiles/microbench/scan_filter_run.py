"""the generic Scan filter on config 3's model, a few runs (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import genjax_amd as genjax
from genjax_amd import C, workloads
from genjax_amd.inference import BootstrapFilter
import test_gpu_scan_filter as T
K, Tn = 1 << 18, 256
s = workloads.ssm_problem()
scan, carry0 = T._lgssm_scan(s, Tn, float(s["q"]))
ys = np.asarray(s["y"], np.float32)
bf = BootstrapFilter(scan, K)
bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    o = bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
torch.cuda.synchronize()
print("us per step", (time.perf_counter() - t0) / (n * Tn) * 1e6, "log_ml", float(o["log_ml"]))
# host-side cost of the loop alone: the same run with the stream left to drain afterwards
t0 = time.perf_counter()
o = bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue us per step", (t1 - t0) / Tn * 1e6, "drain us per step", (t2 - t1) / Tn * 1e6)
