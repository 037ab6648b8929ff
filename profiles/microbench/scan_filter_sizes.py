import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import genjax_amd as genjax
from genjax_amd import C, workloads
from genjax_amd.inference import BootstrapFilter
import test_gpu_scan_filter as T
K, Tn = int(os.environ.get("KK", 1 << 20)), 256
s = workloads.ssm_problem()
scan, carry0 = T._lgssm_scan(s, Tn, float(s["q"]))
ys = np.asarray(s["y"], np.float32)
bf = BootstrapFilter(scan, K)
import warnings
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        o = bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
    torch.cuda.synchronize()
    print("K", K, "us per step", (time.perf_counter() - t0) / (n * Tn) * 1e6, "log_ml", float(o["log_ml"]), "warnings", len(w))
