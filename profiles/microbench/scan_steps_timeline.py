"""per-block phase stamps (s_memrealtime, 100 MHz) of step T / 2 inside the steps kernel of the generic filter (gjx_gen_steps) on config 3's model
written as @gen + .scan, K = 2^18: min / median / max over the 256 blocks, us from the first block's step start"""
import ctypes as C_, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import genjax_amd as genjax
from genjax_amd import C, workloads
from genjax_amd._lib import load
from genjax_amd.inference import BootstrapFilter
import test_gpu_scan_filter as T
K, Tn = int(os.environ.get("KK", 1 << 18)), 256
s = workloads.ssm_problem()
scan, carry0 = T._lgssm_scan(s, Tn, float(s["q"]))
ys = np.asarray(s["y"], np.float32)
bf = BootstrapFilter(scan, K)
bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
nb = K // 1024
tl = torch.zeros((nb, 16), dtype=torch.int64, device="cuda")
load().gjx_debug_timeline(C_.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
torch.cuda.synchronize()
load().gjx_debug_timeline(None, 0)
t = tl.cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
print(f"steps kernel of the generic filter, step T/2, {nb} blocks")
for j, n in [(0, "step start"), (1, "table + derived constants in LDS"), (9, "granules gathered, E known"), (10, "shifted totals + prefix"),
             (11, "source tiles found"), (2, "ancestors known"), (3, "sites done (gathers, draws, densities)"), (4, "tile totals reduced"),
             (5, "stores drained (granule goes out)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:40s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
