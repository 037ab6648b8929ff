"""the same stamps as pf_gen_timeline.py (the skeleton's, csrc/gjx_pfcore.h) for the HAND-WRITTEN instance of the skeleton, k_pf_persistent
(LgssmModel, config 3): min / median / max over the blocks, us from the first block's step start.  KK = particles (default 2^18)."""
import ctypes as C_, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import genjax_amd as genjax
from genjax_amd import workloads
from genjax_amd._lib import load
from genjax_amd.inference import BootstrapFilter, LinearGaussianSSM
K = int(os.environ.get("KK", 1 << 18))
s = workloads.ssm_problem()
ys = torch.as_tensor(s["y"], device="cuda")
bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"], q0=float(s["q"])), K, weights="tile_scaled")
for i in range(3):
    bf.run(genjax.key(i), ys, device="cuda")
nb = 4096
tl = torch.zeros((nb, 16), dtype=torch.int64, device="cuda")
load().gjx_debug_timeline(C_.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
bf.run(genjax.key(9), ys, device="cuda")
torch.cuda.synchronize()
load().gjx_debug_timeline(None, 0)
t = tl.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print(f"hand-written filter kernel on the shared skeleton, step T/2, {len(t)} blocks, K = {K}")
for j, n in [(0, "step start"), (1, "granules {e_b, S_b} published"), (2, "y_t staged, hash words of the step done"), (3, "granules gathered, E known"),
             (4, "shifted totals + prefix"), (5, "tiles found, peers ready"), (6, "ancestors known"), (7, "slots done (gathers, propagate, stores issued)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:44s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
