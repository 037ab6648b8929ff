"""per-block phase stamps (s_memrealtime, 100 MHz) of step T / 2 inside the filter kernel GENERATED for config 3's model written as
@gen + .scan (gjx_gen_pf on the shared skeleton, csrc/gjx_pfcore.h): min / median / max over the blocks, us from the first block's
step start.  KK = particles (default 2^18)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import genjax_amd as genjax
from genjax_amd import C, workloads
from genjax_amd.inference import BootstrapFilter
K, Tn = int(os.environ.get("KK", 1 << 18)), 256
scan, carry0, s = workloads.lgssm_scan(8, Tn)
ys = np.asarray(s["y"], np.float32)
bf = BootstrapFilter(scan, K)
o = bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
nb = o["info"]["grid"]
tl = torch.zeros((nb, 16), dtype=torch.int64, device="cuda")
bf.timeline = tl
o = bf.run(genjax.key(1), C["y"].set(ys), (carry0, None))
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
print(f"generated filter kernel ({o['info']['form_name']}), step T/2, {nb} blocks x {o['info']['tiles_per_block']} tiles, K = {K}")
for j, n in [(0, "step start"), (1, "granules {e_b, S_b} published"), (2, "table staged, draws of the step done"), (3, "granules gathered, E known"),
             (4, "shifted totals + prefix"), (5, "tiles found, peers ready"), (6, "ancestors known"), (7, "slots done (gathers, sites, stores issued)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:44s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
