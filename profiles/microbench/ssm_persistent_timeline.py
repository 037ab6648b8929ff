import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import core, workloads
from genjax_amd.inference.pf import BootstrapFilter, LinearGaussianSSM
s = workloads.ssm_problem()
W = os.environ.get("SSM_WEIGHTS", "global_max")     # or tile_scaled
bf = BootstrapFilter(LinearGaussianSSM(s["A"], s["q"], s["r"]), int(os.environ.get("SSM_K", 1 << 18)), weights=W)
ys = torch.as_tensor(s["y"]).cuda()
bf.run(core.key(1), ys)
tl = torch.zeros((4096, 16), dtype=torch.int64, device="cuda")
from genjax_amd._lib import load as _load
import ctypes as _C
_load().gjx_debug_timeline(_C.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
bf.run(core.key(2), ys)
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("persistent filter (%s weights), step T/2, %d blocks" % (W, len(t)))
names = ([(0, "step start"), (1, "block max published"), (2, "global max known"), (3, "tile total published"), (4, "totals gathered + prefix"),
          (5, "ancestors known"), (6, "step end (stores issued)")] if W == "global_max" else
         [(0, "step start"), (7, "all waves of the block in"), (1, "{e_b, S_b} published"), (2, "draws of the step done"), (3, "granules gathered, E known"), (4, "shifted totals + prefix"),
          (9, "tile found, peers ready"), (10, "source tiles' log w loaded"), (11, "source tiles scanned"), (5, "ancestors known"),
          (8, "x[ancestor] loaded"), (6, "step end (stores issued)")])
for j, n in names:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:26s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
if os.environ.get("SSM_TL_DETAIL"):
    ids = np.nonzero(tl.cpu().numpy()[:, 0] > 0)[0]
    c = (t - t0) * 0.01
    print("phase durations per block (us): min / median / p95 / max, and the 4 slowest block ids")
    nm = dict(names)
    for a, b in ([(0, 7), (7, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6)] if W == "global_max" else
                 [(0, 7), (7, 1), (1, 2), (2, 3), (3, 4), (4, 9), (9, 10), (10, 11), (11, 5), (5, 8), (8, 6)]):
        d = c[:, b] - c[:, a]
        o = np.argsort(-d)[:4]
        print(f"  {nm.get(a, '?'):>28s} -> {nm.get(b, '?'):<28s} {d.min():6.2f} {np.median(d):6.2f} {np.percentile(d, 95):6.2f} {d.max():6.2f}   " +
              " ".join(f"{ids[i]}({d[i]:.2f})" for i in o))
    for jn in (0, 3, 6):
        o = np.argsort(-c[:, jn])[:6]
        print(f"  latest at '{nm[jn]}': " + " ".join(f"{ids[i]}({c[i, jn]:.2f})" for i in o))
