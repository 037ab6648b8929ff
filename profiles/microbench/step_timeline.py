import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import kernels as K_, workloads
prog, _ = workloads.gmm_program(D=16, C=8)
K = 1 << 20
out = K_.importance_step(prog, (0, 7), K, 0.3718, allow_fallback=False)
tl = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
from genjax_amd._lib import load as _load
import ctypes as _C
_load().gjx_debug_timeline(_C.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
for i in range(3):
    K_.importance_step(prog, (0, 7 + i), K, 0.3718, out=out, allow_fallback=False)
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
names = ["start", "propagate done", "A done (global max)", "B done (scan+expand)", "C done (ancestors visible)", "D done (gather)"]
for j, n in enumerate(names):
    c = (t[:, j] - t0) * 0.01
    print(f"{n:30s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
