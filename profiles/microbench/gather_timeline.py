import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import kernels, workloads, _abi as A
K = 1 << 20
prog, _ = workloads.gmm_program()
ws = kernels.workspace(A.OP_RUN, K, "cuda"); ws2 = kernels.workspace(A.OP_RESAMPLE, K, "cuda")
out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False, want_lse=False)
n_part = kernels.run_partials_count(prog, K, 0)
rows = torch.empty_like(out["choices"]); lse = torch.empty(4, device="cuda")
def go():
    kernels.run_program(prog, (0, 2), K, ws=ws, out=out, want_weight=False, want_lse=False)
    kernels.resample_gather(out["logw"], 0.37, out["choices"], partials=(ws, n_part), lse_out=lse, out=rows, ws=ws2, allow_fallback=False)
for _ in range(20): go()
tl = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
from genjax_amd._lib import load as _load
import ctypes as _C
_load().gjx_debug_timeline(_C.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
go(); torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64); t0 = t[:, 0].min()
for j, n in [(0, "start"), (1, "tile total published"), (2, "totals gathered"), (6, "prefix of totals"), (7, "source tiles found"), (3, "tile list known"), (4, "ancestors known"), (5, "rows copied (end)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:24s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
