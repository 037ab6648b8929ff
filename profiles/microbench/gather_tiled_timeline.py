import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import kernels, workloads, _abi as A
K = 1 << 20
prog, _ = workloads.gmm_program()
ws = kernels.workspace(A.OP_RUN, K, "cuda"); ws2 = kernels.workspace(A.OP_RESAMPLE, K, "cuda")
out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False, want_lse=False, want_tiles=True)
part = out["_partials"]; print("tiles offset", part.tiles, "n partials", part.count())
rows = torch.empty_like(out["choices"]); lse = torch.empty(4, device="cuda")
def step_old():
    kernels.run_program(prog, (0, 2), K, ws=ws, out=out, want_weight=False, want_lse=False)
    kernels.resample_gather(out["logw"], 0.37, out["choices"], partials=(ws, part.count()), lse_out=lse, out=rows, ws=ws2, allow_fallback=False)
def step_new():
    kernels.run_program(prog, (0, 2), K, ws=ws, out=out, want_weight=False, want_lse=False, want_tiles=True)
    kernels.resample_gather_tiled(out["logw"], 0.37, out["choices"], partials=(ws, part.count()), tiles=part.tiles, lse_out=lse, out=rows, ws=ws2)
def step_new_notiles():
    kernels.run_program(prog, (0, 2), K, ws=ws, out=out, want_weight=False, want_lse=False)
    kernels.resample_gather_tiled(out["logw"], 0.37, out["choices"], partials=(ws, part.count()), tiles=0, lse_out=lse, out=rows, ws=ws2)
for name, fn in (("old", step_old), ("tiled+tiles", step_new), ("tiled no tiles", step_new_notiles), ("old", step_old), ("tiled+tiles", step_new)):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): fn()
    torch.cuda.synchronize(); print(name, "us/step %.2f" % ((time.perf_counter() - t0) / 400 * 1e6), "lse", lse.tolist()[2:])
tl = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
from genjax_amd._lib import load as _load
import ctypes as _C
_load().gjx_debug_timeline(_C.c_void_p(tl.data_ptr()), tl.numel() * tl.element_size())
step_new(); torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64); t0 = t[:, 0].min()
for j, n in [(0, "start"), (1, "totals read, Emax"), (2, "prefix"), (3, "tiles found + window cum"), (4, "ancestors known"), (5, "rows copied (end)")]:
    c = (t[:, j] - t0) * 0.01
    print(f"{n:26s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
