"""The matrix-core flavour of generated ImportanceK kernels under its profiling variants (GJX_GEN_MFMA_DEBUG, gjx_codegen.hip: 1 = no matrix
instructions, 2 = no elementwise phase, 4 = no scheduling barrier, 8 = dependent order, 16 / 32 = one / two row tiles per trip): run from the repo root."""
import sys, os, torch
sys.path.insert(0, ".")
from genjax_amd import kernels, workloads, _abi as A
K = 1 << 20
for dbg in (sys.argv[1:] or ["0", "1", "2", "16", "32"]):
    os.environ["GJX_GEN_MFMA_DEBUG"] = dbg
    os.environ["GJX_JIT_NO_DISK"] = "1"
    prog, _ = workloads.logreg_importance_program()
    ws = kernels.workspace(A.OP_RUN, K, "cuda")
    out = kernels.run_program(prog, (0, 1), K, ws=ws, want_weight=False)
    for i in range(100):
        kernels.run_program(prog, (0, 2 + i), K, ws=ws, out=out, want_weight=False)
    tm = [kernels.DispatchTimer() for _ in range(5)]
    for i, t in enumerate(tm):
        kernels.run_program(prog, (0, 2 + i), K, ws=ws, out=out, want_weight=False, timer=t)
    torch.cuda.synchronize()
    print("debug", dbg, sorted(t.elapsed_us() for t in tm)[2], "us")
