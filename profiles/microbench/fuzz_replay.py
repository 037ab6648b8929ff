"""Replay ONE trial of the random-program campaign (profiles/fuzz.sh; tests/test_gpu_parity.py::test_random_programs_against_oracle)
and print where device and oracle differ:  python profiles/microbench/fuzz_replay.py SEED RNG TRIAL"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                     # noqa: E402
import test_gpu_parity as P                      # noqa: E402
from genjax_amd import _abi as A, kernels as K_  # noqa: E402
from genjax_amd.program import PackedProgram     # noqa: E402
from oracle import cpu as oracle                 # noqa: E402

seed, rng, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K_PART = int(os.environ.get("FUZZ_K", "600"))
rs = np.random.default_rng(seed + rng)
K = K_PART
for trial in range(want + 1):
    sl = P._random_program(rs, rng)
    key = (int(rs.integers(1 << 30)), int(rs.integers(1 << 30)))
    subdraw = [rs.random() for _ in sl.sites]
    if trial < want:
        continue
    os.environ["GJX_ENGINE"] = "interp" if trial & 1 else ""
    if not (trial & 1):
        os.environ.pop("GJX_ENGINE")
    prog = PackedProgram(sl, rng_mode=rng)
    print("engine", K_.program_engine(prog), "key", key)
    for s in sl.sites:
        print(" site", s.addr, A.KIND_NAMES[s.kind], "dim", s.dim, [(p.op, p.src, p.src_elem, p.length, p.xf, None if p.values is None else np.asarray(p.values).ravel()[:6]) for p in s.params])
    g = K_.run_program(prog, key, K, want_site_scores=True)
    o = oracle.run_program(prog, key, K, want_site_scores=True, want_margin=True)
    gc, gs = g["choices"].cpu().numpy(), g["score"].cpu().numpy()
    ok = P._close_cols(gc, o["choices"], rt=1e-3, at=5e-4) & P._close_cols(gs[None], o["score"][None], rt=2e-3, at=2e-3)
    bad = np.nonzero(~ok)[0]
    print("differing particles", bad)
    np.set_printoptions(precision=8, linewidth=200)
    for i in bad[:8]:
        print("particle", i, "margin", o["margin"][i])
        print("  dev choices", gc[:, i])
        print("  ora choices", o["choices"][:, i])
        print("  dev score", gs[i], "ora score", o["score"][i])
        print("  dev site scores", g["site_scores"].cpu().numpy()[:, i])
        print("  ora site scores", o["site_scores"][:, i])

    # the second half of the test: a random subset of sites constrained to the oracle's own draws, site scores compared
    sub = [s_.addr for s_, u in zip(sl.sites, subdraw) if u < 0.5]
    if sub:
        prog2 = PackedProgram(sl, {a: A.MODE_OBS_SLOT for a in sub}, rng_mode=rng)
        ch = o["choices"].copy()
        g2 = K_.run_program(prog2, key, K, choices=torch.as_tensor(ch).cuda(), want_site_scores=True)
        o2 = oracle.run_program(prog2, key, K, choices=ch.copy(), want_site_scores=True)
        idx = [j for j, s_ in enumerate(sl.sites) if s_.addr in sub]
        gs, os_ = g2["site_scores"].cpu().numpy()[idx], o2["site_scores"][idx]
        with np.errstate(invalid="ignore"):
            off = ~(np.abs(gs - os_) <= 2e-3 + 2e-3 * np.abs(os_)) & np.isfinite(os_) & (np.abs(os_) < 1e4)
        for r, i in zip(*np.nonzero(off)):
            s_ = sl.sites[idx[r]]
            print("constrained site", s_.addr, A.KIND_NAMES[s_.kind], "particle", i, "dev", gs[r, i], "ora", os_[r, i], "value", ch[prog2.slot_of[s_.addr], i])
            print("   all choices of the particle", ch[:, i])
