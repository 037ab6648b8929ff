"""replay one trial of the differential campaign (tests/test_gpu_parity.py::test_random_programs_against_oracle) and print what differs"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from genjax_amd import _abi as A, kernels
from genjax_amd.program import PackedProgram
from oracle import cpu as oracle
import test_gpu_parity as TP

def replay(seed, rng, target):
    rs = np.random.default_rng(seed + rng)
    K = 600
    for trial in range(target + 1):
        sl = TP._random_program(rs, rng)
        key = (int(rs.integers(1 << 30)), int(rs.integers(1 << 30)))
        if trial == target:
            return sl, key, trial
        [rs.random() for _ in sl.sites]

def main():
    seed, target = int(sys.argv[1]), int(sys.argv[2])
    for rng in (A.RNG_FLAT, A.RNG_JAX32):
        sl, key, trial = replay(seed, rng, target)
        if trial & 1:
            os.environ["GJX_ENGINE"] = "interp"
        else:
            os.environ.pop("GJX_ENGINE", None)
        prog = PackedProgram(sl, rng_mode=rng)
        K = 600
        g, o = TP._run_both(kernels, oracle, prog, key, K, want_site_scores=True)
        ok = TP._close_cols(g["choices"], o["choices"], rt=1e-3, at=5e-4) & TP._close_cols(g["score"][None], o["score"][None], rt=2e-3, at=2e-3)
        fin = np.isfinite(o["score"]) & (np.abs(o["score"]) < 1e4)
        miss = ~ok & fin
        print(f"== seed {seed} rng {rng} trial {trial} engine {g.get('_engine')} kinds {[A.KIND_NAMES[s.kind] for s in sl.sites]}: {int(miss.sum())} differing, margins {o['margin'][miss][:10]}")
        if not miss.any():
            continue
        for s in sl.sites:
            print("  site", s.addr, A.KIND_NAMES[s.kind], "dim", s.dim, "ncat", s.ncat, "slot", prog.slot_of[s.addr],
                  [(p.op, p.xf, getattr(p, 'src', None), np.asarray(p.values).ravel()[:6] if getattr(p, 'values', None) is not None else None) for p in s.params])
        for i in np.flatnonzero(miss)[:4]:
            print(f"  particle {i}: score dev {g['score'][i]} ora {o['score'][i]} margin {o['margin'][i]}")
            for r in range(prog.n_slots):
                d, e = g["choices"][r, i], o["choices"][r, i]
                flag = "" if abs(d - e) <= 5e-4 + 1e-3 * abs(e) else "   <-- differs"
                print(f"     row {r}: dev {d!r} ora {e!r}{flag}")
            ss_d, ss_o = g["site_scores"][:, i], o["site_scores"][:, i]
            print("     site scores dev", ss_d, "\n     site scores ora", ss_o)

main()
