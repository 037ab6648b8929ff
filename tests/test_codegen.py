"""Per-program generated kernels (csrc/gjx_codegen.hip): emission and hipRTC compilation need no GPU (hipRTC
cross-compiles for gfx950), so they are checked here; execution and parity are in the -m gpu tests, where every program
outside the mixture shape runs on its generated kernel by default."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _programs():
    from genjax_amd import _abi as A
    from genjax_amd import workloads
    from genjax_amd.program import PackedProgram, Param, SiteList
    out = {}
    out["gmm"] = workloads.gmm_program(D=16, C=8)[0]
    out["gmm_jax32"] = workloads.gmm_program(D=4, C=3, rng=A.RNG_JAX32)[0]
    out["logreg"] = workloads.logreg_program(N=64, P=4)[0]
    sl = SiteList()
    sl.add("p", A.BETA, [np.float32(2.0), np.float32(2.0)])
    sl.add("v", A.FLIP, [Param.value("p", 1)])
    out["beta_bernoulli"] = PackedProgram(sl, {"v": A.MODE_OBS_TAB}, {"v": np.float32(1.0)})
    return out


def test_emitter_covers_the_workloads_and_compiles(tmp_path, monkeypatch):
    from genjax_amd import kernels
    monkeypatch.setenv("GJX_JIT_CACHE", str(tmp_path))
    for name, prog in _programs().items():
        for ppt in (1, 2):
            src = kernels.program_source(prog, ppt)
            assert 'extern "C" __global__' in src and "gjx_gen(GenArgs a)" in src, name
            assert src.count("---- site") == prog.n_sites
            kernels.program_precompile(prog, ppt)                 # hipRTC, gfx950, no GPU needed
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2 * len(_programs())
    # the structure, not the table values, keys the cache: new observations reuse the code object
    from genjax_amd import workloads
    p2 = workloads.gmm_program(D=16, C=8, seed=3)[0]
    kernels.program_precompile(p2, 2)
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2 * len(_programs())


def test_constants_are_hoisted_into_the_prologue():
    """log-softmax / running CDF of constant logits and log / reciprocal of table scales are computed once per block"""
    from genjax_amd import kernels
    src = kernels.program_source(_programs()["gmm"], 2)
    body = src[src.index("for (int64_t tix"):]
    assert "running CDF" in src and "#define TSRC(i) TAB(i)" in src and "fast_log(TSRC(" in src and "fast_rcp(TSRC(" in src
    assert "fast_log(" not in body and "fast_rcp(" not in body        # nothing transcendental per particle but the samplers


def test_uncovered_programs_report_unsupported():
    from genjax_amd import _abi as A
    from genjax_amd import kernels
    from genjax_amd._lib import GjxError
    from genjax_amd.program import PackedProgram, SiteList
    sl = SiteList()
    sl.add("w", A.DIRICHLET, [np.ones(3, np.float32)], dim=3)
    prog = PackedProgram(sl, {}, {})
    with pytest.raises(GjxError, match="coverage"):
        kernels.program_source(prog, 1)
