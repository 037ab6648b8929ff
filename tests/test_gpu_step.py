"""One-launch importance step (gjx_importance_step) and the robustness of the kernels that synchronise their blocks
through memory: same bits as the three calls they replace, oracle-exact ancestors while another stream keeps the GPU
busy, identity ancestors + status bit for a dead collection, fallback when the grid cannot be co-resident."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def K_():
    from genjax_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu
    return cpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("D,C,K", [(16, 8, 1 << 20), (16, 8, 1 << 14), (8, 5, 3 * 1024), (2, 3, 1024), (1, 4, 2048)])
def test_importance_step_equals_three_calls(K_, D, C, K):
    import torch
    from genjax_amd import workloads
    prog, _ = workloads.gmm_program(D=D, C=C)
    assert K_.program_engine(prog) == 1
    a = K_.importance_step(prog, (0, 7), K, 0.3718, allow_fallback=False)
    assert a["fused"]
    os.environ["GJX_NO_FUSED_STEP"] = "1"
    try:
        b = K_.importance_step(prog, (0, 7), K, 0.3718)
    finally:
        del os.environ["GJX_NO_FUSED_STEP"]
    assert not b["fused"]
    torch.cuda.synchronize()
    assert K_.workspace_status(a["_ws"]) == 0
    for name in ("choices", "score", "logw", "ancestors", "rows"):
        assert torch.equal(a[name], b[name]), name
    assert float(a["lse"][0]) == float(b["lse"][0])              # the maximum (the fixed-point reference) is exact
    np.testing.assert_allclose(_np(a["lse"]), _np(b["lse"]), rtol=1e-6)   # the sum is reduced in a different order
    # and the resampled rows are the gather of the particles by the ancestors
    assert torch.equal(a["rows"], a["choices"][:, a["ancestors"].long()])
    # back-to-back calls on one workspace (epoch tags) with different keys stay independent
    c = K_.importance_step(prog, (0, 8), K, 0.11, out=dict(_ws=a["_ws"]), allow_fallback=False)
    d = K_.importance_step(prog, (0, 7), K, 0.3718, out=dict(_ws=a["_ws"]), allow_fallback=False)
    torch.cuda.synchronize()
    assert not torch.equal(c["ancestors"], a["ancestors"]) and torch.equal(d["ancestors"], a["ancestors"]) and torch.equal(d["rows"], a["rows"])


def test_importance_step_ancestors_match_oracle(K_, oracle):
    """ancestors of the one-launch step == oracle systematic resampling of the step's own log-weights, except where the
    device exp and libm exp quantise a weight one unit apart (reported, bounded)"""
    from genjax_amd import workloads
    prog, _ = workloads.gmm_program(D=16, C=8)
    K = 1 << 16
    a = K_.importance_step(prog, (0, 3), K, 0.625, allow_fallback=False)
    lw = _np(a["logw"])
    lse = oracle.logsumexp(lw, K)
    cum, _ = oracle.weight_cumsum(lw, True, lse)
    want = oracle.resample_systematic(cum, 0.625, K)
    got = _np(a["ancestors"])
    assert (got != want).mean() <= 1e-3
    assert np.abs(got.astype(np.int64) - want).max() <= 1


def test_importance_step_unsupported_sizes_fall_back(K_):
    import torch
    from genjax_amd import workloads
    from genjax_amd._lib import GjxError
    prog, _ = workloads.gmm_program(D=16, C=8)
    with pytest.raises(GjxError):
        K_.importance_step(prog, (0, 1), 1000, 0.5, allow_fallback=False)          # not a multiple of 1024
    out = K_.importance_step(prog, (0, 1), 1000, 0.5)
    assert not out["fused"] and int(out["ancestors"].max()) < 1000
    os.environ["GJX_CORESIDENT_BLOCKS"] = "8"                                      # pretend the device holds 8 blocks
    try:
        out = K_.importance_step(prog, (0, 1), 1 << 15, 0.5)
        assert not out["fused"]
        lw = torch.randn(1 << 15, device="cuda")
        anc = K_.resample_indices(lw, 0.25, lse=K_.logsumexp(lw))                  # one-launch resampler falls back too
    finally:
        del os.environ["GJX_CORESIDENT_BLOCKS"]
    anc2 = K_.resample_indices(lw, 0.25, lse=K_.logsumexp(lw))
    assert torch.equal(anc, anc2)
    # a device that holds even fewer blocks (a partition, a smaller GPU): the wrapper asks the library, gets GJX_EUNSUPPORTED
    # for the one-launch form and retries on the three-launch path with prefix-sum buffers of its own (no size constant)
    os.environ["GJX_CORESIDENT_BLOCKS"] = "4"
    try:
        anc3 = K_.resample_indices(lw, 0.25, lse=K_.logsumexp(lw))
    finally:
        del os.environ["GJX_CORESIDENT_BLOCKS"]
    assert torch.equal(anc3, anc2)


def test_one_launch_resampler_under_concurrent_load(K_, oracle):
    """the co-resident kernels while another stream keeps every CU busy: blocks that cannot start at once wait for the
    other kernel's blocks to retire; the result is still exact and no poll budget runs out"""
    import torch
    K = 1 << 20
    rs = np.random.default_rng(5)
    lw_h = (rs.standard_normal(K) * 1.5).astype(np.float32)
    lw = torch.as_tensor(lw_h).cuda()
    lse = K_.logsumexp(lw)
    ws = K_.workspace(4, K)
    ref = K_.resample_indices(lw, 0.4242, lse=lse, ws=ws)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    outs = []
    with torch.cuda.stream(side):
        for _ in range(40):
            a = (a @ a).clamp_(-1, 1)                       # ~ tens of ms of work occupying the whole chip
    for _ in range(20):
        outs.append(K_.resample_indices(lw, 0.4242, lse=lse, ws=ws))
    from genjax_amd import workloads
    prog, _ = workloads.gmm_program(D=16, C=8)
    s1 = K_.importance_step(prog, (0, 7), K, 0.3718, allow_fallback=False)
    torch.cuda.synchronize()
    assert K_.workspace_status(ws) == 0 and K_.workspace_status(s1["_ws"]) == 0
    for o in outs:
        assert torch.equal(o, ref)
    s2 = K_.importance_step(prog, (0, 7), K, 0.3718, allow_fallback=False)
    torch.cuda.synchronize()
    assert torch.equal(s1["rows"], s2["rows"]) and torch.equal(s1["ancestors"], s2["ancestors"])
    cum, _ = oracle.weight_cumsum(lw_h, True, oracle.logsumexp(lw_h, K))
    assert (_np(ref) != oracle.resample_systematic(cum, 0.4242, K)).mean() <= 1e-3


def test_dead_collection_gives_identity_ancestors_and_status(K_):
    import torch
    from genjax_amd._lib import GjxError
    K = 5000
    lw = torch.full((K,), float("-inf"), device="cuda")
    ws = K_.workspace(4, K)
    anc = K_.resample_indices(lw, 0.5, lse=K_.logsumexp(lw), ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(anc, torch.arange(K, device="cuda", dtype=torch.int32))      # in bounds for any later gather
    with pytest.raises(GjxError, match="zero"):
        K_.workspace_status(ws)
    assert K_.workspace_status(ws) == 0                                            # cleared by the read
    lw[17] = 0.0
    anc = K_.resample_indices(lw, 0.5, lse=K_.logsumexp(lw), ws=ws)
    assert int((anc != 17).sum()) == 0 and K_.workspace_status(ws) == 0


def test_filter_and_resampler_fail_soft_when_the_grid_is_not_coresident(K_):
    """A co-resident kernel whose grid does not fit (here: the launcher is told the device holds far more blocks than it
    does; in production: another stream holds compute units) runs out of its ~0.1 s poll budget and flags the workspace.
    BootstrapFilter.run and inference.pf.resample(check=True) then repeat the call on the plain multi-launch path, warn
    once, and return the SAME result as an undisturbed call — they do not raise."""
    import warnings
    import torch
    from genjax_amd import core, workloads
    from genjax_amd.inference import pf
    s = workloads.ssm_problem(T=10)
    K = 1 << 20
    bf = pf.BootstrapFilter(pf.LinearGaussianSSM(s["A"], s["q"], s["r"]), K, weights="tile_scaled")
    ref = bf.run(core.key(5), s["y"])
    rows = torch.randn(2, 1 << 22, device="cuda")
    lw = torch.randn(1 << 22, device="cuda") * 1.5
    ref_rows, ref_anc = pf.resample(rows, lw, core.key(8), check=True)
    pf._warned_timeout[0] = False
    os.environ["GJX_CORESIDENT_BLOCKS"] = "100000"
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = bf.run(core.key(5), s["y"])
            got_rows, got_anc = pf.resample(rows, lw, core.key(8), check=True)
        assert sum("timed out" in str(x.message) for x in w) == 1          # logged once
    finally:
        del os.environ["GJX_CORESIDENT_BLOCKS"]
    assert torch.equal(got["x"], ref["x"]) and torch.equal(got["logw"], ref["logw"]) and not got["degenerate"]
    np.testing.assert_allclose(_np(got["increments"]), _np(ref["increments"]), rtol=2e-6, atol=2e-6)
    assert torch.equal(got_anc, ref_anc) and torch.equal(got_rows, ref_rows)
