"""CPU tests of the host logic: data types, model tracing, program packing, the C-ABI library's exports.
No compute call is made here (there is no GPU in the build container and no CPU fallback in the product)."""
import ctypes
import os
import re

import numpy as np
import pytest

import genjax_amd as genjax
from genjax_amd import C, S, ChoiceMap, Selection
from genjax_amd import _abi as A
from genjax_amd.inference import Target
from genjax_amd.program import PackedProgram, Param, SiteList

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/gjx.h <-> libgjx_hip.so <-> genjax_amd/_abi.py agree on the symbol set."""
    hdr = open(os.path.join(ROOT, "include", "gjx.h")).read()
    declared = set(re.findall(r"\b(gjx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gjx_status"}
    assert declared == set(A.PROTOTYPES), (declared ^ set(A.PROTOTYPES))
    from genjax_amd import _lib
    lib = _lib.load()                                   # binds all of them; AttributeError if one is missing
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gjx_version() == A.ABI_VERSION
    assert lib.gjx_workspace_bytes(A.OP_RUN, 1 << 20) >= 8 * (1 << 20) // 256 + 256
    assert ctypes.sizeof(A.GjxSite) == 240 and ctypes.sizeof(A.GjxParam) == 48


def test_abi_struct_layout_matches_header():
    hdr = open(os.path.join(ROOT, "include", "gjx.h")).read()
    for name, val in (("GJX_NORMAL", A.NORMAL), ("GJX_GAMMA", A.GAMMA), ("GJX_P_AFFINE", A.P_AFFINE), ("GJX_XF_SIGMOID", A.XF_SIGMOID),
                      ("GJX_MODE_OBS_SLOT", A.MODE_OBS_SLOT), ("GJX_RNG_JAX32", A.RNG_JAX32), ("GJX_OP_SSM", A.OP_SSM)):
        m = re.search(name + r"\s*=\s*(\d+)", hdr)
        assert m and int(m.group(1)) == val, name
    assert f"#define GJX_FLAT_SITE_SHIFT {A.FLAT_SITE_SHIFT}" in hdr


def test_host_threefry_native_equals_python():
    """core.threefry2x32 goes through the library's host entry point; the Python restatement stays the checked twin."""
    from genjax_amd import core
    rs = np.random.default_rng(0)
    for k0, k1, c0, c1 in rs.integers(0, 1 << 32, (200, 4), dtype=np.uint64).tolist() + [[0, 0, 0, 0], [2**32 - 1] * 4]:
        assert core.threefry2x32(k0, k1, c0, c1) == core.threefry2x32_py(k0, k1, c0, c1)
    assert core.threefry2x32_py(0, 0, 0, 0) == (0x6B200159, 0x99BA4EFE)          # Random123 KAT
    assert core._native, "the library's gjx_host_threefry2x32 should be in use once the library loads"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from genjax_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgjx_hip.so"))
    with pytest.raises(_lib.GjxError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "genjax_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "gjx_oracle" not in src, f


def test_choice_map_and_selection_algebra():
    chm = C["y"].set(3.0)
    assert "y" in chm and chm["y"] == 3.0 and "x" not in chm
    assert C.kw(y=3.0) == chm and C.d({"y": 3.0}) == chm
    assert chm.get_submap("y").get_value() == 3.0 and chm.get_submap("zz").static_is_empty()
    both = chm | C["x"].set(1.0)
    assert set(both.addresses()) == {"x", "y"}
    assert (C["y"].set(1.0) | C["y"].set(2.0))["y"] == 1.0            # left-biased merge (choice_map.py:1227)
    assert both.at["z"].set(5.0)["z"] == 5.0 and "z" not in both
    assert C.v(2.5).get_value() == 2.5 and C.n().static_is_empty()
    with pytest.raises(KeyError):
        chm["nope"]
    sel = S["x"] | S["y"]
    assert "x" in sel and "z" not in sel and "z" in ~sel and "x" not in ~sel
    assert Selection.all().check("anything") and not Selection.none().check("anything")
    assert both.filter(S["x"]).addresses() == ["x"] and both.filter(~S["x"]).addresses() == ["y"]
    assert ((~S["x"]) | S["x"]).check("x") and ((~S["x"]) & (~S["y"])).check("z") and not ((~S["x"]) & (~S["y"])).check("y")
    assert Selection.at["x"].check("x")
    assert set(both.get_selection().addrs) == {"x", "y"}


def test_keys():
    k = genjax.key(314159)
    assert k == (0, 314159)
    a, b = genjax.split(k)
    assert a == genjax.fold_in(k, 0) and b == genjax.fold_in(k, 1) and a != b
    assert len(genjax.split(k, 50)) == 50


def test_tracing_builds_the_expected_program():
    @genjax.gen
    def model(scale):
        z = genjax.categorical(probs=[0.2, 0.8]) @ "z"
        mu = genjax.const([[0.0, 1.0], [5.0, 6.0]])[z]
        x = genjax.mv_normal_diag(mu, np.array([1.0, 2.0])) @ "x"
        t = genjax.normal(0.0, 1.0) @ "t"
        y = genjax.normal(np.array([[1.0, -1.0]]) @ x + 0.5, genjax.exp(t) ) @ "y"
        b = genjax.flip(genjax.where(z, 0.9, 0.3)) @ "b"
        w = genjax.normal(2.0 * x[1] - 1.0, scale) @ "w"
        return y

    sl, ret = model.site_list((3.0,))
    assert sl.addresses() == ["z", "x", "t", "y", "b", "w"]
    assert [s.dim for s in sl.sites] == [1, 2, 1, 1, 1, 1] and sl["z"].ncat == 2
    prog, shared, pp = model.pack((3.0,), C["y"].set(0.25), True)
    assert prog.slot_of == {"z": 0, "x": 1, "t": 3, "y": -1, "b": 4, "w": 5} and prog.n_slots == 6
    s = {a: prog.c_sites[j] for j, a in enumerate(sl.addresses())}
    assert s["x"].p[0].op == A.P_GATHER and s["x"].p[0].n == 2 and s["x"].p[0].len == 2 and s["x"].p[0].slot == 0
    assert s["y"].mode == A.MODE_OBS_TAB and prog.tab[s["y"].obs_off] == 0.25
    assert s["y"].p[0].op == A.P_AFFINE and s["y"].p[0].n == 2 and s["y"].p[0].slot == 1
    np.testing.assert_array_equal(prog.tab[s["y"].p[0].moff: s["y"].p[0].moff + 2], [1.0, -1.0])
    assert prog.tab[s["y"].p[0].off] == 0.5
    assert s["y"].p[1].op == A.P_VALUE and s["y"].p[1].xf == A.XF_EXP and s["y"].p[1].slot == 3
    assert s["b"].p[0].op == A.P_GATHER and list(prog.tab[s["b"].p[0].off: s["b"].p[0].off + 2]) == pytest.approx([0.3, 0.9])
    assert s["w"].p[0].op == A.P_AFFINE and s["w"].p[0].slot == 2 and s["w"].p[0].n == 1       # trimmed to x[1]
    assert prog.tab[s["w"].p[0].moff] == 2.0 and prog.tab[s["w"].p[0].off] == -1.0 and prog.tab[s["w"].p[1].off] == 3.0
    assert model.site_list((3.0,))[0] is sl and model.site_list((4.0,))[0] is not sl            # cached per args


def test_observed_parents_fold_into_constants():
    sl = SiteList()
    sl.add("x", A.MVNORMAL_DIAG, [np.zeros(2, np.float32), np.ones(2, np.float32)], dim=2)
    sl.add("k", A.CATEGORICAL_PROBS, [np.array([0.5, 0.5], np.float32)])
    sl.add("y", A.NORMAL, [Param.affine(np.array([[2.0, 3.0]], np.float32), "x", bias=1.0), Param.gather([0.1, 0.2], "k")])
    prog = PackedProgram(sl, {"x": A.MODE_OBS_TAB, "k": A.MODE_OBS_TAB}, {"x": [1.0, -1.0], "k": 1.0})
    y = prog.c_sites[2]
    assert prog.n_slots == 1 and y.slot == 0
    assert y.p[0].op == A.P_CONST and prog.tab[y.p[0].off] == 0.0            # 2*1 + 3*(-1) + 1
    assert y.p[1].op == A.P_CONST and prog.tab[y.p[1].off] == pytest.approx(0.2)
    prog.set_obs("x", [2.0, 2.0])
    assert prog.tab[y.p[0].off] == 11.0


def test_errors_mirror_the_reference():
    @genjax.gen
    def dup():
        _ = genjax.normal(0.0, 1.0) @ "a"
        _ = genjax.normal(0.0, 1.0) @ "a"

    with pytest.raises(genjax.AddressReuse):                 # static.py:139
        dup.site_list(())

    @genjax.gen
    def model():
        x = genjax.normal(0.0, 1.0) @ "x"
        _ = genjax.normal(x, 1.0) @ "y"

    with pytest.raises(genjax.MissingAddress):               # static.py:316-318
        model.pack((), C["x"].set(0.0), False)
    with pytest.raises(TypeError):                           # sp.py:46-49, tests/inference/test_smc.py:89-106
        Target(model.marginal(selection=S["x"]), (), C["x"].set(1.0))
    t = Target(model, (), C["y"].set(3.0))
    assert t["y"] == 3.0 and t.constraint.get_submap("y").get_value() == 3.0
    assert t.filter_to_unconstrained(C.d({"x": 1.0, "y": 3.0})).addresses() == ["x"]

    @genjax.gen
    def nonlinear():
        a = genjax.normal(0.0, 1.0) @ "a"
        b = genjax.normal(0.0, 1.0) @ "b"
        _ = genjax.normal(a * b, 1.0) @ "c"

    # (rounds 1-5 refused a product of choices; since ABI 9 it is a general expression block: tests/test_expr_cpu.py)
    sl_, _ = nonlinear.site_list(())
    assert sl_["c"].params[0].op == A.P_EXPR

    @genjax.gen
    def gather_arith():
        z = genjax.categorical(np.zeros(3, np.float32)) @ "z"
        a = genjax.normal(0.0, 1.0) @ "a"
        _ = genjax.normal(genjax.take(np.arange(3.0), z) * a, 1.0) @ "c"     # arithmetic on a table gather: still not expressible

    with pytest.raises(TypeError):
        gather_arith.site_list(())
    with pytest.warns(DeprecationWarning):                   # distribution.py:479-500
        genjax.categorical([0.0, 1.0])
    with pytest.raises(RuntimeError):
        genjax.normal(0.0, 1.0) @ "outside"


def test_defaults_mirror_the_reference():
    from genjax_amd.inference import HMC, ImportanceK

    @genjax.gen
    def m():
        _ = genjax.normal(0.0, 1.0) @ "x"

    t = Target(m, (), C.n())
    assert ImportanceK(t).get_num_particles() == 2            # smc.py:290
    assert HMC(S["x"], 0.1).L == 10                           # hmc.py:154
    assert ImportanceK(t, k_particles=7).get_final_target() is t


def test_shard_arithmetic():
    from genjax_amd.distributed import shard
    for K in (1, 7, 1 << 20, (1 << 20) + 3):
        for G in (1, 2, 3, 8):
            parts = [shard(K, r, G) for r in range(G)]
            assert parts[0][0] == 0 and sum(k for _, k in parts) == K
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(G - 1))


def test_scan_unrolls_into_chained_sites():
    @genjax.gen
    def kernel(z, scanned_in):
        z = genjax.normal(z, 1.0) @ "x"
        _ = genjax.normal(2.0 * z + scanned_in, 0.01) @ "y"
        return z, None

    model = kernel.scan(n=4)
    sl, (carry, outs) = model.site_list((0.5, np.arange(4, dtype=np.float32)))
    assert sl.addresses() == [("x", 0), ("y", 0), ("x", 1), ("y", 1), ("x", 2), ("y", 2), ("x", 3), ("y", 3)]
    prog, shared, _ = model.pack((0.5, np.arange(4, dtype=np.float32)), C["y"].set(np.full(4, 3.0, np.float32)), True)
    assert prog.n_slots == 4 and all(prog.slot_of[("y", t)] == -1 for t in range(4))
    x0, x1, y1 = prog.c_sites[0], prog.c_sites[2], prog.c_sites[3]
    assert x0.p[0].op == A.P_CONST and prog.tab[x0.p[0].off] == 0.5                    # initial carry
    assert x1.p[0].op == A.P_VALUE and x1.p[0].slot == prog.slot_of[("x", 0)]            # x_1 ~ N(x_0, 1)
    assert y1.p[0].op == A.P_AFFINE and prog.tab[y1.p[0].moff] == 2.0 and prog.tab[y1.p[0].off] == 1.0   # 2 x_1 + xs[1]
    assert Selection.at["x"].check(("x", 2)) and not Selection.at["x"].check(("y", 2))
    chm = C.d({("x", 1): 7.0, "x": np.array([5.0, 7.0])})
    assert chm[1, "x"] == 7.0 and chm[:, "x"][0] == 5.0


def test_hierarchical_addresses_and_nested_calls():
    """`callee(...) @ "addr"`, `kernel.scan(n)(...) @ "tracks"`, `kernel.vmap(...)(...) @ "ys"` inline the callee's
    sites under a path prefix; ChoiceMap / Selection accept the reference's address forms
    (tests/inference/test_requests.py:257-420: ["tracks", :, "obs_pos"], ["x", "x"], Selection.at["tracks", ..., "pos"])."""
    import genjax_amd as genjax
    from genjax_amd import C, ChoiceMap, S, Selection
    from genjax_amd.core import key_of, norm_addr, ALL

    assert norm_addr(("tracks", 3, "pos")) == (("tracks", "pos"), 3)
    assert norm_addr(("tracks", slice(None), "pos")) == (("tracks", "pos"), ALL)
    assert norm_addr((2, "x")) == ("x", 2) and key_of(("x", "x")) == ("x", "x") and key_of("x") == "x"
    # nested combinators: the indices stack, outermost first (a vmap inside a scan step, a scan inside a vmap instance)
    assert norm_addr(("a", 1, 2, "b")) == (("a", "b"), (1, 2)) and key_of((1, 2, "x")) == ("x", (1, 2))
    assert key_of((slice(None), 2, "x")) == "x"                       # a wildcard component: the whole array
    with pytest.raises(KeyError):
        norm_addr(("a", 1, 2, 3, 4, "b"))
    nested = C[1, 2, "x"].set(5.0)
    assert (1, 2, "x") in nested and nested[1, 2, "x"] == 5.0 and (1, 3, "x") not in nested
    grid = C["x"].set(np.arange(6.0).reshape(2, 3))
    assert (1, 2, "x") in grid and grid[1, 2, "x"] == 5.0 and np.array_equal(grid[:, :, "x"], np.arange(6.0).reshape(2, 3))
    assert np.array_equal(grid[1, :, "x"], np.array([3.0, 4.0, 5.0]))

    @genjax.gen
    def submodel():
        x = genjax.normal(0.0, 1.0) @ "x"
        y = genjax.normal(x, 0.01) @ "y"
        return y

    @genjax.gen
    def step(carry, _):
        pos = genjax.mv_normal_diag(carry, np.array([0.1, 0.1])) @ "pos"
        genjax.mv_normal_diag(pos, np.array([0.2, 0.2])) @ "obs_pos"
        return pos, pos

    @genjax.gen
    def noisy(x, w):
        return genjax.normal(w * x, 0.5) @ "y"

    @genjax.gen
    def model(xs):
        a = submodel() @ "a"
        _ = submodel() @ "b"
        y0 = genjax.normal(0.5, 0.01) @ "init_pos"
        _, tracks = step.scan(n=3)(genjax.array([0.0, y0]), None) @ "tracks"
        w = genjax.normal(0.0, 1.0) @ "w"
        ys = noisy.vmap(in_axes=(0, None))(xs, w) @ "ys"
        return a

    sl, _ = model.site_list((np.arange(4.0),))
    addrs = [s.addr for s in sl.sites]
    assert addrs[:5] == [("a", "x"), ("a", "y"), ("b", "x"), ("b", "y"), "init_pos"]
    assert (("tracks", "pos"), 0) in addrs and (("tracks", "obs_pos"), 2) in addrs and (("ys", "y"), 3) in addrs and "w" in addrs
    # the scan's initial carry [0, y0] is an affine read of init_pos
    p0 = sl[(("tracks", "pos"), 0)].params[0]
    assert p0.src == "init_pos"
    # address reuse across callees is detected on the full path only
    @genjax.gen
    def bad():
        submodel() @ "a"
        submodel() @ "a"
    with pytest.raises(genjax.AddressReuse):
        bad.site_list(())

    obs = ChoiceMap.empty().at["tracks", :, "obs_pos"].set(np.arange(6.0).reshape(3, 2)).at["a", "y"].set(3.0)
    assert ("tracks", 1, "obs_pos") in obs and ("a", "y") in obs and "a" not in obs
    np.testing.assert_array_equal(obs["tracks", 2, "obs_pos"], [4.0, 5.0])
    assert obs.get_submap("a")["y"] == 3.0 and obs("a")["y"] == 3.0
    sel = Selection.at["tracks", ..., "pos"]
    assert sel.check((("tracks", "pos"), 1)) and not sel.check((("tracks", "obs_pos"), 1)) and not sel.check("pos")
    assert S["a"].check(("a", "x")) and not S["a"].check(("b", "x")) and (S["a"] | S["w"]).check("w")
    assert Selection.at["x"].prefixed("b").check(("b", "x")) and Selection.all().prefixed("b").check(("b", "y"))
    # packing with a hierarchical constraint: observed sites leave the slot table
    prog, shared, pp = model.pack((np.arange(4.0),), obs, True)
    assert prog.slot_of[("a", "y")] == -1 and prog.slot_of[(("tracks", "obs_pos"), 1)] == -1 and prog.slot_of[("a", "x")] >= 0


def test_shard_message_counts_are_pairwise_consistent():
    """The C++ message-size arithmetic that drives the grouped ncclSend/ncclRecv (gjx_shard_message_counts, host
    only): for random and degenerate weight totals and 1..8 ranks, what rank r sends to d is exactly what d expects
    from r (the no-hang condition of the exchange), every output slot is produced once and stored once, and the
    counts equal the Python exchange's (distributed.resample_exchange uses the same plan)."""
    import ctypes as C
    from genjax_amd import _lib
    from genjax_amd.distributed import HostPlan, shard
    lib = _lib.load()
    rs = np.random.default_rng(0)
    cases = 0
    for G in (1, 2, 3, 4, 7, 8):
        for trial in range(40):
            N = int(rs.integers(G, 5000)) if trial % 3 else G * 1024
            kind = trial % 5
            if kind == 0:
                tot = rs.integers(1, 1 << 40, G)
            elif kind == 1:
                tot = np.zeros(G, np.int64); tot[rs.integers(0, G)] = 12345            # all mass on one rank
            elif kind == 2:
                tot = rs.integers(0, 3, G) * rs.integers(1, 1 << 30, G); tot[0] += (tot.sum() == 0)
            elif kind == 3:
                tot = np.full(G, 1 << 30, np.int64) + rs.integers(-1000, 1000, G)      # nearly balanced
            else:
                tot = (rs.random(G) ** 6 * (1 << 35)).astype(np.int64) + 1             # a few dominant ranks
            totals = [int(t) for t in tot]
            u = float(rs.random())
            send = np.zeros((G, G), np.int64)
            recv = np.zeros((G, G), np.int64)
            for r in range(G):
                hp = HostPlan(totals, r, u, N)
                p = A.GjxShardPlan()
                p.base, p.total, p.slot0, p.n_valid = hp.base, hp.total, hp.slot0, hp.n_valid
                p.own_lo, p.own_n, p.keep_lo, p.keep_hi, p.n_ranks, p.status = hp.own_lo, hp.own_n, hp.keep_lo, hp.keep_hi, G, hp.status
                for k, b in enumerate(hp.bounds):
                    p.bounds[k] = b
                sc, rc_, parts = (C.c_int64 * G)(), (C.c_int64 * G)(), (C.c_int64 * 4)()
                assert lib.gjx_shard_message_counts(C.byref(p), r, N, C.cast(sc, C.c_void_p), C.cast(rc_, C.c_void_p),
                                                    C.cast(parts, C.c_void_p)) == 0, lib.gjx_last_error()
                send[r], recv[r] = list(sc), list(rc_)
                # the Python exchange's counts (resample_exchange): overlaps of runs and shards
                owners = [shard(N, d, G) for d in range(G)]
                ov = lambda a0, a1, b0, b1: max(0, min(a1, b1) - max(a0, b0))
                want_s = [0 if d == r else ov(hp.slot0, hp.slot0 + hp.n_valid, lo, lo + k) for d, (lo, k) in enumerate(owners)]
                want_r = [0 if s_ == r else ov(hp.bounds[s_], hp.bounds[s_ + 1], hp.own_lo, hp.own_lo + hp.own_n) for s_ in range(G)]
                assert list(sc) == want_s and list(rc_) == want_r
                assert parts[0] + parts[1] == sum(want_s) and parts[2] + parts[3] == sum(want_r)
                assert hp.keep_hi - hp.keep_lo + sum(want_r) == hp.own_n          # every stored slot arrives exactly once
            np.testing.assert_array_equal(send, recv.T)                            # pairwise agreement: no rank waits for bytes nobody sends
            assert send.sum() + sum(HostPlan(totals, r, u, N).keep_hi - HostPlan(totals, r, u, N).keep_lo for r in range(G)) == N
            cases += 1
    assert cases == 240


class TestChoiceMapAlgebra:
    """The reference's ChoiceMap / Selection unit tests on the host mirror (genjax_amd/core.py), restated from
    /root/reference/tests/core/test_choice_maps.py (line numbers per case).  Out of scope there: Switch / dynamic-index
    choice maps, pytree validation."""

    def test_builder_set_and_membership(self):      # :295-310
        from genjax_amd.core import ChoiceMapBuilder as C
        chm = C["a", "b"].set(1)
        assert chm["a", "b"] == 1
        assert ("a", "b") in chm and "a" not in chm and "b" in chm("a")
        nested = C["x"].set(C["y"].set(2))
        assert nested["x", "y"] == 2 and ("x", "y") in nested and "y" not in nested
        assert C[()].set(1.0).get_value() == 1.0

    def test_builder_update(self):                  # :312-329
        from genjax_amd.core import ChoiceMapBuilder as C
        chm = C["x", "y"].set(2)
        assert chm.at["x"].update(lambda m: C["z"].set(m))["x", "z", "y"] == 2
        assert chm.at["x", "y"].update(lambda v: v * v)["x", "y"] == 4
        assert chm.at["q"].update(lambda m: C["z"].set(m))(("q", "z")).static_is_empty()
        assert chm.at["q"].update(lambda m: C["z"].set(2))["q", "z"] == 2

    def test_builder_empty_v_from_mapping_d_kw(self):   # :331-375
        from genjax_amd.core import ChoiceMap, ChoiceMapBuilder as C
        assert C.n() == ChoiceMap.empty() and C["x", "y"].n() == ChoiceMap.empty()
        assert C["a", "b"].set(1) == C["a", "b"].v(1)
        chm = C["base"].from_mapping([("a", 1.0), (("b", "c"), 2.0), (("b", "d", "e"), {"f": 3.0})])
        assert chm["base", "a"] == 1 and chm["base", "b", "c"] == 2 and chm["base", "b", "d", "e", "f"] == 3
        assert ("base", "a") in chm and ("base", "b", "c") in chm and ("b", "c") in chm("base")
        d = C["top"].d({"x": 3, "y": {"z": 4, "w": C["bottom"].d({"v": 5})}})
        assert d["top", "x"] == 3 and d["top", "y", "z"] == 4 and d["top", "y", "w", "bottom", "v"] == 5
        kw = C["root"].kw(a=1, b=C["nested"].kw(c=2, d={"deep": 3}))
        assert kw["root", "a"] == 1 and kw["root", "b", "nested", "c"] == 2 and kw["root", "b", "nested", "d", "deep"] == 3

    def test_extend_through_at(self):               # :465-499
        from genjax_amd.core import ChoiceMap
        initial = ChoiceMap.kw(x=1, y={"z": 2})
        ext = initial.at["y", "w"].set(3)
        assert ext["x"] == 1 and ext["y", "z"] == 2 and ext["y", "w"] == 3
        multi = initial.at["y", "w"].set(3).at["a", "b", "c"].set(4)
        assert multi["y", "w"] == 3 and multi["a", "b", "c"] == 4 and multi["x"] == 1
        assert initial.at["y", "z"].set(5)["y", "z"] == 5
        nested = initial.at["nested"].set(ChoiceMap.kw(a=6, b=7))
        assert nested["nested", "a"] == 6 and nested["nested", "b"] == 7 and nested["y", "z"] == 2

    def test_filter_mask_extend(self):              # :501-523
        from genjax_amd.core import ChoiceMap, SelectionBuilder as S
        chm = ChoiceMap.kw(x=1, y=2, z=3)
        f = (S["x"] | S["y"]).filter(chm)
        assert f["x"] == 1 and f["y"] == 2 and "z" not in f
        assert ChoiceMap.kw(x=1, y=2).mask(True) == ChoiceMap.kw(x=1, y=2)
        assert ChoiceMap.kw(x=1, y=2).mask(False).static_is_empty()
        e = ChoiceMap.choice(1).extend("a", "b")
        assert e["a", "b"] == 1 and e.get_value() is None and e.get_submap("a", "b").get_value() == 1
        assert ChoiceMap.empty().extend("a", "b").static_is_empty()

    def test_merge_xor_or_and(self):                # :709-793
        from genjax_amd.core import ChoiceMap
        a, b = ChoiceMap.kw(x=1), ChoiceMap.kw(y=2)
        m = a.merge(b)
        assert m["x"] == 1 and m["y"] == 2 and m == (a | b)
        x = a ^ b
        assert x["x"] == 1 and x["y"] == 2
        assert (ChoiceMap.empty() ^ ChoiceMap.empty()).static_is_empty()
        assert (a ^ ChoiceMap.empty()) == a and (ChoiceMap.empty() ^ a) == a
        with pytest.raises(ValueError):
            ChoiceMap.kw(x=1) ^ ChoiceMap.kw(x=2)
        o = a | b
        assert o.get_value() is None and (a | ChoiceMap.empty()) == a and (ChoiceMap.empty() | a) == a
        assert (ChoiceMap.kw(x=1) | ChoiceMap.kw(x=2))["x"] == 1          # left-biased
        c1, c2 = ChoiceMap.kw(x=1, y=2, z=3), ChoiceMap.kw(y=20, z=30, w=40)
        n = c1 & c2
        assert "x" not in n and "w" not in n and n["y"] == 20 and n["z"] == 30
        assert (c1 & ChoiceMap.empty()).static_is_empty() and (ChoiceMap.empty() & c1).static_is_empty()
        n1, n2 = ChoiceMap.kw(a={"b": 1, "c": 2}, d=3), ChoiceMap.kw(a={"b": 10, "d": 20}, d=30)
        nn = n1 & n2
        assert nn["a", "b"] == 10 and "c" not in nn("a") and "d" not in nn("a") and nn["d"] == 30

    def test_call_getitem_contains_selection(self):     # :719-724, :795-810
        from genjax_amd.core import ChoiceMap, ChoiceMapNoValueAtAddress
        chm = ChoiceMap.kw(x={"y": 1})
        assert chm("x")("y") == ChoiceMap.choice(1)
        assert "x" not in chm and "y" in chm("x") and ("x", "y") in chm and "z" not in chm
        with pytest.raises(ChoiceMapNoValueAtAddress, match="y"):
            ChoiceMap.kw(x=1)["y"]
        sel = ChoiceMap.kw(x=1, y=2).get_selection()
        assert sel["x"] and sel["y"] and not sel["z"]
        assert ChoiceMap.empty().static_is_empty() and not ChoiceMap.kw(x=1).static_is_empty()

    def test_submap_path_can_be_split_or_splatted(self):    # :1170-1202
        from genjax_amd.core import ChoiceMap
        chm = ChoiceMap.from_mapping([(("a", "b", "c"), 1.0), (("a", "d"), 2.0)])
        assert chm.get_submap("a", "b", "c") == chm.get_submap(("a", "b", "c")) == chm("a")("b")("c")
        assert chm.get_submap("a").get_submap("b", "c").get_value() == 1.0
        assert chm.get_submap("a", "d").get_value() == 2.0

    def test_choice_kv_d_from_mapping_simplify(self):   # :412-463, :662-687
        from genjax_amd import Mask
        from genjax_amd.core import ChoiceMap, ChoiceMapBuilder as C, ChoiceMapNoValueAtAddress, SelectionBuilder as S
        assert ChoiceMap.empty().static_is_empty()
        choice = ChoiceMap.choice(42.0)
        assert choice.get_value() == 42.0 and choice.has_value() and () in choice
        assert ChoiceMap.choice(Mask(42.0, False)).static_is_empty()
        assert ChoiceMap.choice(Mask(42.0, True)) == ChoiceMap.choice(42.0)
        mv = Mask(42.0, np.array(False))
        assert ChoiceMap.choice(mv).get_value() == mv
        assert ChoiceMap.choice(np.ones((0,))).static_is_empty()
        kv = ChoiceMap.kw(x=1, y=2)
        assert kv["x"] == 1 and kv["y"] == 2 and "x" in kv and "y" in kv and "other_value" not in kv
        d = ChoiceMap.d({"a": 1, "b": {"c": 2, "d": {"e": 3}}})
        assert d["a"] == 1 and d["b", "c"] == 2 and d["b", "d", "e"] == 3 and "a" in d and ("b", "c") in d and ("b", "d", "e") in d
        fm = ChoiceMap.from_mapping([("x", 1), (("y", "z"), 2), (("w", "v", "u"), 3)])
        assert fm["x"] == 1 and fm["y", "z"] == 2 and fm["w", "v", "u"] == 3 and ("w", "v", "u") in fm
        outer = ChoiceMap.kw(x=ChoiceMap.kw(a=1, b=2), y=3)
        assert outer["x", "a"] == 1 and outer["x", "b"] == 2 and outer["y"] == 3
        root = ChoiceMap.kw(r=ChoiceMap.kw(p=ChoiceMap.kw(m=4, n=5), q=6), s=7)
        assert root["r", "p", "m"] == 4 and root["r", "p", "n"] == 5 and root["r", "q"] == 6 and root["s"] == 7
        xyz = ChoiceMap.d({"x": 1, "y": 2, "z": 3})
        or_chm, xor_chm = xyz.filter(S["x"]) | xyz.filter(S["y"]), xyz.filter(S["x"]) ^ xyz.filter(S["y"])
        assert or_chm.simplify() == xor_chm.simplify() == ChoiceMap.d({"x": 1, "y": 2})
        assert or_chm["x"] == 1 and or_chm["y"] == 2
        with pytest.raises(ChoiceMapNoValueAtAddress, match="z"):
            or_chm["z"]

    def test_choicemap_validation(self):            # :875-927
        import genjax_amd as genjax
        from genjax_amd.core import ChoiceMap

        @genjax.gen
        def model(x):
            y = genjax.normal(x, 1.0) @ "y"
            z = genjax.bernoulli(probs=0.5) @ "z"
            return y + z

        assert ChoiceMap.kw(y=1.0, z=1).invalid_subset(model, (0.0,)) is None
        bad1 = ChoiceMap.kw(x=1.0)
        assert bad1.invalid_subset(model, (0.0,)) == bad1
        assert ChoiceMap.kw(y=1.0, z=1, extra=0.5).invalid_subset(model, (0.0,)) == ChoiceMap.kw(extra=0.5)

        @genjax.gen
        def inner_model():
            a = genjax.normal(0.0, 1.0) @ "a"
            b = genjax.bernoulli(probs=0.5) @ "b"
            return a + b

        @genjax.gen
        def outer_model():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = inner_model() @ "y"
            return x + y

        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5, b=1)).invalid_subset(outer_model, ()) is None
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5)).invalid_subset(outer_model, ()) is None          # a missing address is fine
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5, b=1, c=2.0)).invalid_subset(outer_model, ()) == ChoiceMap.kw(y=ChoiceMap.kw(c=2.0))
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=0.5, b=1), z=3.0).invalid_subset(outer_model, ()) == ChoiceMap.kw(z=3.0)

    def test_validation_of_sequences_and_slices(self):   # :929-979, :1025-1081
        import genjax_amd as genjax
        from genjax_amd.core import ChoiceMap, ChoiceMapBuilder as C

        @genjax.gen
        def inner_model(x):
            a = genjax.normal(x, 1.0) @ "a"
            b = genjax.bernoulli(probs=0.5) @ "b"
            return a + b

        @genjax.gen
        def outer_model():
            x = genjax.normal(0.0, 1.0) @ "x"
            y = inner_model.vmap(in_axes=(0,))(np.array([1.0, 2.0, 3.0], np.float32)) @ "y"
            return x

        a_, b_ = np.array([0.5, 1.5, 2.5]), np.array([1, 0, 1])
        assert ChoiceMap.kw(x=1.0, y=C[:].set(ChoiceMap.kw(a=a_, b=b_))).invalid_subset(outer_model, ()) is None
        assert ChoiceMap.kw(x=1.0, y=ChoiceMap.kw(a=a_, b=b_)).invalid_subset(outer_model, ()) is None    # the index layer is optional
        c_ = np.array([0.1, 0.2, 0.3])
        bad = ChoiceMap.kw(x=1.0, y=C[:].set(ChoiceMap.kw(a=a_, b=b_, c=c_))).invalid_subset(outer_model, ())
        assert bad == C["y", :, "c"].set(c_)

        @genjax.gen
        def step(mean):
            return genjax.normal(mean, 1.0) @ "x"

        chain = step.iterate(n=4)
        xs = np.array([0.5, 1.2, 0.8, 0.9])
        assert C[:, "x"].set(xs).invalid_subset(chain, (1.0,)) is None
        assert C["x"].set(xs).invalid_subset(chain, (1.0,)) is None
        assert C[:].set({"x": xs, "z": xs}).invalid_subset(chain, (1.0,)) == C[:, "z"].set(xs)

        for partial in (slice(None, 3), slice(0, 3), slice(0, 3, 1)):
            with pytest.raises(ValueError):
                C[partial, "x"].set(np.array([1, 2]))
        vals = np.arange(10)
        chm = C[:, "x"].set(vals)
        assert np.array_equal(chm[:, "x"], vals) and chm[1, "x"] == vals[1] and chm[np.int64(5), "x"] == vals[5]
        assert np.array_equal(chm[0:4, "x"], vals[0:4])

    def test_index_only_addresses(self):            # :812-834, :864-869
        from genjax_amd.core import ChoiceMapBuilder as C, ChoiceMapNoValueAtAddress, SelectionBuilder as S
        xs, ys = np.array([1.0, 2.0, 3.0]), np.array([4.0, 5.0, 6.0])
        chm = C[:].set({"x": xs, "y": ys})
        only_x = chm.filter(S["x"])
        assert (only_x[:, "x"] == xs).all() and only_x[0, "x"] == 1.0 and only_x[1, "x"] == 2.0 and only_x[2, "x"] == 3.0
        with pytest.raises(ChoiceMapNoValueAtAddress):
            only_x[:, "y"]
        c0 = C[0].set({"x": 1.0, "y": 2.0})
        assert c0[0, "x"] == 1.0 and c0[0, "y"] == 2.0

    def test_wildcard_and_leaf_selections(self):    # :55-60, :81-116
        from genjax_amd.core import Selection, SelectionBuilder as S
        sel = S["x"] | S[..., "y"]
        assert sel["x"] and sel["any_address", "y"] and sel["rando", "y", "tail"] and not sel["q"]
        # the wildcard consumes exactly ONE component (the reference's StaticSel with an Ellipsis component): a top-level "y"
        # is not selected, step 3 of a sequence "y" (the reference's chm[3, "y"]) is
        wild = S[..., "y"]
        assert not wild["y"] and wild.check(("y", 3)) and not wild.check(("x", 3)) and wild["z", "y"]
        assert S["tracks", ..., "pos"].check((("tracks", "pos"), 1)) and not S["tracks", ..., "pos"].check(("tracks", "pos"))
        assert S.all == Selection.all() and S.all["x"] and S.all["y", "z"] and S.all[()]
        assert S.none == Selection.none() and not S.none["x"] and not S.none["y", "z"] and not S.none[()]
        leaf = S.leaf
        assert leaf == Selection.leaf()
        leaf = leaf.extend("a", "b")
        assert leaf["a", "b"] and not leaf["a"] and not leaf["a", "b", "c"]
        assert S[()] == Selection.leaf() and () in S[()]
        exact = Selection.leaf().extend("x", "y")
        assert not exact["x"] and exact["x", "y"] and not exact["x", "y", "z"]
        with pytest.raises(TypeError):
            exact[..., "y"]

    def test_selection_filter_combination_contains_subselection(self):   # :183-292
        from genjax_amd.core import ChoiceMap, ChoiceMapBuilder as C, Selection, SelectionBuilder as S
        chm = ChoiceMap.kw(x=1, y=2, z=3)
        f_ = (S["x"] | S["y"]).filter(chm)
        assert "x" in f_ and "y" in f_ and "z" not in f_ and f_["x"] == 1 and f_["y"] == 2
        assert Selection.none().filter(chm).static_is_empty() and Selection.all().filter(chm) == chm
        nf = (S["a", "b"] | S["d"]).filter(ChoiceMap.kw(a={"b": 1, "c": 2}, d=3))
        assert "d" in nf and "b" in nf("a") and "c" not in nf("a")
        comb = ((S["x"] | S["y"]) & (S["y"] | S["z"])) | S["w"]
        assert not comb["x"] and comb["y"] and not comb["z"] and comb["w"]
        sel = S["x"] | S["y", "z"]
        assert "x" in sel and sel["x"] and ("y", "z") in sel and sel["y", "z"] and "y" not in sel and not sel["y"] and "w" not in sel
        nested = S["c"].extend("a", "b")
        assert ("a", "b", "c") in nested and nested["a", "b", "c"] and ("a", "b") not in nested and not nested["a", "b"]
        assert not nested("a")("b").check() and nested("a")("b")("c").check()
        with pytest.raises(TypeError):
            (S["a", "b", "c"] | S["x", "y", "z"])["a", ..., ...]
        xy = Selection.at["x", "y"]
        assert not xy[()] and xy["x", "y"] and not xy["other_address"]
        n = Selection.at["x"].extend("y")
        assert n["y", "x"] and not n["y"]
        cs = (C["x", "y"].set(3.0) | C["z"].set(5.0)).get_selection()
        assert cs["x", "y"] and cs["z"] and not cs["w"] and cs("x")["y"]
        assert ChoiceMap.empty().get_selection() == Selection.none()

    def test_selections(self):                      # :40-53, :62-79, :118-181, :228-253
        from genjax_amd.core import Selection, SelectionBuilder as S
        new = S["x"] | S["z", "y"]
        assert new["x"] and new["z", "y"] and new["z", "y", "tail"]
        assert S["x"]["x", "y", "z"] and S["x", "y", "z"]["x", "y", "z"] and not S["x", "y", "z"]["x"] and not S["x", "y", "z"]["x", "y"]
        all_sel, none_sel = Selection.all(), Selection.none()
        assert all_sel == ~~all_sel and all_sel["x"] and all_sel["y", "z"] and all_sel[()]
        assert none_sel == ~~none_sel and not none_sel["x"] and not none_sel["y", "z"] and not none_sel[()]
        assert Selection.none().extend("a", "b") == Selection.none()
        assert S.all == Selection.all() and S.none == Selection.none() and ~all_sel == none_sel and ~none_sel == all_sel
        sel = S["x"] | S["y"]
        assert not (~sel)["x"] and not (~sel)["y"] and (~sel)["z"] and ~~sel == sel
        both = (S["x"] | S["y"]) & (S["y"] | S["z"])
        assert both["y"] and not both["x"] and not both["z"]
        assert "x" in S["x"] and ("x", "y") in S["x", "y"] and "y" not in S["x"]
        ext = S["x"].extend("a", "b")
        assert ext["a", "b", "x"] and not ext["x"]


class TestMask:
    """reference tests/core/generative/test_functional_types.py:28-63, 65-75, 139-150, 154-227, 329-366 on the host type
    (genjax_amd.Mask): concrete flags are Python bools, flag arrays hold one flag per leading index."""

    def test_constructor_unmask_build_maybe(self):
        from genjax_amd import Mask
        m = Mask(value=42, flag=True)
        assert m.value == 42 and m.flag is True and Mask(value=42).flag is True
        assert Mask(42, True).unmask() == 42 and Mask(42, True).unmask(default=0) == 42 and Mask(42, False).unmask(default=0) == 0
        with pytest.raises(Exception):
            Mask(42, False).unmask()
        tree = {"a": 1, "b": [2, 3], "c": {"d": 4}}
        assert Mask(tree, True).unmask() == tree
        default = {"a": 0, "b": [0, 0], "c": {"d": 0}}
        assert Mask(tree, False).unmask(default=default) == default
        b = Mask.build(42, True)
        assert isinstance(b, Mask) and b.flag is True and b.value == 42
        nested = Mask.build(Mask.build(42, True), False)
        assert isinstance(nested, Mask) and nested.flag is False and nested.value == 42
        v = Mask.build(np.arange(10), np.ones(10, bool))
        n2 = Mask.build(v, False)
        assert np.array_equal(n2.value, np.arange(10)) and np.array_equal(n2.primal_flag(), np.zeros(10, bool))
        assert Mask.maybe_mask(42, True) == 42 and Mask.maybe_mask(42, False) is None
        assert Mask.maybe_mask(Mask(42, True), True) == 42 and Mask.maybe_mask(Mask(42, True), False) is None

    def test_or_xor_not_indexing(self):
        from genjax_amd import Mask
        for fa, fb, flag, val in ((True, True, True, 42), (True, False, True, 42), (False, True, True, 43), (False, False, False, None)):
            r = Mask(42, fa) | Mask(43, fb)
            assert r.primal_flag() is flag and (val is None or r.value == val)
        for fa, fb, flag, val in ((True, True, False, None), (True, False, True, 42), (False, True, True, 43), (False, False, False, None)):
            r = Mask(42, fa) ^ Mask(43, fb)
            assert r.primal_flag() is flag and (val is None or r.value == val)
        a = Mask(np.array([42, 42, 42, 42]), np.array([True, True, False, False]))
        b = Mask(np.array([43, 43, 43, 43]), np.array([False, True, False, True]))
        assert np.array_equal((a | b).primal_flag(), [True, True, False, True]) and np.array_equal((a | b).value[[0, 3]], [42, 43])
        assert np.array_equal((a ^ b).primal_flag(), [True, False, False, True]) and np.array_equal((a ^ b).value[[0, 3]], [42, 43])
        assert ~Mask(1.0, True) == Mask(1.0, False) and ~Mask(2.0, False) == Mask(2.0, True)
        assert ~Mask(np.array([1.0, 2.0]), np.array([True, False])) == Mask(np.array([1.0, 2.0]), np.array([False, True]))
        s_ = Mask(np.array([[1, 2], [3, 4]]), True)
        assert s_[0, 1].value == 2 and s_[0, 1].primal_flag() is True
        v = Mask(np.array([[1, 2], [3, 4]]), np.array([True, False]))
        assert v[0, 1].value == 2 and v[0, 1].primal_flag() is True and v[1, 0].value == 3 and v[1, 0].primal_flag() is False


def test_gen_transfers_the_function_metadata():
    """reference tests/generative_functions/test_static_gen_fn.py:39-84"""
    import genjax_amd as genjax

    def original_function(x: float, y: float) -> float:
        """This is a test function that adds two numbers."""
        return x + y

    w = genjax.gen(original_function)
    assert w.__doc__ == original_function.__doc__ and w.__name__ == original_function.__name__
    assert w.__module__ == original_function.__module__ and w.__qualname__ == original_function.__qualname__
    assert getattr(w, "__wrapped__") == original_function


def test_reference_names_are_importable():
    """names a reference user imports from `genjax` for this path (src/genjax/{core,generative_functions,inference}/__init__.py)"""
    import genjax_amd as g
    for n in ("gen", "ChoiceMap", "ChoiceMapBuilder", "Selection", "SelectionBuilder", "Mask", "Diff", "NoChange", "UnknownChange",
              "normal", "beta", "flip", "bernoulli", "categorical", "mv_normal_diag", "gamma", "exponential", "laplace", "poisson",
              "scan", "vmap", "repeat", "iterate", "iterate_final", "accumulate", "reduce", "Scan", "Vmap",
              "Target", "ImportanceK", "Importance", "ChangeTarget", "Algorithm", "SMCAlgorithm", "ParticleCollection",
              "Update", "Regenerate", "Rejuvenate", "HMC", "SafeHMC", "StaticRequest", "IndexRequest", "smc", "requests", "marginal"):
        assert hasattr(g, n), n
    assert g.smc.ImportanceK is g.ImportanceK and g.requests.HMC is g.HMC


def test_tracing_of_inline_methods_closures_and_scan_lengths():
    """host-side lowering of the forms added for the reference's static / scan tests: the site lists (no device needed)"""
    import genjax_amd as genjax

    @genjax.gen
    def simple_normal():
        y1 = genjax.normal(0.0, 1.0) @ "y1"
        y2 = genjax.normal(0.0, 1.0) @ "y2"
        return y1 + y2

    @genjax.gen
    def higher():
        return simple_normal.inline()

    @genjax.gen
    def nested():
        return simple_normal() @ "sub"

    addrs = lambda g, args=(): [s.addr for s in g.site_list(args)[0].sites]
    assert addrs(higher) == ["y1", "y2"] and addrs(nested) == [("sub", "y1"), ("sub", "y2")]

    class Model:
        def __init__(self, loc):
            self.loc = loc

        @genjax.gen
        def run(self, x):
            return genjax.normal(self.loc, 1.0) @ "y" + genjax.normal(x, 1.0) @ "z"

    m = Model(4.0)
    assert addrs(m.run, (1.0,)) == ["y", "z"] and m.run.partial_args == (m,) and Model.run.partial_args == ()

    @genjax.gen
    def model3(x, y, z):
        return genjax.normal(x, y + z) @ "x"

    assert model3.partial_apply(1.0).partial_apply(1.0).partial_args == (1.0, 1.0)

    @genjax.gen
    def walk(x, std):
        nx = genjax.normal(x, std) @ "x"
        return nx, nx

    xs = np.array([2.0, 4.0, 3.0], np.float32)
    assert addrs(walk.scan(), (0.0, xs)) == [("x", 0), ("x", 1), ("x", 2)] == addrs(walk.scan(n=3), (0.0, xs))
    assert addrs(walk.scan(n=0), (0.0, np.zeros(0, np.float32))) == []
    with pytest.raises(ValueError, match="different leading axis sizes: 1, 2"):
        @genjax.gen
        def foo(shift, d):
            return genjax.normal(d["loc"], d["scale"]) @ "x" + shift, None
        foo.scan().site_list((1.0, {"loc": np.array([10.0, 12.0], np.float32), "scale": np.array([1.0], np.float32)}))

    @genjax.gen
    def add(acc, v):
        return acc + genjax.normal(v, 1.0) @ "n"

    assert addrs(add.accumulate(), (0.0, xs)) == [("n", 0), ("n", 1), ("n", 2)] == addrs(add.reduce(), (0.0, xs))


def test_round6_distribution_wrappers_follow_the_tfp_signatures():
    """negative_binomial(total_count, logits) — a bare second argument is LOGITS, probs= is folded to logits on the host (the
    reference passes tfd.NegativeBinomial through unwrapped, tensorflow_probability/__init__.py:249) — and the two-parameter forms of
    the other round-6 wrappers"""
    import genjax_amd as genjax
    from genjax_amd import _abi as A

    @genjax.gen
    def m():
        r = genjax.gamma(2.0, 1.0) @ "r"
        genjax.negative_binomial(r, probs=0.57) @ "k"
        genjax.negative_binomial(3.0, 0.2) @ "k2"
        genjax.negative_binomial(total_count=3.0, logits=r) @ "k3"
        genjax.von_mises(0.1, r) @ "th"
        genjax.half_student_t(4.0, 0.0, r) @ "h"
        genjax.truncated_cauchy(0.0, 1.0, -2.0, r) @ "tc"

    sl, _ = m.site_list(())
    assert sl["k"].kind == A.NEGATIVE_BINOMIAL and sl["k"].params[0].src == "r"
    np.testing.assert_allclose(sl["k"].params[1].values, [np.log(0.57 / 0.43)], rtol=1e-6)
    np.testing.assert_allclose(sl["k2"].params[1].values, [0.2])
    assert sl["k3"].params[1].src == "r" and sl["th"].kind == A.VON_MISES and sl["th"].params[1].src == "r"
    assert len(sl["h"].params) == 3 and len(sl["tc"].params) == 4 and sl["tc"].params[3].src == "r"
    with pytest.raises(TypeError):
        genjax.negative_binomial(3.0) @ "x"
    assert A.NEGATIVE_BINOMIAL in A.NO_GRADIENT_KINDS and A.VON_MISES not in A.NO_GRADIENT_KINDS and A.KIND_MAX == 37
