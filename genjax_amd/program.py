"""Model program: the explicit site list that replaces the reference's Jaxpr interpretation.

The reference stages a ``@gen`` Python function to a Jaxpr and interprets it once per GFI call
(generative_functions/static.py:340-399, core/compiler/interpreters/stateful.py:49-72).  Here a
generative function is a straight-line list of *sites* whose distribution parameters are simple
expression forms over earlier values and constants; that list is packed into the ``gjx_program``
struct of ``include/gjx.h`` and handed to the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import itertools
from dataclasses import dataclass, field
from typing import Any, Sequence

import numpy as np

from . import _abi as A


# ---------------------------------------------------------------------------------------------
# parameter expressions
# ---------------------------------------------------------------------------------------------
@dataclass
class Param:
    """One distribution parameter as a function of earlier choices (host-side description)."""

    op: int
    values: np.ndarray | None = None   # CONST: vector; GATHER: table [n][len]; AFFINE: bias
    matrix: np.ndarray | None = None   # AFFINE: [dim][n]
    src: str | None = None             # address of the source site (VALUE/AFFINE/GATHER index)
    src_elem: int = 0                  # first element of the source site used
    length: int = 1                    # VALUE: number of source elements (1 = broadcast)
    xf: int = A.XF_NONE
    terms: list | None = None          # AFFINE over several sites: [(addr, matrix [dim][site dim]), ...]
    # inside a plate (gjx.h "Plates"): the source element advances by d_elem per instance; `values` / `matrix` (and the
    # matrices of `terms`) carry a leading instance axis when inst_values / inst_matrix is set
    d_elem: int = 0
    inst_values: bool = False
    inst_matrix: bool = False
    # VGATHER (gjx.h GJX_P_VGATHER): row `src`-th of the earlier vector-valued choice `vsrc` (from element vsrc_elem: vn rows of vlen values)
    vsrc: Any = None
    vsrc_elem: int = 0
    vn: int = 0
    vlen: int = 1
    # EXPR (gjx.h GJX_P_EXPR): the output nodes of a general elementwise expression (expr.py); len(outs) = 1 or the site's dim
    outs: tuple | None = None
    inst_outs: list | None = None      # EXPR parameter of a plate body site: the output nodes of every instance (expr.lower_plate)

    @staticmethod
    def expr(outs, xf=A.XF_NONE) -> "Param":
        """a general elementwise expression of earlier choices and constants: ``outs`` = its output nodes (expr.py)"""
        return Param(A.P_EXPR, outs=tuple(outs), xf=xf)

    def sources(self) -> list:
        """addresses of the choices this parameter reads"""
        if self.op == A.P_CONST:
            return []
        if self.op == A.P_EXPR:
            from . import expr as E
            return E.sources(self.outs)
        if self.terms:
            return [a for a, _ in self.terms]
        return [self.src, self.vsrc] if self.op == A.P_VGATHER else [self.src]

    @staticmethod
    def const(v, xf=A.XF_NONE) -> "Param":
        return Param(A.P_CONST, values=np.atleast_1d(np.asarray(v, np.float32)).ravel(), xf=xf)

    @staticmethod
    def value(src: str, length: int = 1, elem: int = 0, xf=A.XF_NONE) -> "Param":
        return Param(A.P_VALUE, src=src, src_elem=elem, length=length, xf=xf)

    @staticmethod
    def gather(table, src: str, xf=A.XF_NONE) -> "Param":
        t = np.asarray(table, np.float32)
        if t.ndim == 1:
            t = t[:, None]
        return Param(A.P_GATHER, values=t, src=src, xf=xf)

    @staticmethod
    def vgather(vsrc, n: int, src: str, vlen: int = 1, xf=A.XF_NONE) -> "Param":
        """``value(vsrc)[idx]``: one of the n rows (vlen values each) of an earlier choice, picked by the discrete choice ``src``"""
        return Param(A.P_VGATHER, src=src, vsrc=vsrc, vn=int(n), vlen=int(vlen), xf=xf)

    @staticmethod
    def affine(matrix, src: str, bias=0.0, elem: int = 0, xf=A.XF_NONE) -> "Param":
        m = np.atleast_2d(np.asarray(matrix, np.float32))
        return Param(A.P_AFFINE, values=np.atleast_1d(np.asarray(bias, np.float32)).ravel(),
                     matrix=m, src=src, src_elem=elem, xf=xf)


def _affine_multi(terms, bias=0.0, xf=A.XF_NONE) -> Param:
    """bias + sum_t M_t @ value(site t): one P_AFFINE over the slot range spanning the latent sources; observed
    sources fold into the bias when the program is packed."""
    ts = [(a, np.atleast_2d(np.asarray(m, np.float32))) for a, m in terms]
    return Param(A.P_AFFINE, values=np.atleast_1d(np.asarray(bias, np.float32)).ravel(), src=ts[0][0], xf=xf, terms=ts)


Param.affine_multi = staticmethod(_affine_multi)


def as_param(p: Any) -> Param:
    if isinstance(p, Param):
        return p
    if hasattr(p, "as_param"):
        return p.as_param()
    return Param.const(p)


@dataclass
class Site:
    addr: str
    kind: int
    params: list[Param]
    dim: int = 1
    ncat: int = 0
    slot: int = -1
    scan: int = 0     # gjx_site.scan: (scan_id << 20) | (step + 1) for the sites of a Scan step, else 0
    plate: int = 0    # gjx_site.plate / plate_n: body site of a device plate of plate_n instances (dim = ONE instance's size)
    plate_n: int = 0

    @property
    def rows(self) -> int:
        """rows of choices[][] the site owns"""
        return self.dim * (self.plate_n if self.plate else 1)


N_PARAMS = {
    A.NORMAL: 2, A.FLIP: 1, A.BERNOULLI_LOGITS: 1, A.BETA: 2, A.CATEGORICAL_LOGITS: 1,
    A.CATEGORICAL_PROBS: 1, A.UNIFORM: 2, A.MVNORMAL_DIAG: 2, A.EXPONENTIAL: 1, A.HALF_NORMAL: 1,
    A.LAPLACE: 2, A.LOG_NORMAL: 2, A.CAUCHY: 2, A.GAMMA: 2,
    A.STUDENT_T: 3, A.TRUNCATED_NORMAL: 4, A.POISSON: 1, A.GEOMETRIC: 1, A.DIRICHLET: 1, A.GUMBEL: 2, A.HALF_CAUCHY: 2,
    A.INVERSE_GAMMA: 2, A.WEIBULL: 2, A.LOGIT_NORMAL: 2, A.CHI2: 1,
    A.CHI: 1, A.EXP_GAMMA: 2, A.EXP_INVERSE_GAMMA: 2, A.HALF_STUDENT_T: 3, A.KUMARASWAMY: 2, A.MOYAL: 2, A.TRUNCATED_CAUCHY: 4,
    A.DOUBLESIDED_MAXWELL: 2, A.INVERSE_GAUSSIAN: 2, A.NEGATIVE_BINOMIAL: 2, A.VON_MISES: 2,
}


class AddressReuse(Exception):
    """Same address used twice in one program (reference: static.py:139)."""


class MissingAddress(Exception):
    """assess() without a value for a site (reference: static.py:147, 316-318)."""


_UIDS = itertools.count(1)     # gjx_program.uid of every PackedProgram of this process


@dataclass
class SiteList:
    """Ordered sites of a generative function (program order == the reference's trace order)."""

    sites: list[Site] = field(default_factory=list)
    n_slots: int = 0
    _index: dict = field(default_factory=dict, repr=False, compare=False)     # addr -> position (hashable addresses)

    def _pos(self, addr):
        if len(self._index) != len(self.sites):
            self._index = {s.addr: j for j, s in enumerate(self.sites)}
        return self._index.get(addr)

    def add(self, addr: str, kind: int, params: Sequence[Any], dim: int | None = None) -> Site:
        if self._pos(addr) is not None:
            raise AddressReuse(addr)
        ps = [as_param(p) for p in params]
        if len(ps) != N_PARAMS[kind]:
            raise TypeError(f"{A.KIND_NAMES[kind]} takes {N_PARAMS[kind]} parameters, got {len(ps)}")
        ncat = 0
        if kind in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS):
            p0 = ps[0]
            if p0.op == A.P_CONST:
                ncat = int(p0.values.size)
            elif p0.op == A.P_GATHER:
                ncat = int(p0.values.shape[1])
            elif p0.op == A.P_AFFINE:
                ncat = int((p0.terms[0][1] if p0.terms else p0.matrix).shape[0])
            elif p0.op == A.P_VGATHER:
                ncat = int(p0.vlen)
            elif p0.op == A.P_EXPR:
                ncat = len(p0.outs)
            else:
                ncat = int(p0.length)
            dim = 1
        if dim is None:
            dim = 1
            for p in ps:
                if p.op == A.P_CONST:
                    dim = max(dim, int(p.values.size))
                elif p.op == A.P_VALUE:
                    dim = max(dim, int(p.length))
                elif p.op == A.P_GATHER:
                    dim = max(dim, int(p.values.shape[1]))
                elif p.op == A.P_VGATHER:
                    dim = max(dim, int(p.vlen))
                elif p.op == A.P_EXPR:
                    dim = max(dim, len(p.outs))
                elif p.op == A.P_AFFINE:
                    dim = max(dim, int((p.terms[0][1] if p.terms else p.matrix).shape[0]))
        if kind == A.DIRICHLET and int(dim) > 256:
            raise ValueError("dirichlet: at most 256 components per site")
        site = Site(addr, kind, ps, int(dim), ncat, self.n_slots)
        self.sites.append(site)
        self._index[addr] = len(self.sites) - 1
        self.n_slots += int(dim)
        return site

    def __getitem__(self, addr: str) -> Site:
        j = self._pos(addr)
        if j is None:
            raise KeyError(addr)
        return self.sites[j]

    def __contains__(self, addr: str) -> bool:
        return self._pos(addr) is not None

    def addresses(self) -> list[str]:
        return [s.addr for s in self.sites]


# ---------------------------------------------------------------------------------------------
# plates: the instances of a vmapped kernel as ONE device site per kernel site
# ---------------------------------------------------------------------------------------------
PLATE_MIN = 8     # fewer instances stay unrolled (nothing to gain)


def _is_cat(kind: int) -> bool:
    return kind in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS)


def compact_plates(sl: SiteList, modes: dict, obs: dict, selected: Sequence, draws_ok: bool = True,
                   tagged: bool = True, vector_needs_table: bool = False) -> tuple[SiteList, dict, dict, tuple, dict]:
    """The host tracer unrolls ``kernel.vmap(...)(args)`` into n instances of the kernel's m sites, addressed
    ``(name, i)`` (combinators/vmap.py:193-218: per-instance sub-traces, summed weights).  The device gets ONE site per
    kernel site, in one of two forms (both lay a kernel site's values out instance after instance — element i * d + c is
    element c of instance i — so the logical addresses ``(name, i)`` map to the same rows either way):

    * VECTOR form (``_try_compact``): a plain site with n * d elements whose log-density is the sum over the instances
      (distribution.py:392-396 sums a shaped log-pdf) — the N rows of a regression likelihood become one site with an
      [N x P] affine parameter.  Possible when every parameter is a constant, the value of a site outside the plate, the
      value of an earlier site of the SAME instance with the same dimension, or an affine form over a site outside the
      plate; not for categorical sites, gathers, masks.  It is what the hand-written and the generated HMC kernels see.
    * PLATE form (``_try_plate``, ``tagged``): the m sites carry ``gjx_site.plate`` and the engines run ONE instance loop
      over them (include/gjx.h "Plates") with per-instance rows, observations, masks, tables and sources — categorical
      sites, gathers on an index of the same instance, per-instance masks, sources in an earlier plate.

    Anything neither form covers (instances that differ in structure or mode) stays unrolled, which is always correct.
    ``vector_needs_table`` (the HMC packing): the vector form only for plates whose values all come from the table (OBS_TAB: the
    regression likelihood, which the HMC kernels run on the matrix cores); a plate with per-chain values takes the plate form,
    whose rows stay in memory and are read instance by instance — a vector site's values would all be chain registers.

    ``draws_ok`` False (GJX_RNG_JAX32, whose streams follow the reference's key structure site by site): the vector form is
    used only for plates in which nothing is drawn; the plate form follows the reference's instance-key rule and may draw.

    -> (device site list, its modes / obs / selected, {logical addr: (device addr, element offset)})"""
    sel = set(selected)
    out = SiteList()
    dmodes, dobs, dsel, where = {}, {}, [], {}
    sites = sl.sites
    j = 0
    n_plates = 0

    def is_inst(s, i=None):
        return (isinstance(s.addr, tuple) and len(s.addr) == 2 and isinstance(s.addr[1], (int, np.integer))
                and (i is None or s.addr[1] == i))

    def resolve_param(p: Param) -> Param:
        """a parameter of a site OUTSIDE the plates that reads a compacted instance: read the device site at the instance's offset"""
        if p.op == A.P_CONST:
            return p
        if p.op == A.P_EXPR:            # leaves that name a compacted instance: the device site at the instance's offset
            from . import expr as E
            if not any(a_ in where for a_ in p.sources()):
                return p
            return dataclasses.replace(p, outs=tuple(E.rewrite_leaves(p.outs, lambda a_, e_: E.value(where[a_][0], e_ + where[a_][1]) if a_ in where else None)))
        if p.terms:
            if not any(a_ in where for a_, _ in p.terms):
                return p
            raise _Unrollable()         # (an affine form over several sites, one of them a compacted instance)
        if p.op == A.P_VGATHER and p.vsrc in where:
            daddr, off = where[p.vsrc]
            p = dataclasses.replace(p, vsrc=daddr, vsrc_elem=p.vsrc_elem + off)
        if p.src in where:
            daddr, off = where[p.src]
            return dataclasses.replace(p, src=daddr, src_elem=p.src_elem + off)
        return p

    def copy_plain(s):
        ns = Site(s.addr, s.kind, [resolve_param(p) for p in s.params], s.dim, s.ncat, out.n_slots, s.scan)
        out.sites.append(ns)
        out.n_slots += s.dim
        if s.addr in modes:
            dmodes[s.addr] = modes[s.addr]
        if s.addr in obs:
            dobs[s.addr] = obs[s.addr]
        if s.addr in sel:
            dsel.append(s.addr)

    while j < len(sites):
        s0 = sites[j]
        if not is_inst(s0, 0):
            copy_plain(s0)
            j += 1
            continue
        # instance 0: the kernel's sites; then count the instances that repeat its names
        m = 0
        while j + m < len(sites) and is_inst(sites[j + m], 0) and sites[j + m].scan == s0.scan:
            m += 1
        names = [sites[j + l].addr[0] for l in range(m)]
        n = 1
        while j + (n + 1) * m <= len(sites) and all(is_inst(sites[j + n * m + l], n) and sites[j + n * m + l].addr[0] == names[l]
                                                    and sites[j + n * m + l].scan == s0.scan for l in range(m)):
            n += 1
        group = [[sites[j + i * m + l] for i in range(n)] for l in range(m)]      # [kernel site][instance]
        plate = None
        if n >= PLATE_MIN:
            if s0.scan == 0:
                plate = _try_compact(group, names, n, modes, obs, sel, where)
                if plate is not None and not draws_ok and any(mode not in (A.MODE_OBS_TAB, A.MODE_OBS_SLOT) for _, mode, _, _ in plate):
                    plate = None
                if plate is not None and vector_needs_table and tagged and any(mode != A.MODE_OBS_TAB for _, mode, _, _ in plate):
                    plate = None
            if plate is None and tagged and m <= 48:
                plate = _try_plate(group, names, n, modes, obs, sel, where, n_plates + 1)
                if plate is not None:
                    n_plates += 1
        if plate is None:
            for i in range(n):
                for l in range(m):
                    copy_plain(sites[j + i * m + l])
        else:
            for l, (ns, mode, ov, is_sel) in enumerate(plate):
                ns.slot = out.n_slots
                ns.scan = s0.scan
                out.sites.append(ns)
                out.n_slots += ns.rows
                if mode is not None:
                    dmodes[ns.addr] = mode
                if ov is not None:
                    dobs[ns.addr] = ov
                if is_sel:
                    dsel.append(ns.addr)
                d = group[l][0].dim
                for i in range(n):
                    where[group[l][i].addr] = (ns.addr, i * d)
        j += n * m
    return out, dmodes, dobs, tuple(dsel), where


def fold_known(p: Param, known: dict, rows: int) -> Param:
    """a parameter with the sources whose values are KNOWN constants (observed sites of an earlier Scan step) folded in:
    VALUE -> the constant slice, GATHER -> the row, AFFINE -> bias + M v; an affine form over several sites keeps its unknown
    terms.  ``rows``: the reading site's event size."""
    if p.op == A.P_CONST:
        return p
    if p.op == A.P_EXPR:
        from . import expr as E
        if not any(a_ in known for a_ in p.sources()):
            return p
        outs = E.rewrite_leaves(p.outs, lambda a_, e_: E.const(np.asarray(known[a_], np.float64).ravel()[e_]) if a_ in known else None)
        if all(E.is_const(n) for n in outs):
            return Param.const([n[1] for n in outs], xf=p.xf)
        return dataclasses.replace(p, outs=tuple(outs))
    if p.terms:
        if not any(a_ in known for a_, _ in p.terms):
            return p
        bias = np.broadcast_to(p.values.astype(np.float64), (rows,)).copy() if p.values.size in (1, rows) else p.values.astype(np.float64).copy()
        rest = []
        for a_, m in p.terms:
            if a_ in known:
                bias = bias + m.astype(np.float64) @ np.asarray(known[a_], np.float64).ravel()[: m.shape[1]]
            else:
                rest.append((a_, m))
        if not rest:
            return Param.const(bias, xf=p.xf)
        return Param(A.P_AFFINE, values=bias.astype(np.float32), src=rest[0][0], xf=p.xf, terms=rest)
    if p.op == A.P_VGATHER:
        if p.vsrc in known:          # the indexed choice is known: an ordinary row gather of a constant table
            t = np.asarray(known[p.vsrc], np.float32).ravel()[p.vsrc_elem:p.vsrc_elem + p.vn * p.vlen].reshape(p.vn, p.vlen)
            return fold_known(Param(A.P_GATHER, values=t, src=p.src, src_elem=p.src_elem, xf=p.xf, d_elem=p.d_elem), known, rows)
        if p.src in known:           # the index is known: a plain read of that row
            idx = int(np.clip(int(np.asarray(known[p.src]).ravel()[p.src_elem]), 0, p.vn - 1))
            return Param.value(p.vsrc, length=p.vlen, elem=p.vsrc_elem + idx * p.vlen, xf=p.xf)
        return p
    if p.src not in known:
        return p
    v = np.asarray(known[p.src], np.float64).ravel()[p.src_elem:]
    if p.op == A.P_VALUE:
        return Param.const(v[: p.length], xf=p.xf)
    if p.op == A.P_GATHER:
        idx = int(np.clip(int(v[0]), 0, p.values.shape[0] - 1))
        return Param.const(p.values[idx], xf=p.xf)
    if p.op == A.P_AFFINE:
        n = p.matrix.shape[1]
        bias = np.broadcast_to(p.values, (rows,)) if p.values.size in (1, rows) else p.values
        return Param.const(bias + p.matrix.astype(np.float64) @ v[:n], xf=p.xf)
    raise ValueError(p.op)


class _Unrollable(Exception):
    """a site outside the plates reads compacted instances in a form the device sites cannot express: pack without plates"""


def _same(arrs) -> bool:
    a0 = arrs[0]
    return all(a.shape == a0.shape and np.array_equal(a, a0) for a in arrs[1:])


def _try_plate(group, names, n: int, modes: dict, obs: dict, sel: set, where: dict, pid: int):
    """PLATE form -> [(body Site, mode or None, observed values [n * d] or None, selected)] per kernel site, or None.
    Every parameter must have the same form in all instances; what differs between them is per-instance DATA: constants,
    gather tables, affine bias / matrices (stacked along a leading instance axis) and the source element, which must
    advance by a fixed number of elements per instance (an earlier site of the same instance: its dimension; a site
    outside the plate: 0; instance i of an earlier plate: that site's dimension)."""
    inside = {s.addr: (l, i) for l, col in enumerate(group) for i, s in enumerate(col)}
    res = []

    def source(addr, l, i):
        """-> (device addr, element offset) of a source address as instance i of kernel site l sees it, or None"""
        if addr in inside:
            ls, isrc = inside[addr]
            if ls >= l or isrc != i:
                return None
            return ("@plate", names[ls]), i * group[ls][0].dim
        if addr in where:
            return where[addr]
        return addr, 0

    for l, col in enumerate(group):
        s0 = col[0]
        d = s0.dim
        if s0.kind == A.DIRICHLET:
            return None
        mode0 = modes.get(s0.addr)
        for s in col:
            if (s.kind != s0.kind or s.dim != d or s.ncat != s0.ncat or len(s.params) != len(s0.params) or modes.get(s.addr) != mode0
                    or (s.addr in sel) != (s0.addr in sel)):
                return None
        params = []
        for k, p0 in enumerate(s0.params):
            ps = [s.params[k] for s in col]
            if any(q.op != p0.op or q.xf != p0.xf or bool(q.terms) != bool(p0.terms) for q in ps):
                return None
            if p0.op == A.P_EXPR:
                # every instance's expression with its leaves renamed to the DEVICE sites (an earlier site of the same instance, a site
                # outside the plate, an instance of an earlier plate); one node list with linear strides, or the plate stays unrolled
                from . import expr as E
                inst_outs = []
                for i, q in enumerate(ps):
                    if len(q.outs) != len(p0.outs):
                        return None
                    bad = []

                    def ren(a_, e_, i=i, bad=bad):
                        r = source(a_, l, i)
                        if r is None:
                            bad.append(a_)
                            return None
                        return E.value(r[0], e_ + r[1])
                    outs_i = E.rewrite_leaves(q.outs, ren)
                    if bad:
                        return None
                    inst_outs.append(tuple(outs_i))
                try:
                    fake = {}
                    E.lower_plate(inst_outs, lambda a_, e_: ("slot", fake.setdefault(a_, 100000 * (len(fake) + 1)) + e_), None)
                except (E.IrregularPlate, E.ExprTooLarge):
                    return None
                params.append(Param(A.P_EXPR, outs=inst_outs[0], xf=p0.xf, inst_outs=inst_outs))
                continue
            d_elem, src, elem0 = 0, None, 0
            if p0.op != A.P_CONST and not p0.terms:
                rs = [source(q.src, l, i) for i, q in enumerate(ps)]
                if any(r is None for r in rs) or any(r[0] != rs[0][0] for r in rs):
                    return None
                el = [r[1] + q.src_elem for r, q in zip(rs, ps)]
                d_elem = el[1] - el[0]
                if any(el[i] != el[0] + i * d_elem for i in range(n)) or any(q.length != p0.length for q in ps):
                    return None
                src, elem0 = rs[0][0], el[0]
            if p0.op == A.P_CONST:
                vs = [q.values for q in ps]
                if any(v.size != p0.values.size for v in vs):
                    return None
                params.append(Param.const(p0.values, xf=p0.xf) if _same(vs) else Param(A.P_CONST, values=np.stack(vs), xf=p0.xf, inst_values=True))
            elif p0.op == A.P_VALUE:
                params.append(Param(A.P_VALUE, src=src, src_elem=elem0, length=p0.length, xf=p0.xf, d_elem=d_elem))
            elif p0.op == A.P_GATHER:
                vs = [q.values for q in ps]
                if any(v.shape != p0.values.shape for v in vs):
                    return None
                share = _same(vs)
                params.append(Param(A.P_GATHER, values=p0.values if share else np.stack(vs), src=src, src_elem=elem0, xf=p0.xf,
                                    d_elem=d_elem, inst_values=not share))
            elif p0.op == A.P_VGATHER:
                # the indexed choice lives outside this plate and is the same for every instance; the index advances like a gather's
                if any(q.vsrc != p0.vsrc or q.vsrc_elem != p0.vsrc_elem or q.vn != p0.vn or q.vlen != p0.vlen for q in ps) or p0.vsrc in inside:
                    return None
                vs_, ve_ = where.get(p0.vsrc, (p0.vsrc, 0))
                params.append(Param(A.P_VGATHER, src=src, src_elem=elem0, xf=p0.xf, d_elem=d_elem, vsrc=vs_, vsrc_elem=p0.vsrc_elem + ve_,
                                    vn=p0.vn, vlen=p0.vlen))
            elif p0.op == A.P_AFFINE and not p0.terms:
                if any(q.matrix.shape != p0.matrix.shape or q.values.size != p0.values.size for q in ps):
                    return None
                sb, sm = _same([q.values for q in ps]), _same([q.matrix for q in ps])
                params.append(Param(A.P_AFFINE, values=p0.values if sb else np.stack([q.values for q in ps]),
                                    matrix=p0.matrix if sm else np.stack([q.matrix for q in ps]), src=src, src_elem=elem0, xf=p0.xf,
                                    d_elem=d_elem, inst_values=not sb, inst_matrix=not sm))
            elif p0.op == A.P_AFFINE:
                # an affine form over several sites: every source outside this plate and the same in all instances
                t0 = [a_ for a_, _ in p0.terms]
                if any([a_ for a_, _ in q.terms] != t0 or q.values.size != p0.values.size for q in ps) or any(a_ in inside or a_ in where for a_ in t0):
                    return None
                if any(any(mq.shape != m0.shape for (_, mq), (_, m0) in zip(q.terms, p0.terms)) for q in ps):
                    return None
                sb = _same([q.values for q in ps])
                sm = all(_same([q.terms[t][1] for q in ps]) for t in range(len(t0)))
                terms = p0.terms if sm else [(a_, np.stack([q.terms[t][1] for q in ps])) for t, a_ in enumerate(t0)]
                params.append(Param(A.P_AFFINE, values=p0.values if sb else np.stack([q.values for q in ps]), src=t0[0], xf=p0.xf,
                                    terms=terms, inst_values=not sb, inst_matrix=not sm))
            else:
                return None
        ns = Site(("@plate", names[l]), s0.kind, params, d, s0.ncat, -1, 0, pid, n)
        ov = None
        if mode0 == A.MODE_OBS_TAB:
            if any(s.addr not in obs for s in col):
                return None
            ov = np.concatenate([np.broadcast_to(np.asarray(obs[s.addr], np.float32).ravel(), (d,)) for s in col])
        res.append((ns, mode0, ov, s0.addr in sel))
    return res


def _try_compact(group, names, n: int, modes: dict, obs: dict, sel: set, where: dict | None = None):
    """-> [(vector Site, mode or None, observed values or None, selected)] per kernel site, or None"""
    inside = {s.addr: (l, i) for l, col in enumerate(group) for i, s in enumerate(col)}
    where = where or {}
    res = []
    for l, col in enumerate(group):
        s0 = col[0]
        d = s0.dim
        if s0.kind in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS, A.DIRICHLET) or s0.ncat:
            return None
        if any(q.op == A.P_EXPR for s in col for q in s.params):
            return None                     # (expression blocks: instances stay unrolled or take the plate form)
        if any(q.op != A.P_CONST and (q.src in where or (q.terms and any(a_ in where for a_, _ in q.terms))) for s in col for q in s.params):
            return None                     # reads instances of an earlier plate: the plate form expresses that
        mode0 = modes.get(s0.addr)
        if mode0 == A.MODE_OBS_MASK:
            return None
        for s in col:
            if s.kind != s0.kind or s.dim != d or len(s.params) != len(s0.params) or modes.get(s.addr) != mode0 or (s.addr in sel) != (s0.addr in sel):
                return None
        params = []
        for k, p0 in enumerate(s0.params):
            ps = [s.params[k] for s in col]
            if any(q.op != p0.op or q.xf != p0.xf or bool(q.terms) for q in ps):
                return None
            if p0.op == A.P_CONST:
                if any(q.values.size not in (1, d) for q in ps):
                    return None
                if all(q.values.size == p0.values.size and np.array_equal(q.values, p0.values) for q in ps):
                    params.append(Param.const(p0.values, xf=p0.xf))
                else:
                    params.append(Param.const(np.concatenate([np.broadcast_to(q.values, (d,)) for q in ps]), xf=p0.xf))
            elif p0.op == A.P_VALUE:
                if p0.src in inside:          # an earlier site of the same instance, elementwise
                    ls, _ = inside[p0.src]
                    if ls >= l or group[ls][0].dim != d or p0.length != d or p0.src_elem != 0:
                        return None
                    if any(q.src != group[ls][i].addr or q.length != d or q.src_elem != 0 for i, q in enumerate(ps)):
                        return None
                    params.append(Param.value(("@plate", names[ls]), length=n * d, xf=p0.xf))
                else:                         # a site outside the plate: the same read in every instance
                    if any(q.src != p0.src or q.length != p0.length or q.src_elem != p0.src_elem for q in ps) or p0.length not in (1, d):
                        return None
                    params.append(Param.value(p0.src, length=p0.length, elem=p0.src_elem, xf=p0.xf))
            elif p0.op == A.P_AFFINE:
                if p0.src in inside or any(q.src != p0.src or q.src_elem != p0.src_elem or q.matrix.shape != p0.matrix.shape for q in ps):
                    return None
                if p0.matrix.shape[0] != d or any(q.values.size not in (1, d) for q in ps):
                    return None
                M = np.concatenate([q.matrix for q in ps], axis=0)
                b = np.concatenate([np.broadcast_to(q.values, (d,)) for q in ps])
                params.append(Param.affine(M, p0.src, bias=b, elem=p0.src_elem, xf=p0.xf))
            else:
                return None
        ns = Site(("@plate", names[l]), s0.kind, params, n * d, 0, -1, 0)
        ov = None
        if mode0 == A.MODE_OBS_TAB:
            if any(s.addr not in obs for s in col):
                return None
            ov = np.concatenate([np.broadcast_to(np.asarray(obs[s.addr], np.float32).ravel(), (d,)) for s in col])
        res.append((ns, mode0, ov, s0.addr in sel))
    return res


# ---------------------------------------------------------------------------------------------
# packing
# ---------------------------------------------------------------------------------------------
class PackedProgram:
    """A SiteList bound to per-site modes, packed into the C structs of ``include/gjx.h``.

    ``modes`` maps address -> MODE_*; ``obs`` maps address -> observed value for MODE_OBS_TAB
    sites (the Target's constraint, inference/sp.py:52-94).  Addresses absent from ``modes``
    are MODE_SAMPLE.  ``selected`` flags sites for HMC.
    """

    def __init__(self, sl: SiteList, modes: dict[str, int] | None = None,
                 obs: dict[str, Any] | None = None, selected: Sequence[str] = (),
                 rng_mode: int = A.RNG_FLAT, plates: bool | str = False,
                 proposal: Sequence = (), proposed_by: dict | None = None,
                 carried: Sequence = (), input_row_of: dict | None = None):
        """``carried``: GJX_MODE_INPUT sites flagged GJX_SITE_CARRIED (their gathered value is stored into their own rows: a value that
        travels with the particle through a filter's steps); ``input_row_of``: address -> the row an INPUT site reads (gjx_site.obs_off;
        default: the inputs' rows in site order, from 0) — with GJX_FILTER_ABSOLUTE_INPUTS the absolute row of the previous step's buffer.
        ``plates``: lower the instances of vmapped kernels to vector sites (compact_plates).  The LOGICAL view stays
        per instance — ``site_list``, ``slot_of`` and ``obs_off`` answer for the addresses ``(name, i)`` — while the
        device program (``c_sites``, ``n_sites``) holds one site per kernel site; per-site scores are then per plate,
        so callers that need one score per instance pack without it."""
        self.logical_site_list = sl
        self.logical_modes = dict(modes or {})
        self.plate_of: dict = {}
        obs = obs or {}
        if plates:
            # True: vector form where it applies, else the plate form (gjx_site.plate); "vector": vector form only; "hmc": the
            # packing of an HMC move — vector form for plates whose values are all in the table, the plate form otherwise
            try:
                sl, modes, obs, selected, self.plate_of = compact_plates(sl, dict(modes or {}), dict(obs), tuple(selected),
                                                                         draws_ok=int(rng_mode) == A.RNG_FLAT, tagged=plates in (True, "hmc"),
                                                                         vector_needs_table=plates == "hmc")
            except _Unrollable:
                sl, modes, self.plate_of = self.logical_site_list, self.logical_modes, {}
            if not self.plate_of:
                sl = self.logical_site_list
        self.site_list = sl
        self.modes = dict(modes or {})
        self.rng_mode = int(rng_mode)
        tab: list[np.ndarray] = []
        self._tab_parts = tab
        ntab = 0

        shared_mats: dict = {}

        def push(arr, align: int = 1, share: bool = False) -> int:
            """append to the float table; ``align`` (in floats): 16-byte alignment for what a kernel reads with b128 loads;
            ``share``: a large CONSTANT array (the matrix of an affine form: never rewritten by set_obs) that is already in the table
            is not stored again — two likelihood sites over the same design matrix read one copy"""
            nonlocal ntab
            a = np.asarray(arr, np.float32).ravel()
            if share and a.size >= 64:
                k_ = (a.size, align, a.tobytes())
                if k_ in shared_mats:
                    return shared_mats[k_]
            if align > 1 and ntab % align:
                pad = align - ntab % align
                tab.append(np.zeros(pad, np.float32))
                ntab += pad
            off = ntab
            tab.append(a)
            ntab += a.size
            if share and a.size >= 64:
                shared_mats[(a.size, align, a.tobytes())] = off
            return off

        n = len(sl.sites)
        self.c_sites = (A.GjxSite * max(n, 1))()
        self._uid = next(_UIDS)     # gjx_program.uid: names this site list for the library's per-program caches
        self.obs_off: dict[str, int] = {}
        self.slot_of: dict[str, int] = {}
        self._derived: list[tuple[int, Param, str]] = []  # CONST blocks computed from observed values
        # Sites constrained to one shared value (OBS_TAB) own no row of choices[][]: their value
        # lives in tab and later sites read it from there.
        n_slots = 0
        proposal, proposed_by, carried = set(proposal), dict(proposed_by or {}), set(carried)
        for s in sl.sites:
            if self.modes.get(s.addr, A.MODE_SAMPLE) == A.MODE_OBS_PROPOSED:
                q = proposed_by.get(s.addr)
                if q is None or q not in self.slot_of or self.slot_of[q] < 0 or sl[q].rows != s.rows:
                    raise ValueError(f"site {s.addr!r}: MODE_OBS_PROPOSED needs an earlier proposal site of the same size (proposed_by)")
                self.slot_of[s.addr] = self.slot_of[q]          # scored at the proposal's draw: the two sites share their rows
                continue
            if self.modes.get(s.addr, A.MODE_SAMPLE) == A.MODE_OBS_TAB:
                if s.addr not in obs:
                    raise MissingAddress(s.addr)
                v = np.asarray(obs[s.addr], np.float32).ravel()
                if v.size != s.rows:
                    v = np.broadcast_to(v, (s.rows,))
                self.obs_off[s.addr] = push(v, 4 if v.size >= 64 else 1)
                self.slot_of[s.addr] = -1
            else:
                self.slot_of[s.addr] = n_slots
                n_slots += s.rows
        # Mask(value, flag) per particle: one extra row of choices[][] per masked site (per instance of a plate) holds its flags
        self.flag_slot_of: dict[str, int] = {}
        for s in sl.sites:
            if self.modes.get(s.addr, A.MODE_SAMPLE) == A.MODE_OBS_MASK:
                self.flag_slot_of[s.addr] = n_slots
                n_slots += s.plate_n if s.plate else 1
        order = {s.addr: j for j, s in enumerate(sl.sites)}
        self.input_row: dict = {}
        self.n_input_rows = 0
        for j, s in enumerate(sl.sites):
            cs = self.c_sites[j]
            cs.kind, cs.dim, cs.slot, cs.ncat = s.kind, s.dim, self.slot_of[s.addr], s.ncat
            cs.mode = self.modes.get(s.addr, A.MODE_SAMPLE)
            cs.flags = ((A.SITE_HMC_SELECTED if s.addr in selected else 0) | (A.SITE_PROPOSAL if s.addr in proposal else 0)
                        | (A.SITE_CARRIED if s.addr in carried else 0))
            cs.scan = int(s.scan)
            cs.plate, cs.plate_n = int(s.plate), int(s.plate_n)
            cs.obs_off = self.flag_slot_of[s.addr] if s.addr in self.flag_slot_of else self.obs_off.get(s.addr, 0)
            if cs.mode == A.MODE_INPUT:          # row of the site in the INPUT rows of gjx_run_program_ex (in_rows): inputs in site order
                cs.obs_off = self.n_input_rows if not input_row_of or s.addr not in input_row_of else int(input_row_of[s.addr])
                self.input_row[s.addr] = int(cs.obs_off)
                self.n_input_rows += s.dim
            if s.plate:
                cs.d_obs = 1 if cs.mode == A.MODE_OBS_MASK else (s.dim if cs.mode == A.MODE_OBS_TAB else 0)
            rows = s.ncat if s.ncat else s.dim
            for k in range(len(s.params), A.MAX_PARAMS):   # unused parameter slots read tab[0]
                cs.p[k].op, cs.p[k].len, cs.p[k].off = A.P_CONST, 1, 0
            for k, p in enumerate(s.params):
                cp = cs.p[k]
                cp.op, cp.xf = p.op, p.xf
                if p.op != A.P_CONST:
                    for a_src in p.sources():
                        if a_src not in order or order[a_src] >= j:
                            raise ValueError(f"site {s.addr!r} reads {a_src!r} before it is traced")
                if p.op == A.P_EXPR:
                    self._pack_expr(cp, p, s, rows, push)
                    continue
                if p.op == A.P_VGATHER:
                    self._pack_vgather(cp, p, s)
                    continue
                if p.terms:
                    self._pack_affine_multi(cp, p, s, rows, push)
                    continue
                src_obs = p.op != A.P_CONST and self.slot_of[p.src] < 0
                ninst = s.plate_n if s.plate else 1
                if p.op == A.P_CONST:
                    cp.off, cp.len = push(p.values), int(p.values.size // (ninst if p.inst_values else 1))
                    cp.d_off = cp.len if p.inst_values else 0
                elif p.op == A.P_VALUE and src_obs:
                    cp.op = A.P_CONST  # read the observed value straight from tab
                    cp.off, cp.len = self.obs_off[p.src] + p.src_elem, int(p.length)
                    cp.d_off = int(p.d_elem)
                elif src_obs:
                    cp.op = A.P_CONST  # parameter is a function of observed data only: fold on the host (per instance in a plate)
                    if s.plate:
                        vals = np.stack([self._fold_observed(p, rows, i) for i in range(ninst)])
                        cp.off, cp.len, cp.d_off = push(vals), int(vals.shape[1]), int(vals.shape[1])
                    else:
                        vals = self._fold_observed(p, rows)
                        cp.off, cp.len = push(vals), int(vals.size)
                    self._derived.append((cp.off, p, s.addr))
                elif p.op == A.P_VALUE:
                    cp.slot, cp.len = self.slot_of[p.src] + p.src_elem, int(p.length)
                    cp.d_slot = int(p.d_elem)
                elif p.op == A.P_GATHER:
                    cp.slot = self.slot_of[p.src] + p.src_elem
                    cp.d_slot = int(p.d_elem)
                    tshape = p.values.shape[1:] if p.inst_values else p.values.shape
                    cp.n, cp.len = int(tshape[0]), int(tshape[1])
                    cp.off = push(p.values)
                    cp.d_off = cp.n * cp.len if p.inst_values else 0
                elif p.op == A.P_AFFINE:
                    mshape = p.matrix.shape[1:] if p.inst_matrix else p.matrix.shape
                    if mshape[0] != rows:
                        raise ValueError(f"affine matrix of {s.addr!r} has {mshape[0]} rows, site dim {rows}")
                    cp.slot, cp.n = self.slot_of[p.src] + p.src_elem, int(mshape[1])
                    cp.d_slot = int(p.d_elem)
                    cp.off, cp.len = push(p.values, 4 if p.values.size >= 64 else 1), int(p.values.size // (ninst if p.inst_values else 1))
                    cp.d_off = cp.len if p.inst_values else 0
                    cp.moff = push(p.matrix, 4 if mshape[1] % 4 == 0 else 1, share=True)      # (rows of a matrix with 4 k columns: b128 loads)
                    cp.d_moff = int(mshape[0] * mshape[1]) if p.inst_matrix else 0
                else:
                    raise ValueError(p.op)
        self.tab = np.concatenate(tab).astype(np.float32) if tab else np.zeros(1, np.float32)
        self.n_sites = n
        self.n_slots = n_slots
        if self.plate_of:
            # the logical view: instance i of a compacted kernel site lives at element offset i * d of the vector site
            self.device_site_list = self.site_list
            for addr, (daddr, off) in self.plate_of.items():
                self.slot_of[addr] = self.slot_of[daddr] + off if self.slot_of[daddr] >= 0 else -1
                if daddr in self.obs_off:
                    self.obs_off[addr] = self.obs_off[daddr] + off
                if daddr in self.flag_slot_of:      # a masked plate site: one flag row per instance
                    self.flag_slot_of[addr] = self.flag_slot_of[daddr] + off // max(self.device_site_list[daddr].dim, 1)
            for daddr in {d for d, _ in self.plate_of.values()}:
                self.flag_slot_of.pop(daddr, None)
            self.site_list = self.logical_site_list
            self.modes = self.logical_modes
        else:
            self.device_site_list = self.site_list
        self._dev = None  # (sites tensor, tab tensor)
        self._aux = None  # device floats of gjx_program_prepare (None: not prepared yet)
        self._aux_n = 0

    # -- observed values can be replaced in place (same program, new data) --
    def _obs_value(self, addr: str) -> np.ndarray:
        off = self.obs_off[addr]
        # (after __init__ `site_list` is the LOGICAL list; derived sources may be plate body sites, which only the device list names)
        dsl = getattr(self, "device_site_list", self.site_list)
        site = dsl[addr] if addr in dsl else self.site_list[addr]
        return self._tab_view()[off:off + site.rows]

    def _tab_view(self) -> np.ndarray:
        return self.tab if hasattr(self, "tab") else np.concatenate(self._tab_parts)

    def _pack_vgather(self, cp, p: Param, s: Site) -> None:
        """value(vsrc)[idx]: four cases by who owns storage.  A choice constrained to one value for every particle lives in the table:
        the indexed choice there -> an ordinary GATHER whose table IS the observed value (set_obs keeps working); the index there ->
        VGATHER with slot = -1 and the index read from the table; both there -> GATHER on the table index is not expressible, so the
        row is a CONST at the observed offsets only when not in a plate."""
        v_lat, i_lat = self.slot_of[p.vsrc] >= 0, self.slot_of[p.src] >= 0
        cp.n, cp.len = int(p.vn), int(p.vlen)
        if v_lat:
            cp.op = A.P_VGATHER
            cp.moff = self.slot_of[p.vsrc] + p.vsrc_elem
            if i_lat:
                cp.slot, cp.d_slot = self.slot_of[p.src] + p.src_elem, int(p.d_elem)
            else:
                cp.slot, cp.off, cp.d_off = -1, self.obs_off[p.src] + p.src_elem, int(p.d_elem)
        elif i_lat:
            cp.op = A.P_GATHER
            cp.off = self.obs_off[p.vsrc] + p.vsrc_elem
            cp.slot, cp.d_slot = self.slot_of[p.src] + p.src_elem, int(p.d_elem)
        else:
            if s.plate and p.d_elem:
                raise NotImplementedError("take(observed vector, observed index) inside a plate: constrain one of them per particle")
            idx = int(np.clip(int(self._obs_value(p.src)[p.src_elem]), 0, p.vn - 1))
            # the index is baked into the offset: a later set_obs on it could not move the row (set_obs refuses, see there)
            self._folded_index = getattr(self, "_folded_index", set()) | {p.src}
            cp.op = A.P_CONST
            cp.off, cp.len = self.obs_off[p.vsrc] + p.vsrc_elem + idx * p.vlen, int(p.vlen)

    def _pack_expr(self, cp, p: Param, s: Site, rows: int, push) -> None:
        """GJX_P_EXPR: the expression's nodes as a block of 6-float records in the table (include/gjx.h).  Latent sources are read
        from their rows; a source constrained to one shared value is read from ITS table entries (set_obs is seen without repacking)."""
        from . import expr as E
        if len(p.outs) not in (1, rows):
            raise ValueError(f"site {s.addr!r}: an expression parameter with {len(p.outs)} elements for an event of {rows}")

        def place(addr, elem):
            sl_ = self.slot_of[addr]
            return ("slot", sl_ + elem) if sl_ >= 0 else ("tab", self.obs_off[addr] + elem)
        if s.plate:                            # ONE node list for the plate's instances, strides in the nodes (include/gjx.h)
            nodes, n = E.lower_plate(p.inst_outs, place, push)
        else:
            nodes, n = E.lower(p.outs, place, push)
        cp.op, cp.off, cp.n, cp.len = A.P_EXPR, push(nodes), int(n), len(p.outs)

    def _pack_affine_multi(self, cp, p: Param, s: Site, rows: int, push) -> None:
        ninst = s.plate_n if s.plate else 1
        per_inst = bool(s.plate) and (p.inst_matrix or p.inst_values)
        def mat(m, i):
            return m[i] if p.inst_matrix else m
        latent = [(a, m) for a, m in p.terms if self.slot_of[a] >= 0]
        for a, m in p.terms:
            m0 = mat(m, 0)
            if m0.shape[0] != rows or m0.shape[1] != self.site_list[a].dim:
                raise ValueError(f"affine term of {s.addr!r} on {a!r} has shape {m0.shape}, want ({rows}, {self.site_list[a].dim})")
        observed = len(latent) < len(p.terms)
        per_inst = per_inst or (bool(s.plate) and observed and p.inst_matrix)
        insts = range(ninst) if per_inst else range(1)
        bias = np.stack([self._fold_observed(p, rows, i) for i in insts])
        if not latent:
            cp.op = A.P_CONST
            cp.off, cp.len = push(bias), int(bias.shape[1])
            cp.d_off = cp.len if per_inst else 0
        else:
            lo = min(self.slot_of[a] for a, _ in latent)
            hi = max(self.slot_of[a] + mat(m, 0).shape[1] for a, m in latent)
            dense = np.zeros((len(insts), rows, hi - lo), np.float32)
            for i in insts:
                for a, m in latent:
                    mi = mat(m, i)
                    dense[i, :, self.slot_of[a] - lo: self.slot_of[a] - lo + mi.shape[1]] += mi
            cp.slot, cp.n = lo, hi - lo
            cp.off, cp.len = push(bias), int(bias.shape[1])
            cp.moff = push(dense)
            if per_inst:
                cp.d_off, cp.d_moff = cp.len, rows * (hi - lo)
        if observed:
            self._derived.append((cp.off, p, s.addr))

    def _fold_observed(self, p: Param, rows: int, inst: int = 0) -> np.ndarray:
        """the part of a parameter that depends on observed data only, evaluated on the host (for instance `inst` of a plate)"""
        vals = p.values[inst] if p.inst_values else p.values
        if p.terms:
            bias = np.broadcast_to(vals, (rows,)).astype(np.float32).copy() if vals.size in (1, rows) else vals.astype(np.float32).copy()
            for a, m in p.terms:
                if self.slot_of[a] < 0:
                    mi = m[inst] if p.inst_matrix else m
                    bias = bias + mi @ self._obs_value(a)[: mi.shape[1]]
            return bias.astype(np.float32)
        src = self._obs_value(p.src)[p.src_elem + inst * p.d_elem:]
        if p.op == A.P_GATHER:
            idx = int(np.clip(int(src[0]), 0, vals.shape[0] - 1))
            return vals[idx].astype(np.float32).copy()
        if p.op == A.P_AFFINE:
            mi = p.matrix[inst] if p.inst_matrix else p.matrix
            n = mi.shape[1]
            bias = np.broadcast_to(vals, (rows,)) if vals.size in (1, rows) else vals
            return (bias + mi @ src[:n]).astype(np.float32)
        raise ValueError(p.op)

    def set_obs(self, addr: str, value) -> None:
        """Replace an observed value in place (same program, new data)."""
        if addr in getattr(self, "_folded_index", ()):
            raise NotImplementedError(f"set_obs({addr!r}): this observed index picks a row of an observed vector and was folded into the program "
                                      "when it was packed (take(observed vector, observed index)); build the program again with the new index")
        off = self.obs_off[addr]
        v = np.asarray(value, np.float32).ravel()
        self.tab[off:off + v.size] = v
        dirty = [(off, v.size)]
        daddr = self.plate_of.get(addr, (addr, 0))[0]
        for doff, p, site_addr in self._derived:
            if p.src in (addr, daddr) or (p.terms and any(a in (addr, daddr) for a, _ in p.terms)):
                s = self.device_site_list[site_addr]
                r_ = s.ncat if s.ncat else s.dim
                # the branch __init__ took: a plate site's folded parameter is stacked per instance; an affine form over several sites
                # (`terms`) only when its matrix or bias differs by instance (_pack_affine_multi)
                per = bool(s.plate) and (p.inst_matrix or p.inst_values) if p.terms else bool(s.plate)
                vals = np.stack([self._fold_observed(p, r_, i) for i in range(s.plate_n)]).ravel() if per else self._fold_observed(p, r_)
                self.tab[doff:doff + vals.size] = vals
                dirty.append((doff, vals.size))
        if self._dev is not None:
            import torch
            for o, n in dirty:
                self._dev[1][o:o + n] = torch.from_numpy(self.tab[o:o + n].copy()).to(self._dev[1].device)
            self._aux = None                             # derived constants follow the table

    def sites_bytes(self) -> bytes:
        return bytes(self.c_sites)[: self.n_sites * C.sizeof(A.GjxSite)]

    def c_program(self, device=None) -> A.GjxProgram:
        """Struct for a call.  With ``device`` (a torch device) the *_dev pointers are filled."""
        p = A.GjxProgram()
        p.n_sites, p.n_slots, p.n_tab, p.rng_mode = self.n_sites, self.n_slots, int(self.tab.size), self.rng_mode
        p.sites = C.cast(self.c_sites, C.c_void_p)
        p.uid = self._uid
        p.tab = self.tab.ctypes.data_as(C.c_void_p)
        if device is not None:
            import torch
            if self._dev is None or self._dev[1].device != torch.device(device):
                sb = np.frombuffer(self.sites_bytes() or b"\0" * 96, dtype=np.uint8).copy()
                self._dev = (torch.from_numpy(sb).to(device), torch.from_numpy(self.tab.copy()).to(device))
                self._aux = None
            p.sites_dev = self._dev[0].data_ptr()
            p.tab_dev = self._dev[1].data_ptr()
            if self._dev[1].is_cuda:
                if self._aux is None:                    # constants derived from tab, computed once per upload
                    from ._lib import check, load
                    lib = load()
                    n = int(lib.gjx_program_aux_floats(C.byref(p)))
                    self._aux = torch.empty(max(n, 1), dtype=torch.float32, device=device) if n >= 0 else False
                    if n > 0:
                        check(lib.gjx_program_prepare(C.byref(p), C.c_void_p(self._aux.data_ptr()), n,
                                                      C.c_void_p(torch.cuda.current_stream(self._aux.device).cuda_stream)),
                              "gjx_program_prepare")
                    self._aux_n = max(n, 0)
                if self._aux is not False and self._aux_n > 0:
                    p.aux_dev, p.n_aux = self._aux.data_ptr(), self._aux_n
        else:
            p.sites_dev = p.sites
            p.tab_dev = p.tab
        return p
