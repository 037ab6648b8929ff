"""``@gen`` modelling language, primitive distributions and traces — host side.

The reference stages a Python function to a Jaxpr and re-interprets it for every GFI method
(generative_functions/static.py:725-1049).  Here the function body is run ONCE per argument tuple
with symbolic values; every ``dist(*params) @ "addr"`` appends a site to a SiteList
(program.py) and the GFI methods become launches of the HIP site interpreter / fused kernels
with per-site modes:

    simulate   all sites SAMPLE                                   static.py:254-278, 787-793
    assess     all sites constrained, MissingAddress otherwise    static.py:297-321, 983-989
    generate / importance   constrained sites OBS, rest SAMPLE    static.py:340-399, 795-810
    update     all sites keep/replace their value, w = new - old  static.py:827-865

Batched execution (the reference's ``jax.vmap`` over particles or chains) is the native mode: a
Trace holds choices[n_slots][K] on the device; K == 1 is the un-vmapped case.
"""
from __future__ import annotations

import warnings
from typing import Any, Callable, Sequence

import numpy as np

from . import _abi as A
from . import expr as E
from .core import ChoiceMap, Key, Masked, Selection, _VALUE, split
from .program import MissingAddress, PackedProgram, Param, SiteList

__all__ = [
    "gen", "StaticGenerativeFunction", "Trace", "Distribution", "take", "where", "cond", "const", "exp",
    "softplus", "sigmoid", "tanh", "log", "sqrt", "square", "sin", "cos", "log1p", "maximum", "minimum", "dot", "normal", "flip", "bernoulli", "beta", "categorical", "uniform", "mv_normal_diag",
    "exponential", "half_normal", "laplace", "log_normal", "cauchy", "gamma", "Marginal", "ScanCombinator",
]


# ---------------------------------------------------------------------------------------------
# symbolic values seen by the model body while it is traced
# ---------------------------------------------------------------------------------------------
class NotSupportedInModelBody(TypeError):
    """The model body computed something the parameter-expression forms cannot express."""


class Sym:
    __array_ufunc__ = None  # numpy binary operators defer to our reflected methods
    __array_priority__ = 1000

    def as_param(self) -> Param:  # pragma: no cover - abstract
        raise NotImplementedError

    # Arithmetic tries the CLOSED forms first — affine maps of earlier choices, which the engines have fast paths for (Affine) — and
    # falls back to a general elementwise expression (Expr -> GJX_P_EXPR: any computation between sites, static.py:383-399)
    def _affine(self) -> "Affine":
        raise NotSupportedInModelBody(f"arithmetic on {type(self).__name__} is not supported in a model body")

    def _expr(self) -> "Expr":
        raise NotSupportedInModelBody(f"arithmetic on {type(self).__name__} is not supported in a model body")

    def _try(self, closed, general):
        try:
            return closed()
        except NotSupportedInModelBody:
            return general()

    def __add__(self, o): return self._try(lambda: self._affine().add(o), lambda: _ebin("add", self, o))
    def __radd__(self, o): return self._try(lambda: self._affine().add(o), lambda: _ebin("add", o, self))
    def __sub__(self, o): return self._try(lambda: self._affine().add(_neg(o)), lambda: _ebin("sub", self, o))
    def __rsub__(self, o): return self._try(lambda: self._affine().scale(-1.0).add(o), lambda: _ebin("sub", o, self))
    def __mul__(self, o): return self._try(lambda: self._affine().mul(o), lambda: _ebin("mul", self, o))
    def __rmul__(self, o): return self._try(lambda: self._affine().mul(o), lambda: _ebin("mul", o, self))

    def __truediv__(self, o):
        if isinstance(o, Sym):
            return _ebin("div", self, o)
        return self._try(lambda: self._affine().mul(1.0 / np.asarray(o, np.float64)), lambda: _ebin("div", self, o))

    def __rtruediv__(self, o): return _ebin("div", o, self)
    def __neg__(self): return self._try(lambda: self._affine().scale(-1.0), lambda: _eun("neg", self))
    def __rmatmul__(self, o): return self._try(lambda: self._affine().lmatmul(o), lambda: self._expr().lmatmul(o))

    def __matmul__(self, o):                # x @ w (a dot product); x @ W == W.T @ x
        o = np.asarray(o, np.float64)
        return self.__rmatmul__(o if o.ndim == 1 else o.T)

    def __pow__(self, k):
        k = float(k)
        if k == 1.0:
            return self
        if k == 2.0:
            return _eun("square", self)
        if k == 0.5:
            return _eun("sqrt", self)
        if k == -1.0:
            return _eun("recip", self)
        if k == 3.0:
            return _ebin("mul", _eun("square", self), self)
        return _eun("exp", _ebin("mul", _eun("log", self), k))

    # comparisons give 0 / 1 values (for where(...)); no gradient flows through them (jax.grad of a comparison)
    def __gt__(self, o): return _ebin("gt", self, o)
    def __lt__(self, o): return _ebin("gt", o, self)
    def __ge__(self, o): return _ebin("sub", 1.0, _ebin("gt", o, self))
    def __le__(self, o): return _ebin("sub", 1.0, _ebin("gt", self, o))


def _neg(o):
    return -o if isinstance(o, Sym) else -np.asarray(o, np.float64)


def _to_expr(x) -> "Expr":
    if isinstance(x, Sym):
        return x._expr()
    return Expr([E.const(v) for v in np.atleast_1d(np.asarray(x, np.float64)).ravel()])


def _ebin(op: str, a, b) -> "Expr":
    """elementwise binary op on general expressions (length-1 operands broadcast)"""
    a, b = _to_expr(a), _to_expr(b)
    n = max(a.dim, b.dim)
    if a.dim not in (1, n) or b.dim not in (1, n):
        raise NotSupportedInModelBody(f"cannot broadcast expressions of {a.dim} and {b.dim} elements")
    f = {"add": E.add, "sub": E.sub}.get(op, lambda x, y: E.binary(op, x, y))
    return Expr([f(a.elems[i % a.dim], b.elems[i % b.dim]) for i in range(n)])


def _eun(op: str, a) -> "Expr":
    a = _to_expr(a)
    return Expr([E.unary(op, e) for e in a.elems])


class Expr(Sym):
    """A vector of general elementwise expressions of earlier choices and constants (expr.py): products of choices, arithmetic
    after transforms, tanh / log / sqrt / ..., comparisons and where, matrix products of such vectors.  As a distribution parameter
    it lowers to GJX_P_EXPR — emitted inline by the generated kernels, evaluated by the interpreters and the oracle, differentiated
    in reverse by the HMC engines; as a return value it is evaluated from the trace."""

    def __init__(self, elems):
        self.elems = list(elems)
        self.dim = len(self.elems)

    def _expr(self) -> "Expr":
        return self

    def __getitem__(self, i):
        e = np.asarray(range(self.dim))[i]
        return Expr([self.elems[int(j)] for j in np.atleast_1d(e)])

    def lmatmul(self, o) -> "Expr":
        o = np.atleast_2d(np.asarray(o, np.float64))
        if o.shape[1] != self.dim:
            raise NotSupportedInModelBody(f"matrix of shape {o.shape} times an expression of {self.dim} elements")
        return Expr([E.lin(0.0, [(self.elems[c], o[r, c]) for c in range(self.dim)]) for r in range(o.shape[0])])

    def as_param(self) -> Param:
        if all(E.is_const(e) for e in self.elems):
            return Param.const([e[1] for e in self.elems])
        return Param.expr(self.elems)

    def __repr__(self):
        return f"<expression of {self.dim} element(s) over {E.sources(self.elems)!r}>"


class HostExpr(Sym):
    """An expression of several choices: fine as a RETURN value (evaluated from the trace on demand),
    not expressible as a distribution parameter."""

    def __init__(self, fn):
        self.fn = fn

    def as_param(self):
        raise NotSupportedInModelBody("a distribution parameter may depend on ONE earlier site (plus constants)")

    def _bin(self, o, op):
        return HostExpr(lambda ev: op(ev(self), ev(o) if isinstance(o, Sym) else o))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    __radd__ = __add__
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    __rmul__ = __mul__
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __neg__(self): return HostExpr(lambda ev: -ev(self))


class Affine(Sym):
    """``sum_s M_s @ value(s) + b`` over one or more earlier sites s (M_s over ALL elements of s); optional outer
    transform.  One source lowers to a VALUE / AFFINE parameter on that site; several lower to one AFFINE
    parameter over the slot range that spans them (observed sources fold into the bias at pack time)."""

    def __init__(self, src, M=None, b=None, xf: int = A.XF_NONE):
        if isinstance(src, dict):                         # {addr: (SiteVal, M)}
            self.terms = {a: (sv, np.atleast_2d(np.asarray(m, np.float64))) for a, (sv, m) in src.items()}
        else:
            self.terms = {src.addr: (src, np.atleast_2d(np.asarray(M, np.float64)))}
        self.b, self.xf = np.atleast_1d(np.asarray(b, np.float64)), xf
        self.dim = next(iter(self.terms.values()))[1].shape[0] if self.terms else self.b.size

    # single-source views (most expressions)
    @property
    def src(self) -> "SiteVal":
        if len(self.terms) != 1:
            raise NotSupportedInModelBody("this expression depends on several sites")
        return next(iter(self.terms.values()))[0]

    @property
    def M(self) -> np.ndarray:
        if len(self.terms) != 1:
            raise NotSupportedInModelBody("this expression depends on several sites")
        return next(iter(self.terms.values()))[1]

    def _affine(self):
        if self.xf != A.XF_NONE:
            raise NotSupportedInModelBody("arithmetic after exp/softplus/sigmoid is not supported")
        return self

    def _map(self, f, b) -> "Affine":
        return Affine({a: (sv, f(m)) for a, (sv, m) in self.terms.items()}, b=b)

    def _bcast(self, m: int) -> "Affine":
        if self.dim == m:
            return self
        if self.dim != 1:
            raise NotSupportedInModelBody(f"cannot broadcast a length-{self.dim} expression to {m}")
        return self._map(lambda M: np.repeat(M, m, 0), np.repeat(self.b, m))

    def add(self, o):
        if isinstance(o, Sym):
            if isinstance(o, HostExpr):
                a0 = self
                return HostExpr(lambda ev: ev(a0) + ev(o))
            if isinstance(o, Expr):
                raise NotSupportedInModelBody("affine + general expression")            # (-> the general expression)
            o = o._affine()
            m = max(self.dim, o.dim)
            a, c = self._bcast(m), o._bcast(m)
            terms = dict(a.terms)
            for addr, (sv, M) in c.terms.items():
                terms[addr] = (sv, terms[addr][1] + M) if addr in terms else (sv, M)
            return Affine(terms, b=a.b + c.b)
        o = np.atleast_1d(np.asarray(o, np.float64)).ravel()
        a = self._bcast(max(self.dim, o.size))
        return Affine(a.terms, b=a.b + o)

    def scale(self, s: float):
        return self._map(lambda M: M * s, self.b * s)

    def _expr(self) -> Expr:
        rows = []
        for r in range(self.dim):
            terms = [(E.value(addr, c), M[r, c]) for addr, (sv, M) in self.terms.items() for c in range(M.shape[1]) if M[r, c] != 0.0]
            rows.append(E.lin(self.b[r], terms))
        op = {A.XF_EXP: "exp", A.XF_SOFTPLUS: "softplus", A.XF_SIGMOID: "sigmoid"}.get(self.xf)
        return Expr([E.unary(op, e) for e in rows] if op else rows)

    def mul(self, o):
        if isinstance(o, Sym):
            raise NotSupportedInModelBody("a product of choices is not affine")        # (Sym.__mul__ then takes the general expression)
        o = np.atleast_1d(np.asarray(o, np.float64)).ravel()
        a = self._bcast(max(self.dim, o.size))
        return a._map(lambda M: M * o[:, None], a.b * o)

    def lmatmul(self, o):
        o = np.atleast_2d(np.asarray(o, np.float64))
        return self._map(lambda M: o @ M, o @ self.b)

    def with_xf(self, xf: int) -> "Affine":
        self._affine()
        return Affine(self.terms, b=self.b, xf=xf)

    def __getitem__(self, i):
        self._affine()
        return self._map(lambda M: M[i], self.b[i])

    def as_param(self) -> Param:
        if len(self.terms) > 1:
            return Param.affine_multi([(a, m) for a, (sv, m) in self.terms.items()], bias=self.b, xf=self.xf)
        M, b = self.M, self.b
        n = self.src.dim
        # identity on a prefix (or a single broadcast element) is a plain VALUE read
        if not b.any():
            if M.shape == (n, n) and np.array_equal(M, np.eye(n)):
                return Param.value(self.src.addr, length=n, xf=self.xf)
            nz = np.nonzero(M.any(axis=0))[0]
            if M.shape[0] == 1 and nz.size == 1 and M[0, nz[0]] == 1.0:
                return Param.value(self.src.addr, length=1, elem=int(nz[0]), xf=self.xf)
        nz = np.nonzero(M.any(axis=0))[0]
        c0, c1 = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 1)
        return Param.affine(M[:, c0:c1], self.src.addr, bias=b, elem=c0, xf=self.xf)


class SiteVal(Sym):
    """The value of a traced site (what ``dist(...) @ addr`` returns inside a model body)."""

    def __init__(self, addr, dim: int, kind: int):
        self.addr, self.dim, self.kind = addr, dim, kind

    def _affine(self) -> Affine:
        if self.kind in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS):
            raise NotSupportedInModelBody("arithmetic on a categorical index; use take(table, idx)")
        return Affine(self, np.eye(self.dim), np.zeros(self.dim))

    def _expr(self) -> Expr:
        return Expr([E.value(self.addr, e) for e in range(self.dim)])

    def __getitem__(self, i):
        if isinstance(i, SiteVal):          # mu[z]: a row of this choice picked by a discrete choice
            return take(self, i)
        return self._try(lambda: self._affine()[i], lambda: self._expr()[i])

    def as_param(self) -> Param:
        return Param.value(self.addr, length=self.dim)

    def __repr__(self):
        return f"<choice {self.addr!r}>"


class Gather(Sym):
    def __init__(self, table: np.ndarray, idx: SiteVal, xf: int = A.XF_NONE):
        t = np.asarray(table, np.float32)
        self.table, self.idx, self.xf = (t[:, None] if t.ndim == 1 else t.reshape(t.shape[0], -1)), idx, xf
        self.dim = self.table.shape[1]

    def as_param(self) -> Param:
        return Param.gather(self.table, self.idx.addr, xf=self.xf)


class VGather(Sym):
    """``vec[idx]``: one of the rows of an EARLIER vector-valued choice picked by a discrete choice (GJX_P_VGATHER) — the component
    mean ``mu[z]`` of a mixture with latent means.  ``rows``: number of rows the choice's elements are read as (default: one
    element per row)."""

    def __init__(self, vec: SiteVal, idx: SiteVal, rows: int | None = None, xf: int = A.XF_NONE):
        n = int(rows) if rows else vec.dim
        if vec.dim % n:
            raise NotSupportedInModelBody(f"take: {vec.dim} elements do not split into {n} rows")
        self.vec, self.idx, self.n, self.dim, self.xf = vec, idx, n, vec.dim // n, xf

    def with_xf(self, xf: int) -> "VGather":
        return VGather(self.vec, self.idx, self.n, xf)

    def as_param(self) -> Param:
        return Param.vgather(self.vec.addr, self.n, self.idx.addr, vlen=self.dim, xf=self.xf)


class const:
    """Wrap a constant table so that it can be indexed by a random choice: ``const(means)[idx]``."""

    def __init__(self, table):
        self.table = np.asarray(table, np.float32)

    def __getitem__(self, idx):
        return take(self.table, idx) if isinstance(idx, Sym) else self.table[idx]


def take(table, idx, rows: int | None = None):
    """``table[idx]`` for a categorical / boolean choice ``idx``: rows of a constant table, or — ``table`` an earlier vector-valued
    choice — its element (``rows``: its row of ``dim / rows`` elements) number ``idx``."""
    if not isinstance(idx, SiteVal):
        raise NotSupportedInModelBody("take(table, idx): idx must be a random choice")
    if isinstance(table, SiteVal):
        return VGather(table, idx, rows)
    return Gather(np.asarray(table, np.float32), idx)


def where(flag, if_true, if_false):
    """``jnp.where(flag, a, b)`` / ``jax.lax.cond(flag, lambda: a, lambda: b)``: on constants picked by a boolean choice a table
    gather (the closed form); with a computed condition (``x > 0``) or branches that depend on choices a general expression"""
    if isinstance(flag, SiteVal) and not isinstance(if_true, Sym) and not isinstance(if_false, Sym):
        a, b = np.atleast_1d(np.asarray(if_true, np.float32)), np.atleast_1d(np.asarray(if_false, np.float32))
        return take(np.stack([np.broadcast_to(b, np.broadcast(a, b).shape), np.broadcast_to(a, np.broadcast(a, b).shape)]), flag)
    if not isinstance(flag, Sym):
        c = np.asarray(flag)
        if c.ndim == 0:
            return if_true if bool(c) else if_false
    c, a, b = _to_expr(flag), _to_expr(if_true), _to_expr(if_false)
    n = max(c.dim, a.dim, b.dim)
    if any(x.dim not in (1, n) for x in (c, a, b)):
        raise NotSupportedInModelBody("where: operands do not broadcast")
    return Expr([E.where(c.elems[i % c.dim], a.elems[i % a.dim], b.elems[i % b.dim]) for i in range(n)])


def cond(flag, true_fn, false_fn):
    t = true_fn() if callable(true_fn) else true_fn
    f = false_fn() if callable(false_fn) else false_fn
    return where(flag, t, f)


def array(items) -> "Affine | Expr | np.ndarray":
    """``jnp.array([0.0, y])`` inside a model body: a vector whose entries are numbers and scalar expressions of earlier sites
    (affine entries keep the closed form; anything else makes it a general expression)"""
    items = list(items)
    if not any(isinstance(it, Sym) for it in items):
        return np.asarray(items, np.float32)
    try:
        n = len(items)
        terms: dict = {}
        b = np.zeros(n, np.float64)
        for r, it in enumerate(items):
            if not isinstance(it, Sym):
                b[r] = float(it)
                continue
            a = it._affine()
            if a.dim != 1:
                raise NotSupportedInModelBody("array([...]): entries must be scalars")
            b[r] = a.b[0]
            for addr, (sv, M) in a.terms.items():
                if addr not in terms:
                    terms[addr] = (sv, np.zeros((n, sv.dim), np.float64))
                terms[addr][1][r] = M[0]
        return Affine(terms, b=b)
    except NotSupportedInModelBody:
        es = []
        for it in items:
            e = _to_expr(it)
            if e.dim != 1:
                raise NotSupportedInModelBody("array([...]): entries must be scalars")
            es.append(e.elems[0])
        return Expr(es)


def _xf(x, code: int, fn: Callable, name: str):
    if isinstance(x, Gather):
        if x.xf != A.XF_NONE:
            raise NotSupportedInModelBody("nested transforms of a table gather")
        return Gather(x.table, x.idx, code)
    if isinstance(x, VGather):
        if x.xf != A.XF_NONE:
            raise NotSupportedInModelBody("nested transforms of a row gather")
        return x.with_xf(code)
    if isinstance(x, Sym):
        # one transform on top of an affine map is a closed form (GJX_XF_*); anything else a general expression
        return x._try(lambda: x._affine().with_xf(code), lambda: _eun(name, x))
    return fn(np.asarray(x, np.float64))


def _fn(name: str, fn: Callable):
    def f(x):
        return _eun(name, x) if isinstance(x, Sym) else fn(np.asarray(x, np.float64))
    f.__name__ = name
    f.__doc__ = f"``jnp.{name}`` of numbers or of expressions of earlier choices (a general expression: GJX_P_EXPR)"
    return f


def exp(x): return _xf(x, A.XF_EXP, np.exp, "exp")
def softplus(x): return _xf(x, A.XF_SOFTPLUS, lambda v: np.logaddexp(0.0, v), "softplus")
def sigmoid(x): return _xf(x, A.XF_SIGMOID, lambda v: 1.0 / (1.0 + np.exp(-v)), "sigmoid")


tanh = _fn("tanh", np.tanh)
log = _fn("log", np.log)
sqrt = _fn("sqrt", np.sqrt)
square = _fn("square", np.square)
sin = _fn("sin", np.sin)
cos = _fn("cos", np.cos)
log1p = _fn("log1p", np.log1p)
abs_ = _fn("abs", np.abs)


def maximum(a, b):
    return _ebin("max", a, b) if isinstance(a, Sym) or isinstance(b, Sym) else np.maximum(a, b)


def minimum(a, b):
    return _ebin("min", a, b) if isinstance(a, Sym) or isinstance(b, Sym) else np.minimum(a, b)


def dot(a, b):
    """``jnp.dot`` of a constant vector / matrix and a vector of expressions (either order), or of two expression vectors"""
    if isinstance(a, Sym) and isinstance(b, Sym):
        ea, eb = _to_expr(a), _to_expr(b)
        if ea.dim != eb.dim:
            raise NotSupportedInModelBody("dot: lengths differ")
        acc = E.binary("mul", ea.elems[0], eb.elems[0])
        for x, y in zip(ea.elems[1:], eb.elems[1:]):
            acc = E.add(acc, E.binary("mul", x, y))
        return Expr([acc])
    if isinstance(b, Sym):
        return np.asarray(a, np.float64) @ b
    if isinstance(a, Sym):
        return a @ np.asarray(b, np.float64)
    return np.dot(a, b)


# ---------------------------------------------------------------------------------------------
# tracer
# ---------------------------------------------------------------------------------------------
class _Tracer:
    stack: list["_Tracer"] = []

    def __init__(self):
        self.sites = SiteList()
        self.step = None   # inside a scan / vmap: the iteration index, appended to every address as (name, step); nested
                           # combinators: the tuple of indices, outermost first
        self.in_scan = False
        self.scan = 0      # inside a scan: gjx_site.scan tag of the current step (chained step keys, gjx.h)
        self.n_scans = 0
        self.prefix = ()   # inside `callee(...) @ "addr"`: the path of enclosing call addresses

    def __enter__(self):
        _Tracer.stack.append(self)
        return self

    def __exit__(self, *exc):
        _Tracer.stack.pop()

    @staticmethod
    def current() -> "_Tracer":
        if not _Tracer.stack:
            raise RuntimeError("`dist(...) @ addr` is only meaningful inside a @gen function")
        return _Tracer.stack[-1]


class DistCall:
    def __init__(self, dist: "Distribution", kind: int, params: list, dim):
        self.dist, self.kind, self.params, self.dim = dist, kind, params, dim

    def __matmul__(self, addr):
        t = _Tracer.current()
        path = t.prefix + (tuple(addr) if isinstance(addr, tuple) else (addr,))
        addr = path[0] if len(path) == 1 else (path if path else _VALUE)
        if t.step is not None:
            addr = (addr, t.step)
        site = t.sites.add(addr, self.kind, self.params, self.dim)
        site.scan = t.scan
        return SiteVal(addr, site.dim, self.kind)


def _distcall_api(cls):
    """``genjax.normal(0.0, 1.0).simulate(key, ())``: a distribution applied to its parameters is itself a generative
    function without arguments (distribution.py:108-147)."""
    def bound(self):
        return self.dist._as_gen_with_args(), tuple(self.params)
    cls.simulate = lambda self, key, args=(), K=None: bound(self)[0].simulate(key, bound(self)[1], K)
    cls.assess = lambda self, chm, args=(): self.dist.assess(chm, bound(self)[1])
    cls.importance = cls.generate = lambda self, key, chm, args=(), K=None: bound(self)[0].generate(key, chm, bound(self)[1], K)
    return cls


_distcall_api(DistCall)


class GenCall:
    """``callee(*args)`` inside a model body; ``@ "addr"`` inlines the callee's sites under that address
    (static.py:340-399: the handler recurses into the callee with the sub-choicemap at ``addr``)."""

    def __init__(self, inline: Callable):
        self._inline = inline

    def __matmul__(self, addr):
        t = _Tracer.current()
        old = t.prefix
        t.prefix = old + (tuple(addr) if isinstance(addr, tuple) else (addr,))
        try:
            return self._inline(t)
        finally:
            t.prefix = old


class GenClosure(GenCall):
    """``model(*args, **kwargs)``: inlines under an address inside a body (GenCall), and outside one behaves as the generative
    function with its arguments filled in — ``gfc.simulate(key, ())``, ``gfc.importance(key, chm, ())``, ``gfc.assess(chm, ())``,
    ``gfc(key)`` -> return value, keyword arguments overridable per call (tests/generative_functions/test_static_gen_fn.py:824-885)."""

    def __init__(self, gen_fn, args: tuple, kwargs: dict):
        super().__init__(lambda t: gen_fn.source(*args, **kwargs))
        self.gen_fn, self.args, self.kwargs = gen_fn, tuple(args), dict(kwargs)

    def _bound(self, **over):
        src, a, kw = self.gen_fn.source, self.args, {**self.kwargs, **over}
        g = StaticGenerativeFunction(lambda: src(*a, **kw))
        g.__name__ = getattr(self.gen_fn, "__name__", "gen_fn")
        return g

    def simulate(self, key, args=()):
        return self._bound().simulate(key, ())

    def importance(self, key, constraint, args=()):
        return self._bound().importance(key, constraint, ())

    generate = importance

    def assess(self, chm, args=()):
        return self._bound().assess(chm, ())

    def __call__(self, key, **over):
        return self._bound(**over).simulate(key, ()).get_retval()


# ---------------------------------------------------------------------------------------------
# traces
# ---------------------------------------------------------------------------------------------
def _to_device_value(kind: int, t):
    import torch
    if kind in (A.FLIP, A.BERNOULLI_LOGITS):
        return t != 0
    if kind in (A.CATEGORICAL_LOGITS, A.CATEGORICAL_PROBS):
        return t.to(torch.int32)
    return t


class Trace:
    """Batched execution record (static.py:80-119 StaticTrace, distribution.py:59-82).

    ``choices`` f32[n_slots][K] and ``score`` f32[K] live on the device; constrained-to-a-shared-value
    sites (the Target's constraint) are kept once in ``shared``.  ``batched`` False means the trace was
    produced by an un-vmapped call and accessors drop the particle axis.
    """

    def __init__(self, gen_fn, args, prog: PackedProgram, choices, score, shared: dict, batched: bool,
                 retval_sym=None):
        self.gen_fn, self.args, self.prog = gen_fn, args, prog
        self.choices, self.score, self.shared, self.batched = choices, score, dict(shared), batched
        self.retval_sym = retval_sym

    @property
    def K(self) -> int:
        return int(self.score.shape[0])

    def get_gen_fn(self): return self.gen_fn
    def get_args(self): return self.args

    def get_score(self):
        return self.score if self.batched else self.score[0]

    def _site_value(self, addr):
        import torch
        s = self.prog.site_list[addr]
        slot = self.prog.slot_of[addr]
        if slot < 0:
            v = torch.as_tensor(np.asarray(self.shared[addr], np.float32).reshape(-1)[: s.dim].copy(), device=self.score.device)
            v = v[0] if s.dim == 1 else v
            if self.batched:
                v = v.expand(self.K, *v.shape)
            return _to_device_value(s.kind, v)
        rows = self.choices[slot:slot + s.dim]                      # [dim][K]
        v = rows[0] if s.dim == 1 else rows.t()                      # [K] or [K][dim]
        if not self.batched:
            v = v[0]
        return _to_device_value(s.kind, v)

    def get_choices(self) -> ChoiceMap:
        import torch
        d = {s.addr: self._site_value(s.addr) for s in self.prog.site_list.sites}
        # scan sites ("x", t): also expose the stacked sequence under "x" (leading axes: particles, then steps)
        seqs: dict = {}
        nested: dict = {}
        for s in self.prog.site_list.sites:
            if isinstance(s.addr, tuple) and len(s.addr) == 2 and isinstance(s.addr[1], int):
                seqs.setdefault(s.addr[0], []).append(d[s.addr])
            elif isinstance(s.addr, tuple) and len(s.addr) == 2 and isinstance(s.addr[1], tuple):
                nested.setdefault(s.addr[0], {})[s.addr[1]] = d[s.addr]
        lead = 1 if self.batched else 0
        for name, vals in seqs.items():
            d[name] = torch.stack(vals, dim=lead)
        for name, by_idx in nested.items():
            # nested combinators: the full grid of indices (a rectangular nest) stacks to [particles?][n0][n1](...)
            shape = tuple(max(ix[k] for ix in by_idx) + 1 for k in range(len(next(iter(by_idx)))))
            if len(by_idx) == int(np.prod(shape)):
                flat = torch.stack([by_idx[ix] for ix in sorted(by_idx)], dim=lead)
                d[name] = flat.reshape(flat.shape[:lead] + shape + flat.shape[lead + 1:])
        return ChoiceMap(d, lead)

    def get_retval(self):
        r = self.retval_sym
        import torch

        def ev(x):
            if isinstance(x, SiteVal):
                return self._site_value(x.addr)
            if isinstance(x, Affine):
                out = torch.as_tensor(x.b, dtype=torch.float32, device=self.score.device)
                for addr, (sv, Mx) in x.terms.items():
                    v = self._site_value(addr).float()
                    v = v[..., None] if sv.dim == 1 else v
                    out = out + v @ torch.as_tensor(Mx, dtype=torch.float32, device=v.device).t()
                out = out[..., 0] if x.dim == 1 else out
                return {A.XF_EXP: torch.exp, A.XF_SOFTPLUS: torch.nn.functional.softplus,
                        A.XF_SIGMOID: torch.sigmoid}.get(x.xf, lambda t: t)(out)
            if isinstance(x, Gather):
                idx = self._site_value(x.idx.addr).long()
                out = torch.as_tensor(x.table, device=idx.device)[idx]
                return out[..., 0] if x.dim == 1 else out
            if isinstance(x, Expr):
                def leaf(addr, elem):
                    v = self._site_value(addr).float()
                    return v if self.prog.site_list[addr].dim == 1 else v[..., elem]
                outs = [E._as_arr(o, torch, self.score) * torch.ones_like(self.score if self.batched else self.score[0])
                        for o in E.evaluate(x.elems, leaf, torch)]
                return outs[0] if x.dim == 1 else torch.stack(outs, dim=-1)
            if isinstance(x, HostExpr):
                return x.fn(ev)
            if isinstance(x, (tuple, list)):
                return type(x)(ev(e) for e in x)
            return x
        return ev(r)

    def get_particle(self, idx) -> "Trace":
        """Row gather of every leaf (smc.py:90-91)."""
        i = int(idx)
        return Trace(self.gen_fn, self.args, self.prog, self.choices[:, i:i + 1].contiguous(),
                     self.score[i:i + 1].contiguous(), self.shared, False, self.retval_sym)

    # -- edits ---------------------------------------------------------------------------------
    def update(self, key: Key, constraint: ChoiceMap, argdiffs=None):
        """-> (new trace, weight, retdiff, discard ChoiceMap)  (generative_function.py:168-183, 611-627)"""
        return self.gen_fn.update(key, self, constraint, argdiffs)

    def edit(self, key: Key, request, argdiffs=None):
        return request.edit(key, self, argdiffs)

    def project(self, key: Key, selection: Selection):
        """Sum of the scores of the selected choices (generative_function.py:184-194; static.py project: the
        per-site score of every selected site).  One assess launch with per-site scores."""
        import torch
        from . import kernels
        from .inference.requests import _rows_and_shared
        shared, rows = _rows_and_shared(self)
        prog, _, _ = self.gen_fn.pack(self.args, shared, False, rng_mode=self.prog.rng_mode, per_particle=tuple(rows), plates=False)
        out = kernels.run_program(prog, (0, 0), self.K, choices=self.rows_for(prog), want_site_scores=True, want_lse=False)
        sel = [j for j, s in enumerate(prog.site_list.sites) if selection.check(s.addr)]
        tot = out["site_scores"][sel].sum(dim=0) if sel else torch.zeros(self.K, device=self.score.device)
        return tot if self.batched else tot[0]

    def _site_scores(self):
        """(sites, f32[n_sites][K]) per-site scores of this trace: one assess launch with every site constrained to its value"""
        from . import kernels
        from .inference.requests import _rows_and_shared
        shared, rows = _rows_and_shared(self)
        prog, _, _ = self.gen_fn.pack(self.args, shared, False, rng_mode=self.prog.rng_mode, per_particle=tuple(rows), plates=False)
        out = kernels.run_program(prog, (0, 0), self.K, choices=self.rows_for(prog), want_site_scores=True, want_lse=False)
        return prog.site_list.sites, out["site_scores"]

    def get_subtrace(self, *addrs) -> "SubTrace":
        """the part of this trace below an address path (generative_function.py get_subtrace; tests/core/generative/
        test_core.py:57-170): ``tr.get_subtrace("f", "x")`` == ``tr.get_subtrace("f").get_subtrace("x")``; its score is the
        sum of the scores of the sites below the path — one score per step / instance when the path names the sites of a
        Scan or Vmap."""
        return SubTrace(self, _flat_path(addrs))

    def rows_for(self, prog: PackedProgram):
        """this trace's values laid out for ANOTHER packing of the same site list (a program packed with / without plates,
        or with other modes, orders its rows differently): f32[prog.n_slots][K]"""
        import torch
        if prog.slot_of == self.prog.slot_of and prog.n_slots == self.prog.n_slots:
            return self.choices.clone()
        out = torch.zeros((max(prog.n_slots, 1), self.K), dtype=torch.float32, device=self.score.device)
        for s in self.prog.site_list.sites:
            src, dst = self.prog.slot_of[s.addr], prog.slot_of.get(s.addr, -1)
            if src >= 0 and dst >= 0:
                out[dst:dst + s.dim] = self.choices[src:src + s.dim]
        return out

    def full_choice_rows(self) -> dict:
        """addr -> device rows [dim][K] for every site (shared values broadcast)."""
        import torch
        out = {}
        for s in self.prog.site_list.sites:
            slot = self.prog.slot_of[s.addr]
            if slot >= 0:
                out[s.addr] = self.choices[slot:slot + s.dim]
            else:
                v = torch.as_tensor(np.broadcast_to(np.asarray(self.shared[s.addr], np.float32).ravel(), (s.dim,)).copy(),
                                    device=self.score.device)
                out[s.addr] = v[:, None].expand(s.dim, self.K)
        return out


def _flat_path(addrs) -> tuple:
    out = []
    for a in addrs:
        out.extend(_flat_path(a) if isinstance(a, tuple) else [a])
    return tuple(out)


class SubTrace:
    """View of a trace below an address path: ``get_score``, ``get_choices``, ``get_subtrace`` (static.py:80-119: the
    reference keeps one sub-trace object per address; here it is a view that assesses the parent's sites on demand)."""

    def __init__(self, trace: Trace, prefix: tuple):
        self.trace, self.prefix = trace, tuple(prefix)

    def get_subtrace(self, *addrs) -> "SubTrace":
        return SubTrace(self.trace, self.prefix + _flat_path(addrs))

    def _members(self):
        from .core import norm_addr
        sites, scores = self.trace._site_scores()
        mem = []
        for j, st in enumerate(sites):
            name, idx = norm_addr(st.addr)
            path = name if isinstance(name, tuple) else (name,)
            if path[: len(self.prefix)] == self.prefix:
                mem.append((j, idx))
        if not mem:
            raise KeyError(f"no choices below {self.prefix!r}")
        return mem, scores

    def get_score(self):
        import torch
        mem, scores = self._members()
        tr = self.trace
        if all(idx is None for _, idx in mem):
            tot = scores[[j for j, _ in mem]].sum(dim=0)
            return tot if tr.batched else tot[0]
        order = sorted({idx for _, idx in mem if idx is not None}, key=lambda i: i if isinstance(i, tuple) else (i,))
        per = [scores[[j for j, idx in mem if idx == i]].sum(dim=0) for i in order]       # one row per step / instance
        st = torch.stack(per, dim=-1)                                                    # [K][n]
        return st if tr.batched else st[0]

    def get_choices(self):
        return self.trace.get_choices().get_submap(self.prefix) if self.prefix else self.trace.get_choices()


# ---------------------------------------------------------------------------------------------
# generative functions
# ---------------------------------------------------------------------------------------------
def _value_rows(v, dim: int):
    """Constraint value -> (np [dim] shared, or torch/np [dim][K] per particle)."""
    if hasattr(v, "detach"):
        t = v.float()
        if t.ndim == 0:
            return np.asarray([float(t)], np.float32), None
        if dim == 1 and t.ndim == 1 and t.shape[0] != 1:
            return None, t[None, :]
        if dim > 1 and t.ndim == 1:
            return t.detach().cpu().numpy().astype(np.float32), None
        if t.ndim == 2:
            return None, t.t()
        return t.detach().cpu().numpy().astype(np.float32).ravel(), None
    a = np.asarray(v, np.float32)
    if a.ndim == 0:
        return a.reshape(1), None
    if a.ndim == 1 and (dim > 1 or a.shape[0] == 1) and a.shape[0] == dim:
        return a, None
    if a.ndim == 1:
        return None, a[None, :]
    if a.ndim == 2:
        return None, a.T
    raise ValueError(f"constraint of shape {a.shape} for a site of dimension {dim}")


def _constraint_value(constraint: ChoiceMap, addr):
    """(found, value) for a site key ``name`` or ``(name, t)``; a scan / vmap site also matches a whole-sequence
    entry under its name (ChoiceMap.__getitem__ indexes the leading axis)."""
    if addr in constraint:
        return True, constraint[addr]
    return False, None


class GenerativeFunction:
    """GFI surface (core/generative/generative_function.py:238-689), restated for sited programs."""

    def site_list(self, args) -> tuple[SiteList, Any]:  # pragma: no cover - abstract
        raise NotImplementedError

    # -- program construction -------------------------------------------------------------------
    def pack(self, args, constraint: ChoiceMap, sample_rest: bool, selected: Sequence = (), rng_mode=None,
             per_particle: Sequence = (), plates: bool = True):
        """-> (PackedProgram, shared dict, per-particle dict addr -> rows).  Sites in ``constraint`` are
        OBS_TAB (shared value) or OBS_SLOT (value per particle); sites in ``per_particle`` are OBS_SLOT with
        rows supplied later; the rest are SAMPLE when ``sample_rest`` else MissingAddress.
        ``plates``: the instances of vmapped kernels become one vector site per kernel site on the device (program.py
        compact_plates; the logical addresses stay per instance); off for callers that need one score per instance."""
        from . import config
        # Packed programs are kept per (arguments, constraint content, flags): the same Target used again — every step of
        # an SMC loop, every call of the GenSP interface — finds its program with the table already on the device and
        # its engine already chosen, instead of re-tracing the body, re-packing and re-uploading (the reference gets the
        # same effect from jax.jit's trace cache).  Constraints that carry one value per particle are not cached.
        ckey = None
        cache = self.__dict__.setdefault("_pack_cache", {})
        try:
            items = []
            for addr, v in constraint._d.items():
                if isinstance(v, Masked) or np.size(_np_value(v)) > 4096:
                    items = None
                    break
                items.append((repr(addr), _value_key(v)))
            if items is not None:
                ckey = (_args_key(args), tuple(items), bool(sample_rest), tuple(selected),
                        config.rng_mode() if rng_mode is None else rng_mode, tuple(per_particle), plates if isinstance(plates, str) else bool(plates))
                hit = cache.get(ckey)
                if hit is not None and not hit[2]:
                    return hit[0], hit[1], {}
        except TypeError:
            ckey = None
        sl, _ = self.site_list(args)
        modes, shared, pp, mask_rows = {}, {}, {}, {}
        for s in sl.sites:
            found, cval = _constraint_value(constraint, s.addr)
            if found and isinstance(cval, Masked) and isinstance(cval.flag, bool):
                found, cval = cval.flag, cval.value         # a concrete flag is decided here: a plain constraint, or none
            if found and isinstance(cval, Masked):
                K = cval.flag.size
                sv, rows = _value_rows(cval.value, s.dim)
                if rows is None:
                    rows = np.broadcast_to(np.asarray(sv, np.float32).reshape(s.dim, 1), (s.dim, K)).copy()
                modes[s.addr] = A.MODE_OBS_MASK
                pp[s.addr] = rows
                mask_rows[s.addr] = cval.flag.astype(np.float32)
            elif found:
                sv, rows = _value_rows(cval, s.dim)
                if rows is None:
                    modes[s.addr] = A.MODE_OBS_TAB
                    shared[s.addr] = np.broadcast_to(sv, (s.dim,)).astype(np.float32)
                else:
                    modes[s.addr] = A.MODE_OBS_SLOT
                    pp[s.addr] = rows
            elif s.addr in per_particle:
                modes[s.addr] = A.MODE_OBS_SLOT
            elif not sample_rest:
                raise MissingAddress(s.addr)
        rm = config.rng_mode() if rng_mode is None else rng_mode
        prog = PackedProgram(sl, modes, shared, selected=tuple(selected), rng_mode=rm, plates=plates if isinstance(plates, str) else bool(plates))
        prog.mask_flags = mask_rows          # addr -> f32[K] validity flags of Mask(value, flag) constraints
        if ckey is not None and not pp and not mask_rows:
            if len(cache) >= 64:
                cache.pop(next(iter(cache)))
            cache[ckey] = (prog, shared, False)
        return prog, shared, pp

    def _run(self, key: Key, K: int, args, constraint: ChoiceMap, sample_rest: bool, batched: bool,
             prev_rows: dict | None = None, logw_in=None, sub=None, want_lse=False, device=None, offset=0,
             K_total=None, want_site_scores=False, plates: bool = True):
        import torch
        from . import kernels
        prog, shared, pp = self.pack(args, constraint, sample_rest, per_particle=tuple(prev_rows or ()), plates=plates)
        _, retval = self.site_list(args)
        dev = kernels._dev(device)
        rows = dict(prev_rows or {})
        rows.update(pp)
        for r in rows.values():
            if r.shape[-1] not in (1, K):
                raise ValueError(f"per-particle constraint has {r.shape[-1]} particles, expected {K}")
        choices = torch.empty((max(prog.n_slots, 1), K), dtype=torch.float32, device=dev)
        for addr, r in rows.items():
            slot = prog.slot_of[addr]
            dim = prog.site_list[addr].dim
            choices[slot:slot + dim] = torch.as_tensor(r, dtype=torch.float32, device=dev).expand(dim, K)
        for addr, fslot in prog.flag_slot_of.items():
            fl = torch.as_tensor(prog.mask_flags[addr], dtype=torch.float32, device=dev).reshape(-1)
            if fl.numel() != K:
                raise ValueError(f"mask of {addr!r} has {fl.numel()} flags, expected one per particle ({K})")
            choices[fslot] = fl
        out = kernels.run_program(prog, key, K, offset=offset, choices=choices, logw_in=logw_in, sub=sub,
                                  want_lse=want_lse, K_total=K_total, device=dev, want_site_scores=want_site_scores,
                                  ws=kernels.shared_workspace(A.OP_RUN, K, dev))
        tr = Trace(self, args, prog, out["choices"], out["score"], shared, batched, retval)
        return tr, out

    # -- GFI ------------------------------------------------------------------------------------
    def simulate(self, key: Key, args=(), K: int | None = None) -> Trace:
        tr, _ = self._run(key, K or 1, args, ChoiceMap.empty(), True, K is not None)
        return tr

    def generate(self, key: Key, constraint: ChoiceMap, args=(), K: int | None = None):
        kk = K or 1
        tr, out = self._run(key, kk, args, constraint, True, K is not None)
        w = out["weight"]
        return tr, (w if K is not None else w[0])

    importance = generate

    def assess(self, chm: ChoiceMap, args=()):
        K = 1
        batched = False
        sl, _ = self.site_list(args)
        for s in sl.sites:
            found, cval = _constraint_value(chm, s.addr)
            if found:
                _, rows = _value_rows(cval, s.dim)
                if rows is not None:
                    K, batched = int(rows.shape[-1]), True
        tr, out = self._run((0, 0), K, args, chm, False, batched)
        return (out["score"] if batched else out["score"][0]), tr.get_retval()

    def update(self, key: Key, trace: "Trace", constraint: ChoiceMap, argdiffs=None):
        """generative_function.py:611-627: ``Update(constraint).edit`` with the backward request's choice map as the
        discard."""
        from .inference.requests import Update
        tr, w, rd, bwd = Update(constraint).edit(key, trace, argdiffs)
        return tr, w, rd, bwd.constraint

    def propose(self, key: Key, args=(), K: int | None = None):
        tr = self.simulate(key, args, K)
        return tr.get_choices(), tr.get_score(), tr.get_retval()

    def marginal(self, selection: Selection | None = None, algorithm=None) -> "Marginal":
        return Marginal(self, selection or Selection.all(), algorithm)


class StaticGenerativeFunction(GenerativeFunction):
    """``@gen`` function (static.py:725-1036)."""

    def __init__(self, source: Callable):
        self.source = source
        self.__name__ = getattr(source, "__name__", "gen_fn")
        self.__doc__ = getattr(source, "__doc__", None)
        for a in ("__module__", "__qualname__"):              # static.py:1044-1049 (functools.wraps)
            if hasattr(source, a):
                try:
                    setattr(self, a, getattr(source, a))
                except (AttributeError, TypeError):
                    pass
        self.__wrapped__ = source
        self.partial_args: tuple = ()      # arguments filled in by partial_apply / method binding (static.py:1011-1036)
        self._cache: dict = {}

    def site_list(self, args):
        k = _args_key(args)
        if k not in self._cache:
            with _Tracer() as t:
                retval = self.source(*args)
            self._cache[k] = (t.sites, retval)
        return self._cache[k]

    def __call__(self, *args, **kwargs):
        """``callee(*args) @ "addr"`` inside a model body; outside one, the result is also a closure over the arguments
        with the generative-function interface taking ``()`` (generative_function.py GenerativeFunctionClosure)."""
        return GenClosure(self, args, kwargs)

    def partial_apply(self, *args) -> "StaticGenerativeFunction":
        """static.py:1011-1036: the same function with its leading arguments pre-filled (``partial_args`` lists them)."""
        src = self.source
        g = StaticGenerativeFunction(lambda *rest, **kw: src(*args, *rest, **kw))
        g.__name__ = self.__name__
        g.partial_args = self.partial_args + tuple(args)
        return g

    def __get__(self, obj, objtype=None):
        """``@gen`` on a method: ``instance.method`` is the generative function with ``self`` filled in (static.py gen on
        methods; tests/generative_functions/test_static_gen_fn.py:1116-1145)"""
        return self if obj is None else self.partial_apply(obj)

    def inline(self, *args, **kwargs):
        """inside a model body: the callee's choices at the CALLER's address level, no prefix (static.py `inline`)"""
        _Tracer.current()                     # (only meaningful while a body is traced)
        return self.source(*args, **kwargs)

    def vmap(self, in_axes=0) -> "VmapCombinator":
        """``kernel.vmap(in_axes=...)`` (combinators/vmap.py:193-218): one independent instance per index of the
        mapped (host-data) arguments."""
        return VmapCombinator(self, in_axes)

    def repeat(self, n: int) -> "VmapCombinator":
        """``kernel.repeat(n=...)`` (combinators/repeat.py): n i.i.d. instances with the same arguments."""
        return VmapCombinator(self, None, n=int(n))

    def scan(self, n: int | None = None) -> "ScanCombinator":
        """``kernel.scan(n=T)`` (combinators/scan.py): the kernel ``(carry, x) -> (carry, out)`` unrolled T times; without ``n``
        the length is the leading axis of the scanned argument (scan.py:380-400)."""
        return ScanCombinator(self, None if n is None else int(n))

    def iterate(self, n: int) -> "_Iterate":
        """``f.iterate(n=T)`` (scan.py:916-990): ``a -> a`` applied T times, returns [init, f(init), ..., f^T(init)]"""
        return iterate(n)(self)

    def iterate_final(self, n: int) -> "_Iterate":
        """``f.iterate_final(n=T)``: as iterate, returning only f^T(init)"""
        return iterate_final(n)(self)

    def accumulate(self) -> "_Accumulate":
        """``f.accumulate()`` (scan.py:992-1064): ``(acc, v) -> acc`` folded over the leading axis of ``vs``; returns
        [init, acc_1, ..., acc_T]"""
        return _Accumulate(self, final=False)

    def reduce(self) -> "_Accumulate":
        """``f.reduce()`` (scan.py:1066-1130): as accumulate, returning only the final accumulator"""
        return _Accumulate(self, final=True)

    def __repr__(self):
        return f"<gen {self.__name__}>"


def _nest(outer, i: int):
    """index of iteration i under the enclosing combinators' index `outer` (None, an int, or a tuple)"""
    if outer is None:
        return i
    return (outer if isinstance(outer, tuple) else (outer,)) + (i,)


class ScanCombinator(GenerativeFunction):
    """Time recursion of a kernel generative function (combinators/scan.py:200-294), lowered by unrolling:
    step t's sites get the addresses ``(addr, t)``, its parameters read step t-1's choices through the carry.
    Scores and weights add over steps (scan.py:283-294); whole-sequence constraints / selections use the bare
    address (``C["y"].set(vector)``, ``Selection.at["x"]``, ``chm[:, "x"]``)."""

    def __init__(self, kernel: StaticGenerativeFunction, n: int | None):
        self.kernel, self.n = kernel, n
        self._cache: dict = {}

    def _length(self, xs) -> int:
        """the number of steps: ``n``, or the common leading axis of the scanned argument's leaves (scan.py:380-400, 422-438)"""
        if self.n is not None:
            return self.n
        lens = sorted({int(np.shape(_np_arg(v))[0]) for v in _leaves(xs) if np.ndim(_np_arg(v)) > 0})
        if len(lens) != 1:
            raise ValueError("scan got values with different leading axis sizes: " + ", ".join(map(str, lens)) + "."
                             if lens else "scan without n needs a scanned argument with a leading axis")
        return lens[0]

    def _unroll(self, t: _Tracer, carry, xs):
        # a scan inside a vmap instance is a scan of its own (its own id, its own chained keys).  A scan inside a scan STEP (the
        # reference nests freely, scan.py:237-294) is a scan of its own as well — a fresh id per instantiation, i.e. per step of
        # the enclosing scan — and the sites of the enclosing step BEHIND it continue under a fresh id too: the device numbers
        # sites from 1 within a run of equal tags (include/gjx.h "Scan steps"), so re-entering the enclosing step's tag would
        # repeat its site numbers, hence its random streams.  Every run of sites thus has its own key chain.
        outer, outer_scan, nested = t.step, t.scan, t.in_scan
        outs = []
        sid = t.n_scans
        t.n_scans += 1
        n = self._length(xs)
        if n >= (1 << 20) - 1 or t.n_scans + (1 if nested else 0) > 2048:
            raise NotSupportedInModelBody("scan: at most 2^20 - 2 steps and 2048 scan instantiations per model (a scan inside a scan step takes two per step of the enclosing scan)")
        try:
            t.in_scan = True
            for i in range(n):
                t.step = _nest(outer, i)
                t.scan = (sid << 20) | (i + 1)      # step keys chain on the device: key_t = fold_in(key_{t-1}, t) (scan.py:268)
                carry, out = self.kernel.source(carry, None if xs is None else _step_of(xs, i))
                outs.append(out)
        finally:
            t.step = outer
            t.in_scan = nested
            if nested:                               # the rest of the enclosing step: a run of its own (one step, number 0)
                t.scan = (t.n_scans << 20) | 1
                t.n_scans += 1
            else:
                t.scan = outer_scan if outer_scan else 0
        return carry, outs

    def __call__(self, carry, xs=None):
        return GenCall(lambda t: self._unroll(t, carry, xs))

    def site_list(self, args):
        k = _args_key(tuple(a for a in args if a is not None)) + (len(args),)
        if k not in self._cache:
            with _Tracer() as t:
                ret = self._unroll(t, args[0], args[1] if len(args) > 1 else None)
            self._cache[k] = (t.sites, ret)
        return self._cache[k]


class VmapCombinator(GenerativeFunction):
    """Plate over a kernel generative function (combinators/vmap.py:193-218), lowered by unrolling like
    ``ScanCombinator``: instance i's sites get the addresses ``(addr, i)``; whole-plate constraints / selections
    use the bare address (``C["y"].set(vector)``, ``chm[:, "y"]``, ``chm[i, "y"]``); scores and weights add
    (vmap.py:206-218 sums the per-instance weights).  Mapped arguments are host data (arrays indexed on the
    mapped axis at trace time); ``in_axes`` follows jax.vmap: an int for every argument, or a tuple with
    ``None`` for broadcast arguments.  The reference derives instance keys by ``split(key, n)`` (vmap.py:186);
    here every site has its own counter-based stream, which is the same independence structure."""

    def __init__(self, kernel: StaticGenerativeFunction, in_axes=0, n: int | None = None):
        self.kernel, self.in_axes, self.n = kernel, in_axes, n
        self._cache: dict = {}

    def _axes(self, args) -> tuple:
        if self.n is not None:
            return (None,) * len(args)
        ax = self.in_axes
        if isinstance(ax, (int, type(None))):
            ax = (ax,) * len(args)
        ax = tuple(ax) + (None,) * (len(args) - len(ax))
        if len(ax) != len(args):
            raise ValueError(f"vmap in_axes {self.in_axes!r} does not match {len(args)} argument(s)")
        return ax

    def _unroll(self, t: _Tracer, args):
        outer = t.step                      # nested in a scan step or another vmap: the indices stack, outermost first
        if isinstance(outer, tuple) and len(outer) >= 3:
            raise NotSupportedInModelBody("combinators nest at most three deep")
        axes = self._axes(args)
        moved = [a if ax is None else np.moveaxis(_np_arg(a), ax, 0) for a, ax in zip(args, axes)]
        lens = {m.shape[0] for m, ax in zip(moved, axes) if ax is not None}
        if self.n is None and len(lens) != 1:
            raise ValueError(f"vmap: mapped arguments must share one axis length, got {sorted(lens)}")
        n = self.n if self.n is not None else lens.pop()
        outs = []
        try:
            for i in range(n):
                t.step = _nest(outer, i)
                outs.append(self.kernel.source(*[m if ax is None else m[i] for m, ax in zip(moved, axes)]))
        finally:
            t.step = outer
        return outs

    def __call__(self, *args):
        return GenCall(lambda t: self._unroll(t, args))

    def site_list(self, args):
        k = _args_key(args)
        if k not in self._cache:
            with _Tracer() as t:
                outs = self._unroll(t, args)
            self._cache[k] = (t.sites, outs)
        return self._cache[k]


def _np_arg(a) -> np.ndarray:
    if isinstance(a, Sym):
        raise NotSupportedInModelBody("vmap over a traced value is not supported: mapped arguments must be host data")
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def vmap(in_axes=0) -> Callable:
    """``@genjax.vmap(in_axes=...)`` decorator form (combinators/vmap.py:115)."""
    return lambda f: (f if isinstance(f, StaticGenerativeFunction) else gen(f)).vmap(in_axes)


def repeat(n: int) -> Callable:
    return lambda f: (f if isinstance(f, StaticGenerativeFunction) else gen(f)).repeat(n)


def iterate(n: int) -> Callable:
    """``@genjax.iterate(n=T)`` (combinators/scan.py:916-990): ``a -> a`` applied T times; all traced values nested
    under the step index; returns [init, f(init), f(f(init)), ...]."""
    def deco(f):
        g = f if isinstance(f, StaticGenerativeFunction) else gen(f)

        def kernel(carry, _):
            out = g.source(carry)
            return out, out
        sc = ScanCombinator(StaticGenerativeFunction(kernel), int(n))
        return _Iterate(sc, final=False)
    return deco


def iterate_final(n: int) -> Callable:
    """``@genjax.iterate_final(n=T)``: as iterate, returning only the final value."""
    def deco(f):
        it = iterate(n)(f)
        it.final = True
        return it
    return deco


def _leaves(x):
    if isinstance(x, dict):
        for v in x.values():
            yield from _leaves(v)
    elif isinstance(x, (tuple, list)):
        for v in x:
            yield from _leaves(v)
    elif x is not None:
        yield x


def _step_of(xs, i: int):
    """element i of the scanned argument: arrays are indexed on their leading axis, containers leaf by leaf"""
    if isinstance(xs, dict):
        return {k: _step_of(v, i) for k, v in xs.items()}
    if isinstance(xs, tuple):
        return tuple(_step_of(v, i) for v in xs)
    return xs[i]


class _Accumulate(GenerativeFunction):
    """``f.accumulate()`` / ``f.reduce()``: the kernel ``(acc, v) -> acc`` as a Scan over ``vs`` (scan.py:992-1130)"""

    def __init__(self, f: "StaticGenerativeFunction", final: bool):
        src = f.source

        def kernel(carry, v):
            out = src(carry, v)
            return out, out
        self.sc, self.final = ScanCombinator(StaticGenerativeFunction(kernel), None), final

    def site_list(self, args):
        sl, (carry, outs) = self.sc.site_list((args[0], args[1]))
        return sl, (carry if self.final else [args[0]] + list(outs))

    def __call__(self, init, vs):
        def inline(t):
            carry, outs = self.sc._unroll(t, init, vs)
            return carry if self.final else [init] + list(outs)
        return GenCall(inline)


class _Iterate(GenerativeFunction):
    def __init__(self, sc: "ScanCombinator", final: bool):
        self.sc, self.final = sc, final

    def site_list(self, args):
        sl, (carry, outs) = self.sc.site_list((args[0], None))
        return sl, (carry if self.final else [args[0]] + list(outs))

    def __call__(self, init):
        def inline(t):
            carry, outs = self.sc._unroll(t, init, None)
            return carry if self.final else [init] + list(outs)
        return GenCall(inline)


def scan(n: int | None = None) -> Callable:
    """``@genjax.scan(n=T)`` decorator form (combinators/scan.py)."""
    return lambda f: (f if isinstance(f, StaticGenerativeFunction) else gen(f)).scan(n)


def accumulate() -> Callable:
    """``@genjax.accumulate()`` decorator form (combinators/scan.py:791-852)"""
    return lambda f: (f if isinstance(f, StaticGenerativeFunction) else gen(f)).accumulate()


def reduce() -> Callable:
    """``@genjax.reduce()`` decorator form (combinators/scan.py:854-914)"""
    return lambda f: (f if isinstance(f, StaticGenerativeFunction) else gen(f)).reduce()


Scan, Vmap = ScanCombinator, VmapCombinator          # the reference's class names (scan.py:110, vmap.py:98)


def gen(fn: Callable) -> StaticGenerativeFunction:
    """static.py:1044-1049"""
    return StaticGenerativeFunction(fn)


class _IdKey:
    """Cache-key component for an argument that has no value identity: compares by object identity and keeps the
    object alive, so its id() cannot be recycled for a different object while the cache entry exists."""
    __slots__ = ("obj",)

    def __init__(self, obj):
        self.obj = obj

    def __hash__(self):
        return id(self.obj)

    def __eq__(self, other):
        return isinstance(other, _IdKey) and other.obj is self.obj


def _value_key(a):
    """Content key of one argument / constraint value (arrays by shape + bytes)."""
    if isinstance(a, (int, float, str, bool, type(None))):
        return a
    if isinstance(a, (np.ndarray, np.generic, list, tuple)) or hasattr(a, "detach"):
        arr = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        if arr.dtype == object:
            return tuple(_value_key(x) for x in a)
        return (arr.shape, arr.dtype.str, arr.tobytes())
    ck = getattr(a, "_cache_key", None)
    if ck is not None:
        return ck()
    return _IdKey(a)


def _np_value(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


def _args_key(args) -> tuple:
    return tuple(_value_key(a) for a in args)


class Marginal(GenerativeFunction):
    """The marginal of a generative function over a selection of addresses, as a sample distribution
    (inference/sp.py:207-252).  ``Target`` refuses it as a model (sp.py:46-49); as a PROPOSAL it is what
    ``gen_fn.marginal()`` hands to ``ImportanceK(target, q=...)``."""

    def __init__(self, gen_fn, selection, algorithm=None):
        self.gen_fn, self.selection, self.algorithm = gen_fn, selection, algorithm

    def random_weighted(self, key: Key, *args):
        """-> (score estimate, choices of the selection)  (sp.py:216-238): simulate, keep the selected choices; the
        weight is the projection on the UNSELECTED choices, or, with an inference algorithm, its estimate of the
        reciprocal normalising constant of Target(gen_fn, args, selected choices)."""
        from .inference.smc import Target
        key, sub_key = split(key)
        tr = self.gen_fn.simulate(sub_key, args)
        choices = tr.get_choices()
        latent_choices = choices.filter(self.selection)
        key, sub_key = split(key)
        weight = tr.project(sub_key, ~self.selection)
        if self.algorithm is None:
            return weight, latent_choices
        target = Target(self.gen_fn, args, latent_choices)
        other_choices = choices.filter(~self.selection)
        return self.algorithm.estimate_reciprocal_normalizing_constant(key, target, other_choices, weight), latent_choices

    def estimate_logpdf(self, key: Key, v: ChoiceMap, *args):
        """sp.py:240-252: importance weight of the given choices, or the algorithm's normalising-constant estimate"""
        from .inference.smc import Target
        if self.algorithm is None:
            _, weight = self.gen_fn.importance(key, v, args)
            return weight
        return self.algorithm.estimate_normalizing_constant(key, Target(self.gen_fn, args, v))


def marginal(selection: Selection | None = None, algorithm=None) -> Callable:
    """``@genjax.marginal(selection, algorithm)`` (sp.py:260-275)"""
    return lambda gen_fn: Marginal(gen_fn, selection or Selection.all(), algorithm)


# ---------------------------------------------------------------------------------------------
# primitive distributions (distribution.py:359-476 ExactDensity; tfp/__init__.py wrappers)
# ---------------------------------------------------------------------------------------------
class Distribution(GenerativeFunction):
    def __init__(self, name: str, kind: int, param_names: Sequence[str]):
        self.name, self.kind, self.param_names = name, kind, tuple(param_names)

    def _params(self, args, kwargs) -> tuple[int, list]:
        if kwargs:
            args = tuple(args) + tuple(kwargs[n] for n in self.param_names[len(args):])
        if len(args) != len(self.param_names):
            raise TypeError(f"{self.name}() takes parameters {self.param_names}")
        return self.kind, list(args)

    def _dim(self, params):
        return None

    def __call__(self, *args, **kwargs) -> DistCall:
        kind, params = self._params(args, kwargs)
        return DistCall(self, kind, params, self._dim(params))

    # one-site program for the stand-alone API
    def _as_gen(self, args, kwargs=None) -> StaticGenerativeFunction:
        d = self
        def body():
            return d(*args, **(kwargs or {})) @ _VALUE
        return StaticGenerativeFunction(body)

    def _as_gen_with_args(self) -> StaticGenerativeFunction:
        """one-site @gen function that takes the distribution's parameters as ITS arguments, so that a trace can be
        updated under new arguments (distribution.py:179-244)"""
        if getattr(self, "_argful", None) is None:
            d = self
            self._argful = StaticGenerativeFunction(lambda *a: d(*a) @ _VALUE)
            self._argful.__name__ = self.name
        return self._argful

    def site_list(self, args):
        return self._as_gen(args).site_list(())

    def simulate(self, key, args=(), K=None):
        return self._as_gen_with_args().simulate(key, tuple(args), K)

    def update(self, key, trace, constraint: ChoiceMap, argdiffs=None):
        return self._as_gen_with_args().update(key, trace, constraint, argdiffs)

    def sample(self, key, *args, **kwargs):
        return self._as_gen(args, kwargs).simulate(key, ()).get_choices()[_VALUE]

    def logpdf(self, v, *args, **kwargs):
        return self._as_gen(args, kwargs).assess(ChoiceMap.v(v), ())[0]

    def assess(self, chm: ChoiceMap, args=()):
        g = self._as_gen(args)
        v = chm.get_value() if chm.has_value() else chm[_VALUE]
        return g.assess(ChoiceMap.v(v), ())

    def generate(self, key, constraint: ChoiceMap, args=(), K=None):
        return self._as_gen_with_args().generate(key, constraint, tuple(args), K)

    importance = generate

    def vmap(self, in_axes=0) -> "VmapCombinator":
        """``genjax.normal.vmap()(locs, scales) @ "a"``: one independent draw per index, addressed ["a", i]."""
        d = self
        return VmapCombinator(StaticGenerativeFunction(lambda *a: d(*a) @ _VALUE), in_axes)

    def random_weighted(self, key, *args):
        tr = self.simulate(key, args)
        return tr.get_score(), tr.get_choices()[_VALUE]

    def estimate_logpdf(self, key, v, *args):
        return self.logpdf(v, *args)

    def __repr__(self):
        return f"genjax_amd.{self.name}"


class _VectorDist(Distribution):
    def _dim(self, params):
        d = 1
        for p in params:
            if isinstance(p, Sym):
                d = max(d, getattr(p, "dim", 1))
            else:
                d = max(d, int(np.asarray(p).size))
        return d


class _Categorical(Distribution):
    def _params(self, args, kwargs):
        if "probs" in kwargs:
            return A.CATEGORICAL_PROBS, [kwargs["probs"]]
        if "logits" in kwargs:
            return A.CATEGORICAL_LOGITS, [kwargs["logits"]]
        if len(args) != 1:
            raise TypeError("categorical(logits) / categorical(probs=...) / categorical(logits=...)")
        warnings.warn("bare argument to categorical is interpreted as logits (distribution.py:479-500)",
                      DeprecationWarning, stacklevel=3)
        return A.CATEGORICAL_LOGITS, [args[0]]


class _Bernoulli(Distribution):
    def _params(self, args, kwargs):
        if "probs" in kwargs:
            return A.FLIP, [kwargs["probs"]]
        if "logits" in kwargs:
            return A.BERNOULLI_LOGITS, [kwargs["logits"]]
        if len(args) != 1:
            raise TypeError("bernoulli(logits) / bernoulli(probs=...) / bernoulli(logits=...)")
        return A.BERNOULLI_LOGITS, [args[0]]


class _Geometric(_VectorDist):
    """tfd.Geometric(logits=None, probs=None): a bare argument is LOGITS (the reference passes tfd.Geometric through
    unwrapped, tensorflow_probability/__init__.py:169)."""

    def _params(self, args, kwargs):
        if "probs" in kwargs:
            return self.kind, [kwargs["probs"]]
        l = kwargs["logits"] if "logits" in kwargs else (args[0] if len(args) == 1 else None)
        if l is None:
            raise TypeError("geometric(logits) / geometric(probs=...) / geometric(logits=...)")
        return self.kind, [sigmoid(l)]


class _Poisson(_VectorDist):
    def _params(self, args, kwargs):
        if "log_rate" in kwargs:
            return self.kind, [exp(kwargs["log_rate"])]
        return super()._params(args, kwargs)


class _NegativeBinomial(_VectorDist):
    """tfd.NegativeBinomial(total_count, logits=None, probs=None): a bare second argument is LOGITS (the reference passes the TFP
    distribution through unwrapped, tensorflow_probability/__init__.py:249)"""

    def _params(self, args, kwargs):
        tc = kwargs["total_count"] if "total_count" in kwargs else (args[0] if args else None)
        if tc is None:
            raise TypeError("negative_binomial(total_count, logits) / negative_binomial(total_count, probs=...)")
        if "probs" in kwargs:
            p = kwargs["probs"]
            return self.kind, [tc, log(p) - log1p(-p)]
        l = kwargs["logits"] if "logits" in kwargs else (args[1] if len(args) == 2 else None)
        if l is None:
            raise TypeError("negative_binomial(total_count, logits) / negative_binomial(total_count, probs=...)")
        return self.kind, [tc, l]


normal = _VectorDist("normal", A.NORMAL, ("loc", "scale"))
mv_normal_diag = _VectorDist("mv_normal_diag", A.MVNORMAL_DIAG, ("loc", "scale_diag"))
flip = _VectorDist("flip", A.FLIP, ("p",))
bernoulli = _Bernoulli("bernoulli", A.BERNOULLI_LOGITS, ("logits",))
beta = Distribution("beta", A.BETA, ("concentration1", "concentration0"))
categorical = _Categorical("categorical", A.CATEGORICAL_LOGITS, ("logits",))
uniform = _VectorDist("uniform", A.UNIFORM, ("low", "high"))
exponential = _VectorDist("exponential", A.EXPONENTIAL, ("rate",))
half_normal = _VectorDist("half_normal", A.HALF_NORMAL, ("scale",))
laplace = _VectorDist("laplace", A.LAPLACE, ("loc", "scale"))
log_normal = _VectorDist("log_normal", A.LOG_NORMAL, ("loc", "scale"))
cauchy = _VectorDist("cauchy", A.CAUCHY, ("loc", "scale"))
gamma = Distribution("gamma", A.GAMMA, ("concentration", "rate"))
student_t = _VectorDist("student_t", A.STUDENT_T, ("df", "loc", "scale"))
truncated_normal = _VectorDist("truncated_normal", A.TRUNCATED_NORMAL, ("loc", "scale", "low", "high"))
poisson = _Poisson("poisson", A.POISSON, ("rate",))
geometric = _Geometric("geometric", A.GEOMETRIC, ("probs",))
dirichlet = _VectorDist("dirichlet", A.DIRICHLET, ("concentration",))
gumbel = _VectorDist("gumbel", A.GUMBEL, ("loc", "scale"))
half_cauchy = _VectorDist("half_cauchy", A.HALF_CAUCHY, ("loc", "scale"))
inverse_gamma = _VectorDist("inverse_gamma", A.INVERSE_GAMMA, ("concentration", "scale"))
weibull = _VectorDist("weibull", A.WEIBULL, ("concentration", "scale"))
logit_normal = _VectorDist("logit_normal", A.LOGIT_NORMAL, ("loc", "scale"))
chi2 = _VectorDist("chi2", A.CHI2, ("df",))
# (round 6: nine more of the reference's TFP wrappers, tensorflow_probability/__init__.py:115-284)
chi = _VectorDist("chi", A.CHI, ("df",))
exp_gamma = _VectorDist("exp_gamma", A.EXP_GAMMA, ("concentration", "rate"))
exp_inverse_gamma = _VectorDist("exp_inverse_gamma", A.EXP_INVERSE_GAMMA, ("concentration", "scale"))
half_student_t = _VectorDist("half_student_t", A.HALF_STUDENT_T, ("df", "loc", "scale"))
kumaraswamy = _VectorDist("kumaraswamy", A.KUMARASWAMY, ("concentration1", "concentration0"))
moyal = _VectorDist("moyal", A.MOYAL, ("loc", "scale"))
truncated_cauchy = _VectorDist("truncated_cauchy", A.TRUNCATED_CAUCHY, ("loc", "scale", "low", "high"))
double_sided_maxwell = _VectorDist("double_sided_maxwell", A.DOUBLESIDED_MAXWELL, ("loc", "scale"))
inverse_gaussian = _VectorDist("inverse_gaussian", A.INVERSE_GAUSSIAN, ("loc", "concentration"))
negative_binomial = _NegativeBinomial("negative_binomial", A.NEGATIVE_BINOMIAL, ("total_count", "logits"))
von_mises = _VectorDist("von_mises", A.VON_MISES, ("loc", "concentration"))
