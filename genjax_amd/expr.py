"""General elementwise expressions between sites — host side of ``GJX_P_EXPR`` (include/gjx.h).

The reference stages a ``@gen`` body and interprets ANY JAX computation between two trace sites
(generative_functions/static.py:383-399, core/compiler/staging.py:286-298); ``selection_gradient`` differentiates
through it (inference/requests/hmc.py:70-96).  The closed parameter forms of program.py (constant, value, gather,
affine) cover what the engines have fast paths for; everything else — ``normal(a * b, 1)``, ``exp(a) + exp(b)``,
``w2 @ tanh(W1 @ x)`` — is recorded here as a DAG of scalar nodes and lowered to a block of SSA nodes in the program's
float table, which the generated kernels emit as straight-line code, the site interpreters and the oracle evaluate, and
the HMC engines sweep backwards.

A node is a hashable tuple:
    ("c", float)                         constant
    ("v", addr, elem)                    element `elem` of the choice at `addr`
    (op, a) / (op, a, b)                 unary / binary elementwise op, op in UNARY / BINARY
    ("where", cond, a, b)                cond != 0 ? a : b
    ("lin", bias, ((node, weight), ...)) bias + sum_k weight_k * node_k     (rows of W @ x: one device node per run of sources)
"""
from __future__ import annotations

import math
from typing import Callable, Sequence

import numpy as np

from . import _abi as A

UNARY = {"neg": A.E_NEG, "exp": A.E_EXP, "log": A.E_LOG, "sqrt": A.E_SQRT, "square": A.E_SQUARE, "tanh": A.E_TANH,
         "sigmoid": A.E_SIGMOID, "softplus": A.E_SOFTPLUS, "abs": A.E_ABS, "sin": A.E_SIN, "cos": A.E_COS, "log1p": A.E_LOG1P,
         "recip": A.E_RECIP}
BINARY = {"add": A.E_ADD, "sub": A.E_SUB, "mul": A.E_MUL, "div": A.E_DIV, "max": A.E_MAX, "min": A.E_MIN, "gt": A.E_GT}

_NP_UNARY = {
    "neg": lambda x: -x, "exp": np.exp, "log": np.log, "sqrt": np.sqrt, "square": lambda x: x * x, "tanh": np.tanh,
    "sigmoid": lambda x: 1.0 / (1.0 + np.exp(-x)), "softplus": lambda x: np.logaddexp(0.0, x), "abs": np.abs, "sin": np.sin,
    "cos": np.cos, "log1p": np.log1p, "recip": lambda x: 1.0 / x,
}


class ExprTooLarge(TypeError):
    """the block of one distribution parameter exceeds GJX_EXPR_MAX_NODES"""


# ---------------------------------------------------------------------------------------------
# construction (with the constant folding a tracer's user expects: 2 * 3 * x is one multiply)
# ---------------------------------------------------------------------------------------------
def const(x) -> tuple:
    return ("c", float(x))


def value(addr, elem: int = 0) -> tuple:
    return ("v", addr, int(elem))


def is_const(n) -> bool:
    return n[0] == "c"


def unary(op: str, a: tuple) -> tuple:
    if is_const(a):
        with np.errstate(all="ignore"):
            return const(_NP_UNARY[op](np.float64(a[1])))
    if op == "neg" and a[0] == "neg":
        return a[1]
    return (op, a)


def binary(op: str, a: tuple, b: tuple) -> tuple:
    if is_const(a) and is_const(b):
        x, y = np.float64(a[1]), np.float64(b[1])
        with np.errstate(all="ignore"):
            r = {"add": x + y, "sub": x - y, "mul": x * y, "div": x / y, "max": max(x, y), "min": min(x, y), "gt": float(x > y)}[op]
        return const(r)
    if a == b:                      # the same node on both sides: decided here, not by a float comparison of a value with itself
        if op in ("max", "min"):
            return a
        if op == "gt":
            return const(0.0)
        if op == "sub":
            return const(0.0)
    if op == "add":
        if is_const(a) and a[1] == 0.0:
            return b
        if is_const(b) and b[1] == 0.0:
            return a
    if op == "sub" and is_const(b) and b[1] == 0.0:
        return a
    if op == "mul":
        if (is_const(a) and a[1] == 1.0):
            return b
        if (is_const(b) and b[1] == 1.0):
            return a
        # a constant factor is a one-term linear form (the device's fused multiply-add against a table entry)
        if is_const(a):
            return lin(0.0, [(b, a[1])])
        if is_const(b):
            return lin(0.0, [(a, b[1])])
    if op == "div" and is_const(b):
        return lin(0.0, [(a, 1.0 / b[1])]) if b[1] != 0.0 else (op, a, b)
    return (op, a, b)


def where(c: tuple, a: tuple, b: tuple) -> tuple:
    if is_const(c):
        return a if c[1] != 0.0 else b
    return ("where", c, a, b)


def lin(bias: float, terms: Sequence) -> tuple:
    """bias + sum w * node, flattened (a linear form of linear forms stays one form), constants folded, zero weights dropped"""
    acc: dict = {}
    order: list = []
    b = float(bias)
    for n, w in terms:
        w = float(w)
        if w == 0.0:
            continue
        if is_const(n):
            b += w * n[1]
        elif n[0] == "lin":
            b += w * n[1]
            for m, wm in n[2]:
                if m not in acc:
                    acc[m] = 0.0
                    order.append(m)
                acc[m] += w * wm
        else:
            if n not in acc:
                acc[n] = 0.0
                order.append(n)
            acc[n] += w
    ts = tuple((n, acc[n]) for n in order if acc[n] != 0.0)
    if not ts:
        return const(b)
    if len(ts) == 1 and ts[0][1] == 1.0 and b == 0.0:
        return ts[0][0]
    return ("lin", b, ts)


def add(a, b):
    # sums of (scaled) nodes stay ONE linear form: a + b + 2 c is one device node over three sources
    if a[0] in ("lin", "c") or b[0] in ("lin", "c"):
        return lin(0.0, [(a, 1.0), (b, 1.0)])
    return binary("add", a, b)


def sub(a, b):
    if a[0] in ("lin", "c") or b[0] in ("lin", "c"):
        return lin(0.0, [(a, 1.0), (b, -1.0)])
    return binary("sub", a, b)


# ---------------------------------------------------------------------------------------------
# queries and rewrites
# ---------------------------------------------------------------------------------------------
def _children(n):
    k = n[0]
    if k in ("c", "v"):
        return ()
    if k == "lin":
        return tuple(m for m, _ in n[2])
    return n[1:]


def sources(outs: Sequence[tuple]) -> list:
    """addresses of the choices the expression reads, in first-use order"""
    seen, out, stack, done = set(), [], list(reversed(list(outs))), set()
    while stack:
        n = stack.pop()
        if n in done:
            continue
        done.add(n)
        if n[0] == "v":
            if n[1] not in seen:
                seen.add(n[1])
                out.append(n[1])
        else:
            stack.extend(reversed(_children(n)))
    return out


def rewrite_leaves(outs: Sequence[tuple], fn: Callable) -> list:
    """every ("v", addr, elem) leaf replaced by fn(addr, elem) -> node (or None: unchanged); constants re-folded"""
    memo: dict = {}

    def go(n):
        if n in memo:
            return memo[n]
        k = n[0]
        if k == "c":
            r = n
        elif k == "v":
            r = fn(n[1], n[2])
            r = n if r is None else r
        elif k == "lin":
            r = lin(n[1], [(go(m), w) for m, w in n[2]])
        elif k == "where":
            r = where(go(n[1]), go(n[2]), go(n[3]))
        elif k in UNARY:
            r = unary(k, go(n[1]))
        else:
            r = binary(k, go(n[1]), go(n[2]))
        memo[n] = r
        return r
    return [go(n) for n in outs]


def evaluate(outs: Sequence[tuple], leaf: Callable, xp=np):
    """values of the output nodes; ``leaf(addr, elem)`` returns an array (numpy or torch: ``xp`` the module); results
    broadcast like the leaves.  Used for return values (Trace.get_retval) and by the tests as the float64 reference."""
    memo: dict = {}
    is_t = xp is not np

    def un(op, x):
        if not is_t:
            with np.errstate(all="ignore"):
                return _NP_UNARY[op](x)
        import torch
        return {"neg": lambda v: -v, "exp": torch.exp, "log": torch.log, "sqrt": torch.sqrt, "square": lambda v: v * v, "tanh": torch.tanh,
                "sigmoid": torch.sigmoid, "softplus": torch.nn.functional.softplus, "abs": torch.abs, "sin": torch.sin, "cos": torch.cos,
                "log1p": torch.log1p, "recip": lambda v: 1.0 / v}[op](x)

    def go(n):
        if n in memo:
            return memo[n]
        k = n[0]
        if k == "c":
            r = n[1]
        elif k == "v":
            r = leaf(n[1], n[2])
        elif k == "lin":
            r = n[1]
            for m, w in n[2]:
                r = r + w * go(m)
        elif k == "where":
            c, a, b = go(n[1]), go(n[2]), go(n[3])
            r = xp.where(_as_arr(c, xp) != 0, _as_arr(a, xp, c), _as_arr(b, xp, c))
        elif k in UNARY:
            r = un(k, _as_arr(go(n[1]), xp))
        else:
            a, b = go(n[1]), go(n[2])
            if k == "add":
                r = a + b
            elif k == "sub":
                r = a - b
            elif k == "mul":
                r = a * b
            elif k == "div":
                r = a / b
            elif k == "max":
                r = xp.maximum(_as_arr(a, xp, b), _as_arr(b, xp, a))
            elif k == "min":
                r = xp.minimum(_as_arr(a, xp, b), _as_arr(b, xp, a))
            else:
                r = (_as_arr(a, xp, b) > _as_arr(b, xp, a)) * 1.0
        memo[n] = r
        return r
    return [go(n) for n in outs]


def _as_arr(x, xp, like=None):
    if xp is np:
        return np.asarray(x, np.float64)
    import torch
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(like, torch.Tensor):
        return torch.as_tensor(x, dtype=like.dtype, device=like.device)
    return torch.as_tensor(x, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------
# lowering to the device block
# ---------------------------------------------------------------------------------------------
def lower(outs: Sequence[tuple], leaf_place: Callable, push: Callable) -> tuple[np.ndarray, int]:
    """-> (float32 [n][6] node list {op, a, b, c, da, db}, n) for include/gjx.h GJX_P_EXPR; the LAST len(outs) nodes are the outputs.

    ``leaf_place(addr, elem)`` -> ("slot", s) for a latent choice (a row of choices[][]) or ("tab", off) for a choice constrained
    to one shared value (it lives in the table: a later set_obs is seen without repacking); ``push(arr) -> off`` appends floats
    to the program's table (constants, the [bias, weights...] of linear nodes)."""
    nodes: list[list[int]] = []
    index: dict = {}                 # host node -> device node (first emission)
    const_off: dict = {}
    pushed: dict = {}                # equal float arrays are stored once per block (the 8 rows of W1 @ x all carry the weights x)
    push_raw = push

    def push(arr):
        a_ = np.asarray(arr, np.float32).ravel()
        k_ = a_.tobytes()
        if k_ not in pushed:
            pushed[k_] = push_raw(a_)
        return pushed[k_]

    def emit(op, a=0, b=0, c=0) -> int:
        nodes.append([int(op), int(a), int(b), int(c), 0, 0])           # (da, db: plate strides, filled by lower_plate)
        if len(nodes) > A.EXPR_MAX_NODES:
            raise ExprTooLarge(f"a distribution parameter's expression needs more than {A.EXPR_MAX_NODES} nodes: give an intermediate result a "
                               "site of its own, or use the closed forms (affine maps, gathers) where they apply")
        return len(nodes) - 1

    def const_at(x: float) -> int:
        k = np.float32(x).tobytes()
        if k not in const_off:
            const_off[k] = push(np.asarray([x], np.float32))
        return const_off[k]

    def top(n, fresh: bool = False) -> int:
        """device node of host node n (its children are lowered first); fresh: emit the top node again even if it exists —
        what puts the operands of a LINN node, and the outputs of the block, at consecutive indices"""
        if not fresh and n in index:
            return index[n]
        k = n[0]
        if k == "c":
            i = emit(A.E_CONST, const_at(n[1]))
        elif k == "v":
            where_, at = leaf_place(n[1], n[2])
            i = emit(A.E_VALUE if where_ == "slot" else A.E_CONST, at)
        elif k == "lin":
            i = lower_lin(n)
        elif k == "where":
            c_, a_, b_ = top(n[1]), top(n[2]), top(n[3])
            i = emit(A.E_WHERE, c_, a_, b_)
        elif k in UNARY:
            a_ = top(n[1])
            i = emit(UNARY[k], a_)
        else:
            a_, b_ = top(n[1]), top(n[2])
            i = emit(BINARY[k], a_, b_)
        index.setdefault(n, i)
        return i

    def lower_lin(n) -> int:
        bias, terms = n[1], n[2]
        lat, rest = [], []
        for m, w in terms:
            if m[0] == "v":
                where_, at = leaf_place(m[1], m[2])
                if where_ == "slot":
                    lat.append((at, w))
                    continue
            rest.append((m, w))
        parts: list[int] = []
        lat.sort()
        # runs of consecutive slots -> LINV (choices read straight from registers / rows); small gaps are bridged with zero weights
        j = 0
        while j < len(lat):
            run = [lat[j]]
            while j + 1 < len(lat) and 0 < lat[j + 1][0] - run[-1][0] <= 2 and (lat[j + 1][0] - run[0][0]) < 64:
                for g in range(run[-1][0] + 1, lat[j + 1][0]):
                    run.append((g, 0.0))
                run.append(lat[j + 1])
                j += 1
            j += 1
            b_here = bias if not parts else 0.0
            off = push(np.asarray([b_here] + [w for _, w in run], np.float32))
            parts.append(emit(A.E_LINV, off, run[0][0], len(run)))
        # everything else -> LINN over operands emitted at consecutive indices (chunks of 64)
        for c0 in range(0, len(rest), 64):
            chunk = rest[c0:c0 + 64]
            for m, _ in chunk:                     # children first, so that the fresh tops below are consecutive
                if m[0] not in ("c", "v"):
                    for ch in _children(m):
                        top(ch)
            first = None
            for m, _ in chunk:
                i = top(m, fresh=True)
                first = i if first is None else first
            b_here = bias if not parts else 0.0
            off = push(np.asarray([b_here] + [w for _, w in chunk], np.float32))
            parts.append(emit(A.E_LINN, off, first, len(chunk)))
        acc = parts[0]
        for p_ in parts[1:]:
            acc = emit(A.E_ADD, acc, p_)
        return acc

    outs = list(outs)
    for n in outs:                                  # everything below the outputs
        if n[0] == "lin":
            continue                                # (a linear output is emitted whole, below)
        for ch in _children(n):
            top(ch)
    first_out = None
    if len(outs) == 1:
        i = top(outs[0])
        if i != len(nodes) - 1:
            i = top(outs[0], fresh=True) if outs[0][0] != "lin" else emit(A.E_MAX, i, i)
    else:
        # outputs are the LAST len nodes, in order: linear outputs may need several nodes each, so they go first and are copied
        pre = [top(n) if n[0] == "lin" else None for n in outs]
        for n, p_ in zip(outs, pre):
            i = emit(A.E_MAX, p_, p_) if p_ is not None else top(n, fresh=True)     # max(x, x): the identity, gradient to x
            first_out = i if first_out is None else first_out
    return np.asarray(nodes, np.float32).reshape(-1, A.EXPR_NODE_FLOATS), len(nodes)


def count_nodes(outs: Sequence[tuple]) -> int:
    """a lower bound of the device nodes the outputs need (distinct host nodes): lets the tracer refuse early"""
    seen: set = set()
    stack = list(outs)
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        stack.extend(_children(n))
    return len(seen)


POOL_BASE = 1 << 22      # lower_plate: table offsets at or above it are relative to the instance's own constant pool


class IrregularPlate(ValueError):
    """the instances of a vmapped kernel do not lower to ONE node list with linear strides"""


def lower_plate(inst_outs: Sequence[Sequence[tuple]], leaf_place: Callable, push: Callable | None) -> tuple[np.ndarray, int]:
    """The expression of a parameter of a PLATE body site (include/gjx.h "Plates"): ``inst_outs[i]`` = its output nodes in instance i
    (leaves already name device sites: ("v", device addr, element incl. the instance's offset)).  Every instance is lowered on its
    own into a private constant pool; the instances must give the SAME node list up to fields that advance linearly with i — the
    strides da / db of the device nodes: VALUE / CONST leaves and LINV slots by what the leaves say, everything that points into the
    pool by the pool's length (the pools are stored one after the other) — else IrregularPlate (the caller keeps the plate unrolled).
    ``push`` None: a dry run (regularity check only)."""
    lowered = []
    for outs in inst_outs:
        pool: list = []

        def lpush(arr, pool=pool):
            off = len(pool)
            pool.extend(np.asarray(arr, np.float32).ravel().tolist())
            return POOL_BASE + off
        nodes, n = lower(outs, leaf_place, lpush)
        lowered.append((nodes.astype(np.int64), pool))
    nd0, pool0 = lowered[0]
    L = len(pool0)
    n = nd0.shape[0]
    if any(nd.shape != nd0.shape or len(pool) != L for nd, pool in lowered):
        raise IrregularPlate("instances lower to different node lists")
    out = nd0.copy()
    n_inst = len(lowered)
    for i_node in range(n):
        op = int(nd0[i_node, 0])
        for nd, _ in lowered:
            if int(nd[i_node, 0]) != op or int(nd[i_node, 3]) != int(nd0[i_node, 3]):
                raise IrregularPlate("instances differ in an operation")
        strided = {A.E_CONST: (1,), A.E_VALUE: (1,), A.E_LINV: (1, 2), A.E_LINN: (1,)}.get(op, ())
        for f in (1, 2):
            col = [int(nd[i_node, f]) for nd, _ in lowered]
            if f not in strided:
                if any(c != col[0] for c in col):
                    raise IrregularPlate("instances differ in an operand")
                continue
            if col[0] >= POOL_BASE:                 # into the instance's pool: same relative entry, stride = pool length
                if any(c != col[0] for c in col):
                    raise IrregularPlate("instances use different pool entries")
                out[i_node, f] = col[0]             # (rebased below)
                out[i_node, 3 + f] = L
            else:
                d = col[1] - col[0] if n_inst > 1 else 0
                if any(col[i] != col[0] + i * d for i in range(n_inst)):
                    raise IrregularPlate("a source does not advance linearly with the instance")
                out[i_node, 3 + f] = d
    if push is None:
        return out.astype(np.float32), n
    base = push(np.asarray([v for _, pool in lowered for v in pool], np.float32)) if L else 0
    for i_node in range(n):
        for f in (1, 2):
            if out[i_node, f] >= POOL_BASE:
                out[i_node, f] = base + (out[i_node, f] - POOL_BASE)
    return out.astype(np.float32), n
