"""Host-side data types with the reference's names: PRNG keys, ChoiceMap, Selection.

Only the static-address forms that the hot path needs are restated:
  - ChoiceMap / ChoiceMapBuilder ``C``      core/generative/choice_map.py:847-1395
  - Selection / SelectionBuilder ``S``      core/generative/choice_map.py:124-361
  - key / split / fold_in                   jax.random (Threefry-2x32, partitionable layout)
Values are Python scalars, NumPy arrays or torch tensors; a leading batch axis means "one value per
particle".  No arithmetic happens here — this is trace bookkeeping.
"""
from __future__ import annotations

from typing import Any, Iterable

import numpy as np

# ---------------------------------------------------------------------------------------------
# PRNG keys (host side).  key(seed) == (0, seed); split(k, n)[i] == fold_in(k, i) == Threefry(k, (0, i))
# (jax 0.5.2, jax_threefry_partitionable=True).  The same function runs on the device for per-particle
# keys; this host copy only serves the few scalar splits of the inference drivers (smc.py:154, 299).
# ---------------------------------------------------------------------------------------------
_M = 0xFFFFFFFF
_ROT = (13, 15, 26, 6, 17, 29, 16, 24)


_native = None      # gjx_host_threefry2x32 once the library is loaded (False: not available, stay in Python)


def threefry2x32(k0: int, k1: int, c0: int, c1: int) -> tuple[int, int]:
    """Threefry-2x32-20 of one counter.  The drivers call it a few times per step: through the library's host entry point
    when it is loaded (0.5 us instead of 4 us of Python integer arithmetic), the Python restatement below otherwise
    (`threefry2x32_py`; both are checked against the Random123 KATs)."""
    global _native
    if _native is None:
        try:
            from ._lib import load
            _native = load().gjx_host_threefry2x32
        except Exception:
            _native = False
    if _native:
        v = _native(k0 & _M, k1 & _M, c0 & _M, c1 & _M)
        return (v >> 32) & _M, v & _M
    return threefry2x32_py(k0, k1, c0, c1)


def threefry2x32_py(k0: int, k1: int, c0: int, c1: int) -> tuple[int, int]:
    ks = (k0 & _M, k1 & _M, (k0 ^ k1 ^ 0x1BD11BDA) & _M)
    x0, x1 = (c0 + ks[0]) & _M, (c1 + ks[1]) & _M
    for g in range(5):
        rs = _ROT[4:] if g & 1 else _ROT[:4]
        for r in rs:
            x0 = (x0 + x1) & _M
            x1 = ((x1 << r) | (x1 >> (32 - r))) & _M
            x1 ^= x0
        x0 = (x0 + ks[(g + 1) % 3]) & _M
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & _M
    return x0, x1


Key = tuple  # (uint32, uint32)


def key(seed: int) -> Key:
    seed = int(seed)
    return ((seed >> 32) & _M, seed & _M)


def fold_in(k: Key, data: int) -> Key:
    data = int(data)
    return threefry2x32(k[0], k[1], (data >> 32) & _M, data & _M)


def split(k: Key, num: int = 2) -> list[Key]:
    return [fold_in(k, i) for i in range(num)]


# ---------------------------------------------------------------------------------------------
# Argument change tags (core/interpreters/incremental.py:  Diff, NoChange, UnknownChange).  The kernels always
# recompute every visited density, so the tag only matters as API surface: ``Diff.tree_primal`` strips it.
# ---------------------------------------------------------------------------------------------
class _ChangeTag:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


NoChange, UnknownChange = _ChangeTag("NoChange"), _ChangeTag("UnknownChange")


class Diff:
    def __init__(self, primal, tangent=UnknownChange):
        self.primal, self.tangent = primal, tangent

    @staticmethod
    def no_change(v):
        return tuple(Diff(x, NoChange) for x in v) if isinstance(v, tuple) else Diff(v, NoChange)

    @staticmethod
    def unknown_change(v):
        return tuple(Diff(x, UnknownChange) for x in v) if isinstance(v, tuple) else Diff(v, UnknownChange)

    @staticmethod
    def tree_primal(v):
        if isinstance(v, Diff):
            return Diff.tree_primal(v.primal)
        if isinstance(v, (tuple, list)):
            return type(v)(Diff.tree_primal(x) for x in v)
        return v


# ---------------------------------------------------------------------------------------------
# Addresses.  The reference addresses a choice by a path of strings with at most one index component for a
# Scan/Vmap level: "x", ("sub", "x"), ("tracks", 3, "pos"), ("tracks", slice(None), "pos"), (3, "x")
# (choice_map.py:847-1395, scan.py:56-97).  Internally a site is keyed by ``name`` or ``(name, i)`` where
# ``name`` is a string or a tuple of strings and ``i`` the step / instance index.
# ---------------------------------------------------------------------------------------------
ALL = "<all indices>"
_ANY = "<any component>"


def norm_addr(addr):
    """user or internal address -> (name, idx); idx is None, an int, ALL (slice / Ellipsis), or — for a site inside nested
    combinators (a vmap inside a scan step, a scan inside a vmap instance, ...) — a tuple of ints / ALL, outermost first."""
    if not isinstance(addr, tuple) or addr == ():
        return addr, None
    names, idx = [], []

    def walk(c):
        if isinstance(c, str):
            names.append(c)
        elif isinstance(c, tuple):
            for e in c:
                walk(e)
        elif isinstance(c, slice) or c is Ellipsis:
            idx.append(ALL)
        elif isinstance(c, (int, np.integer)) and not isinstance(c, bool):
            idx.append(int(c))
        else:
            raise KeyError(f"unsupported address component {c!r} in {addr!r}")

    walk(addr)
    if not names:
        raise KeyError(f"address {addr!r} has no name component")
    if len(idx) > 3:
        raise KeyError(f"address {addr!r} has more than three index components")
    return (names[0] if len(names) == 1 else tuple(names)), (None if not idx else (idx[0] if len(idx) == 1 else tuple(idx)))


def _has_all(idx) -> bool:
    return idx is ALL or (isinstance(idx, tuple) and any(c is ALL for c in idx))


def key_of(addr):
    """canonical dictionary key of an address: name, or (name, i) for one step / instance ((name, (i, j)) nested); an
    address with a wildcard component stands for the whole sequence: name"""
    name, idx = norm_addr(addr)
    return name if idx is None or _has_all(idx) else (name, idx)


def _prefixes(name):
    """proper prefixes of a name path, as keys ("a" for ("a", "b"), ("a", "b") for ("a", "b", "c"))"""
    if not isinstance(name, tuple):
        return []
    return [name[0] if n == 1 else name[:n] for n in range(1, len(name))]


# ---------------------------------------------------------------------------------------------
# Selection
# ---------------------------------------------------------------------------------------------
class Selection:
    """Set of addresses (choice_map.py:124-361): ``S["x"] | S["y"]``, ``~sel``, ``Selection.all()``."""

    def __init__(self, addrs: Iterable[str] = (), complement: bool = False):
        self.addrs = frozenset(addrs)
        self.complement = bool(complement)

    @staticmethod
    def all() -> "Selection":
        return Selection((), True)

    @staticmethod
    def none() -> "Selection":
        return Selection((), False)

    @staticmethod
    def leaf() -> "Selection":
        """exactly the address ``()``; extended, exactly that path and nothing below it (choice_map.py:203-230)"""
        return Selection((("=", ()),))

    class _At:
        def __getitem__(self, addr) -> "Selection":
            if isinstance(addr, tuple) and addr == ():
                return Selection.leaf()
            comps = addr if isinstance(addr, tuple) else (addr,)
            if any(c is Ellipsis for c in comps) and any(isinstance(c, str) for c in comps):
                # S[..., "y"]: any ONE component, then "y" (choice_map.py:55-60).  A step / instance index is a component of
                # the reference's addresses (chm[3, "y"]), so the wildcard also stands for the index of a sequence "y" —
                # check() matches the pattern against the index-first form of an address; a bare "y" is NOT selected
                names = tuple(_ANY if c is Ellipsis else c for c in comps if c is Ellipsis or isinstance(c, str))
                return Selection((("*", names),))
            return Selection((key_of(addr),))

    at = _At()

    def check(self, addr=()) -> bool:
        """a site is selected by its own key, by its whole sequence (name), by any prefix of its path, by an exact-path
        entry equal to its path, or by a wildcard entry matching a prefix of its path; without an address: is ``()``"""
        if addr is None or (isinstance(addr, tuple) and addr == ()):
            hit = ("=", ()) in self.addrs
            return hit != self.complement if (self.addrs or not self.complement) else True
        name, idx = norm_addr(addr)
        path = name if isinstance(name, tuple) else (name,)
        hit = key_of(addr) in self.addrs or name in self.addrs or any(p in self.addrs for p in _prefixes(name))
        if not hit:
            for e in self.addrs:
                if isinstance(e, tuple) and len(e) == 2 and e[0] == "=" and isinstance(e[1], tuple):
                    hit = e[1] == path and e[1] != ()
                elif isinstance(e, tuple) and len(e) == 2 and e[0] == "*" and isinstance(e[1], tuple):
                    pat = e[1]
                    hit = len(path) >= len(pat) and all(pc is _ANY or pc == c for pc, c in zip(pat, path))
                    if not hit and idx is not None and not _has_all(idx):
                        # the reference's form of the same address carries the index components in front of one of the names
                        # (chm[3, "y"], chm["tracks", 1, "pos"]); only a wildcard matches an index
                        blk = (_ANY,) * (len(idx) if isinstance(idx, tuple) else 1)
                        for at in range(len(path)):
                            ipath = path[:at] + blk + path[at:]
                            if len(ipath) >= len(pat) and all(pc is _ANY or (c is not _ANY and pc == c) for pc, c in zip(pat, ipath)):
                                hit = True
                                break
                if hit:
                    break
        return hit != self.complement

    def prefixed(self, prefix) -> "Selection":
        """the same selection seen from an enclosing generative function that calls this one at ``prefix``"""
        pre = tuple(prefix) if isinstance(prefix, tuple) else (prefix,)

        def join(a):
            if isinstance(a, tuple) and len(a) == 2 and a[0] in ("=", "*") and isinstance(a[1], tuple):
                return (a[0], pre + a[1])                   # exact-path and wildcard entries move below the prefix
            name, idx = norm_addr(a)
            full = pre + (name if isinstance(name, tuple) else (name,))
            return full if idx is None or _has_all(idx) else (full, idx)
        if self.complement and not self.addrs:          # all() under a prefix = the prefix itself
            return Selection((pre[0] if len(pre) == 1 else pre,))
        if self.complement:
            raise NotImplementedError("complement selections cannot be re-rooted under a prefix")
        return Selection(join(a) for a in self.addrs)

    def __contains__(self, addr) -> bool:
        return self.check(addr)

    def __call__(self, *addr) -> "Selection":
        """the selection seen from below the path ``addr`` (choice_map.py:262-290): ``S["a", "b", "c"]("a")("b")["c"]``"""
        a = tuple(c for x in addr for c in (x if isinstance(x, tuple) else (x,)))
        if not a:
            return self
        out, everything = set(), False
        for e in self.addrs:
            if isinstance(e, tuple) and len(e) == 2 and e[0] == "=" and isinstance(e[1], tuple):
                if e[1][: len(a)] == a:
                    out.add(("=", e[1][len(a):]))
                continue
            if isinstance(e, tuple) and len(e) == 2 and e[0] == "*" and isinstance(e[1], tuple):
                pat = e[1]
                n = min(len(a), len(pat))
                if all(pc is _ANY or pc == c for pc, c in zip(pat[:n], a[:n])):
                    if len(a) >= len(pat):
                        everything = True
                    else:
                        out.add(("*", pat[len(a):]))
                continue
            name, idx = norm_addr(e)
            path = name if isinstance(name, tuple) else (name,)
            if path[: len(a)] == a and len(path) > len(a):
                rest = path[len(a):]
                rest = rest[0] if len(rest) == 1 else rest
                out.add(rest if idx is None or _has_all(idx) else (rest, idx))
            elif a[: len(path)] == path:                     # the entry is the path itself or one of its prefixes
                everything = True
        sub = Selection.all() if everything else Selection(out)
        return ~sub if self.complement else sub

    def __getitem__(self, addr) -> bool:
        """``sel["x"]``, ``sel["z", "y"]``: is the address selected (choice_map.py:262-290); a query holds no wildcard"""
        if addr is Ellipsis or (isinstance(addr, tuple) and any(c is Ellipsis for c in addr)):
            raise TypeError("a selection is queried with a concrete address: `...` is only meaningful when building one")
        return self.check(addr)

    def __eq__(self, other) -> bool:
        return isinstance(other, Selection) and self.addrs == other.addrs and self.complement == other.complement

    def __hash__(self) -> int:
        return hash((self.addrs, self.complement))

    def extend(self, *addrs) -> "Selection":
        """the same selection below the path ``addrs`` (choice_map.py:291-318); ``none`` stays ``none``"""
        if not self.addrs and not self.complement:
            return self
        return self.prefixed(tuple(addrs))

    def filter(self, chm: "ChoiceMap") -> "ChoiceMap":
        """``sel.filter(chm)`` == ``chm.filter(sel)`` (choice_map.py:319-361)"""
        return chm.filter(self)

    def __invert__(self) -> "Selection":
        return Selection(self.addrs, not self.complement)

    def __or__(self, other: "Selection") -> "Selection":
        if not self.complement and not other.complement:
            return Selection(self.addrs | other.addrs)
        if self.complement and other.complement:
            return Selection(self.addrs & other.addrs, True)
        c, p = (self, other) if self.complement else (other, self)
        return Selection(c.addrs - p.addrs, True)

    def __and__(self, other: "Selection") -> "Selection":
        return ~((~self) | (~other))

    def __repr__(self) -> str:
        return ("~" if self.complement else "") + "S" + repr(sorted(map(str, self.addrs)))


class _SelectionBuilder:
    def __getitem__(self, addr) -> Selection:
        return Selection.at[addr]

    @property
    def all(self) -> Selection:
        return Selection.all()

    @property
    def none(self) -> Selection:
        return Selection.none()

    @property
    def leaf(self) -> Selection:
        return Selection.leaf()


SelectionBuilder = S = _SelectionBuilder()


# ---------------------------------------------------------------------------------------------
# ChoiceMap
# ---------------------------------------------------------------------------------------------
class Masked:
    """``Mask(value, flag)`` (core/generative/functional_types.py:40-330): a value that is only meaningful where its flag is
    set.  A Python-bool flag is decided on the host (a constraint under ``Mask(v, False)`` is no constraint); a flag array
    holds one flag per particle and goes to the kernels as GJX_MODE_OBS_MASK (distribution.py:129-143)."""

    def __init__(self, value, flag=True):
        self.value = value
        self.flag = bool(flag) if isinstance(flag, (bool, np.bool_)) else np.asarray(_np(flag), bool)

    # -- the reference's constructors ---------------------------------------------------------
    @staticmethod
    def build(value, flag=True) -> "Masked":
        """flattens a nested mask: the flags combine with AND (functional_types.py `build`)"""
        if isinstance(value, Masked):
            return Masked(value.value, _flag_and(value.flag, flag if isinstance(flag, (bool, np.bool_)) else np.asarray(_np(flag), bool)))
        return Masked(value, flag)

    @staticmethod
    def maybe_mask(value, flag):
        """the bare value for a concrete True flag, None for a concrete False one, a mask otherwise"""
        if isinstance(flag, (bool, np.bool_)):
            if not flag:
                return None
            return value.unmask() if isinstance(value, Masked) and value.flag is True else value
        return Masked.build(value, flag)

    def primal_flag(self):
        return self.flag

    def unmask(self, default=None):
        """the value; invalid entries are replaced by ``default`` (without a default an invalid scalar mask raises)"""
        if isinstance(self.flag, bool):
            if self.flag:
                return self.value
            if default is None:
                raise ValueError("Mask.unmask: the mask is invalid and no default was given")
            return default
        if default is None:
            if not self.flag.all():
                raise ValueError("Mask.unmask: some entries are invalid and no default was given")
            return self.value
        v = _np(self.value)
        f = self.flag.reshape(self.flag.shape + (1,) * (v.ndim - self.flag.ndim))
        return np.where(f, v, _np(default))

    def __invert__(self) -> "Masked":
        return Masked(self.value, (not self.flag) if isinstance(self.flag, bool) else ~self.flag)

    def _pick(self, other: "Masked", flag):
        """value of ``self`` where it is valid, ``other``'s elsewhere"""
        if isinstance(self.flag, bool) and isinstance(other.flag, bool):
            return Masked(self.value if self.flag else other.value, flag)
        a, b = _np(self.value), _np(other.value)
        f = np.broadcast_to(np.asarray(self.flag), np.broadcast(np.asarray(self.flag), np.asarray(other.flag)).shape)
        fe = f.reshape(f.shape + (1,) * (max(a.ndim, b.ndim) - f.ndim))
        return Masked(np.where(fe, a, b), flag)

    def __or__(self, other: "Masked") -> "Masked":
        return self._pick(other, _flag_or(self.flag, other.flag))

    def __xor__(self, other: "Masked") -> "Masked":
        return self._pick(other, _flag_xor(self.flag, other.flag))

    def __getitem__(self, idx) -> "Masked":
        """index the value; a flag array is indexed by as many leading components as it has axes"""
        idx = idx if isinstance(idx, tuple) else (idx,)
        v = _np(self.value)[idx]
        if isinstance(self.flag, bool):
            return Masked(v, self.flag)
        fl = self.flag[idx[: self.flag.ndim]]
        return Masked(v, bool(fl) if np.ndim(fl) == 0 else fl)

    def __eq__(self, other) -> bool:
        return (isinstance(other, Masked) and np.array_equal(np.asarray(self.flag), np.asarray(other.flag))
                and np.array_equal(_np(self.value), _np(other.value)))

    __hash__ = None

    def __repr__(self):
        if isinstance(self.flag, bool):
            return f"Mask({self.value!r}, {self.flag})"
        return f"Mask({int(self.flag.sum())}/{self.flag.size} valid)"


def _flag_and(a, b):
    r = np.logical_and(a, b)
    return bool(r) if np.ndim(r) == 0 and isinstance(a, bool) and isinstance(b, (bool, np.bool_)) else np.asarray(r, bool)


def _flag_or(a, b):
    r = np.logical_or(a, b)
    return bool(r) if isinstance(a, bool) and isinstance(b, bool) else np.asarray(r, bool)


def _flag_xor(a, b):
    r = np.logical_xor(a, b)
    return bool(r) if isinstance(a, bool) and isinstance(b, bool) else np.asarray(r, bool)


Mask = Masked


class ChoiceMapNoValueAtAddress(KeyError):
    """choice_map.py:672"""


_VALUE = ()  # address of a bare value (C.v(x)), as in the reference's `()` address


def _rekey(prefix: tuple, key):
    """the canonical key of ``key`` seen below the path ``prefix`` (a tuple of strings)"""
    if isinstance(key, tuple) and key == ():
        name, idx = (), None
    else:
        name, idx = norm_addr(key)
        name = name if isinstance(name, tuple) else (name,)
    full = tuple(prefix) + name
    if not full:
        return _VALUE
    k = full[0] if len(full) == 1 else full
    return k if idx is None or _has_all(idx) else (k, idx)


def _flat_into(out: dict, prefix: tuple, value) -> None:
    """nested dictionaries and choice maps become path keys (choice_map.py `d` / `kw` / `from_mapping`: dict values are
    converted into choice maps)"""
    if isinstance(value, dict):
        for k, v in value.items():
            name, idx = norm_addr(k) if not (isinstance(k, tuple) and k == ()) else ((), None)
            if idx is None:
                _flat_into(out, tuple(prefix) + (name if isinstance(name, tuple) else (name,)), v)
            else:
                out[_rekey(prefix, k)] = v
    elif isinstance(value, ChoiceMap):
        for k, v in value._d.items():
            out[_rekey(prefix, k)] = v
    else:
        out[_rekey(prefix, ())] = value


class ChoiceMap:
    """Static-address choice map: ``{addr: value}`` (choice_map.py:847-1395, Static 1535)."""

    def __init__(self, entries: dict | None = None, lead_axes: int = 0):
        self._d: dict = dict(entries or {})
        self._lead_axes = int(lead_axes)   # stacked values of a batched trace carry the particle axis in front of the indices

    # -- builders (ChoiceMapBuilder C) ------------------------------------------------------
    @staticmethod
    def empty() -> "ChoiceMap":
        return ChoiceMap()

    n = empty

    @staticmethod
    def kw(**kwargs) -> "ChoiceMap":
        return ChoiceMap.d(kwargs)

    @staticmethod
    def d(entries: dict) -> "ChoiceMap":
        """nested dictionaries (and choice maps) as values nest below their key (choice_map.py:1023-1056)"""
        out: dict = {}
        _flat_into(out, (), entries)
        return ChoiceMap(out)

    @staticmethod
    def from_mapping(pairs) -> "ChoiceMap":
        """``[(addr, value), ...]`` with string or tuple addresses (choice_map.py:1057-1085)"""
        out: dict = {}
        for a, v in pairs:
            _flat_into(out, a if isinstance(a, tuple) else (a,), v)
        return ChoiceMap(out)

    @staticmethod
    def v(value) -> "ChoiceMap":
        """a bare value (choice_map.py `choice`): a concrete-False mask or an empty array is no choice at all, a concrete-True
        mask is its value"""
        if isinstance(value, Masked) and isinstance(value.flag, bool):
            return ChoiceMap.v(value.value) if value.flag else ChoiceMap()
        if not isinstance(value, (Masked, dict, ChoiceMap)) and value is not None and hasattr(value, "shape") and int(np.prod(value.shape)) == 0:
            return ChoiceMap()
        return ChoiceMap({_VALUE: value})

    choice = v

    class _AtSetter:
        def __init__(self, base: "ChoiceMap", addr):
            self.base, self.addr = base, addr

        def _path(self) -> tuple | None:
            """the address as a pure path of strings, or None when it carries an index"""
            if isinstance(self.addr, tuple) and self.addr == ():
                return ()
            name, idx = norm_addr(self.addr)
            return None if idx is not None else (name if isinstance(name, tuple) else (name,))

        def set(self, value) -> "ChoiceMap":
            d = dict(self.base._d)
            if isinstance(value, ChoiceMap) and value.has_value():
                value = value.get_value()
            comps = self.addr if isinstance(self.addr, tuple) else (self.addr,)
            if any(isinstance(c, slice) and c != slice(None) for c in comps):
                raise ValueError("a choice map is set at whole sequences ([:]) or single indices, not at partial slices")
            if comps and not any(isinstance(c, str) for c in comps) and isinstance(value, (dict, ChoiceMap)):
                # C[:].set({"x": xs}) / C[0].set({"x": 1.0}): the index applies to every address of the value
                flat: dict = {}
                _flat_into(flat, (), value)
                for k, v in flat.items():
                    name, idx = norm_addr(k)
                    if idx is not None:
                        raise KeyError(f"{k!r} already carries an index")
                    d[key_of(tuple(comps) + (name if isinstance(name, tuple) else (name,)))] = v
                return ChoiceMap(d, self.base._lead_axes)
            pre = self._path()
            if pre is not None and isinstance(value, (dict, ChoiceMap)):
                _flat_into(d, pre, value)              # a choice map / dict nests below the address
            else:
                d[key_of(self.addr) if self.addr != () else _VALUE] = value
            return ChoiceMap(d, self.base._lead_axes)

        def v(self, value) -> "ChoiceMap":
            return self.set(value)

        def n(self) -> "ChoiceMap":
            return ChoiceMap.empty()

        def d(self, entries: dict) -> "ChoiceMap":
            return self.set(ChoiceMap.d(entries))

        def kw(self, **kwargs) -> "ChoiceMap":
            return self.set(ChoiceMap.d(kwargs))

        def from_mapping(self, pairs) -> "ChoiceMap":
            return self.set(ChoiceMap.from_mapping(pairs))

        def update(self, f) -> "ChoiceMap":
            """replace what sits at the address by ``f`` of it: of the value for a leaf, of the sub-map otherwise
            (choice_map.py builder `update`); an empty spot hands ``f`` the empty choice map"""
            k = key_of(self.addr)
            if k in self.base._d:
                return self.set(f(self.base._d[k]))
            sub = self.base.get_submap(self.addr)
            pre = self._path() or ()
            keep = {a: v for a, v in self.base._d.items() if not (a != _VALUE and _starts_with(a, pre))}
            return ChoiceMap._AtSetter(ChoiceMap(keep, self.base._lead_axes), self.addr).set(f(sub))

    class _At:
        def __init__(self, base: "ChoiceMap"):
            self.base = base

        def __getitem__(self, addr) -> "ChoiceMap._AtSetter":
            if isinstance(addr, tuple) and len(addr) == 1:
                addr = addr[0]
            return ChoiceMap._AtSetter(self.base, addr)

    @property
    def at(self) -> "ChoiceMap._At":
        return ChoiceMap._At(self)

    # -- queries ------------------------------------------------------------------------------
    def has_value(self) -> bool:
        return _VALUE in self._d

    def get_value(self):
        return self._d.get(_VALUE)

    def static_is_empty(self) -> bool:
        return not self._d

    def __contains__(self, addr) -> bool:
        if isinstance(addr, tuple) and addr == ():
            return self.has_value()
        try:
            name, idx = norm_addr(addr)
        except KeyError:
            return False
        return key_of(addr) in self._d or (idx is not None and not _has_all(idx) and name in self._d)

    def _n_steps(self, name) -> int:
        return sum(1 for k in self._d if isinstance(k, tuple) and len(k) == 2 and k[0] == name and isinstance(k[1], int))

    def __getitem__(self, addr):
        # chm["x"], chm["sub", "x"], chm[t, "x"] / chm["tracks", t, "pos"] (one step), chm[:, "x"] (stacked);
        # nested combinators: chm[i, t, "x"] (one instance, one step), chm[:, :, "x"] (stacked over both), chm[i, :, "x"]
        if isinstance(addr, tuple) and any(isinstance(c, slice) and c != slice(None) for c in addr):
            # a partial slice of a sequence (chm[0:4, "x"], choice_map.py slices): the whole sequence, then the slice on the
            # axis of that index component
            full = tuple(slice(None) if isinstance(c, slice) else c for c in addr)
            whole = self[full]
            cut = tuple(c for c in addr if isinstance(c, slice) or c is Ellipsis)
            return whole[(slice(None),) * self._lead_axes + tuple(slice(None) if c is Ellipsis else c for c in cut)]
        name, idx = norm_addr(addr)
        if isinstance(idx, tuple):
            if not _has_all(idx):
                if (name, idx) in self._d:
                    return self._d[(name, idx)]
                if name in self._d:                   # a whole-array value set by the user: leading axes = the indices
                    return self._d[name][idx]
                raise ChoiceMapNoValueAtAddress(addr)
            if name not in self._d:
                raise ChoiceMapNoValueAtAddress(addr)
            whole = self._d[name]
            if all(c is ALL for c in idx):
                return whole
            lead = self._lead_axes                    # a batched trace's stacked values carry the particle axis in front
            return whole[(slice(None),) * lead + tuple(slice(None) if c is ALL else c for c in idx)]
        if isinstance(idx, int):
            if idx < 0:
                n = self._n_steps(name)
                idx = idx + n if n else idx
            if (name, idx) in self._d:
                return self._d[(name, idx)]
            if name in self._d:                       # a whole-sequence value set by the user: leading axis = steps
                return self._d[name][idx]
            raise ChoiceMapNoValueAtAddress(addr)
        if name not in self._d:
            raise ChoiceMapNoValueAtAddress(addr)
        return self._d[name]

    def get(self, addr, default=None):
        try:
            return self[addr]
        except (KeyError, IndexError):
            return default

    def __call__(self, addr):
        return self.get_submap(addr)

    def get_submap(self, *addr) -> "ChoiceMap":
        """value at a leaf address, or the choices below a path prefix with the prefix removed; the path may be given as one
        tuple or splatted (choice_map.py:1170-1202)"""
        addr = addr[0] if len(addr) == 1 else tuple(addr)
        if isinstance(addr, tuple) and addr == ():
            return self
        k = key_of(addr)
        if k in self._d:
            return ChoiceMap.v(self._d[k])
        pre = k if isinstance(k, tuple) and all(isinstance(c, str) for c in k) else (k,)
        out = {}
        for a, v in self._d.items():
            name, idx = norm_addr(a)
            path = name if isinstance(name, tuple) else (name,)
            if a != _VALUE and len(path) > len(pre) and path[: len(pre)] == pre:
                rest = path[len(pre):]
                rest = rest[0] if len(rest) == 1 else rest
                out[rest if idx is None else (rest, idx)] = v
        return ChoiceMap(out)

    def addresses(self) -> list:
        return [a for a in self._d if a != _VALUE]

    def items(self):
        return [(a, v) for a, v in self._d.items() if a != _VALUE]

    def __len__(self) -> int:
        return len(self._d)

    # -- algebra ------------------------------------------------------------------------------
    def merge(self, other: "ChoiceMap") -> "ChoiceMap":
        """``self | other`` — left-biased union (choice_map.py:1227-1251)."""
        d = dict(other._d)
        d.update(self._d)
        return ChoiceMap(d, max(self._lead_axes, getattr(other, "_lead_axes", 0)))

    __or__ = merge

    def __xor__(self, other: "ChoiceMap") -> "ChoiceMap":
        """disjoint union: the two maps may not hold a value at the same address (choice_map.py Xor)"""
        both = set(self._d) & set(other._d)
        if both:
            raise ValueError(f"ChoiceMap ^: both sides hold a value at {sorted(map(str, both))}")
        return self.merge(other)

    def __and__(self, other: "ChoiceMap") -> "ChoiceMap":
        """the addresses both maps hold, with the RIGHT side's values (choice_map.py And)"""
        return ChoiceMap({a: v for a, v in other._d.items() if a in self._d}, max(self._lead_axes, other._lead_axes))

    def extend(self, *addrs) -> "ChoiceMap":
        """the same choices below the path ``addrs`` (choice_map.py:1203-1226)"""
        return ChoiceMap({_rekey(tuple(addrs), a): v for a, v in self._d.items()}, self._lead_axes)

    def invalid_subset(self, gen_fn, args: tuple):
        """the choices of this map at addresses ``gen_fn(*args)`` does not produce, or None when every address is one of the
        function's (choice_map.py `invalid_subset`; a missing address is fine).  A whole-sequence entry (``C["x"].set(xs)``)
        is valid when the function has sites ``("x", i)``."""
        sl, _ = gen_fn.site_list(tuple(args))
        known = set()
        for st in sl.sites:
            known.add(st.addr)
            name, idx = norm_addr(st.addr) if st.addr != _VALUE else (_VALUE, None)
            known.add(name)
        bad = {a: v for a, v in self._d.items() if a not in known}
        return ChoiceMap(bad, self._lead_axes) if bad else None

    def simplify(self) -> "ChoiceMap":
        """this representation is always flat: nothing to push down (choice_map.py `simplify`)"""
        return self

    def mask(self, flag) -> "ChoiceMap":
        """``chm.mask(flag)`` (choice_map.py Mask).  A scalar flag keeps or drops the choices on the host; a flag per
        particle (length-K array) wraps every value in ``Masked``: the kernels then apply the constrained rule to the
        flagged particles and sample the others (distribution.py:129-143)."""
        f = np.asarray(flag.detach().cpu() if hasattr(flag, "detach") else flag)
        if f.ndim == 0:
            return self if bool(f) else ChoiceMap()
        if f.ndim != 1:
            raise ValueError("ChoiceMap.mask: the flag is a scalar or one boolean per particle")
        return ChoiceMap({a: Masked(v, f.astype(bool)) for a, v in self._d.items()})

    def filter(self, selection: Selection) -> "ChoiceMap":
        return ChoiceMap({a: v for a, v in self._d.items() if selection.check(a)}, self._lead_axes)

    def get_selection(self) -> Selection:
        return Selection(self._d.keys())

    def __eq__(self, other) -> bool:
        if not isinstance(other, ChoiceMap) or set(self._d) != set(other._d):
            return False
        return all(np.array_equal(_np(self._d[a]), _np(other._d[a])) for a in self._d)

    def __repr__(self) -> str:
        return "ChoiceMap(" + ", ".join(f"{a!r}: {_short(v)}" for a, v in self._d.items()) + ")"


def _starts_with(key, pre: tuple) -> bool:
    name, _ = norm_addr(key)
    path = name if isinstance(name, tuple) else (name,)
    return len(path) > len(pre) and path[: len(pre)] == tuple(pre)


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def _short(v: Any) -> str:
    a = _np(v)
    return repr(a.item()) if a.ndim == 0 else f"array{a.shape}"


class _ChoiceMapBuilder:
    """``C["y"].set(v)``, ``C.kw(y=3.0)``, ``C.d({...})``, ``C.v(x)``, ``C.n()`` (choice_map.py builders)."""

    def __getitem__(self, addr):
        return ChoiceMap().at[addr]

    kw = staticmethod(ChoiceMap.kw)
    d = staticmethod(ChoiceMap.d)
    v = staticmethod(ChoiceMap.v)
    n = staticmethod(ChoiceMap.empty)
    choice = staticmethod(ChoiceMap.v)
    from_mapping = staticmethod(ChoiceMap.from_mapping)


ChoiceMapBuilder = C = _ChoiceMapBuilder()
