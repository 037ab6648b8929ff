"""Runtime knobs (the reference has none beyond static dataclass fields)."""
from __future__ import annotations

import os

from . import _abi as A

_RNG = {"flat": A.RNG_FLAT, "jax32": A.RNG_JAX32, "jax": A.RNG_JAX32}
_override: int | None = None


def rng_mode() -> int:
    """Random-stream layout for new programs: env GJX_RNG = flat (default) | jax32."""
    if _override is not None:
        return _override
    return _RNG.get(os.environ.get("GJX_RNG", "flat").lower(), A.RNG_FLAT)


def set_rng_mode(mode) -> None:
    global _override
    _override = None if mode is None else (_RNG[mode.lower()] if isinstance(mode, str) else int(mode))
