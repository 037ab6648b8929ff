"""Loader for the HIP library.  There is NO CPU fallback: if ``libgjx_hip.so`` is missing or does
not export the ABI of ``include/gjx.h`` the import of any compute entry point fails loudly."""
from __future__ import annotations

import ctypes
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgjx_hip.so")

_lib = None


class GjxError(RuntimeError):
    """Non-zero gjx_status returned by the HIP library."""


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        # torch first: libgjx_hip.so needs libamdhip64.so.7 and must share the HIP runtime torch
        # already loaded (one runtime per process, so torch device pointers and streams are ours too)
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise GjxError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C genjax_amd/csrc). "
                "genjax_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        A.bind(lib)
        v = lib.gjx_version()
        if v != A.ABI_VERSION:
            raise GjxError(f"libgjx_hip.so ABI version {v}, expected {A.ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gjx_last_error().decode("utf-8", "replace")
        raise GjxError(f"{what or 'gjx call'} failed with status {rc}: {msg}")
