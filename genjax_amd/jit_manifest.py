"""Manifest of the generated kernels a deployment precompiles.

Generated kernels (csrc/gjx_codegen.hip) are compiled by hipRTC the first time a program structure is seen and cached
on disk by the hash of their source (``csrc/jit_cache/`` next to the library).  A compile at run time costs 0.3 - 3 s and
drops the GPU's clocks while it runs, so a build step walks THIS list — ``precompile_all()``, called by
``__graft_entry__.build()``; hipRTC cross-compiles for gfx950 without a GPU — and ``gjx_jit_stats`` (``kernels.jit_stats``)
counts what was still compiled at run time (``bench.py`` prints it as ``jit_compiles_at_runtime``).

An entry = (name, factory of the PackedProgram, kind, variants):
    kind "run"     gjx_gen            variants = particles-per-lane codes (1, 2, 4; | 256 matrix-core flavour)
    kind "filter"  gjx_gen_pf         variants = tiles per block
    kind "hmc"     gjx_hmc_gen        variants = (None,)
A user's own models join with ``register(name, factory, kind, variants)`` before ``precompile_all()``.
"""
from __future__ import annotations

import sys
from typing import Callable, List, Tuple

from . import workloads

Entry = Tuple[str, Callable[[], object], str, tuple]

MANIFEST: List[Entry] = [
    # config 2: the mixture target's generated kernel (every particles-per-lane variant the run-time picker may choose)
    ("gmm_c8_d16", lambda: workloads.gmm_program()[0], "run", (1, 2, 4)),
    # config 5's model as an ImportanceK target: scalar forms and the matrix-core flavour of its big affine site
    ("hier_logreg_importance", lambda: workloads.logreg_importance_program()[0], "run", (1, 2, 4, 1 | 256)),
    # config 3's model written as @gen + .scan: the filter kernel on the shared skeleton, one tile per block (K = 2^18) and four (2^20)
    ("lgssm_scan_step", lambda: workloads.lgssm_scan_step_program(), "filter", (1, 4)),
    # config 5: the generated HMC kernel
    ("hier_logreg_hmc", lambda: workloads.logreg_program()[0], "hmc", (None,)),
]


def register(name: str, factory: Callable[[], object], kind: str, variants: tuple) -> None:
    assert kind in ("run", "filter", "hmc")
    MANIFEST.append((name, factory, kind, tuple(variants)))


def precompile_all(verbose: bool = True) -> dict:
    """Compile every entry into the on-disk cache.  -> {"compiled": n, "cached": n, "skipped": [(name, variant, why)]}.
    An entry the emitter (or a box without hipRTC) refuses is reported, not fatal: at run time such a program takes the
    hand-fused or interpreter engine."""
    from . import kernels
    skipped = []
    before = kernels.jit_stats()
    for name, factory, kind, variants in MANIFEST:
        try:
            prog = factory()
        except Exception as e:                              # noqa: BLE001
            skipped.append((name, None, repr(e)))
            continue
        for v in variants:
            try:
                if kind == "run":
                    kernels.program_precompile(prog, v)
                elif kind == "filter":
                    kernels.program_filter_precompile(prog, v)
                else:
                    kernels.program_hmc_precompile(prog)
            except Exception as e:                          # noqa: BLE001
                skipped.append((name, v, str(e)[:200]))
    after = kernels.jit_stats()
    out = dict(compiled=after["hiprtc_compiles"] - before["hiprtc_compiles"], cached=after["disk_hits"] - before["disk_hits"],
               skipped=skipped)
    if verbose:
        print(f"[jit_manifest] {len(MANIFEST)} entries: {out['compiled']} compiled, {out['cached']} already cached"
              + (f", {len(skipped)} skipped: {skipped}" if skipped else ""), file=sys.stderr)
    return out
