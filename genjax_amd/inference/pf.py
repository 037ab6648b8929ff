"""Bootstrap particle filter for a linear-Gaussian state-space model (BASELINE configs 3 and 4) and
N-of-K resampling.  The reference library has neither: it supplies the ingredients
(Scan.generate's chained keys and per-step weights, scan.py:237-294; the cookbook's categorical-draw +
gather idiom) and closed-form answers come from a Kalman filter.  Per step and per particle the HIP
path does: resample-index (prefix sum + comb search) then ONE fused kernel (ancestor gather +
propagate + reweight + LSE partials).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _abi as A
from .._lib import GjxError
from .. import config
from ..core import Key, fold_in, split, threefry2x32


_warned_timeout = [False]


def _note_timeout(what: str) -> None:
    if not _warned_timeout[0]:
        import warnings
        warnings.warn(f"{what}: a co-resident kernel timed out waiting for its peer blocks (another kernel holds compute units); "
                      "the call was repeated on the multi-launch path (logged once)")
        _warned_timeout[0] = True


def _unit_from_key(k: Key) -> float:
    """uniform [0,1) from one key: 23 mantissa bits of x0^x1 of Threefry(k, (0,0)) (jax _uniform)."""
    a, b = threefry2x32(k[0], k[1], 0, 0)
    bits = (a ^ b) >> 9
    return float(bits) / float(1 << 23)


def resample(rows: torch.Tensor, logw: torch.Tensor, key: Key, method: str = "systematic", lse=None, n_out=None,
             check: bool = False, collection=None, weights: str | None = None):
    """N-of-K resampling of SoA rows by log-weights.  -> (new_rows, ancestors int32[N]).
    ``weights`` picks the fixed-point scheme of include/gjx.h; default: ``"global_max"`` (weights -> ancestors -> children in
    one co-resident launch) up to K = 2^20, ``"tile_scaled"`` beyond, where that launch no longer fits the device and the
    global-maximum scheme would take three launches (K = 2^21: 104 vs 127 us per ImportanceK step).
    ``weights="tile_scaled"`` (systematic, N = K): the tile-scaled fixed point of include/gjx.h through
    gjx_resample_gather_tiled — ONE plain launch at any K (no co-resident grid, nothing to time out), the tile totals
    taken from the collection's producing run when it left them.
    check=True reads the status word of the co-resident kernel afterwards (one stream synchronisation): raises GjxError
    on a time-out (grid not co-resident: results undefined) or a dead collection (all weights zero).
    ``collection``: the ParticleCollection the weights come from — if its producing run's block partials are still in
    place, the resampling kernel reduces them in its prologue and hands the finished LSE record back to the collection
    (no separate LSE launch anywhere on the path)."""
    from .. import kernels
    K = logw.numel()
    N = int(n_out or K)
    if weights is None:
        weights = "tile_scaled" if (K > (1 << 20) and method == "systematic" and N == K and rows.stride(1) == 1) else "global_max"
    if weights not in ("global_max", "tile_scaled"):
        raise ValueError("weights must be 'global_max' or 'tile_scaled'")
    if weights == "tile_scaled":
        if method != "systematic" or N != K or rows.stride(1) != 1:
            raise ValueError("weights='tile_scaled' resamples N = K particles systematically from contiguous rows")
        part = collection.lse_partials() if (collection is not None and lse is None) else None
        ws = kernels.shared_workspace(A.OP_RESAMPLE, K, logw.device)
        anc = torch.empty(K, dtype=torch.int32, device=logw.device)
        lse_out = torch.empty(4, dtype=torch.float32, device=logw.device) if (part is not None and collection.K_total == K) else None
        out = kernels.resample_gather_tiled(logw, _unit_from_key(key), rows, partials=None if lse_out is None else part.as_arg(),
                                            tiles=0 if lse_out is None else part.tiles, lse_out=lse_out, K_total=K, anc=anc, ws=ws)
        if check and kernels.workspace_status(ws, raise_on_error=False) & 2:
            raise GjxError("resample: all weights are zero / -inf / NaN: nothing to resample from")
        if lse_out is not None:
            collection._lse = lse_out
        return out, anc
    part = collection.lse_partials() if (collection is not None and lse is None and method == "systematic") else None
    lse_out = None
    if part is not None and collection.K_total == K:
        lse_out = torch.empty(4, dtype=torch.float32, device=logw.device)
    else:
        part = None
        if lse is None:
            lse = collection.lse() if collection is not None else kernels.logsumexp(logw, ws=kernels.shared_workspace(A.OP_LSE, K, logw.device))
    if method == "systematic":
        ws = kernels.shared_workspace(A.OP_RESAMPLE, K, logw.device)
        pa = None if part is None else part.as_arg()
        if N == K and rows.stride(1) == 1:
            # weights -> ancestors -> children in one launch (falls back to two beyond the co-resident grid)
            anc = torch.empty(K, dtype=torch.int32, device=logw.device)
            out = kernels.resample_gather(logw, _unit_from_key(key), rows, True, lse, partials=pa, lse_out=lse_out, K_total=K, anc=anc, ws=ws)
        else:
            # weights -> ancestors in one launch, then the slot-oriented row copy
            anc = kernels.resample_indices(logw, _unit_from_key(key), N, True, lse, partials=pa, lse_out=lse_out, K_total=K, ws=ws)
            out = kernels.gather_rows(rows, anc)
        if check:
            st = kernels.workspace_status(ws, raise_on_error=False)
            if st & 1:
                # the grid was not co-resident (results undefined): the same resampling as plain launches — prefix sums,
                # slot-run expansion, row gather: identical integers, identical ancestors
                _note_timeout("resample")
                cum, bt = kernels.weight_cumsum(logw, True, lse, ws=ws, partials=pa, lse_out=lse_out, K_total=K)
                anc = kernels.resample_systematic(cum, bt, _unit_from_key(key), N, prefill=False)
                out = kernels.gather_rows(rows, anc)
                st = kernels.workspace_status(ws, raise_on_error=False)
            if st & 2:
                raise GjxError("resample: all weights are zero / -inf / NaN: nothing to resample from")
        if lse_out is not None:
            collection._lse = lse_out
        return out, anc
    cum, bt = kernels.weight_cumsum(logw, True, lse, ws=kernels.shared_workspace(A.OP_RESAMPLE, K, logw.device))
    if method == "multinomial":
        anc = kernels.resample_multinomial(cum, bt, key, N)
        return kernels.gather_rows(rows, anc), anc
    raise ValueError(method)


class LinearGaussianSSM:
    """x_0 ~ N(0, q0^2 I); x_t ~ N(A x_{t-1}, q^2 I); y_t ~ N(H x_t, r^2 I)."""

    def __init__(self, A_mat, q: float, r: float, H=None, q0: float = 1.0):
        self.A = np.ascontiguousarray(A_mat, np.float32)
        self.H = None if H is None else np.ascontiguousarray(H, np.float32)
        self.q, self.r, self.q0 = float(q), float(r), float(q0)
        self.dx = self.A.shape[0]
        self.dy = self.dx if self.H is None else self.H.shape[0]
        self._dev = None

    def c_struct(self, device) -> A.GjxSsm:
        if self._dev is None or self._dev[0].device != torch.device(device):
            Ad = torch.from_numpy(self.A).to(device)
            Hd = None if self.H is None else torch.from_numpy(self.H).to(device)
            self._dev = (Ad, Hd)
        s = A.GjxSsm()
        s.dx, s.dy, s.q, s.r, s.q0 = self.dx, self.dy, self.q, self.r, self.q0
        s.A_dev = self._dev[0].data_ptr()
        s.H_dev = None if self._dev[1] is None else self._dev[1].data_ptr()
        return s


class ParticleHistory:
    """Append-only record of a filter run: the particles of every step (SoA rows) and the ancestor indices of every
    resampling.  The reference's ScanTrace keeps the whole stacked trace per particle and its resampling idiom gathers
    whole particles (scan.py:56-97; O(T^2) copies over a run); here a trajectory is reconstructed lazily by following
    the ancestors back from the last step — one row gather per step."""

    def __init__(self):
        self.states: list = []      # x_t f32[dx][K], the particles as propagated at step t (before the next resampling)
        self.ancestors: list = []   # a_t i32[K] for t >= 1: particle i of step t descends from particle a_t[i] of step t-1
        self.logw = None            # log-weights of the last step

    def append(self, x, anc):
        self.states.append(x.clone())
        if anc is not None:
            self.ancestors.append(anc.clone())

    def replace_parents(self, x_moved, anc):
        """Resample-move: the particles of the current step descend from MOVED copies of the previous step's particles.
        ``x_moved`` f32[dx][K] are those moved parents in the current step's slot order and ``anc`` the ancestors the
        resampling picked for the slots: the stored previous step becomes the moved parents, its own ancestor links are
        composed with ``anc`` (entry i now descends from what particle anc[i] descended from), and the current step is
        then appended with the identity as its ancestors."""
        from .. import kernels
        self.states[-1] = x_moved.clone()
        if self.ancestors:
            a = self.ancestors[-1]
            self.ancestors[-1] = kernels.gather_rows(a.view(torch.float32).reshape(1, -1), anc).reshape(-1).view(torch.int32).contiguous()

    def __len__(self):
        return len(self.states)

    def paths(self, idx=None):
        """-> f32[T][dx][n]: the trajectories that end in particles ``idx`` (int32 device tensor; default all) of the last step"""
        import torch
        from .. import kernels
        T = len(self.states)
        K = self.states[-1].shape[1]
        cur = torch.arange(K, dtype=torch.int32, device=self.states[-1].device) if idx is None else idx.to(torch.int32).contiguous()
        out = [None] * T
        for t in range(T - 1, -1, -1):
            out[t] = kernels.gather_rows(self.states[t], cur)
            if t > 0:
                a = self.ancestors[t - 1]
                cur = kernels.gather_rows(a.view(torch.float32).reshape(1, -1), cur).reshape(-1).view(torch.int32).contiguous()
        return torch.stack(out)

    def smoothed_means(self):
        """E[x_t | y_{1:T}] estimated from the reconstructed trajectories, weighted by the last step's weights: f32[T][dx]"""
        import torch
        w = torch.softmax(self.logw.double(), dim=0)
        return (self.paths().double() * w).sum(dim=2).float()


class BootstrapFilter:
    """SMC with the prior as proposal and systematic resampling before every propagate step.
    ``BootstrapFilter(LinearGaussianSSM(...), K)``: the hand-written kernels of BASELINE configs 3 / 4;
    ``BootstrapFilter(kernel.scan(n=T), K)``: any Scan kernel (inference/scan_filter.py, gjx_scan_filter);
    ``BootstrapFilter(model, K)`` with a ``@gen`` model whose body draws static parameters and then calls ONE ``kernel.scan(n=T)(...)``:
    the parameters are drawn with step 0 and travel with the particles (``run(key, constraint, model_args)``)."""

    def __new__(cls, model, *args, **kwargs):
        from ..gen import ScanCombinator, StaticGenerativeFunction
        if isinstance(model, (ScanCombinator, StaticGenerativeFunction)):      # (a @gen model whose body is sites in front of ONE Scan)
            from .scan_filter import ScanBootstrapFilter
            return ScanBootstrapFilter(model, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, ssm: LinearGaussianSSM, k_particles: int, rng_mode: int | None = None, rejuvenate: dict | None = None,
                 weights: str | None = None):
        """``rejuvenate=dict(n_moves=.., scale=..)``: resample-move — after every resampling each particle takes n_moves
        random-walk Metropolis steps (proposal scale ``scale``) that leave p(x_{t-1} | parent, y_{t-1}) invariant, fused
        into the propagate kernel (gjx_ssm_step_move).
        ``weights``: fixed-point scheme of the systematic resampler (include/gjx.h): ``"global_max"`` quantises every
        weight against the exact global maximum, ``"tile_scaled"`` against a power of two per 1024-particle tile — one
        grid-wide exchange per step instead of two in the one-launch filter, resample-move inside the launch, and the
        peer-mapped exchange when sharded.  Default (None): tile-scaled wherever that path exists, i.e. everywhere except a
        sharded run that cannot use the peer-mapped exchange (keep_means / step_by_step, ranks that are not peers, shards that
        are not whole tiles), which runs the global-maximum scheme over the collective transport."""
        if weights not in (None, "global_max", "tile_scaled"):
            raise ValueError("weights must be 'global_max' or 'tile_scaled'")
        self._auto_weights = weights is None
        self.weights = A.WEIGHTS_GLOBAL_MAX if weights == "global_max" else A.WEIGHTS_TILE_SCALED
        self.ssm, self.K = ssm, int(k_particles)
        self.rng_mode = config.rng_mode() if rng_mode is None else rng_mode
        self.rejuvenate = dict(rejuvenate) if rejuvenate else None
        self.last_accept_rate = None

    def _run_peer(self, key, ys_d, dev, world, check_status):
        from .. import kernels
        K_local = self.K // world
        k_ = (K_local, self.ssm.dx, world, str(dev))
        if getattr(self, "_peer_key", None) != k_:
            if getattr(self, "_peer", None) is not None:
                self._peer.close()
            self._peer, self._peer_key = kernels.PeerContext(K_local, self.ssm.dx, dev), k_
        mv = None
        if self.rejuvenate:
            mv = (int(self.rejuvenate.get("n_moves", 1)), float(self.rejuvenate.get("scale", 0.5)))
        out = self._peer.ssm_filter(self.ssm.c_struct(dev), key, self.rng_mode, ys_d, move=mv)
        incs = out["lse_steps"][:, 3]
        res = dict(log_ml=incs.sum(), increments=incs, x=out["x"], logw=out["logw"], means=None, transport="peer")
        if mv is not None:
            acc = out["accepted_total"].double()
            if world > 1:
                torch.distributed.all_reduce(acc)
            self.last_accept_rate = float(acc[0]) / (self.K * (ys_d.shape[0] - 1) * max(mv[0], 1))
        if check_status:
            st = self._peer.status()
            if st & 1:
                raise GjxError("sharded bootstrap filter: a rendezvous between the ranks timed out; results are undefined")
            res["degenerate"] = bool(st & 2)
        return res

    def close(self):
        """release the exchange contexts of a sharded filter (collective: every rank calls it)"""
        if getattr(self, "_peer", None) is not None:
            self._peer.close()
            self._peer, self._peer_key = None, None
        if getattr(self, "_resampler", None) is not None:
            self._resampler.close()
            self._resampler, self._resampler_key = None, None

    @staticmethod
    def _checked(res, ws, check_status):
        """Read (and clear) the status word of the co-resident resampling kernels: a time-out means the grid was not
        co-resident and the results are undefined (raises); a step whose weights were all zero kept its particles and
        the run is flagged ``degenerate``.  One stream synchronisation per run."""
        from .. import kernels
        if check_status and ws is not None:
            st = kernels.workspace_status(ws, raise_on_error=False)
            if st & 1:
                raise GjxError("bootstrap filter: a co-resident resampling kernel timed out waiting for its peers; results are undefined")
            res["degenerate"] = bool(st & 2)
        return res

    def run(self, key: Key, ys, device=None, rank: int = 0, world: int = 1, keep_means: bool = False,
            step_by_step: bool = False, keep_history: bool = False, check_status: bool = True):
        """-> dict(log_ml 0-d device tensor, increments f32[T], x f32[dx][K_local], logw, means?, degenerate?).
        With world > 1 the K particles are sharded (distributed.py) and ``ys`` is the same on all ranks."""
        from .. import kernels
        from .. import distributed as D
        if self._auto_weights and (world > 1 or D._forced()):
            # default scheme, sharded: tile-scaled only where the peer-mapped filter can run (the test is rank-symmetric)
            dev_ = kernels._dev(device)
            T_ = ys.shape[0] if hasattr(ys, "shape") else len(ys)
            peer_ok = (not keep_means and not step_by_step and not keep_history and T_ >= 2 and self.K % world == 0
                       and (self.K // world) % 1024 == 0 and D.peer_available(dev_))
            if not peer_ok and not self.rejuvenate:
                self.weights = A.WEIGHTS_GLOBAL_MAX
                try:
                    return self._run(key, ys, device, rank, world, keep_means, step_by_step, keep_history, check_status)
                finally:
                    self.weights = A.WEIGHTS_TILE_SCALED
        return self._run(key, ys, device, rank, world, keep_means, step_by_step, keep_history, check_status)

    def _run(self, key: Key, ys, device, rank, world, keep_means, step_by_step, keep_history, check_status):
        from .. import kernels
        from .. import distributed as D
        dev = kernels._dev(device)
        ys_d = torch.as_tensor(np.asarray(ys, np.float32), device=dev) if not torch.is_tensor(ys) else ys.to(dev)
        ys_d = ys_d.contiguous()
        T = ys_d.shape[0]
        if keep_history and (world > 1 or D._forced()):
            raise NotImplementedError("keep_history: the ancestor history is kept per process; run the filter on one GPU")
        if self.rejuvenate and (world > 1 or D._forced()) and self.weights != A.WEIGHTS_TILE_SCALED:
            raise NotImplementedError("sharded resample-move rejuvenation needs weights='tile_scaled' (the peer-mapped one-launch filter)")
        sharded_ = world > 1 or D._forced()
        if self.weights == A.WEIGHTS_TILE_SCALED and sharded_:
            # the whole sharded filter in two launches per rank through peer-mapped windows (gjx_ssm_filter_peer); the
            # collective transport (global-maximum scheme, one exchange call per step) only if that is not available
            plain = not keep_means and not step_by_step and T >= 2
            if plain and self.K % world == 0 and (self.K // world) % 1024 == 0 and D.peer_available(dev):
                try:
                    res, ok = self._run_peer(key, ys_d, dev, world, check_status), True
                except GjxError:
                    # a rendezvous between the ranks timed out: their grids do not run side by side (ranks sharing one
                    # device in a dry run, a device held by something else)
                    res, ok = None, False
                if D.all_agree(ok, dev):
                    return res
                if self.rejuvenate or not getattr(self, "_allow_collective_fallback", True):
                    raise GjxError("sharded bootstrap filter: the peer-mapped launch timed out and there is no collective form of this run")
                _note_timeout("sharded bootstrap filter (peer-mapped)")
                plain = False      # -> the collective transport below
            if (not plain and (keep_means or step_by_step or T < 2)) or not getattr(self, "_allow_collective_fallback", True):
                raise NotImplementedError("sharded tile-scaled filter: needs the peer-mapped exchange (K / world a multiple of 1024, "
                                          "no keep_means / step_by_step)")
            import warnings
            warnings.warn("sharded bootstrap filter: peer-mapped exchange unavailable, using the collective transport with "
                          "global-maximum weights")
            if self.rejuvenate:
                raise NotImplementedError("sharded resample-move rejuvenation needs the peer-mapped exchange")
        if (self.rejuvenate and self.weights == A.WEIGHTS_TILE_SCALED and world == 1 and not keep_means and not step_by_step
                and not keep_history and not D._forced() and T >= 2):
            # resample-move inside the one-launch filter (gjx_ssm_filter_move); None: outside that kernel -> the loop below
            nm, sc = int(self.rejuvenate.get("n_moves", 1)), float(self.rejuvenate.get("scale", 0.5))
            out = kernels.ssm_filter_move(self.ssm.c_struct(dev), key, self.rng_mode, ys_d, self.K, nm, sc)
            if out is not None:
                st = kernels.workspace_status(out["_status_ws"], raise_on_error=False) if check_status else 0
                if not (st & 1):
                    incs = out["lse_steps"][:, 3]
                    self.last_accept_rate = float(out["accepted_total"][0]) / (self.K * (T - 1) * max(nm, 1))
                    res = dict(log_ml=incs.sum(), increments=incs, x=out["x"], logw=out["logw"], means=None, history=None)
                    if check_status:
                        res["degenerate"] = bool(st & 2)
                    return res
                _note_timeout("bootstrap filter (resample-move)")      # the grid was not co-resident: the loop below, plain launches
        if world == 1 and not keep_means and not step_by_step and not keep_history and not self.rejuvenate and not D._forced():
            out = kernels.ssm_filter(self.ssm.c_struct(dev), key, self.rng_mode, ys_d, self.K, weights=self.weights)
            st = kernels.workspace_status(out["_status_ws"], raise_on_error=False) if check_status else 0
            if st & 1:
                # the one-launch filter needs its whole grid resident; something else held compute units.  Same filter, same
                # keys, as one plain launch per stage (bit-identical results: tests/test_gpu_tiled.py) — log once, do not raise
                _note_timeout("bootstrap filter")
                out = kernels.ssm_filter(self.ssm.c_struct(dev), key, self.rng_mode, ys_d, self.K, weights=self.weights | A.WEIGHTS_PLAIN_LAUNCHES)
                st = kernels.workspace_status(out["_status_ws"], raise_on_error=False)
            incs = out["lse_steps"][:, 3]
            res = dict(log_ml=incs.sum(), increments=incs, x=out["x"], logw=out["logw"], means=None)
            if check_status:
                res["degenerate"] = bool(st & 2)
            return res
        off, K = D.shard(self.K, rank, world)
        sharded = world > 1 or D._forced()
        cs = self.ssm.c_struct(dev)
        ws = kernels.workspace(A.OP_SSM, K, dev)
        bufs = [torch.empty((self.ssm.dx, K), dtype=torch.float32, device=dev) for _ in range(2)]
        logw = torch.empty(K, dtype=torch.float32, device=dev)
        incs = torch.empty(T, dtype=torch.float32, device=dev)
        lse = torch.empty(4, dtype=torch.float32, device=dev)
        means = torch.empty((T, self.ssm.dx), dtype=torch.float32, device=dev) if keep_means else None
        if sharded:
            # equal shards only: every rank keeps K/world particles through every resampling
            if self.K % world:
                raise ValueError("sharded bootstrap filter: k_particles must be a multiple of the world size")
            key_ = (K, self.ssm.dx, self.K, str(dev))
            if getattr(self, "_resampler_key", None) != key_:
                self._resampler, self._resampler_key = D.ShardedResampler(K, self.ssm.dx, self.K, dev), key_
            resampler = self._resampler
            if resampler.transport == "rccl" and not keep_means and not step_by_step:
                out = kernels.ssm_filter_sharded(cs, key, self.rng_mode, ys_d, resampler.ctx, off)
                incs = out["lse_steps"][:, 3]
                return dict(log_ml=incs.sum(), increments=incs, x=out["x"], logw=out["logw"], means=None)
        else:
            ws2 = kernels.workspace(A.OP_RESAMPLE, K, dev)
            cum_buf = torch.empty(K, dtype=torch.int64, device=dev)
            bt_buf = torch.empty(2, dtype=torch.int64, device=dev)
        x_prev, anc = None, None
        hist = ParticleHistory() if keep_history else None
        k = key
        for t in range(T):
            k = fold_in(k, t)                      # chained step key (scan.py:268)
            k_prop, k_res = split(k)
            if t > 0 and not sharded:
                if self.weights == A.WEIGHTS_TILE_SCALED:
                    anc = kernels.resample_indices_tiled(logw, _unit_from_key(k_res), self.K, ws=ws2, cum=cum_buf)
                else:
                    anc = kernels.resample_indices(logw, _unit_from_key(k_res), self.K, True, lse, ws=ws2, cum=cum_buf, bt=bt_buf)
            x_out = bufs[t & 1]
            if self.rejuvenate:
                if t == 0:
                    mbufs = [torch.empty((self.ssm.dx, K), dtype=torch.float32, device=dev) for _ in range(2)]
                    acc_buf = torch.zeros(K, dtype=torch.float32, device=dev)
                    acc_sum = torch.zeros((), dtype=torch.float64, device=dev)
                    moved_buf = torch.empty((self.ssm.dx, K), dtype=torch.float32, device=dev) if keep_history else None
                kernels.ssm_step_move(cs, k_prop, self.rng_mode, t, K, x_prev, mbufs[(t + 1) & 1] if t > 1 else None, anc,
                                      ys_d[t - 1] if t > 0 else None, ys_d[t], int(self.rejuvenate.get("n_moves", 1)),
                                      float(self.rejuvenate.get("scale", 0.5)), x_out=x_out, m_out=mbufs[t & 1], logw=logw,
                                      accepted=acc_buf, lse=lse, offset=off, K_total=self.K, ws=ws, x_moved=moved_buf)
                if hist is not None and t > 0:
                    # the parents that were actually propagated are the MOVED ones: keep those (and their lineage)
                    hist.replace_parents(moved_buf, anc)
                    anc = torch.arange(K, dtype=torch.int32, device=dev)
                if t > 0:
                    acc_sum += acc_buf.double().mean()
            else:
                kernels.ssm_step(cs, k_prop, self.rng_mode, t, K, x_prev, anc, ys_d[t], x_out=x_out, logw=logw, lse=lse,
                                 offset=off, K_total=self.K, ws=ws)
            rec = lse
            if sharded:
                if t + 1 < T:
                    # resampling for step t+1 (its u comes from step t+1's key, as in the unsharded loop) also
                    # yields the global LSE record of step t
                    k_next = split(fold_in(k, t + 1))[1]
                    x_res, rec = resampler.step(x_out, logw, lse, _unit_from_key(k_next))
                else:
                    rec = D.global_lse(lse, self.K)
            incs[t] = rec[3]
            if hist is not None:
                hist.append(x_out, anc if t > 0 else None)
            if keep_means:
                w = torch.exp(logw - rec[2])
                m = (x_out * w).sum(dim=1)
                if world > 1:
                    torch.distributed.all_reduce(m)
                means[t] = m
            x_prev = x_res if (sharded and t + 1 < T) else x_out
        if self.rejuvenate and T > 1:
            self.last_accept_rate = float(acc_sum) / ((T - 1) * max(int(self.rejuvenate.get("n_moves", 1)), 1))
        if hist is not None:
            hist.logw = logw.clone()
        res = dict(log_ml=incs.sum(), increments=incs, x=x_prev, logw=logw, means=means, history=hist)
        return self._checked(res, None if sharded else ws2, check_status)
