// gjx_pfilter.inl — k_pf_persistent: steps 1 .. T-1 of the bootstrap filter of the linear-Gaussian state-space model in ONE
// launch for any number of particles a device holds, on one GPU or on a collection sharded over the GPUs of a node
// (gjx_peer.hip).  The filter's skeleton — rendezvous, tile-scaled systematic search, `ready` words, LSE ring, peer-mapped
// addressing — is pf_core (gjx_pfcore.h), shared with the kernels generated for any Scan kernel's step program
// (gjx_codegen.hip); this file is the hand-written MODEL of BASELINE configs 3 / 4 for that skeleton:
//   x_t ~ N(A x_{t-1}[ancestor], q^2 I),  log w_t = log N(y_t; H x_t, r^2 I)          (k_ssm_step's arithmetic and streams)
// with an optional resample-move rejuvenation of the gathered x_{t-1} (requests/rejuvenate.py:70-94 fused into the filter).
//
// k_ssm_persistent<TILED> (gjx_ssm.hip) gives every lane ONE slot, so its grid of K / 1024 blocks must be co-resident:
// K <= 2^18 on a full MI355X.  Here a block owns SPL consecutive quantisation tiles of 1024 slots (a lane: SPL slots, one
// per tile); the rendezvous cost (one per step) is shared by SPL tiles.  Same scheme (GJX_WEIGHTS_TILE_SCALED,
// include/gjx.h), same streams, same ancestors.
#pragma once
#include "gjx_pfcore.h"
#include "gjx_pfilter_host.h"

namespace gjx {

static_assert(kPfThreads == kPfCoreThreads && kPfGranulePad == kPfCorePad, "k_pf_persistent runs on pf_core's geometry");

template <int RNG, int DX, bool MOVE>
struct LgssmModel {
  const PfArgs& f;
  float* sA; float* sH; float* sY; float* sYp;     // LDS: A, H, y_t, y_{t-1} (MOVE)
  float rr, lconst;
  unsigned acc_lane;                               // MOVE: accepted Metropolis moves of this lane's slots, all steps
  typedef SsmNoiseBits<RNG, DX> Draws;

  GJX_DEV LgssmModel(const PfArgs& a, float* sA_, float* sH_, float* sY_, float* sYp_) : f(a), sA(sA_), sH(sH_), sY(sY_), sYp(sYp_), acc_lane(0u) {
    rr = fast_rcp(f.r);
    lconst = -(float)f.dy * (kHalfLog2Pi + fast_log(f.r));
  }
  GJX_DEV float* x_buf(int t) const { return (t & 1) ? f.x_b : f.x_a; }
  GJX_DEV float* m_buf(int t) const { return (t & 1) ? f.m_b : f.m_a; }   // MOVE: A x'_{t-1} of step t (step 0: the prior mean, zero — never read)

  GJX_DEV void prologue(int tid) {
    for (int e = tid; e < DX * DX; e += kPfThreads) sA[e] = f.A[e];
    if (f.H) for (int e = tid; e < f.dy * DX; e += kPfThreads) sH[e] = f.H[e];
  }
  GJX_DEV void stage(int t, int tid) {             // y_t (and y_{t-1}) by lanes of the second wave, while the granules travel
    const int lane = tid & 63;
    if ((tid >> 6) == 1 && lane < f.dy) {
      sY[lane] = f.ys[(size_t)t * f.dy + lane];
      if (MOVE) sYp[lane] = f.ys[(size_t)(t - 1) * f.dy + lane];
    }
  }
  GJX_DEV void draw(int, key2 key, uint64_t gidx, Draws& d) { ssm_noise_bits<RNG, DX>(key, gidx, d); }
  // verify mode: the check word of a row the previous launch (step 0) wrote
  GJX_DEV void seal_prev(int t_prev, int j, uint32_t gslot, const PfSlotCtx& cx) {
    uint32_t h = row_check_init(t_prev, gslot);
#pragma unroll
    for (int d = 0; d < DX; ++d) h = row_check_mix(h, x_buf(t_prev)[(int64_t)d * cx.K + j]);
    store_scoped_u32(cx.chk_cur + j, cx.verify == 2 ? h ^ 1u : h, cx.sys);
  }
  // ---- propagate + reweight one slot (k_ssm_step's arithmetic and streams) ----
  GJX_DEV float slot(int t, key2 key, int j, bool a, int sg, int sl, uint64_t gidx, const Draws* hoisted, const PfSlotCtx& cx) {
    const int64_t K = cx.K;
    const bool sys = cx.sys;
    const float* xs = peer_ptr((const float*)x_buf(t - 1), cx.sPD[sg]) + sl;
    float* x_out = x_buf(t);
    float xp[DX], nz[DX], xn[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) xp[d] = load_scoped(xs + (int64_t)d * K, sys);
    if (cx.verify) {                             // the pulled row against its owner's check word (step t - 1, global index)
      const unsigned want = load_scoped_u32(peer_ptr((const unsigned*)cx.chk_prev, cx.sPD[sg]) + sl, sys);
      uint32_t h = row_check_init(t - 1, (uint32_t)((int64_t)sg * K + sl));
#pragma unroll
      for (int d = 0; d < DX; ++d) h = row_check_mix(h, xp[d]);
      if (a && cx.live && h != want) cx.mismatch();
    }
    if constexpr (MOVE) {
      // resample-move: n_moves random-walk Metropolis steps on the gathered x_{t-1} with p(x_{t-1} | parent, y_{t-1}) as
      // invariant density — k_ssm_step<.., MOVE>'s arithmetic and draws (site 2 of the step's stream), so the one-launch
      // filter equals the step-by-step one bit for bit.  The parent's transition mean A x'_{t-2} was stored by the
      // previous step; step 1 moves x_0 under the prior N(0, q0^2 I).
      float mp[DX];
      const float* ms = peer_ptr((const float*)m_buf(t - 1), cx.sPD[sg]) + sl;
#pragma unroll
      for (int d = 0; d < DX; ++d) mp[d] = t > 1 ? load_scoped(ms + (int64_t)d * K, sys) : 0.0f;
      const float rq = fast_rcp(t > 1 ? f.q : f.q0);
      auto logpi = [&](const float (&xx)[DX]) {
        float sq = 0.0f;
#pragma unroll
        for (int d = 0; d < DX; ++d) { const float z = (xx[d] - mp[d]) * rq; sq = fmaf(z, z, sq); }
        if (f.H) {
          for (int o = 0; o < f.dy; ++o) {
            float mm = 0.0f;
#pragma unroll
            for (int e = 0; e < DX; ++e) mm = fmaf(sH[o * DX + e], xx[e], mm);
            const float z = (sYp[o] - mm) * rr;
            sq = fmaf(z, z, sq);
          }
        } else {
#pragma unroll
          for (int d = 0; d < DX; ++d) { const float z = (sYp[d] - xx[d]) * rr; sq = fmaf(z, z, sq); }
        }
        return -0.5f * sq;
      };
      BitStreamRT<RNG> bmv;
      bmv.open(key, gidx, 2u);
      float cur = logpi(xp), nacc = 0.0f;
      for (int n = 0; n < f.n_moves; ++n) {
        float xq[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) xq[d] = fmaf(f.move_scale, stream_normal<RNG>(bmv, (uint32_t)(n * (DX + 2) + d)), xp[d]);
        const float prop = logpi(xq);
        const float lu = safe_log(uniform_from_bits(bmv.get((uint32_t)(n * (DX + 2) + DX)), kTiny, 1.0f));
        if (lu < prop - cur) {
#pragma unroll
          for (int d = 0; d < DX; ++d) xp[d] = xq[d];
          cur = prop;
          nacc += 1.0f;
        }
      }
      if (a) acc_lane += (unsigned)nacc;         // summed over the launch: one atomic per wave at the very end
    }
    if (hoisted) ssm_noise_normals<RNG, DX>(*hoisted, nz);           // behind the loads above
    else {
      Draws late;                                                  // (tiles after the first hash here)
      ssm_noise_bits<RNG, DX>(key, gidx, late);
      ssm_noise_normals<RNG, DX>(late, nz);
    }
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      float acc = 0.0f;
#pragma unroll
      for (int e = 0; e < DX; ++e) acc = fmaf(sA[d * DX + e], xp[e], acc);
      if (MOVE && a) store_scoped(m_buf(t) + (int64_t)d * K + j, acc, sys);
      xn[d] = fmaf(f.q, nz[d], acc);
    }
    if (a) {
#pragma unroll
      for (int d = 0; d < DX; ++d) store_scoped(x_out + (int64_t)d * K + j, xn[d], sys);
      if (cx.verify) {
        uint32_t h = row_check_init(t, (uint32_t)gidx);
#pragma unroll
        for (int d = 0; d < DX; ++d) h = row_check_mix(h, xn[d]);
        store_scoped_u32(cx.chk_cur + j, cx.verify == 2 ? h ^ 1u : h, sys);
      }
    }
    float qsum = 0.0f;
    if (f.H) {
      for (int o = 0; o < f.dy; ++o) {
        float mm = 0.0f;
#pragma unroll
        for (int e = 0; e < DX; ++e) mm = fmaf(sH[o * DX + e], xn[e], mm);
        const float z = (sY[o] - mm) * rr;
        qsum = fmaf(z, z, qsum);
      }
    } else {
#pragma unroll
      for (int d = 0; d < DX; ++d) { const float z = (sY[d] - xn[d]) * rr; qsum = fmaf(z, z, qsum); }
    }
    return fmaf(-0.5f, qsum, lconst);
  }
  GJX_DEV void epilogue(int lane) {
    if constexpr (MOVE) {
      if (f.acc_total) {
        const unsigned wacc = wave_scan_u32(acc_lane);     // lane 63: the wave's total (< 2^32: 64 lanes x SPL x T x n_moves)
        if (lane == 63 && wacc) atomicAdd(f.acc_total, (unsigned long long)wacc);
      }
    }
  }
};

GJX_DEV PfCoreArgs pf_core_args(const PfArgs& f) {
  PfCoreArgs c;
  c.T = f.T; c.K = f.K; c.K_total = f.K_total; c.offset = f.offset; c.G = f.G; c.rank = f.rank; c.nt = f.nt; c.NT = f.NT;
  c.lw_even = f.lw_even; c.lw_odd = f.lw_odd; c.aggA = f.aggA; c.aggB = f.aggB; c.bsum = f.bsum; c.bmax = f.bmax; c.ready = f.ready;
  c.peer_data = f.peer_data; c.peer_flag = f.peer_flag; c.keys = f.keys; c.us = f.us; c.lse_steps = f.lse_steps;
  c.ancestors = f.ancestors; c.ancestors_all = nullptr; c.ctrl = f.ctrl; c.log_k = f.log_k; c.first_budget = f.first_budget;
  c.zero_ptr = f.zero_ptr; c.zero_n = f.zero_n; c.verify = f.verify; c.chk_a = f.chk_a; c.chk_b = f.chk_b; c.timeline = f.timeline;
  return c;
}

template <int RNG, int DX, int SPL, bool MOVE = false>
__global__ __launch_bounds__(kPfThreads) void k_pf_persistent(PfArgs f) {
  extern __shared__ __align__(16) unsigned char pf_dyn[];
  // model constants in LDS: inside the step loop the compiler must assume the kernel's own stores may alias A, H, ys and
  // re-reads them with VECTOR loads every step
  __shared__ float sA[DX * DX], sH[kSsmPersistMaxDy * DX], sY[kSsmPersistMaxDy];
  __shared__ float sYp[MOVE ? kSsmPersistMaxDy : 1];       // MOVE: y_{t-1}, the observation the moved particle is conditioned on
  LgssmModel<RNG, DX, MOVE> m(f, sA, sH, sY, sYp);
  const PfCoreArgs c = pf_core_args(f);
  pf_core<LgssmModel<RNG, DX, MOVE>, SPL>(c, m, pf_dyn);
}

// kernel of one (dx, spl) for this translation unit's RNG; NULL when the combination is not instantiated
template <int RNG>
const void* pf_kernel_of(int dx, int spl, bool move) {
#define GJX_PF(DXV, SPLV) if (dx == DXV && spl == SPLV) return move ? (const void*)k_pf_persistent<RNG, DXV, SPLV, true> : (const void*)k_pf_persistent<RNG, DXV, SPLV, false>;
  GJX_PF(2, 1) GJX_PF(2, 2) GJX_PF(2, 4) GJX_PF(2, 8)
  GJX_PF(4, 1) GJX_PF(4, 2) GJX_PF(4, 4) GJX_PF(4, 8)
  GJX_PF(8, 1) GJX_PF(8, 2) GJX_PF(8, 4) GJX_PF(8, 8) GJX_PF(8, 16)   // 16: config 4's dry run (8 ranks share ONE device's blocks)
  GJX_PF(16, 1) GJX_PF(16, 2) GJX_PF(16, 4)
#undef GJX_PF
  return nullptr;
}

}  // namespace gjx
