// gjx_tile.h — device helpers shared by the one-launch filters (gjx_ssm.hip, gjx_pfilter.inl): agent- / system-scope
// accesses for data other blocks (or other ranks) of the SAME launch read, the propagation noise in two halves, and the
// tile-scaled fixed point (GJX_WEIGHTS_TILE_SCALED, include/gjx.h).
#pragma once
#include "gjx_device.h"
#include "gjx_scan.h"

namespace gjx {

constexpr int kSsmPersistMaxDy = 32;              // observation dimension the one-launch filters stage in LDS

GJX_DEV float load_agent(const float* p) { return __int_as_float(__hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
// four consecutive floats through ONE 16-byte sc1 load (4-byte sc1 accesses run at a fraction of the 16-byte rate); the
// wait sits inside the asm because the compiler does not count an asm's memory operations
GJX_DEV void load_agent_x4(const float* p, float (&v)[4]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
// four consecutive floats through ONE 16-byte write-through store (completion: the caller's s_waitcnt vmcnt(0) — the compiler
// does not count an asm's memory operations)
GJX_DEV void store_agent_x4(float* p, const float (&v)[4]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 r = {v[0], v[1], v[2], v[3]};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(r) : "memory");
}
GJX_DEV void store_agent(float* p, float v) { __hip_atomic_store((int*)p, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The same accesses with the scope as a (wave-uniform) argument: `sys` selects SYSTEM scope (sc0 sc1) for memory that
// kernels of OTHER ranks read or write during the same launch — peer-mapped windows over xGMI, gjx_peer.hip.  One scalar
// branch per access; on one GPU the agent-scope form runs.
GJX_DEV float load_scoped(const float* p, bool sys) {
  return sys ? __int_as_float(__hip_atomic_load((const int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) : load_agent(p);
}
GJX_DEV void load_scoped_x4(const float* p, float (&v)[4], bool sys) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 r;
  if (sys) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
GJX_DEV void store_scoped(float* p, float v, bool sys) {
  if (sys) __hip_atomic_store((int*)p, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else store_agent(p, v);
}
GJX_DEV void store_scoped_u64(unsigned long long* p, unsigned long long v, bool sys) {
  if (sys) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
GJX_DEV unsigned long long load_scoped_u64(const unsigned long long* p, bool sys) {
  return sys ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
GJX_DEV void store_scoped_u32(unsigned* p, unsigned v, bool sys) {
  if (sys) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
GJX_DEV unsigned load_scoped_u32(const unsigned* p, bool sys) {
  return sys ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- verify mode of the sharded kernels (GJX_PEER_VERIFY=1, gjx_peer.hip): a check word per particle row, stored by the
// owner beside the row and recomputed by every reader that pulls the row.  It folds in the STEP the row belongs to and the
// particle's global index, so a reader that is served an older generation of the same addresses (the buffers ping-pong:
// the previous occupant is the row of step t - 2), a torn row, or a row of another particle cannot pass.
GJX_DEV uint32_t row_check_init(int t, uint32_t gslot) { return (0x9E3779B9u * (uint32_t)(t + 1)) ^ (gslot * 0x85EBCA6Bu); }
GJX_DEV uint32_t row_check_mix(uint32_t h, float v) {
  h ^= __float_as_uint(v);
  h *= 0x01000193u;
  return h ^ (h >> 15);
}
// pointer into rank g's copy of a window: this rank's pointer + the byte distance between the two mappings (0 for g == rank)
template <class Tp>
GJX_DEV Tp* peer_ptr(Tp* local, long long delta) { return (Tp*)((char*)local + delta); }

// standard-normal draws of slot gidx under the step's propagation key (k_ssm_step's streams), in two halves: the random
// words (the hashes: most of the VALU work, done inside the granule wait) and the normals from them (Box-Muller, done
// behind the loads of the ancestor's state).  A block's window work must stay below the ~1 us of slack the LAST block
// to publish has over the others, or that block is late again in the next step and its chain sets the step time.
template <int RNG, int DX>
struct SsmNoiseBits { uint32_t w[RNG == GJX_RNG_FLAT ? 2 * GJX_FLAT_BLOCKS(DX + (DX & 1)) : DX]; };

template <int RNG, int DX>
GJX_DEV void ssm_noise_bits(key2 skj, uint64_t gidx, SsmNoiseBits<RNG, DX>& nb) {
  if (RNG == GJX_RNG_JAX32) skj = fold_in(fold_in64(skj, gidx), 1u);
  else if (gidx >> 32) skj = threefry2x32(skj, 0xFFFFFFFFu, (uint32_t)(gidx >> 32));
  if (RNG == GJX_RNG_FLAT) {
    constexpr int NB = GJX_FLAT_BLOCKS(DX + (DX & 1));
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      const key2 hh = threefry2x32(skj, (uint32_t)gidx, (1u << GJX_FLAT_SITE_SHIFT) | (uint32_t)h);
      nb.w[2 * h] = hh.a; nb.w[2 * h + 1] = hh.b;
    }
  } else {
#pragma unroll
    for (int d0 = 0; d0 < DX; ++d0) {
      const key2 h0 = threefry2x32(skj, 0u, (uint32_t)d0);
      nb.w[d0] = h0.a ^ h0.b;
    }
  }
}
template <int RNG, int DX>
GJX_DEV void ssm_noise_normals(const SsmNoiseBits<RNG, DX>& nb, float (&nz)[DX]) {
  if (RNG == GJX_RNG_FLAT) {
#pragma unroll
    for (int d0 = 0; d0 < DX; d0 += 2) {
      float n0, n1;
      box_muller(GJX_FIELD(nb.w, d0), GJX_FIELD(nb.w, d0 + 1), n0, n1);
      nz[d0] = n0;
      if (d0 + 1 < DX) nz[d0 + 1] = n1;
    }
  } else {
#pragma unroll
    for (int d0 = 0; d0 < DX; ++d0) nz[d0] = normal_from_bits_fast(nb.w[d0]);
  }
}

// ---- tile-scaled fixed point (shared by k_ssm_persistent<TILED> and the gjx_resample_indices_tiled kernels) ----
constexpr int kTileQ = 1024;                       // particles per quantisation tile
constexpr float kTileScale = 536870912.0f;         // 2^29
constexpr int kTileDead = -524288;                 // exponent of a tile without a finite positive weight (-2^19)
constexpr float kLog2e = 1.44269504f;

GJX_DEV int tile_exponent(float tile_max) {        // e_b = ceil(max * log2 e), clamped to 20 bits; tile_max never NaN (fmaxf)
  if (!(tile_max > -INFINITY)) return kTileDead;
  const float t = ceilf(tile_max * kLog2e);
  return t < -524287.0f ? -524287 : (t > 524287.0f ? 524287 : (int)t);
}
GJX_DEV uint64_t tile_q(float lw, int e) {         // floor(2^29 * min(1, 2^(lw * log2 e - e))); NaN / -inf / dead tile -> 0
  if (e == kTileDead) return 0;
  float w = __builtin_amdgcn_exp2f(fmaf(lw, kLog2e, -(float)e));   // one rounding: the oracle uses fmaf too
  w = w > 0.0f ? w : 0.0f;
  w = w < 1.0f ? w : 1.0f;
  return (uint64_t)(uint32_t)(w * kTileScale);    // <= 2^29: one v_cvt_u32_f32 (a float -> u64 conversion is eight instructions)
}
// granule of the tiled rendezvous: tag (4 bits, != 0) | e_b + 2^19 (20 bits) | S_b (40 bits, S_b <= 2^39).  Four tag
// bits are plenty: a block rewrites its granule of one parity every second step, so a reader can only ever meet the
// tag of step t or of step t - 2
GJX_DEV unsigned long long tile_granule(unsigned long long tag, int e, uint64_t S) {
  return (tag << 60) | ((unsigned long long)(unsigned)(e - kTileDead) << 40) | (S & ((1ull << 40) - 1));
}

// ---- the search of the tile-scaled systematic resampler for ONE tile of 1024 output slots (include/gjx.h,
// GJX_WEIGHTS_TILE_SCALED): lane t of a 256-thread block owns the slots tix 1024 + 4 t .. + 3 and gets their ancestors.
// Shared by k_resample_gather_tiled (gjx_resample.hip) and by generated step kernels that resample in their prologue
// (gjx_codegen.hip, gjx_run_resample): ONE body, hence the same ancestors bit for bit.  x: log-weights; S, E: {S_b, e_b} of
// every tile; !PLANNED: Pl [nt + 2], Ebl [nt] are block-shared scratch (every block reduces all nt totals), PLANNED: Pg, shg
// hold the prefix and the shifts (k_tiled_plan).  tix == 0 also finishes the LSE record of the producing run (lse_mode 2) and
// flags a dead collection.  Every thread of the block must call it (block barriers inside).
constexpr int kLiveGranulePad = 8;         // words between the granules of the steps kernel (gjx_gen_steps)
struct TiledSearchShared {
  float fred[8];
  uint64_t wsum[4];
  uint64_t cumL[3 * 1024];                                     // cumulative q of the tiles being searched, relative to the tile's start
  uint64_t s_wtot[3][4];
  int s_range[2];
};

// LIVE: the collection being resampled was written by other blocks of THIS launch (the steps kernel of a generated filter,
// gjx_codegen.hip): S holds one tagged granule {rtag, e_b, S_b} per tile (tile_granule), published by its block once the block's
// write-through stores of the step had completed; this block polls them — the rendezvous of the step — and reads everything else
// of the previous step (log-weights, block pairs) with agent-scope loads.  E is unused.
template <bool PLANNED, bool LIVE = false>
GJX_DEV void tiled_search_tile(const float* __restrict__ x, int64_t K, const uint64_t* __restrict__ S, const int32_t* __restrict__ E,
                               const uint64_t* __restrict__ Pg, const int32_t* __restrict__ shg, const int nt, const int tix, uint64_t* const Pl,
                               int32_t* const Ebl, TiledSearchShared& sh, int lse_mode, const float* lse, int n_partials, float* lse_out,
                               float log_k_total, double u, unsigned* ctrl, unsigned long long* timeline, int32_t (&anc)[4],
                               unsigned long long rtag = 0ull) {
  static_assert(!(PLANNED && LIVE), "the live form reduces the granules itself");
#define GJX_STAMP(n) do { if (timeline && threadIdx.x == 0) timeline[tix * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  constexpr int ITEMS = 4, TILE = 256 * ITEMS, CH = 3;
  const uint64_t* const P = PLANNED ? Pg : Pl;
  const int32_t* const Eb = PLANNED ? E : Ebl;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i0 = ((int64_t)tix * 256 + threadIdx.x) * ITEMS;
  auto load_tile = [&](int64_t p0, float (&v)[ITEMS]) {
    if constexpr (LIVE) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (p0 + k < K) ? load_agent(x + p0 + k) : -INFINITY;
    } else if (p0 + 4 <= K) {
      const float4 q4 = *(const float4*)(x + p0);
      v[0] = q4.x; v[1] = q4.y; v[2] = q4.z; v[3] = q4.w;
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (p0 + k < K) ? x[p0 + k] : -INFINITY;
    }
  };
  // ---- every load of the head, at once: the three tiles of log-weights around the block, all tile totals (LIVE: the granules
  //      first — a tile's log-weights are complete only once its granule is there) ----
  const int w0 = tix - 1;
  float xw[CH][ITEMS];
  auto load_window = [&]() {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) xw[c][k] = -INFINITY;
      const int tc = w0 + c;
      if (tc >= 0 && tc < nt) load_tile((int64_t)tc * TILE + (int64_t)threadIdx.x * ITEMS, xw[c]);
    }
  };
  if constexpr (!LIVE) load_window();
  int Emax = 0;
  if constexpr (!PLANNED) {
  float em = (float)kTileDead;
  if constexpr (LIVE) {
    unsigned budget = (ctrl && (__hip_atomic_load(&ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusPollTimeout)) ? 0u : kPollBudget;
    for (int b = threadIdx.x; b < nt; b += 256) {
      // (one granule per 64-byte line, kLiveGranulePad: blocks that store into a shared line serialise in the L2)
      unsigned long long v = __hip_atomic_load((const unsigned long long*)&S[(size_t)b * kLiveGranulePad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((v >> 60) != rtag && budget) {
        --budget;
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load((const unsigned long long*)&S[(size_t)b * kLiveGranulePad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((v >> 60) != rtag) { if (ctrl) __hip_atomic_fetch_or(&ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0; }
      const uint64_t sv = v & ((1ull << 40) - 1);
      const int e = sv ? (int)((v >> 40) & 0xFFFFFu) + kTileDead : kTileDead;
      Pl[b + 1] = sv;
      Ebl[b] = e;
      em = fmaxf(em, (float)e);
    }
    __syncthreads();        // every granule of the step has been seen by some lane of this block: the whole previous step is readable
    load_window();
  } else {
  for (int b = threadIdx.x; b < nt; b += 256) {
    const uint64_t sv = S[b];
    const int e = sv ? E[b] : kTileDead;
    Pl[b + 1] = sv;
    Ebl[b] = e;
    em = fmaxf(em, (float)e);
  }
  }
  em = wave_max(em);
  if (lane == 0) sh.fred[wid] = em;
  if (threadIdx.x == 0) Pl[0] = 0;
  __syncthreads();
  Emax = (int)fmaxf(fmaxf(sh.fred[0], sh.fred[1]), fmaxf(sh.fred[2], sh.fred[3]));
  GJX_STAMP(1);
  {   // shifted totals and their prefix, in place: lane t owns entries [t per, (t+1) per)
    const int per = (nt + 255) >> 8;
    const int e0 = threadIdx.x * per < nt ? threadIdx.x * per : nt, e1 = (e0 + per) < nt ? (e0 + per) : nt;
    uint64_t loc = 0;
    for (int e = e0; e < e1; ++e) {
      const int sh = Emax - Ebl[e];
      const uint64_t g = sh < 64 ? Pl[e + 1] >> sh : 0;
      Pl[e + 1] = g;
      loc += g;
    }
    const uint64_t inc = wave_scan_u64(loc);
    if (lane == 63) sh.wsum[wid] = inc;
    __syncthreads();
    uint64_t run = inc - loc;
    for (int w = 0; w < wid; ++w) run += sh.wsum[w];
    for (int e = e0; e < e1; ++e) { run += Pl[e + 1]; Pl[e + 1] = run; }
    __syncthreads();
  }
  }
  auto shift_of = [&](int t) { return PLANNED ? shg[t] : Emax - Eb[t]; };   // (a tile shifted out entirely is never a source)
  const uint64_t total = P[nt];
  GJX_STAMP(2);
  if (tix == 0) {   // block-uniform: the LSE record of the producing run (its block partials), the dead-collection flag
    if (lse_mode == 2 && lse_out) {
      float sm_sum;
      const float mx = LIVE ? block_ref_max_live(lse, n_partials, sh.fred, &sm_sum) : block_ref_max(2, lse, n_partials, sh.fred, &sm_sum);
      if (threadIdx.x == 0) {
        const float l = mx > -INFINITY ? mx + logf(sm_sum) : -INFINITY;
        lse_out[0] = mx; lse_out[1] = sm_sum; lse_out[2] = l; lse_out[3] = l - log_k_total;
      }
    }
    if (threadIdx.x == 0 && total == 0 && ctrl) __hip_atomic_fetch_or(&ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) anc[k] = (int32_t)((i0 + k < K) ? i0 + k : K - 1);   // dead collection: identity (flagged)
  if (total > 0) {   // block-uniform
    const double step = (double)total / (double)K;
    uint64_t T[ITEMS];
    int tile[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int64_t j = (i0 + k < K) ? i0 + k : K - 1;
      T[k] = comb_threshold(j, u, step, total);
      tile[k] = 0;
    }
    auto descend = [&](int k0, int k1) {   // first tile t with P[t + 1] > T (fixed trip count; two slots per pass)
      for (int sft = 1 << (31 - __builtin_clz((unsigned)nt)); sft >= 1; sft >>= 1) {
        const int pa = tile[k0] + sft, pb = tile[k1] + sft;
        if (pa <= nt - 1 && P[pa] <= T[k0]) tile[k0] = pa;
        if (k1 != k0 && pb <= nt - 1 && P[pb] <= T[k1]) tile[k1] = pb;
      }
    };
    const int wlo = tix - 4 < 0 ? 0 : (tix - 4 > nt - 8 ? (nt - 8 < 0 ? 0 : nt - 8) : tix - 4);
    uint64_t Pw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Pw[k] = P[wlo + k < nt ? wlo + k : nt];
    if (T[0] >= Pw[0] && T[ITEMS - 1] < Pw[8]) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        int tl = wlo;
#pragma unroll
        for (int w = 1; w < 8; ++w) tl += Pw[w] <= T[k] ? 1 : 0;
        tile[k] = tl;
      }
    } else {
      descend(0, ITEMS - 1);
      if (tile[0] == tile[ITEMS - 1]) {
#pragma unroll
        for (int k = 1; k < ITEMS - 1; ++k) tile[k] = tile[0];
      } else {
        descend(1, 2);
      }
    }
    // residual of every slot in its source tile's own units (< S of that tile)
    uint64_t Tr[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) Tr[k] = (T[k] - P[tile[k]]) << shift_of(tile[k]);
    if (threadIdx.x == 0) sh.s_range[0] = tile[0];
    if (threadIdx.x == 255) sh.s_range[1] = tile[ITEMS - 1];
    __syncthreads();
    const int tmin = sh.s_range[0], tmax = sh.s_range[1];
    GJX_STAMP(3);
    const bool windowed = tmin >= w0 && tmax <= w0 + 2;       // block-uniform: every source tile is among the three loaded at the top
    if (windowed) {
      // cumulative q of the window tiles the slots fall into (usually two), each against its own exponent, relative to
      // the tile's start: sh.cumL[c] for tile w0 + c
      uint64_t qi[CH][ITEMS], sacc[CH], inc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int tc = w0 + c;
        sacc[c] = 0; inc[c] = 0;
        if (tc < tmin || tc > tmax) continue;                 // block-uniform
        const int es = Eb[tc];
        const int64_t p0 = (int64_t)tc * TILE + (int64_t)threadIdx.x * ITEMS;
        if (p0 + ITEMS <= K) {
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) { sacc[c] += tile_q(xw[c][k], es); qi[c][k] = sacc[c]; }
        } else {
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) { sacc[c] += p0 + k < K ? tile_q(xw[c][k], es) : 0; qi[c][k] = sacc[c]; }
        }
        inc[c] = wave_scan_u64(sacc[c]);
        if (lane == 63) sh.s_wtot[c][wid] = inc[c];
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int tc = w0 + c;
        if (tc < tmin || tc > tmax) continue;
        uint64_t base = inc[c] - sacc[c];
        for (int w = 0; w < wid; ++w) base += sh.s_wtot[c][w];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) sh.cumL[c * TILE + threadIdx.x * ITEMS + k] = base + qi[c][k];
      }
      __syncthreads();
    }
    auto search = [&](const uint64_t* cm, uint64_t r) {   // entries <= r among TILE (4-ary: three independent probes per level)
      int pos = 0;
#pragma unroll
      for (int q = TILE >> 2; q >= 1; q >>= 2) {
        const uint64_t pa = cm[pos + q - 1], pb = cm[pos + 2 * q - 1], pc = cm[pos + 3 * q - 1];
        pos += (pa <= r ? q : 0) + (pb <= r ? q : 0) + (pc <= r ? q : 0);
      }
      return pos;
    };
    if (windowed) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) anc[k] = (int32_t)((int64_t)tile[k] * TILE + search(sh.cumL + (tile[k] - w0) * TILE, Tr[k]));
    } else {
      // collapsed or strongly drifting weights: re-scan the source tiles CH at a time (tiles without weight are skipped)
      const int ntiles = tmax - tmin + 1;
      auto next_live = [&](int at) { while (at < ntiles && P[tmin + at + 1] == P[tmin + at]) ++at; return at; };
      for (int idx = 0; idx < ntiles;) {
        const int nidx = next_live(idx + CH);
        uint64_t qi[CH][ITEMS], sacc[CH], inc[CH];
        __syncthreads();         // every lane is done searching the previous contents of sh.cumL
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const bool on = idx + c < ntiles;
          const int tsrc = on ? tmin + idx + c : 0;
          const int64_t p0 = (int64_t)tsrc * TILE + (int64_t)threadIdx.x * ITEMS;
          float nv[ITEMS];
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) nv[k] = -INFINITY;
          if (on) load_tile(p0, nv);
          const int es = Eb[tsrc];
          sacc[c] = 0;
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) { sacc[c] += (on && p0 + k < K) ? tile_q(nv[k], es) : 0; qi[c][k] = sacc[c]; }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) inc[c] = wave_scan_u64(sacc[c]);
        if (lane == 63) {
#pragma unroll
          for (int c = 0; c < CH; ++c) sh.s_wtot[c][wid] = inc[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          uint64_t base = inc[c] - sacc[c];
          for (int w = 0; w < wid; ++w) base += sh.s_wtot[c][w];
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) sh.cumL[c * TILE + threadIdx.x * ITEMS + k] = base + qi[c][k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
          const int kp = tile[k] - tmin;
          if (kp >= idx && kp < idx + CH) anc[k] = (int32_t)((int64_t)tile[k] * TILE + search(sh.cumL + (kp - idx) * TILE, Tr[k]));
        }
        idx = nidx;
      }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) anc[k] = anc[k] < K ? anc[k] : (int32_t)(K - 1);
  }
#undef GJX_STAMP
}

}  // namespace gjx
