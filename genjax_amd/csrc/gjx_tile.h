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


}  // namespace gjx
