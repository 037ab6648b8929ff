// gjx_scan.h — device-side pieces shared by the resampling kernels (gjx_resample.hip) and the one-launch importance
// step (gjx_run.hip): fixed-point weights, the reference maximum from block partials, the tagged-granule all-gather
// among co-resident blocks, and the tile scan + systematic slot-run expansion.
#pragma once
#include "gjx_device.h"

namespace gjx {

// ---- fixed-point weights ---------------------------------------------------------------------
constexpr float kWeightScale = 1073741824.0f;  // 2^30
constexpr int kScanItems = 4;                  // items per thread
constexpr int kScanTile = 256 * kScanItems;    // items per block

GJX_DEV uint64_t weight_q(const float* x, int64_t i, int is_log, float mx) {
  float w = is_log ? fast_exp(x[i] - mx) : x[i];
  w = w > 0.0f ? w : 0.0f;
  // log-weights: w <= 1 up to the rounding of v_exp_f32, the product fits 32 bits (one v_cvt_u32_f32; a float -> u64
  // conversion is eight instructions); linear weights may be anything
  return is_log ? (uint64_t)(uint32_t)(w * kWeightScale) : (uint64_t)(w * kWeightScale);
}

// reference maximum of the log-weights for the fixed-point conversion.
//   mode 1: lse[0] of a finished LSE record.
//   mode 2: `lse` points at n_partials per-block {max, sumexp} pairs left by the producing kernel
//           (gjx_run_program with lse == NULL): every block reduces them itself (a few KB from L2) — the LSE
//           "finish" rides in the consumer's prologue instead of being a serial tail of the producer.
// Block-uniform result; `red` is LDS scratch of >= 8 floats; ends with a barrier.
GJX_DEV float block_ref_max(int mode, const float* lse, int n_partials, float* red, float* sum_out) {
  if (mode != 2) { if (sum_out) *sum_out = 0.0f; return mode == 1 ? lse[0] : 0.0f; }
  const float2* parts = (const float2*)lse;
  float tmax = -INFINITY, tsum = 0.0f;
  for (int t = threadIdx.x; t < n_partials; t += 256) {
    const float2 p = parts[t];
    const float nm = fmaxf(tmax, p.x);
    if (nm > -INFINITY) tsum = tsum * fast_exp(tmax - nm) + p.y * fast_exp(p.x - nm);
    tmax = nm;
  }
  const float wm = wave_max(tmax);
  const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wm; red[4 + (threadIdx.x >> 6)] = ws; }
  __syncthreads();
  const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (sum_out) {
    float sm = 0.0f;
    for (int w = 0; w < 4; ++w) sm += m > -INFINITY ? red[4 + w] * fast_exp(red[w] - m) : 0.0f;
    *sum_out = sm;
  }
  return m;
}

// the same reduction over block pairs that other blocks of THIS launch stored (agent-scope 8-byte stores of pack_f2)
GJX_DEV float block_ref_max_live(const float* lse, int n_partials, float* red, float* sum_out) {
  const unsigned long long* parts = (const unsigned long long*)lse;
  float tmax = -INFINITY, tsum = 0.0f;
  for (int t = threadIdx.x; t < n_partials; t += 256) {
    const unsigned long long w = __hip_atomic_load(&parts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float px = __uint_as_float((unsigned)w), py = __uint_as_float((unsigned)(w >> 32));
    const float nm = fmaxf(tmax, px);
    if (nm > -INFINITY) tsum = tsum * fast_exp(tmax - nm) + py * fast_exp(px - nm);
    tmax = nm;
  }
  const float wm = wave_max(tmax);
  const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wm; red[4 + (threadIdx.x >> 6)] = ws; }
  __syncthreads();
  const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.0f;
  for (int w = 0; w < 4; ++w) sm += m > -INFINITY ? red[4 + w] * fast_exp(red[w] - m) : 0.0f;
  if (sum_out) *sum_out = sm;
  return m;
}

// ---- u64 wave scans on the DPP network (dpp_mov and the float forms: gjx_device.h) ----
template <unsigned CTRL, unsigned ROW_MASK>
GJX_DEV uint64_t dpp_add_u64(uint64_t v) {
  const uint32_t lo = dpp_mov<CTRL, ROW_MASK>(0u, (uint32_t)v), hi = dpp_mov<CTRL, ROW_MASK>(0u, (uint32_t)(v >> 32));
  return v + (((uint64_t)hi << 32) | lo);
}
// inclusive prefix sum over the 64 lanes (lane 63 = the wave total)
GJX_DEV uint64_t wave_scan_u64(uint64_t v) {
  v = dpp_add_u64<kDppShr1, 0xf>(v);
  v = dpp_add_u64<kDppShr2, 0xf>(v);
  v = dpp_add_u64<kDppShr4, 0xf>(v);
  v = dpp_add_u64<kDppShr8, 0xf>(v);
  v = dpp_add_u64<kDppBcast15, 0xa>(v);
  v = dpp_add_u64<kDppBcast31, 0xc>(v);
  return v;
}
GJX_DEV uint32_t wave_scan_u32(uint32_t v) {
  v += dpp_mov<kDppShr1, 0xf>(0u, v);
  v += dpp_mov<kDppShr2, 0xf>(0u, v);
  v += dpp_mov<kDppShr4, 0xf>(0u, v);
  v += dpp_mov<kDppShr8, 0xf>(0u, v);
  v += dpp_mov<kDppBcast15, 0xa>(0u, v);
  v += dpp_mov<kDppBcast31, 0xc>(0u, v);
  return v;
}
GJX_DEV uint64_t lane63_u64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}
GJX_DEV uint64_t wave_total_u64(uint64_t v) { return lane63_u64(wave_scan_u64(v)); }
// inclusive prefix sum inside each row of 16 lanes (lane 15 of a row = the row total)
GJX_DEV uint64_t row_scan_u64(uint64_t v) {
  v = dpp_add_u64<kDppShr1, 0xf>(v);
  v = dpp_add_u64<kDppShr2, 0xf>(v);
  v = dpp_add_u64<kDppShr4, 0xf>(v);
  v = dpp_add_u64<kDppShr8, 0xf>(v);
  return v;
}
GJX_DEV uint64_t wave_sum_u64(uint64_t v) { return wave_total_u64(v); }   // the total, in every lane


// status bits a co-resident kernel leaves in workspace word 10 (read and cleared by gjx_workspace_status)
enum { kStatusPollTimeout = 1u, kStatusZeroTotal = 2u, kStatusVerifyMismatch = 4u };

// ------------------------------------------------------------------------------------------
// All-gather of one 8-byte granule per block among the blocks of a CO-RESIDENT grid (no kernel boundary, no fence:
// MI355X guide, G16 form R2).  granule = (tag << 50) | value, value < 2^50, tag = (epoch mod 16383) + 1 != 0; `epoch`
// lives in the workspace control block and is bumped by block 0 once it has seen every granule of the call's LAST
// all-gather (by then every block has read the old epoch), so consecutive calls never mistake each other's granules
// and the workspace needs zeroing only once.  Every lane carries a poll budget (~0.1 s): a grid that is not co-resident
// must not hang — it sets kStatusPollTimeout and carries on with zeros.
// `visit(b, value)` is called by thread (b mod blockDim.x) for every block b.
// ------------------------------------------------------------------------------------------
constexpr unsigned long long kAggMask = (1ull << 50) - 1;
// polls (sleep + one L2-bypassing load: ~1 us each) a lane spends on ONE rendezvous before it gives up and flags
// kStatusPollTimeout; the host side then repeats the call on the plain multi-launch path (inference/pf.py)
constexpr unsigned kPollBudget = 1u << 16;

GJX_DEV unsigned long long grid_tag(const unsigned* ctrl, unsigned* epoch_out) {
  const unsigned epoch = __hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *epoch_out = epoch;
  return (unsigned long long)(epoch % 16383u) + 1ull;
}

// PAD: one granule every PAD words (PAD = 8: one per 64-byte line).  Blocks that store into a shared line serialise in the
// L2, readers pay PAD times the lines: worth it for the 256-block grids of the one-launch filters, not for 1024 blocks.
template <int PAD = 1>
GJX_DEV void grid_publish(unsigned long long* agg, unsigned long long tag, unsigned long long value) {
  __hip_atomic_store(&agg[(size_t)blockIdx.x * PAD], (tag << 50) | (value & kAggMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by every thread of the block; thread t polls the granules of blocks t, t + 256, ... one after the other (a
// variant with all of a lane's loads in flight together measured 2 us SLOWER per all-gather: the blocks arrive over
// several microseconds and the re-issued batches crowd the L2) and `visit(b, value)` runs on thread (b mod 256).
template <int PAD = 1, class Visit>
GJX_DEV void grid_gather(const unsigned long long* agg, unsigned long long tag, unsigned* ctrl, Visit&& visit) {
  unsigned budget = kPollBudget;   // polls this lane may spend in total (~0.1 s): a grid that is not co-resident must not hang
  // (sticky: blocks that start after the flag went up — the grid never was co-resident — do not wait at all)
  if (__hip_atomic_load(&ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusPollTimeout) budget = 0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += (int)blockDim.x) {
    unsigned long long v = 0;
    while (budget) {
      v = __hip_atomic_load(&agg[(size_t)b * PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 50) == tag) break;
      --budget;
      __builtin_amdgcn_s_sleep(1);
    }
    if ((v >> 50) != tag) { __hip_atomic_fetch_or(&ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0; }
    visit(b, v & kAggMask);
  }
}

// LDS of tile_scan_expand (one per block)
struct ScanSmem {
  static constexpr int kHeavyCap = 1024;   // particles with more than kOwn offspring, filled cooperatively
  uint64_t wsum[4], red2[8];
  int64_t jlast[256];
  int n_heavy;
  int32_t h_i[kHeavyCap];
  int64_t h_lo[kHeavyCap], h_hi[kHeavyCap];
  static constexpr int kStage = 2048;      // ancestors of the block's own slot run, staged for coalesced stores
  int32_t stage[kStage];
  int64_t run_lo, run_hi;
};

// Fixed-point weights of this block's tile (ITEMS consecutive particles per lane, first one i0) -> tile scan in
// registers -> all-gather of the tile totals -> systematic ancestors of the output slots this tile's particles own
// (slot-run expansion, see k_systematic_expand).  xv: the (log-)weights; mode != 0: log-weights against maximum mx.
// A zero grand total (all weights -inf / NaN / 0) yields the identity ancestors and sets kStatusZeroTotal.
// STORE_SC1: ancestors are stored write-through (sc1) so that other blocks of the SAME launch may read them.
template <int ITEMS, bool STORE_SC1>
GJX_DEV void tile_scan_expand(const float (&xv)[ITEMS], int mode, float mx, int64_t i0, int64_t K, double u, int64_t N,
                              int32_t* ancestors, uint64_t* cum_out, uint64_t* base_total_out, unsigned long long* agg,
                              unsigned long long tag, unsigned* ctrl, unsigned epoch, bool last_gather, ScanSmem& sm) {
  constexpr int kOwn = 8;
  if (threadIdx.x == 0) sm.n_heavy = 0;
  auto put = [&](int64_t j, int32_t v) {
    if (STORE_SC1) __hip_atomic_store(&ancestors[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else ancestors[j] = v;
  };
  // ---- tile scan in registers ----
  uint64_t q[ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    s += (i0 + k < K) ? weight_q(&xv[k], 0, mode, mx) : 0;
    q[k] = s;  // thread-local inclusive
  }
  uint64_t inc = wave_scan_u64(s);
  if ((threadIdx.x & 63) == 63) sm.wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint64_t off = inc - s;  // exclusive offset of this thread inside the block
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += sm.wsum[w];
  const uint64_t tile_total = sm.wsum[0] + sm.wsum[1] + sm.wsum[2] + sm.wsum[3];
  if (threadIdx.x == 0) grid_publish(agg, tag, tile_total);
  // ---- all-gather of the tile totals ----
  uint64_t pre = 0, tot = 0;
  grid_gather(agg, tag, ctrl, [&](int b, unsigned long long val) {
    tot += val;
    if (b < (int)blockIdx.x) pre += val;
  });
  pre = wave_sum_u64(pre);
  tot = wave_sum_u64(tot);
  if ((threadIdx.x & 63) == 0) { sm.red2[threadIdx.x >> 6] = pre; sm.red2[4 + (threadIdx.x >> 6)] = tot; }
  __syncthreads();
  const uint64_t prefix = sm.red2[0] + sm.red2[1] + sm.red2[2] + sm.red2[3];
  const uint64_t total = sm.red2[4] + sm.red2[5] + sm.red2[6] + sm.red2[7];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (last_gather) __hip_atomic_store(&ctrl[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every block has read `epoch` by now
    if (base_total_out) { base_total_out[0] = 0; base_total_out[1] = total; }
    if (total == 0) __hip_atomic_fetch_or(&ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (cum_out) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) if (i0 + k < K) cum_out[i0 + k] = prefix + off + q[k];
  }
  // ---- systematic ancestors by slot-range expansion (see k_systematic_expand) ----
  if (!ancestors) return;
  if (total == 0) {   // nothing to resample from: identity ancestors keep every later gather in bounds
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256) put(j, (int32_t)(j % K));
    return;
  }
  const double step = (double)total / (double)N;
  const double inv_step = (double)N / (double)total;
  const uint64_t c_last = prefix + off + s;  // inclusive prefix of this thread's last item
  const int64_t j_mine = slots_below(c_last, u, step, inv_step, total, N);
  sm.jlast[threadIdx.x] = j_mine;
  if (threadIdx.x == 0) sm.run_lo = slots_below(prefix, u, step, inv_step, total, N);
  if (threadIdx.x == 255) sm.run_hi = j_mine > N ? N : j_mine;
  __syncthreads();
  // write-through stores of single dwords from lanes that own 0..8 scattered slots each are slow (one fabric write
  // per lane); when the block's whole slot run fits the staging array the ancestors go to LDS first and leave as
  // lane-contiguous stores
  const int64_t run_lo = sm.run_lo, run_hi = sm.run_hi;
  const bool staged = STORE_SC1 && (run_hi - run_lo) <= ScanSmem::kStage;
  auto put2 = [&](int64_t j, int32_t v) {
    if (staged) sm.stage[j - run_lo] = v; else put(j, v);
  };
  int64_t j_prev = threadIdx.x > 0 ? sm.jlast[threadIdx.x - 1] : run_lo;
  uint64_t c_prev = prefix + off;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t i = i0 + k;
    const uint64_t c_cur = prefix + off + q[k];
    const int64_t j_cur = (k == ITEMS - 1) ? j_mine : ((c_cur > c_prev) ? slots_below(c_cur, u, step, inv_step, total, N) : j_prev);
    if (i < K && c_cur > c_prev) {
      const int64_t lo = j_prev, hi = j_cur > N ? N : j_cur;
      const int h = (hi - lo > kOwn) ? atomicAdd(&sm.n_heavy, 1) : ScanSmem::kHeavyCap;
      if (h < ScanSmem::kHeavyCap) {
        sm.h_i[h] = (int32_t)i; sm.h_lo[h] = lo; sm.h_hi[h] = hi;
      } else {   // few offspring, or the cooperative list is full (ITEMS > 4 with collapsed weights): write them here
        for (int64_t j = lo; j < hi; ++j) put2(j, (int32_t)i);
      }
    }
    j_prev = j_cur;
    c_prev = c_cur;
  }
  __syncthreads();
  const int nh = sm.n_heavy < ScanSmem::kHeavyCap ? sm.n_heavy : ScanSmem::kHeavyCap;
  for (int h = 0; h < nh; ++h) {
    const int32_t pi = sm.h_i[h];
    for (int64_t j = sm.h_lo[h] + threadIdx.x; j < sm.h_hi[h]; j += 256) put2(j, pi);
  }
  if (staged) {
    __syncthreads();
    for (int64_t j = run_lo + threadIdx.x; j < run_hi; j += 256) put(j, sm.stage[j - run_lo]);
  }
}

}  // namespace gjx
