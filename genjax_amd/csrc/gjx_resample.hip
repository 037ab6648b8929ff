// gjx_resample.hip — 1-of-K categorical pick (Gumbel-max argmax reduce), fixed-point weight
// prefix sums, systematic / multinomial ancestor search, SoA row gather.
//
// All index work is integer: weights become q_i = (uint64)(w_i * 2^30), prefix sums are exact
// uint64 adds (associative, so the block decomposition cannot change a single bit), comb
// thresholds are one IEEE double multiply + truncation per output slot.  HBM traffic per
// particle: pick 4 B read; cumsum 2x4 B read + 8 B write; search 4 B write (+ L2-resident
// binary-search probes); gather 4 B read + 4 B write per row.
#include "gjx_device.h"
#include "gjx_host.h"
#include "gjx_scan.h"
#include "gjx_tile.h"

namespace gjx {

struct PickPair {
  float v;
  int32_t i;
};

GJX_DEV void pick_better(float& v, int32_t& i, float ov, int32_t oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <int RNG>
__global__ __launch_bounds__(256) void k_pick_partial(const float* logw, int64_t K, int64_t offset, const float* lse,
                                                     key2 key, PickPair* partials) {
  __shared__ float rv[4];
  __shared__ int32_t ri[4];
  const float l = lse[2];
  float bv = -INFINITY;
  int32_t bi = 0x7FFFFFFF;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < K; i += (int64_t)gridDim.x * 256) {
    const uint64_t gi = (uint64_t)(offset + i);
    uint32_t bits;
    if (RNG == GJX_RNG_JAX32) {
      const key2 h = fold_in64(key, gi);
      bits = h.a ^ h.b;
    } else {
      const key2 h = fold_in64(key, gi >> 1);
      bits = (gi & 1u) ? h.b : h.a;
    }
    const float v = (logw[i] - l) + gumbel_from_bits(bits);
    pick_better(bv, bi, v, (int32_t)gi);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int32_t oi = __shfl_xor(bi, o, 64);
    pick_better(bv, bi, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = bv; ri[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) pick_better(bv, bi, rv[w], ri[w]);
    partials[blockIdx.x] = PickPair{bv, bi};
  }
}

__global__ __launch_bounds__(256) void k_pick_finish(const PickPair* partials, int n, PickPair* out) {
  __shared__ float rv[4];
  __shared__ int32_t ri[4];
  float bv = -INFINITY;
  int32_t bi = 0x7FFFFFFF;
  for (int t = threadIdx.x; t < n; t += 256) pick_better(bv, bi, partials[t].v, partials[t].i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int32_t oi = __shfl_xor(bi, o, 64);
    pick_better(bv, bi, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = bv; ri[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) pick_better(bv, bi, rv[w], ri[w]);
    *out = PickPair{bv, bi};
  }
}

// Independent trials (jax.vmap(alg.run_smc)(split(key, n)), README.md:110-113): trial t owns particles [t K, (t + 1) K) of one
// n K-particle run.  One block per trial: its LSE record {max, sumexp, lse, lse - log K} and, if asked, its 1-of-K draw
// argmax_i (logw_i - lse_t) + Gumbel(bits(key, global index)) — the same rule as k_pick_partial with particle_offset = t K.
template <int RNG>
__global__ __launch_bounds__(256) void k_trials_lse_pick(const float* logw, int64_t K, int64_t offset, key2 key, int want_pick, float log_k,
                                                        float* lse_out, int32_t* pick_out) {
  __shared__ float red[8];
  __shared__ float rv[4];
  __shared__ int32_t ri[4];
  const int64_t t = blockIdx.x;
  const float* lw = logw + t * K;
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < K; i += 256) m = fmaxf(m, lw[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.0f;
  if (bm > -INFINITY)
    for (int64_t i = threadIdx.x; i < K; i += 256) sm += fast_exp(lw[i] - bm);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sm;
  __syncthreads();
  const float bs = (red[4] + red[5]) + (red[6] + red[7]);
  const float l = bm > -INFINITY ? bm + fast_log(bs) : -INFINITY;
  if (threadIdx.x == 0) {
    float* o = lse_out + 4 * t;
    o[0] = bm; o[1] = bs; o[2] = l; o[3] = l - log_k;
  }
  if (!want_pick) return;
  float bv = -INFINITY;
  int32_t bi = 0x7FFFFFFF;
  for (int64_t i = threadIdx.x; i < K; i += 256) {
    const uint64_t gi = (uint64_t)(offset + t * K + i);
    uint32_t bits;
    if (RNG == GJX_RNG_JAX32) {
      const key2 h = fold_in64(key, gi);
      bits = h.a ^ h.b;
    } else {
      const key2 h = fold_in64(key, gi >> 1);
      bits = (gi & 1u) ? h.b : h.a;
    }
    pick_better(bv, bi, (lw[i] - l) + gumbel_from_bits(bits), (int32_t)gi);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int32_t oi = __shfl_xor(bi, o, 64);
    pick_better(bv, bi, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = bv; ri[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) pick_better(bv, bi, rv[w], ri[w]);
    pick_out[t] = bi == 0x7FFFFFFF ? (int32_t)(offset + t * K) : bi;   // a dead trial (all weights zero, lse = -inf): its first particle
  }
}

__global__ __launch_bounds__(256) void k_wsum_blocks(const float* x, int64_t K, int is_log, const float* lse,
                                                    int n_partials, float* lse_out, float log_k_total,
                                                    uint64_t* block_sums) {
  __shared__ uint64_t red[4];
  __shared__ float fred[8];
  float sm;
  const float mx = block_ref_max(is_log, lse, n_partials, fred, &sm);
  if (is_log == 2 && lse_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const float l = mx > -INFINITY ? mx + logf(sm) : -INFINITY;
    lse_out[0] = mx; lse_out[1] = sm; lse_out[2] = l; lse_out[3] = l - log_k_total;
  }
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < K) s += weight_q(x, base + k, is_log, mx);
  s = wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Second (final) pass: every block sums the totals of the blocks before it (nblocks u64 values, L2-resident:
// K/2048 of them) instead of waiting for a separate single-block scan kernel, then scans its own tile.
// The last block also publishes base_total = {0, sum of all weights}.
__global__ __launch_bounds__(256) void k_wscan_write(const float* x, int64_t K, int is_log, const float* lse,
                                                    int n_partials, const uint64_t* block_sums, uint64_t* cum,
                                                    uint64_t* base_total) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t red[4];
  __shared__ float fred[8];
  const float mx = block_ref_max(is_log, lse, n_partials, fred, nullptr);
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  // offset of this block = sum of block_sums[0 .. blockIdx.x)
  uint64_t pre = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) pre += block_sums[t];
  pre = wave_sum_u64(pre);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pre;
  uint64_t q[kScanItems];
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    q[k] = (base + k < K) ? weight_q(x, base + k, is_log, mx) : 0;
    s += q[k];
    q[k] = s;  // thread-local inclusive
  }
  uint64_t inc = wave_scan_u64(s);
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint64_t off = red[0] + red[1] + red[2] + red[3] + inc - s;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < K) cum[base + k] = off + q[k];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
    base_total[0] = 0;
    base_total[1] = off + s;
  }
}

// first i in [0, K) with cum[i] > t   (requires t < cum[K-1])
GJX_DEV int64_t upper_search(const uint64_t* __restrict__ cum, int64_t K, uint64_t t) {
  int64_t lo = 0, hi = K - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (cum[mid] > t) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// first i in [lo, hi] with cum[i] > t   (requires cum[hi] > t)
GJX_DEV int64_t upper_search_in(const uint64_t* __restrict__ cum, int64_t lo, int64_t hi, uint64_t t) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (cum[mid] > t) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Systematic resampling, particle-oriented ("expand"): particle i owns the output slots
// [J(base + cum[i-1]), J(base + cum[i])) — a few consecutive slots, usually 0..3 — so one lane per particle
// reads two prefix sums (coalesced), computes its slot range in O(1), and writes its index (and, with
// GATHER, its SoA rows) to those slots.  Consecutive particles own consecutive slot runs, so the writes
// coalesce; particles without offspring read nothing.  Particles with many offspring (degenerate weights)
// are filled cooperatively by the whole block.  No binary search, no dependent-load chain.
template <bool GATHER>
__global__ __launch_bounds__(256) void k_systematic_expand(const uint64_t* __restrict__ cum, int64_t K,
                                                          const uint64_t* base_total, double u, int64_t N_total,
                                                          int64_t out_begin, int64_t n_out, int32_t* ancestors,
                                                          const float* __restrict__ src, int64_t src_stride, int rows,
                                                          float* __restrict__ dst, int64_t dst_stride,
                                                          const int64_t* range_dev = nullptr) {
  constexpr int kOwn = 8;  // offspring a lane writes by itself
  if (range_dev) {         // sharded plan: {first slot, slot count} of this rank's run live on the device
    out_begin = range_dev[0];
    const int64_t n = range_dev[1];
    n_out = n < n_out ? n : n_out;   // n_out carries the capacity of ancestors[]
  }
  __shared__ int n_heavy;
  __shared__ int32_t h_i[256];
  __shared__ int64_t h_lo[256], h_hi[256];
  if (threadIdx.x == 0) n_heavy = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t base = base_total[0], total = base_total[1];
  const double step = (double)total / (double)N_total;
  const double inv_step = (double)N_total / (double)total;
  // J(base + cum[i]) once per lane; the lower end of a particle's run is its left neighbour's upper end
  const bool in = i < K && total > 0;
  const uint64_t c_cur = in ? base + cum[i] : 0;
  const int64_t j_cur = in ? slots_below(c_cur, u, step, inv_step, total, N_total) : 0;
  int64_t j_prev = __shfl_up((long long)j_cur, 1, 64);
  uint64_t c_prev = __shfl_up((unsigned long long)c_cur, 1, 64);
  if ((threadIdx.x & 63) == 0) {
    c_prev = base + ((in && i > 0) ? cum[i - 1] : 0);
    j_prev = in ? slots_below(c_prev, u, step, inv_step, total, N_total) : 0;
  }
  if (in) {
    if (c_cur > c_prev) {
      int64_t lo = j_prev - out_begin;
      int64_t hi = j_cur - out_begin;
      lo = lo < 0 ? 0 : lo;
      hi = hi > n_out ? n_out : hi;
      if (hi - lo > kOwn) {
        const int h = atomicAdd(&n_heavy, 1);
        h_i[h] = (int32_t)i; h_lo[h] = lo; h_hi[h] = hi;
      } else if (hi > lo) {
        if (ancestors) for (int64_t j = lo; j < hi; ++j) ancestors[j] = (int32_t)i;
        if (GATHER) {
          for (int r = 0; r < rows; ++r) {
            const float v = src[(int64_t)r * src_stride + i];
            for (int64_t j = lo; j < hi; ++j) dst[(int64_t)r * dst_stride + j] = v;
          }
        }
      }
    }
  }
  __syncthreads();
  const int nh = n_heavy;
  for (int h = 0; h < nh; ++h) {
    const int32_t pi = h_i[h];
    const int64_t lo = h_lo[h], hi = h_hi[h];
    if (ancestors) for (int64_t j = lo + threadIdx.x; j < hi; j += 256) ancestors[j] = pi;
    if (GATHER) {
      for (int r = 0; r < rows; ++r) {
        const float v = src[(int64_t)r * src_stride + pi];
        for (int64_t j = lo + threadIdx.x; j < hi; j += 256) dst[(int64_t)r * dst_stride + j] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// One-launch resampling indices for a single GPU: fixed-point weights, their prefix sums and the systematic
// ancestor expansion in ONE kernel (gjx_scan.h: tile_scan_expand).  Each block scans its tile in registers, publishes
// the tile total as one tagged 8-byte agent-scope granule, and then reads EVERY block's granule to get both its own
// offset and the grand total — an all-gather of <= 1024 words instead of two kernel boundaries and a 16 MB round trip
// of the prefix-sum array.  Requires all blocks co-resident: the launcher sizes the grid with the occupancy query.
// ------------------------------------------------------------------------------------------
template <int ITEMS>
__global__ __launch_bounds__(256) void k_resample_fused(const float* __restrict__ x, int64_t K, int mode,
                                                       const float* lse, int n_partials, float* lse_out,
                                                       float log_k_total, double u, int64_t N, int32_t* ancestors,
                                                       uint64_t* cum_out, uint64_t* base_total_out,
                                                       unsigned long long* agg, unsigned* ctrl) {
  __shared__ float fred[8];
  __shared__ ScanSmem sm;
  unsigned epoch;
  const unsigned long long tag = grid_tag(ctrl, &epoch);
  // the weights are requested before the reference maximum is reduced: their latency overlaps the reduction
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ITEMS;
  float xv[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) xv[k] = (i0 + k < K) ? x[i0 + k] : 0.0f;
  float sm_sum;
  const float mx = block_ref_max(mode, lse, n_partials, fred, &sm_sum);
  if (mode == 2 && lse_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const float l = mx > -INFINITY ? mx + logf(sm_sum) : -INFINITY;
    lse_out[0] = mx; lse_out[1] = sm_sum; lse_out[2] = l; lse_out[3] = l - log_k_total;
  }
  tile_scan_expand<ITEMS, false>(xv, mode, mx, i0, K, u, N, ancestors, cum_out, base_total_out, agg, tag, ctrl, epoch, true, sm);
}

// ------------------------------------------------------------------------------------------
// One-launch resampling + gather for a single GPU (N = K): weights -> systematic ancestors -> rows of the children,
// the ancestors never written to memory unless asked for.  Block b scans its own tile of 256 ITEMS fixed-point weights,
// the tile totals are all-gathered (the one rendezvous, as in k_resample_fused), and then the block is the CONSUMER of
// the output slots [b TILE, (b+1) TILE): each lane takes ITEMS consecutive slots, computes their comb thresholds T_j,
// finds the source tile of each by binary search in the prefix of the tile totals (LDS), and the block re-scans the few
// distinct source tiles its thresholds fall into (log-weights from L2, the next tile's loads in flight while the current
// one is searched); ancestor = first particle of the tile with inclusive cumulative weight > T_j.  Same ancestors as
// the slot-run expansion: particle i owns slot j  <=>  cum_excl(i) <= T_j < cum_incl(i).  Then the rows are copied,
// ITEMS slots per lane (16-byte stores at ITEMS = 4).  Balanced whatever the weights are; a window that touches many
// source tiles (collapsed weights) just loops longer.
// ------------------------------------------------------------------------------------------
constexpr int kGatherMaxTiles = 1024;

template <int ITEMS>
__global__ __launch_bounds__(256) void k_resample_gather(const float* __restrict__ x, int64_t K, int mode,
                                                        const float* lse, int n_partials, float* lse_out,
                                                        float log_k_total, double u, const float* __restrict__ src,
                                                        int64_t src_stride, int rows, float* __restrict__ dst,
                                                        int64_t dst_stride, int32_t* ancestors,
                                                        unsigned long long* agg, unsigned* ctrl,
                                                        unsigned long long* timeline) {
#define GJX_STAMP(n) do { if (timeline && threadIdx.x == 0) timeline[blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GJX_STAMP(0);
  constexpr int TILE = 256 * ITEMS;
  __shared__ float fred[8];
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t P[kGatherMaxTiles + 1];   // P[t] = total weight of tiles < t; P[nb] = grand total
  constexpr int CH = 3;                          // source tiles re-scanned per round
  __shared__ uint64_t cumL[CH * TILE];          // inclusive cumulative weights (absolute) of the tiles being searched
  __shared__ uint64_t s_wtot[CH][4];
  __shared__ int s_range[2];                    // first and last source tile of this block's slots
  unsigned epoch;
  const unsigned long long tag = grid_tag(ctrl, &epoch);
  const int nb = (int)gridDim.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ITEMS;   // first particle this lane scans
  auto load_tile = [&](int64_t p0, float (&v)[ITEMS]) {
    if (ITEMS == 4 && p0 + 4 <= K) {
      const float4 q4 = *(const float4*)(x + p0);
      v[0] = q4.x; v[1] = q4.y; v[2] = q4.z; v[3] = q4.w;
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (p0 + k < K) ? x[p0 + k] : 0.0f;
    }
  };
  // the block's own tile and its two neighbours: with weights that are not collapsed the slots of block b draw from
  // tiles b - 1 .. b + 1, so their cumulative weights are laid out in LDS while the tile totals travel (below)
  const int w0 = (int)blockIdx.x - 1;            // first tile of the prefetched window
  float xv[ITEMS], xn[2][ITEMS];
  load_tile(i0, xv);
  float sm_sum;
  const float mx = block_ref_max(mode, lse, n_partials, fred, &sm_sum);
  if (mode == 2 && lse_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const float l = mx > -INFINITY ? mx + logf(sm_sum) : -INFINITY;
    lse_out[0] = mx; lse_out[1] = sm_sum; lse_out[2] = l; lse_out[3] = l - log_k_total;
  }
  // ---- own tile total -> all-gather ----
  uint64_t qown[ITEMS], sown = 0, incown;
  {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { sown += (i0 + k < K) ? weight_q(xv, k, mode, mx) : 0; qown[k] = sown; }
    incown = wave_scan_u64(sown);
    if (lane == 63) { wsum[wid] = incown; s_wtot[1][wid] = incown; }
    __syncthreads();
    if (threadIdx.x == 0) grid_publish(agg, tag, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  }
  GJX_STAMP(1);
  // ---- while the totals are in flight: cumulative weights of tiles b - 1, b, b + 1 RELATIVE to each tile's start
  //      (the neighbours' log-weights are requested only now: nothing may delay the publication above, every block waits
  //      for the last one) ----
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int tc = w0 + 2 * c;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) xn[c][k] = 0.0f;
    if (tc >= 0 && tc < nb) load_tile((int64_t)tc * TILE + (int64_t)threadIdx.x * ITEMS, xn[c]);
  }
  {
    uint64_t qn[2][ITEMS], sn[2], incn[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int tc = w0 + 2 * c;
      const bool on = tc >= 0 && tc < nb;
      const int64_t p0 = (int64_t)tc * TILE + (int64_t)threadIdx.x * ITEMS;
      sn[c] = 0;
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) { sn[c] += (on && p0 + k < K) ? weight_q(xn[c], k, mode, mx) : 0; qn[c][k] = sn[c]; }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) incn[c] = wave_scan_u64(sn[c]);
    if (lane == 63) { s_wtot[0][wid] = incn[0]; s_wtot[2][wid] = incn[1]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uint64_t base = c == 1 ? incown - sown : incn[c >> 1] - sn[c >> 1];
      for (int w = 0; w < wid; ++w) base += s_wtot[c][w];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) cumL[c * TILE + threadIdx.x * ITEMS + k] = base + (c == 1 ? qown[k] : qn[c >> 1][k]);
    }
  }
  grid_gather(agg, tag, ctrl, [&](int b, unsigned long long val) { P[b + 1] = val; });
  if (threadIdx.x == 0) P[0] = 0;
  __syncthreads();
  GJX_STAMP(2);
  {   // prefix of the tile totals, in place: lane t owns entries [t per, (t+1) per)
    const int per = (nb + 255) >> 8;
    const int e0 = threadIdx.x * per, e1 = (e0 + per) < nb ? (e0 + per) : nb;
    uint64_t loc = 0;
    for (int e = e0; e < e1; ++e) loc += P[e + 1];
    uint64_t inc = wave_scan_u64(loc);
    if (lane == 63) wsum[wid] = inc;     // the publish above read wsum before the barrier after the gather
    __syncthreads();
    uint64_t run = inc - loc;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    for (int e = e0; e < e1; ++e) { run += P[e + 1]; P[e + 1] = run; }
    __syncthreads();
  }
  const uint64_t total = P[nb];
  GJX_STAMP(6);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(&ctrl[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (total == 0) __hip_atomic_fetch_or(&ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- ancestors of slots i0 .. i0 + ITEMS - 1 (consecutive slots per lane: 16-byte row stores at ITEMS = 4; the
  //      striped assignment, 64 consecutive slots per wave instruction, measured 1.3 us slower at K = 2^20) ----
  int32_t anc[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) anc[k] = (int32_t)((i0 + k < K) ? i0 + k : 0);   // dead collection: identity (flagged)
  if (total > 0) {   // block-uniform
    const double step = (double)total / (double)K;
    uint64_t T[ITEMS];
    int tile[ITEMS], kpos[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int64_t j = (i0 + k < K) ? i0 + k : K - 1;
      T[k] = comb_threshold(j, u, step, total);
      tile[k] = 0;
    }
    // first tile t with P[t + 1] > T = number of tiles with inclusive prefix <= T: fixed-trip binary descent for the
    // lane's first and last slot together (independent LDS reads per round); the slots between them are in the same
    // tile unless the lane straddles a tile boundary (then they are searched too)
    auto descend = [&](int k0, int k1) {
#pragma unroll
      for (int sft = kGatherMaxTiles >> 1; sft >= 1; sft >>= 1) {
        const int pa = tile[k0] + sft, pb = tile[k1] + sft;     // P[probe] = inclusive prefix of tile probe - 1
        if (pa <= nb - 1 && P[pa] <= T[k0]) tile[k0] = pa;
        if (k1 != k0 && pb <= nb - 1 && P[pb] <= T[k1]) tile[k1] = pb;
      }
    };
    // A block's slots draw from tiles near its own index (equal tile totals would make it exactly its own): the nine
    // boundaries of the eight tiles around blockIdx.x are read together (one LDS latency, the same addresses in every
    // lane) and the tile of each slot is counted off; only a lane with a threshold outside that window descends
    const int wlo = (int)blockIdx.x - 4 < 0 ? 0 : ((int)blockIdx.x - 4 > nb - 8 ? (nb - 8 < 0 ? 0 : nb - 8) : (int)blockIdx.x - 4);
    uint64_t Pw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Pw[k] = P[wlo + k < nb ? wlo + k : nb];
    if (T[0] >= Pw[0] && T[ITEMS - 1] < Pw[8]) {            // thresholds are non-decreasing in the slot index
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        int tl = wlo;
#pragma unroll
        for (int w = 1; w < 8; ++w) tl += Pw[w] <= T[k] ? 1 : 0;
        tile[k] = tl;
      }
    } else {
      descend(0, ITEMS - 1);
      if (ITEMS > 2) {
        if (tile[0] == tile[ITEMS - 1]) {
#pragma unroll
          for (int k = 1; k < ITEMS - 1; ++k) tile[k] = tile[0];
        } else {
          descend(1, ITEMS > 3 ? 2 : 1);
        }
      }
    }
    if (timeline) { asm volatile("" :: "v"(tile[0]), "v"(tile[ITEMS - 1])); GJX_STAMP(7); }
    // the block's source tiles: the tile index is non-decreasing in the slot index, so they are the range between the
    // first and the last slot's tile — no list to build, the two ends travel through LDS behind one barrier.  A round
    // starts at a tile that has weight (block-uniform skip; the range's last tile always has), so collapsed weights far
    // apart cost as many rounds as there are live tiles
    if (threadIdx.x == 0) s_range[0] = tile[0];
    if (threadIdx.x == 255) s_range[1] = tile[ITEMS - 1];
    __syncthreads();
    const int tmin = s_range[0], ntiles = s_range[1] - tmin + 1;
    if (tmin >= w0 && s_range[1] <= w0 + 2) {       // block-uniform: every source tile is in the prefetched window
      GJX_STAMP(3);
      int pos[ITEMS];
      const uint64_t* cm[ITEMS];
      uint64_t Tr[ITEMS];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) { cm[k] = cumL + (tile[k] - w0) * TILE; Tr[k] = T[k] - P[tile[k]]; pos[k] = 0; }
#pragma unroll
      for (int q = TILE >> 2; q >= 1; q >>= 2) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
          const uint64_t pa = cm[k][pos[k] + q - 1], pb = cm[k][pos[k] + 2 * q - 1], pc = cm[k][pos[k] + 3 * q - 1];
          pos[k] += (pa <= Tr[k] ? q : 0) + (pb <= Tr[k] ? q : 0) + (pc <= Tr[k] ? q : 0);
        }
      }
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) anc[k] = (int32_t)((int64_t)tile[k] * TILE + pos[k]);
    } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) kpos[k] = tile[k] - tmin;
    auto next_live = [&](int at) { while (at < ntiles && P[tmin + at + 1] == P[tmin + at]) ++at; return at; };
    GJX_STAMP(3);
    // re-scan the source tiles, CH at a time (a window of TILE slots rarely touches more than 3)
    float nv[CH][ITEMS];
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < ntiles) load_tile((int64_t)(tmin + c) * TILE + (int64_t)threadIdx.x * ITEMS, nv[c]);
    for (int idx = 0; idx < ntiles;) {
      const int nidx = next_live(idx + CH);
      uint64_t qi[CH][ITEMS], sacc[CH], inc[CH];
      int tsrc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const bool on = idx + c < ntiles;
        tsrc[c] = on ? tmin + idx + c : 0;
        const int64_t p0 = (int64_t)tsrc[c] * TILE + (int64_t)threadIdx.x * ITEMS;
        sacc[c] = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { sacc[c] += (on && p0 + k < K) ? weight_q(nv[c], k, mode, mx) : 0; qi[c][k] = sacc[c]; }
        if (nidx + c < ntiles) load_tile((int64_t)(tmin + nidx + c) * TILE + (int64_t)threadIdx.x * ITEMS, nv[c]);   // next round, in flight
        inc[c] = sacc[c];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) inc[c] = wave_scan_u64(inc[c]);     // three independent DPP chains
      if (lane == 63) {
#pragma unroll
        for (int c = 0; c < CH; ++c) s_wtot[c][wid] = inc[c];
      }
      __syncthreads();           // also: every lane is done searching the previous round's cumL
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        uint64_t base = P[tsrc[c]] + (inc[c] - sacc[c]);
        for (int w = 0; w < wid; ++w) base += s_wtot[c][w];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) cumL[c * TILE + threadIdx.x * ITEMS + k] = base + qi[c][k];
      }
      __syncthreads();
      // first particle p of the slot's tile with cum_incl(p) > T
      int pos[ITEMS];
      const uint64_t* cm[ITEMS];
      bool mine[ITEMS];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        mine[k] = kpos[k] >= idx && kpos[k] < idx + CH;
        cm[k] = cumL + (mine[k] ? (kpos[k] - idx) * TILE : 0);
        pos[k] = 0;
      }
      // (number of entries <= T, 4-ary: three independent probes per level and slot, 5 LDS latencies for 1024 entries)
#pragma unroll
      for (int q = TILE >> 2; q >= 1; q >>= 2) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
          const uint64_t pa = cm[k][pos[k] + q - 1], pb = cm[k][pos[k] + 2 * q - 1], pc = cm[k][pos[k] + 3 * q - 1];
          pos[k] += (pa <= T[k] ? q : 0) + (pb <= T[k] ? q : 0) + (pc <= T[k] ? q : 0);
        }
      }
#pragma unroll
      for (int k = 0; k < ITEMS; ++k)
        if (mine[k]) anc[k] = (int32_t)((int64_t)tile[k] * TILE + pos[k]);
      idx = nidx;
    }
    }
  }
  GJX_STAMP(4);
  // if a rendezvous timed out (grid not co-resident) the totals are partial and the ancestors undefined (the caller
  // repeats the call): keep them inside the rows whatever happened — one v_med3_i32 per slot
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) anc[k] = max(0, min(anc[k], (int32_t)(K - 1)));
  // ---- children: rows of the ancestors, ITEMS consecutive slots per lane ----
  const bool whole = i0 + ITEMS <= K;
  if (ancestors) {
    if (ITEMS == 4 && whole) *(int4*)(ancestors + i0) = make_int4(anc[0], anc[1], anc[2], anc[3]);
    else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) if (i0 + k < K) ancestors[i0 + k] = anc[k];
    }
  }
  const bool vec = ITEMS == 4 && whole && (dst_stride & 3) == 0 && (((uintptr_t)dst) & 15) == 0;
  if (vec) {
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      const float* sr = src + (int64_t)r * src_stride;
      float4 v;
      v.x = sr[anc[0]]; v.y = sr[anc[ITEMS > 1 ? 1 : 0]]; v.z = sr[anc[ITEMS > 2 ? 2 : 0]]; v.w = sr[anc[ITEMS > 3 ? 3 : 0]];
      *(float4*)(dst + (int64_t)r * dst_stride + i0) = v;
    }
  } else {
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      const float* sr = src + (int64_t)r * src_stride;
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) if (i0 + k < K) dst[(int64_t)r * dst_stride + i0 + k] = sr[anc[k]];
    }
  }
  GJX_STAMP(5);
#undef GJX_STAMP
}

// ------------------------------------------------------------------------------------------
// Resampling + gather under the TILE-SCALED scheme (GJX_WEIGHTS_TILE_SCALED, include/gjx.h), N = K: a PLAIN kernel — no
// rendezvous, no co-resident grid, any K up to kGatherTiledMaxTiles tiles.  The {e_b, S_b} of every 1024-particle tile
// exist before the launch: the propagate kernel leaves them beside its LSE partials when a block of it covers whole
// tiles (gjx_run_program, k_run_gmm_flat<.., TILES>), otherwise k_tile_totals computes them (one small launch).  Every
// block reads all of them (12 B per tile), takes the maximum exponent, shifts, prefixes, and is the consumer of the
// output slots [b 1024, (b+1) 1024) exactly as k_resample_gather: thresholds, source tile in a window of the prefix,
// cumulative q of tiles b-1 .. b+1 laid out in LDS from the log-weights (quantised against each tile's OWN exponent),
// 4-ary search of the residual, rows copied.  Every load the head needs is issued at the top: the head is one memory
// latency plus LDS work instead of two fabric round trips.  Ancestors are those of gjx_resample_indices_tiled bit for bit.
// ------------------------------------------------------------------------------------------
constexpr int kGatherTiledMaxTiles = 65536;

// {S_b, E_b} of every tile from the log-weights (what k_tiled_quantise computes, without the cumulative array)
__global__ __launch_bounds__(kTileQ) void k_tile_totals(const float* __restrict__ logw, int64_t K, uint64_t* S, int32_t* E) {
  constexpr int NW = kTileQ / 64;
  __shared__ float fred[NW];
  __shared__ uint64_t wsum[NW];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kTileQ + threadIdx.x;
  const float lw = i < K ? logw[i] : -INFINITY;
  const float wm = wave_max(lw);
  if (lane == 0) fred[wid] = wm;
  __syncthreads();
  float bm = fred[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) bm = fmaxf(bm, fred[w]);
  const int e = tile_exponent(bm);
  const uint64_t tot_w = wave_total_u64(i < K ? tile_q(lw, e) : 0);
  if (lane == 0) wsum[wid] = tot_w;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t tot = 0;
    for (int w = 0; w < NW; ++w) tot += wsum[w];
    S[blockIdx.x] = tot;
    E[blockIdx.x] = tot ? e : kTileDead;
  }
}

// PLANNED (more than 1024 tiles): the maximum exponent, the shifts and the prefix were computed ONCE by k_tiled_plan (one
// small launch) and are read from memory — a block touches the nine prefix entries around its own index — instead of
// every block reducing all nt granules again (O(nt^2) reads and 72 KB of LDS at nt = 4096).
template <int ITEMS, bool PLANNED>
__global__ __launch_bounds__(256) void k_resample_gather_tiled(const float* __restrict__ x, int64_t K, const uint64_t* __restrict__ S,
                                                              const int32_t* __restrict__ E, const uint64_t* __restrict__ Pg,
                                                              const int32_t* __restrict__ shg, int lse_mode, const float* lse,
                                                              int n_partials, float* lse_out, float log_k_total, double u,
                                                              const float* __restrict__ src, int64_t src_stride, int rows,
                                                              float* __restrict__ dst, int64_t dst_stride, int32_t* ancestors,
                                                              unsigned* ctrl, unsigned long long* timeline) {
#define GJX_STAMP(n) do { if (timeline && threadIdx.x == 0) timeline[blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GJX_STAMP(0);
  static_assert(ITEMS == 4, "a block consumes one quantisation tile: 256 lanes x 4 slots");
  extern __shared__ __align__(16) unsigned char gt_dyn[];
  const int nt = (int)gridDim.x;
  uint64_t* const Pl = (uint64_t*)gt_dyn;                      // [nt + 1] prefix of the shifted tile totals (!PLANNED)
  int32_t* const Ebl = (int32_t*)(Pl + ((nt + 2) & ~1));       // [nt] tile exponents (!PLANNED)
  __shared__ TiledSearchShared sh;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ITEMS;
  int32_t anc[ITEMS];
  tiled_search_tile<PLANNED>(x, K, S, E, Pg, shg, nt, (int)blockIdx.x, Pl, Ebl, sh, lse_mode, lse, n_partials, lse_out, log_k_total, u, ctrl, timeline, anc);
  GJX_STAMP(4);
  // ---- children: rows of the ancestors, ITEMS consecutive slots per lane ----
  const bool whole = i0 + ITEMS <= K;
  if (ancestors) {
    if (whole) *(int4*)(ancestors + i0) = make_int4(anc[0], anc[1], anc[2], anc[3]);
    else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) if (i0 + k < K) ancestors[i0 + k] = anc[k];
    }
  }
  const bool vec = whole && (dst_stride & 3) == 0 && (((uintptr_t)dst) & 15) == 0;
  if (vec) {
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      const float* sr = src + (int64_t)r * src_stride;
      float4 v;
      v.x = sr[anc[0]]; v.y = sr[anc[1]]; v.z = sr[anc[2]]; v.w = sr[anc[3]];
      *(float4*)(dst + (int64_t)r * dst_stride + i0) = v;
    }
  } else {
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      const float* sr = src + (int64_t)r * src_stride;
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) if (i0 + k < K) dst[(int64_t)r * dst_stride + i0 + k] = sr[anc[k]];
    }
  }
  GJX_STAMP(5);
#undef GJX_STAMP
}

__global__ __launch_bounds__(256) void k_multinomial(const uint64_t* cum, int64_t K, const uint64_t* base_total,
                                                    key2 key, int64_t out_begin, int64_t n_out, int32_t* ancestors) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const uint64_t base = base_total[0], total = base_total[1];
  const key2 h = fold_in64(key, (uint64_t)(out_begin + j));
  const uint64_t r = (((uint64_t)h.a << 32) | h.b) >> 11;
  const double uj = (double)r * (1.0 / 9007199254740992.0);
  const uint64_t T = (uint64_t)(uj * (double)total);
  int32_t a = -1;
  if (K > 0) {
    const uint64_t local = cum[K - 1];
    if (T >= base && T < base + local) a = (int32_t)upper_search(cum, K, T - base);
  }
  ancestors[j] = a;
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ src, int64_t src_stride,
                                                    const int32_t* __restrict__ anc, int64_t n_out, int rows,
                                                    float* __restrict__ dst, int64_t dst_stride) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const int32_t a = anc[j];
  // a < 0: a slot another rank fills (prefilled -1).  a >= src_stride cannot come from a completed resampling, but the
  // ancestors of a co-resident kernel that timed out are undefined (the caller repeats the call): never read outside the rows
  if ((uint64_t)(int64_t)a >= (uint64_t)src_stride) return;
  for (int r = 0; r < rows; ++r) dst[(int64_t)r * dst_stride + j] = src[(int64_t)r * src_stride + a];
}


// planned expansion for a sharded collection (gjx_shard.hip): slot run read from the device plan
int launch_expand_planned(const uint64_t* cum, int64_t K, const gjx_shard_plan* plan_dev, double u, int64_t N_total,
                          int32_t* ancestors, int64_t anc_capacity, hipStream_t st) {
  const uint64_t* bt = reinterpret_cast<const uint64_t*>(plan_dev);                       // {base, total}
  const int64_t* range = reinterpret_cast<const int64_t*>(plan_dev) + 2;                  // {slot0, n_valid}
  hipLaunchKernelGGL(k_systematic_expand<false>, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, cum, K, bt, u, N_total,
                     (int64_t)0, anc_capacity, ancestors, (const float*)nullptr, (int64_t)0, 0, (float*)nullptr, (int64_t)0,
                     range);
  GJX_CHECK_LAUNCH("gjx_shard_resample(expand)");
  return GJX_OK;
}

}  // namespace gjx

using namespace gjx;

extern "C" int gjx_categorical_pick(const float* logw, int64_t K, int64_t particle_offset, const float* lse,
                                    uint32_t key0, uint32_t key1, int32_t rng_mode, void* out_dev,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!logw || !lse || !out_dev || K <= 0) return gjx_fail(GJX_EINVAL, "gjx_categorical_pick: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_PICK, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_categorical_pick: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (K + 1023) / 1024;
  const int nblocks = (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
  PickPair* partials = (PickPair*)((char*)workspace + kWsHeaderBytes);
  if (rng_mode == GJX_RNG_JAX32)
    hipLaunchKernelGGL(k_pick_partial<GJX_RNG_JAX32>, dim3(nblocks), dim3(256), 0, st, logw, K, particle_offset, lse, key2{key0, key1}, partials);
  else
    hipLaunchKernelGGL(k_pick_partial<GJX_RNG_FLAT>, dim3(nblocks), dim3(256), 0, st, logw, K, particle_offset, lse, key2{key0, key1}, partials);
  GJX_CHECK_LAUNCH("gjx_categorical_pick/partial");
  hipLaunchKernelGGL(k_pick_finish, dim3(1), dim3(256), 0, st, (const PickPair*)partials, nblocks, (PickPair*)out_dev);
  GJX_CHECK_LAUNCH("gjx_categorical_pick/finish");
  return GJX_OK;
}

extern "C" int gjx_trials_lse_pick(const float* logw, int64_t n_trials, int64_t K, int64_t particle_offset, uint32_t key0, uint32_t key1,
                                   int32_t rng_mode, float* lse_out, int32_t* pick_out, void* stream) {
  if (!logw || !lse_out || n_trials <= 0 || K <= 0 || n_trials > 0x7fffffffLL || particle_offset < 0 ||
      (pick_out && particle_offset + n_trials * K > 0x7fffffffLL))
    return gjx_fail(GJX_EINVAL, "gjx_trials_lse_pick: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const float log_k = (float)log((double)K);
  if (rng_mode == GJX_RNG_JAX32)
    hipLaunchKernelGGL(k_trials_lse_pick<GJX_RNG_JAX32>, dim3((unsigned)n_trials), dim3(256), 0, st, logw, K, particle_offset, key2{key0, key1},
                       pick_out ? 1 : 0, log_k, lse_out, pick_out);
  else
    hipLaunchKernelGGL(k_trials_lse_pick<GJX_RNG_FLAT>, dim3((unsigned)n_trials), dim3(256), 0, st, logw, K, particle_offset, key2{key0, key1},
                       pick_out ? 1 : 0, log_k, lse_out, pick_out);
  GJX_CHECK_LAUNCH("gjx_trials_lse_pick");
  return GJX_OK;
}

extern "C" int gjx_weight_cumsum(const float* x, int64_t K, int32_t is_log, const float* lse, int32_t n_partials,
                                 uint64_t* cum, uint64_t* base_total_dev, float* lse_out, int64_t K_total,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !cum || !base_total_dev || K <= 0 || is_log < 0 || is_log > 2 || (is_log && !lse) || (is_log == 2 && n_partials <= 0))
    return gjx_fail(GJX_EINVAL, "gjx_weight_cumsum: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_weight_cumsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblocks = (int)((K + kScanTile - 1) / kScanTile);
  uint64_t* bs = (uint64_t*)((char*)workspace + kWsHeaderBytes);
  const float log_k = (float)log((double)(K_total > 0 ? K_total : K));
  hipLaunchKernelGGL(k_wsum_blocks, dim3(nblocks), dim3(256), 0, st, x, K, (int)is_log, lse, (int)n_partials, lse_out, log_k, bs);
  GJX_CHECK_LAUNCH("gjx_weight_cumsum/sum");
  hipLaunchKernelGGL(k_wscan_write, dim3(nblocks), dim3(256), 0, st, x, K, (int)is_log, lse, (int)n_partials, (const uint64_t*)bs, cum, base_total_dev);
  GJX_CHECK_LAUNCH("gjx_weight_cumsum/write");
  return GJX_OK;
}

extern "C" int gjx_resample_systematic(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev, double u,
                                       int64_t N_total, int64_t out_begin, int64_t n_out, int32_t* ancestors,
                                       void* stream) {
  if (!cum || !base_total_dev || !ancestors || K <= 0 || N_total <= 0 || n_out < 0 || !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_resample_systematic: bad argument");
  if (n_out == 0) return GJX_OK;
  hipLaunchKernelGGL(k_systematic_expand<false>, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cum, K,
                     base_total_dev, u, N_total, out_begin, n_out, ancestors, (const float*)nullptr, (int64_t)0, 0,
                     (float*)nullptr, (int64_t)0);
  GJX_CHECK_LAUNCH("gjx_resample_systematic");
  return GJX_OK;
}

static int env_items() {
  const char* e = getenv("GJX_RESAMPLE_ITEMS");
  const int v = e ? atoi(e) : 0;
  return (v == 1 || v == 4 || v == 16) ? v : 0;
}

extern "C" int gjx_resample_indices(const float* x, int64_t K, int32_t is_log, const float* lse, int32_t n_partials,
                                    double u, int64_t N, int32_t* ancestors, uint64_t* cum, uint64_t* base_total_dev,
                                    float* lse_out, int64_t K_total, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || K <= 0 || N <= 0 || is_log < 0 || is_log > 2 || (is_log && !lse) || (is_log == 2 && n_partials <= 0) ||
      !(u >= 0.0 && u < 1.0) || (!ancestors && !cum))
    return gjx_fail(GJX_EINVAL, "gjx_resample_indices: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_resample_indices: workspace too small");
  // One co-resident grid: the blocks all-gather their tile totals through memory, so every block must be running.
  // The capacity comes from the occupancy query (70 / 118 registers per lane at 4 / 16 particles per lane, 30 KB of
  // LDS: 4 blocks per CU on a full MI355X, fewer on a partitioned device); 1 particle per lane while the grid fits
  // (4x the waves to hide the round trips), then 4, then 16; beyond that the three-launch path runs.
  int items = env_items();
  int64_t nblocks = 0;
  const int cand[3] = {1, 4, 16};
  bool fits = false;
  for (int c = 0; c < 3 && !fits; ++c) {
    if (items && cand[c] != items) continue;
    const void* fn = cand[c] == 1 ? (const void*)k_resample_fused<1> : cand[c] == 4 ? (const void*)k_resample_fused<4> : (const void*)k_resample_fused<16>;
    const int cap = gjx_coresident_blocks(fn, 256, 0);
    nblocks = (K + 256 * (int64_t)cand[c] - 1) / (256 * (int64_t)cand[c]);
    if (nblocks <= cap) { items = cand[c]; fits = true; }
  }
  hipStream_t st = (hipStream_t)stream;
  if (!fits || !ancestors) {
    if (!cum || !base_total_dev) return gjx_fail(GJX_EUNSUPPORTED, "gjx_resample_indices: the grid would not be co-resident on this device and there are no cum/base_total buffers for the three-launch path");
    int rc = gjx_weight_cumsum(x, K, is_log, lse, n_partials, cum, base_total_dev, lse_out, K_total, workspace, workspace_bytes, stream);
    if (rc || !ancestors) return rc;
    return gjx_resample_systematic(cum, K, base_total_dev, u, N, 0, N, ancestors, stream);
  }
  unsigned* ctrl = (unsigned*)workspace + 8;  // control block words 8..10: epoch, -, error (0..7 belong to the LSE ticket)
  unsigned long long* agg = (unsigned long long*)((char*)workspace + kWsHeaderBytes);
  const float log_k = (float)log((double)(K_total > 0 ? K_total : K));
#define GJX_RF(IT) hipLaunchKernelGGL((k_resample_fused<IT>), dim3((unsigned)nblocks), dim3(256), 0, st, x, K, (int)is_log, lse, \
                                      (int)n_partials, lse_out, log_k, u, N, ancestors, cum, base_total_dev, agg, ctrl)
  if (items == 1) GJX_RF(1); else if (items == 4) GJX_RF(4); else GJX_RF(16);
#undef GJX_RF
  GJX_CHECK_LAUNCH("gjx_resample_indices");
  return GJX_OK;
}

extern "C" int gjx_resample_gather(const float* x, int64_t K, int32_t is_log, const float* lse, int32_t n_partials, double u,
                                   const float* src, int64_t src_stride, int32_t rows, float* dst, int64_t dst_stride,
                                   int32_t* ancestors, float* lse_out, int64_t K_total, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (!x || K <= 0 || is_log < 0 || is_log > 2 || (is_log && !lse) || (is_log == 2 && n_partials <= 0) ||
      !(u >= 0.0 && u < 1.0) || rows < 0 || (rows > 0 && (!src || !dst)) || K > 0x7fffffffLL)
    return gjx_fail(GJX_EINVAL, "gjx_resample_gather: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_resample_gather: workspace too small");
  int items = 0;
  int64_t nblocks = 0;
  const int cand[2] = {1, 4};
  const int forced = env_items();
  for (int c = 0; c < 2 && !items; ++c) {
    if (forced && cand[c] != forced) continue;
    const void* fn = cand[c] == 1 ? (const void*)k_resample_gather<1> : (const void*)k_resample_gather<4>;
    const int cap = gjx_coresident_blocks(fn, 256, 0);
    nblocks = (K + 256 * (int64_t)cand[c] - 1) / (256 * (int64_t)cand[c]);
    if (nblocks <= cap && nblocks <= kGatherMaxTiles) items = cand[c];
  }
  if (!items) return gjx_fail(GJX_EUNSUPPORTED, "gjx_resample_gather: K is too large for one co-resident grid on this device (use gjx_resample_indices + gjx_gather_rows)");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctrl = (unsigned*)workspace + 8;
  unsigned long long* agg = (unsigned long long*)((char*)workspace + kWsHeaderBytes);
  const float log_k = (float)log((double)(K_total > 0 ? K_total : K));
  unsigned long long* timeline = nullptr;   // debug: per-block phase stamps (profiles/microbench/gather_timeline.py)
  timeline = gjx::debug_timeline(64 * (size_t)nblocks);
#define GJX_RG(IT) hipLaunchKernelGGL((k_resample_gather<IT>), dim3((unsigned)nblocks), dim3(256), 0, st, x, K, (int)is_log, lse, \
                                      (int)n_partials, lse_out, log_k, u, src, src_stride, (int)rows, dst, dst_stride, ancestors, agg, ctrl, timeline)
  if (items == 1) GJX_RG(1); else GJX_RG(4);
#undef GJX_RG
  GJX_CHECK_LAUNCH("gjx_resample_gather");
  return GJX_OK;
}

extern "C" int gjx_resample_gather_tiled(const float* logw, int64_t K, const uint64_t* tile_S, const int32_t* tile_E, int32_t lse_mode,
                                         const float* lse, int32_t n_partials, double u, const float* src, int64_t src_stride,
                                         int32_t rows, float* dst, int64_t dst_stride, int32_t* ancestors, float* lse_out,
                                         int64_t K_total, void* workspace, size_t workspace_bytes, void* stream) {
  if (!logw || K <= 0 || !(u >= 0.0 && u < 1.0) || rows < 0 || (rows > 0 && (!src || !dst)) || K > 0x7fffffffLL ||
      (lse_mode != 0 && lse_mode != 2) || (lse_mode == 2 && (!lse || n_partials <= 0)) || ((tile_S == nullptr) != (tile_E == nullptr)))
    return gjx_fail(GJX_EINVAL, "gjx_resample_gather_tiled: bad argument");
  const int64_t nt = (K + kTileQ - 1) / kTileQ;
  if (nt > kGatherTiledMaxTiles) return gjx_fail(GJX_EUNSUPPORTED, "gjx_resample_gather_tiled: more than 65536 tiles (K > 2^26): use gjx_resample_indices_tiled + gjx_gather_rows");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K) || kWsHeaderBytes + 12 * (size_t)nt > workspace_bytes)
    return gjx_fail(GJX_EWORKSPACE, "gjx_resample_gather_tiled: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ctrl = (unsigned*)workspace + 8;
  if (!tile_S) {   // the producer left no tile totals: one small launch over the log-weights
    uint64_t* S = (uint64_t*)((char*)workspace + kWsHeaderBytes);
    int32_t* E = (int32_t*)(S + nt);
    hipLaunchKernelGGL(k_tile_totals, dim3((unsigned)nt), dim3(kTileQ), 0, st, logw, K, S, E);
    GJX_CHECK_LAUNCH("gjx_resample_gather_tiled/totals");
    tile_S = S; tile_E = E;
  }
  const float log_k = (float)log((double)(K_total > 0 ? K_total : K));
  unsigned long long* timeline = gjx::debug_timeline(64 * (size_t)nt);
  if (nt > 1024 || getenv("GJX_TILED_PLANNED")) {       // (the variable: tests run the planned form at small sizes)
    // many tiles: maximum exponent, shifts and prefix once (k_tiled_plan), behind the tile totals in the workspace
    uint64_t* Pg = (uint64_t*)((char*)workspace + kWsHeaderBytes + 12 * (size_t)nt + 16);
    Pg = (uint64_t*)(((uintptr_t)Pg + 15) & ~(uintptr_t)15);
    int32_t* shg = (int32_t*)(Pg + nt + 1);
    if ((char*)(shg + nt) > (char*)workspace + workspace_bytes) return gjx_fail(GJX_EWORKSPACE, "gjx_resample_gather_tiled: workspace too small");
    const int rc = gjx::launch_tiled_plan(tile_S, tile_E, (int)nt, Pg, shg, ctrl, st);
    if (rc) return rc;
    hipLaunchKernelGGL((k_resample_gather_tiled<4, true>), dim3((unsigned)nt), dim3(256), 0, st, logw, K, tile_S, tile_E, (const uint64_t*)Pg,
                       (const int32_t*)shg, (int)lse_mode, lse, (int)n_partials, lse_out, log_k, u, src, src_stride, (int)rows, dst, dst_stride,
                       ancestors, ctrl, timeline);
  } else {
    const size_t lds = 8 * (size_t)((nt + 2) & ~1) + 4 * (size_t)nt;
    hipLaunchKernelGGL((k_resample_gather_tiled<4, false>), dim3((unsigned)nt), dim3(256), lds, st, logw, K, tile_S, tile_E, (const uint64_t*)nullptr,
                       (const int32_t*)nullptr, (int)lse_mode, lse, (int)n_partials, lse_out, log_k, u, src, src_stride, (int)rows, dst, dst_stride,
                       ancestors, ctrl, timeline);
  }
  GJX_CHECK_LAUNCH("gjx_resample_gather_tiled");
  return GJX_OK;
}

extern "C" int gjx_resample_gather_systematic(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev, double u,
                                              int64_t N_total, int64_t out_begin, int64_t n_out, const float* src,
                                              int64_t src_stride, int32_t rows, float* dst, int64_t dst_stride,
                                              int32_t* ancestors, void* stream) {
  if (!cum || !base_total_dev || !src || !dst || !ancestors || K <= 0 || N_total <= 0 || n_out < 0 || rows < 0 ||
      !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_resample_gather_systematic: bad argument");
  if (n_out == 0) return GJX_OK;
  // particle-oriented expand of the ancestor indices (no search), then the slot-oriented, fully coalesced row copy
  const int rc = gjx_resample_systematic(cum, K, base_total_dev, u, N_total, out_begin, n_out, ancestors, stream);
  if (rc) return rc;
  return gjx_gather_rows(src, src_stride, ancestors, n_out, rows, dst, dst_stride, stream);
}

extern "C" int gjx_resample_multinomial(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev,
                                        uint32_t key0, uint32_t key1, int64_t N_total, int64_t out_begin,
                                        int64_t n_out, int32_t* ancestors, void* stream) {
  (void)N_total;
  if (!cum || !base_total_dev || !ancestors || K <= 0 || n_out < 0) return gjx_fail(GJX_EINVAL, "gjx_resample_multinomial: bad argument");
  if (n_out == 0) return GJX_OK;
  hipLaunchKernelGGL(k_multinomial, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cum, K,
                     base_total_dev, key2{key0, key1}, out_begin, n_out, ancestors);
  GJX_CHECK_LAUNCH("gjx_resample_multinomial");
  return GJX_OK;
}

extern "C" int gjx_gather_rows(const float* src, int64_t src_stride, const int32_t* anc, int64_t n_out, int32_t rows,
                               float* dst, int64_t dst_stride, void* stream) {
  if (!src || !anc || !dst || n_out < 0 || rows < 0) return gjx_fail(GJX_EINVAL, "gjx_gather_rows: bad argument");
  if (n_out == 0 || rows == 0) return GJX_OK;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     src_stride, anc, n_out, (int)rows, dst, dst_stride);
  GJX_CHECK_LAUNCH("gjx_gather_rows");
  return GJX_OK;
}

