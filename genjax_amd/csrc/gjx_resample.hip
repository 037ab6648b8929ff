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

// ---- fixed-point weights ---------------------------------------------------------------------
constexpr float kWeightScale = 1073741824.0f;  // 2^30
constexpr int kScanItems = 8;                  // items per thread
constexpr int kScanTile = 256 * kScanItems;    // items per block

GJX_DEV uint64_t weight_q(const float* x, int64_t i, int is_log, float mx) {
  float w = is_log ? fast_exp(x[i] - mx) : x[i];
  w = w > 0.0f ? w : 0.0f;
  return (uint64_t)(w * kWeightScale);
}

GJX_DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((unsigned long long)v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_wsum_blocks(const float* x, int64_t K, int is_log, const float* lse,
                                                    uint64_t* block_sums) {
  __shared__ uint64_t red[4];
  const float mx = is_log ? lse[0] : 0.0f;
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

// exclusive scan of the block sums in place (single block, sequential carry over chunks of 256)
__global__ __launch_bounds__(256) void k_scan_block_sums(uint64_t* block_sums, int n, uint64_t* total) {
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int t = c0 + threadIdx.x;
    const uint64_t v = t < n ? block_sums[t] : 0;
    uint64_t inc = v;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t up = __shfl_up((unsigned long long)inc, o, 64);
      if ((threadIdx.x & 63) >= o) inc += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    const uint64_t carry = carry_s;
    if (t < n) block_sums[t] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(256) void k_wscan_write(const float* x, int64_t K, int is_log, const float* lse,
                                                    const uint64_t* block_offsets, uint64_t* cum) {
  __shared__ uint64_t wsum[4];
  const float mx = is_log ? lse[0] : 0.0f;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint64_t q[kScanItems];
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    q[k] = (base + k < K) ? weight_q(x, base + k, is_log, mx) : 0;
    s += q[k];
    q[k] = s;  // thread-local inclusive
  }
  uint64_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t up = __shfl_up((unsigned long long)inc, o, 64);
    if ((threadIdx.x & 63) >= o) inc += up;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint64_t off = block_offsets[blockIdx.x] + inc - s;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < K) cum[base + k] = off + q[k];
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

__global__ __launch_bounds__(256) void k_systematic(const uint64_t* cum, int64_t K, const uint64_t* base_total,
                                                   double u, int64_t N_total, int64_t out_begin, int64_t n_out,
                                                   int32_t* ancestors) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const uint64_t base = base_total[0], total = base_total[1];
  const double step = (double)total / (double)N_total;
  const double pj = ((double)(out_begin + j) + u) * step;
  uint64_t T = (uint64_t)pj;
  if (total > 0 && T > total - 1) T = total - 1;
  int32_t a = -1;
  if (K > 0) {
    const uint64_t local = cum[K - 1];
    if (T >= base && T < base + local) a = (int32_t)upper_search(cum, K, T - base);
  }
  ancestors[j] = a;
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
  if (a < 0) return;
  for (int r = 0; r < rows; ++r) dst[(int64_t)r * dst_stride + j] = src[(int64_t)r * src_stride + a];
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

extern "C" int gjx_weight_cumsum(const float* x, int64_t K, int32_t is_log, const float* lse, uint64_t* cum,
                                 uint64_t* total_dev, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !cum || !total_dev || K <= 0 || (is_log && !lse)) return gjx_fail(GJX_EINVAL, "gjx_weight_cumsum: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_weight_cumsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblocks = (int)((K + kScanTile - 1) / kScanTile);
  uint64_t* bs = (uint64_t*)((char*)workspace + kWsHeaderBytes);
  hipLaunchKernelGGL(k_wsum_blocks, dim3(nblocks), dim3(256), 0, st, x, K, (int)is_log, lse, bs);
  GJX_CHECK_LAUNCH("gjx_weight_cumsum/sum");
  hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(256), 0, st, bs, nblocks, total_dev);
  GJX_CHECK_LAUNCH("gjx_weight_cumsum/scan");
  hipLaunchKernelGGL(k_wscan_write, dim3(nblocks), dim3(256), 0, st, x, K, (int)is_log, lse, (const uint64_t*)bs, cum);
  GJX_CHECK_LAUNCH("gjx_weight_cumsum/write");
  return GJX_OK;
}

extern "C" int gjx_resample_systematic(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev, double u,
                                       int64_t N_total, int64_t out_begin, int64_t n_out, int32_t* ancestors,
                                       void* stream) {
  if (!cum || !base_total_dev || !ancestors || K <= 0 || N_total <= 0 || n_out < 0 || !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_resample_systematic: bad argument");
  if (n_out == 0) return GJX_OK;
  hipLaunchKernelGGL(k_systematic, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cum, K,
                     base_total_dev, u, N_total, out_begin, n_out, ancestors);
  GJX_CHECK_LAUNCH("gjx_resample_systematic");
  return GJX_OK;
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
