// gjx_shard.hip — a particle collection split over the GPUs of one node: this rank's part of a global
// systematic resampling (the reference has no multi-device path; this extends smc.py:90-101).
//
//   * plan / planned expansion / kept-children gather / pack / unpack kernels (building blocks, also driven
//     from Python over torch.distributed);
//   * gjx_shard_ctx: the whole exchange as ONE host call over RCCL — two 8-byte all-gathers, the plan, and an
//     all-to-all-v of the surplus children issued as grouped send/recv on the caller's stream, with the local
//     gather overlapped on a second stream.  RCCL is resolved at run time from the library the process already
//     uses (the path is passed in), so there is one RCCL instance per process.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string.h>

#include <chrono>
#include <new>

#include "gjx_device.h"
#include "gjx_host.h"

namespace gjx {

// Sharded resampling plan.  Every rank holds the G per-rank weight totals (one 8-byte all-gather) and derives
// the same slot bounds B_r = J(total_0 + ... + total_{r-1}): rank r's particles produce exactly the output
// slots [B_r, B_{r+1}).  One wave computes the G+1 bounds; the plan goes to device memory (read by the
// expansion / gather kernels queued behind it) and to a pinned host copy (read by the host only to size the
// all-to-all, while those kernels run).
__global__ __launch_bounds__(64) void k_shard_plan(const uint64_t* __restrict__ totals, int G, int rank, double u,
                                                  int64_t N_total, int64_t seq, gjx_shard_plan* plan_dev,
                                                  gjx_shard_plan* plan_host) {
  __shared__ gjx_shard_plan p;
  const int t = threadIdx.x;
  uint64_t total = 0;
  for (int r = 0; r < G; ++r) total += totals[r];
  for (int b = t; b <= G; b += 64) {          // G + 1 bounds, G <= GJX_MAX_RANKS = 64: lane 0 also takes bound 64
    uint64_t below = 0;
    for (int r = 0; r < b; ++r) below += totals[r];
    const double step = (double)total / (double)N_total;
    const double inv_step = (double)N_total / (double)total;
    p.bounds[b] = total > 0 ? slots_below(below, u, step, inv_step, total, N_total) : 0;
    if (b == rank) p.base = below;
  }
  __syncthreads();
  if (t == 0) {
    p.total = total;
    p.slot0 = p.bounds[rank];
    p.n_valid = p.bounds[rank + 1] - p.bounds[rank];
    const int64_t q = N_total / G, rem = N_total % G;
    p.own_lo = rank * q + (rank < rem ? rank : rem);
    p.own_n = q + (rank < rem ? 1 : 0);
    const int64_t lo = p.slot0 > p.own_lo ? p.slot0 : p.own_lo;
    int64_t hi = p.slot0 + p.n_valid < p.own_lo + p.own_n ? p.slot0 + p.n_valid : p.own_lo + p.own_n;
    p.keep_lo = lo;
    p.keep_hi = hi > lo ? hi : lo;
    p.n_ranks = G;
    p.status = total > 0 ? 0 : 1;
    p.seq = seq;
    p.reserved = 0;
  }
  __syncthreads();
  constexpr int kWords = sizeof(gjx_shard_plan) / 8;
  constexpr int kSeqWord = offsetof(gjx_shard_plan, seq) / 8;
  const uint64_t* sp = reinterpret_cast<const uint64_t*>(&p);
  for (int w = t; w < kWords; w += 64) {
    reinterpret_cast<uint64_t*>(plan_dev)[w] = sp[w];
    if (plan_host && w != kSeqWord) reinterpret_cast<uint64_t*>(plan_host)[w] = sp[w];
  }
  if (plan_host) {         // one wave: every lane's stores are ordered before lane 0's release of the sequence word
    __threadfence_system();
    if (t == 0) __hip_atomic_store(&plan_host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// pack: msg[j][r] = src[r][anc[idx(j)]]; unpack: dst[r][col(j)] = msg[j][r]  (lanes along r: messages are row-major)
__global__ __launch_bounds__(256) void k_shard_pack(const float* __restrict__ src, int64_t src_stride, int rows,
                                                   const int32_t* __restrict__ anc, int64_t n_valid, int64_t n_pre,
                                                   int64_t n_suf, float* __restrict__ msg) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (n_pre + n_suf) * rows) return;
  const int64_t j = e / rows, r = e % rows;
  const int32_t a = anc[j < n_pre ? j : n_valid - n_suf + (j - n_pre)];
  msg[e] = src[r * src_stride + a];
}

__global__ __launch_bounds__(256) void k_shard_unpack(const float* __restrict__ msg, int64_t n_lo, int64_t n_hi, int rows,
                                                     float* __restrict__ dst, int64_t dst_stride, int64_t own_n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (n_lo + n_hi) * rows) return;
  const int64_t r = e / (n_lo + n_hi), j = e % (n_lo + n_hi);   // lanes along j: coalesced stores into the SoA rows
  const int64_t col = j < n_lo ? j : own_n - n_hi + (j - n_lo);
  dst[r * dst_stride + col] = msg[j * rows + r];
}

// children that stay on this rank: dst[r][j - own_lo] = src[r][anc[j - slot0]] for the slots j this rank both
// produces and owns, keep_lo <= j < keep_hi (all four read from the device plan)
__global__ __launch_bounds__(256) void k_gather_kept(const gjx_shard_plan* __restrict__ plan, const float* __restrict__ src,
                                                    int64_t src_stride, const int32_t* __restrict__ anc, int rows,
                                                    float* __restrict__ dst, int64_t dst_stride) {
  const int64_t j = plan->keep_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= plan->keep_hi) return;
  const int32_t a = anc[j - plan->slot0];
  const int64_t o = j - plan->own_lo;
  for (int r = 0; r < rows; ++r) dst[(int64_t)r * dst_stride + o] = src[(int64_t)r * src_stride + a];
}

// general strided form: dst[r*drs + j*dcs] = src[r*srs + idx(j)*scs]; packs children into [n][rows] messages
// and unpacks received ones.  Lanes run along r when the destination is row-contiguous.
__global__ __launch_bounds__(256) void k_gather_rows_strided(const float* __restrict__ src, int64_t srs, int64_t scs,
                                                            const int32_t* __restrict__ anc, int64_t n, int rows,
                                                            float* __restrict__ dst, int64_t drs, int64_t dcs) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * rows) return;
  int64_t j, r;
  if (drs == 1) { j = e / rows; r = e % rows; } else { r = e / n; j = e % n; }
  const int64_t a = anc ? anc[j] : j;
  if (a < 0) return;
  dst[r * drs + j * dcs] = src[r * srs + a * scs];
}


// first i in [0, K) with cum[i] > t   (requires t < cum[K-1])
GJX_DEV int64_t upper_search_u64(const uint64_t* __restrict__ cum, int64_t K, uint64_t t) {
  int64_t lo = 0, hi = K - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (cum[mid] > t) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// ---- sharded multinomial resampling --------------------------------------------------------------------------
// Output slot j of N_total draws its own uniform (hash of the GLOBAL slot index, as k_multinomial), so the result
// does not depend on the number of ranks.  The slots that land on a rank's particles are scattered over [0, N_total):
// every rank scans all N_total thresholds (one hash each), keeps those inside its stretch of the global weight line,
// and ships each child to the rank that owns the slot — a true all-to-all (systematic resampling only talks to
// neighbours in slot order).  Two passes over the thresholds: count per owner, then fill the messages; children are
// tagged with their slot, so the order inside a message does not matter and the fill can use atomics.
struct MultiArgs {
  const uint64_t* cum;       // local inclusive prefix sums [K]
  int64_t K;
  const uint64_t* totals;    // [G] all-gathered local totals
  int G, rank;
  key2 key;
  int64_t N_total;
};

GJX_DEV int owner_of(int64_t j, int64_t q, int64_t rem) {   // contiguous shards, remainder to the low ranks
  const int64_t cut = rem * (q + 1);
  return (int)(j < cut ? j / (q + 1) : rem + (j - cut) / (q > 0 ? q : 1));
}
GJX_DEV int64_t shard_lo(int d, int64_t q, int64_t rem) { return d * q + (d < rem ? d : rem); }

GJX_DEV bool multi_threshold(const MultiArgs& a, int64_t j, uint64_t base, uint64_t local, uint64_t total, uint64_t* t_local) {
  const key2 h = fold_in64(a.key, (uint64_t)j);
  const uint64_t r = (((uint64_t)h.a << 32) | h.b) >> 11;
  const double uj = (double)r * (1.0 / 9007199254740992.0);
  const uint64_t T = (uint64_t)(uj * (double)total);
  if (T < base || T >= base + local) return false;
  *t_local = T - base;
  return true;
}

__global__ __launch_bounds__(256) void k_multi_count(MultiArgs a, unsigned long long* counts /*[G], zeroed*/) {
  __shared__ unsigned cnt[GJX_MAX_RANKS];
  for (int t = threadIdx.x; t < a.G; t += 256) cnt[t] = 0u;
  __syncthreads();
  uint64_t total = 0, base = 0;
  for (int r = 0; r < a.G; ++r) { if (r < a.rank) base += a.totals[r]; total += a.totals[r]; }
  const uint64_t local = a.totals[a.rank];
  const int64_t q = a.N_total / a.G, rem = a.N_total % a.G;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < a.N_total; j += (int64_t)gridDim.x * 256) {
    uint64_t tl;
    if (total > 0 && multi_threshold(a, j, base, local, total, &tl)) atomicAdd(&cnt[owner_of(j, q, rem)], 1u);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < a.G; t += 256) if (cnt[t]) atomicAdd(&counts[t], (unsigned long long)cnt[t]);
}

// counts matrix [G][G] (row = source rank) -> pinned host mirror, sequence word last (as the systematic plan)
__global__ __launch_bounds__(64) void k_multi_publish(const unsigned long long* matrix, int n, int64_t seq, unsigned long long* host) {
  for (int t = threadIdx.x; t < n; t += 64) host[1 + t] = matrix[t];
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(&host[0], (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_multi_fill(MultiArgs a, const unsigned long long* my_counts /*[G]: row `rank` of the matrix*/,
                                                   unsigned long long* cursor /*[G], zeroed*/, const float* __restrict__ src,
                                                   int64_t src_stride, int rows, float* __restrict__ msg /*[n][rows+1]*/,
                                                   float* __restrict__ dst, int64_t dst_stride) {
  uint64_t total = 0, base = 0;
  for (int r = 0; r < a.G; ++r) { if (r < a.rank) base += a.totals[r]; total += a.totals[r]; }
  const uint64_t local = a.totals[a.rank];
  const int64_t q = a.N_total / a.G, rem = a.N_total % a.G;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < a.N_total; j += (int64_t)gridDim.x * 256) {
    uint64_t tl;
    if (!(total > 0 && multi_threshold(a, j, base, local, total, &tl))) continue;
    const int64_t anc = upper_search_u64(a.cum, a.K, tl);
    const int d = owner_of(j, q, rem);
    const int64_t slot = j - shard_lo(d, q, rem);
    if (d == a.rank) {
      for (int r = 0; r < rows; ++r) dst[(int64_t)r * dst_stride + slot] = src[(int64_t)r * src_stride + anc];
    } else {
      uint64_t off = 0;
      for (int e = 0; e < d; ++e) if (e != a.rank) off += my_counts[e];
      const uint64_t pos = off + atomicAdd(&cursor[d], 1ull);
      float* m = msg + pos * (uint64_t)(rows + 1);
      for (int r = 0; r < rows; ++r) m[r] = src[(int64_t)r * src_stride + anc];
      m[rows] = __uint_as_float((uint32_t)slot);
    }
  }
}

__global__ __launch_bounds__(256) void k_multi_unpack(const float* __restrict__ msg, int64_t n, int rows, float* __restrict__ dst,
                                                     int64_t dst_stride) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float* m = msg + e * (int64_t)(rows + 1);
  const int64_t slot = (int64_t)__float_as_uint(m[rows]);
  for (int r = 0; r < rows; ++r) dst[(int64_t)r * dst_stride + slot] = m[r];
}

}  // namespace gjx

using namespace gjx;

extern "C" int gjx_shard_plan_build(const uint64_t* totals_dev, int32_t n_ranks, int32_t rank, double u, int64_t N_total,
                                    int64_t seq, gjx_shard_plan* plan_dev, gjx_shard_plan* plan_host_pinned, void* stream) {
  if (!totals_dev || !plan_dev || n_ranks < 1 || n_ranks > GJX_MAX_RANKS || rank < 0 || rank >= n_ranks || N_total <= 0 ||
      !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_shard_plan_build: bad argument");
  gjx_shard_plan* mapped = nullptr;
  if (plan_host_pinned && hipHostGetDevicePointer((void**)&mapped, plan_host_pinned, 0) != hipSuccess) {
    (void)hipGetLastError();
    return gjx_fail(GJX_EINVAL, "gjx_shard_plan_build: plan_host_pinned is not pinned (device-mapped) host memory");
  }
  hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, totals_dev, (int)n_ranks, (int)rank, u, N_total,
                     seq, plan_dev, mapped);
  GJX_CHECK_LAUNCH("gjx_shard_plan_build");
  return GJX_OK;
}

extern "C" int gjx_shard_resample(const uint64_t* cum, int64_t K, const gjx_shard_plan* plan_dev, double u, int64_t N_total,
                                  int32_t* ancestors, int64_t anc_capacity, const float* src, int64_t src_stride,
                                  int32_t rows, float* dst, int64_t dst_stride, int64_t own_n, void* stream) {
  if (!cum || !plan_dev || !ancestors || K <= 0 || N_total <= 0 || anc_capacity < 0 || rows < 0 || own_n < 0 ||
      (rows > 0 && own_n > 0 && (!src || !dst)) || !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_shard_resample: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_expand_planned(cum, K, plan_dev, u, N_total, ancestors, anc_capacity, st);
  if (rc != GJX_OK) return rc;
  if (rows > 0 && own_n > 0) {
    hipLaunchKernelGGL(k_gather_kept, dim3((unsigned)((own_n + 255) / 256)), dim3(256), 0, st, plan_dev, src, src_stride,
                       (const int32_t*)ancestors, (int)rows, dst, dst_stride);
    GJX_CHECK_LAUNCH("gjx_shard_resample(gather)");
  }
  return GJX_OK;
}

extern "C" int gjx_gather_rows_strided(const float* src, int64_t src_row_stride, int64_t src_col_stride, const int32_t* anc,
                                       int64_t n, int32_t rows, float* dst, int64_t dst_row_stride, int64_t dst_col_stride,
                                       void* stream) {
  if (!src || !dst || n < 0 || rows < 0) return gjx_fail(GJX_EINVAL, "gjx_gather_rows_strided: bad argument");
  if (n == 0 || rows == 0) return GJX_OK;
  hipLaunchKernelGGL(k_gather_rows_strided, dim3((unsigned)((n * rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     src_row_stride, src_col_stride, anc, n, (int)rows, dst, dst_row_stride, dst_col_stride);
  GJX_CHECK_LAUNCH("gjx_gather_rows_strided");
  return GJX_OK;
}

extern "C" int gjx_shard_pack(const float* src, int64_t src_stride, int32_t rows, const int32_t* ancestors, int64_t n_valid,
                              int64_t n_pre, int64_t n_suf, float* msg, void* stream) {
  if (n_pre < 0 || n_suf < 0 || rows < 0 || n_pre + n_suf > n_valid) return gjx_fail(GJX_EINVAL, "gjx_shard_pack: bad argument");
  const int64_t n = (n_pre + n_suf) * rows;
  if (n == 0) return GJX_OK;
  if (!src || !ancestors || !msg) return gjx_fail(GJX_EINVAL, "gjx_shard_pack: bad argument");
  hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, src_stride, (int)rows,
                     ancestors, n_valid, n_pre, n_suf, msg);
  GJX_CHECK_LAUNCH("gjx_shard_pack");
  return GJX_OK;
}

extern "C" int gjx_shard_unpack(const float* msg, int64_t n_lo, int64_t n_hi, int32_t rows, float* dst, int64_t dst_stride,
                                int64_t own_n, void* stream) {
  if (n_lo < 0 || n_hi < 0 || rows < 0 || n_lo + n_hi > own_n) return gjx_fail(GJX_EINVAL, "gjx_shard_unpack: bad argument");
  const int64_t n = (n_lo + n_hi) * rows;
  if (n == 0) return GJX_OK;
  if (!msg || !dst) return gjx_fail(GJX_EINVAL, "gjx_shard_unpack: bad argument");
  hipLaunchKernelGGL(k_shard_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, msg, n_lo, n_hi, (int)rows,
                     dst, dst_stride, own_n);
  GJX_CHECK_LAUNCH("gjx_shard_unpack");
  return GJX_OK;
}

// ------------------------------------------------------------------------------------------------------------
// RCCL-driven exchange
// ------------------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int rccl_open(const char* path, RcclApi* api) {
  if (!path) return gjx_fail(GJX_EINVAL, "rccl: library path is NULL");
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return gjx_fail(GJX_EUNSUPPORTED, dlerror());
  api->lib = h;
  bool ok = true;
  auto sym = [&](const char* name) { void* p = dlsym(h, name); ok = ok && p; return p; };
  api->GetUniqueId = (decltype(api->GetUniqueId))sym("ncclGetUniqueId");
  api->CommInitRank = (decltype(api->CommInitRank))sym("ncclCommInitRank");
  api->CommDestroy = (decltype(api->CommDestroy))sym("ncclCommDestroy");
  api->AllGather = (decltype(api->AllGather))sym("ncclAllGather");
  api->Send = (decltype(api->Send))sym("ncclSend");
  api->Recv = (decltype(api->Recv))sym("ncclRecv");
  api->GroupStart = (decltype(api->GroupStart))sym("ncclGroupStart");
  api->GroupEnd = (decltype(api->GroupEnd))sym("ncclGroupEnd");
  api->GetErrorString = (decltype(api->GetErrorString))sym("ncclGetErrorString");
  if (!ok) return gjx_fail(GJX_EUNSUPPORTED, "rccl: the library does not export the nccl* entry points");
  return GJX_OK;
}

}  // namespace

struct gjx_shard_ctx {
  RcclApi api;
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, rows = 0;
  int64_t K = 0, N_total = 0, own_lo = 0, own_n = 0, seq = 0;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  uint64_t* cum = nullptr;
  uint64_t* bt = nullptr;        // {0, local total}
  float* pairs = nullptr;        // [world][2]
  uint64_t* totals = nullptr;    // [world]
  gjx_shard_plan* plan_dev = nullptr;
  gjx_shard_plan* plan_host = nullptr;   // pinned
  gjx_shard_plan* plan_host_mapped = nullptr;
  int32_t* anc = nullptr;        // [N_total]
  float* send = nullptr;
  float* recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;     // in floats
  int64_t stat_steps = 0, stat_sent = 0, stat_received = 0;   // since creation (gjx_shard_ctx_stats)
  unsigned long long* mcounts = nullptr;   // [G] this rank's children per owner + [G] fill cursors
  unsigned long long* mmatrix = nullptr;   // [G][G] all-gathered counts (row = source)
  unsigned long long* mhost = nullptr;     // pinned: {seq, matrix[G*G]}
  unsigned long long* mhost_mapped = nullptr;
};

#define GJX_HIP(call, where)                                   \
  do {                                                         \
    hipError_t e__ = (call);                                   \
    if (e__ != hipSuccess) return gjx_fail_hip(e__, where);    \
  } while (0)
#define GJX_NCCL(c, call, where)                                                    \
  do {                                                                              \
    ncclResult_t r__ = (call);                                                      \
    if (r__ != ncclSuccess) return gjx_fail(GJX_EHIP, (c)->api.GetErrorString(r__)); \
  } while (0)

extern "C" int gjx_rccl_unique_id(const char* rccl_library_path, uint8_t* out128) {
  if (!out128) return gjx_fail(GJX_EINVAL, "gjx_rccl_unique_id: bad argument");
  RcclApi api;
  int rc = rccl_open(rccl_library_path, &api);
  if (rc != GJX_OK) return rc;
  ncclUniqueId id;
  ncclResult_t r = api.GetUniqueId(&id);
  if (r != ncclSuccess) return gjx_fail(GJX_EHIP, api.GetErrorString(r));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return GJX_OK;
}

extern "C" int gjx_shard_ctx_destroy(gjx_shard_ctx* c) {
  if (!c) return GJX_OK;
  (void)hipDeviceSynchronize();
  if (c->comm) c->api.CommDestroy(c->comm);
  void* dev_bufs[] = {c->ws, c->cum, c->bt, c->pairs, c->totals, c->plan_dev, c->anc, c->send, c->recv, c->mcounts, c->mmatrix};
  for (void* b : dev_bufs)
    if (b) (void)hipFree(b);
  if (c->plan_host) (void)hipHostFree(c->plan_host);
  if (c->mhost) (void)hipHostFree(c->mhost);
  delete c;
  return GJX_OK;
}

extern "C" int gjx_shard_ctx_create(const char* rccl_library_path, const uint8_t* unique_id128, int32_t n_ranks, int32_t rank,
                                    int64_t K_local, int32_t rows, int64_t N_total, gjx_shard_ctx** out) {
  if (!unique_id128 || !out || n_ranks < 1 || n_ranks > GJX_MAX_RANKS || rank < 0 || rank >= n_ranks || K_local <= 0 || rows < 0 ||
      N_total <= 0)
    return gjx_fail(GJX_EINVAL, "gjx_shard_ctx_create: bad argument");
  gjx_shard_ctx* c = new (std::nothrow) gjx_shard_ctx();
  if (!c) return gjx_fail(GJX_EINVAL, "gjx_shard_ctx_create: out of host memory");
  int rc = rccl_open(rccl_library_path, &c->api);
  if (rc != GJX_OK) { delete c; return rc; }
  c->world = n_ranks; c->rank = rank; c->rows = rows; c->K = K_local; c->N_total = N_total;
  const int64_t q = N_total / n_ranks, rem = N_total % n_ranks;
  c->own_lo = rank * q + (rank < rem ? rank : rem);
  c->own_n = q + (rank < rem ? 1 : 0);
  c->ws_bytes = gjx_workspace_bytes(GJX_OP_RESAMPLE, K_local);
  auto fail = [&](int code) { gjx_shard_ctx_destroy(c); return code; };
#define GJX_TRY(call, where)                                          \
  do {                                                                \
    hipError_t e__ = (call);                                          \
    if (e__ != hipSuccess) return fail(gjx_fail_hip(e__, where));     \
  } while (0)
  GJX_TRY(hipMalloc(&c->ws, c->ws_bytes), "shard ctx: workspace");
  GJX_TRY(hipMemset(c->ws, 0, c->ws_bytes), "shard ctx: workspace");
  GJX_TRY(hipMalloc((void**)&c->cum, sizeof(uint64_t) * K_local), "shard ctx: cum");
  GJX_TRY(hipMalloc((void**)&c->bt, sizeof(uint64_t) * 2), "shard ctx: bt");
  GJX_TRY(hipMalloc((void**)&c->pairs, sizeof(float) * 2 * n_ranks), "shard ctx: pairs");
  GJX_TRY(hipMalloc((void**)&c->totals, sizeof(uint64_t) * n_ranks), "shard ctx: totals");
  GJX_TRY(hipMalloc((void**)&c->plan_dev, sizeof(gjx_shard_plan)), "shard ctx: plan");
  GJX_TRY(hipMalloc((void**)&c->anc, sizeof(int32_t) * N_total), "shard ctx: ancestors");
  GJX_TRY(hipHostMalloc((void**)&c->plan_host, sizeof(gjx_shard_plan), hipHostMallocMapped), "shard ctx: pinned plan");
  memset(c->plan_host, 0, sizeof(gjx_shard_plan));
  GJX_TRY(hipHostGetDevicePointer((void**)&c->plan_host_mapped, c->plan_host, 0), "shard ctx: pinned plan");
  GJX_TRY(hipMalloc((void**)&c->mcounts, sizeof(unsigned long long) * 2 * n_ranks), "shard ctx: multinomial counts");
  GJX_TRY(hipMalloc((void**)&c->mmatrix, sizeof(unsigned long long) * n_ranks * n_ranks), "shard ctx: multinomial matrix");
  GJX_TRY(hipHostMalloc((void**)&c->mhost, sizeof(unsigned long long) * (1 + n_ranks * n_ranks), hipHostMallocMapped), "shard ctx: pinned counts");
  memset(c->mhost, 0, sizeof(unsigned long long) * (1 + n_ranks * n_ranks));
  GJX_TRY(hipHostGetDevicePointer((void**)&c->mhost_mapped, c->mhost, 0), "shard ctx: pinned counts");
  if (rows > 0 && n_ranks > 1) {
    // message buffers for the worst case (every child this rank produces leaves it / every slot it owns is filled
    // from elsewhere), capped at 1 GiB each: nothing is allocated or freed inside the resampling loop below that
    const size_t cap = (size_t)1 << 28;
    // (rows + 1 floats per child: the multinomial exchange tags each child with its slot)
    size_t s_need = (size_t)(N_total - c->own_n) * (rows + 1), r_need = (size_t)c->own_n * (rows + 1);
    s_need = s_need < cap ? s_need : cap; r_need = r_need < cap ? r_need : cap;
    if (s_need) { GJX_TRY(hipMalloc((void**)&c->send, s_need * sizeof(float)), "shard ctx: send buffer"); c->send_cap = s_need; }
    if (r_need) { GJX_TRY(hipMalloc((void**)&c->recv, r_need * sizeof(float)), "shard ctx: recv buffer"); c->recv_cap = r_need; }
  }
  GJX_TRY(hipDeviceSynchronize(), "shard ctx: init");
#undef GJX_TRY
  ncclUniqueId id;
  memcpy(&id, unique_id128, 128);
  ncclResult_t r = c->api.CommInitRank(&c->comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return fail(gjx_fail(GJX_EHIP, c->api.GetErrorString(r)));
  }
  *out = c;
  return GJX_OK;
}

static int grow(float** buf, size_t* cap, size_t need, hipStream_t st) {
  if (need <= *cap) return GJX_OK;
  GJX_HIP(hipStreamSynchronize(st), "shard step: grow");
  if (*buf) GJX_HIP(hipFree(*buf), "shard step: grow");
  *buf = nullptr; *cap = 0;
  size_t want = need + need / 4 + 4096;
  GJX_HIP(hipMalloc((void**)buf, want * sizeof(float)), "shard step: message buffer");
  *cap = want;
  return GJX_OK;
}

// Message sizes of one rank, from the plan alone (host arithmetic, no device access).  Rank r produces the output
// slots [bounds[r], bounds[r+1]) and stores the slots of its own shard of N_total; what it produces outside its
// shard goes to the owners (send_counts), what others produce inside it comes in (recv_counts).  Every rank calls
// this with the same bounds, so send_counts[d] on rank r equals recv_counts[r] on rank d by construction.
extern "C" int gjx_shard_message_counts(const gjx_shard_plan* plan, int32_t rank, int64_t N_total, int64_t* send_counts,
                                        int64_t* recv_counts, int64_t* parts4) {
  if (!plan || !send_counts || !recv_counts || plan->n_ranks < 1 || plan->n_ranks > GJX_MAX_RANKS || rank < 0 ||
      rank >= plan->n_ranks || N_total <= 0)
    return gjx_fail(GJX_EINVAL, "gjx_shard_message_counts: bad argument");
  const int G = (int)plan->n_ranks;
  const int64_t q = N_total / G, rem = N_total % G;
  auto lo_of = [&](int d) { return d * q + (d < rem ? d : rem); };
  const int64_t slot0 = plan->bounds[rank], run_hi = plan->bounds[rank + 1];
  const int64_t own_lo = lo_of(rank), own_hi = lo_of(rank + 1);
  auto overlap = [](int64_t a0, int64_t a1, int64_t b0, int64_t b1) {
    const int64_t lo = a0 > b0 ? a0 : b0, hi = a1 < b1 ? a1 : b1;
    return hi > lo ? hi - lo : (int64_t)0;
  };
  int64_t n_pre = 0, n_suf = 0, n_lo = 0, n_hi = 0;
  for (int d = 0; d < G; ++d) {
    send_counts[d] = d == rank ? 0 : overlap(slot0, run_hi, lo_of(d), lo_of(d + 1));
    recv_counts[d] = d == rank ? 0 : overlap(plan->bounds[d], plan->bounds[d + 1], own_lo, own_hi);
    (d < rank ? n_pre : n_suf) += send_counts[d];
    (d < rank ? n_lo : n_hi) += recv_counts[d];
  }
  // the two pieces of the run outside the shard / of the shard outside the run, as intervals (cross-check)
  auto clamp0 = [](int64_t v) { return v > 0 ? v : (int64_t)0; };
  if (n_pre != clamp0((run_hi < own_lo ? run_hi : own_lo) - slot0) || n_suf != clamp0(run_hi - (slot0 > own_hi ? slot0 : own_hi)) ||
      n_lo != clamp0((own_hi < slot0 ? own_hi : slot0) - own_lo) || n_hi != clamp0(own_hi - (own_lo > run_hi ? own_lo : run_hi)))
    return gjx_fail(GJX_EINVAL, "gjx_shard_message_counts: the plan's bounds are not monotone");
  if (parts4) { parts4[0] = n_pre; parts4[1] = n_suf; parts4[2] = n_lo; parts4[3] = n_hi; }
  return GJX_OK;
}

extern "C" int gjx_shard_resample_step(gjx_shard_ctx* c, const float* logw, const float* local_lse, const float* rows_in,
                                       int64_t in_stride, float* rows_out, int64_t out_stride, double u, float* lse_out,
                                       int64_t* info_host, void* stream) {
  if (!c || !logw || !local_lse || !lse_out || (c->rows > 0 && (!rows_in || !rows_out)) || !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_shard_resample_step: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int G = c->world, rank = c->rank, R = c->rows;
  // global {max, sumexp}: one pair per rank; the prefix-sum kernels reduce the pairs in their prologue
  GJX_NCCL(c, c->api.AllGather(local_lse, c->pairs, 2, ncclFloat, c->comm, st), "all-gather lse pairs");
  int rc = gjx_weight_cumsum(logw, c->K, 2, c->pairs, G, c->cum, c->bt, lse_out, c->N_total, c->ws, c->ws_bytes, st);
  if (rc != GJX_OK) return rc;
  GJX_NCCL(c, c->api.AllGather(c->bt + 1, c->totals, 1, ncclUint64, c->comm, st), "all-gather totals");
  const int64_t seq = ++c->seq;
  hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(64), 0, st, (const uint64_t*)c->totals, G, rank, u, c->N_total, seq, c->plan_dev,
                     c->plan_host_mapped);
  GJX_CHECK_LAUNCH("gjx_shard_resample_step(plan)");
  rc = launch_expand_planned(c->cum, c->K, c->plan_dev, u, c->N_total, c->anc, c->N_total, st);
  if (rc != GJX_OK) return rc;
  // children that stay are gathered in place while the host reads the plan.  (A second stream for this gather,
  // overlapped with the send/recv, was measured: the two cross-queue dependencies cost ~25 us, the gather 24 us.)
  if (R > 0 && c->own_n > 0) {
    hipLaunchKernelGGL(k_gather_kept, dim3((unsigned)((c->own_n + 255) / 256)), dim3(256), 0, st, (const gjx_shard_plan*)c->plan_dev,
                       rows_in, in_stride, (const int32_t*)c->anc, R, rows_out, out_stride);
    GJX_CHECK_LAUNCH("gjx_shard_resample_step(gather)");
  }
  // the host needs only the G+1 slot bounds to size the messages: spin on the pinned mirror of the plan
  {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (__atomic_load_n(&c->plan_host->seq, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
        return gjx_fail(GJX_EHIP, "gjx_shard_resample_step: the plan kernel did not complete within 60 s");
    }
  }
  const gjx_shard_plan* p = c->plan_host;
  if (p->status) return gjx_fail(GJX_EINVAL, "gjx_shard_resample_step: all weights are zero");
  int64_t send_counts[GJX_MAX_RANKS], recv_counts[GJX_MAX_RANKS], parts[4];
  rc = gjx_shard_message_counts(p, rank, c->N_total, send_counts, recv_counts, parts);
  if (rc != GJX_OK) return rc;
  const int64_t n_pre = parts[0], n_suf = parts[1], n_lo = parts[2], n_hi = parts[3];
  c->stat_steps += 1; c->stat_sent += n_pre + n_suf; c->stat_received += n_lo + n_hi;
  if (info_host) {
    info_host[0] = n_pre + n_suf;  // children sent
    info_host[1] = n_lo + n_hi;    // children received
    info_host[2] = p->slot0;
    info_host[3] = p->n_valid;
  }
  if (R > 0 && G > 1) {
    rc = grow(&c->send, &c->send_cap, (size_t)(n_pre + n_suf) * R, st);
    if (rc != GJX_OK) return rc;
    rc = grow(&c->recv, &c->recv_cap, (size_t)(n_lo + n_hi) * R, st);
    if (rc != GJX_OK) return rc;
    if (n_pre + n_suf) {
      const int64_t n = (n_pre + n_suf) * R;
      hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rows_in, in_stride, R, (const int32_t*)c->anc,
                         p->n_valid, n_pre, n_suf, c->send);
      GJX_CHECK_LAUNCH("gjx_shard_resample_step(pack)");
    }
    // all-to-all-v as grouped send/recv.  Messages are in slot order, so the block for rank d follows the blocks
    // of the lower ranks; every rank derives every count from the same bounds (gjx_shard_message_counts).
    GJX_NCCL(c, c->api.GroupStart(), "group");
    int64_t s_off = 0, r_off = 0;
    for (int d = 0; d < G; ++d) {
      if (send_counts[d]) {
        GJX_NCCL(c, c->api.Send(c->send + s_off * R, (size_t)send_counts[d] * R, ncclFloat, d, c->comm, st), "send");
        s_off += send_counts[d];
      }
      if (recv_counts[d]) {
        GJX_NCCL(c, c->api.Recv(c->recv + r_off * R, (size_t)recv_counts[d] * R, ncclFloat, d, c->comm, st), "recv");
        r_off += recv_counts[d];
      }
    }
    GJX_NCCL(c, c->api.GroupEnd(), "group");
    if (n_lo + n_hi) {
      const int64_t n = (n_lo + n_hi) * R;
      hipLaunchKernelGGL(k_shard_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)c->recv, n_lo, n_hi, R, rows_out,
                         out_stride, c->own_n);
      GJX_CHECK_LAUNCH("gjx_shard_resample_step(unpack)");
    }
  }
  return GJX_OK;
}


extern "C" int gjx_shard_resample_multinomial_step(gjx_shard_ctx* c, const float* logw, const float* local_lse, const float* rows_in,
                                                   int64_t in_stride, float* rows_out, int64_t out_stride, uint32_t key0, uint32_t key1,
                                                   float* lse_out, int64_t* info_host, void* stream) {
  if (!c || !logw || !local_lse || !lse_out || (c->rows > 0 && (!rows_in || !rows_out)))
    return gjx_fail(GJX_EINVAL, "gjx_shard_resample_multinomial_step: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int G = c->world, rank = c->rank, R = c->rows;
  GJX_NCCL(c, c->api.AllGather(local_lse, c->pairs, 2, ncclFloat, c->comm, st), "all-gather lse pairs");
  int rc = gjx_weight_cumsum(logw, c->K, 2, c->pairs, G, c->cum, c->bt, lse_out, c->N_total, c->ws, c->ws_bytes, st);
  if (rc != GJX_OK) return rc;
  GJX_NCCL(c, c->api.AllGather(c->bt + 1, c->totals, 1, ncclUint64, c->comm, st), "all-gather totals");
  GJX_HIP(hipMemsetAsync(c->mcounts, 0, sizeof(unsigned long long) * 2 * G, st), "multinomial step: counters");
  MultiArgs a;
  a.cum = c->cum; a.K = c->K; a.totals = c->totals; a.G = G; a.rank = rank; a.key = key2{key0, key1}; a.N_total = c->N_total;
  const int64_t want = (c->N_total + 255) / 256;
  const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(k_multi_count, dim3(grid), dim3(256), 0, st, a, c->mcounts);
  GJX_CHECK_LAUNCH("gjx_shard_resample_multinomial_step(count)");
  GJX_NCCL(c, c->api.AllGather(c->mcounts, c->mmatrix, (size_t)G, ncclUint64, c->comm, st), "all-gather counts");
  const int64_t seq = ++c->seq;
  hipLaunchKernelGGL(k_multi_publish, dim3(1), dim3(64), 0, st, (const unsigned long long*)c->mmatrix, G * G, seq, c->mhost_mapped);
  GJX_CHECK_LAUNCH("gjx_shard_resample_multinomial_step(publish)");
  {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (__atomic_load_n(&c->mhost[0], __ATOMIC_ACQUIRE) != (unsigned long long)seq) {
      if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
        return gjx_fail(GJX_EHIP, "gjx_shard_resample_multinomial_step: the count kernels did not complete within 60 s");
    }
  }
  const unsigned long long* M = c->mhost + 1;
  int64_t n_send = 0, n_recv = 0, n_all = 0;
  for (int s = 0; s < G; ++s)
    for (int d = 0; d < G; ++d) n_all += (int64_t)M[s * G + d];
  if (n_all != c->N_total) return gjx_fail(GJX_EINVAL, n_all == 0 ? "gjx_shard_resample_multinomial_step: all weights are zero"
                                                                  : "gjx_shard_resample_multinomial_step: the ranks' counts do not add up to N_total");
  for (int d = 0; d < G; ++d) if (d != rank) { n_send += (int64_t)M[rank * G + d]; n_recv += (int64_t)M[d * G + rank]; }
  c->stat_steps += 1; c->stat_sent += n_send; c->stat_received += n_recv;
  if (info_host) { info_host[0] = n_send; info_host[1] = n_recv; info_host[2] = (int64_t)M[rank * G + rank]; info_host[3] = c->N_total; }
  if (R > 0) {
    rc = grow(&c->send, &c->send_cap, (size_t)n_send * (R + 1), st);
    if (rc != GJX_OK) return rc;
    rc = grow(&c->recv, &c->recv_cap, (size_t)n_recv * (R + 1), st);
    if (rc != GJX_OK) return rc;
    hipLaunchKernelGGL(k_multi_fill, dim3(grid), dim3(256), 0, st, a, (const unsigned long long*)(c->mmatrix + (size_t)rank * G),
                       c->mcounts + G, rows_in, in_stride, R, c->send, rows_out, out_stride);
    GJX_CHECK_LAUNCH("gjx_shard_resample_multinomial_step(fill)");
    if (G > 1) {
      GJX_NCCL(c, c->api.GroupStart(), "group");
      int64_t s_off = 0, r_off = 0;
      for (int d = 0; d < G; ++d) {
        if (d == rank) continue;
        const int64_t ns = (int64_t)M[rank * G + d], nr = (int64_t)M[d * G + rank];
        if (ns) { GJX_NCCL(c, c->api.Send(c->send + s_off * (R + 1), (size_t)ns * (R + 1), ncclFloat, d, c->comm, st), "send"); s_off += ns; }
        if (nr) { GJX_NCCL(c, c->api.Recv(c->recv + r_off * (R + 1), (size_t)nr * (R + 1), ncclFloat, d, c->comm, st), "recv"); r_off += nr; }
      }
      GJX_NCCL(c, c->api.GroupEnd(), "group");
      if (n_recv) {
        hipLaunchKernelGGL(k_multi_unpack, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, st, (const float*)c->recv, n_recv, R,
                           rows_out, out_stride);
        GJX_CHECK_LAUNCH("gjx_shard_resample_multinomial_step(unpack)");
      }
    }
  }
  return GJX_OK;
}

extern "C" int gjx_shard_ctx_stats(const gjx_shard_ctx* c, int64_t* out4) {
  if (!c || !out4) return gjx_fail(GJX_EINVAL, "gjx_shard_ctx_stats: bad argument");
  out4[0] = c->stat_steps; out4[1] = c->stat_sent; out4[2] = c->stat_received; out4[3] = c->world;
  return GJX_OK;
}

extern "C" int gjx_shard_ctx_shape(const gjx_shard_ctx* c, int64_t* out5) {
  if (!c || !out5) return gjx_fail(GJX_EINVAL, "gjx_shard_ctx_shape: bad argument");
  out5[0] = c->K; out5[1] = c->rows; out5[2] = c->N_total; out5[3] = c->world; out5[4] = c->rank;
  return GJX_OK;
}

extern "C" int gjx_shard_global_lse(gjx_shard_ctx* c, const float* local_lse, float* lse_out, void* stream) {
  if (!c || !local_lse || !lse_out) return gjx_fail(GJX_EINVAL, "gjx_shard_global_lse: bad argument");
  GJX_NCCL(c, c->api.AllGather(local_lse, c->pairs, 2, ncclFloat, c->comm, (hipStream_t)stream), "all-gather lse pairs");
  return gjx_lse_combine(c->pairs, c->world, c->N_total, lse_out, stream);
}
