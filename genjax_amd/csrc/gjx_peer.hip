// gjx_peer.hip — a particle collection sharded over the GPUs of one node with NO host in the loop: every rank maps
// every other rank's exchange windows (hipIpc; xGMI peer access between GPUs) and the kernels talk through them —
// tagged granules pushed with system-scope stores, source tiles and ancestors' rows pulled with system-scope loads.
// The reference has no multi-device path; this is the build's config-4 design.
//
//   gjx_peer_ctx            two windows per rank: DATA (rows x2, log-weights x2: what other ranks read) and FLAG
//                           (granules, LSE ring, `ready` words: what other ranks write), exported as IPC handles
//   gjx_ssm_filter_peer     the whole sharded bootstrap filter: step 0 + k_pf_persistent (gjx_pfilter.inl), two
//                           launches per run on every rank whatever T is
//   gjx_peer_resample_gather  one sharded ImportanceK resampling step in ONE launch (k_peer_resample_gather below)
//
// gjx_shard.hip (RCCL collectives + a host-sized all-to-all-v) stays as the fallback transport.
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "gjx_device.h"
#include "gjx_host.h"
#include "gjx_pfilter_host.h"
#include "gjx_scan.h"
#include "gjx_tile.h"

using namespace gjx;

namespace {
constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }
}  // namespace

struct gjx_peer_ctx {
  int world = 0, rank = 0, rows = 0, share = 1;
  int64_t K = 0;                       // particles per rank
  int nt = 0, NT = 0;                  // quantisation tiles per rank / in total
  char* data = nullptr;                // this rank's DATA window
  char* flag = nullptr;                // this rank's FLAG window
  size_t data_bytes = 0, flag_bytes = 0;
  size_t off_rows[2] = {0, 0}, off_lw[2] = {0, 0};
  size_t off_region[2] = {0, 0}, region_bytes = 0;
  // inside a flag region
  size_t r_aggA = 0, r_aggB = 0, r_bsum = 0, r_bmax = 0, r_ready = 0, r_gmm = 0;
  char* peer_data[GJX_MAX_RANKS];
  char* peer_flag[GJX_MAX_RANKS];
  long long* delta_dev = nullptr;      // [2][world]: byte distance to rank g's data window, then to its flag window
  bool connected = false;
  uint64_t n_filter = 0, n_gmm = 0;    // launches so far (select the flag region / the tags)
  double* us_dev = nullptr;
  uint32_t* keys_dev = nullptr;
  int t_cap = 0;
};

#define GJX_HIP(call, where)                                   \
  do {                                                         \
    hipError_t e__ = (call);                                   \
    if (e__ != hipSuccess) return gjx_fail_hip(e__, where);    \
  } while (0)

extern "C" int gjx_peer_ctx_destroy(gjx_peer_ctx* c) {
  if (!c) return GJX_OK;
  (void)hipDeviceSynchronize();
  if (c->connected) {
    for (int g = 0; g < c->world; ++g) {
      if (g == c->rank) continue;
      if (c->peer_data[g]) (void)hipIpcCloseMemHandle(c->peer_data[g]);
      if (c->peer_flag[g]) (void)hipIpcCloseMemHandle(c->peer_flag[g]);
    }
  }
  void* bufs[] = {c->data, c->flag, c->delta_dev, c->us_dev, c->keys_dev};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  (void)hipGetLastError();
  delete c;
  return GJX_OK;
}

extern "C" int gjx_peer_ctx_create(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device,
                                   gjx_peer_ctx** out) {
  if (!out || n_ranks < 1 || n_ranks > GJX_MAX_RANKS || rank < 0 || rank >= n_ranks || K_local <= 0 || rows < 1 ||
      ranks_on_this_device < 1)
    return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_create: bad argument");
  if (n_ranks > 1 && K_local % kPfHostThreads)
    return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_create: a sharded collection needs K_local % 1024 == 0 (whole quantisation tiles per rank)");
  gjx_peer_ctx* c = new (std::nothrow) gjx_peer_ctx();
  if (!c) return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_create: out of host memory");
  c->world = n_ranks; c->rank = rank; c->rows = rows; c->K = K_local; c->share = ranks_on_this_device;
  c->nt = (int)((K_local + kPfHostThreads - 1) / kPfHostThreads);
  c->NT = c->nt * n_ranks;
  for (int g = 0; g < GJX_MAX_RANKS; ++g) { c->peer_data[g] = nullptr; c->peer_flag[g] = nullptr; }
  // DATA window: rows[2][rows][K], logw[2][K]
  size_t o = 0;
  for (int p = 0; p < 2; ++p) { c->off_rows[p] = o; o = align_up(o + sizeof(float) * (size_t)rows * (size_t)K_local); }
  for (int p = 0; p < 2; ++p) { c->off_lw[p] = o; o = align_up(o + sizeof(float) * (size_t)K_local); }
  c->data_bytes = o;
  // FLAG window: [256 B control][region 0][region 1]; a region: granules A, B [NT] | ring sum, max [3][NT] | ready words |
  // the words of the one-launch resampling step (gmm: 4 x [world] u64 + this rank's tile granules [nt])
  const size_t NT = (size_t)c->NT;
  size_t r = 0;
  c->r_aggA = r; r = align_up(r + 8 * NT);
  c->r_aggB = r; r = align_up(r + 8 * NT);
  c->r_bsum = r; r = align_up(r + 12 * NT);
  c->r_bmax = r; r = align_up(r + 12 * NT);
  c->r_ready = r; r = align_up(r + 4 * (size_t)kPfHostMaxTiles);
  c->r_gmm = r; r = align_up(r + 8 * (4 * (size_t)GJX_MAX_RANKS + (size_t)c->nt));
  c->region_bytes = r;
  c->off_region[0] = kAlign;
  c->off_region[1] = kAlign + r;
  c->flag_bytes = kAlign + 2 * r;
  auto fail = [&](int code) { gjx_peer_ctx_destroy(c); return code; };
  if (hipMalloc((void**)&c->data, c->data_bytes) != hipSuccess) return fail(gjx_fail(GJX_EHIP, "gjx_peer_ctx_create: data window"));
  // flags are polled while other devices write them: uncached device memory (what RCCL uses for its own flags); plain
  // device memory as the fallback (every access to it carries its scope in the instruction anyway)
  if (hipExtMallocWithFlags((void**)&c->flag, c->flag_bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    c->flag = nullptr;
    if (hipMalloc((void**)&c->flag, c->flag_bytes) != hipSuccess) return fail(gjx_fail(GJX_EHIP, "gjx_peer_ctx_create: flag window"));
  }
  if (hipMemset(c->flag, 0, c->flag_bytes) != hipSuccess || hipMemset(c->data, 0, c->data_bytes) != hipSuccess ||
      hipMalloc((void**)&c->delta_dev, sizeof(long long) * 2 * (size_t)n_ranks) != hipSuccess ||
      hipMemset(c->delta_dev, 0, sizeof(long long) * 2 * (size_t)n_ranks) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    return fail(gjx_fail(GJX_EHIP, "gjx_peer_ctx_create: initialisation"));
  c->peer_data[rank] = c->data;
  c->peer_flag[rank] = c->flag;
  c->connected = n_ranks == 1;
  *out = c;
  return GJX_OK;
}

// out128 = {hipIpcMemHandle_t of the data window, hipIpcMemHandle_t of the flag window}
extern "C" int gjx_peer_ctx_export(gjx_peer_ctx* c, uint8_t* out128) {
  if (!c || !out128) return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_export: bad argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h[2];
  GJX_HIP(hipIpcGetMemHandle(&h[0], c->data), "gjx_peer_ctx_export(data)");
  GJX_HIP(hipIpcGetMemHandle(&h[1], c->flag), "gjx_peer_ctx_export(flag)");
  memcpy(out128, h, 128);
  return GJX_OK;
}

// handles = every rank's 128 bytes of gjx_peer_ctx_export, in rank order (this rank's own entry is ignored)
extern "C" int gjx_peer_ctx_connect(gjx_peer_ctx* c, const uint8_t* handles) {
  if (!c || !handles) return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_connect: bad argument");
  if (c->connected) return GJX_OK;
  std::vector<long long> delta(2 * (size_t)c->world, 0);
  for (int g = 0; g < c->world; ++g) {
    if (g == c->rank) continue;
    hipIpcMemHandle_t h[2];
    memcpy(h, handles + 128 * (size_t)g, 128);
    void* pd = nullptr;
    void* pf = nullptr;
    GJX_HIP(hipIpcOpenMemHandle(&pd, h[0], hipIpcMemLazyEnablePeerAccess), "gjx_peer_ctx_connect(data window)");
    c->peer_data[g] = (char*)pd;
    GJX_HIP(hipIpcOpenMemHandle(&pf, h[1], hipIpcMemLazyEnablePeerAccess), "gjx_peer_ctx_connect(flag window)");
    c->peer_flag[g] = (char*)pf;
    delta[g] = (long long)(c->peer_data[g] - c->data);
    delta[c->world + g] = (long long)(c->peer_flag[g] - c->flag);
  }
  GJX_HIP(hipMemcpy(c->delta_dev, delta.data(), sizeof(long long) * delta.size(), hipMemcpyHostToDevice), "gjx_peer_ctx_connect");
  c->connected = true;
  return GJX_OK;
}

// out6 = device pointers of this rank's {rows[0], rows[1], logw[0], logw[1]} and {data window bytes, flag window bytes}
extern "C" int gjx_peer_ctx_buffers(gjx_peer_ctx* c, uint64_t* out6) {
  if (!c || !out6) return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_buffers: bad argument");
  out6[0] = (uint64_t)(uintptr_t)(c->data + c->off_rows[0]);
  out6[1] = (uint64_t)(uintptr_t)(c->data + c->off_rows[1]);
  out6[2] = (uint64_t)(uintptr_t)(c->data + c->off_lw[0]);
  out6[3] = (uint64_t)(uintptr_t)(c->data + c->off_lw[1]);
  out6[4] = (uint64_t)c->data_bytes;
  out6[5] = (uint64_t)c->flag_bytes;
  return GJX_OK;
}

// status word of the context's kernels (bit 0: a rendezvous timed out, results undefined; bit 1: a dead collection), read
// and cleared; synchronises the stream
extern "C" int gjx_peer_ctx_status(gjx_peer_ctx* c, int32_t* status_host, void* stream) {
  if (!c || !status_host) return gjx_fail(GJX_EINVAL, "gjx_peer_ctx_status: bad argument");
  unsigned v = 0;
  unsigned* word = (unsigned*)c->flag + 10;
  GJX_HIP(hipMemcpyAsync(&v, word, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream), "gjx_peer_ctx_status");
  GJX_HIP(hipStreamSynchronize((hipStream_t)stream), "gjx_peer_ctx_status");
  if (v) GJX_HIP(hipMemsetAsync(word, 0, sizeof(unsigned), (hipStream_t)stream), "gjx_peer_ctx_status");
  *status_host = (int32_t)v;
  return GJX_OK;
}

extern "C" int gjx_ssm_step(const gjx_ssm*, uint32_t, uint32_t, int32_t, int32_t, int64_t, int64_t, const float*, int64_t, const int32_t*,
                            const float*, float*, float*, float*, int64_t, void*, size_t, void*);

// The bootstrap filter of gjx_ssm_filter_scheme(GJX_WEIGHTS_TILE_SCALED) on a collection sharded over the ranks of the
// context (created with rows == dx): every rank calls this with the same key and ys.  Two launches per rank whatever T is
// (step 0, then k_pf_persistent for steps 1 .. T-1); no host synchronisation, no collective call.  Results do not depend
// on the number of ranks.  The particles of the last step end up in rows[(T - 1) & 1] of the context, their log-weights
// in logw[0]; lse_steps f32[T][4] receives the GLOBAL record of every step on every rank; ancestors (optional) int32[K_local]
// the GLOBAL ancestor index of every slot at the last resampling.
extern "C" int gjx_ssm_filter_peer(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* c,
                                   const float* ys_dev, float* lse_steps, int32_t* ancestors, void* stream) {
  if (!m || !c || !ys_dev || !lse_steps || T < 2) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: bad argument (T >= 2)");
  if (!c->connected) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: the context is not connected (gjx_peer_ctx_connect)");
  if (c->rows != m->dx) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: the context must have rows == dx");
  if (!m->H_dev && m->dy != m->dx) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: H == NULL needs dy == dx");
  hipStream_t st = (hipStream_t)stream;
  const int64_t K = c->K, K_total = K * c->world;
  PfPlan pf;
  if (pf_plan(rng_mode, m->dx, m->dy, K, c->world, c->share, &pf) != GJX_OK)
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter_peer: this shape does not fit the one-launch filter (dx in {2,4,8,16}, dy <= 32, "
                                       "K_total <= 2^22, K_local / 1024 tiles co-resident at <= 8 tiles per block)");
  if (T > c->t_cap) {   // (per-run scratch grows outside every loop)
    if (c->us_dev) (void)hipFree(c->us_dev);
    if (c->keys_dev) (void)hipFree(c->keys_dev);
    c->us_dev = nullptr; c->keys_dev = nullptr; c->t_cap = 0;
    GJX_HIP(hipMalloc((void**)&c->us_dev, sizeof(double) * (size_t)T), "gjx_ssm_filter_peer: step offsets");
    GJX_HIP(hipMalloc((void**)&c->keys_dev, sizeof(uint32_t) * 2 * (size_t)T), "gjx_ssm_filter_peer: step keys");
    c->t_cap = T;
  }
  static thread_local std::vector<uint32_t> h_keys;
  static thread_local std::vector<double> h_us;
  pf_step_keys(key0, key1, T, h_keys, h_us);
  GJX_HIP(hipMemcpyAsync(c->us_dev, h_us.data(), sizeof(double) * (size_t)T, hipMemcpyHostToDevice, st), "gjx_ssm_filter_peer(step offsets)");
  GJX_HIP(hipMemcpyAsync(c->keys_dev, h_keys.data(), sizeof(uint32_t) * 2 * (size_t)T, hipMemcpyHostToDevice, st), "gjx_ssm_filter_peer(step keys)");
  float* x_a = (float*)(c->data + c->off_rows[0]);
  float* x_b = (float*)(c->data + c->off_rows[1]);
  float* lw_even = (float*)(c->data + c->off_lw[0]);
  float* lw_odd = (float*)(c->data + c->off_lw[1]);
  float* lw0 = ((T - 1) & 1) ? lw_odd : lw_even;     // log-weights of step 0
  // step 0 from the prior; its global LSE record comes out of the ring like every other step's (T > 1)
  int rc = gjx_ssm_step(m, h_keys[0], h_keys[1], rng_mode, 0, K, (int64_t)c->rank * K, nullptr, K, nullptr, ys_dev, x_a, lw0,
                        nullptr, K_total, nullptr, 0, stream);
  if (rc) return rc;
  const int region = (int)(c->n_filter & 1), other = region ^ 1;
  c->n_filter += 1;
  char* rg = c->flag + c->off_region[region];
  PfArgs f;
  memset(&f, 0, sizeof(f));
  f.A = m->A_dev; f.H = m->H_dev; f.ys = ys_dev; f.q = m->q; f.r = m->r; f.dy = m->dy; f.T = T;
  f.K = K; f.K_total = K_total; f.offset = (int64_t)c->rank * K; f.G = c->world; f.rank = c->rank; f.nt = c->nt; f.NT = c->NT;
  f.x_a = x_a; f.x_b = x_b; f.lw_even = lw_even; f.lw_odd = lw_odd;
  f.aggA = (unsigned long long*)(rg + c->r_aggA); f.aggB = (unsigned long long*)(rg + c->r_aggB);
  f.bsum = (float*)(rg + c->r_bsum); f.bmax = (float*)(rg + c->r_bmax); f.ready = (unsigned*)(rg + c->r_ready);
  f.peer_data = c->world > 1 ? c->delta_dev : nullptr;
  f.peer_flag = c->world > 1 ? c->delta_dev + c->world : nullptr;
  f.keys = c->keys_dev; f.us = c->us_dev; f.lse_steps = lse_steps; f.ancestors = ancestors;
  f.ctrl = (unsigned*)c->flag + 8; f.log_k = (float)log((double)K_total);
  // the other ranks' launches may be queued behind host work of their own: the first rendezvous waits for seconds, later ones ~0.1 s
  f.first_budget = c->world > 1 ? (1u << 24) : (1u << 17);
  f.zero_ptr = (unsigned long long*)(c->flag + c->off_region[other] + c->r_aggA);
  f.zero_n = (int)((c->r_bsum - c->r_aggA) / 8);
  void* args[] = {&f};
  const hipError_t e = hipLaunchKernel(pf.fn, dim3((unsigned)pf.grid), dim3(kPfHostThreads), args, pf.lds, st);
  if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter_peer(k_pf_persistent)");
  return GJX_OK;
}
