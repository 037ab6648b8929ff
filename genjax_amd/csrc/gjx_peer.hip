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
#include "gjx_pfcore.h"

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
  size_t off_rows[2] = {0, 0}, off_lw[2] = {0, 0}, off_m[2] = {0, 0};   // off_m: transition means of the resample-move filter
  size_t off_chk[2] = {0, 0};          // verify mode: one check word per particle row (u32[K], ping-pong like the rows)
  bool verify = false;                 // GJX_PEER_VERIFY=1 when the context was created
  bool verify_fault = false;           // GJX_PEER_VERIFY_FAULT=<this rank>: publish wrong check words (test hook)
  bool data_fine = false;              // DATA window in fine-grained memory (GJX_PEER_DATA=fine)
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
  for (int g = 0; g < c->world; ++g) {   // (whatever `connected` says: a connect that failed half-way has opened some of them)
    if (g == c->rank) continue;
    if (c->peer_data[g]) (void)hipIpcCloseMemHandle(c->peer_data[g]);
    if (c->peer_flag[g]) (void)hipIpcCloseMemHandle(c->peer_flag[g]);
    c->peer_data[g] = nullptr; c->peer_flag[g] = nullptr;
  }
  void* bufs[] = {c->data, c->flag, c->delta_dev, c->us_dev, c->keys_dev};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  (void)hipGetLastError();
  delete c;
  return GJX_OK;
}

static int peer_ctx_create(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device, int32_t flags,
                           gjx_peer_ctx** out);
// the switches from the environment (GJX_PEER_VERIFY, GJX_PEER_DATA, GJX_PEER_VERIFY_FAULT), read here and nowhere else
extern "C" int gjx_peer_ctx_create(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device,
                                   gjx_peer_ctx** out) {
  int32_t flags = 0;
  { const char* v = getenv("GJX_PEER_VERIFY"); if (v && atoi(v) != 0) flags |= GJX_PEER_VERIFY_ON; }
  { const char* v = getenv("GJX_PEER_DATA"); if (v && !strcmp(v, "fine")) flags |= GJX_PEER_DATA_FINE; }
  { const char* v = getenv("GJX_PEER_VERIFY_FAULT"); if ((flags & GJX_PEER_VERIFY_ON) && v && atoi(v) == rank) flags |= GJX_PEER_VERIFY_FAULTY; }
  return peer_ctx_create(n_ranks, rank, K_local, rows, ranks_on_this_device, flags, out);
}
// the same with the switches as an argument (no environment): GJX_PEER_VERIFY_ON | GJX_PEER_DATA_FINE | GJX_PEER_VERIFY_FAULTY
extern "C" int gjx_peer_ctx_create_ex(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device,
                                      int32_t flags, gjx_peer_ctx** out) {
  return peer_ctx_create(n_ranks, rank, K_local, rows, ranks_on_this_device, flags, out);
}
static int peer_ctx_create(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device, int32_t flags,
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
  for (int p = 0; p < 2; ++p) { c->off_m[p] = o; o = align_up(o + sizeof(float) * (size_t)rows * (size_t)K_local); }
  for (int p = 0; p < 2; ++p) { c->off_chk[p] = o; o = align_up(o + sizeof(uint32_t) * (size_t)K_local); }
  c->data_bytes = o;
  c->verify = (flags & GJX_PEER_VERIFY_ON) != 0;
  c->data_fine = (flags & GJX_PEER_DATA_FINE) != 0;
  // test hook of the verify mode: this rank publishes check words that do not belong to its rows (tests/test_gpu_config4.py)
  c->verify_fault = c->verify && (flags & GJX_PEER_VERIFY_FAULTY) != 0;
  // FLAG window: [256 B control][region 0][region 1]; a region: granules A, B [NT] | ring sum, max [3][NT] | ready words |
  // the words of the one-launch resampling step (twice, for alternating calls: 4 x [MAX_RANKS] u64 + this rank's tile granules [nt])
  const size_t NT = (size_t)c->NT;
  size_t r = 0;
  c->r_aggA = r; r = align_up(r + 8 * NT * kPfGranulePad);   // (one granule per 64-byte line)
  c->r_aggB = r; r = align_up(r + 8 * NT * kPfGranulePad);
  c->r_bsum = r; r = align_up(r + 12 * NT);
  c->r_bmax = r; r = align_up(r + 12 * NT);
  c->r_ready = r; r = align_up(r + 4 * (size_t)kPfHostMaxTiles);
  c->r_gmm = r; r = align_up(r + 2 * 8 * (4 * (size_t)GJX_MAX_RANKS + (size_t)c->nt));   // x2: alternating calls
  c->region_bytes = r;
  c->off_region[0] = kAlign;
  c->off_region[1] = kAlign + r;
  c->flag_bytes = kAlign + 2 * r;
  auto fail = [&](int code) { gjx_peer_ctx_destroy(c); return code; };
  // DATA window.  Default: ordinary (coarse-grained) device memory — every access another rank can observe carries its scope
  // in the instruction (sc0 sc1 write-through stores, sc0 sc1 loads: DESIGN.md §8 "memory model of the peer windows").
  // GJX_PEER_DATA=fine: fine-grained device memory (hipDeviceMallocFinegrained) — coherent between agents by its memory
  // type, whatever the instructions say; the fallback for a fabric on which the coarse window shows stale rows
  // (GJX_PEER_VERIFY=1 raises GJX_STATUS_VERIFY_MISMATCH then).  Every rank of a run must use the same setting.
  if (c->data_fine) {
    if (hipExtMallocWithFlags((void**)&c->data, c->data_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      c->data = nullptr;
      return fail(gjx_fail(GJX_EHIP, "gjx_peer_ctx_create: fine-grained data window (GJX_PEER_DATA=fine)"));
    }
  } else if (hipMalloc((void**)&c->data, c->data_bytes) != hipSuccess) return fail(gjx_fail(GJX_EHIP, "gjx_peer_ctx_create: data window"));
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
  // per-run scratch of the filter (step keys, comb offsets): sized here for T <= 4096 so that no call allocates
  c->t_cap = 4096;
  if (hipMalloc((void**)&c->us_dev, sizeof(double) * (size_t)c->t_cap) != hipSuccess ||
      hipMalloc((void**)&c->keys_dev, sizeof(uint32_t) * 2 * (size_t)c->t_cap) != hipSuccess)
    return fail(gjx_fail(GJX_EHIP, "gjx_peer_ctx_create: step-key scratch"));
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
static int filter_peer(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* c, const float* ys_dev,
                       float* lse_steps, int32_t* ancestors, bool move, int n_moves, float move_scale, unsigned long long* acc_total,
                       void* stream);
extern "C" int gjx_ssm_filter_peer(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* c,
                                   const float* ys_dev, float* lse_steps, int32_t* ancestors, void* stream) {
  return filter_peer(m, key0, key1, rng_mode, T, c, ys_dev, lse_steps, ancestors, false, 0, 0.0f, nullptr, stream);
}
// the sharded filter with resample-move rejuvenation (gjx_ssm_filter_move on a sharded collection): the transition means
// live in the DATA window like the states, a moved particle's parent mean is pulled from the rank that holds the ancestor
extern "C" int gjx_ssm_filter_peer_move(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* c,
                                        const float* ys_dev, float* lse_steps, int32_t* ancestors, int32_t n_moves, float move_scale,
                                        uint64_t* accepted_total, void* stream) {
  if (n_moves < 0) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer_move: bad argument");
  return filter_peer(m, key0, key1, rng_mode, T, c, ys_dev, lse_steps, ancestors, true, (int)n_moves, move_scale,
                     (unsigned long long*)accepted_total, stream);
}
static int filter_peer(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* c, const float* ys_dev,
                       float* lse_steps, int32_t* ancestors, bool move, int n_moves, float move_scale, unsigned long long* acc_total,
                       void* stream) {
  if (!m || !c || !ys_dev || !lse_steps || T < 2) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: bad argument (T >= 2)");
  if (!c->connected) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: the context is not connected (gjx_peer_ctx_connect)");
  if (c->rows != m->dx) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: the context must have rows == dx");
  if (!m->H_dev && m->dy != m->dx) return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_peer: H == NULL needs dy == dx");
  hipStream_t st = (hipStream_t)stream;
  const int64_t K = c->K, K_total = K * c->world;
  PfPlan pf;
  if (pf_plan(rng_mode, m->dx, m->dy, K, c->world, c->share, &pf, move) != GJX_OK)
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter_peer: this shape does not fit the one-launch filter (dx in {2,4,8,16}, dy <= 32, "
                                       "K_total <= 2^22, K_local / 1024 tiles co-resident at <= 8 tiles per block)");
  // (nothing is allocated after gjx_peer_ctx_create: the context's per-step scratch holds GJX_PEER_MAX_STEPS steps; longer runs are
  // cut by the caller — the carry of one call is the start of the next)
  if (T > c->t_cap) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter_peer: more steps than a peer context holds (GJX_PEER_MAX_STEPS = 4096 per call)");
  std::vector<uint32_t> h_keys;        // per-call staging; the words travel as kernel arguments (upload_words): nothing of it is read later
  std::vector<double> h_us;
  pf_step_keys(key0, key1, T, h_keys, h_us);
  if (int rcu = upload_words(c->us_dev, h_us.data(), (size_t)T, st)) return rcu;
  if (int rck = upload_words(c->keys_dev, h_keys.data(), (size_t)T, st)) return rck;      // (T pairs of 32-bit words)
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
  f.first_budget = c->world > 1 ? (1u << 24) : (1u << 16);
  f.zero_ptr = (unsigned long long*)(c->flag + c->off_region[other] + c->r_aggA);
  f.zero_n = (int)((c->r_bsum - c->r_aggA) / 8);
  f.q0 = m->q0;
  f.verify = c->verify ? (c->verify_fault ? 2 : 1) : 0;
  f.chk_a = (unsigned*)(c->data + c->off_chk[0]); f.chk_b = (unsigned*)(c->data + c->off_chk[1]);
  if (move) {
    f.m_a = (float*)(c->data + c->off_m[0]); f.m_b = (float*)(c->data + c->off_m[1]);
    f.n_moves = n_moves; f.move_scale = move_scale; f.acc_total = acc_total;
    if (acc_total) GJX_HIP(hipMemsetAsync(acc_total, 0, sizeof(unsigned long long), st), "gjx_ssm_filter_peer_move(accept counter)");
  }
  void* args[] = {&f};
  const hipError_t e = hipLaunchKernel(pf.fn, dim3((unsigned)pf.grid), dim3(kPfHostThreads), args, pf.lds, st);
  if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter_peer(k_pf_persistent)");
  return GJX_OK;
}

// ------------------------------------------------------------------------------------------------------------
// The bootstrap filter for ANY Scan kernel (gjx_scan_filter, GJX_FILTER_FORM_WIDE) on a collection sharded over the ranks of the
// context: step 0 by the step program's own kernel, then steps 1 .. T-1 in ONE launch of the filter kernel generated for the step
// program (gjx_gen_pf, sharded flavour) — the model of the skeleton k_pf_persistent runs on, so the exchange is that filter's:
// granules pushed into every rank's flag window, source tiles' log-weights and the ancestors' carry rows pulled from the
// owner's data window, one rendezvous per step, no host, no collective call.  The context must hold rows >= every step's
// n_slots; every rank passes its OWN copies of the step programs (same structure, same tables) and the same key.
// Particles and weights do not depend on the number of ranks (streams and resampling integers are global).
// ------------------------------------------------------------------------------------------------------------
static int scan_filter_peer_impl(gjx_peer_ctx* c, const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, float* lse_steps,
                                 int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream, const gjx_filter_opts* opts,
                                 gjx_filter_info* info_out, bool prepare_only);

extern "C" int gjx_scan_filter_peer(gjx_peer_ctx* c, const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, float* lse_steps,
                                    int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream, gjx_filter_info* info_out) {
  return scan_filter_peer_impl(c, steps, T, key0, key1, lse_steps, ancestors, workspace, workspace_bytes, stream, nullptr, info_out, false);
}
// ... with the options of gjx_scan_filter that the sharded kernel carries: GJX_FILTER_MULTINOMIAL (resampling by sorted uniforms — the
// spacing sums of every tile travel to every rank as the second word of its granule, no exchange beyond the systematic filter's)
extern "C" int gjx_scan_filter_peer_opts(gjx_peer_ctx* c, const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, float* lse_steps,
                                         int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream, const gjx_filter_opts* opts,
                                         gjx_filter_info* info_out) {
  return scan_filter_peer_impl(c, steps, T, key0, key1, lse_steps, ancestors, workspace, workspace_bytes, stream, opts, info_out, false);
}
extern "C" int gjx_scan_filter_peer_prepare_opts(gjx_peer_ctx* c, const gjx_program* steps, int32_t T, const gjx_filter_opts* opts, gjx_filter_info* info_out) {
  return scan_filter_peer_impl(c, steps, T, 0u, 0u, nullptr, nullptr, nullptr, 0, nullptr, opts, info_out, true);
}
// everything of the call above that takes unpredictable host time — generating, compiling (hipRTC) and loading the kernels of the
// step programs, the occupancy queries behind the choice of tiles per block — and no launch: the ranks of a job call it, meet at a
// HOST barrier, and only then enter the filter together (a rank that waits for a peer still compiling would run out of its poll budget)
extern "C" int gjx_scan_filter_peer_prepare(gjx_peer_ctx* c, const gjx_program* steps, int32_t T, gjx_filter_info* info_out) {
  return scan_filter_peer_impl(c, steps, T, 0u, 0u, nullptr, nullptr, nullptr, 0, nullptr, nullptr, info_out, true);
}

static int scan_filter_peer_impl(gjx_peer_ctx* c, const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, float* lse_steps,
                                 int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream, const gjx_filter_opts* opts,
                                 gjx_filter_info* info_out, bool prepare_only) {
  gjx_filter_info finfo = {GJX_FILTER_FORM_WIDE, 0, 0, 0};
  if (info_out) *info_out = finfo;
  const int fflags = opts ? opts->flags : 0;
  if (opts && opts->n_moves > 0) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: the sharded filter kernel has no rejuvenation move (gjx_scan_filter on one GPU has)");
  if (fflags & ~(GJX_FILTER_MULTINOMIAL)) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: of gjx_filter_opts.flags the sharded kernel takes GJX_FILTER_MULTINOMIAL only (it has one form)");
  const bool multinomial = (fflags & GJX_FILTER_MULTINOMIAL) != 0;
  const int flavour = 256 | (multinomial ? 1024 : 0);
  if (!c || !steps || (!lse_steps && !prepare_only) || T < 2) return gjx_fail(GJX_EINVAL, "gjx_scan_filter_peer: bad argument (T >= 2)");
  if (!c->connected) return gjx_fail(GJX_EINVAL, "gjx_scan_filter_peer: the context is not connected (gjx_peer_ctx_connect)");
  auto input_rows = [](const gjx_program& p) {
    int n = 0;
    for (int j = 0; j < p.n_sites; ++j) if (p.sites[j].mode == GJX_MODE_INPUT) n += p.sites[j].dim;
    return n;
  };
  for (int t = 0; t < T; ++t)
    if (steps[t].n_slots > c->rows) return gjx_fail(GJX_EINVAL, "gjx_scan_filter_peer: a step program has more rows than the context (rows >= n_slots of every step)");
  if (!steps[1].tab_dev || !gen_pf_supported(&steps[1]))
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: the step program is outside the filter emitter's coverage (sites SAMPLE / OBS_TAB / INPUT)");
  for (int u = 2; u < T; ++u)
    if (steps[u].n_tab != steps[1].n_tab || steps[u].n_slots != steps[1].n_slots || input_rows(steps[u]) != input_rows(steps[1]) || !steps[u].tab_dev ||
        !gen_pf_same_kernel(&steps[1], &steps[u]))
      return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: the step programs 1 .. T-1 must be one kernel (a periodic Scan)");
  if (input_rows(steps[1]) > steps[0].n_slots - input_rows(steps[0])) return gjx_fail(GJX_EINVAL, "gjx_scan_filter_peer: step 1 reads more carry rows than step 0 produced");
  if (c->NT > kPfHostMaxTiles) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: K_total <= 2^22");
  const size_t need_run = gjx_workspace_bytes(GJX_OP_RUN, c->K);
  if (!prepare_only && (!workspace || workspace_bytes < need_run + 8 * (size_t)T + 256)) return gjx_fail(GJX_EWORKSPACE, "gjx_scan_filter_peer: workspace too small (OP_RUN + 8 T + 256)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t K = c->K, K_total = K * c->world;
  const size_t dyn = pf_core_dyn_lds(c->NT, multinomial);
  int spl = 0, grid = 0;
  const int spls[5] = {1, 2, 4, 8, 16};
  for (int i = 0; i < 5 && !spl; ++i) {
    const int64_t g = ((int64_t)c->nt + spls[i] - 1) / spls[i];
    int cap = gen_pf_resident_blocks(&steps[1], spls[i] | flavour, dyn);
    if (cap <= 0) break;
    if (c->share > 1) cap /= c->share;               // ranks that share one device (dry runs): every rank's grid must be resident
    if (g <= cap && g * c->world <= kPfHostMaxTiles) { spl = spls[i]; grid = (int)g; }
  }
  if (!spl) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: no co-resident grid for this size (or the kernel could not be generated)");
  if (T > c->t_cap) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter_peer: more steps than a peer context holds (GJX_PEER_MAX_STEPS = 4096 per call)");
  if (prepare_only) {
    // step 0's own kernel (gjx_run_program_ex picks and compiles it on first use), then nothing is launched
    finfo.launches = 0; finfo.grid = grid; finfo.tiles_per_block = spl;
    if (info_out) *info_out = finfo;
    (void)gjx_program_precompile(&steps[0], gen_pick_ppt(&steps[0], c->K, false));     // (step 0 may run on another engine: not an error)
    return GJX_OK;
  }
  std::vector<uint32_t> h_keys, h_res;
  std::vector<double> h_us;
  pf_step_keys_res(key0, key1, T, h_keys, h_us, h_res);
  if (multinomial)      // (the sorted-uniform resampler takes the resampling KEY of a step where the comb takes its offset: two words as one 64-bit pattern)
    for (int u = 0; u < T; ++u) { const uint64_t w = (uint64_t)h_res[2 * u] | ((uint64_t)h_res[2 * u + 1] << 32); memcpy(&h_us[u], &w, 8); }
  if (int rcu = upload_words(c->us_dev, h_us.data(), (size_t)T, st)) return rcu;
  if (int rcu = upload_words(c->keys_dev, h_keys.data(), (size_t)T, st)) return rcu;
  const float** tabs_dev = (const float**)((char*)workspace + ((need_run + 255) & ~(size_t)255));
  std::vector<const float*> h_tabs((size_t)T, nullptr);
  for (int u = 0; u < T; ++u) h_tabs[u] = steps[u].tab_dev;
  if (int rcu = upload_words(tabs_dev, h_tabs.data(), (size_t)T, st)) return rcu;
  float* rows_a = (float*)(c->data + c->off_rows[0]);
  float* rows_b = (float*)(c->data + c->off_rows[1]);
  float* lw_even = (float*)(c->data + c->off_lw[0]);
  float* lw_odd = (float*)(c->data + c->off_lw[1]);
  float* lw0 = ((T - 1) & 1) ? lw_odd : lw_even;     // log-weights of step 0
  // step 0 (no carry to read) on this rank's particle range; its global LSE record comes out of the ring like every other step's
  gjx_run_opts o;
  memset(&o, 0, sizeof(o));
  gjx_run_info info = {0, 0, 0};
  int rc = gjx_run_program_ex(&steps[0], h_keys[0], h_keys[1], K, (int64_t)c->rank * K, rows_a, nullptr, nullptr, lw0, nullptr, nullptr, nullptr, nullptr,
                              K_total, workspace, need_run, stream, &o, &info);
  if (rc) return rc;
  const int region = (int)(c->n_filter & 1), other = region ^ 1;
  c->n_filter += 1;
  char* rg = c->flag + c->off_region[region];
  GenPfArgs ga;
  memset(&ga, 0, sizeof(ga));
  PfCoreArgs& f = ga.core;
  f.T = T; f.K = K; f.K_total = K_total; f.offset = (int64_t)c->rank * K; f.G = c->world; f.rank = c->rank; f.nt = c->nt; f.NT = c->NT;
  f.lw_even = lw_even; f.lw_odd = lw_odd;
  f.aggA = (unsigned long long*)(rg + c->r_aggA); f.aggB = (unsigned long long*)(rg + c->r_aggB);
  f.bsum = (float*)(rg + c->r_bsum); f.bmax = (float*)(rg + c->r_bmax); f.ready = (unsigned*)(rg + c->r_ready);
  f.peer_data = c->world > 1 ? c->delta_dev : nullptr;
  f.peer_flag = c->world > 1 ? c->delta_dev + c->world : nullptr;
  f.keys = c->keys_dev; f.us = c->us_dev; f.lse_steps = lse_steps; f.ancestors = ancestors; f.ancestors_all = nullptr;
  f.ctrl = (unsigned*)c->flag + 8; f.log_k = (float)log((double)K_total);
  f.first_budget = c->world > 1 ? (1u << 24) : (1u << 16);
  f.zero_ptr = (unsigned long long*)(c->flag + c->off_region[other] + c->r_aggA);
  f.zero_n = (int)((c->r_bsum - c->r_aggA) / 8);
  f.verify = c->verify ? (c->verify_fault ? 2 : 1) : 0;
  f.chk_a = (unsigned*)(c->data + c->off_chk[0]); f.chk_b = (unsigned*)(c->data + c->off_chk[1]);
  f.timeline = nullptr;
  ga.tabs = tabs_dev;
  ga.rows_a = rows_a; ga.rows_b = rows_b; ga.rows_all = nullptr; ga.rows_step = 0;
  ga.in_row0_first = (int64_t)input_rows(steps[0]) * K;
  ga.in_row0 = (int64_t)input_rows(steps[1]) * K;
  rc = gen_pf_launch(&steps[1], spl | flavour, ga, grid, dyn, st);
  if (rc) return rc;
  finfo.launches = 2; finfo.grid = grid; finfo.tiles_per_block = spl;
  if (info_out) *info_out = finfo;
  return GJX_OK;
}

// ------------------------------------------------------------------------------------------------------------
// One sharded ImportanceK resampling step in ONE launch per rank: log-weights -> systematic ancestors over the WHOLE
// collection -> this rank's children, pulled from whichever rank holds the ancestor.  k_resample_gather's consumer-side
// scheme (gjx_resample.hip) under the tile-scaled fixed point (GJX_WEIGHTS_TILE_SCALED): a block quantises its own tile
// of 1024 particles against the tile's power of two, the blocks of a rank all-gather their {e_b, S_b} granules locally,
// and the ranks meet twice through their flag windows — hop 1: the largest tile exponent E of the collection, hop 2:
// the rank totals at that exponent (G words each) — after which every block knows the global comb, finds the source
// rank of its slots from the G + 1 rank bounds, and searches that rank's tile prefix (its own: already in LDS; another
// rank's: that rank's granule array, pulled through the mapping).  Same integers as gjx_resample_indices_tiled on the
// unsharded collection: G_b = S_b >> (E - e_b), prefix over all tiles in rank order, thresholds floor((j + u) total / N),
// residual << (E - e_b) looked up in the tile's own cumulative q.  One rank: both hops vanish.
// ------------------------------------------------------------------------------------------------------------
namespace gjx {

struct PrgArgs {
  const float* x;                  // [K] log-weights of this rank (DATA window: other ranks read them)
  int64_t K, K_total, offset;      // particles here / in total; global index of this rank's first particle
  int G, rank, nt;                 // ranks, this rank, tiles per rank (= gridDim.x)
  int mode;                        // 2: `lse` = n_partials per-block {max, sumexp} pairs of the producing kernel; 0: no LSE record
  const float* lse;
  int n_partials;
  float* lse_out;                  // [4] global record (every rank), or NULL
  float log_k_total;
  double u;
  const float* src;                // [rows][src_stride] this rank's particle rows (DATA window)
  int64_t src_stride;
  int rows;
  float* dst;                      // [rows][dst_stride] children of this rank's slots (private)
  int64_t dst_stride;
  int32_t* ancestors;              // [K] global ancestor index (or NULL)
  unsigned long long* agg;         // [nt] this rank's tile granules (FLAG window: other ranks read them)
  unsigned long long* wE;          // [G] hop 1: tagged max tile exponent of every rank (this rank's copy)
  unsigned long long* wR;          // [G] hop 2: tagged rank total at the global exponent
  unsigned long long* wP;          // [G] {max, sumexp} of every rank's log-weights (lands before its hop-1 word)
  const long long* peer_data;      // [G] byte distances to the other ranks' windows (NULL: one rank)
  const long long* peer_flag;
  unsigned* ctrl;
  unsigned seq;                    // call counter of the context (same on every rank): the tags
  unsigned first_budget;
  int verify;                      // GJX_PEER_VERIFY=1: check words per particle row + tile totals against the granules
  unsigned* chk;                   // [K] this rank's check words of the rows of this call (DATA window)
};

constexpr int kPrgMaxTiles = 1024;

template <int ITEMS>
__global__ __launch_bounds__(256) void k_peer_resample_gather(PrgArgs a) {
  constexpr int TILE = 256 * ITEMS;
  static_assert(TILE == kTileQ, "a block owns one quantisation tile");
  constexpr int CH = 3;                          // source tiles re-scanned per round
  __shared__ float fred[8], s_tm[4], s_em[4];
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t P[kPrgMaxTiles + 1];
  __shared__ int32_t Eb[kPrgMaxTiles];
  __shared__ uint64_t cumL[CH * TILE];
  __shared__ uint64_t s_wtot[CH][4];
  __shared__ uint64_t RB[GJX_MAX_RANKS + 1];
  __shared__ long long sPD[GJX_MAX_RANKS], sPF[GJX_MAX_RANKS];
  __shared__ int s_range[2], s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool sys = a.G > 1;
  const int nb = (int)gridDim.x, G = a.G;
  const int64_t K = a.K;
  const unsigned long long tag4 = (unsigned long long)(a.seq % 15u) + 1ull;
  const unsigned long long tag14 = (unsigned long long)(a.seq % 16383u) + 1ull;
  if (tid < GJX_MAX_RANKS) {
    sPD[tid] = (a.peer_data && tid < G) ? a.peer_data[tid] : 0;
    sPF[tid] = (a.peer_flag && tid < G) ? a.peer_flag[tid] : 0;
  }
  if (tid == 0) s_dead = 0;
  unsigned budget = a.first_budget;
  auto timed_out = [&]() {
    __hip_atomic_fetch_or(&a.ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dead = 1;
    budget = 0;
  };
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + tid) * ITEMS;   // first particle this lane scans = first slot it produces
  float xv[ITEMS];
  if (ITEMS == 4 && i0 + 4 <= K) {
    const float4 q4 = *(const float4*)(a.x + i0);
    xv[0] = q4.x; xv[1] = q4.y; xv[2] = q4.z; xv[3] = q4.w;
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) xv[k] = (i0 + k < K) ? a.x[i0 + k] : -INFINITY;
  }
  float sm_sum = 0.0f, mx_r = -INFINITY;
  if (a.mode == 2) mx_r = block_ref_max(2, a.lse, a.n_partials, fred, &sm_sum);      // this rank's {max, sumexp} (ends with a barrier)
  else __syncthreads();                                                             // (sPF / s_dead visible)
  // ---- own tile: exponent, fixed-point weights, total -> local granule ----
  {
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) m = fmaxf(m, xv[k]);
    const float wm = wave_max_dpp(m);
    if (lane == 0) s_tm[wid] = wm;
  }
  __syncthreads();
  const int eb = tile_exponent(fmaxf(fmaxf(s_tm[0], s_tm[1]), fmaxf(s_tm[2], s_tm[3])));
  {
    uint64_t qs = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) qs += (i0 + k < K) ? tile_q(xv[k], eb) : 0;
    const uint64_t wt = wave_sum_u64(qs);
    if (lane == 0) wsum[wid] = wt;
  }
  __syncthreads();
  const bool verify = a.verify != 0;
  auto verify_failed = [&]() { __hip_atomic_fetch_or(&a.ctrl[2], kStatusVerifyMismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  if (verify) {
    // the rows of this block's own particles (the caller's kernels wrote them) get their check words before the block's
    // granule goes out: a remote reader pulls a row of this rank only after this rank's hop-2 word, which follows every granule
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      if (i0 + k >= K) continue;
      uint32_t h = row_check_init((int)a.seq, (uint32_t)(a.offset + i0 + k));
      for (int r = 0; r < a.rows; ++r) h = row_check_mix(h, a.src[(int64_t)r * a.src_stride + i0 + k]);
      store_scoped_u32(a.chk + i0 + k, a.verify == 2 ? h ^ 1u : h, sys);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (tid == 0) store_scoped_u64(&a.agg[blockIdx.x], tile_granule(tag4, eb, wsum[0] + wsum[1] + wsum[2] + wsum[3]), sys);
  if (blockIdx.x == 0 && a.mode == 2 && sys && tid < G)          // this rank's LSE pair, to every rank (ordered before the hop-1 word below)
    store_scoped_u64(peer_ptr(a.wP + a.rank, sPF[tid]), pack_f2(mx_r, sm_sum), true);
  // ---- all-gather of this rank's granules (local) ----
  auto read_granules = [&](const unsigned long long* arr, bool wait_tag) {
    float em = (float)kTileDead;
    for (int b = tid; b < nb; b += 256) {
      unsigned long long v = load_scoped_u64(&arr[b], sys);
      if (wait_tag) {
        while ((v >> 60) != tag4 && budget) {
          --budget;
          __builtin_amdgcn_s_sleep(1);
          v = load_scoped_u64(&arr[b], sys);
        }
        if ((v >> 60) != tag4) { timed_out(); v = 0; }
      }
      const uint64_t S = v & ((1ull << 40) - 1);
      const int e = S ? (int)((v >> 40) & 0xFFFFFu) + kTileDead : kTileDead;
      P[b + 1] = S;
      Eb[b] = e;
      em = fmaxf(em, (float)e);
    }
    em = wave_max_dpp(em);
    if (lane == 0) s_em[wid] = em;
    if (tid == 0) P[0] = 0;
    __syncthreads();
    return (int)fmaxf(fmaxf(s_em[0], s_em[1]), fmaxf(s_em[2], s_em[3]));
  };
  // P[b + 1] = S_b >> (E - e_b), then the inclusive prefix in place: thread t owns the entries [t per, (t + 1) per)
  auto shift_and_prefix = [&](int E) {
    const int per = (nb + 255) >> 8;
    const int e0 = tid * per < nb ? tid * per : nb, e1 = (e0 + per) < nb ? (e0 + per) : nb;
    uint64_t loc = 0;
    for (int e = e0; e < e1; ++e) {
      const int sh = E - Eb[e];
      const uint64_t g = sh < 64 ? P[e + 1] >> sh : 0;
      P[e + 1] = g;
      loc += g;
    }
    const uint64_t inc = wave_scan_u64(loc);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint64_t run = inc - loc;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    for (int e = e0; e < e1; ++e) { run += P[e + 1]; P[e + 1] = run; }
    __syncthreads();
  };
  budget = kPollBudget;                          // the local gather: blocks of one launch
  int E = read_granules(a.agg, true);
  // ---- hop 1: the largest tile exponent over all ranks ----
  if (sys) {
    budget = s_dead ? 0u : a.first_budget;       // the other ranks' launches may start later than this one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (block 0: the LSE pair has landed before the hop-1 word goes out)
    if (blockIdx.x == 0 && tid < G)
      store_scoped_u64(peer_ptr(a.wE + a.rank, sPF[tid]), (tag14 << 50) | (unsigned long long)(unsigned)(E - kTileDead), true);
    float em = (float)kTileDead;
    if (tid < G) {
      unsigned long long v = load_scoped_u64(&a.wE[tid], true);
      while ((v >> 50) != tag14 && budget) {
        --budget;
        __builtin_amdgcn_s_sleep(1);
        v = load_scoped_u64(&a.wE[tid], true);
      }
      if ((v >> 50) != tag14) { timed_out(); v = 0; }
      em = (float)((int)(v & 0xFFFFFu) + kTileDead);
    }
    em = wave_max_dpp(em);
    __syncthreads();                             // s_em of read_granules has been read by everybody
    if (lane == 0) s_em[wid] = em;
    __syncthreads();
    E = (int)fmaxf(fmaxf(s_em[0], s_em[1]), fmaxf(s_em[2], s_em[3]));
  }
  E = __builtin_amdgcn_readfirstlane(E);
  shift_and_prefix(E);
  const uint64_t R_own = P[nb];
  // ---- hop 2: the rank totals at that exponent -> rank bounds on the global weight line ----
  if (sys) {
    budget = s_dead ? 0u : kPollBudget;
    if (blockIdx.x == 0 && tid < G)
      store_scoped_u64(peer_ptr(a.wR + a.rank, sPF[tid]), (tag14 << 50) | (R_own & kAggMask), true);
    if (tid < G) {
      unsigned long long v = load_scoped_u64(&a.wR[tid], true);
      while ((v >> 50) != tag14 && budget) {
        --budget;
        __builtin_amdgcn_s_sleep(1);
        v = load_scoped_u64(&a.wR[tid], true);
      }
      if ((v >> 50) != tag14) { timed_out(); v = 0; }
      RB[tid + 1] = v & kAggMask;
    }
    __syncthreads();
    if (tid == 0) {
      uint64_t run = 0;
      RB[0] = 0;
      for (int g = 0; g < G; ++g) { run += RB[g + 1]; RB[g + 1] = run; }
    }
    __syncthreads();
  } else {
    if (tid == 0) { RB[0] = 0; RB[1] = R_own; }
    __syncthreads();
  }
  const uint64_t total = RB[G];
  // the global LSE record (every rank writes its own copy): the G pairs landed before the hop-1 words this block has seen
  if (a.mode == 2 && a.lse_out && blockIdx.x == 0 && tid == 0) {
    float m = mx_r, s = sm_sum;
    if (sys) {
      m = -INFINITY; s = 0.0f;
      for (int g = 0; g < G; ++g) {
        const unsigned long long pr = g == a.rank ? pack_f2(mx_r, sm_sum) : load_scoped_u64(&a.wP[g], true);
        const float pm = __uint_as_float((uint32_t)pr), ps = __uint_as_float((uint32_t)(pr >> 32));
        const float nm = fmaxf(m, pm);
        if (nm > -INFINITY) s = s * fast_exp(m - nm) + ps * fast_exp(pm - nm);
        m = nm;
      }
    }
    const float l = m > -INFINITY ? m + logf(s) : -INFINITY;
    a.lse_out[0] = m; a.lse_out[1] = s; a.lse_out[2] = l; a.lse_out[3] = l - a.log_k_total;
  }
  if (total == 0 && blockIdx.x == 0 && tid == 0) __hip_atomic_fetch_or(&a.ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- ancestors of slots i0 .. i0 + ITEMS - 1, in TILE space: (global tile) * 1024 + index in the tile ----
  const int kpad = nb * TILE;
  int anc[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) anc[k] = a.rank * kpad + (int)((i0 + k < K) ? i0 + k : 0);   // dead collection: identity (flagged)
  if (total > 0) {                               // block-uniform
    const double step = (double)total / (double)a.K_total;
    const double inv_step = (double)a.K_total / (double)total;
    uint64_t T[ITEMS];
    int gk[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int64_t j = a.offset + ((i0 + k < K) ? i0 + k : K - 1);
      T[k] = comb_threshold(j, a.u, step, total);
      int g = a.rank;                            // source rank: the one whose stretch [RB[g], RB[g + 1]) holds the threshold
      if (T[k] < RB[g] || T[k] >= RB[g + 1]) {
        g = 0;
        for (int h = 1; h < G; ++h) g += RB[h] <= T[k] ? 1 : 0;
      }
      gk[k] = g;
    }
    if (tid == 0) s_range[0] = gk[0];
    if (tid == 255) s_range[1] = gk[ITEMS - 1];
    __syncthreads();
    const int gmin = __builtin_amdgcn_readfirstlane(s_range[0]), gmax = __builtin_amdgcn_readfirstlane(s_range[1]);
    bool own_prefix = true;                      // P / Eb hold this rank's tiles
    const int64_t blk0 = a.offset + (int64_t)blockIdx.x * TILE;      // first global slot of the block
    for (int gs = gmin; gs <= gmax; ++gs) {
      if (RB[gs + 1] == RB[gs]) continue;        // a rank without weight produces no slot
      // the block's slots that fall on rank gs: [J(RB[gs]), J(RB[gs + 1])) cut to the block
      const int64_t jlo = slots_below(RB[gs], a.u, step, inv_step, total, a.K_total);
      const int64_t jhi = slots_below(RB[gs + 1], a.u, step, inv_step, total, a.K_total);
      const int64_t sb0 = jlo > blk0 ? jlo : blk0, sb1 = jhi < blk0 + TILE ? jhi : blk0 + TILE;
      if (sb1 <= sb0) continue;                  // block-uniform
      __syncthreads();                           // the previous rank's search is over (P, Eb, cumL, s_range free)
      if (gs != a.rank || !own_prefix) {         // that rank's granules: complete since its hop-2 word was seen
        const int Eg = read_granules(peer_ptr(a.agg, sPF[gs]), false);
        (void)Eg;
        shift_and_prefix(E);
        own_prefix = false;
      }
      uint64_t Tl[ITEMS];
      int tile[ITEMS];
      bool in[ITEMS];
      // rank-local thresholds; a block's slots draw from tiles near its own index when the source is its own rank: the nine
      // boundaries of the eight tiles around blockIdx.x are read together, otherwise (and outside the window) a descent
      const int wlo = (int)blockIdx.x - 4 < 0 ? 0 : ((int)blockIdx.x - 4 > nb - 8 ? (nb - 8 < 0 ? 0 : nb - 8) : (int)blockIdx.x - 4);
      uint64_t Pw[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) Pw[k] = P[wlo + k < nb ? wlo + k : nb];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        in[k] = gk[k] == gs;
        Tl[k] = in[k] ? T[k] - RB[gs] : 0;
        int tl = wlo;
        if (Tl[k] >= Pw[0] && Tl[k] < Pw[8]) {
#pragma unroll
          for (int w = 1; w < 8; ++w) tl += Pw[w] <= Tl[k] ? 1 : 0;
        } else {
          tl = 0;
#pragma unroll
          for (int sft = kPrgMaxTiles >> 1; sft >= 1; sft >>= 1) {
            const int p = tl + sft;              // P[p] = inclusive prefix of tile p - 1
            if (p <= nb - 1 && P[p] <= Tl[k]) tl = p;
          }
        }
        tile[k] = tl;
        Tl[k] = (Tl[k] - P[tl]) << (E - Eb[tl]);                  // residual in the source tile's own units
        const int64_t jslot = a.offset + i0 + k;
        if (in[k] && jslot == sb0) s_range[0] = tl;
        if (in[k] && jslot == sb1 - 1) s_range[1] = tl;
      }
      __syncthreads();
      const int tmin = __builtin_amdgcn_readfirstlane(s_range[0]), ntiles = __builtin_amdgcn_readfirstlane(s_range[1]) - tmin + 1;
      auto next_live = [&](int at) { while (at < ntiles && P[tmin + at + 1] == P[tmin + at]) ++at; return at; };
      const float* xg = peer_ptr(a.x, sPD[gs]);
      auto load_tile = [&](int t, float (&v)[ITEMS]) {
        const int64_t p0 = (int64_t)t * TILE + (int64_t)tid * ITEMS;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) v[k] = -INFINITY;
        if (ITEMS == 4 && p0 + 4 <= K) load_scoped_x4(xg + p0, v, sys);
        else {
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) if (p0 + k < K) v[k] = load_scoped(xg + p0 + k, sys);
        }
      };
      for (int idx = next_live(0); idx < ntiles;) {
        const int nidx = next_live(idx + CH);
        uint64_t qi[CH][ITEMS], sacc[CH], inc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const bool on = idx + c < ntiles;
          float nv[ITEMS];
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) nv[k] = -INFINITY;
          if (on) load_tile(tmin + idx + c, nv);
          const int es = on ? Eb[tmin + idx + c] : kTileDead;
          sacc[c] = 0;
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) { sacc[c] += on ? tile_q(nv[k], es) : 0; qi[c][k] = sacc[c]; }
          inc[c] = sacc[c];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) inc[c] = wave_scan_u64(inc[c]);
        if (lane == 63) {
#pragma unroll
          for (int c = 0; c < CH; ++c) s_wtot[c][wid] = inc[c];
        }
        __syncthreads();                         // also: every lane is done searching the previous round's cumL
        if (verify && tid < CH && idx + tid < ntiles) {
          // the log-weights just pulled must quantise to the total their owner published in the tile's granule
          const int ts = tmin + idx + tid, sh = E - Eb[ts];
          const uint64_t tot = s_wtot[tid][0] + s_wtot[tid][1] + s_wtot[tid][2] + s_wtot[tid][3];
          if ((sh < 64 ? tot >> sh : 0) != P[ts + 1] - P[ts]) verify_failed();
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          uint64_t base = inc[c] - sacc[c];
          for (int w = 0; w < wid; ++w) base += s_wtot[c][w];
#pragma unroll
          for (int k = 0; k < ITEMS; ++k) cumL[c * TILE + tid * ITEMS + k] = base + qi[c][k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
          const int kpos = tile[k] - tmin;
          if (in[k] && kpos >= idx && kpos < idx + CH) {
            const uint64_t* cm = cumL + (kpos - idx) * TILE;
            int pos = 0;                           // number of entries <= the residual, 4-ary
#pragma unroll
            for (int q = TILE >> 2; q >= 1; q >>= 2) {
              const uint64_t pa = cm[pos + q - 1], pb = cm[pos + 2 * q - 1], pc = cm[pos + 3 * q - 1];
              pos += (pa <= Tl[k] ? q : 0) + (pb <= Tl[k] ? q : 0) + (pc <= Tl[k] ? q : 0);
            }
            anc[k] = (gs * nb + tile[k]) * TILE + pos;
          }
        }
        idx = nidx;
      }
    }
  }
  // ---- children: rows of the ancestors, pulled from the rank that holds them ----
  int sl[ITEMS];
  long long dl[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int sg = anc[k] / kpad;
    sl[k] = anc[k] - sg * kpad;
    dl[k] = sPD[sg];
    if (a.ancestors && i0 + k < K) a.ancestors[i0 + k] = (int32_t)((int64_t)sg * K + sl[k]);
  }
  const bool whole = i0 + ITEMS <= K;
  const bool vec = ITEMS == 4 && whole && (a.dst_stride & 3) == 0 && (((uintptr_t)a.dst) & 15) == 0;
  uint32_t hk[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) hk[k] = row_check_init((int)a.seq, (uint32_t)((int64_t)(anc[k] / kpad) * K + sl[k]));
#pragma unroll 4
  for (int r = 0; r < a.rows; ++r) {
    const float* sr = a.src + (int64_t)r * a.src_stride;
    float v[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) v[k] = sys ? load_scoped(peer_ptr(sr, dl[k]) + sl[k], true) : sr[sl[k]];
    if (verify) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) hk[k] = row_check_mix(hk[k], v[k]);
    }
    if (vec) *(float4*)(a.dst + (int64_t)r * a.dst_stride + i0) = make_float4(v[0], v[ITEMS > 1 ? 1 : 0], v[ITEMS > 2 ? 2 : 0], v[ITEMS > 3 ? 3 : 0]);
    else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) if (i0 + k < K) a.dst[(int64_t)r * a.dst_stride + i0 + k] = v[k];
    }
  }
  if (verify && !s_dead) {                       // every pulled row against its owner's check word
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const unsigned want = load_scoped_u32(peer_ptr((const unsigned*)a.chk, dl[k]) + sl[k], sys);
      if (i0 + k < K && want != hk[k]) verify_failed();
    }
  }
}

}  // namespace gjx

// One resampling step of a sharded collection (every rank calls it with the same u): this rank's log-weights logw[p]
// and particle rows rows[p] (p = parity: the buffers of the context's DATA window the producing kernel wrote) -> the
// children of this rank's K_local output slots in rows_out f32[rows][out_stride] (caller's memory).  Alternate p from
// call to call: a rank that is one call ahead then never overwrites what a slower rank still reads.
//   partials / n_partials: the per-block {max, sumexp} pairs gjx_run_program left in ITS workspace (workspace + 256) — the
//   global LSE record then comes out in lse_out f32[4] on every rank; partials == NULL: no record.
//   ancestors (or NULL): int32[K_local], global ancestor index of every slot.
// One launch, no collective call, no host synchronisation.  GJX_EUNSUPPORTED when K_local / 1024 blocks are not
// co-resident (K_local <= 2^20 on a full MI355X).
extern "C" int gjx_peer_resample_gather(gjx_peer_ctx* c, int32_t parity, const float* partials, int32_t n_partials, double u,
                                        float* rows_out, int64_t out_stride, int32_t* ancestors, float* lse_out, void* stream) {
  if (!c || parity < 0 || parity > 1 || !rows_out || !(u >= 0.0 && u < 1.0) || (partials && n_partials <= 0))
    return gjx_fail(GJX_EINVAL, "gjx_peer_resample_gather: bad argument");
  if (!c->connected) return gjx_fail(GJX_EINVAL, "gjx_peer_resample_gather: the context is not connected (gjx_peer_ctx_connect)");
  const void* fn = (const void*)k_peer_resample_gather<4>;
  int cap = gjx_coresident_blocks(fn, 256, 0);
  if (c->share > 1) cap /= c->share;
  if (c->nt > cap || c->nt > kPrgMaxTiles)
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_peer_resample_gather: K_local / 1024 blocks are not co-resident on this device");
  // the exchange words alternate between two sets from call to call: a rank that is already one call ahead (nothing stops
  // it once it has seen everybody's hop-2 word) must not overwrite granules or LSE pairs a slower rank still reads; it
  // cannot be two calls ahead (its next hop 1 needs the slower rank's next hop-1 word)
  char* gm = c->flag + c->off_region[0] + c->r_gmm + (size_t)(c->n_gmm & 1) * 8 * (4 * (size_t)GJX_MAX_RANKS + (size_t)c->nt);
  PrgArgs a;
  memset(&a, 0, sizeof(a));
  a.x = (const float*)(c->data + c->off_lw[parity]);
  a.K = c->K; a.K_total = c->K * c->world; a.offset = (int64_t)c->rank * c->K; a.G = c->world; a.rank = c->rank; a.nt = c->nt;
  a.mode = partials ? 2 : 0; a.lse = partials; a.n_partials = n_partials; a.lse_out = lse_out;
  a.log_k_total = (float)log((double)a.K_total);
  a.u = u;
  a.src = (const float*)(c->data + c->off_rows[parity]); a.src_stride = c->K; a.rows = c->rows;
  a.dst = rows_out; a.dst_stride = out_stride; a.ancestors = ancestors;
  a.wE = (unsigned long long*)gm; a.wR = a.wE + GJX_MAX_RANKS; a.wP = a.wR + GJX_MAX_RANKS;
  a.agg = a.wP + 2 * GJX_MAX_RANKS;
  a.peer_data = c->world > 1 ? c->delta_dev : nullptr;
  a.peer_flag = c->world > 1 ? c->delta_dev + c->world : nullptr;
  a.ctrl = (unsigned*)c->flag + 8;
  a.seq = (unsigned)(c->n_gmm++);
  a.verify = c->verify ? (c->verify_fault ? 2 : 1) : 0;
  a.chk = (unsigned*)(c->data + c->off_chk[parity]);
  a.first_budget = c->world > 1 ? (1u << 24) : (1u << 16);
  void* args[] = {&a};
  const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)c->nt), dim3(256), args, 0, (hipStream_t)stream);
  if (e != hipSuccess) return gjx_fail_hip(e, "gjx_peer_resample_gather");
  return GJX_OK;
}
