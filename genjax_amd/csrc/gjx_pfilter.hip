// gjx_pfilter.hip — host side of k_pf_persistent (gjx_pfilter.inl): grid / tiles-per-block plan, the step keys of a
// filter run, and the single-GPU launch used by gjx_ssm_filter_scheme for particle counts beyond one slot per lane.
#include <math.h>
#include <string.h>

#include <vector>

#include "gjx_host.h"
#include "gjx_pfilter_host.h"

namespace gjx {

const void* pf_kernel_flat(int dx, int spl, bool move);   // gjx_pfilter_flat.hip
const void* pf_kernel_jax(int dx, int spl, bool move);    // gjx_pfilter_jax.hip

int pf_plan(int rng_mode, int dx, int dy, int64_t K_local, int n_ranks, int share, PfPlan* out, bool move) {
  if (K_local <= 0 || n_ranks < 1 || n_ranks > GJX_MAX_RANKS || dy > 32) return GJX_EUNSUPPORTED;
  const int64_t nt = (K_local + kPfHostThreads - 1) / kPfHostThreads;
  if (n_ranks > 1 && K_local % kPfHostThreads) return GJX_EUNSUPPORTED;   // sharded: whole tiles per rank
  if (nt * n_ranks > kPfHostMaxTiles) return GJX_EUNSUPPORTED;
  const size_t lds = pf_host_dyn_lds((int)(nt * n_ranks));
  const int spls[5] = {1, 2, 4, 8, 16};
  for (int i = 0; i < 5; ++i) {
    const int spl = spls[i];
    const void* fn = rng_mode == GJX_RNG_JAX32 ? pf_kernel_jax(dx, spl, move) : pf_kernel_flat(dx, spl, move);
    if (!fn) continue;
    // (static + dynamic LDS is above the 64 KB default once NT > 2048)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); }
    int cap = gjx_coresident_blocks(fn, kPfHostThreads, lds);
    if (share > 1) cap /= share;                 // ranks that share one device (dry runs): every rank's grid must be resident
    const int64_t grid = (nt + spl - 1) / spl;
    if (grid <= cap && grid * n_ranks <= kPfHostMaxTiles) {
      out->fn = fn; out->spl = spl; out->grid = (int)grid; out->lds = lds; out->nt = (int)nt;
      return GJX_OK;
    }
  }
  return GJX_EUNSUPPORTED;
}

// ---- small per-run argument arrays (step keys, comb offsets, table pointers) reach the device as KERNEL ARGUMENTS: the runtime
//      copies a launch's argument block before the launch call returns, so no host buffer has to outlive the call (an asynchronous
//      copy from pageable memory may still read its source afterwards) and the library keeps no per-thread staging state ----
namespace {
constexpr int kWordsPerLaunch = 120;
struct WordChunk {
  unsigned long long w[kWordsPerLaunch];
  unsigned long long* dst;
  int n;
};
__global__ void k_upload_words(WordChunk c) {
  const int i = (int)threadIdx.x;
  if (i < c.n) c.dst[i] = c.w[i];
}
}  // namespace

int upload_words(void* dst_dev, const void* src_host, size_t n_words, hipStream_t st) {
  const unsigned long long* src = (const unsigned long long*)src_host;
  unsigned long long* dst = (unsigned long long*)dst_dev;
  for (size_t at = 0; at < n_words; at += kWordsPerLaunch) {
    WordChunk c;
    c.n = (int)(n_words - at < (size_t)kWordsPerLaunch ? n_words - at : (size_t)kWordsPerLaunch);
    memcpy(c.w, src + at, sizeof(unsigned long long) * (size_t)c.n);
    c.dst = dst + at;
    hipLaunchKernelGGL(k_upload_words, dim3(1), dim3(128), 0, st, c);
    GJX_CHECK_LAUNCH("upload_words");
  }
  return GJX_OK;
}

void host_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int g = 0; g < 5; ++g) {
    const int* r = (g & 1) ? R + 4 : R;
    for (int j = 0; j < 4; ++j) { x0 += x1; x1 = (x1 << r[j]) | (x1 >> (32 - r[j])); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0; out[1] = x1;
}

// Key discipline of inference/pf.py: k_t = fold_in(k_{t-1}, t) (scan.py:268); (k_prop, k_res) = split(k_t); the comb
// offset of the resampling in front of step t is uniform(k_res).
void pf_step_keys(uint32_t key0, uint32_t key1, int T, std::vector<uint32_t>& keys, std::vector<double>& us) {
  std::vector<uint32_t> unused;
  pf_step_keys_res(key0, key1, T, keys, us, unused);
}

void pf_step_keys_res(uint32_t key0, uint32_t key1, int T, std::vector<uint32_t>& keys, std::vector<double>& us, std::vector<uint32_t>& res_keys) {
  keys.assign(2 * (size_t)T, 0u);
  us.assign((size_t)T, 0.0);
  res_keys.assign(2 * (size_t)T, 0u);
  uint32_t k[2] = {key0, key1};
  for (int t = 0; t < T; ++t) {
    uint32_t kt[2], kp[2], kr[2], b[2];
    host_threefry2x32(k[0], k[1], 0u, (uint32_t)t, kt);
    k[0] = kt[0]; k[1] = kt[1];
    host_threefry2x32(k[0], k[1], 0u, 0u, kp);
    host_threefry2x32(k[0], k[1], 0u, 1u, kr);
    host_threefry2x32(kr[0], kr[1], 0u, 0u, b);
    keys[2 * t] = kp[0]; keys[2 * t + 1] = kp[1];
    res_keys[2 * t] = kr[0]; res_keys[2 * t + 1] = kr[1];
    us[t] = (double)((b[0] ^ b[1]) >> 9) / 8388608.0;
  }
}

}  // namespace gjx
