// gmmv.hip — standalone experiment harness (not product code): variants of the fused mixture kernel and
// Threefry issue-rate microbenchmarks.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gmmv gmmv.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>

#include "../../genjax_amd/csrc/gjx_device.h"

using namespace gjx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------
// 1. hash issue-rate microbenchmarks: NH hashes per lane, ILP independent chains interleaved by hand
// ------------------------------------------------------------------------------------------
template <int ILP>
__global__ __launch_bounds__(256) void k_hash(uint32_t* out, key2 key, int nh) {
  uint32_t c0 = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (int h = 0; h < nh; h += ILP) {
    key2 r[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) r[j] = threefry2x32(key, c0, (uint32_t)(h + j));
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc ^= r[j].a + r[j].b;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------
// 2. fused mixture kernel variants
// ------------------------------------------------------------------------------------------
struct GArgs {
  const float* tab;   // raw tables: logits[C], mu[C][D], sig[C][D], r[D], y[D]
  const float* aux;   // prepared LDS image (PRO = 1)
  int C;
  key2 key;
  int64_t K;
  float* choices;
  float* score;
  float* logw;
  unsigned long long* partials;
  unsigned long long* tl;   // timeline: 3 realtime stamps per wave (or NULL)
};

template <int PPT>
struct VS;
template <> struct VS<1> { static GJX_DEV void st(float* p, const float (&v)[1]) { *p = v[0]; } };
template <> struct VS<2> { static GJX_DEV void st(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); } };
template <> struct VS<4> { static GJX_DEV void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); } };

// store flavours: 0 plain, 1 nontemporal, 2 sc1 (write-through, agent scope), 3 sc0 sc1
template <int FL>
GJX_DEV void st4(float* p, const float (&v)[4]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 x = {v[0], v[1], v[2], v[3]};
  if (FL == 1) __builtin_nontemporal_store(x, reinterpret_cast<f4*>(p));
  else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
  else if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
  else *reinterpret_cast<f4*>(p) = x;
}
template <int FL, int PPT>
GJX_DEV void stv(float* p, const float (&v)[PPT]) {
  if constexpr (PPT == 4) st4<FL>(p, v); else VS<PPT>::st(p, v);
}

// Threefry with the key schedule held in VGPRs (kv[0..2] = ks0, ks1, ks2): SGPR operands make v_add_u32 a half-rate op
GJX_DEV key2 threefry_kv(const uint32_t (&kv)[3], uint32_t c0, uint32_t c1) {
  uint32_t x0 = c0 + kv[0], x1 = c1 + kv[1];
#define GJX_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  GJX_R(13) GJX_R(15) GJX_R(26) GJX_R(6)
  x0 += kv[1]; x1 += kv[2]; x1 += 1u;
  GJX_R(17) GJX_R(29) GJX_R(16) GJX_R(24)
  x0 += kv[2]; x1 += kv[0]; x1 += 2u;
  GJX_R(13) GJX_R(15) GJX_R(26) GJX_R(6)
  x0 += kv[0]; x1 += kv[1]; x1 += 3u;
  GJX_R(17) GJX_R(29) GJX_R(16) GJX_R(24)
  x0 += kv[1]; x1 += kv[2]; x1 += 4u;
  GJX_R(13) GJX_R(15) GJX_R(26) GJX_R(6)
  x0 += kv[2]; x1 += kv[0]; x1 += 5u;
#undef GJX_R
  return key2{x0, x1};
}

// LDS image layout (floats): mu[C][DS] sig[C][DS] zlp[C] cdf[C] y[D] rr[D] misc[8]
template <int D>
__host__ __device__ constexpr int aux_floats(int C) { return 2 * C * (D + 4) + 2 * C + 2 * D + 8; }

template <int D>
__global__ void k_prepare(const float* tab, int C, float* aux) {
  // single block; computes exactly what the old prologue computed
  constexpr int DS = D + 4;
  const float* logits = tab; const float* mu = tab + C; const float* sig = mu + C * D; const float* r = sig + C * D; const float* y = r + D;
  float* s_mu = aux; float* s_sig = s_mu + C * DS; float* s_zlp = s_sig + C * DS; float* s_cdf = s_zlp + C; float* s_y = s_cdf + C; float* s_rr = s_y + D; float* s_misc = s_rr + D;
  if (threadIdx.x == 0) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[c]);
    float se = 0.0f;
    for (int c = 0; c < C; ++c) { se += fast_exp(logits[c] - mx); s_cdf[c] = se; }
    const float lse = mx + fast_log(se);
    for (int c = 0; c < C; ++c) {
      float sl = 0.0f;
      for (int d = 0; d < D; ++d) { s_mu[c * DS + d] = mu[c * D + d]; s_sig[c * DS + d] = sig[c * D + d]; sl += fast_log(sig[c * D + d]); }
      for (int d = D; d < DS; ++d) { s_mu[c * DS + d] = 0; s_sig[c * DS + d] = 0; }
      s_zlp[c] = (logits[c] - lse) - sl - (float)D * kHalfLog2Pi;
    }
    float sl = 0.0f;
    for (int d = 0; d < D; ++d) { s_y[d] = y[d]; s_rr[d] = fast_rcp(r[d]); sl += fast_log(r[d]); }
    s_misc[0] = -sl - (float)D * kHalfLog2Pi;
    for (int j = 1; j < 8; ++j) s_misc[j] = 0;
  }
}

// window of 32 stream bits starting at bit 23*c of the word stream w[] (compile-time c after unrolling)
#define FIELD(w, c) ((((23 * (c)) & 31) == 0) ? (w)[(23 * (c)) >> 5] : __builtin_amdgcn_alignbit((w)[((23 * (c)) >> 5) + 1], (w)[(23 * (c)) >> 5], (23 * (c)) & 31))

// NH9 = 1: old layout (one hash per pair of draws); 0: bit-packed (6 hashes for 16 draws)
template <int D, int PPT, int THREADS, int NH9, int PRO, int MINW, int CAT = 0, int OPT = 0>
__global__ __launch_bounds__(THREADS, (MINW % 100)) void k_gmm(GArgs a) {
  static_assert(D == 16, "harness is for D = 16");
  constexpr int DS = D + 4;
  const unsigned long long tl0 = a.tl ? __builtin_amdgcn_s_memrealtime() : 0ull;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C;
  float* s_mu = smem; float* s_sig = s_mu + C * DS; float* s_zlp = s_sig + C * DS; float* s_cdf = s_zlp + C;
  float* s_y = s_cdf + C; float* s_rr = s_y + D; float* s_misc = s_rr + D;
  if (PRO) {
    const int n = aux_floats<D>(C);
    for (int t = threadIdx.x; t < n; t += THREADS) smem[t] = a.aux[t];
    __syncthreads();
  } else {
    float* s_logit = s_misc + 8; float* s_lsig = s_logit + C; float* s_lr = s_lsig + C * DS;
    const float* logits = a.tab; const float* mu = a.tab + C; const float* sig = mu + C * D; const float* r = sig + C * D; const float* y = r + D;
    for (int t = threadIdx.x; t < C * D; t += THREADS) {
      const int c = t / D, d = t % D;
      const float sg = sig[t];
      s_mu[c * DS + d] = mu[t]; s_sig[c * DS + d] = sg; s_lsig[c * DS + d] = fast_log(sg);
    }
    for (int t = threadIdx.x; t < D; t += THREADS) { s_y[t] = y[t]; s_rr[t] = fast_rcp(r[t]); s_lr[t] = fast_log(r[t]); }
    for (int t = threadIdx.x; t < C; t += THREADS) s_logit[t] = logits[t];
    __syncthreads();
    if (threadIdx.x < 64) {
      float mx = -INFINITY;
      for (int c = threadIdx.x; c < C; c += 64) mx = fmaxf(mx, s_logit[c]);
      mx = wave_max(mx);
      float se = 0.0f;
      for (int c = threadIdx.x; c < C; c += 64) se += fast_exp(s_logit[c] - mx);
      se = wave_sum(se);
      const float lse = mx + fast_log(se);
      for (int c = threadIdx.x; c < C; c += 64) {
        float sl = 0.0f;
        for (int d = 0; d < D; ++d) sl += s_lsig[c * DS + d];
        s_zlp[c] = (s_logit[c] - lse) - sl - (float)D * kHalfLog2Pi;
      }
      if (threadIdx.x == 63) { float sl = 0.0f; for (int d = 0; d < D; ++d) sl += s_lr[d]; s_misc[0] = -sl - (float)D * kHalfLog2Pi; }
      if (threadIdx.x == 0) { float run = 0.0f; for (int c = 0; c < C; ++c) { run += fast_exp(s_logit[c] - mx); s_cdf[c] = run; } }
    }
    __syncthreads();
  }
  const unsigned long long tl1 = a.tl ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int64_t K = a.K;
  const int64_t tile = (int64_t)THREADS * PPT;
  const int64_t ntiles = (K + tile - 1) / tile;
  const key2 fkey = a.key;
  constexpr int FL = (OPT >> 4) & 3;
  uint32_t kv[3] = {fkey.a, fkey.b, fkey.a ^ fkey.b ^ 0x1BD11BDAu};
  if (OPT & 8) { asm volatile("v_mov_b32 %0, %0" : "+v"(kv[0])); asm volatile("v_mov_b32 %0, %0" : "+v"(kv[1])); asm volatile("v_mov_b32 %0, %0" : "+v"(kv[2])); }
  float cdf_s[8];
  if (CAT == 1) {
    const float* cdfp = a.aux + 2 * C * DS + C;
#pragma unroll
    for (int c = 0; c < 8; ++c) cdf_s[c] = cdfp[c];
  }
  float tmax = -INFINITY, tsum = 0.0f;
  for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const int64_t i0 = tix * tile + (int64_t)threadIdx.x * PPT;
    uint32_t c0[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) c0[p] = (uint32_t)(i0 + p);
    int z[PPT];
    float zf[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const key2 h = (OPT & 8) ? threefry_kv(kv, c0[p], (1u << GJX_FLAT_SITE_SHIFT)) : threefry2x32(fkey, c0[p], (1u << GJX_FLAT_SITE_SHIFT));
      const float target = (__uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, h.a, 9)) - 1.0f) * s_cdf[C - 1];
      int best = 0;
      if (CAT == 1) {
        // count of c in [0, 7) with cdf[c] <= target: sign bit of (target - cdf[c]) is 0  <=>  7 - sum of sign bits
        const float target1 = (__uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, h.a, 9)) - 1.0f) * cdf_s[7];
        unsigned neg = 0;
#pragma unroll
        for (int c = 0; c < 7; ++c) neg += __float_as_uint(target1 - cdf_s[c]) >> 31;
        best = 7 - (int)neg;
      } else {
        for (int c = 0; c < C - 1; ++c) best += (s_cdf[c] > target) ? 0 : 1;
      }
      z[p] = best; zf[p] = (float)best;
    }
    float* ch = a.choices;
    VS<PPT>::st(ch + i0, zf);
    float qx[PPT], qy[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) { qx[p] = 0.0f; qy[p] = 0.0f; }
    if (NH9) {
#pragma unroll 2
      for (int d0 = 0; d0 < D; d0 += 2) {
        float xa[PPT], xb[PPT];
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
          const key2 h = threefry2x32(fkey, c0[p], (2u << GJX_FLAT_SITE_SHIFT) | (uint32_t)(d0 >> 1));
          const int zo = z[p] * DS + d0;
          float n0, n1;
          box_muller(h.a, h.b, n0, n1);
          { const float x = fmaf(s_sig[zo], n0, s_mu[zo]); qx[p] = fmaf(n0, n0, qx[p]); const float zy = (s_y[d0] - x) * s_rr[d0]; qy[p] = fmaf(zy, zy, qy[p]); xa[p] = x; }
          { const float x = fmaf(s_sig[zo + 1], n1, s_mu[zo + 1]); qx[p] = fmaf(n1, n1, qx[p]); const float zy = (s_y[d0 + 1] - x) * s_rr[d0 + 1]; qy[p] = fmaf(zy, zy, qy[p]); xb[p] = x; }
        }
        float* r0 = ch + (int64_t)(1 + d0) * K + i0;
        VS<PPT>::st(r0, xa);
        VS<PPT>::st(r0 + K, xb);
      }
    } else {
      // bit-packed: 16 draws x 23 bits from 6 hashes (12 words); fully unrolled so every index is a constant
      uint32_t w[PPT][12];
#pragma unroll
      for (int k = 0; k < 8; ++k) {        // pair k = draws 2k, 2k+1
        float xa[PPT], xb[PPT];
        const int d0 = 2 * k;
        if (OPT & 4) { if (k < 2) __builtin_amdgcn_s_setprio(3); else if (k < 4) __builtin_amdgcn_s_setprio(2); else if (k < 6) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        float2 mu2[PPT], sg2[PPT];
        float2 y2, rr2;
        if (MINW >= 200) {                 // LDS operands of this pair requested before the hash work that hides their latency
          y2 = *reinterpret_cast<const float2*>(&s_y[d0]);
          rr2 = *reinterpret_cast<const float2*>(&s_rr[d0]);
#pragma unroll
          for (int p = 0; p < PPT; ++p) {
            mu2[p] = *reinterpret_cast<const float2*>(&s_mu[z[p] * DS + d0]);
            sg2[p] = *reinterpret_cast<const float2*>(&s_sig[z[p] * DS + d0]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
          const int need = (((23 * (2 * k + 1)) >> 5) + 1) >> 1;      // highest hash index needed
          const int have = k == 0 ? -1 : ((((23 * (2 * (k - 1) + 1)) >> 5) + 1) >> 1);
#pragma unroll
          for (int h = 0; h < 6; ++h) if (h > have && h <= need) {
            const key2 hh = (OPT & 8) ? threefry_kv(kv, c0[p], (2u << GJX_FLAT_SITE_SHIFT) | (uint32_t)h) : threefry2x32(fkey, c0[p], (2u << GJX_FLAT_SITE_SHIFT) | (uint32_t)h);
            w[p][2 * h] = hh.a; w[p][2 * h + 1] = hh.b;
          }
        }
        if (MINW >= 300) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
          const uint32_t wa = FIELD(w[p], 2 * k), wb = FIELD(w[p], 2 * k + 1);
          float n0, n1;
          box_muller(wa, wb, n0, n1);
          if (MINW < 200) {
            const int zo = z[p] * DS + d0;
            mu2[p] = make_float2(s_mu[zo], s_mu[zo + 1]); sg2[p] = make_float2(s_sig[zo], s_sig[zo + 1]);
            y2 = make_float2(s_y[d0], s_y[d0 + 1]); rr2 = make_float2(s_rr[d0], s_rr[d0 + 1]);
          }
          { const float x = fmaf(sg2[p].x, n0, mu2[p].x); qx[p] = fmaf(n0, n0, qx[p]); const float zy = (y2.x - x) * rr2.x; qy[p] = fmaf(zy, zy, qy[p]); xa[p] = x; }
          { const float x = fmaf(sg2[p].y, n1, mu2[p].y); qx[p] = fmaf(n1, n1, qx[p]); const float zy = (y2.y - x) * rr2.y; qy[p] = fmaf(zy, zy, qy[p]); xb[p] = x; }
        }
        float* r0 = ch + (int64_t)(1 + 2 * k) * K + i0;
        stv<FL, PPT>(r0, xa);
        stv<FL, PPT>(r0 + K, xb);
        if (MINW >= 100) __builtin_amdgcn_sched_barrier(0);
      }
    }
    float sc[PPT], lw[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      lw[p] = fmaf(-0.5f, qy[p], s_misc[0]);
      sc[p] = fmaf(-0.5f, qx[p], s_zlp[z[p]]) + lw[p];
    }
    stv<FL, PPT>(a.score + i0, sc);
    stv<FL, PPT>(a.logw + i0, lw);
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float nm = fmaxf(tmax, lw[p]);
      if (nm > -INFINITY) tsum = tsum * fast_exp(tmax - nm) + fast_exp(lw[p] - nm);
      tmax = nm;
    }
  }
  {
    constexpr int NW = THREADS / 64;
    __shared__ float red[2 * NW];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float wm = wave_max(tmax);
    const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
    if (lane == 0) { red[wid] = wm; red[NW + wid] = ws; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float bm = red[0];
      for (int w2 = 1; w2 < NW; ++w2) bm = fmaxf(bm, red[w2]);
      float bsum = 0.0f;
      for (int w2 = 0; w2 < NW; ++w2) bsum += bm > -INFINITY ? red[NW + w2] * fast_exp(red[w2] - bm) : 0.0f;
      a.partials[blockIdx.x] = pack_f2(bm, bsum);
    }
  }
  if (a.tl && (threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    a.tl[3 * w] = tl0; a.tl[3 * w + 1] = tl1; a.tl[3 * w + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

template <class F>
float time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch(i);
  CK(hipDeviceSynchronize());
  // per-launch timing with events around each launch; report the median and min
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  printf("   median %.2f us  min %.2f us", t[t.size() / 2], t[0]);
  // back-to-back throughput
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  back-to-back %.2f us/launch\n", ms * 1e3f / reps);
  return t[t.size() / 2];
}

struct Bufs { float *tab, *aux, *choices, *score, *logw; unsigned long long* partials; uint32_t* out; unsigned long long* tl; };

template <int PPT, int THREADS, int NH9, int PRO, int MINW, int CAT = 0, int OPT = 0>
void run_gmm(const char* name, const Bufs& b, int64_t K, int maxgrid, double* check) {
  constexpr int D = 16; const int C = 8;
  GArgs a; a.tab = b.tab; a.aux = b.aux; a.C = C; a.K = K; a.choices = b.choices; a.score = b.score; a.logw = b.logw; a.partials = b.partials; a.tl = nullptr;
  const int64_t tile = (int64_t)THREADS * PPT;
  int64_t nt = (K + tile - 1) / tile;
  const int grid = (int)(nt < maxgrid ? nt : maxgrid);
  const size_t lds = sizeof(float) * (size_t)(aux_floats<D>(C) + C + C * (D + 4) + D + 16);
  printf("%-44s grid %5d x %3d ppt %d:", name, grid, THREADS, PPT);
  auto launch = [&](int i) { a.key = key2{0u, (uint32_t)(1 + i)}; hipLaunchKernelGGL((k_gmm<D, PPT, THREADS, NH9, PRO, MINW, CAT, OPT>), dim3(grid), dim3(THREADS), lds, 0, a); };
  time_us(launch, 200);
  // checksum: mean of logw and mean of x row 3 for key (0,1)
  launch(0);
  CK(hipDeviceSynchronize());
  std::vector<float> lw(K), x3(K);
  CK(hipMemcpy(lw.data(), b.logw, K * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x3.data(), b.choices + 4 * K, K * 4, hipMemcpyDeviceToHost));
  double s = 0, s2 = 0, m = -1e30;
  for (int64_t i = 0; i < K; ++i) { m = lw[i] > m ? lw[i] : m; }
  for (int64_t i = 0; i < K; ++i) { s += exp(lw[i] - m); s2 += x3[i]; }
  printf("      logML %.5f  mean x3 %.5f\n", m + log(s) - log((double)K), s2 / K);
  if (check) *check = m + log(s) - log((double)K);
  // timeline of one launch (realtime counter, 100 MHz)
  a.tl = b.tl;
  launch(0); CK(hipDeviceSynchronize());
  launch(0); CK(hipDeviceSynchronize());
  const int nw = grid * (THREADS / 64);
  std::vector<unsigned long long> tl(3 * nw);
  CK(hipMemcpy(tl.data(), b.tl, tl.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0; double pro = 0, dur = 0;
  for (int w = 0; w < nw; ++w) { s0 = std::min(s0, tl[3 * w]); s1 = std::max(s1, tl[3 * w]); e0 = std::min(e0, tl[3 * w + 2]); e1 = std::max(e1, tl[3 * w + 2]); pro += (double)(tl[3 * w + 1] - tl[3 * w]); dur += (double)(tl[3 * w + 2] - tl[3 * w + 1]); }
  printf("      timeline us: first start 0, last start %.2f, first end %.2f, last end %.2f | mean prologue %.2f, mean main+epilogue %.2f\n", (s1 - s0) * 0.01, (e0 - s0) * 0.01, (e1 - s0) * 0.01, pro / nw * 0.01, dur / nw * 0.01);
  a.tl = nullptr;
}

int main(int argc, char** argv) {
  const int64_t K = 1 << 20;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  Bufs b;
  const int C = 8, D = 16;
  std::vector<float> tab(C + 2 * C * D + 2 * D);
  srand(1);
  for (int c = 0; c < C; ++c) tab[c] = (rand() / (float)RAND_MAX) * 2 - 1;
  for (int t = 0; t < C * D; ++t) { tab[C + t] = (rand() / (float)RAND_MAX) * 2 - 1; tab[C + C * D + t] = 1.0f; }
  for (int d = 0; d < D; ++d) { tab[C + 2 * C * D + d] = 4.0f; tab[C + 2 * C * D + D + d] = (rand() / (float)RAND_MAX) * 4 - 2; }
  CK(hipMalloc(&b.tab, tab.size() * 4)); CK(hipMemcpy(b.tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&b.aux, 4096 * 4));
  CK(hipMalloc(&b.choices, (size_t)17 * K * 4)); CK(hipMalloc(&b.score, K * 4)); CK(hipMalloc(&b.logw, K * 4));
  CK(hipMalloc(&b.partials, 65536 * 8)); CK(hipMalloc(&b.tl, 3 * 8 * 65536)); CK(hipMalloc(&b.out, 2048 * 256 * 8 * 4));
  hipLaunchKernelGGL((k_prepare<16>), dim3(1), dim3(64), 0, 0, b.tab, C, b.aux);
  CK(hipDeviceSynchronize());

  if (argc > 1 && !strcmp(argv[1], "hash")) {
    // hashes per lane 64; grid = 256 CUs * waves-per-SIMD blocks
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nh = 256;
    for (int ilp : {1, 2, 4}) for (int wps : {1, 2, 4, 8}) {
      const int grid = 256 * wps;
      auto L = [&]() {
        if (ilp == 1) hipLaunchKernelGGL((k_hash<1>), dim3(grid), dim3(256), 0, 0, b.out, key2{1u, 2u}, nh);
        else if (ilp == 2) hipLaunchKernelGGL((k_hash<2>), dim3(grid), dim3(256), 0, 0, b.out, key2{1u, 2u}, nh);
        else hipLaunchKernelGGL((k_hash<4>), dim3(grid), dim3(256), 0, 0, b.out, key2{1u, 2u}, nh);
      };
      L(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) L(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / 10;
      // wave-hashes per SIMD = grid*4 waves * nh / 1024 SIMDs
      const double per_simd = (double)grid * 4 * nh / 1024.0;
      printf("hash ilp %d waves/SIMD %d: %.2f us -> %.1f cycles/wave-hash @2.4GHz (%.1f @2.1)\n", ilp, wps, us, us * 1e-6 * 2.4e9 / per_simd, us * 1e-6 * 2.1e9 / per_simd);
    }
    return 0;
  }
  double ref;
  run_gmm<4, 256, 1, 0, 1>("base: 9 hashes, computed prologue", b, K, 2048, &ref);
  run_gmm<4, 256, 0, 1, 101, 1>("7h copy ppt4 sb CAT1", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 4>("7h ppt4 sb CAT1 setprio", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 8>("7h ppt4 sb CAT1 vgprkeys", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 12>("7h ppt4 sb CAT1 setprio+vgprkeys", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 16>("7h ppt4 sb CAT1 st nt", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 32>("7h ppt4 sb CAT1 st sc1", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 48>("7h ppt4 sb CAT1 st sc0sc1", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 12+16>("7h ppt4 sb CAT1 setprio+vgprkeys+nt", b, K, 2048, nullptr);
  run_gmm<4, 256, 0, 1, 101, 1, 12+32>("7h ppt4 sb CAT1 setprio+vgprkeys+sc1", b, K, 2048, nullptr);
  run_gmm<2, 256, 0, 1, 101, 1, 12>("7h ppt2 sb CAT1 setprio+vgprkeys 2048blk", b, K, 2048, nullptr);
  return 0;
}
