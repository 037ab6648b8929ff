// ub_mfma.hip — does VALU work hide beside v_mfma_f32_16x16x4_f32 on gfx950?  Shader cycles (s_memtime) per MFMA for a
// stream of MFMAs on two accumulators with F filler VALU instructions behind each, at 1 / 2 / 4 waves per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o ub_mfma ub_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

// KIND 0: v_fma_f32 fillers, 1: v_exp_f32 fillers; F fillers behind each MFMA (independent registers, round robin over 8);
// M = 0: no MFMAs.  Inline asm: the instruction order is exactly the source order (fillers never touch MFMA registers; the two
// accumulators alternate as in k_hmc_logreg_mfma, where hipcc emits the same pattern without wait states).
template <int F, int KIND, int M>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed) {
  v4f c0 = {seed, 0.f, 1.f, 2.f}, c1 = {1.f, seed, 0.f, 3.f};
  float a = seed + (float)(threadIdx.x & 15), b = seed * 0.5f;
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = seed * (float)(j + 1) * 1e-3f;
  const float m = 0.999f, q = 0.25f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (M) {
        if (u & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int j = 0; j < F; ++j) {
        const int r = (u * F + j) % 8;
        if (KIND) asm volatile("v_exp_f32 %0, %0" : "+v"(f[r]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(m), "v"(q));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
  for (int j = 0; j < 8; ++j) s += f[j];
  if (s == 12345.678f) out[4096] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// the HMC mix (per 8 MFMAs: 16 v_exp + 16 v_fma), issued in three orders.  ORDER 0: (MFMA, 2 exp, 2 fma) x 8;
// 1: 8 MFMAs, then 16 exp, then 16 fma; 2: (4 MFMAs, 8 exp, 8 fma) x 2
template <int ORDER>
__global__ __launch_bounds__(1024) void kmix(unsigned long long* out, float seed) {
  v4f c0 = {seed, 0.f, 1.f, 2.f}, c1 = {1.f, seed, 0.f, 3.f};
  float a = seed + (float)(threadIdx.x & 15), b = seed * 0.5f;
  float f[8], g[8];
  for (int j = 0; j < 8; ++j) { f[j] = seed * (float)(j + 1) * 1e-3f; g[j] = f[j] + 1.0f; }
  const float m = 0.999f, q = 0.25f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define MF(u) { if ((u) & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b)); \
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b)); }
#define EX(r) asm volatile("v_exp_f32 %0, %0" : "+v"(f[(r) % 8]));
#define FM(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[(r) % 8]) : "v"(m), "v"(q));
  for (int i = 0; i < ITER; ++i) {
    if (ORDER == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { MF(u) EX(2 * u) FM(2 * u) EX(2 * u + 1) FM(2 * u + 1) }
    } else if (ORDER == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) MF(u)
#pragma unroll
      for (int u = 0; u < 16; ++u) EX(u)
#pragma unroll
      for (int u = 0; u < 16; ++u) FM(u)
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 4; ++u) MF(u)
#pragma unroll
        for (int u = 0; u < 8; ++u) EX(u)
#pragma unroll
        for (int u = 0; u < 8; ++u) FM(u)
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
  for (int j = 0; j < 8; ++j) s += f[j] + g[j];
  if (s == 12345.678f) out[4096] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ORDER>
static void runmix(const char* name, unsigned long long* d) {
  for (int threads : {256, 512, 1024}) {
    CK(hipMemset(d, 0, 8 * 8192));
    hipLaunchKernelGGL((kmix<ORDER>), dim3(8), dim3(threads), 0, 0, d, 1.25f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(8 * 16);
    CK(hipMemcpy(h.data(), d, 8 * 16 * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int b = 0; b < 8; ++b) for (int w = 0; w < threads / 64; ++w) mx = std::max(mx, (double)h[b * 16 + w]);
    printf("%-34s waves/SIMD %d: %7.1f cycles of SIMD time per 8 MFMAs + 16 v_exp + 16 v_fma (256 + 131 + 41 if they just add up)\n", name, threads / 256,
           mx / ITER / (threads / 256));
  }
}

// LDS reads beside the MFMAs: per 8 MFMAs, R reads of kind W (0: ds_read_b32, 1: ds_read2_b32, 2: ds_read_b128); M = 0: reads alone
template <int R, int W, int M>
__global__ __launch_bounds__(1024) void klds(unsigned long long* out, float seed) {
  __shared__ float sm[16384];
  for (int t = threadIdx.x; t < 16384; t += blockDim.x) sm[t] = seed * (float)t;
  __syncthreads();
  v4f c0 = {seed, 0.f, 1.f, 2.f}, c1 = {1.f, seed, 0.f, 3.f};
  float a = seed + (float)(threadIdx.x & 15), b = seed * 0.5f;
  const unsigned addr = (unsigned)(uintptr_t)(sm) + (threadIdx.x & 63) * (W == 2 ? 16 : 4);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (M) {
        if (u & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
      }
      if (u < R) {
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (W == 0) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(v[0]) : "v"(addr));
        else if (W == 1) { float2 w; asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:20" : "=v"(w) : "v"(addr)); v[0] = w.x; v[1] = w.y; }
        else asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3] + acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 12345.678f) out[4096] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int R, int W, int M>
static void runlds(const char* name, unsigned long long* d) {
  for (int threads : {256, 512, 1024}) {
    CK(hipMemset(d, 0, 8 * 8192));
    hipLaunchKernelGGL((klds<R, W, M>), dim3(8), dim3(threads), 0, 0, d, 1.25f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(8 * 16);
    CK(hipMemcpy(h.data(), d, 8 * 16 * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int b = 0; b < 8; ++b) for (int w = 0; w < threads / 64; ++w) mx = std::max(mx, (double)h[b * 16 + w]);
    printf("%-34s waves/SIMD %d: %7.1f cycles of SIMD time per 8 MFMA slots (256 for the MFMAs)\n", name, threads / 256, mx / ITER / (threads / 256));
  }
}

// the gradient trip of k_hmc_logreg_mfma2 without its LDS reads: 16 forward MFMAs on 4 accumulators, 16 exp + 16 add + 16 rcp on
// their results, 16 backward MFMAs that take the sigmoids as B.  DROP bit 0: no exp, bit 1: no add, bit 2: no rcp
template <int DROP>
__global__ __launch_bounds__(1024) void khmc(unsigned long long* out, float seed) {
  v4f g[4], beta[2], x0, x1, a0, a1;
  for (int j = 0; j < 4; ++j) g[j] = v4f{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 2; ++c) beta[c] = v4f{seed * 1e-3f, -seed * 1e-3f, 2e-3f, -1e-3f} * (float)(c + 1);
  x0 = v4f{seed, 1.f, -1.f, 0.5f}; x1 = x0 * 0.5f; a0 = x0 * 0.25f; a1 = x0 * 0.125f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITER; ++i) {
    v4f s0[2], s1[2];
    for (int c = 0; c < 2; ++c) { s0[c] = v4f{0.f, 0.f, 0.f, 0.f}; s1[c] = s0[c]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        s0[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k], beta[c][k], s0[c], 0, 0, 0);
        s1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[k], beta[c][k], s1[c], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (!(DROP & 1)) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[c][r] = __builtin_amdgcn_exp2f(s0[c][r]); s1[c][r] = __builtin_amdgcn_exp2f(s1[c][r]); }
    }
    if (!(DROP & 2)) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[c][r] = 1.0f + s0[c][r]; s1[c][r] = 1.0f + s1[c][r]; }
    }
    if (!(DROP & 4)) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[c][r] = __builtin_amdgcn_rcpf(s0[c][r]); s1[c][r] = __builtin_amdgcn_rcpf(s1[c][r]); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        g[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], s0[c][r], g[c], 0, 0, 0);
        g[2 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], s1[c][r], g[2 + c], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += g[j][0] + g[j][1] + g[j][2] + g[j][3];
  if (s == 12345.678f) out[4096] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int DROP>
static void runhmc(const char* name, unsigned long long* d) {
  for (int threads : {256, 512, 1024}) {
    CK(hipMemset(d, 0, 8 * 8192));
    hipLaunchKernelGGL((khmc<DROP>), dim3(8), dim3(threads), 0, 0, d, 1.25f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(8 * 16);
    CK(hipMemcpy(h.data(), d, 8 * 16 * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int b = 0; b < 8; ++b) for (int w = 0; w < threads / 64; ++w) mx = std::max(mx, (double)h[b * 16 + w]);
    printf("%-34s waves/SIMD %d: %7.1f cycles of SIMD time per trip (32 MFMAs = 1024)\n", name, threads / 256, mx / ITER / (threads / 256));
  }
}

template <int F, int KIND, int M>
static void run(const char* name, unsigned long long* d) {
  for (int threads : {256, 512, 1024}) {
    CK(hipMemset(d, 0, 8 * 8192));
    hipLaunchKernelGGL((k<F, KIND, M>), dim3(8), dim3(threads), 0, 0, d, 1.25f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(8 * 16);
    CK(hipMemcpy(h.data(), d, 8 * 16 * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int b = 0; b < 8; ++b) for (int w = 0; w < threads / 64; ++w) mx = std::max(mx, (double)h[b * 16 + w]);
    const double per = mx / (ITER * 8.0);       // per MFMA slot (8 per iteration) of ONE wave
    printf("%-34s waves/SIMD %d: %7.1f cycles per slot per wave, %6.1f cycles of SIMD time per slot\n", name, threads / 256, per, per / (threads / 256));
  }
}

int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 8 * 8192));
  run<0, 0, 1>("mfma only", d);
  run<1, 0, 1>("mfma + 1 v_fma", d);
  run<2, 0, 1>("mfma + 2 v_fma", d);
  run<4, 0, 1>("mfma + 4 v_fma", d);
  run<8, 0, 1>("mfma + 8 v_fma", d);
  run<1, 1, 1>("mfma + 1 v_exp", d);
  run<2, 1, 1>("mfma + 2 v_exp", d);
  run<4, 1, 1>("mfma + 4 v_exp", d);
  run<4, 0, 0>("4 v_fma alone", d);
  run<8, 0, 0>("8 v_fma alone", d);
  run<2, 1, 0>("2 v_exp alone", d);
  run<4, 1, 0>("4 v_exp alone", d);
  runmix<0>("mix interleaved", d);
  runmix<1>("mix batched 8", d);
  runmix<2>("mix batched 4", d);
  runlds<8, 0, 1>("8 mfma + 8 ds_read_b32 (+wait+4 add)", d);
  runlds<4, 1, 1>("8 mfma + 4 ds_read2_b32", d);
  runlds<4, 2, 1>("8 mfma + 4 ds_read_b128", d);
  runlds<8, 2, 1>("8 mfma + 8 ds_read_b128", d);
  runlds<8, 2, 0>("8 ds_read_b128 alone", d);
  runlds<8, 0, 0>("8 ds_read_b32 alone", d);
  runhmc<0>("hmc trip: 32 mfma + 16 exp,add,rcp", d);
  runhmc<4>("hmc trip without the rcp", d);
  runhmc<1>("hmc trip without the exp", d);
  runhmc<2>("hmc trip without the add", d);
  runhmc<7>("hmc trip: the 32 mfma only", d);
  return 0;
}
