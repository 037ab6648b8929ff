#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../genjax_amd/csrc/gjx_device.h"
using namespace gjx;
#define ITER 512
// A: plain threefry with uniform key, per-lane counter; ILP chains = NCH
template <int NCH>
__global__ __launch_bounds__(256) void k_hash(uint32_t* out, key2 key) {
  uint32_t c[NCH]; uint32_t acc = 0;
  for (int j = 0; j < NCH; ++j) c[j] = threadIdx.x * 16 + j + blockIdx.x * 4096;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) { key2 h = threefry2x32(key, c[j], (uint32_t)i); acc ^= h.a + h.b; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// B: fma chain with distinct regs + literal constants (like the polynomial)
template <int NCH>
__global__ __launch_bounds__(256) void k_poly(float* out, float seed) {
  float w[NCH]; float acc = 0;
  for (int j = 0; j < NCH; ++j) w[j] = seed + threadIdx.x * 0.001f + j;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float p = 3.97e-08f; float ww = w[j];
      p = fmaf(p, ww, 4.85e-07f); p = fmaf(p, ww, -4.98e-06f); p = fmaf(p, ww, -6.21e-06f); p = fmaf(p, ww, 0.000309f);
      p = fmaf(p, ww, -0.00177f); p = fmaf(p, ww, -0.0059f); p = fmaf(p, ww, 0.3488f); p = fmaf(p, ww, 2.1233f);
      w[j] = p * 0.37f; 
    }
  }
  for (int j = 0; j < NCH; ++j) acc += w[j];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// C: normal_from_bits_fast throughput
template <int NCH>
__global__ __launch_bounds__(256) void k_normal(float* out, uint32_t seed) {
  uint32_t b[NCH]; float acc = 0;
  for (int j = 0; j < NCH; ++j) b[j] = seed * 2654435761u + threadIdx.x * 40503u + j * 7919u;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) { acc += normal_from_bits_fast(b[j]); b[j] = b[j] * 1664525u + 1013904223u; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <class T, class K, class S> void run(const char* name, K kern, S seed, double units_per_thread_iter) {
  T* out; hipMalloc(&out, 256 * 4096 * sizeof(T));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {1, 2, 4}) {
    int grid = 256 * wpc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, seed);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, seed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double units = (double)grid * 4 * ITER * units_per_thread_iter / 1024.0; // per SIMD wave-units
    printf("%-10s waves/SIMD %d: %.3f ms -> %.1f cycles per wave-unit @2.1GHz\n", name, wpc, ms, ms * 1e-3 * 2.1e9 / units);
  }
}
int main() {
  run<uint32_t>("hash ilp1", k_hash<1>, key2{1, 2}, 1); run<uint32_t>("hash ilp2", k_hash<2>, key2{1, 2}, 2); run<uint32_t>("hash ilp4", k_hash<4>, key2{1, 2}, 4);
  run<float>("poly ilp1", k_poly<1>, 0.5f, 1); run<float>("poly ilp4", k_poly<4>, 0.5f, 4);
  run<float>("normal ilp1", k_normal<1>, 3u, 1); run<float>("normal ilp4", k_normal<4>, 3u, 4);
  return 0;
}
