#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
// A: acc[j] = fma(x[j], y[j], acc[j]) : 3 distinct VGPR sources
__global__ __launch_bounds__(256) void k_fma3(float* out, float seed) {
  float x[16], y[16], acc[16];
  for (int j = 0; j < 16; ++j) { x[j] = seed + threadIdx.x * 0.001f + j; y[j] = seed * 0.5f + j * 0.01f; acc[j] = j; }
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "v"(x[j]), "v"(y[j]));
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// B: acc[j] = fma(x[j], yS, acc[j]) : y in SGPR
__global__ __launch_bounds__(256) void k_fma2s(float* out, float seed) {
  float x[16], acc[16];
  for (int j = 0; j < 16; ++j) { x[j] = seed + threadIdx.x * 0.001f + j; acc[j] = j; }
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "s"(seed), "v"(x[j]));
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// C: one accumulator chain (dependent), distinct sources: s = fma(x[j], y[j], s)
__global__ __launch_bounds__(256) void k_fma_chain(float* out, float seed) {
  float x[16], y[16]; float s0 = 0, s1 = 1;
  for (int j = 0; j < 16; ++j) { x[j] = seed + threadIdx.x * 0.001f + j; y[j] = seed * 0.5f + j * 0.01f; }
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int j = 0; j < 16; j += 2) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s0) : "v"(x[j]), "v"(y[j])); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s1) : "v"(x[j+1]), "v"(y[j+1])); }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s0 + s1;
}
template <class K> void run(const char* name, K kern, double instr_per_iter) {
  float* out; hipMalloc(&out, 256 * 4096 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {1, 2, 4}) {
    int grid = 256 * wpc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)grid * 4 * ITER * instr_per_iter / 1024.0;
    printf("%-10s waves/SIMD %d: %.3f ms -> %.2f cycles/instr @2.1GHz\n", name, wpc, ms, ms * 1e-3 * 2.1e9 / per_simd);
  }
}
int main() { run("fma3", k_fma3, 16); run("fma2s", k_fma2s, 16); run("fma_chain2", k_fma_chain, 16); return 0; }
