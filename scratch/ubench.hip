#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
#define DEF(NAME, TYPE, INIT, BODY) \
__global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE seed) { \
  TYPE a0 = INIT + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  TYPE s = seed; \
  for (int i = 0; i < ITER; ++i) { BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7) } \
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
#define B_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(s));
#define B_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(s));
#define B_ROT(x) asm volatile("v_alignbit_b32 %0, %0, %0, 19" : "+v"(x));
#define B_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
#define B_FMA(x) x = __builtin_fmaf(x, s, s);
#define B_FADD(x) x = x + s;
#define B_FMUL(x) x = x * s;
#define B_LOG(x) x = __builtin_amdgcn_logf(x);
#define B_EXP(x) x = __builtin_amdgcn_exp2f(x);
#define B_RCP(x) x = __builtin_amdgcn_rcpf(x);
#define B_SQRT(x) x = __builtin_amdgcn_sqrtf(x);
#define B_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#define B_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
#define B_PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#define B_PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#define B_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(s));
#define B_LSHL(x) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
#define B_PERM(x) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(x) : "v"(s));
#define B_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(s));
#define B_XORS(x) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "s"(sseed));
#define B_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(x) : "v"(s));
#define B_XAD(x) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
#define B_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(s));
#define B_MULHI(x) x = __umulhi(x, s);
DEF(k_addu, uint32_t, 1u, B_ADDU) DEF(k_xor, uint32_t, 1u, B_XOR) DEF(k_rot, uint32_t, 1u, B_ROT) DEF(k_add3, uint32_t, 1u, B_ADD3)
DEF(k_fma, float, 1.0f, B_FMA) DEF(k_fadd, float, 1.0f, B_FADD) DEF(k_fmul, float, 1.0f, B_FMUL) DEF(k_log, float, 1.0f, B_LOG) DEF(k_exp, float, 1.0f, B_EXP)
DEF(k_rcp, float, 1.0f, B_RCP) DEF(k_sqrt, float, 1.0f, B_SQRT) DEF(k_max, float, 1.0f, B_MAX) DEF(k_xad, uint32_t, 1u, B_XAD)
DEF(k_mullo, uint32_t, 1u, B_MULLO) DEF(k_cnd, uint32_t, 1u, B_CNDMASK) DEF(k_lshl, uint32_t, 1u, B_LSHL) DEF(k_perm, uint32_t, 1u, B_PERM) DEF(k_lshlor, uint32_t, 1u, B_LSHLOR) DEF(k_bfi, uint32_t, 1u, B_BFI) DEF(k_mulhi, uint32_t, 1u, B_MULHI)
__global__ __launch_bounds__(256) void k_pkfma(v2f* out, v2f seed) {
  v2f a0 = seed + (float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; v2f s = seed;
  for (int i = 0; i < ITER; ++i) { B_PKFMA(a0) B_PKFMA(a1) B_PKFMA(a2) B_PKFMA(a3) B_PKFMA(a4) B_PKFMA(a5) B_PKFMA(a6) B_PKFMA(a7) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
__global__ __launch_bounds__(256) void k_pkmul(v2f* out, v2f seed) {
  v2f a0 = seed + (float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; v2f s = seed;
  for (int i = 0; i < ITER; ++i) { B_PKMUL(a0) B_PKMUL(a1) B_PKMUL(a2) B_PKMUL(a3) B_PKMUL(a4) B_PKMUL(a5) B_PKMUL(a6) B_PKMUL(a7) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
template <class T, class K> void run(const char* name, K kern, T seed, int lanes_per_instr) {
  T* out; hipMalloc(&out, 256 * 2048 * sizeof(T));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4}) {  // blocks per CU multiples
    int grid = 256 * wpc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, seed);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, seed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)grid * 4 /*waves*/ * ITER * 8;
    double per_simd = instr / 1024.0;
    printf("%-8s grid %4d: %.3f ms  -> %.2f cycles/wave-instr/SIMD @2.4GHz (%.2f @2.1)\n", name, grid, ms, ms * 1e-3 * 2.4e9 / per_simd, ms * 1e-3 * 2.1e9 / per_simd);
  }
  hipFree(out);
}
int main() {
  run<uint32_t>("add_u32", k_addu, 3u, 64); run<uint32_t>("xor", k_xor, 3u, 64); run<uint32_t>("alignbit", k_rot, 3u, 64);
  run<uint32_t>("add3", k_add3, 3u, 64); run<uint32_t>("xad", k_xad, 3u, 64); run<uint32_t>("mul_lo", k_mullo, 3u, 64); run<uint32_t>("mul_hi", k_mulhi, 3u, 64); run<uint32_t>("cndmask", k_cnd, 3u, 64); run<uint32_t>("lshl", k_lshl, 3u, 64); run<uint32_t>("perm", k_perm, 3u, 64); run<uint32_t>("lshl_or", k_lshlor, 3u, 64); run<uint32_t>("bfi", k_bfi, 3u, 64);
  run<float>("fma", k_fma, 1.0001f, 64); run<float>("fadd", k_fadd, 1.0001f, 64); run<float>("fmul", k_fmul, 1.0001f, 64); run<float>("max", k_max, 1.0001f, 64);
  run<float>("log", k_log, 1.0001f, 64); run<float>("exp", k_exp, 1.0001f, 64); run<float>("rcp", k_rcp, 1.0001f, 64); run<float>("sqrt", k_sqrt, 1.0001f, 64);
  v2f s2 = {1.0001f, 0.9999f};
  run<v2f>("pk_fma", k_pkfma, s2, 128); run<v2f>("pk_mul", k_pkmul, s2, 128);
  return 0;
}
