// ub5.hip — does VGPR bank placement change the cost of a Threefry round?  Explicit physical registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define ITER 2048

#define STR2(x) #x
#define STR(x) STR2(x)
// one round on chain (A = x0 reg number, B = x1 reg number)
#define ROUND(A, B, R) "v_add_u32 v" STR(A) ", v" STR(A) ", v" STR(B) "\n v_alignbit_b32 v" STR(B) ", v" STR(B) ", v" STR(B) ", " STR(R) "\n v_xor_b32 v" STR(B) ", v" STR(B) ", v" STR(A) "\n"
// grouped: all adds, all rots, all xors for 4 chains
#define ADD(A, B) "v_add_u32 v" STR(A) ", v" STR(A) ", v" STR(B) "\n"
#define ROT(B, R) "v_alignbit_b32 v" STR(B) ", v" STR(B) ", v" STR(B) ", " STR(R) "\n"
#define XOR(A, B) "v_xor_b32 v" STR(B) ", v" STR(B) ", v" STR(A) "\n"
#define XAD(A, B) "v_xad_u32 v" STR(A) ", v" STR(B) ", v" STR(A) ", v" STR(A) "\n"   /* placeholder */

#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25"

#define KERNEL(NAME, BODY, NINSTR)                                                              \
  __global__ __launch_bounds__(256) void NAME(unsigned long long* out, unsigned seed) {          \
    asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n" \
                 "v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n" : : "v"(seed + threadIdx.x) : CLOB); \
    __builtin_amdgcn_s_barrier();                                                                \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                    \
    for (int i = 0; i < ITER; ++i) { asm volatile(BODY : : : CLOB); }                            \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                    \
    unsigned s;                                                                                  \
    asm volatile("v_xor_b32 %0, v10, v11\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v13\n v_xor_b32 %0, %0, v17\n v_xor_b32 %0, %0, v21" : "=v"(s) : : CLOB); \
    if (s == 0x12345678u) out[4096 * 8] = s;                                                     \
    if ((threadIdx.x & 63) == 0) {                                                               \
      const int w = blockIdx.x * 4 + (threadIdx.x >> 6);                                         \
      out[4 * w] = t1 - t0; out[4 * w + 1] = r1 - r0; out[4 * w + 2] = r0; out[4 * w + 3] = r1;  \
    }                                                                                            \
  }                                                                                              \
  static const double NAME##_n = (double)(NINSTR) * ITER;

// A: 4 chains, x0/x1 in different banks: (10,11) (12,13) (14,15) (16,17); sequential rounds per chain
KERNEL(r_diffbank_seq, ROUND(10,11,19) ROUND(12,13,19) ROUND(14,15,19) ROUND(16,17,19), 12)
// B: same bank: (10,14) (11,15) (12,16) (13,17)
KERNEL(r_samebank_seq, ROUND(10,14,19) ROUND(11,15,19) ROUND(12,16,19) ROUND(13,17,19), 12)
// C: grouped by op type, different banks
KERNEL(r_diffbank_grp, ADD(10,11) ADD(12,13) ADD(14,15) ADD(16,17) ROT(11,19) ROT(13,19) ROT(15,19) ROT(17,19) XOR(10,11) XOR(12,13) XOR(14,15) XOR(16,17), 12)
// D: grouped, same bank
KERNEL(r_samebank_grp, ADD(10,14) ADD(11,15) ADD(12,16) ADD(13,17) ROT(14,19) ROT(15,19) ROT(16,19) ROT(17,19) XOR(10,14) XOR(11,15) XOR(12,16) XOR(13,17), 12)
// E: 8 chains grouped, different banks (x0 in 10..17 even/odd mix: pairs (10,19),(11,18)? keep simple: (10,11)...(24,25))
KERNEL(r_diffbank_grp8, ADD(10,11) ADD(12,13) ADD(14,15) ADD(16,17) ADD(18,19) ADD(20,21) ADD(22,23) ADD(24,25) ROT(11,19) ROT(13,19) ROT(15,19) ROT(17,19) ROT(19,19) ROT(21,19) ROT(23,19) ROT(25,19) XOR(10,11) XOR(12,13) XOR(14,15) XOR(16,17) XOR(18,19) XOR(20,21) XOR(22,23) XOR(24,25), 24)
// F: rot first then add+xor adjacent per chain (F F pairs): rot, add, xor
KERNEL(r_rot_first, ROT(11,19) ADD(10,11) XOR(10,11) ROT(13,19) ADD(12,13) XOR(12,13) ROT(15,19) ADD(14,15) XOR(14,15) ROT(17,19) ADD(16,17) XOR(16,17), 12)
// G: only adds and xors of the round (no rot): cost of the fast part alone
KERNEL(r_no_rot, ADD(10,11) XOR(10,11) ADD(12,13) XOR(12,13) ADD(14,15) XOR(14,15) ADD(16,17) XOR(16,17), 8)
// H: only rots
KERNEL(r_only_rot, ROT(11,19) ROT(13,19) ROT(15,19) ROT(17,19), 4)
// I: rot via two-source alignbit with different regs (non-identical operands): v_alignbit v11, v11, v13 (not a rotation; cost probe)
KERNEL(r_rot2src, "v_alignbit_b32 v11, v11, v13, 19\n v_alignbit_b32 v13, v13, v15, 19\n v_alignbit_b32 v15, v15, v17, 19\n v_alignbit_b32 v17, v17, v11, 19\n", 4)
// J: v_perm-based rot16 + add + xor
KERNEL(r_mov_dpp, "v_mov_b32_dpp v11, v11 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v13, v13 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v15, v15 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v17, v17 row_ror:4 row_mask:0xf bank_mask:0xf\n", 4)
// K: SDWA probe (may not assemble on gfx950): rot16 as two SDWA ops
// L: add with DPP? no.
// M: v_xad with rot: x0' = (x1r ^ x0) + x0?  not the same function; cost probe of xad in the chain: rot, xad(x1 = x1r ^ x0 ... )
// N: 64-bit shift based rotate: v_lshlrev_b64 on {x,x}
KERNEL(r_lshl64, "v_lshlrev_b64 v[10:11], 19, v[10:11]\n v_lshlrev_b64 v[12:13], 19, v[12:13]\n v_lshlrev_b64 v[14:15], 19, v[14:15]\n v_lshlrev_b64 v[16:17], 19, v[16:17]\n", 4)
// O: fast ops only forming a rotate substitute?  lshr + add-chain (x<<1 via add) cost probe: lshr, add, or
KERNEL(r_lshr_or, "v_lshrrev_b32 v18, 13, v11\n v_or_b32 v11, v11, v18\n v_lshrrev_b32 v19, 13, v13\n v_or_b32 v13, v13, v19\n v_lshrrev_b32 v20, 13, v15\n v_or_b32 v15, v15, v20\n v_lshrrev_b32 v21, 13, v17\n v_or_b32 v17, v17, v21\n", 8)

template <class K>
void run(const char* name, K kern, double ninstr, unsigned long long* out) {
  printf("%-18s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int grid = 256 * wps;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 3u);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 3u);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(4 * grid * 4);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, rt;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int w = 0; w < grid * 4; ++w) { cyc.push_back((double)h[4 * w]); rt.push_back((double)h[4 * w + 1]); r0 = std::min(r0, h[4 * w + 2]); r1 = std::max(r1, h[4 * w + 3]); }
    std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
    const double c = cyc[cyc.size() / 2], r = rt[rt.size() / 2];
    const double mhz = r > 0 ? c / r * 100.0 : 0.0;
    const double span_cyc = (double)(r1 - r0) * mhz / 100.0;
    printf("  w%d: %5.2f cyc/instr (%4.0f MHz)", wps, span_cyc / (ninstr * wps), mhz);
  }
  printf("\n");
}
#define RUN(N) run(#N, N, N##_n, out)
int main() {
  unsigned long long* out; CK(hipMalloc(&out, 1 << 20));
  RUN(r_diffbank_seq); RUN(r_samebank_seq); RUN(r_diffbank_grp); RUN(r_samebank_grp); RUN(r_diffbank_grp8); RUN(r_rot_first);
  RUN(r_no_rot); RUN(r_only_rot); RUN(r_rot2src); RUN(r_mov_dpp); RUN(r_lshl64); RUN(r_lshr_or);
  return 0;
}
