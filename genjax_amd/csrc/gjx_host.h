// gjx_host.h — host-side helpers shared by the launchers (error reporting only; no state).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/gjx.h"

int gjx_fail(int status, const char* msg);           // records the thread-local message, returns status
int gjx_fail_hip(hipError_t e, const char* where);   // same for a HIP error

#define GJX_CHECK_LAUNCH(where)                                  \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) return gjx_fail_hip(e__, where);      \
  } while (0)

// {max,sumexp} block partials -> out[4] = {max, sumexp, lse, lse - log(K_total)}  (gjx_run.hip)
int gjx_launch_lse_finish(const void* partials_float2, int n, int64_t K_total, float* out, hipStream_t st);

// Number of thread blocks of `kernel` (block size `threads`, `dyn_lds` bytes of dynamic LDS) that are resident on the
// current device AT THE SAME TIME: occupancy query x CU count, cached per (kernel, device).  Kernels that synchronise
// their blocks through memory (granule all-gathers) must not be launched with a larger grid.  GJX_CORESIDENT_BLOCKS
// overrides the answer (tests of the fallback paths).  Returns 0 when the query fails.
int gjx_coresident_blocks(const void* kernel, int threads, size_t dyn_lds);
// While one of these lives on a thread, gjx_coresident_blocks answers 0 on that thread: every launcher below it takes its
// plain multi-launch path (GJX_WEIGHTS_PLAIN_LAUNCHES of gjx_ssm_filter_scheme: the repeat after a poll time-out).
struct gjx_plain_launch_scope {
  explicit gjx_plain_launch_scope(bool on);
  ~gjx_plain_launch_scope();
  bool on_;
};
bool gjx_plain_launches_forced();

namespace gjx {
// phase-stamp buffer of the profiling scripts (gjx_debug_timeline): the registered device buffer if it holds `need` bytes
unsigned long long* debug_timeline(size_t need);
struct GenArgs;
// per-program generated kernels (gjx_codegen.hip)
int gen_pick_ppt(const gjx_program* prog, int64_t K, bool prefer4 = false);
int gen_available(const gjx_program* prog, int ppt);
int gen_launch(const gjx_program* prog, int ppt, const GenArgs& args, int grid, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
// the steps kernel of a generated filter (gjx_gen_steps, gjx_codegen.hip): every step of a run in one launch
struct GenStepsArgs;
bool gen_same_kernel(const gjx_program* p, const gjx_program* q, int ppt);
int gen_steps_resident_blocks(const gjx_program* prog, int ppt);
int gen_steps_launch(const gjx_program* prog, int ppt, const GenStepsArgs& args, int grid, hipStream_t st);
// the filter kernel generated for a step program (gjx_gen_pf on the skeleton of gjx_pfcore.h): steps 1 .. T-1 of a run in one launch,
// 16 waves per 1024-particle tile, `spl` tiles per block
struct GenPfArgs;
bool gen_pf_supported(const gjx_program* p);
bool gen_pf_moves_supported(const gjx_program* p);   // the kernel can carry the rejuvenation move (spl | 512)
bool gen_pf_same_kernel(const gjx_program* p, const gjx_program* q);
int gen_pf_precompile(const gjx_program* prog, int spl);
int gen_pf_resident_blocks(const gjx_program* prog, int spl, size_t dyn_lds);
int gen_pf_launch(const gjx_program* prog, int spl, const GenPfArgs& args, int grid, size_t dyn_lds, hipStream_t st);
// n 8-byte words from HOST memory to device memory through kernel arguments (small per-run argument arrays: step keys, comb
// offsets, table pointers): no host buffer has to outlive the call, unlike an asynchronous copy from pageable memory
int upload_words(void* dst_dev, const void* src_host, size_t n_words, hipStream_t st);
// per-program generated HMC kernels (gjx_codegen.hip)
struct HmcGenArgs;
int hmc_gen_available(const gjx_program* prog);
int hmc_gen_launch(const gjx_program* prog, const HmcGenArgs& args, hipStream_t st);
// systematic ancestor expansion with the slot run {slot0, n_valid} read from a device plan (gjx_resample.hip)
int launch_expand_planned(const uint64_t* cum, int64_t K, const gjx_shard_plan* plan_dev, double u, int64_t N_total,
                          int32_t* ancestors, int64_t anc_capacity, hipStream_t st);
// maximum tile exponent, shifts and prefix of the shifted tile totals for the tile-scaled resampler (k_tiled_plan, gjx_ssm.hip):
// P u64[nt + 1], sh i32[nt]
int launch_tiled_plan(const uint64_t* S, const int32_t* E, int nt, uint64_t* P, int32_t* sh, unsigned* ctrl, hipStream_t st);
}  // namespace gjx
