// gjx_pfilter_host.h — argument block and launch plan of k_pf_persistent (gjx_pfilter.inl), shared by the kernel's
// translation units and the host code that launches it (gjx_ssm.hip on one GPU, gjx_peer.hip on a sharded collection).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/gjx.h"

namespace gjx {

constexpr int kPfThreads = 1024;
constexpr int kPfGranulePad = 8;                         // granules one per 64-byte line (as k_ssm_persistent: gjx_ssm.hip kGranulePad)
constexpr int kPfMaxTiles = 4096;                        // quantisation tiles over all ranks (K_total <= 2^22)
constexpr int kPfPer = kPfMaxTiles / kPfThreads;         // granules / ring entries / ready words a thread looks at

struct PfArgs {
  const float* A; const float* H; const float* ys;       // ys [T][dy]
  float q, r;
  int dy, T;
  int64_t K;                                             // particles of THIS rank
  int64_t K_total;
  int64_t offset;                                        // global index of this rank's first particle (stream index)
  int G, rank;
  int nt;                                                // tiles of this rank = ceil(K / 1024); sharded: K % 1024 == 0
  int NT;                                                // G * nt
  float* x_a; float* x_b;                                // [DX][K] ping-pong: step t writes x_b when t is odd
  float* lw_even; float* lw_odd;                         // log-weights of step t in lw_odd when (T - 1 - t) is odd
  unsigned long long* aggA; unsigned long long* aggB;    // [NT * kPfGranulePad] this rank's copy of the granules (alternating steps), one per 64-byte line
  float* bsum; float* bmax;                              // [3][NT] LSE ring: per tile {max, sum exp(lw - max)}
  unsigned* ready;                                       // [G * gridDim.x]
  const long long* peer_data;                            // [G] byte distance from this rank's data window to rank g's mapping (NULL: one rank)
  const long long* peer_flag;                            // [G] the same for the flag window
  const uint32_t* keys;                                  // [T][2] propagation key of every step
  const double* us;                                      // [T]    comb offset of every step
  float* lse_steps;                                      // [T][4]
  int32_t* ancestors;                                    // [K] GLOBAL ancestor index of every slot at the last step (or NULL)
  unsigned* ctrl;                                        // control block words: [0] epoch, [2] status
  float log_k;                                           // log K_total
  unsigned first_budget;                                 // polls a lane may spend in the FIRST rendezvous (peers launch later)
  unsigned long long* zero_ptr;                          // sharded: the granule arrays of the NEXT launch's flag region, cleared here
  int zero_n;
  // resample-move rejuvenation (MOVE kernels; requests/rejuvenate.py:70-94, k_ssm_step<.., MOVE>): after the ancestor gather
  // every particle takes n_moves random-walk Metropolis steps that leave p(x_{t-1} | parent, y_{t-1}) invariant
  float* m_a; float* m_b;                                // [DX][K] ping-pong like x: A x'_{t-1}, the mean step t propagated from
  float q0;                                              // prior scale (the transition that produced x_0)
  int n_moves;
  float move_scale;
  unsigned long long* acc_total;                         // [1] accepted moves of this rank's particles over the launch (or NULL)
  // verify mode (GJX_PEER_VERIFY=1, gjx_peer.hip): every propagated particle leaves a check word beside its row (gjx_tile.h
  // row_check_*), every reader of a row recomputes it; the re-scanned total of a source tile is compared with the tile's
  // granule.  A mismatch raises GJX_STATUS_VERIFY_MISMATCH.  chk_a / chk_b ping-pong like x_a / x_b (DATA window).
  int verify;
  unsigned* chk_a; unsigned* chk_b;                      // [K]
  unsigned long long* timeline;                          // debug (gjx_debug_timeline): the skeleton's 16 stamps per block for step T / 2, or NULL
};

constexpr int kPfHostThreads = kPfThreads;
constexpr int kPfHostMaxTiles = kPfMaxTiles;
// dynamic LDS of k_pf_persistent for NT tiles: prefix [NT + 1] u64 (padded to even), cumulative q [4][1024] u64, exponents [NT] i32
inline size_t pf_host_dyn_lds(int NT) { return 8 * (size_t)((NT + 2) & ~1) + 8 * (size_t)(kPfThreads / 256) * kPfThreads + 4 * (size_t)NT; }

struct PfPlan {
  const void* fn;    // kernel
  int spl;           // tiles per block
  int grid;          // blocks
  int nt;            // tiles of this rank
  size_t lds;        // dynamic LDS bytes
};
// Picks the smallest number of tiles per block whose grid is co-resident on the current device (`share` ranks on one
// device split its capacity: dry runs).  GJX_EUNSUPPORTED when the shape does not fit this kernel.
int pf_plan(int rng_mode, int dx, int dy, int64_t K_local, int n_ranks, int share, PfPlan* out, bool move = false);
void host_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]);
void pf_step_keys(uint32_t key0, uint32_t key1, int T, std::vector<uint32_t>& keys, std::vector<double>& us);
// the same, and the resampling key k_res of every step ([T][2]): multinomial resampling draws one uniform per slot from it
void pf_step_keys_res(uint32_t key0, uint32_t key1, int T, std::vector<uint32_t>& keys, std::vector<double>& us, std::vector<uint32_t>& res_keys);

}  // namespace gjx
