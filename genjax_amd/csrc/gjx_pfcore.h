// gjx_pfcore.h — the skeleton of the one-launch particle filters: steps 1 .. T-1 of an SMC run over a Scan in ONE launch, for
// ANY step model.  What a filter step does that does not depend on the model lives here ONCE —
//   the rendezvous (one tagged granule {e_b, S_b} per 1024-particle tile: GJX_WEIGHTS_TILE_SCALED, include/gjx.h), the prefix of
//   the shifted tile totals, the systematic comb, the source tiles, the re-scan of their log-weights, the in-tile search, the
//   `ready` words that order a step's stores before the next step's foreign reads, the LSE ring, peer-mapped addressing for a
//   collection sharded over the GPUs of a node (gjx_peer.hip) —
// and the model is a policy class: what it stages per launch and per step, what it may draw while the granules travel, and how
// ONE slot is produced from its ancestor (gather the carry, [move], propagate, reweight).  Two models instantiate it:
//   * the hand-written linear-Gaussian model (gjx_pfilter.inl: k_pf_persistent, BASELINE configs 3 / 4), and
//   * the model GENERATED from the site list of any Scan kernel's step program (gjx_codegen.hip: gjx_gen_pf — the step programs of
//     inference/scan_filter.py), which therefore runs with the same search, the same 16 waves per tile and the same sharding.
// Reference: there is no filter in the reference; the recursion is Scan.generate's (combinators/scan.py:237-294).
//
// Model concept (every method is called by ALL threads of the block unless stated; force-inlined):
//   struct Draws;                                   random words / standard draws of ONE slot that do not depend on the carry
//   void  prologue(int tid);                        constants of the launch -> LDS (a block barrier follows)
//   void  stage(int t, int tid);                    constants of step t -> LDS; runs while the granules travel (every wave has left
//                                                   step t-1), read only behind later barriers
//   void  draw(int t, key2 key, uint64_t gidx, Draws&);
//   float slot(int t, key2 key, int j, bool active, int sg, int sl, uint64_t gidx, const Draws* hoisted, const PfSlotCtx&);
//                                                   slot j (local index) of step t from the particle at local index sl of rank sg:
//                                                   stores its rows, returns its incremental log-weight
//   void  seal_prev(int t_prev, int j, uint32_t gslot, const PfSlotCtx&);   verify mode: check word of the rows the PREVIOUS launch wrote
//   void  epilogue(int lane);
#pragma once
#include "gjx_tile.h"

namespace gjx {

constexpr int kPfCoreThreads = 1024;
constexpr int kPfCorePad = 8;             // granules one per 64-byte line

struct PfCoreArgs {
  int T;                                  // the run has T steps; this launch runs t = 1 .. T-1 and finishes the records of 0 .. T-1
  int64_t K;                              // particles of THIS rank
  int64_t K_total;
  int64_t offset;                         // global index of this rank's first particle (stream index)
  int G, rank;
  int nt;                                 // tiles of this rank = ceil(K / 1024); sharded: K % 1024 == 0
  int NT;                                 // G * nt
  float* lw_even; float* lw_odd;          // log-weights of step t in lw_odd when (T - 1 - t) is odd
  unsigned long long* aggA; unsigned long long* aggB;    // [NT * kPfCorePad] this rank's copy of the granules (alternating steps)
  float* bsum; float* bmax;               // [3][NT] LSE ring: per tile {max, sum exp(lw - max)}
  unsigned* ready;                        // [G * gridDim.x]
  const long long* peer_data;             // [G] byte distance from this rank's data window to rank g's mapping (NULL: one rank)
  const long long* peer_flag;             // [G] the same for the flag window
  const uint32_t* keys;                   // [T][2] propagation key of every step
  const double* us;                       // [T]    comb offset of every step
  float* lse_steps;                       // [T][4]
  int32_t* ancestors;                     // [K] GLOBAL ancestor index of every slot at the last step (or NULL)
  int32_t* ancestors_all;                 // [T-1][K] the same for every step (or NULL)
  unsigned* ctrl;                         // control block words: [0] epoch, [2] status
  float log_k;                            // log K_total
  unsigned first_budget;                  // polls a lane may spend in the FIRST rendezvous (peers launch later)
  unsigned long long* zero_ptr;           // sharded: the granule arrays of the NEXT launch's flag region, cleared here
  int zero_n;
  int verify;                             // GJX_PEER_VERIFY (gjx_peer.hip): 1 check, 2 check against deliberately wrong words
  unsigned* chk_a; unsigned* chk_b;       // [K] check words of the rows of even / odd steps
  unsigned long long* timeline;           // debug (gjx_debug_timeline): 16 stamps per block for step T / 2, or NULL
};

// what a model's slot() needs to address other ranks and to report
struct PfSlotCtx {
  const long long* sPD;                   // LDS copy of peer_data (zeros on one rank)
  bool sys;                               // system-scope accesses (G > 1)
  int verify;
  bool live;                              // the collection has weight (a dead step keeps every particle: nothing foreign is read)
  int64_t K, offset;
  unsigned* chk_prev; unsigned* chk_cur;
  unsigned* ctrl;
  GJX_DEV void mismatch() const { __hip_atomic_fetch_or(&ctrl[2], kStatusVerifyMismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

// ---- arguments of `gjx_gen_pf`, the filter kernel GENERATED for a step program (gjx_codegen.hip generate_pf): the skeleton's, the
// step programs' tables (one structure, T tables: they differ in the step's observation / scanned-over argument) and the rows ----
struct GenPfArgs {
  PfCoreArgs core;
  const float* const* tabs;             // [T] device tables of the step programs (tabs[0] unused: step 0 ran before)
  float* rows_a; float* rows_b;         // choices f32[n_slots][K] of even / odd steps ...
  float* rows_all; int64_t rows_step;   // ... or, when the run is recorded, step t at rows_all + t * rows_step
  int64_t in_row0_first, in_row0;       // floats in front of the OWN rows of step 0 / of the later steps in their buffer
  // rejuvenation (kernels generated with the move: generate_pf | 512): behind every resampling from the second on, n_moves random-walk
  // Metropolis steps of scale move_scale on the gathered carry, target = the previous step's density given ITS inputs
  int n_moves;
  float move_scale;
  unsigned long long* acc_total;        // [1] accepted moves over the launch (or NULL)
};

GJX_DEV uint64_t pfc_readlane_u64(uint64_t v, int l) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
// block-uniform values that come out of LDS sit in VGPRs unless the compiler is told: move them to SGPRs
GJX_DEV int pfc_uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
GJX_DEV uint64_t pfc_uni_u64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
GJX_DEV double pfc_uni_f64(double v) { return __longlong_as_double((long long)pfc_uni_u64((uint64_t)__double_as_longlong(v))); }
GJX_DEV float pfc_uni_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// dynamic LDS of pf_core for NT tiles: prefix [NT + 1] u64 (padded to even), cumulative q [4][1024] u64, exponents [NT] i32
// (+ MULTI: the prefix of the tiles' spacing sums, [NT + 1] u64 behind the exponents)
inline __host__ __device__ size_t pf_core_dyn_lds(int NT, bool multi = false) {
  return 8 * (size_t)((NT + 2) & ~1) + 8 * (size_t)(kPfCoreThreads / 256) * kPfCoreThreads + 4 * (size_t)((NT + 1) & ~1) + (multi ? 8 * (size_t)((NT + 2) & ~1) : 0);
}

// SPL: tiles per block (a lane produces one slot of each).  `pf_dyn`: the block's dynamic LDS (pf_core_dyn_lds bytes).
// SYSM: scope of the accesses other blocks / ranks observe — 0 agent (one rank), 1 system (peer-mapped windows), 2 decided at run
// time by f.G (the hand-written kernels: one instantiation for both).  VERM: 0 verify mode compiled out, 2 decided by f.verify.
// MULTI: MULTINOMIAL resampling by sorted uniforms instead of the systematic comb (include/gjx.h GJX_FILTER_MULTINOMIAL): slot j's
// threshold is floor(U_(j) total), U_(j) = S_j / S_{N+1} the j-th smallest of N uniforms built from exponential spacings of the
// slots' words under the step's resampling key (exp_spacing, gjx_device.h: exact integers).  A lane draws its slots' spacings (they
// depend on the key only), the in-tile running sums ride on the barrier of the tile maxima, a tile's spacing total travels in the
// second word of its granule, the prefix over the tiles' totals is taken beside the prefix of the weights; everything behind the
// threshold — tile search, re-scan of the source tiles, in-tile search — is the comb's.  f.us[t] then holds the resampling KEY of
// step t (its two words as one 64-bit pattern), not a comb offset.
template <class Model, int SPL, int SYSM = 2, int VERM = 2, bool MULTI = false>
GJX_DEV void pf_core(const PfCoreArgs& f_in, Model& m, unsigned char* pf_dyn) {
  struct ArgsView : PfCoreArgs {                 // (compile-time modes fold the fields they fix)
    GJX_DEV explicit ArgsView(const PfCoreArgs& a) : PfCoreArgs(a) { if (VERM == 0) verify = 0; }
  };
  const ArgsView f(f_in);
  constexpr int THREADS = kPfCoreThreads;
  constexpr int NW = THREADS / 64;               // waves per block
  constexpr int WPT = THREADS / 256;             // waves that re-scan one source tile together (256 particles each)
  constexpr int kChunk = NW / WPT;               // source tiles re-scanned per round (= 4)
  // (the draws of the block's FIRST tile are done inside the granule wait — about what the wait can hide: one tile's hashes
  // are ~1.4 us of a SIMD's issue with four waves on it — and kept in registers; the other tiles draw behind the loads of
  // their ancestor's state.  128 VGPRs is all a lane of a 1024-thread block has.)
  uint64_t* const P = (uint64_t*)pf_dyn;                               // [NT + 1] prefix of the shifted tile totals
  uint64_t* const cumL = P + ((f.NT + 2) & ~1);                        // [kChunk][THREADS] cumulative q of the tiles being searched
  int32_t* const Eb = (int32_t*)(cumL + kChunk * THREADS);             // [NT] tile exponents
  uint64_t* const P2 = (uint64_t*)(Eb + ((f.NT + 1) & ~1));            // MULTI: [NT + 1] prefix of the tiles' spacing sums
  __shared__ uint64_t wexp[MULTI ? SPL : 1][NW];                       // MULTI: the waves' spacing sums of the block's tiles
  __shared__ uint32_t sRes[2][2];                                      // MULTI: resampling key of the step (staged a step ahead)
  __shared__ float fred[SPL][NW], fsum[SPL][NW], fmx[NW];
  __shared__ uint64_t wtot[SPL][NW], wq[NW];
  __shared__ float lse_pm[NW], lse_ps[NW];
  __shared__ double sU;
  __shared__ uint32_t sKey[2][2];
  __shared__ int s_range[2], s_dead;
  __shared__ long long sPD[GJX_MAX_RANKS], sPF[GJX_MAX_RANKS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool sys = SYSM == 2 ? f.G > 1 : SYSM == 1;
  m.prologue(tid);
  if (tid < GJX_MAX_RANKS) {
    sPD[tid] = (f.peer_data && tid < f.G) ? f.peer_data[tid] : 0;
    sPF[tid] = (f.peer_flag && tid < f.G) ? f.peer_flag[tid] : 0;
  }
  if (tid == 0) s_dead = 0;
  // Sharded: consecutive launches alternate between two flag regions, and a launch clears the granules of the region the
  // NEXT launch uses — quiescent by now: its last writers (launch n-1) had all landed before this rank left that launch,
  // and no rank can enter launch n+1 before this one has published its last granule of launch n.  (A memset between
  // launches would race with a faster rank's first pushes.)
  if (f.zero_ptr) for (int i = (int)blockIdx.x * THREADS + tid; i < f.zero_n; i += (int)gridDim.x * THREADS) f.zero_ptr[i] = 0ull;
  const unsigned epoch = (unsigned)pfc_uni_i32((int)__hip_atomic_load(&f.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const int64_t K = f.K;
  const int nb = (int)gridDim.x, T = f.T, NT = f.NT, nt = f.nt, G = f.G;
  const int lt0 = (int)blockIdx.x * SPL;         // the block's first tile, local and global index
  const int gt0 = f.rank * nt + lt0;
  const int jl0 = lt0 * THREADS + tid;           // the slot this lane produces in tile s = the particle it scans: jl0 + s * 1024
  unsigned actm = 0u, tonm = 0u;                 // per tile s: the tile exists / this lane's slot in it exists
#pragma unroll
  for (int s = 0; s < SPL; ++s) {
    if (lt0 + s < nt) tonm |= 1u << s;
    if (lt0 + s < nt && (int64_t)jl0 + s * THREADS < K) actm |= 1u << s;
  }
#define jl(s) (jl0 + (s) * THREADS)
#define act(s) ((actm >> (s)) & 1u)
#define ton(s) ((tonm >> (s)) & 1u)
#define GJX_CSTAMP(n) do { if (f.timeline && t == T / 2 && tid == 0) f.timeline[blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  auto lw_buf = [&](int t) { return ((T - 1 - t) & 1) ? f.lw_odd : f.lw_even; };
  auto chk_buf = [&](int t) { return (t & 1) ? f.chk_b : f.chk_a; };   // verify mode: check words of the rows of step t
  PfSlotCtx cx;
  cx.sPD = sPD; cx.sys = sys; cx.verify = f.verify; cx.live = true; cx.K = K; cx.offset = f.offset; cx.chk_prev = nullptr; cx.chk_cur = nullptr; cx.ctrl = f.ctrl;
  float lw_own[SPL];
#pragma unroll
  for (int s = 0; s < SPL; ++s) lw_own[s] = act(s) ? lw_buf(0)[jl(s)] : -INFINITY;    // step 0 ran in the previous launch
  if (f.verify) {                                // step 0's rows (written by the previous launch) get their check words here:
    cx.chk_cur = chk_buf(0);                     // complete before this block's first `ready` word, like every store of a step
#pragma unroll 1
    for (int s = 0; s < SPL; ++s) {
      if (!act(s)) continue;
      m.seal_prev(0, jl(s), (uint32_t)(f.offset + jl(s)), cx);
    }
  }
  // ---- LSE record of step s from ring slot s % 3: loads out (fixed trip count), reduced later, written by thread 0 ----
  auto lse_ring_issue = [&](int s, float& rpm, float& rps) {       // the first entry of this thread: stays in flight
    const int b = (tid + THREADS / 2) % THREADS;
    rpm = b < NT ? load_scoped(f.bmax + (size_t)(s % 3) * NT + b, sys) : -INFINITY;
    rps = b < NT ? load_scoped(f.bsum + (size_t)(s % 3) * NT + b, sys) : 0.0f;
  };
  auto lse_ring_reduce = [&](int s, float rpm, float rps) {        // + the entries beyond the first 1024 (NT > 1024)
    const float* rm = f.bmax + (size_t)(s % 3) * NT;
    const float* rs = f.bsum + (size_t)(s % 3) * NT;
    float mm = rpm, sm = rps;
    for (int b = (tid + THREADS / 2) % THREADS + THREADS; b < NT; b += THREADS) {
      const float pm = load_scoped(rm + b, sys), ps = load_scoped(rs + b, sys);
      const float nm = fmaxf(mm, pm);
      if (nm > -INFINITY) sm = sm * fast_exp(mm - nm) + ps * fast_exp(pm - nm);
      mm = nm;
    }
    const float wm = wave_max_dpp(mm);
    const float wsm = wave_sum_dpp(wm > -INFINITY ? sm * fast_exp(mm - wm) : 0.0f);
    if (lane == 0) { lse_pm[wid] = wm; lse_ps[wid] = wsm; }
  };
  auto lse_ring_write = [&](int s) {             // thread 0, behind a barrier after lse_ring_reduce
    float mm = lse_pm[0];
    for (int w = 1; w < NW; ++w) mm = fmaxf(mm, lse_pm[w]);
    float se = 0.0f;
    for (int w = 0; w < NW; ++w) se += mm > -INFINITY ? lse_ps[w] * fast_exp(lse_pm[w] - mm) : 0.0f;
    const float l = mm > -INFINITY ? mm + logf(se) : -INFINITY;
    float* rec = f.lse_steps + 4 * (size_t)s;
    rec[0] = mm; rec[1] = se; rec[2] = l; rec[3] = l - f.log_k;
  };
  auto stage_step_constants = [&](int t) {
    if (t < T && wid == 1) {
      if (lane == 63) sU = f.us[t];
      if (lane >= 61 && lane < 63 && t + 1 < T) sKey[(t + 1) & 1][lane - 61] = f.keys[2 * (t + 1) + (lane - 61)];
      if (MULTI && lane >= 59 && lane < 61 && t + 1 < T) sRes[(t + 1) & 1][lane - 59] = ((const uint32_t*)f.us)[2 * (t + 1) + (lane - 59)];
    }
    if (t < T) m.stage(t, tid);
  };
  if (tid < 2) sKey[1][tid] = f.keys[2 + tid];   // step 1's key (T > 1)
  if (MULTI && tid >= 2 && tid < 4) sRes[1][tid - 2] = ((const uint32_t*)f.us)[2 + (tid - 2)];
  __syncthreads();
  for (int t = 1; t <= T; ++t) {
    GJX_CSTAMP(0);
    typename Model::Draws hoisted;
    int eb[SPL], Emax = kTileDead;
    float bm[SPL];
    unsigned rdy0;
    const unsigned rtag = epoch + (unsigned)t;   // `ready` word of this step: never repeats, the epoch advances by 2 T per launch
    const int nready = G * nb;
    // sticky: a launch in which one rendezvous timed out (a rank is missing, the grid is not co-resident) stops waiting
    const bool flagged = (__hip_atomic_load(&f.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusPollTimeout) != 0;
    unsigned step_budget = (s_dead || flagged) ? 0u : (t == 1 ? f.first_budget : kPollBudget);
    auto timed_out = [&]() {
      __hip_atomic_fetch_or(&f.ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_dead = 1;
    };
    auto check_ready = [&]() {                   // every block of every rank has completed its stores of step t-1
      for (int b = tid; b < nready; b += THREADS) {
        unsigned r = b == tid ? rdy0 : load_scoped_u32(&f.ready[b], sys);   // (monotone compare: a block a step ahead has completed this one)
        while ((int)(r - rtag) < 0 && step_budget) {
          __builtin_amdgcn_s_sleep(1);
          r = load_scoped_u32(&f.ready[b], sys);
          --step_budget;
        }
        if ((int)(r - rtag) < 0) timed_out();
      }
    };
    // ---- tile maxima of log w_{t-1} ----
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      const float wm = wave_max_dpp(act(s) ? lw_own[s] : -INFINITY);
      if (lane == 0) fred[s][wid] = wm;          // (last read two barriers ago)
    }
    // MULTI: the spacings of this lane's slots (key and slot index only) and their running sum within the wave
    uint64_t ex_incl[SPL];
    key2 kres{0u, 0u};
    if constexpr (MULTI) {
      if (t < T) {
        kres = key2{(uint32_t)pfc_uni_i32((int)sRes[t & 1][0]), (uint32_t)pfc_uni_i32((int)sRes[t & 1][1])};
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          const uint64_t e = act(s) ? exp_spacing(fold_in64(kres, (uint64_t)(f.offset + jl(s))).a) : 0ull;
          ex_incl[s] = wave_scan_u64(e);
          if (lane == 63) wexp[s][wid] = ex_incl[s];
        }
      }
    }
    __syncthreads();
    uint64_t ex_tile[SPL];
    if constexpr (MULTI) {
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        uint64_t below = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const uint64_t v = wexp[s][w]; tot += v; below += w < wid ? v : 0; }
        ex_incl[s] += below;
        ex_tile[s] = tot;
      }
    }
    // ---- the ONE rendezvous: {e_b, S_b} of every tile, to every rank ----
    const unsigned long long tag = (unsigned long long)((epoch + (unsigned)t) % 15u) + 1ull;
    unsigned long long* agg = (t & 1) ? f.aggA : f.aggB;   // alternate: a slow block may still poll step t-1's granules
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      float mm = fred[s][0];
#pragma unroll
      for (int w = 1; w < NW; ++w) mm = fmaxf(mm, fred[s][w]);
      mm = pfc_uni_f32(mm);
      bm[s] = mm;
      eb[s] = tile_exponent(mm);
      const uint64_t qv = (act(s) && t < T) ? tile_q(lw_own[s], eb[s]) : 0;
      const float e = (act(s) && mm > -INFINITY) ? fast_exp(lw_own[s] - mm) : 0.0f;
      const uint64_t wt = wave_total_u64(qv);
      const float ws = wave_sum_dpp(e);
      if (lane == 0) { wtot[s][wid] = wt; fsum[s][wid] = ws; }
    }
    __syncthreads();
    if (wid == 0) {                              // lane g publishes this block's tiles into rank g's copies
      static_assert(NW <= 16, "the wave partials fit one DPP row");
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        const uint64_t tt = pfc_readlane_u64(row_scan_u64(lane < NW ? wtot[s][lane] : 0), 15);
        const float bs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row_sum_to_lane15(lane < NW ? fsum[s][lane] : 0.0f)), 15));
        if (ton(s) && lane < G) {
          const long long d = sPF[lane];
          if constexpr (MULTI) store_scoped_u64(peer_ptr(agg + (size_t)(gt0 + s) * kPfCorePad + 1, d), (tag << 60) | (ex_tile[s] & ((1ull << 60) - 1)), sys);
          store_scoped_u64(peer_ptr(agg + (size_t)(gt0 + s) * kPfCorePad, d), tile_granule(tag, eb[s], tt), sys);
          const size_t slot = (size_t)((t - 1) % 3) * NT + gt0 + s;
          store_scoped(peer_ptr(f.bsum + slot, d), bs, sys);
          store_scoped(peer_ptr(f.bmax + slot, d), bm[s], sys);
        }
      }
    }
    GJX_CSTAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the record of step t-2 if this block is its finisher: every tile's ring entry was complete before the `ready` words
    // this block checked in step t-1
    const bool fin = t >= 2 && (int)blockIdx.x == (t - 2) % nb;
    float rpm = -INFINITY, rps = 0.0f;
    if (fin) lse_ring_issue(t - 2, rpm, rps);
    __syncthreads();                             // every wave's stores of step t-1 (rows, log w, ring) have completed
    if (tid < G) store_scoped_u32(peer_ptr(f.ready + f.rank * nb + (int)blockIdx.x, sPF[tid]), rtag, sys);
    stage_step_constants(t);
    // a first look at the granules goes out before the draws (for the last block to publish they are all there already)
    const unsigned long long gv0 = tid < NT ? load_scoped_u64(&agg[(size_t)tid * kPfCorePad], sys) : 0ull;
    const key2 key_t = key2{sKey[t & 1][0], sKey[t & 1][1]};
    if (t < T) m.draw(t, key_t, (uint64_t)(f.offset + jl(0)), hoisted);
    if (fin) lse_ring_reduce(t - 2, rpm, rps);
    GJX_CSTAMP(2);
    {
      float em = (float)kTileDead;
      for (int b = tid; b < NT; b += THREADS) {
        unsigned long long v = b == tid ? gv0 : load_scoped_u64(&agg[(size_t)b * kPfCorePad], sys);
        while ((v >> 60) != tag && step_budget) {
          --step_budget;
          __builtin_amdgcn_s_sleep(1);
          v = load_scoped_u64(&agg[(size_t)b * kPfCorePad], sys);
        }
        if ((v >> 60) != tag) { timed_out(); v = 0; }
        const uint64_t S = v & ((1ull << 40) - 1);
        const int e = S ? (int)((v >> 40) & 0xFFFFFu) + kTileDead : kTileDead;
        P[b + 1] = S;
        Eb[b] = e;
        em = fmaxf(em, (float)e);
        if constexpr (MULTI) {                   // the tile's spacing total: the second word of its granule, under the same tag
          unsigned long long v2 = load_scoped_u64(&agg[(size_t)b * kPfCorePad + 1], sys);
          while ((v2 >> 60) != tag && step_budget) {
            --step_budget;
            __builtin_amdgcn_s_sleep(1);
            v2 = load_scoped_u64(&agg[(size_t)b * kPfCorePad + 1], sys);
          }
          if ((v2 >> 60) != tag) { timed_out(); v2 = 0; }
          P2[b + 1] = v2 & ((1ull << 60) - 1);
        }
      }
      // the `ready` words: the first load is issued now and looked at after the tile search
      rdy0 = tid < nready ? load_scoped_u32(&f.ready[tid], sys) : rtag;
      em = wave_max_dpp(em);
      if (lane == 0) fmx[wid] = em;
    }
    if (tid == 0) { P[0] = 0; if (MULTI) P2[0] = 0; }
    __syncthreads();
    GJX_CSTAMP(3);
    {
      float em = fmx[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) em = fmaxf(em, fmx[w]);
      Emax = pfc_uni_i32((int)em);
      if (fin && tid == 0) lse_ring_write(t - 2);
    }
    if (NT <= 512) {
      // few tiles: the prefix by ONE wave — ceil(NT / 64) entries per lane, one DPP scan — instead of 16 waves and a barrier between
      // their partial sums (as k_ssm_persistent does for its 256 tiles)
      if (wid == 0) {
        const int per = (NT + 63) >> 6;
        const int e0 = lane * per < NT ? lane * per : NT, e1 = (e0 + per) < NT ? (e0 + per) : NT;
        uint64_t loc = 0;
        for (int e = e0; e < e1; ++e) {
          const int sh = Emax - Eb[e];
          const uint64_t g = sh < 64 ? P[e + 1] >> sh : 0;
          P[e + 1] = g;
          loc += g;
        }
        uint64_t run = wave_scan_u64(loc) - loc;
        for (int e = e0; e < e1; ++e) { run += P[e + 1]; P[e + 1] = run; }
      }
      if (MULTI && wid == 1) {                   // the prefix of the spacing totals, by the next wave at the same time
        const int per = (NT + 63) >> 6;
        const int e0 = lane * per < NT ? lane * per : NT, e1 = (e0 + per) < NT ? (e0 + per) : NT;
        uint64_t loc = 0;
        for (int e = e0; e < e1; ++e) loc += P2[e + 1];
        uint64_t run = wave_scan_u64(loc) - loc;
        for (int e = e0; e < e1; ++e) { run += P2[e + 1]; P2[e + 1] = run; }
      }
    } else {
      // prefix of the shifted tile totals: thread i owns the entries [i per, (i + 1) per)
      const int per = (NT + THREADS - 1) / THREADS;
      const int e0 = tid * per < NT ? tid * per : NT, e1 = (e0 + per) < NT ? (e0 + per) : NT;
      uint64_t loc = 0;
      for (int e = e0; e < e1; ++e) {
        const int sh = Emax - Eb[e];
        const uint64_t g = sh < 64 ? P[e + 1] >> sh : 0;
        P[e + 1] = g;
        loc += g;
      }
      const uint64_t inc = wave_scan_u64(loc);
      if (lane == 63) wq[wid] = inc;
      __syncthreads();
      uint64_t run = inc - loc;
      for (int w = 0; w < wid; ++w) run += wq[w];
      for (int e = e0; e < e1; ++e) { run += P[e + 1]; P[e + 1] = run; }
      if constexpr (MULTI) {                     // ... and of the spacing totals (a second pass: wq is reused)
        __syncthreads();
        uint64_t loc2 = 0;
        for (int e = e0; e < e1; ++e) loc2 += P2[e + 1];
        const uint64_t inc2 = wave_scan_u64(loc2);
        if (lane == 63) wq[wid] = inc2;
        __syncthreads();
        uint64_t run2 = inc2 - loc2;
        for (int w = 0; w < wid; ++w) run2 += wq[w];
        for (int e = e0; e < e1; ++e) { run2 += P2[e + 1]; P2[e + 1] = run2; }
      }
    }
    __syncthreads();
    GJX_CSTAMP(4);
    const uint64_t total = pfc_uni_u64(P[NT]);
    if (t == T) {                                // the last record, once every tile's ring entry is complete
      check_ready();
      __syncthreads();
      if ((int)blockIdx.x == (T - 1) % nb) {
        float qpm, qps;
        lse_ring_issue(T - 1, qpm, qps);
        lse_ring_reduce(T - 1, qpm, qps);
        __syncthreads();
        if (tid == 0) lse_ring_write(T - 1);
      }
      break;
    }
    if (total == 0 && blockIdx.x == 0 && tid == 0) __hip_atomic_fetch_or(&f.ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* lw_prev = lw_buf(t - 1);
    float* lw_out = lw_buf(t);
    const int kpad = nt * THREADS;               // ancestors live in TILE space: (global tile) * 1024 + index in the tile
    const double step = total > 0 ? (double)total / (double)f.K_total : 0.0;
    const double u_t = pfc_uni_f64(sU);
    // ---- ancestors of ALL the block's tiles together: the SPL slot tiles of a block draw from a common, contiguous range
    //      of source tiles (thresholds ascend with the slot index, so tile s+1's sources start where tile s's end): that
    //      range is re-scanned ONCE, kChunk tiles per round — a source tile shared by two slot tiles is quantised once, and
    //      the barriers of a round are paid per round, not per slot tile ----
    int nton = 0;                                // tiles this block owns (block-uniform: the rank's last block may own fewer)
#pragma unroll
    for (int s = 0; s < SPL; ++s) nton += ton(s) ? 1 : 0;
    int srcs[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) srcs[s] = f.rank * kpad + (act(s) ? jl(s) : 0);     // dead collection: every slot keeps its own particle (flagged)
    if (total > 0) {
      uint64_t Tjs[SPL];
      int tiles[SPL];
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        tiles[s] = 0; Tjs[s] = 0;
        if (!ton(s)) continue;
        // slots past the rank's last particle search that particle's threshold (thresholds stay non-decreasing in the tile)
        uint64_t Tj;
        if constexpr (MULTI) {
          // the slot's sorted uniform: spacings of all the slots up to and including it, over all N + 1 spacings (slots past the
          // rank's last particle drew no spacing: they repeat that particle's threshold)
          const uint64_t s_all = P2[NT] + exp_spacing(fold_in64(kres, (uint64_t)f.K_total).a);
          Tj = sorted_threshold(P2[gt0 + s] + ex_incl[s], s_all, total);
        } else
        Tj = comb_threshold(f.offset + (act(s) ? (int64_t)jl(s) : K - 1), u_t, step, total);
        // a tile's slots draw from tiles near its own index: the nine boundaries around it are read together (one LDS
        // latency, the same addresses in every lane); a threshold outside that window takes the fixed-trip descent
        const int own = gt0 + s;
        const int wlo = own - 4 < 0 ? 0 : (own - 4 > NT - 8 ? (NT - 8 < 0 ? 0 : NT - 8) : own - 4);
        uint64_t Pw[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Pw[k] = P[wlo + k < NT ? wlo + k : NT];
        int tile = wlo;
        if (Tj >= Pw[0] && Tj < Pw[8]) {
#pragma unroll
          for (int k = 1; k < 8; ++k) tile += Pw[k] <= Tj ? 1 : 0;
        } else {
          tile = 0;
          for (int sft = 1 << (31 - __builtin_clz((unsigned)NT)); sft >= 1; sft >>= 1) {
            const int p = tile + sft;                                   // P[p] = inclusive prefix of tile p - 1
            if (p <= NT - 1 && P[p] <= Tj) tile = p;
          }
        }
        Tjs[s] = (Tj - P[tile]) << (Emax - Eb[tile]);                 // residual in the source tile's own units (< S_tile)
        tiles[s] = tile;
      }
      if (tid == 0) s_range[0] = tiles[0];
#pragma unroll
      for (int s = 0; s < SPL; ++s) if (s == nton - 1 && tid == THREADS - 1) s_range[1] = tiles[s];
      check_ready();                                                  // before the barrier in front of the first foreign read
      GJX_CSTAMP(5);
      __syncthreads();                                                // (also: the previous step's cumL has been searched)
      const int tmin = pfc_uni_i32(s_range[0]), ntiles = pfc_uni_i32(s_range[1]) - tmin + 1;
      bool again = false;
      for (int c0 = 0; c0 < ntiles; c0 += kChunk) {
        // a round starts at a tile that has weight (block-uniform; the range's last tile always has)
        while (P[tmin + c0 + 1] == P[tmin + c0]) ++c0;
        if (again) __syncthreads();                                   // previous round's cumL consumed
        again = true;
        const int tl = wid / WPT, part = wid % WPT;                   // this wave: quarter `part` of source tile c0 + tl
        const bool on = c0 + tl < ntiles;
        uint64_t qi[4], sacc = 0, inc = 0;
        if (on) {
          const int tsrc = tmin + c0 + tl;
          const int g = tsrc / nt;
          const int64_t p0 = (int64_t)(tsrc - g * nt) * THREADS + part * 256 + lane * 4;
          const float* lwp = peer_ptr(lw_prev, sPD[g]);
          float lw4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) lw4[k] = -INFINITY;
          if (p0 + 4 <= K) load_scoped_x4(lwp + p0, lw4, sys);        // (p0 is a multiple of 4, the buffer 16-byte aligned)
          else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (p0 + k < K) lw4[k] = load_scoped(lwp + p0 + k, sys);
          }
          const int es = Eb[tsrc];
#pragma unroll
          for (int k = 0; k < 4; ++k) { sacc += p0 + k < K ? tile_q(lw4[k], es) : 0; qi[k] = sacc; }
          inc = wave_scan_u64(sacc);
          if (lane == 63) wq[wid] = inc;                              // the wave's total: offset of the next quarter
        }
        __syncthreads();
        if (f.verify && on && part == 0 && lane == 0) {
          // the log-weights this block just pulled must quantise to the total their owner published in the tile's granule
          const int tsrc = tmin + c0 + tl;
          uint64_t tot = 0;
          for (int w = 0; w < WPT; ++w) tot += wq[tl * WPT + w];
          const int sh = Emax - Eb[tsrc];
          if ((sh < 64 ? tot >> sh : 0) != P[tsrc + 1] - P[tsrc]) cx.mismatch();
        }
        if (on) {
          uint64_t base = inc - sacc;
          for (int w = 0; w < part; ++w) base += wq[tl * WPT + w];
#pragma unroll
          for (int k = 0; k < 4; ++k) cumL[tl * THREADS + part * 256 + lane * 4 + k] = base + qi[k];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          const int kpos = tiles[s] - tmin;
          if (ton(s) && kpos >= c0 && kpos < c0 + kChunk) {
            const uint64_t* cm = cumL + (kpos - c0) * THREADS;
            const uint64_t Tj = Tjs[s];
            int l2 = 0;                                               // number of entries <= the residual, 4-ary descent
#pragma unroll
            for (int q = THREADS >> 2; q >= 1; q >>= 2) {
              const uint64_t pa = cm[l2 + q - 1], pb = cm[l2 + 2 * q - 1], pc = cm[l2 + 3 * q - 1];
              l2 += (pa <= Tj ? q : 0) + (pb <= Tj ? q : 0) + (pc <= Tj ? q : 0);
            }
            srcs[s] = tiles[s] * THREADS + l2;
          }
        }
      }
    } else {
      check_ready();               // (a dead step reads nothing foreign, but the ring entries count on every step's check)
    }
    GJX_CSTAMP(6);
    cx.live = total > 0;
    cx.chk_prev = chk_buf(t - 1);
    cx.chk_cur = chk_buf(t);
    // ---- per tile of the block, one after the other (a rolled loop: 128 VGPRs hold one slot's propagation) ----
#pragma unroll 1
    for (int s = 0; s < SPL; ++s) {
      if (!ton(s)) break;                        // block-uniform
      const bool a = act(s);
      const int j = jl(s);
      int src = srcs[0];
#pragma unroll
      for (int k = 1; k < SPL; ++k) if (k == s) src = srcs[k];
      const int sg = SYSM == 0 ? 0 : src / kpad, sl = src - sg * kpad;          // (one rank: no division)
      const int32_t ganc = (int32_t)((int64_t)sg * K + sl);
      if (a && f.ancestors_all) f.ancestors_all[(size_t)(t - 1) * (size_t)K + j] = ganc;
      if (a && t == T - 1 && f.ancestors) f.ancestors[j] = ganc;
      const float lw = m.slot(t, key_t, j, a, sg, sl, (uint64_t)(f.offset + j), s == 0 ? &hoisted : nullptr, cx);
      if (a) store_scoped(lw_out + j, lw, sys);
#pragma unroll
      for (int k = 0; k < SPL; ++k) if (k == s) lw_own[k] = a ? lw : -INFINITY;
    }
    GJX_CSTAMP(7);
  }
  if (blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(&f.ctrl[0], epoch + 2u * (unsigned)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  m.epilogue(lane);
#undef jl
#undef act
#undef ton
#undef GJX_CSTAMP
}

}  // namespace gjx
