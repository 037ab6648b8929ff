// gjx_pfilter.inl — k_pf_persistent: steps 1 .. T-1 of the bootstrap filter in ONE launch for any number of particles a
// device holds, on one GPU or on a collection sharded over the GPUs of a node (gjx_peer.hip).
//
// k_ssm_persistent<TILED> (gjx_ssm.hip) gives every lane ONE slot, so its grid of K / 1024 blocks must be co-resident:
// K <= 2^18 on a full MI355X.  Here a block owns SPL consecutive quantisation tiles of 1024 slots (a lane: SPL slots, one
// per tile), the granule all-gather carries one {e_b, S_b} per TILE, and everything after it — prefix of the shifted tile
// totals, comb thresholds, source tiles, in-tile search, ancestor gather, propagate, reweight — runs per slot exactly as
// in k_ssm_persistent<TILED>: same scheme (GJX_WEIGHTS_TILE_SCALED, include/gjx.h), same streams, same ancestors.  The
// rendezvous cost (one per step) is shared by SPL tiles.
//
// Sharded (G > 1 ranks, one process per GPU): the tile index space is global — rank r owns tiles [r nt, (r+1) nt) —
// and every rank keeps a full copy of the granule array, the `ready` words and the LSE ring; a block PUSHES its granules
// and ring entries into every rank's copy with system-scope stores through peer-mapped pointers (hipIpc over xGMI), and
// polls only its local copy.  The log-weights of a source tile and the ancestor's state are PULLED from the rank that
// owns them through the same mappings.  No host involvement, no collective call, no kernel boundary inside the T loop;
// results do not depend on G because every integer of the resampling is computed from the same NT granules.
#pragma once
#include "gjx_tile.h"
#include "gjx_pfilter_host.h"

namespace gjx {

GJX_DEV uint64_t readlane_u64(uint64_t v, int l) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}

// block-uniform values that come out of LDS sit in VGPRs unless the compiler is told: move them to SGPRs
GJX_DEV int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
GJX_DEV uint64_t uni_u64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
GJX_DEV double uni_f64(double v) { return __longlong_as_double((long long)uni_u64((uint64_t)__double_as_longlong(v))); }
GJX_DEV float uni_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

template <int RNG, int DX, int SPL, bool MOVE = false>
__global__ __launch_bounds__(kPfThreads) void k_pf_persistent(PfArgs f) {
  constexpr int THREADS = kPfThreads;
  constexpr int NW = THREADS / 64;               // waves per block
  constexpr int WPT = THREADS / 256;             // waves that re-scan one source tile together (256 particles each)
  constexpr int kChunk = NW / WPT;               // source tiles re-scanned per round (= 4)
  // (the hashes of the block's FIRST tile are done inside the granule wait — about what the wait can hide: one tile's
  // hashes are ~1.4 us of a SIMD's issue with four waves on it — and kept in registers; the other tiles hash behind the
  // loads of their ancestor's state.  128 VGPRs is all a lane of a 1024-thread block has.)
  extern __shared__ __align__(16) unsigned char pf_dyn[];
  uint64_t* const P = (uint64_t*)pf_dyn;                               // [NT + 1] prefix of the shifted tile totals
  uint64_t* const cumL = P + ((f.NT + 2) & ~1);                        // [kChunk][THREADS] cumulative q of the tiles being searched
  int32_t* const Eb = (int32_t*)(cumL + kChunk * THREADS);             // [NT] tile exponents
  __shared__ float fred[SPL][NW], fsum[SPL][NW], fmx[NW];
  __shared__ uint64_t wtot[SPL][NW], wq[NW];
  __shared__ float lse_pm[NW], lse_ps[NW];
  __shared__ float sA[DX * DX], sH[kSsmPersistMaxDy * DX], sY[kSsmPersistMaxDy];
  __shared__ float sYp[MOVE ? kSsmPersistMaxDy : 1];       // MOVE: y_{t-1}, the observation the moved particle is conditioned on
  __shared__ double sU;
  __shared__ uint32_t sKey[2][2];
  __shared__ int s_range[2], s_dead;
  __shared__ long long sPD[GJX_MAX_RANKS], sPF[GJX_MAX_RANKS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool sys = f.G > 1;
  for (int e = tid; e < DX * DX; e += THREADS) sA[e] = f.A[e];
  if (f.H) for (int e = tid; e < f.dy * DX; e += THREADS) sH[e] = f.H[e];
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
  const unsigned epoch = (unsigned)uni_i32((int)__hip_atomic_load(&f.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
  auto lw_buf = [&](int t) { return ((T - 1 - t) & 1) ? f.lw_odd : f.lw_even; };
  auto x_buf = [&](int t) { return (t & 1) ? f.x_b : f.x_a; };
  auto m_buf = [&](int t) { return (t & 1) ? f.m_b : f.m_a; };   // MOVE: A x'_{t-1} of step t (step 0: the prior mean, zero — never read)
  auto chk_buf = [&](int t) { return (t & 1) ? f.chk_b : f.chk_a; };   // verify mode: check words of the rows of step t
  const bool verify = f.verify != 0;             // (wave-uniform: scalar branches)
  auto verify_failed = [&]() { __hip_atomic_fetch_or(&f.ctrl[2], kStatusVerifyMismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const float rr = fast_rcp(f.r);
  const float lconst = -(float)f.dy * (kHalfLog2Pi + fast_log(f.r));
  float lw_own[SPL];
#pragma unroll
  for (int s = 0; s < SPL; ++s) lw_own[s] = act(s) ? lw_buf(0)[jl(s)] : -INFINITY;    // step 0 ran in the previous launch
  if (verify) {                                  // step 0's rows (written by the previous launch) get their check words here:
#pragma unroll 1                                  // complete before this block's first `ready` word, like every store of a step
    for (int s = 0; s < SPL; ++s) {
      if (!act(s)) continue;
      uint32_t h = row_check_init(0, (uint32_t)(f.offset + jl(s)));
#pragma unroll
      for (int d = 0; d < DX; ++d) h = row_check_mix(h, x_buf(0)[(int64_t)d * K + jl(s)]);
      store_scoped_u32(chk_buf(0) + jl(s), f.verify == 2 ? h ^ 1u : h, sys);
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
    float m = rpm, sm = rps;
    for (int b = (tid + THREADS / 2) % THREADS + THREADS; b < NT; b += THREADS) {
      const float pm = load_scoped(rm + b, sys), ps = load_scoped(rs + b, sys);
      const float nm = fmaxf(m, pm);
      if (nm > -INFINITY) sm = sm * fast_exp(m - nm) + ps * fast_exp(pm - nm);
      m = nm;
    }
    const float wm = wave_max_dpp(m);
    const float wsm = wave_sum_dpp(wm > -INFINITY ? sm * fast_exp(m - wm) : 0.0f);
    if (lane == 0) { lse_pm[wid] = wm; lse_ps[wid] = wsm; }
  };
  auto lse_ring_write = [&](int s) {             // thread 0, behind a barrier after lse_ring_reduce
    float m = lse_pm[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, lse_pm[w]);
    float se = 0.0f;
    for (int w = 0; w < NW; ++w) se += m > -INFINITY ? lse_ps[w] * fast_exp(lse_pm[w] - m) : 0.0f;
    const float l = m > -INFINITY ? m + logf(se) : -INFINITY;
    float* rec = f.lse_steps + 4 * (size_t)s;
    rec[0] = m; rec[1] = se; rec[2] = l; rec[3] = l - f.log_k;
  };
  auto stage_step_constants = [&](int t) {
    if (t < T && wid == 1) {
      if (lane < f.dy) sY[lane] = f.ys[(size_t)t * f.dy + lane];
      if (MOVE && lane < f.dy) sYp[lane] = f.ys[(size_t)(t - 1) * f.dy + lane];
      if (lane == 63) sU = f.us[t];
      if (lane >= 61 && lane < 63 && t + 1 < T) sKey[(t + 1) & 1][lane - 61] = f.keys[2 * (t + 1) + (lane - 61)];
    }
  };
  if (tid < 2) sKey[1][tid] = f.keys[2 + tid];   // step 1's key (T > 1)
  __syncthreads();
  unsigned acc_lane = 0u;                         // MOVE: accepted Metropolis moves of this lane's slots, all steps
  for (int t = 1; t <= T; ++t) {
    SsmNoiseBits<RNG, DX> nbits;
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
    __syncthreads();
    // ---- the ONE rendezvous: {e_b, S_b} of every tile, to every rank ----
    const unsigned long long tag = (unsigned long long)((epoch + (unsigned)t) % 15u) + 1ull;
    unsigned long long* agg = (t & 1) ? f.aggA : f.aggB;   // alternate: a slow block may still poll step t-1's granules
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      float m = fred[s][0];
#pragma unroll
      for (int w = 1; w < NW; ++w) m = fmaxf(m, fred[s][w]);
      m = uni_f32(m);
      bm[s] = m;
      eb[s] = tile_exponent(m);
      const uint64_t qv = (act(s) && t < T) ? tile_q(lw_own[s], eb[s]) : 0;
      const float e = (act(s) && m > -INFINITY) ? fast_exp(lw_own[s] - m) : 0.0f;
      const uint64_t wt = wave_total_u64(qv);
      const float ws = wave_sum_dpp(e);
      if (lane == 0) { wtot[s][wid] = wt; fsum[s][wid] = ws; }
    }
    __syncthreads();
    if (wid == 0) {                              // lane g publishes this block's tiles into rank g's copies
      static_assert(NW <= 16, "the wave partials fit one DPP row");
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        const uint64_t tt = readlane_u64(row_scan_u64(lane < NW ? wtot[s][lane] : 0), 15);
        const float bs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row_sum_to_lane15(lane < NW ? fsum[s][lane] : 0.0f)), 15));
        if (ton(s) && lane < G) {
          const long long d = sPF[lane];
          store_scoped_u64(peer_ptr(agg + (size_t)(gt0 + s) * kPfGranulePad, d), tile_granule(tag, eb[s], tt), sys);
          const size_t slot = (size_t)((t - 1) % 3) * NT + gt0 + s;
          store_scoped(peer_ptr(f.bsum + slot, d), bs, sys);
          store_scoped(peer_ptr(f.bmax + slot, d), bm[s], sys);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the record of step t-2 if this block is its finisher: every tile's ring entry was complete before the `ready` words
    // this block checked in step t-1
    const bool fin = t >= 2 && (int)blockIdx.x == (t - 2) % nb;
    float rpm = -INFINITY, rps = 0.0f;
    if (fin) lse_ring_issue(t - 2, rpm, rps);
    __syncthreads();                             // every wave's stores of step t-1 (x, log w, ring) have completed
    if (tid < G) store_scoped_u32(peer_ptr(f.ready + f.rank * nb + (int)blockIdx.x, sPF[tid]), rtag, sys);
    stage_step_constants(t);
    // a first look at the granules goes out before the draws (for the last block to publish they are all there already)
    const unsigned long long gv0 = tid < NT ? load_scoped_u64(&agg[(size_t)tid * kPfGranulePad], sys) : 0ull;
    if (t < T) ssm_noise_bits<RNG, DX>(key2{sKey[t & 1][0], sKey[t & 1][1]}, (uint64_t)(f.offset + jl(0)), nbits);
    if (fin) lse_ring_reduce(t - 2, rpm, rps);
    {
      float em = (float)kTileDead;
      for (int b = tid; b < NT; b += THREADS) {
        unsigned long long v = b == tid ? gv0 : load_scoped_u64(&agg[(size_t)b * kPfGranulePad], sys);
        while ((v >> 60) != tag && step_budget) {
          --step_budget;
          __builtin_amdgcn_s_sleep(1);
          v = load_scoped_u64(&agg[(size_t)b * kPfGranulePad], sys);
        }
        if ((v >> 60) != tag) { timed_out(); v = 0; }
        const uint64_t S = v & ((1ull << 40) - 1);
        const int e = S ? (int)((v >> 40) & 0xFFFFFu) + kTileDead : kTileDead;
        P[b + 1] = S;
        Eb[b] = e;
        em = fmaxf(em, (float)e);
      }
      // the `ready` words: the first load is issued now and looked at after the tile search
      rdy0 = tid < nready ? load_scoped_u32(&f.ready[tid], sys) : rtag;
      em = wave_max_dpp(em);
      if (lane == 0) fmx[wid] = em;
    }
    if (tid == 0) P[0] = 0;
    __syncthreads();
    {
      float em = fmx[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) em = fmaxf(em, fmx[w]);
      Emax = uni_i32((int)em);
      if (fin && tid == 0) lse_ring_write(t - 2);
    }
    {
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
    }
    __syncthreads();
    const uint64_t total = uni_u64(P[NT]);
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
    // ---- per tile of the block, one after the other (a rolled loop: 128 VGPRs hold one slot's search and propagation):
    //      comb threshold, source tile, re-scan of the source tiles, in-tile search, ancestor gather, propagate, reweight ----
    const float* lw_prev = lw_buf(t - 1);
    const float* x_prev = x_buf(t - 1);
    float* x_out = x_buf(t);
    float* lw_out = lw_buf(t);
    const int kpad = nt * THREADS;               // ancestors live in TILE space: (global tile) * 1024 + index in the tile
    const double step = total > 0 ? (double)total / (double)f.K_total : 0.0;
    const double u_t = uni_f64(sU);
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
        uint64_t Tj = comb_threshold(f.offset + (act(s) ? (int64_t)jl(s) : K - 1), u_t, step, total);
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
      __syncthreads();                                                // (also: the previous step's cumL has been searched)
      const int tmin = uni_i32(s_range[0]), ntiles = uni_i32(s_range[1]) - tmin + 1;
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
        if (verify && on && part == 0 && lane == 0) {
          // the log-weights this block just pulled must quantise to the total their owner published in the tile's granule
          const int tsrc = tmin + c0 + tl;
          uint64_t tot = 0;
          for (int w = 0; w < WPT; ++w) tot += wq[tl * WPT + w];
          const int sh = Emax - Eb[tsrc];
          if ((sh < 64 ? tot >> sh : 0) != P[tsrc + 1] - P[tsrc]) verify_failed();
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
    // ---- per tile of the block, one after the other (a rolled loop: 128 VGPRs hold one slot's propagation) ----
#pragma unroll 1
    for (int s = 0; s < SPL; ++s) {
      if (!ton(s)) break;                        // block-uniform
      const bool a = act(s);
      const int j = jl(s);
      int src = srcs[0];
#pragma unroll
      for (int k = 1; k < SPL; ++k) if (k == s) src = srcs[k];
      // ---- propagate + reweight the slot (k_ssm_step's arithmetic and streams) ----
      const int sg = src / kpad, sl = src - sg * kpad;
      if (a && t == T - 1 && f.ancestors) f.ancestors[j] = (int32_t)((int64_t)sg * K + sl);
      const float* xs = peer_ptr(x_prev, sPD[sg]) + sl;
      float xp[DX], nz[DX], xn[DX];
#pragma unroll
      for (int d = 0; d < DX; ++d) xp[d] = load_scoped(xs + (int64_t)d * K, sys);
      if (verify) {                              // the pulled row against its owner's check word (step t - 1, global index)
        const unsigned want = load_scoped_u32(peer_ptr((const unsigned*)chk_buf(t - 1), sPD[sg]) + sl, sys);
        uint32_t h = row_check_init(t - 1, (uint32_t)((int64_t)sg * K + sl));
#pragma unroll
        for (int d = 0; d < DX; ++d) h = row_check_mix(h, xp[d]);
        if (a && total > 0 && h != want) verify_failed();
      }
      if constexpr (MOVE) {
        // resample-move: n_moves random-walk Metropolis steps on the gathered x_{t-1} with p(x_{t-1} | parent, y_{t-1}) as
        // invariant density — k_ssm_step<.., MOVE>'s arithmetic and draws (site 2 of the step's stream), so the one-launch
        // filter equals the step-by-step one bit for bit.  The parent's transition mean A x'_{t-2} was stored by the
        // previous step; step 1 moves x_0 under the prior N(0, q0^2 I).
        float mp[DX];
        const float* ms = peer_ptr((const float*)m_buf(t - 1), sPD[sg]) + sl;
#pragma unroll
        for (int d = 0; d < DX; ++d) mp[d] = t > 1 ? load_scoped(ms + (int64_t)d * K, sys) : 0.0f;
        const float rq = fast_rcp(t > 1 ? f.q : f.q0);
        auto logpi = [&](const float (&xx)[DX]) {
          float sq = 0.0f;
#pragma unroll
          for (int d = 0; d < DX; ++d) { const float z = (xx[d] - mp[d]) * rq; sq = fmaf(z, z, sq); }
          if (f.H) {
            for (int o = 0; o < f.dy; ++o) {
              float m = 0.0f;
#pragma unroll
              for (int e = 0; e < DX; ++e) m = fmaf(sH[o * DX + e], xx[e], m);
              const float z = (sYp[o] - m) * rr;
              sq = fmaf(z, z, sq);
            }
          } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) { const float z = (sYp[d] - xx[d]) * rr; sq = fmaf(z, z, sq); }
          }
          return -0.5f * sq;
        };
        BitStreamRT<RNG> bmv;
        bmv.open(key2{sKey[t & 1][0], sKey[t & 1][1]}, (uint64_t)(f.offset + j), 2u);
        float cur = logpi(xp), nacc = 0.0f;
        for (int n = 0; n < f.n_moves; ++n) {
          float xq[DX];
#pragma unroll
          for (int d = 0; d < DX; ++d) xq[d] = fmaf(f.move_scale, stream_normal<RNG>(bmv, (uint32_t)(n * (DX + 2) + d)), xp[d]);
          const float prop = logpi(xq);
          const float lu = safe_log(uniform_from_bits(bmv.get((uint32_t)(n * (DX + 2) + DX)), kTiny, 1.0f));
          if (lu < prop - cur) {
#pragma unroll
            for (int d = 0; d < DX; ++d) xp[d] = xq[d];
            cur = prop;
            nacc += 1.0f;
          }
        }
        if (a) acc_lane += (unsigned)nacc;       // summed over the launch: one atomic per block at the very end
      }
      if (s == 0) ssm_noise_normals<RNG, DX>(nbits, nz);             // behind the loads above
      else {
        SsmNoiseBits<RNG, DX> late;                                 // (tiles after the first hash here)
        ssm_noise_bits<RNG, DX>(key2{sKey[t & 1][0], sKey[t & 1][1]}, (uint64_t)(f.offset + j), late);
        ssm_noise_normals<RNG, DX>(late, nz);
      }
#pragma unroll
      for (int d = 0; d < DX; ++d) {
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < DX; ++e) acc = fmaf(sA[d * DX + e], xp[e], acc);
        if (MOVE && a) store_scoped(m_buf(t) + (int64_t)d * K + j, acc, sys);
        xn[d] = fmaf(f.q, nz[d], acc);
      }
      if (a) {
#pragma unroll
        for (int d = 0; d < DX; ++d) store_scoped(x_out + (int64_t)d * K + j, xn[d], sys);
        if (verify) {
          uint32_t h = row_check_init(t, (uint32_t)(f.offset + j));
#pragma unroll
          for (int d = 0; d < DX; ++d) h = row_check_mix(h, xn[d]);
          store_scoped_u32(chk_buf(t) + j, f.verify == 2 ? h ^ 1u : h, sys);
        }
      }
      float qsum = 0.0f;
      if (f.H) {
        for (int o = 0; o < f.dy; ++o) {
          float m = 0.0f;
#pragma unroll
          for (int e = 0; e < DX; ++e) m = fmaf(sH[o * DX + e], xn[e], m);
          const float z = (sY[o] - m) * rr;
          qsum = fmaf(z, z, qsum);
        }
      } else {
#pragma unroll
        for (int d = 0; d < DX; ++d) { const float z = (sY[d] - xn[d]) * rr; qsum = fmaf(z, z, qsum); }
      }
      const float lw = fmaf(-0.5f, qsum, lconst);
      if (a) store_scoped(lw_out + j, lw, sys);
#pragma unroll
      for (int k = 0; k < SPL; ++k) if (k == s) lw_own[k] = a ? lw : -INFINITY;
    }
  }
  if (blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(&f.ctrl[0], epoch + 2u * (unsigned)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if constexpr (MOVE) {
    if (f.acc_total) {
      const unsigned wacc = wave_scan_u32(acc_lane);     // lane 63: the wave's total (< 2^32: 64 lanes x SPL x T x n_moves)
      if (lane == 63 && wacc) atomicAdd(f.acc_total, (unsigned long long)wacc);
    }
  }
#undef jl
#undef act
#undef ton
}

// kernel of one (dx, spl) for this translation unit's RNG; NULL when the combination is not instantiated
template <int RNG>
const void* pf_kernel_of(int dx, int spl, bool move) {
#define GJX_PF(DXV, SPLV) if (dx == DXV && spl == SPLV) return move ? (const void*)k_pf_persistent<RNG, DXV, SPLV, true> : (const void*)k_pf_persistent<RNG, DXV, SPLV, false>;
  GJX_PF(2, 1) GJX_PF(2, 2) GJX_PF(2, 4) GJX_PF(2, 8)
  GJX_PF(4, 1) GJX_PF(4, 2) GJX_PF(4, 4) GJX_PF(4, 8)
  GJX_PF(8, 1) GJX_PF(8, 2) GJX_PF(8, 4) GJX_PF(8, 8) GJX_PF(8, 16)   // 16: config 4's dry run (8 ranks share ONE device's blocks)
  GJX_PF(16, 1) GJX_PF(16, 2) GJX_PF(16, 4)
#undef GJX_PF
  return nullptr;
}

}  // namespace gjx
