// gjx_ssm.hip — fused bootstrap-filter step for a linear-Gaussian state-space model
// (BASELINE configs 3/4): ancestor gather + propagate x_t ~ N(A x_{t-1}, q) + reweight
// log N(y_t; H x_t, r) + block {max, sum-exp} partials, one particle per lane.
// A, H, y are wave-uniform (scalar loads into SGPRs); the previous state is read through the
// ancestor index (systematic resampling yields monotone ancestors, so a wave's gathers touch a few
// adjacent cache lines per row); the new state is written as SoA rows (256 B per wave per row).
// Algorithmic HBM bytes per particle-step: 4 (ancestor) + 4*dx (gather) + 4*dx (state) + 4 (logw).
#include "gjx_device.h"
#include "gjx_host.h"
#include "gjx_scan.h"
#include "gjx_tile.h"
#include "gjx_pfilter_host.h"
#include <string.h>
// pollers per granule in the one-launch filter's rendezvous (tile-scaled scheme, up to 256 blocks) and the stagger between their
// first looks in units of s_sleep (64 cycles): see k_ssm_persistent
#ifndef GJX_POLLERS
#define GJX_POLLERS 4
#endif
#ifndef GJX_POLL_STAGGER
#define GJX_POLL_STAGGER 12
#endif
#include <vector>

namespace gjx {

struct SsmArgs {
  const float* A;
  const float* H;
  const float* y;
  float q, r, q0;
  int dy, t;
  key2 key;
  int64_t K, offset, prev_stride;
  const float* x_prev;
  const int32_t* anc;
  float* x_out;
  float* logw;
  unsigned long long* partials;
  unsigned* ticket;
  float* lse;
  float log_k_total;
  // resample-move (MOVE kernels): Metropolis-Hastings rejuvenation of the resampled x_{t-1} before it is propagated
  const float* m_prev;   // [DX][K] E[x_{t-1} | parent] of every particle of step t-1 (NULL at t-1 == 0: prior mean 0)
  float* m_out;          // [DX][K] A x_{t-1} of the particles of this step
  const float* y_prev;   // y_{t-1}
  int n_moves;
  float move_scale;
  float* accepted;       // [K] number of accepted moves (or NULL)
  float* x_moved;        // [DX][K] the moved x_{t-1} each slot was propagated from (or NULL)
};

template <int RNG, int DX, bool MOVE = false>
__global__ __launch_bounds__(256) void k_ssm_step(SsmArgs a) {
  __shared__ float red[16];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = i < a.K;
  const int64_t ii = active ? i : a.K - 1;
  const uint64_t gidx = (uint64_t)(a.offset + ii);
  key2 sk = a.key;  // JAX32: site key; FLAT: run key (high index word folded in when needed)
  if (RNG == GJX_RNG_JAX32) sk = fold_in(fold_in64(a.key, gidx), 1u);
  else if (gidx >> 32) sk = threefry2x32(a.key, 0xFFFFFFFFu, (uint32_t)(gidx >> 32));
  float xn[DX];
  if (a.t > 0) {
    // (ancestors left by a co-resident resampler that timed out are undefined — the caller repeats the run —: stay inside the rows)
    const int64_t src = a.anc ? (int64_t)max(0, min(a.anc[ii], (int32_t)(a.prev_stride - 1))) : ii;
    float xp[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) xp[d] = a.x_prev[(int64_t)d * a.prev_stride + src];
    if constexpr (MOVE) {
      // The resampled particle x_{t-1} (equal weights) is a draw from p(x_{t-1} | parent, y_{t-1}) up to the filter's
      // approximation; n_moves random-walk Metropolis steps with that conditional as invariant density restore the
      // diversity the resampling removed (the reference's ingredients: Regenerate / Rejuvenate + the caller-side accept
      // of tests/inference/test_requests.py:131-137; the kernel fuses proposal, both densities and the accept).
      // Draws: site 2 of this step's stream, element n (DX + 2) + d for the proposal, n (DX + 2) + DX for the uniform.
      float mp[DX];
#pragma unroll
      for (int d = 0; d < DX; ++d) mp[d] = a.m_prev ? a.m_prev[(int64_t)d * a.prev_stride + src] : 0.0f;
      const float sdp = a.t > 1 ? a.q : a.q0;
      const float rq = fast_rcp(sdp), rr0 = fast_rcp(a.r);
      auto logpi = [&](const float (&x)[DX]) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < DX; ++d) { const float z = (x[d] - mp[d]) * rq; s = fmaf(z, z, s); }
        if (a.H) {
          for (int o = 0; o < a.dy; ++o) {
            float m = 0.0f;
#pragma unroll
            for (int e = 0; e < DX; ++e) m = fmaf(a.H[o * DX + e], x[e], m);
            const float z = (a.y_prev[o] - m) * rr0;
            s = fmaf(z, z, s);
          }
        } else {
#pragma unroll
          for (int d = 0; d < DX; ++d) { const float z = (a.y_prev[d] - x[d]) * rr0; s = fmaf(z, z, s); }
        }
        return -0.5f * s;
      };
      BitStreamRT<RNG> bm;
      bm.open(a.key, gidx, 2u);
      float cur = logpi(xp), nacc = 0.0f;
      for (int n = 0; n < a.n_moves; ++n) {
        float xq[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) xq[d] = fmaf(a.move_scale, stream_normal<RNG>(bm, (uint32_t)(n * (DX + 2) + d)), xp[d]);
        const float prop = logpi(xq);
        const float lu = safe_log(uniform_from_bits(bm.get((uint32_t)(n * (DX + 2) + DX)), kTiny, 1.0f));
        if (lu < prop - cur) {
#pragma unroll
          for (int d = 0; d < DX; ++d) xp[d] = xq[d];
          cur = prop;
          nacc += 1.0f;
        }
      }
      if (a.accepted && active) a.accepted[i] = nacc;
      if (a.x_moved && active) {
#pragma unroll
        for (int d = 0; d < DX; ++d) a.x_moved[(int64_t)d * a.K + i] = xp[d];
      }
    }
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      float acc = 0.0f;
#pragma unroll
      for (int e = 0; e < DX; ++e) acc = fmaf(a.A[d * DX + e], xp[e], acc);
      xn[d] = acc;
    }
  } else {
#pragma unroll
    for (int d = 0; d < DX; ++d) xn[d] = 0.0f;
  }
  if (MOVE && a.m_out && active) {
#pragma unroll
    for (int d = 0; d < DX; ++d) a.m_out[(int64_t)d * a.K + i] = xn[d];
  }
  const float sd = a.t > 0 ? a.q : a.q0;
  if (RNG == GJX_RNG_FLAT) {
    // site 1 of the FLAT stream: DX elements of 23 bits (an odd DX reads the partner element DX of its last pair)
    constexpr int NE = DX + (DX & 1);
    constexpr int NB = GJX_FLAT_BLOCKS(NE);
    uint32_t w[2 * NB];
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      const key2 hh = threefry2x32(sk, (uint32_t)gidx, (1u << GJX_FLAT_SITE_SHIFT) | (uint32_t)h);
      w[2 * h] = hh.a; w[2 * h + 1] = hh.b;
    }
#pragma unroll
    for (int d0 = 0; d0 < DX; d0 += 2) {
      float n0, n1;
      box_muller(GJX_FIELD(w, d0), GJX_FIELD(w, d0 + 1), n0, n1);
      xn[d0] = fmaf(sd, n0, xn[d0]);
      if (d0 + 1 < DX) xn[d0 + 1] = fmaf(sd, n1, xn[d0 + 1]);
    }
  } else {
#pragma unroll
    for (int d0 = 0; d0 < DX; d0 += 2) {
      const key2 h0 = threefry2x32(sk, 0u, (uint32_t)d0);
      const uint32_t b0 = h0.a ^ h0.b;
      uint32_t b1 = 0u;
      if (d0 + 1 < DX) { const key2 h1 = threefry2x32(sk, 0u, (uint32_t)(d0 + 1)); b1 = h1.a ^ h1.b; }
      xn[d0] = fmaf(sd, normal_from_bits_fast(b0), xn[d0]);
      if (d0 + 1 < DX) xn[d0 + 1] = fmaf(sd, normal_from_bits_fast(b1), xn[d0 + 1]);
    }
  }
  if (active) {
#pragma unroll
    for (int d = 0; d < DX; ++d) a.x_out[(int64_t)d * a.K + i] = xn[d];
  }
  const float rr = fast_rcp(a.r);
  float qsum = 0.0f;
  if (a.H) {
    for (int o = 0; o < a.dy; ++o) {
      float m = 0.0f;
#pragma unroll
      for (int e = 0; e < DX; ++e) m = fmaf(a.H[o * DX + e], xn[e], m);
      const float z = (a.y[o] - m) * rr;
      qsum = fmaf(z, z, qsum);
    }
  } else {
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      const float z = (a.y[d] - xn[d]) * rr;
      qsum = fmaf(z, z, qsum);
    }
  }
  const float lw = fmaf(-0.5f, qsum, -(float)a.dy * (kHalfLog2Pi + fast_log(a.r)));
  if (active) a.logw[i] = lw;
  if (a.partials) {
    float bm, bsum;
    block_lse_partial<256>(lw, active, red, bm, bsum);
    if (a.lse) lse_publish_and_finish<256>(bm, bsum, a.partials, a.ticket, (int)gridDim.x, a.log_k_total, a.lse, red);
    else if (threadIdx.x == 0) a.partials[blockIdx.x] = pack_f2(bm, bsum);  // finished by gjx_weight_cumsum (is_log 2)
  }
}


// ------------------------------------------------------------------------------------------------------------
// One launch per filter step (t >= 1): resample + propagate + reweight with ONE rendezvous.
//
// The two-launch step (k_resample_fused, k_ssm_step) is latency-bound at K = 2^18: 10 us + 10 us for 23 MB.  Here
// block b scans its own tile of fixed-point weights, the tile totals are all-gathered (the one rendezvous), and then the
// block is the CONSUMER of output slots [256 b, 256 b + 256): every lane computes the comb threshold T_j of its slot,
// finds the source tile by binary search in the prefix of the tile totals (LDS), the block re-scans the few distinct
// source tiles its 256 thresholds fall into (their log-weights come from L2; one wave per tile), and each lane finds its
// ancestor by binary search in that tile's cumulative weights.  Ancestors never go to memory (unless asked for), the
// propagation is one slot per lane — balanced whatever the weights are — and the step's {max, sumexp} block partials
// are grouped by slot tile exactly as in k_ssm_step.  Same ancestors as the slot-run expansion of k_resample_fused:
// particle i owns slot j  <=>  cum_excl(i) <= T_j < cum_incl(i)  (slots_below counts the T_j below a boundary).
// A window whose thresholds touch more than kChunk distinct tiles (collapsed weights) walks them kChunk at a time.
// ------------------------------------------------------------------------------------------------------------
struct SsmFusedArgs {
  SsmArgs s;                       // model, key, K, x_prev / x_out, y, t; logw = output of THIS step
  const float* logw_prev;          // [K] log-weights of step t-1
  const float* partials_prev;      // per-block {max, sumexp} pairs of step t-1
  int n_partials_prev;
  float* lse_prev_out;             // [4] finished record of step t-1 (block 0)
  float log_k_total_prev;
  double u;
  int32_t* ancestors;              // [K] or NULL
  unsigned long long* agg;
  unsigned* ctrl;
  unsigned long long* timeline;    // debug (gjx_debug_timeline): 8 realtime stamps per block
};

constexpr int kSsmFusedMaxTiles = 2048;

template <int RNG, int DX>
__global__ __launch_bounds__(256) void k_ssm_fused_step(SsmFusedArgs f) {
#define GJX_STAMP(n) do { if (f.timeline && threadIdx.x == 0) f.timeline[blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  GJX_STAMP(0);
  const SsmArgs& a = f.s;
  constexpr int kChunk = 4;        // source tiles re-scanned together: one per wave
  __shared__ float fred[16];
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t P[kSsmFusedMaxTiles + 1];   // P[t] = total weight of tiles < t; P[nb] = grand total
  __shared__ uint64_t cumL[kChunk * 256];         // inclusive cumulative weights of the loaded tiles (absolute)
  __shared__ int32_t s_tof[256], s_tiles[256];
  __shared__ int s_cnt[4];
  unsigned epoch;
  const unsigned long long tag = grid_tag(f.ctrl, &epoch);
  const int64_t K = a.K;
  const int nb = (int)gridDim.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;     // the slot this lane produces (and the particle it scans)
  const bool active = j < K;
  float xv[1];
  xv[0] = active ? f.logw_prev[j] : 0.0f;
  float sm_sum;
  const float mx = block_ref_max(2, f.partials_prev, f.n_partials_prev, fred, &sm_sum);
  if (f.lse_prev_out && blockIdx.x == 0 && threadIdx.x == 0) {
    const float l = mx > -INFINITY ? mx + logf(sm_sum) : -INFINITY;
    f.lse_prev_out[0] = mx; f.lse_prev_out[1] = sm_sum; f.lse_prev_out[2] = l; f.lse_prev_out[3] = l - f.log_k_total_prev;
  }
  GJX_STAMP(1);
  // ---- own tile total -> all-gather ----
  {
    const uint64_t qv = active ? weight_q(xv, 0, 1, mx) : 0;
    const uint64_t wt = wave_sum_u64(qv);
    if (lane == 0) wsum[wid] = wt;
    __syncthreads();
    if (threadIdx.x == 0) grid_publish(f.agg, tag, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  }
  GJX_STAMP(2);
  grid_gather(f.agg, tag, f.ctrl, [&](int b, unsigned long long val) { P[b + 1] = val; });
  if (threadIdx.x == 0) P[0] = 0;
  __syncthreads();
  // ---- prefix of the tile totals, in place: lane t owns entries [t per, (t+1) per) ----
  {
    const int per = (nb + 255) >> 8;
    const int e0 = threadIdx.x * per, e1 = (e0 + per) < nb ? (e0 + per) : nb;
    uint64_t loc = 0;
    for (int e = e0; e < e1; ++e) loc += P[e + 1];
    uint64_t inc = wave_scan_u64(loc);
    __syncthreads();            // wsum of the publish above has been read by thread 0
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint64_t run = inc - loc;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    for (int e = e0; e < e1; ++e) { run += P[e + 1]; P[e + 1] = run; }
    __syncthreads();
  }
  const uint64_t total = P[nb];
  GJX_STAMP(3);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(&f.ctrl[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (total == 0) __hip_atomic_fetch_or(&f.ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- ancestor of slot j ----
  int64_t src = j;               // dead collection (total == 0): every slot keeps its own particle, flagged above
  if (total > 0) {               // block-uniform
    const double step = (double)total / (double)K;
    const uint64_t T = comb_threshold(active ? j : K - 1, f.u, step, total);
    int lo = 0, hi = nb - 1;     // first tile t with P[t + 1] > T
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (P[mid + 1] > T) hi = mid; else lo = mid + 1;
    }
    const int tile = lo;
    s_tof[threadIdx.x] = tile;
    __syncthreads();
    // distinct source tiles of the window (tile is non-decreasing in the slot index)
    const bool first = threadIdx.x == 0 || s_tof[threadIdx.x - 1] != tile;
    const unsigned long long bal = __ballot(first);
    if (lane == 0) s_cnt[wid] = __popcll(bal);
    __syncthreads();
    int kpos = __popcll(bal & ((2ull << lane) - 1ull)) - 1;      // index of this lane's tile in the distinct list
    for (int w = 0; w < wid; ++w) kpos += s_cnt[w];
    const int ntiles = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (first) s_tiles[kpos] = tile;
    GJX_STAMP(4);
    for (int c0 = 0; c0 < ntiles; c0 += kChunk) {
      __syncthreads();           // s_tiles written / previous chunk's cumL consumed
      if (c0 + wid < ntiles) {   // wave-uniform: this wave re-scans tile s_tiles[c0 + wid], 4 particles per lane
        const int tsrc = s_tiles[c0 + wid];
        const int64_t p0 = (int64_t)tsrc * 256 + lane * 4;
        float lw4[4];
        if (p0 + 4 <= K) {
          const float4 v = *(const float4*)(f.logw_prev + p0);
          lw4[0] = v.x; lw4[1] = v.y; lw4[2] = v.z; lw4[3] = v.w;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) lw4[k] = p0 + k < K ? f.logw_prev[p0 + k] : -INFINITY;
        }
        uint64_t qi[4], sacc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { sacc += p0 + k < K ? weight_q(lw4, k, 1, mx) : 0; qi[k] = sacc; }
        uint64_t inc = wave_scan_u64(sacc);
        const uint64_t base = P[tsrc] + (inc - sacc);
#pragma unroll
        for (int k = 0; k < 4; ++k) cumL[wid * 256 + lane * 4 + k] = base + qi[k];
      }
      __syncthreads();
      if (kpos >= c0 && kpos < c0 + kChunk) {
        const uint64_t* cm = cumL + (kpos - c0) * 256;
        int l2 = 0, h2 = 255;    // first particle p of the tile with cum_incl(p) > T
        while (l2 < h2) {
          const int mid = (l2 + h2) >> 1;
          if (cm[mid] > T) h2 = mid; else l2 = mid + 1;
        }
        src = (int64_t)tile * 256 + l2;
      }
    }
  } else {
    GJX_STAMP(4);
  }
  GJX_STAMP(5);
  // ---- propagate + reweight slot j from its ancestor (k_ssm_step's arithmetic, same streams) ----
  if (active && f.ancestors) f.ancestors[j] = (int32_t)src;
  if (!active) src = 0;
  if (src >= a.K) src = a.K - 1;    // only after a timed-out rendezvous (undefined totals): stay inside the rows
  const uint64_t gidx = (uint64_t)(a.offset + j);
  key2 skj = a.key;
  if (RNG == GJX_RNG_JAX32) skj = fold_in(fold_in64(a.key, gidx), 1u);
  else if (gidx >> 32) skj = threefry2x32(a.key, 0xFFFFFFFFu, (uint32_t)(gidx >> 32));
  const float sd = a.q;
  const float rr = fast_rcp(a.r);
  float xp[DX], xn[DX];
#pragma unroll
  for (int d = 0; d < DX; ++d) xp[d] = a.x_prev[(int64_t)d * a.prev_stride + src];
#pragma unroll
  for (int d = 0; d < DX; ++d) {
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < DX; ++e) acc = fmaf(a.A[d * DX + e], xp[e], acc);
    xn[d] = acc;
  }
  if (f.timeline) { asm volatile("" :: "v"(xn[0])); GJX_STAMP(6); }
  if (RNG == GJX_RNG_FLAT) {
    constexpr int NE = DX + (DX & 1);
    constexpr int NB = GJX_FLAT_BLOCKS(NE);
    uint32_t w[2 * NB];
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      const key2 hh = threefry2x32(skj, (uint32_t)gidx, (1u << GJX_FLAT_SITE_SHIFT) | (uint32_t)h);
      w[2 * h] = hh.a; w[2 * h + 1] = hh.b;
    }
#pragma unroll
    for (int d0 = 0; d0 < DX; d0 += 2) {
      float n0, n1;
      box_muller(GJX_FIELD(w, d0), GJX_FIELD(w, d0 + 1), n0, n1);
      xn[d0] = fmaf(sd, n0, xn[d0]);
      if (d0 + 1 < DX) xn[d0 + 1] = fmaf(sd, n1, xn[d0 + 1]);
    }
  } else {
#pragma unroll
    for (int d0 = 0; d0 < DX; ++d0) {
      const key2 h0 = threefry2x32(skj, 0u, (uint32_t)d0);
      xn[d0] = fmaf(sd, normal_from_bits_fast(h0.a ^ h0.b), xn[d0]);
    }
  }
  if (active) {
#pragma unroll
    for (int d = 0; d < DX; ++d) a.x_out[(int64_t)d * K + j] = xn[d];
  }
  if (f.timeline) { asm volatile("" :: "v"(xn[0])); GJX_STAMP(7); }
  float qsum = 0.0f;
  if (a.H) {
    for (int o = 0; o < a.dy; ++o) {
      float m = 0.0f;
#pragma unroll
      for (int e = 0; e < DX; ++e) m = fmaf(a.H[o * DX + e], xn[e], m);
      const float z = (a.y[o] - m) * rr;
      qsum = fmaf(z, z, qsum);
    }
  } else {
#pragma unroll
    for (int d = 0; d < DX; ++d) { const float z = (a.y[d] - xn[d]) * rr; qsum = fmaf(z, z, qsum); }
  }
  const float lw = fmaf(-0.5f, qsum, -(float)a.dy * (kHalfLog2Pi + fast_log(a.r)));
  if (active) a.logw[j] = lw;
  {
    float bm, bsum;
    block_lse_partial<256>(lw, active, fred, bm, bsum);
    if (a.lse) lse_publish_and_finish<256>(bm, bsum, a.partials, a.ticket, (int)gridDim.x, a.log_k_total, a.lse, fred);
    else if (threadIdx.x == 0) a.partials[blockIdx.x] = pack_f2(bm, bsum);
  }
  if (f.timeline && threadIdx.x == 0) f.timeline[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();   // end (reuses slot 1)
#undef GJX_STAMP
}


// ------------------------------------------------------------------------------------------------------------
// The whole filter in ONE launch (steps 1 .. T-1; step 0 is k_ssm_step).  No kernel boundary: a block keeps the
// log-weights of the slots it produced (they are the particles it scans next), the ancestors are found on the consumer
// side exactly as in k_ssm_fused_step, and the state and log-weights other blocks read cross the chip through
// write-through (sc1) stores and sc1 loads, ordered by `s_waitcnt vmcnt(0)` before a block publishes its granule.
// Ping-pong buffers are safe without further fences: a block can only overwrite the buffer of step t-1 in step t+1,
// after every block has published its granule of step t+1, i.e. has finished reading it.  The standard-normal draws of
// step t depend on (key_t, slot) only, not on the ancestor: they are generated while the granules travel.
//
// TILED == false — the fixed point of gjx_resample_indices (every weight against the exact GLOBAL maximum): the blocks
//   meet twice per step, all-gather of the block maxima of log w_{t-1}, then of the tile totals.  Same ancestors, same
//   streams, same values as the per-step kernels.
// TILED == true — TILE-SCALED fixed point (GJX_WEIGHTS_TILE_SCALED, include/gjx.h): a tile of kTileQ = 1024 consecutive
//   particles is quantised against its OWN power-of-two reference 2^e_b, e_b = ceil(max_tile(log w) * log2 e):
//   q_i = floor(2^29 * exp2(log w_i * log2 e - e_b)), S_b = sum of the tile's q_i.  One granule {e_b, S_b} per block, ONE
//   rendezvous per step; then E = max e_b, the tile counts G_b = S_b >> (E - e_b) units of 2^(E-29) on the global weight
//   line, comb thresholds T_j on the prefix of the G_b, and inside the source tile the residual (T_j - P_b) << (E - e_b)
//   is looked up in the tile's own cumulative q.  A tile loses less than one global unit (< 2^-28 of the largest
//   weight); nothing else is approximated.  Restated by the test oracle (gjxo_resample_systematic_tiled); the
//   multi-launch form is gjx_resample_indices_tiled (bit-identical ancestors).
// The LSE record of step t-1 needs the exact float maximum: blocks leave {block max, block sum-exp} in a 3-deep ring
// (TILED) and block (t-1) mod gridDim finishes the record inside the NEXT step's granule wait, off the critical path.
// ------------------------------------------------------------------------------------------------------------
struct SsmPersistArgs {
  const float* A; const float* H; const float* ys;   // ys [T][dy]
  float q, r;
  int dy, T;
  int64_t K;
  float* x_a; float* x_b;          // [DX][K]: step t writes x_b when t is odd
  float* lw_even; float* lw_odd;   // log-weights of step t in lw_odd when (T - 1 - t) is odd (the last step writes the caller's)
  const uint32_t* keys;            // [T][2] propagation key of every step
  const double* us;                // [T]    comb offset of every step
  float* lse_steps;                // [T][4]
  int32_t* ancestors;              // [K]: of the last step
  unsigned long long* aggA; unsigned long long* aggB;
  float* bsum;                     // [3][gridDim.x] per-block sum of exp(log w - block max) (ring over steps)
  float* bmax;                     // [3][gridDim.x] per-block max (TILED)
  unsigned* ready;                 // [gridDim.x] TILED: epoch + t once the block's stores of step t-1 have completed
  unsigned* ctrl;
  float log_k;
  unsigned long long* timeline;    // debug (gjx_debug_timeline): 16 realtime stamps per block for step T / 2
};

// granules of the persistent filter's rendezvous (both weight schemes) sit one per 64-byte line: 256 blocks storing into 32 shared lines serialise in the
// L2 (11.6 -> 11.0 us per filter step; 128-byte spacing and padding the `ready` words as well measured the same)
constexpr int kGranulePad = 8;
template <int RNG, int DX, int THREADS, bool TILED>
__global__ __launch_bounds__(THREADS) void k_ssm_persistent(SsmPersistArgs f) {
  // THREADS = 1024 (one block per CU, 256 blocks at K = 2^18) quarters the granules of each all-gather: a rendezvous
  // among 256 blocks measures 3.5 us, among 1024 blocks 5.4 us
  static_assert(!TILED || THREADS == kTileQ, "the tile-scaled fixed point quantises per 1024-particle tile");
  constexpr int NW = THREADS / 64;               // waves per block
  constexpr int WPT = THREADS / 256;             // waves that re-scan one source tile together (256 particles each)
  constexpr int kChunk = NW / WPT;               // source tiles re-scanned per round (= 4)
  constexpr int kPer = (kSsmFusedMaxTiles + THREADS - 1) / THREADS;
  __shared__ float fred[2 * NW];
  __shared__ float lse_pm[NW], lse_ps[NW];       // finisher: per-wave {max, sumexp} of the ring entries it read
  __shared__ uint64_t wsum[NW];
  __shared__ uint64_t P[kSsmFusedMaxTiles + 1];
  __shared__ int32_t Eb[TILED ? kSsmFusedMaxTiles : 1];
  __shared__ uint64_t cumL[kChunk * THREADS];
  // model constants in LDS: inside the step loop the compiler must assume the kernel's own stores may alias A, H, ys, us
  // and re-reads them with VECTOR loads every step (16 dependent global_load_dwordx4 for A alone: 1.6 us per step)
  __shared__ float sA[DX * DX], sH[kSsmPersistMaxDy * DX], sY[kSsmPersistMaxDy];
  __shared__ double sU;
  __shared__ uint32_t sKey[2][2];
  __shared__ int s_range[2];
  // TILED, nb <= 256: every granule has kPollers pollers — lanes b, b + 256, ... of the block — whose loads leave a quarter of a
  // round trip apart; whoever sees the tag first publishes the granule in LDS and sets seen[b], the others leave their loop at their
  // next look.  A poll is one fabric round trip (~1.7 us): with one poller a block learns of the last granule up to a round trip
  // after it landed, and that block is then the last to publish in the NEXT step — the skew feeds itself.
  __shared__ unsigned seen[256];
  for (int e = threadIdx.x; e < DX * DX; e += THREADS) sA[e] = f.A[e];
  if (f.H) for (int e = threadIdx.x; e < f.dy * DX; e += THREADS) sH[e] = f.H[e];
  const unsigned epoch = __hip_atomic_load(&f.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t K = f.K;
  const int nb = (int)gridDim.x, T = f.T;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * THREADS + threadIdx.x;  // the slot this lane produces = the particle it scans
  const bool active = j < K;
  auto lw_buf = [&](int t) { return ((T - 1 - t) & 1) ? f.lw_odd : f.lw_even; };
  auto x_buf = [&](int t) { return (t & 1) ? f.x_b : f.x_a; };
  const float rr = fast_rcp(f.r);
  const float lconst = -(float)f.dy * (kHalfLog2Pi + fast_log(f.r));
  float lw_own = active ? lw_buf(0)[j] : -INFINITY;               // step 0 ran in the previous launch
  // finisher of the LSE record of step s from ring slot s % 3 (TILED): all threads read, per-wave partials to LDS
  auto lse_ring_partials = [&](int s) {
    const float* rm = f.bmax + (size_t)(s % 3) * nb;
    const float* rs = f.bsum + (size_t)(s % 3) * nb;
    float m = -INFINITY, sm = 0.0f;
    for (int b = (threadIdx.x + THREADS / 2) % THREADS; b < nb; b += THREADS) {   // the non-polling half of the block first
      const float pm = load_agent(rm + b), ps = load_agent(rs + b);
      const float nm = fmaxf(m, pm);
      if (nm > -INFINITY) sm = sm * fast_exp(m - nm) + ps * fast_exp(pm - nm);
      m = nm;
    }
    const float wm = wave_max_dpp(m);
    const float ws = wave_sum_dpp(wm > -INFINITY ? sm * fast_exp(m - wm) : 0.0f);
    if (lane == 0) { lse_pm[wid] = wm; lse_ps[wid] = ws; }
  };
  // the same in two halves around other work: the loads go out (fixed trip count: they stay in flight, in registers) ...
  auto lse_ring_issue = [&](int s, float (&rpm)[kPer], float (&rps)[kPer]) {
    const float* rm = f.bmax + (size_t)(s % 3) * nb;
    const float* rs = f.bsum + (size_t)(s % 3) * nb;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int b = (threadIdx.x + THREADS / 2) % THREADS + k * THREADS;     // the non-polling half of the block first
      rpm[k] = b < nb ? load_agent(rm + b) : -INFINITY;
      rps[k] = b < nb ? load_agent(rs + b) : 0.0f;
    }
  };
  // ... and are reduced to per-wave partials in LDS (block-uniform call: every wave takes part)
  auto lse_ring_reduce = [&](const float (&rpm)[kPer], const float (&rps)[kPer]) {
    float m = -INFINITY, sm = 0.0f;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const float nm = fmaxf(m, rpm[k]);
      if (nm > -INFINITY) sm = sm * fast_exp(m - nm) + rps[k] * fast_exp(rpm[k] - nm);
      m = nm;
    }
    const float wm = wave_max_dpp(m);
    const float wsm = wave_sum_dpp(wm > -INFINITY ? sm * fast_exp(m - wm) : 0.0f);
    if (lane == 0) { lse_pm[wid] = wm; lse_ps[wid] = wsm; }
  };
  auto lse_ring_write = [&](int s) {             // thread 0, after a barrier behind lse_ring_partials(s)
    float m = lse_pm[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, lse_pm[w]);
    float se = 0.0f;
    for (int w = 0; w < NW; ++w) se += m > -INFINITY ? lse_ps[w] * fast_exp(lse_pm[w] - m) : 0.0f;
    const float l = m > -INFINITY ? m + logf(se) : -INFINITY;
    float* rec = f.lse_steps + 4 * (size_t)s;
    rec[0] = m; rec[1] = se; rec[2] = l; rec[3] = l - f.log_k;
  };
  // y_t and the comb offset of step t into LDS, by lanes of the second wave, while the granules travel (read several
  // barriers later; the previous step's values were last read before this step's first barrier)
  auto stage_step_constants = [&](int t) {
    if (t < T && wid == (NW > 1 ? 1 : 0)) {
      if (lane < f.dy) sY[lane] = f.ys[(size_t)t * f.dy + lane];
      if (lane == 63) sU = f.us[t];
      if (lane >= 61 && lane < 63 && t + 1 < T) sKey[(t + 1) & 1][lane - 61] = f.keys[2 * (t + 1) + (lane - 61)];   // next step's key
    }
  };
  if (threadIdx.x < 2) sKey[1][threadIdx.x] = f.keys[2 + threadIdx.x];        // step 1's key (T > 1)
  __syncthreads();
#define GJX_PSTAMP(n) do { if (f.timeline && t == T / 2 && threadIdx.x == 0) f.timeline[blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  for (int t = 1; t <= T; ++t) {
    // a rendezvous that timed out once (the grid is not co-resident: another kernel holds CUs) ends the launch: every block
    // sees the flag at its next step and leaves; the host repeats the run on the multi-launch path
    if (__hip_atomic_load(&f.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusPollTimeout) break;
    GJX_PSTAMP(0);
    SsmNoiseBits<RNG, DX> nbits;
    float mx = -INFINITY;
    int eb = kTileDead, Emax = kTileDead;
    unsigned rdy[kPer];
    const unsigned rtag = epoch + (unsigned)t;                // `ready` word of this step: never repeats, the epoch advances by 2 T per launch
    auto check_ready = [&]() {                                // every block's stores of step t-1 have completed (TILED)
      unsigned budget = kPollBudget;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int b = threadIdx.x + k * THREADS;
        if (b >= nb) break;
        unsigned r = rdy[k];
        while (r != rtag && budget) {
          __builtin_amdgcn_s_sleep(1);
          r = __hip_atomic_load(&f.ready[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          --budget;
        }
        if (r != rtag) __hip_atomic_fetch_or(&f.ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    // ---- block maximum of log w_{t-1} ----
    float bm;
    {
      const float wm = wave_max_dpp(active ? lw_own : -INFINITY);
      if (lane == 0) fred[wid] = wm;     // (fred[0..NW) was last read at least two barriers ago)
      __syncthreads();
      GJX_PSTAMP(7);         // every wave of the block has finished the previous step
      bm = fred[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) bm = fmaxf(bm, fred[w]);
    }
    if constexpr (!TILED) {
      // ---- rendezvous A: exact global maximum (nothing else rides on it: published as early as possible) ----
      const unsigned long long tagA = (unsigned long long)((epoch + 2u * (unsigned)(t - 1)) % 16383u) + 1ull;
      if (threadIdx.x == 0) grid_publish<kGranulePad>(f.aggA, tagA, (unsigned long long)__float_as_uint(bm));
      GJX_PSTAMP(1);
      // while the granules travel: this block's sum of exp(log w - block max), for the LSE record (read after
      // rendezvous B; ring of 2: a block rewrites its entry only after the finisher published its next granule A),
      // and the draws of step t
      {
        const float e = (active && bm > -INFINITY) ? fast_exp(lw_own - bm) : 0.0f;
        const float ws = wave_sum_dpp(e);
        if (lane == 0) fred[NW + wid] = ws;
        __syncthreads();
        if (threadIdx.x == 0) {
          float bs = 0.0f;
          for (int w = 0; w < NW; ++w) bs += fred[NW + w];
          const size_t slot = (size_t)((t - 1) % 3) * nb + blockIdx.x;   // ring over steps (complete before granule B goes out)
          store_agent(&f.bsum[slot], bs);
          store_agent(&f.bmax[slot], bm);
        }
      }
      stage_step_constants(t);
      // the record of step t-2, if this block is its finisher, off the critical path: ring loads out before the hashes,
      // reduced behind them, written by thread 0 behind the next barrier (every block's entry of step t-2 was complete
      // before its granule B of step t-1, which this block has gathered)
      const bool fin = t >= 2 && (int)blockIdx.x == (t - 2) % nb;
      float rpm[kPer], rps[kPer];
      if (fin) lse_ring_issue(t - 2, rpm, rps);
      if (t < T) ssm_noise_bits<RNG, DX>(key2{sKey[t & 1][0], sKey[t & 1][1]}, (uint64_t)j, nbits);
      if (fin) lse_ring_reduce(rpm, rps);
      grid_gather<kGranulePad>(f.aggA, tagA, f.ctrl, [&](int, unsigned long long v) { mx = fmaxf(mx, __uint_as_float((uint32_t)v)); });
      mx = wave_max_dpp(mx);     // (no acquire fence: everything read from other blocks goes through agent-scope loads)
      __syncthreads();
      if (lane == 0) fred[wid] = mx;
      __syncthreads();
      mx = fred[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) mx = fmaxf(mx, fred[w]);
      if (fin && threadIdx.x == 0) lse_ring_write(t - 2);
      GJX_PSTAMP(2);
      // ---- rendezvous B: tile totals of the fixed-point weights.  The granule goes out only after this lane's
      //      write-through (sc1) stores of step t-1 — x, log w, the block sum — have completed; readers use sc1 loads: no
      //      cache maintenance on either side (MI355X guide G16 form R2; a release fence per wave cost 60 us per step).
      //      At t == T only the ordering matters (the last LSE record is finished behind it). ----
      const unsigned long long tagB = (unsigned long long)((epoch + 2u * (unsigned)(t - 1) + 1u) % 16383u) + 1ull;
      {
        float xv[1] = {lw_own};
        const uint64_t qv = (active && t < T) ? weight_q(xv, 0, 1, mx) : 0;
        const uint64_t wt = wave_total_u64(qv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) wsum[wid] = wt;     // (wsum was last read in the previous step's re-scan, barriers ago)
        __syncthreads();
        if (wid == 0) {
          static_assert(NW <= 16, "the wave partials fit one DPP row");
          const uint64_t tt = row_scan_u64(lane < NW ? wsum[lane] : 0);   // lane 15 = sum of lanes 0..15
          if (lane == 15) __hip_atomic_store(&f.aggB[(size_t)blockIdx.x * kGranulePad], (tagB << 50) | (tt & kAggMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      GJX_PSTAMP(3);
      grid_gather<kGranulePad>(f.aggB, tagB, f.ctrl, [&](int b, unsigned long long val) { P[b + 1] = val; });
    } else {
      // ---- the ONE rendezvous: {e_b, S_b}.  The granule depends on registers only and goes out at once; that the block's
      //      sc1 stores of step t-1 (x, log w, its ring entry) have completed is signalled separately (`ready`), behind
      //      the granule and off the rendezvous' critical path: consumers look at it only right before their first
      //      foreign read, several microseconds later ----
      const unsigned long long tag = (unsigned long long)((epoch + (unsigned)t) % 15u) + 1ull;
      unsigned long long* agg = (t & 1) ? f.aggA : f.aggB;   // alternate: a slow block may still poll step t-1's granules
      eb = tile_exponent(bm);
      {
        const uint64_t qv = (active && t < T) ? tile_q(lw_own, eb) : 0;
        const float e = (active && bm > -INFINITY) ? fast_exp(lw_own - bm) : 0.0f;
        const uint64_t wt = wave_total_u64(qv);
        const float ws = wave_sum_dpp(e);
        if (lane == 0) { wsum[wid] = wt; fred[NW + wid] = ws; }
        __syncthreads();
        if (wid == 0) {                    // the NW wave partials, summed across the first lanes of wave 0
          static_assert(NW <= 16, "the wave partials fit one DPP row");
          uint64_t tt = row_scan_u64(lane < NW ? wsum[lane] : 0);       // lane 15 = sum of lanes 0..15
          float bs = row_sum_to_lane15(lane < NW ? fred[NW + lane] : 0.0f);
          if (lane == 15) {
            __hip_atomic_store(&agg[(size_t)blockIdx.x * kGranulePad], tile_granule(tag, eb, tt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const size_t slot = (size_t)((t - 1) % 3) * nb + blockIdx.x;
            store_agent(&f.bsum[slot], bs);
            store_agent(&f.bmax[slot], bm);
          }
        }
      }
      GJX_PSTAMP(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      constexpr int kPollers = (THREADS == 1024 && GJX_POLLERS > 1) ? GJX_POLLERS : 1;
      const bool multi = kPollers > 1 && nb <= 256;                      // block-uniform
      if (multi && threadIdx.x < 256) seen[threadIdx.x] = 0u;           // (the previous step's pollers are barriers behind; this step's start behind the next barrier)
      // the record of step t-2 if this block is its finisher: its ring entries were complete before the `ready` words
      // this block checked in step t-1; the loads go out here and are consumed behind the draws
      const bool fin = t >= 2 && (int)blockIdx.x == (t - 2) % nb;
      float rpm[kPer], rps[kPer];
      if (fin) lse_ring_issue(t - 2, rpm, rps);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(&f.ready[blockIdx.x], rtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stage_step_constants(t);
      // a first look at the granules goes out BEFORE the draws: for the block that is last to publish (the one everybody
      // else is waiting for, hence the one whose own chain sets the step time) they are all there already, and the load
      // latency hides behind the draws
      unsigned long long gv[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {         // (fixed trip count: the loads stay in flight, in registers)
        const int b = threadIdx.x + k * THREADS;
        gv[k] = b < nb ? __hip_atomic_load(&agg[(size_t)b * kGranulePad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
      // while the granules travel: the draws of step t.  (Looking at the missing granules again between the hashes was
      // measured and dropped: the polling waves then stall inside the draws, 13.5 -> 14.1 us per step.)
      if (t < T) ssm_noise_bits<RNG, DX>(key2{sKey[t & 1][0], sKey[t & 1][1]}, (uint64_t)j, nbits);
      if (fin) lse_ring_reduce(rpm, rps);
      GJX_PSTAMP(2);
      if (multi) {
        unsigned budget = kPollBudget;
        float em = (float)kTileDead;
        const int b = threadIdx.x & 255, grp = threadIdx.x >> 8;
        if (b < nb && grp < kPollers) {
          unsigned long long v = gv[0];
          if (grp > 0) {                                  // the later pollers: first look a fraction of a round trip behind the one before
            v = 0ull;
            __builtin_amdgcn_s_sleep(1);
            for (int w = 0; w < grp; ++w) __builtin_amdgcn_s_sleep(GJX_POLL_STAGGER);
            if (!__hip_atomic_load(&seen[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
              v = __hip_atomic_load(&agg[(size_t)b * kGranulePad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          bool other = false;
          while ((v >> 60) != tag && budget) {
            if (__hip_atomic_load(&seen[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { other = true; break; }
            --budget;
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&agg[(size_t)b * kGranulePad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if ((v >> 60) == tag) {
            const uint64_t S = v & ((1ull << 40) - 1);
            const int e = S ? (int)((v >> 40) & 0xFFFFFu) + kTileDead : kTileDead;
            P[b + 1] = S;
            Eb[b] = e;
            em = (float)e;
            __hip_atomic_store(&seen[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else if (!other && grp == 0) {                // (only the first poller reports: it runs the whole budget)
            __hip_atomic_fetch_or(&f.ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            P[b + 1] = 0;
            Eb[b] = kTileDead;
          }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          const int bb = threadIdx.x + k * THREADS;
          rdy[k] = bb < nb ? __hip_atomic_load(&f.ready[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rtag;
        }
        em = wave_max_dpp(em);
        if (lane == 0) fred[wid] = em;
      } else {
        unsigned budget = kPollBudget;
        float em = (float)kTileDead;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          const int b = threadIdx.x + k * THREADS;
          if (b >= nb) break;
          unsigned long long v = gv[k];
          while ((v >> 60) != tag && budget) {
            --budget;
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&agg[(size_t)b * kGranulePad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if ((v >> 60) != tag) { __hip_atomic_fetch_or(&f.ctrl[2], kStatusPollTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0; }
          const uint64_t S = v & ((1ull << 40) - 1);
          const int e = S ? (int)((v >> 40) & 0xFFFFFu) + kTileDead : kTileDead;
          P[b + 1] = S;
          Eb[b] = e;
          em = fmaxf(em, (float)e);
        }
        // the `ready` words: loads issued now, looked at after the tile search
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          const int b = threadIdx.x + k * THREADS;
          rdy[k] = b < nb ? __hip_atomic_load(&f.ready[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rtag;
        }
        em = wave_max_dpp(em);
        if (lane == 0) fred[wid] = em;   // (fred[0..NW) was last read for the block maximum, two barriers ago)
      }
    }
    if (threadIdx.x == 0) P[0] = 0;
    __syncthreads();
    if constexpr (TILED) {
      float em = fred[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) em = fmaxf(em, fred[w]);
      Emax = (int)em;
      if (t >= 2 && (int)blockIdx.x == (t - 2) % nb && threadIdx.x == 0) lse_ring_write(t - 2);
      GJX_PSTAMP(3);
    }
    if (wid == 0) {
      // prefix of the (shifted) tile totals by ONE wave — ceil(nb / 64) entries per lane, one DPP scan — instead of four
      // waves and a barrier between their partial sums
      const int per = (nb + 63) >> 6;
      const int e0 = lane * per < nb ? lane * per : nb, e1 = (e0 + per) < nb ? (e0 + per) : nb;
      uint64_t loc = 0;
      for (int e = e0; e < e1; ++e) {
        if constexpr (TILED) { const int sh = Emax - Eb[e]; P[e + 1] = sh < 64 ? P[e + 1] >> sh : 0; }
        loc += P[e + 1];
      }
      uint64_t run = wave_scan_u64(loc) - loc;
      for (int e = e0; e < e1; ++e) { run += P[e + 1]; P[e + 1] = run; }
    }
    __syncthreads();
    const uint64_t total = P[nb];
    GJX_PSTAMP(4);
    if constexpr (!TILED) {
      if (t == T && (int)blockIdx.x == (T - 1) % nb) {             // the last record: its ring entries preceded granule B of this step
        lse_ring_partials(T - 1);
        __syncthreads();
        if (threadIdx.x == 0) lse_ring_write(T - 1);
      }
    } else if (t == T) {                                            // the last record, once every block's ring entry is complete
      check_ready();
      __syncthreads();
      if ((int)blockIdx.x == (T - 1) % nb) {
        lse_ring_partials(T - 1);
        __syncthreads();
        if (threadIdx.x == 0) lse_ring_write(T - 1);
      }
    }
    if (t == T) break;
    if (total == 0 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_or(&f.ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- ancestor of slot j (k_ssm_fused_step's search; foreign log-weights through agent-scope loads) ----
    const float* lw_prev = lw_buf(t - 1);
    int64_t src = j;
    if (total > 0) {
      const double step = (double)total / (double)K;
      // the slot's source tile: first tile t with P[t + 1] > T = number of tiles whose inclusive prefix is <= T (fixed-trip
      // descent in the LDS prefix).  The tile index is non-decreasing in the slot index, so the block's source tiles are
      // the range between its first and its last slot's tile: no list to build, the two ends travel through LDS behind
      // the one barrier the `ready` check needs anyway
      uint64_t Tj = comb_threshold(active ? j : K - 1, sU, step, total);
      // A block's slots draw from tiles near its own index (equal tile totals would make it exactly its own): the nine
      // boundaries of the eight tiles around blockIdx.x are read together (one LDS latency, same addresses in every lane)
      // and the tile is counted off; a threshold outside that window takes the fixed-trip descent (nine dependent reads)
      const int wlo = (int)blockIdx.x - 4 < 0 ? 0 : ((int)blockIdx.x - 4 > nb - 8 ? (nb - 8 < 0 ? 0 : nb - 8) : (int)blockIdx.x - 4);
      uint64_t Pw[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) Pw[k] = P[wlo + k < nb ? wlo + k : nb];
      int tile = wlo;
      if (Tj >= Pw[0] && Tj < Pw[8]) {
#pragma unroll
        for (int k = 1; k < 8; ++k) tile += Pw[k] <= Tj ? 1 : 0;
      } else {
        tile = 0;
        for (int sft = 1 << (31 - __builtin_clz((unsigned)nb)); sft >= 1; sft >>= 1) {
          const int p = tile + sft;                                     // P[p] = inclusive prefix of tile p - 1
          if (p <= nb - 1 && P[p] <= Tj) tile = p;
        }
      }
      if (threadIdx.x == 0) s_range[0] = tile;
      if (threadIdx.x == THREADS - 1) s_range[1] = tile;                // (inactive lanes searched slot K - 1)
      if constexpr (TILED) {
        Tj = (Tj - P[tile]) << (Emax - Eb[tile]);                      // residual in the source tile's own units (< S_tile)
        check_ready();                                                // before the barrier in front of the first foreign read
      }
      GJX_PSTAMP(9);
      __syncthreads();
      const int tmin = s_range[0], ntiles = s_range[1] - tmin + 1;    // (tiles without weight inside the range are scanned for nothing)
      const int kpos = tile - tmin;
      bool again = false;
      for (int c0 = 0; c0 < ntiles; c0 += kChunk) {
        // a round starts at a tile that has weight (block-uniform; the range's last tile always has): collapsed weights
        // far apart cost as many rounds as there are live tiles, not as the range is long
        while (P[tmin + c0 + 1] == P[tmin + c0]) ++c0;
        if (again) __syncthreads();                                   // previous round's cumL consumed
        again = true;
        const int tl = wid / WPT, part = wid % WPT;               // this wave: quarter `part` of source tile c0 + tl
        const bool on = c0 + tl < ntiles;
        uint64_t qi[4], sacc = 0, inc = 0;
        int tsrc = 0;
        if (on) {
          tsrc = tmin + c0 + tl;
          const int64_t p0 = (int64_t)tsrc * THREADS + part * 256 + lane * 4;
          float lw4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) lw4[k] = -INFINITY;
          if (p0 + 4 <= K) load_agent_x4(lw_prev + p0, lw4);            // (p0 is a multiple of 4, the buffer 16-byte aligned)
          else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (p0 + k < K) lw4[k] = load_agent(lw_prev + p0 + k);
          }
          if (f.timeline && c0 == 0) { asm volatile("" :: "v"(lw4[0]), "v"(lw4[3])); GJX_PSTAMP(10); }
          const int es = TILED ? Eb[tsrc] : 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if constexpr (TILED) sacc += p0 + k < K ? tile_q(lw4[k], es) : 0;
            else sacc += p0 + k < K ? weight_q(lw4, k, 1, mx) : 0;
            qi[k] = sacc;
          }
          inc = wave_scan_u64(sacc);
          if (WPT > 1 && lane == 63) wsum[wid] = inc;             // the wave's total: offset of the next quarter
        }
        if (WPT > 1) __syncthreads();
        if (on) {
          uint64_t base = (TILED ? 0 : P[tsrc]) + (inc - sacc);
          for (int w = 0; w < part; ++w) base += wsum[tl * WPT + w];
#pragma unroll
          for (int k = 0; k < 4; ++k) cumL[tl * THREADS + part * 256 + lane * 4 + k] = base + qi[k];
        }
        __syncthreads();
        if (c0 == 0) GJX_PSTAMP(11);
        if (kpos >= c0 && kpos < c0 + kChunk) {
          const uint64_t* cm = cumL + (kpos - c0) * THREADS;
          // first particle whose cumulative weight exceeds the threshold = number of entries <= it: a 4-ary descent, three
          // independent probes per level (5 LDS latencies for 1024 entries instead of 10)
          int l2 = 0;
#pragma unroll
          for (int q = THREADS >> 2; q >= 1; q >>= 2) {
            const uint64_t pa = cm[l2 + q - 1], pb = cm[l2 + 2 * q - 1], pc = cm[l2 + 3 * q - 1];
            l2 += (pa <= Tj ? q : 0) + (pb <= Tj ? q : 0) + (pc <= Tj ? q : 0);
          }
          src = (int64_t)tile * THREADS + l2;
        }
      }
    } else if constexpr (TILED) {
      check_ready();              // (a dead step reads nothing foreign, but the ring entries count on every step's check)
    }
    GJX_PSTAMP(5);
    if (active && t == T - 1 && f.ancestors) f.ancestors[j] = (int32_t)src;
    if (!active) src = 0;
    // ---- propagate + reweight slot j (k_ssm_step's arithmetic and streams) ----
    const float* x_prev = x_buf(t - 1);
    float* x_out = x_buf(t);
    float xp[DX], xn[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) xp[d] = load_agent(x_prev + (int64_t)d * K + src);
    float nz[DX];
    ssm_noise_normals<RNG, DX>(nbits, nz);          // behind the loads above
    if (f.timeline) { asm volatile("" :: "v"(xp[0]), "v"(xp[DX - 1])); GJX_PSTAMP(8); }
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      float acc = 0.0f;
#pragma unroll
      for (int e = 0; e < DX; ++e) acc = fmaf(sA[d * DX + e], xp[e], acc);
      xn[d] = fmaf(f.q, nz[d], acc);
    }
    if (active) {
#pragma unroll
      for (int d = 0; d < DX; ++d) store_agent(x_out + (int64_t)d * K + j, xn[d]);
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
    if (active) store_agent(lw_buf(t) + j, lw);
    lw_own = active ? lw : -INFINITY;
    GJX_PSTAMP(6);
  }
#undef GJX_PSTAMP
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(&f.ctrl[0], epoch + 2u * (unsigned)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the tile-scaled systematic resampler as three plain launches (gjx_resample_indices_tiled): the step-by-step
//      form of what k_ssm_persistent<TILED> does inside its loop; bit-identical ancestors ----
__global__ __launch_bounds__(kTileQ) void k_tiled_quantise(const float* logw, int64_t K, uint64_t* cq, uint64_t* S, int32_t* E,
                                                           uint32_t* q_out, int32_t* e_out) {
  constexpr int NW = kTileQ / 64;
  __shared__ float fred[NW];
  __shared__ uint64_t wsum[NW];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kTileQ + threadIdx.x;
  const bool active = i < K;
  const float lw = active ? logw[i] : -INFINITY;
  const float wm = wave_max(lw);
  if (lane == 0) fred[wid] = wm;
  __syncthreads();
  float bm = fred[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) bm = fmaxf(bm, fred[w]);
  const int e = tile_exponent(bm);
  const uint64_t q = active ? tile_q(lw, e) : 0;
  uint64_t inc = wave_scan_u64(q);
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint64_t base = 0, tot = 0;
  for (int w = 0; w < NW; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
  if (active) { cq[i] = base + inc; if (q_out) q_out[i] = (uint32_t)q; }
  if (threadIdx.x == 0) {
    const int es = tot ? e : kTileDead;
    S[blockIdx.x] = tot; E[blockIdx.x] = es;
    if (e_out) e_out[blockIdx.x] = es;
  }
}

__global__ __launch_bounds__(1024) void k_tiled_plan(const uint64_t* S, const int32_t* E, int nt, uint64_t* P, int32_t* sh, unsigned* ctrl) {
  __shared__ float fred[16];
  __shared__ uint64_t wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float em = (float)kTileDead;
  for (int b = threadIdx.x; b < nt; b += 1024) em = fmaxf(em, (float)E[b]);
  em = wave_max(em);
  if (lane == 0) fred[wid] = em;
  __syncthreads();
  em = fred[0];
  for (int w = 1; w < 16; ++w) em = fmaxf(em, fred[w]);
  const int Emax = (int)em;
  uint64_t carry = 0;
  if (threadIdx.x == 0) P[0] = 0;
  for (int b0 = 0; b0 < nt; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    uint64_t g = 0;
    if (b < nt) {
      const int s = Emax - E[b];
      g = s < 64 ? S[b] >> s : 0;
      sh[b] = s < 64 ? s : 64;
    }
    uint64_t inc = wave_scan_u64(g);
    __syncthreads();
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint64_t base = carry, tot = 0;
    for (int w = 0; w < 16; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
    if (b < nt) P[b + 1] = base + inc;
    carry += tot;
  }
  if (threadIdx.x == 0 && carry == 0 && ctrl) __hip_atomic_fetch_or(&ctrl[2], kStatusZeroTotal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_tiled_ancestors(const uint64_t* P, const int32_t* sh, const uint64_t* cq, int nt, int64_t K,
                                                         double u, int64_t N, int32_t* anc) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const uint64_t total = P[nt];
  if (total == 0) { anc[j] = (int32_t)(j < K ? j : K - 1); return; }   // dead collection: identity, flagged by the plan
  const double step = (double)total / (double)N;
  const uint64_t Tj = comb_threshold(j, u, step, total);
  int lo = 0, hi = nt - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (P[mid + 1] > Tj) hi = mid; else lo = mid + 1;
  }
  const uint64_t r = (Tj - P[lo]) << sh[lo];
  const uint64_t* cm = cq + (int64_t)lo * kTileQ;
  const int64_t left = K - (int64_t)lo * kTileQ;
  int l2 = 0, h2 = (int)(left < kTileQ ? left : kTileQ) - 1;
  while (l2 < h2) {
    const int mid = (l2 + h2) >> 1;
    if (cm[mid] > r) h2 = mid; else l2 = mid + 1;
  }
  anc[j] = (int32_t)((int64_t)lo * kTileQ + l2);
}

// ---- multinomial resampling by SORTED uniforms on the tile-scaled weight line (gjx_resample_sorted_multinomial_tiled): slot j's
//      threshold is floor(total * S_j / S_N+1) with S_j the running sum of the exponential spacings exp_spacing(bits(key, j)) — the
//      order statistics of N iid uniforms, so the ancestors are a multinomial draw delivered in non-decreasing order.  The spacings
//      are exact integers (gjx_device.h): their sums do not depend on how the slots are cut into tiles. ----
__global__ __launch_bounds__(kTileQ) void k_spacing_totals(key2 key, int64_t N, uint64_t* SS) {
  constexpr int NW = kTileQ / 64;
  __shared__ uint64_t wsum[NW];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * kTileQ + threadIdx.x;
  uint64_t e = j <= N ? exp_spacing(fold_in64(key, (uint64_t)j).a) : 0ull;     // (slot N: the spacing that closes the unit interval)
  e = wave_scan_u64(e);
  if (lane == 63) wsum[wid] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t tot = 0;
    for (int w = 0; w < NW; ++w) tot += wsum[w];
    SS[blockIdx.x] = tot;
  }
}

__global__ __launch_bounds__(1024) void k_spacing_plan(const uint64_t* SS, int nb, uint64_t* SP) {
  __shared__ uint64_t wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t carry = 0;
  if (threadIdx.x == 0) SP[0] = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const uint64_t inc = wave_scan_u64(b < nb ? SS[b] : 0ull);
    __syncthreads();
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint64_t base = carry, tot = 0;
    for (int w = 0; w < 16; ++w) { if (w < wid) base += wsum[w]; tot += wsum[w]; }
    if (b < nb) SP[b + 1] = base + inc;
    carry += tot;
  }
}

__global__ __launch_bounds__(kTileQ) void k_sorted_ancestors(const uint64_t* P, const int32_t* sh, const uint64_t* cq, int nt, int64_t K, key2 key,
                                                             const uint64_t* SP, int nb, int64_t N, int32_t* anc) {
  constexpr int NW = kTileQ / 64;
  __shared__ uint64_t wsum[NW];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * kTileQ + threadIdx.x;
  uint64_t e = j < N ? exp_spacing(fold_in64(key, (uint64_t)j).a) : 0ull;
  e = wave_scan_u64(e);
  if (lane == 63) wsum[wid] = e;
  __syncthreads();
  uint64_t Sj = SP[blockIdx.x] + e;
  for (int w = 0; w < wid; ++w) Sj += wsum[w];
  if (j >= N) return;
  const uint64_t total = P[nt];
  if (total == 0) { anc[j] = (int32_t)(j < K ? j : K - 1); return; }   // dead collection: identity, flagged by the plan
  const uint64_t Tj = sorted_threshold(Sj, SP[nb], total);
  int lo = 0, hi = nt - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (P[mid + 1] > Tj) hi = mid; else lo = mid + 1;
  }
  const uint64_t r = (Tj - P[lo]) << sh[lo];
  const uint64_t* cm = cq + (int64_t)lo * kTileQ;
  const int64_t left = K - (int64_t)lo * kTileQ;
  int l2 = 0, h2 = (int)(left < kTileQ ? left : kTileQ) - 1;
  while (l2 < h2) {
    const int mid = (l2 + h2) >> 1;
    if (cm[mid] > r) h2 = mid; else l2 = mid + 1;
  }
  anc[j] = (int32_t)((int64_t)lo * kTileQ + l2);
}

int launch_tiled_plan(const uint64_t* S, const int32_t* E, int nt, uint64_t* P, int32_t* sh, unsigned* ctrl, hipStream_t st) {
  hipLaunchKernelGGL(k_tiled_plan, dim3(1), dim3(1024), 0, st, S, E, nt, P, sh, ctrl);
  GJX_CHECK_LAUNCH("k_tiled_plan");
  return GJX_OK;
}

}  // namespace gjx

using namespace gjx;

namespace {
template <int RNG>
int launch_ssm_move(const SsmArgs& a, int dx, int nblocks, hipStream_t st) {
  switch (dx) {
    case 2: hipLaunchKernelGGL((k_ssm_step<RNG, 2, true>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 4: hipLaunchKernelGGL((k_ssm_step<RNG, 4, true>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 8: hipLaunchKernelGGL((k_ssm_step<RNG, 8, true>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 16: hipLaunchKernelGGL((k_ssm_step<RNG, 16, true>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    default: return -1;
  }
}
template <int RNG>
int launch_ssm(const SsmArgs& a, int dx, int nblocks, hipStream_t st) {
  switch (dx) {
    case 1: hipLaunchKernelGGL((k_ssm_step<RNG, 1>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 2: hipLaunchKernelGGL((k_ssm_step<RNG, 2>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 4: hipLaunchKernelGGL((k_ssm_step<RNG, 4>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 8: hipLaunchKernelGGL((k_ssm_step<RNG, 8>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 16: hipLaunchKernelGGL((k_ssm_step<RNG, 16>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    case 32: hipLaunchKernelGGL((k_ssm_step<RNG, 32>), dim3(nblocks), dim3(256), 0, st, a); return 0;
    default: return -1;
  }
}
}  // namespace

extern "C" int gjx_ssm_step(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t t, int64_t K,
                            int64_t particle_offset, const float* x_prev, int64_t prev_stride, const int32_t* anc,
                            const float* y_dev, float* x_out, float* logw, float* lse, int64_t K_total,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !m->A_dev || !y_dev || !x_out || !logw || K <= 0 || t < 0) return gjx_fail(GJX_EINVAL, "gjx_ssm_step: bad argument");
  if (t > 0 && !x_prev) return gjx_fail(GJX_EINVAL, "gjx_ssm_step: x_prev is null for t > 0");
  if (!m->H_dev && m->dy != m->dx) return gjx_fail(GJX_EINVAL, "gjx_ssm_step: H == NULL needs dy == dx");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* partials = nullptr;
  unsigned* ticket = nullptr;
  if (lse || workspace) {  // lse == NULL with a workspace: leave ceil(K/256) per-block partial pairs at workspace + 256
    if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_SSM, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_ssm_step: workspace too small");
    ticket = (unsigned*)workspace;
    partials = (unsigned long long*)((char*)workspace + kWsHeaderBytes);
  }
  SsmArgs a;
  a.A = m->A_dev; a.H = m->H_dev; a.y = y_dev; a.q = m->q; a.r = m->r; a.q0 = m->q0; a.dy = m->dy; a.t = t;
  a.key = key2{key0, key1}; a.K = K; a.offset = particle_offset; a.prev_stride = prev_stride;
  a.x_prev = x_prev; a.anc = anc; a.x_out = x_out; a.logw = logw; a.partials = partials; a.ticket = ticket; a.lse = lse;
  a.log_k_total = (float)log((double)K_total);
  a.m_prev = nullptr; a.m_out = nullptr; a.y_prev = nullptr; a.n_moves = 0; a.move_scale = 0.0f; a.accepted = nullptr; a.x_moved = nullptr;
  const int nblocks = (int)((K + 255) / 256);
  const int rc = rng_mode == GJX_RNG_JAX32 ? launch_ssm<GJX_RNG_JAX32>(a, m->dx, nblocks, st)
                                           : launch_ssm<GJX_RNG_FLAT>(a, m->dx, nblocks, st);
  if (rc) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_step: dx must be one of 1,2,4,8,16,32");
  GJX_CHECK_LAUNCH("gjx_ssm_step");
  return GJX_OK;
}

extern "C" int gjx_ssm_step_move(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t t, int64_t K,
                                 int64_t particle_offset, const float* x_prev, const float* m_prev, int64_t prev_stride,
                                 const int32_t* anc, const float* y_prev_dev, const float* y_dev, int32_t n_moves,
                                 float move_scale, float* x_out, float* m_out, float* logw, float* accepted, float* x_moved_out,
                                 float* lse, int64_t K_total, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !m->A_dev || !y_dev || !x_out || !m_out || !logw || K <= 0 || t < 0 || n_moves < 0)
    return gjx_fail(GJX_EINVAL, "gjx_ssm_step_move: bad argument");
  if (t > 0 && (!x_prev || !y_prev_dev)) return gjx_fail(GJX_EINVAL, "gjx_ssm_step_move: x_prev / y_prev are null for t > 0");
  if (t > 1 && !m_prev) return gjx_fail(GJX_EINVAL, "gjx_ssm_step_move: m_prev is null for t > 1");
  if (!m->H_dev && m->dy != m->dx) return gjx_fail(GJX_EINVAL, "gjx_ssm_step_move: H == NULL needs dy == dx");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* partials = nullptr;
  unsigned* ticket = nullptr;
  if (lse || workspace) {
    if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_SSM, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_ssm_step_move: workspace too small");
    ticket = (unsigned*)workspace;
    partials = (unsigned long long*)((char*)workspace + kWsHeaderBytes);
  }
  SsmArgs a;
  a.A = m->A_dev; a.H = m->H_dev; a.y = y_dev; a.q = m->q; a.r = m->r; a.q0 = m->q0; a.dy = m->dy; a.t = t;
  a.key = key2{key0, key1}; a.K = K; a.offset = particle_offset; a.prev_stride = prev_stride;
  a.x_prev = x_prev; a.anc = anc; a.x_out = x_out; a.logw = logw; a.partials = partials; a.ticket = ticket; a.lse = lse;
  a.log_k_total = (float)log((double)K_total);
  a.m_prev = t > 1 ? m_prev : nullptr; a.m_out = m_out; a.y_prev = y_prev_dev; a.n_moves = t > 0 ? n_moves : 0;
  a.move_scale = move_scale; a.accepted = accepted; a.x_moved = t > 0 ? x_moved_out : nullptr;
  const int nblocks = (int)((K + 255) / 256);
  const int rc = rng_mode == GJX_RNG_JAX32 ? launch_ssm_move<GJX_RNG_JAX32>(a, m->dx, nblocks, st)
                                           : launch_ssm_move<GJX_RNG_FLAT>(a, m->dx, nblocks, st);
  if (rc) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_step_move: dx must be one of 2,4,8,16");
  GJX_CHECK_LAUNCH("gjx_ssm_step_move");
  return GJX_OK;
}


// ---- whole bootstrap filter on one GPU: the T-step loop runs in C++ so that the per-step launches
// (one-launch resampling indices, fused step) are issued back to back without Python in between.
// Key discipline as in inference/pf.py: k_t = fold_in(k_{t-1}, t) (scan.py:268); (k_prop, k_res) = split(k_t);
// the systematic comb offset is uniform(k_res).
extern "C" int gjx_weight_cumsum(const float*, int64_t, int32_t, const float*, int32_t, uint64_t*, uint64_t*, float*, int64_t, void*, size_t, void*);
extern "C" int gjx_resample_systematic(const uint64_t*, int64_t, const uint64_t*, double, int64_t, int64_t, int64_t, int32_t*, void*);
extern "C" int gjx_resample_indices(const float*, int64_t, int32_t, const float*, int32_t, double, int64_t, int32_t*, uint64_t*, uint64_t*, float*, int64_t, void*, size_t, void*);

static void host_threefry(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int g = 0; g < 5; ++g) {
    const int* r = (g & 1) ? R + 4 : R;
    for (int j = 0; j < 4; ++j) { x0 += x1; x1 = (x1 << r[j]) | (x1 >> (32 - r[j])); x1 ^= x0; }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0; out[1] = x1;
}

extern "C" int gjx_resample_indices_tiled(const float* logw, int64_t K, double u, int64_t N, int32_t* ancestors, uint64_t* cum,
                                          uint32_t* q_out, int32_t* e_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!logw || !ancestors || !cum || K <= 0 || N <= 0 || K > (int64_t)1 << 31) return gjx_fail(GJX_EINVAL, "gjx_resample_indices_tiled: bad argument");
  const int64_t nt = (K + kTileQ - 1) / kTileQ;
  // workspace: [256 B control][S u64 nt][P u64 nt + 1][E i32 nt][shift i32 nt]
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K) || 256 + 24 * (size_t)nt + 8 > workspace_bytes)
    return gjx_fail(GJX_EWORKSPACE, "gjx_resample_indices_tiled: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  uint64_t* S = (uint64_t*)((char*)workspace + kWsHeaderBytes);
  uint64_t* P = S + nt;
  int32_t* E = (int32_t*)(P + nt + 1);
  int32_t* sh = E + nt;
  hipLaunchKernelGGL(k_tiled_quantise, dim3((unsigned)nt), dim3(kTileQ), 0, st, logw, K, cum, S, E, q_out, e_out);
  hipLaunchKernelGGL(k_tiled_plan, dim3(1), dim3(1024), 0, st, (const uint64_t*)S, (const int32_t*)E, (int)nt, P, sh, (unsigned*)workspace + 8);
  hipLaunchKernelGGL(k_tiled_ancestors, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (const uint64_t*)P, (const int32_t*)sh,
                     (const uint64_t*)cum, (int)nt, K, u, N, ancestors);
  GJX_CHECK_LAUNCH("gjx_resample_indices_tiled");
  return GJX_OK;
}

extern "C" int gjx_resample_sorted_multinomial_tiled(const float* logw, int64_t K, uint32_t key0, uint32_t key1, int64_t N, int32_t* ancestors,
                                                     uint64_t* cum, uint32_t* q_out, int32_t* e_out, void* workspace, size_t workspace_bytes,
                                                     void* stream) {
  if (!logw || !ancestors || !cum || K <= 0 || N <= 0 || K > (int64_t)1 << 31 || N > (int64_t)1 << 31)
    return gjx_fail(GJX_EINVAL, "gjx_resample_sorted_multinomial_tiled: bad argument");
  const int64_t nt = (K + kTileQ - 1) / kTileQ;
  const int64_t nb = (N + 1 + kTileQ - 1) / kTileQ;        // tiles of SLOTS (N + 1 spacings)
  // workspace: [256 B control][S u64 nt][P u64 nt + 1][E i32 nt][shift i32 nt][SS u64 nb][SP u64 nb + 1]
  const size_t off_ss = (256 + 24 * (size_t)nt + 8 + 7) & ~(size_t)7;
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RESAMPLE, K) || off_ss + 16 * (size_t)nb + 8 > workspace_bytes)
    return gjx_fail(GJX_EWORKSPACE, "gjx_resample_sorted_multinomial_tiled: workspace too small (OP_RESAMPLE of K, and 288 + 24 ceil(K / 1024) + 16 ceil((N + 1) / 1024) bytes)");
  hipStream_t st = (hipStream_t)stream;
  uint64_t* S = (uint64_t*)((char*)workspace + kWsHeaderBytes);
  uint64_t* P = S + nt;
  int32_t* E = (int32_t*)(P + nt + 1);
  int32_t* sh = E + nt;
  uint64_t* SS = (uint64_t*)((char*)workspace + off_ss);
  uint64_t* SP = SS + nb;
  const key2 key{key0, key1};
  hipLaunchKernelGGL(k_tiled_quantise, dim3((unsigned)nt), dim3(kTileQ), 0, st, logw, K, cum, S, E, q_out, e_out);
  hipLaunchKernelGGL(k_tiled_plan, dim3(1), dim3(1024), 0, st, (const uint64_t*)S, (const int32_t*)E, (int)nt, P, sh, (unsigned*)workspace + 8);
  hipLaunchKernelGGL(k_spacing_totals, dim3((unsigned)nb), dim3(kTileQ), 0, st, key, N, SS);
  hipLaunchKernelGGL(k_spacing_plan, dim3(1), dim3(1024), 0, st, (const uint64_t*)SS, (int)nb, SP);
  hipLaunchKernelGGL(k_sorted_ancestors, dim3((unsigned)((N + kTileQ - 1) / kTileQ)), dim3(kTileQ), 0, st, (const uint64_t*)P, (const int32_t*)sh,
                     (const uint64_t*)cum, (int)nt, K, key, (const uint64_t*)SP, (int)nb, N, ancestors);
  GJX_CHECK_LAUNCH("gjx_resample_sorted_multinomial_tiled");
  return GJX_OK;
}


// ---- k_pf_persistent on one GPU: step 0 as a plain launch, then steps 1 .. T-1 in ONE launch (gjx_pfilter.inl) ----
struct PfMove { float* m_a; float* m_b; int n_moves; float move_scale; unsigned long long* acc_total; };
// GJX_EUNSUPPORTED when the shape does not fit the kernel or its grid would not be co-resident (the caller falls back)
static int pf_filter_launch(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K, const float* ys_dev,
                            float* x_a, float* x_b, float* logw, float* lw_alt, int32_t* ancestors, float* lse_steps, char* ws1, char* ws2,
                            size_t need, void* stream, const PfMove* mv) {
  const int64_t nblk = (K + 255) / 256;
  PfPlan pf;
  if (pf_plan(rng_mode, m->dx, m->dy, K, 1, 1, &pf, mv != nullptr) != GJX_OK ||
      256 + (16 * (size_t)kPfGranulePad + 24) * (size_t)pf.nt + 8 * (size_t)pf.grid + 16 * (size_t)T + 64 > need)
    return GJX_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  auto lw_of = [&](int t) { return ((T - 1 - t) & 1) ? lw_alt : logw; };
  std::vector<uint32_t> h_keys;
  std::vector<double> h_us;
  pf_step_keys(key0, key1, T, h_keys, h_us);
  const size_t NT = (size_t)pf.nt;
  // ws2: [256 B control][aggA 64 NT][aggB 64 NT][bsum 12 NT][bmax 12 NT][ready 4 grid, padded to 8][us 8 T][keys 8 T]
  unsigned long long* aggA = (unsigned long long*)(ws2 + kWsHeaderBytes);
  unsigned long long* aggB = aggA + NT * kPfGranulePad;
  float* bsum = (float*)(aggB + NT * kPfGranulePad);
  float* bmax = bsum + 3 * NT;
  unsigned* ready = (unsigned*)(bmax + 3 * NT);
  double* us_dev = (double*)(ready + 2 * (((size_t)pf.grid + 1) / 2));
  uint32_t* keys_dev = (uint32_t*)(us_dev + T);
  hipError_t e = hipMemsetAsync(aggA, 0, (16 * kPfGranulePad + 24) * NT + 8 * (((size_t)pf.grid + 1) / 2), st);   // no stale granule of another kernel may pass
  if (e == hipSuccess && mv && mv->acc_total) e = hipMemsetAsync(mv->acc_total, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter(workspace)");
  // (step keys and comb offsets travel as kernel arguments: no host buffer outlives this call, no per-thread staging state)
  if (int rcu = upload_words(us_dev, h_us.data(), (size_t)T, st)) return rcu;
  if (int rcu = upload_words(keys_dev, h_keys.data(), (size_t)T, st)) return rcu;
  {   // step 0: from the prior (no move: there is nothing to rejuvenate yet)
    SsmArgs a;
    a.A = m->A_dev; a.H = m->H_dev; a.y = ys_dev; a.q = m->q; a.r = m->r; a.q0 = m->q0; a.dy = m->dy; a.t = 0;
    a.key = key2{h_keys[0], h_keys[1]}; a.K = K; a.offset = 0; a.prev_stride = K;
    a.x_prev = nullptr; a.anc = nullptr; a.x_out = x_a; a.logw = lw_of(0);
    a.partials = nullptr; a.ticket = (unsigned*)ws1; a.lse = nullptr; a.log_k_total = (float)log((double)K);
    a.m_prev = nullptr; a.m_out = nullptr; a.y_prev = nullptr; a.n_moves = 0; a.move_scale = 0.0f; a.accepted = nullptr; a.x_moved = nullptr;
    const int rc = rng_mode == GJX_RNG_JAX32 ? launch_ssm<GJX_RNG_JAX32>(a, m->dx, (int)nblk, st) : launch_ssm<GJX_RNG_FLAT>(a, m->dx, (int)nblk, st);
    if (rc) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter: dx must be one of 1,2,4,8,16,32");
    GJX_CHECK_LAUNCH("gjx_ssm_filter(step 0)");
  }
  PfArgs f;
  memset(&f, 0, sizeof(f));
  f.A = m->A_dev; f.H = m->H_dev; f.ys = ys_dev; f.q = m->q; f.r = m->r; f.dy = m->dy; f.T = T;
  f.K = K; f.K_total = K; f.offset = 0; f.G = 1; f.rank = 0; f.nt = pf.nt; f.NT = pf.nt;
  f.x_a = x_a; f.x_b = x_b; f.lw_even = logw; f.lw_odd = lw_alt;
  f.aggA = aggA; f.aggB = aggB; f.bsum = bsum; f.bmax = bmax; f.ready = ready;
  f.peer_data = nullptr; f.peer_flag = nullptr; f.keys = keys_dev; f.us = us_dev;
  f.lse_steps = lse_steps; f.ancestors = ancestors; f.ctrl = (unsigned*)ws2 + 8; f.log_k = (float)log((double)K);
  f.first_budget = 1u << 16; f.zero_ptr = nullptr; f.zero_n = 0;
  f.q0 = m->q0;
  f.timeline = gjx::debug_timeline(128 * (size_t)pf.grid);     // (profiling scripts only: gjx_debug_timeline registers the buffer)
  if (mv) { f.m_a = mv->m_a; f.m_b = mv->m_b; f.n_moves = mv->n_moves; f.move_scale = mv->move_scale; f.acc_total = mv->acc_total; }
  void* args[] = {&f};
  e = hipLaunchKernel(pf.fn, dim3((unsigned)pf.grid), dim3(kPfHostThreads), args, pf.lds, st);
  if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter(k_pf_persistent)");
  return GJX_OK;
}
extern "C" int gjx_ssm_filter_scheme(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                                     const float* ys_dev, float* x_a, float* x_b, float* logw, uint64_t* cum, int32_t* ancestors,
                                     float* lse_steps, int32_t weight_scheme, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int gjx_ssm_filter_move(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                                   const float* ys_dev, float* x_a, float* x_b, float* m_a, float* m_b, float* logw, float* logw_alt,
                                   int32_t* ancestors, float* lse_steps, int32_t n_moves, float move_scale, uint64_t* accepted_total,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !ys_dev || !x_a || !x_b || !m_a || !m_b || !logw || !logw_alt || !lse_steps || T <= 1 || K <= 0 || n_moves < 0)
    return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_move: bad argument");
  const size_t need = gjx_workspace_bytes(GJX_OP_SSM, K);
  if (!workspace || workspace_bytes < 2 * need + 64) return gjx_fail(GJX_EWORKSPACE, "gjx_ssm_filter_move: workspace too small (2x OP_SSM + 64)");
  if (getenv("GJX_SSM_PERSISTENT") && atoi(getenv("GJX_SSM_PERSISTENT")) == 0)
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter_move: one-launch filters are disabled (GJX_SSM_PERSISTENT=0)");
  PfMove mv{m_a, m_b, (int)n_moves, move_scale, (unsigned long long*)accepted_total};
  const int rc = pf_filter_launch(m, key0, key1, rng_mode, T, K, ys_dev, x_a, x_b, logw, logw_alt, ancestors, lse_steps, (char*)workspace,
                                  (char*)workspace + need, need, stream, &mv);
  if (rc == GJX_EUNSUPPORTED) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter_move: shape or size outside the one-launch filter (use gjx_ssm_step_move per step)");
  return rc;
}

extern "C" int gjx_ssm_filter(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                              const float* ys_dev /*[T][dy]*/, float* x_a, float* x_b /*[dx][K] ping-pong*/, float* logw,
                              uint64_t* cum, int32_t* ancestors, float* lse_steps /*[T][4]*/, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return gjx_ssm_filter_scheme(m, key0, key1, rng_mode, T, K, ys_dev, x_a, x_b, logw, cum, ancestors, lse_steps,
                               GJX_WEIGHTS_GLOBAL_MAX, workspace, workspace_bytes, stream);
}

extern "C" int gjx_ssm_filter_scheme(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                                     const float* ys_dev /*[T][dy]*/, float* x_a, float* x_b /*[dx][K] ping-pong*/, float* logw,
                                     uint64_t* cum, int32_t* ancestors, float* lse_steps /*[T][4]*/, int32_t weight_scheme,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !ys_dev || !x_a || !x_b || !logw || !cum || !ancestors || !lse_steps || T <= 0 || K <= 0)
    return gjx_fail(GJX_EINVAL, "gjx_ssm_filter: bad argument");
  // GJX_WEIGHTS_PLAIN_LAUNCHES: no kernel of this call may wait for its own blocks (the caller repeats a timed-out run)
  const gjx_plain_launch_scope plain_scope((weight_scheme & GJX_WEIGHTS_PLAIN_LAUNCHES) != 0);
  weight_scheme &= ~GJX_WEIGHTS_PLAIN_LAUNCHES;
  if (weight_scheme != GJX_WEIGHTS_GLOBAL_MAX && weight_scheme != GJX_WEIGHTS_TILE_SCALED)
    return gjx_fail(GJX_EINVAL, "gjx_ssm_filter: weight_scheme must be GJX_WEIGHTS_GLOBAL_MAX or GJX_WEIGHTS_TILE_SCALED");
  const bool tiled = weight_scheme == GJX_WEIGHTS_TILE_SCALED;
  const size_t need = gjx_workspace_bytes(GJX_OP_SSM, K);
  if (!workspace || workspace_bytes < 2 * need + 64) return gjx_fail(GJX_EWORKSPACE, "gjx_ssm_filter: workspace too small (2x OP_SSM + 64)");
  char* ws1 = (char*)workspace;
  char* ws2 = ws1 + need;
  uint64_t* bt = (uint64_t*)(ws2 + need);
  uint32_t k[2] = {key0, key1};
  // one launch per step when the grid of K / 256 blocks is co-resident (k_ssm_fused_step); GJX_SSM_TWO_LAUNCH=1 keeps
  // the resample + step pair
  const int64_t nblk = (K + 255) / 256;
  const void* fused_fn = nullptr;
  if (!tiled && (!getenv("GJX_SSM_TWO_LAUNCH") || atoi(getenv("GJX_SSM_TWO_LAUNCH")) == 0)) {
    const bool jax = rng_mode == GJX_RNG_JAX32;
    switch (m->dx) {
      case 2: fused_fn = jax ? (const void*)k_ssm_fused_step<GJX_RNG_JAX32, 2> : (const void*)k_ssm_fused_step<GJX_RNG_FLAT, 2>; break;
      case 4: fused_fn = jax ? (const void*)k_ssm_fused_step<GJX_RNG_JAX32, 4> : (const void*)k_ssm_fused_step<GJX_RNG_FLAT, 4>; break;
      case 8: fused_fn = jax ? (const void*)k_ssm_fused_step<GJX_RNG_JAX32, 8> : (const void*)k_ssm_fused_step<GJX_RNG_FLAT, 8>; break;
      case 16: fused_fn = jax ? (const void*)k_ssm_fused_step<GJX_RNG_JAX32, 16> : (const void*)k_ssm_fused_step<GJX_RNG_FLAT, 16>; break;
      default: break;
    }
    if (fused_fn && (nblk > kSsmFusedMaxTiles || nblk > gjx_coresident_blocks(fused_fn, 256, 0) || 256 + 16 * (size_t)nblk > need)) fused_fn = nullptr;
  }
  // the whole filter in one launch (k_ssm_persistent) when everything fits: GJX_SSM_PERSISTENT=0 keeps one launch per step
  const void* pers_fn = nullptr;
  int pthreads = 256;
  int64_t pblk = nblk;
  if ((fused_fn || tiled) && T > 1 && (!gjx_plain_launches_forced() && (!getenv("GJX_SSM_PERSISTENT") || atoi(getenv("GJX_SSM_PERSISTENT")) != 0))) {
    const bool jax = rng_mode == GJX_RNG_JAX32;
    // 1024-thread blocks (one per CU) once the grid would have more than 256 blocks of 256: fewer, cheaper rendezvous;
    // always for the tile-scaled scheme (its quantisation tile is 1024 particles)
    pthreads = (tiled || (nblk > 256 && (!getenv("GJX_SSM_THREADS") || atoi(getenv("GJX_SSM_THREADS")) == 1024))) ? 1024 : 256;
#define GJX_PERS(DXV) (tiled ? (jax ? (const void*)k_ssm_persistent<GJX_RNG_JAX32, DXV, 1024, true> : (const void*)k_ssm_persistent<GJX_RNG_FLAT, DXV, 1024, true>) \
                     : pthreads == 1024 ? (jax ? (const void*)k_ssm_persistent<GJX_RNG_JAX32, DXV, 1024, false> : (const void*)k_ssm_persistent<GJX_RNG_FLAT, DXV, 1024, false>) \
                                        : (jax ? (const void*)k_ssm_persistent<GJX_RNG_JAX32, DXV, 256, false> : (const void*)k_ssm_persistent<GJX_RNG_FLAT, DXV, 256, false>))
    switch (m->dx) {
      case 2: pers_fn = GJX_PERS(2); break;
      case 4: pers_fn = GJX_PERS(4); break;
      case 8: pers_fn = GJX_PERS(8); break;
      case 16: pers_fn = GJX_PERS(16); break;
      default: break;
    }
#undef GJX_PERS
    pblk = (K + pthreads - 1) / pthreads;
    if (pers_fn && (m->dy > kSsmPersistMaxDy || pblk > kSsmFusedMaxTiles || pblk > gjx_coresident_blocks(pers_fn, pthreads, 0) || 256 + (16 * (size_t)kGranulePad + 32) * (size_t)pblk + 16 * (size_t)T + 64 > need)) pers_fn = nullptr;
  }
  // tile-scaled scheme beyond one slot per lane (or GJX_PF=1): k_pf_persistent, several quantisation tiles per block
  // (the kernel on the shared skeleton, k_pf_persistent, first — since its prefix runs in one wave for few tiles it is the faster of
  // the two at config 3's size as well: 11.2 against 11.5 us per step; GJX_PF=0 keeps k_ssm_persistent<TILED>)
  if (tiled && T > 1 && (!pers_fn || !(getenv("GJX_PF") && atoi(getenv("GJX_PF")) == 0)) &&
      (!gjx_plain_launches_forced() && (!getenv("GJX_SSM_PERSISTENT") || atoi(getenv("GJX_SSM_PERSISTENT")) != 0))) {
    const int rc_pf = pf_filter_launch(m, key0, key1, rng_mode, T, K, ys_dev, x_a, x_b, logw, (float*)cum, ancestors, lse_steps, ws1, ws2, need,
                                       stream, nullptr);
    if (rc_pf != GJX_EUNSUPPORTED) return rc_pf;
  }
  if (pers_fn) {
    hipStream_t st = (hipStream_t)stream;
    float* lw_alt = (float*)cum;
    auto lw_of = [&](int t) { return ((T - 1 - t) & 1) ? lw_alt : logw; };
    std::vector<uint32_t> h_keys(2 * (size_t)T, 0u);
    std::vector<double> h_us((size_t)T, 0.0);
    uint32_t kp0[2] = {0u, 0u};
    for (int t = 0; t < T; ++t) {
      uint32_t kt[2], kp[2], kr[2], b[2];
      host_threefry(k[0], k[1], 0u, (uint32_t)t, kt);
      k[0] = kt[0]; k[1] = kt[1];
      host_threefry(k[0], k[1], 0u, 0u, kp);
      host_threefry(k[0], k[1], 0u, 1u, kr);
      host_threefry(kr[0], kr[1], 0u, 0u, b);
      h_keys[2 * t] = kp[0]; h_keys[2 * t + 1] = kp[1];
      h_us[t] = (double)((b[0] ^ b[1]) >> 9) / 8388608.0;
      if (t == 0) { kp0[0] = kp[0]; kp0[1] = kp[1]; }
    }
    // ws2: [256 B control][aggA 8 gp nb][aggB 8 gp nb][bsum ring 12 nb][bmax ring 12 nb][ready 4 nb + 4 nb pad][us 8 T][keys 8 T]
    const size_t gp = (size_t)kGranulePad;                    // one 64-byte line per granule, both schemes
    unsigned long long* aggA = (unsigned long long*)(ws2 + kWsHeaderBytes);
    unsigned long long* aggB = aggA + gp * pblk;
    float* bsum = (float*)(aggB + gp * pblk);
    float* bmax = bsum + 3 * pblk;
    unsigned* ready = (unsigned*)(bmax + 3 * pblk);
    double* us_dev = (double*)(ready + 2 * pblk);
    uint32_t* keys_dev = (uint32_t*)(us_dev + T);
    // the two schemes (and k_ssm_fused_step) tag their granules differently: no stale granule of another kernel may pass for
    // one of this launch
    hipError_t e = hipMemsetAsync(aggA, 0, (16 * gp + 32) * (size_t)pblk, st);
    if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter(workspace)");
    if (int rcu = upload_words(us_dev, h_us.data(), (size_t)T, st)) return rcu;        // (kernel arguments: nothing outlives this call)
    if (int rcu = upload_words(keys_dev, h_keys.data(), (size_t)T, st)) return rcu;
    {   // step 0: from the prior
      SsmArgs a;
      a.A = m->A_dev; a.H = m->H_dev; a.y = ys_dev; a.q = m->q; a.r = m->r; a.q0 = m->q0; a.dy = m->dy; a.t = 0;
      a.key = key2{kp0[0], kp0[1]}; a.K = K; a.offset = 0; a.prev_stride = K;
      a.x_prev = nullptr; a.anc = nullptr; a.x_out = x_a; a.logw = lw_of(0);
      a.partials = nullptr; a.ticket = (unsigned*)ws1; a.lse = nullptr; a.log_k_total = (float)log((double)K);
      a.m_prev = nullptr; a.m_out = nullptr; a.y_prev = nullptr; a.n_moves = 0; a.move_scale = 0.0f; a.accepted = nullptr; a.x_moved = nullptr;
      const int rc = rng_mode == GJX_RNG_JAX32 ? launch_ssm<GJX_RNG_JAX32>(a, m->dx, (int)nblk, st) : launch_ssm<GJX_RNG_FLAT>(a, m->dx, (int)nblk, st);
      if (rc) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter: dx must be one of 1,2,4,8,16,32");
      GJX_CHECK_LAUNCH("gjx_ssm_filter(step 0)");
    }
    SsmPersistArgs f;
    f.A = m->A_dev; f.H = m->H_dev; f.ys = ys_dev; f.q = m->q; f.r = m->r; f.dy = m->dy; f.T = T; f.K = K;
    f.x_a = x_a; f.x_b = x_b; f.lw_even = logw; f.lw_odd = lw_alt;
    f.keys = keys_dev; f.us = us_dev; f.lse_steps = lse_steps; f.ancestors = ancestors;
    f.aggA = aggA; f.aggB = aggB; f.bsum = bsum; f.bmax = bmax; f.ready = ready; f.ctrl = (unsigned*)ws2 + 8; f.log_k = (float)log((double)K);
    f.timeline = nullptr;
    f.timeline = gjx::debug_timeline(128 * (size_t)pblk);
    void* args[] = {&f};
    e = hipLaunchKernel(pers_fn, dim3((unsigned)pblk), dim3((unsigned)pthreads), args, 0, st);
    if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter(persistent)");
    return GJX_OK;
  }
  if (fused_fn) {
    hipStream_t st = (hipStream_t)stream;
    {
      const hipError_t e0 = hipMemsetAsync(ws2 + kWsHeaderBytes, 0, 8 * (size_t)nblk, st);   // see the persistent path
      if (e0 != hipSuccess) return gjx_fail_hip(e0, "gjx_ssm_filter(granules)");
    }
    float* lw_alt = (float*)cum;                     // the prefix-sum buffer is free on this path: second log-weight buffer
    auto lw_of = [&](int t) { return ((T - 1 - t) & 1) ? lw_alt : logw; };   // the last step writes the caller's logw
    auto part_of = [&](int t) { return (unsigned long long*)(ws1 + kWsHeaderBytes) + (size_t)(t & 1) * nblk; };
    for (int t = 0; t < T; ++t) {
      uint32_t kt[2], kp[2], kr[2], b[2];
      host_threefry(k[0], k[1], 0u, (uint32_t)t, kt);
      k[0] = kt[0]; k[1] = kt[1];
      host_threefry(k[0], k[1], 0u, 0u, kp);
      host_threefry(k[0], k[1], 0u, 1u, kr);
      float* x_out = (t & 1) ? x_b : x_a;
      const float* x_prev = (t & 1) ? x_a : x_b;
      SsmArgs a;
      a.A = m->A_dev; a.H = m->H_dev; a.y = ys_dev + (size_t)t * m->dy; a.q = m->q; a.r = m->r; a.q0 = m->q0; a.dy = m->dy; a.t = t;
      a.key = key2{kp[0], kp[1]}; a.K = K; a.offset = 0; a.prev_stride = K;
      a.x_prev = t > 0 ? x_prev : nullptr; a.anc = nullptr; a.x_out = x_out; a.logw = lw_of(t);
      a.partials = part_of(t); a.ticket = (unsigned*)ws1; a.lse = t == T - 1 ? lse_steps + 4 * (size_t)t : nullptr;
      a.log_k_total = (float)log((double)K);
      a.m_prev = nullptr; a.m_out = nullptr; a.y_prev = nullptr; a.n_moves = 0; a.move_scale = 0.0f; a.accepted = nullptr; a.x_moved = nullptr;
      if (t == 0) {
        const int rc = rng_mode == GJX_RNG_JAX32 ? launch_ssm<GJX_RNG_JAX32>(a, m->dx, (int)nblk, st) : launch_ssm<GJX_RNG_FLAT>(a, m->dx, (int)nblk, st);
        if (rc) return gjx_fail(GJX_EUNSUPPORTED, "gjx_ssm_filter: dx must be one of 1,2,4,8,16,32");
        GJX_CHECK_LAUNCH("gjx_ssm_filter(step 0)");
        continue;
      }
      host_threefry(kr[0], kr[1], 0u, 0u, b);
      SsmFusedArgs f;
      f.s = a;
      f.logw_prev = lw_of(t - 1);
      f.partials_prev = (const float*)part_of(t - 1);
      f.n_partials_prev = (int)nblk;
      f.lse_prev_out = lse_steps + 4 * (size_t)(t - 1);
      f.log_k_total_prev = (float)log((double)K);
      f.u = (double)((b[0] ^ b[1]) >> 9) / 8388608.0;
      f.ancestors = t == T - 1 ? ancestors : nullptr;
      f.agg = (unsigned long long*)(ws2 + kWsHeaderBytes);
      f.ctrl = (unsigned*)ws2 + 8;
      f.timeline = nullptr;   // debug: phase stamps of the middle step (profiles/microbench/ssm_timeline.py)
      if (t == T / 2) f.timeline = gjx::debug_timeline(64 * (size_t)nblk);
      void* args[] = {&f};
      const hipError_t e = hipLaunchKernel(fused_fn, dim3((unsigned)nblk), dim3(256), args, 0, st);
      if (e != hipSuccess) return gjx_fail_hip(e, "gjx_ssm_filter(fused step)");
    }
    return GJX_OK;
  }
  for (int t = 0; t < T; ++t) {
    uint32_t kt[2], kp[2], kr[2], b[2];
    host_threefry(k[0], k[1], 0u, (uint32_t)t, kt);
    k[0] = kt[0]; k[1] = kt[1];
    host_threefry(k[0], k[1], 0u, 0u, kp);
    host_threefry(k[0], k[1], 0u, 1u, kr);
    float* x_out = (t & 1) ? x_b : x_a;
    const float* x_prev = (t & 1) ? x_a : x_b;
    float* lse = lse_steps + 4 * (size_t)t;
    if (t > 0) {
      host_threefry(kr[0], kr[1], 0u, 0u, b);
      const double u = (double)((b[0] ^ b[1]) >> 9) / 8388608.0;
      // the previous step left its per-block LSE partials in ws1; the prefix-sum prologue reduces them and
      // block 0 writes the finished record of step t-1
      const int rc = tiled ? gjx_resample_indices_tiled(logw, K, u, K, ancestors, cum, nullptr, nullptr, ws2, need, stream)
                           : gjx_resample_indices(logw, K, 2, (const float*)(ws1 + 256), (int32_t)((K + 255) / 256), u, K, ancestors, cum, bt,
                                                  lse - 4, K, ws2, need, stream);
      if (rc) return rc;
    }
    // tile-scaled scheme: the resampler does not touch the LSE partials, every step finishes its own record
    const int rc = gjx_ssm_step(m, kp[0], kp[1], rng_mode, t, K, 0, t > 0 ? x_prev : nullptr, K, t > 0 ? ancestors : nullptr,
                                ys_dev + (size_t)t * m->dy, x_out, logw, (tiled || t == T - 1) ? lse : nullptr, K, ws1, need, stream);
    if (rc) return rc;
  }
  return GJX_OK;
}

extern "C" int gjx_ssm_filter_sharded(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T,
                                      gjx_shard_ctx* ctx, int64_t particle_offset, const float* ys_dev, float* x_a, float* x_b,
                                      float* logw, float* lse_steps, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !ctx || !ys_dev || !x_a || !x_b || !logw || !lse_steps || T <= 0)
    return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_sharded: bad argument");
  int64_t shape[5];
  int rc = gjx_shard_ctx_shape(ctx, shape);
  if (rc) return rc;
  const int64_t K = shape[0], N_total = shape[2];
  if (shape[1] != m->dx || N_total != K * shape[3])
    return gjx_fail(GJX_EINVAL, "gjx_ssm_filter_sharded: context shape must be (K_local, dx, n_ranks * K_local)");
  const size_t need = gjx_workspace_bytes(GJX_OP_SSM, K);
  if (!workspace || workspace_bytes < need + 64) return gjx_fail(GJX_EWORKSPACE, "gjx_ssm_filter_sharded: workspace too small (OP_SSM + 64)");
  float* lse_local = (float*)((char*)workspace + need);
  uint32_t k[2] = {key0, key1};
  uint32_t kt[2], kp[2], kr[2], b[2];
  host_threefry(k[0], k[1], 0u, 0u, kt);       // k_0 = fold_in(key, 0)
  k[0] = kt[0]; k[1] = kt[1];
  for (int t = 0; t < T; ++t) {
    host_threefry(k[0], k[1], 0u, 0u, kp);
    rc = gjx_ssm_step(m, kp[0], kp[1], rng_mode, t, K, particle_offset, t > 0 ? x_b : nullptr, K, nullptr,
                      ys_dev + (size_t)t * m->dy, x_a, logw, lse_local, N_total, workspace, need, stream);
    if (rc) return rc;
    float* rec = lse_steps + 4 * (size_t)t;
    if (t + 1 < T) {
      // step t+1's key gives the comb offset of the resampling that precedes it; the exchange also yields the
      // global LSE record of step t
      host_threefry(k[0], k[1], 0u, (uint32_t)(t + 1), kt);
      k[0] = kt[0]; k[1] = kt[1];
      host_threefry(k[0], k[1], 0u, 1u, kr);
      host_threefry(kr[0], kr[1], 0u, 0u, b);
      const double u = (double)((b[0] ^ b[1]) >> 9) / 8388608.0;
      rc = gjx_shard_resample_step(ctx, logw, lse_local, x_a, K, x_b, K, u, rec, nullptr, stream);
    } else {
      rc = gjx_shard_global_lse(ctx, lse_local, rec, stream);
    }
    if (rc) return rc;
  }
  return GJX_OK;
}
