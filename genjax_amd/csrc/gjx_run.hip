// gjx_run.hip — particle propagate + reweight kernels (gjx_run_program) for gfx950.
//
//   k_run_generic : site-list interpreter, one particle per lane.  Site descriptors are read
//                   through wave-uniform addresses (scalar loads, SGPRs); particle values live in
//                   the SoA rows of choices[][] (coalesced 256 B per wave per row).
//   k_run_gmm     : hand-fused kernel for the mixture shape of BASELINE config 2
//                   (categorical -> mv_normal_diag(gather) -> observed mv_normal_diag), PPT
//                   particles per lane so that every SoA row is written with 16-byte stores; tables
//                   staged in LDS with a padded row stride (conflict-free ds_read_b128); the
//                   block's {max, sum-exp} of the log-weights reduced with wave shuffles + LDS.
// Both produce identical random streams (same Threefry counters) — tests compare them bitwise on
// the integer side and to float tolerance on the float side.
#include <hip/hip_ext.h>

#include "gjx_device.h"
#include "gjx_host.h"
#include "gjx_scan.h"
#include "gjx_tile.h"

#include <string.h>
#include <type_traits>

namespace gjx {

// ------------------------------------------------------------------------------------------
// generic interpreter
// ------------------------------------------------------------------------------------------
struct RunArgs {
  const gjx_site* sites;
  const float* tab;
  int n_sites, n_slots;
  key2 key;
  int64_t K, offset;
  float* choices;
  float* score;
  float* weight;
  float* logw;
  const float* logw_in;
  const float* sub;
  float* site_scores;
  unsigned long long* partials;  // per block {max, sumexp} of logw (or NULL)
  unsigned* ticket;
  float* lse;
  float log_k_total;
  int preload, n_tab;
  const float* in_rows;          // GJX_MODE_INPUT sites: in_rows[(obs_off + d) * in_stride + ancestor(i)] (NULL: their rows of choices)
  int64_t in_stride;
  const int32_t* anc;
  int store_inputs;
};

// LDSV: the particle's values are mirrored in LDS (vals[slot][lane], conflict-free) so that parameter
// expressions read earlier choices at LDS latency instead of re-reading the SoA rows through L1/L2; used
// whenever n_slots * 1 KB fits (the launcher decides).  The SoA rows in HBM are still written once.
// TABL: the float table (constants, observations) is small enough to be copied into LDS as well, so GATHER /
// AFFINE parameters (per-lane table reads) stop being dependent global loads.
template <int RNG, bool LDSV, bool TABL>
__global__ __launch_bounds__(256) void k_run_generic(RunArgs a) {
  __shared__ float red[16];
  extern __shared__ __attribute__((aligned(16))) float vals_s[];
  float* tab_s = vals_s + (LDSV ? a.n_slots * 256 : 0);
  if (TABL) {
    for (int t = threadIdx.x; t < a.n_tab; t += 256) tab_s[t] = a.tab[t];
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = i < a.K;
  const int64_t ii = active ? i : a.K - 1;  // inactive lanes shadow the last particle (no stores)
  const uint64_t gidx = (uint64_t)(a.offset + ii);
  const float* __restrict__ tab = TABL ? tab_s : a.tab;
  float* ch = a.choices;
  const int64_t K = a.K;
  if (LDSV && a.preload) {  // some site reads its value from choices[][] (OBS_SLOT): stage every row once, coalesced
    for (int s = 0; s < a.n_slots; ++s) vals_s[s * 256 + threadIdx.x] = ch[(int64_t)s * K + ii];
  }
  auto val = [&](int slot) -> float { return LDSV ? vals_s[slot * 256 + threadIdx.x] : ch[(int64_t)slot * K + ii]; };

  float score = 0.0f, weight = 0.0f;
  SiteStreamWalk walk(a.key);     // wave-uniform: chained step keys of Scan sites (gjx.h "Scan steps")
  BitStreamRT<RNG> rs;            // stream of the open scalar-normal run (gjx.h "Scalar-normal runs"): lives across sites
  uint32_t jn = 0u;               // sites seen that are not GJX_MODE_INPUT (those take no site number)
  for (int j0 = 0; j0 < a.n_sites;) {
    if (a.sites[j0].mode == GJX_MODE_INPUT) {
      // the carry of a Scan step / an argument of the kernel: rows that are already there — or the rows of the ancestor the
      // resampling step picked for this slot (the particle gather fused into the read side); no draw, no score
      const gjx_site& s = a.sites[j0];
      const int64_t src_i = a.anc ? (int64_t)a.anc[ii] : ii;
      for (int d = 0; d < s.dim; ++d) {
        const float v = a.in_rows ? a.in_rows[(int64_t)(s.obs_off + d) * a.in_stride + src_i] : ch[(int64_t)(s.slot + d) * K + ii];
        if (LDSV) vals_s[(s.slot + d) * 256 + threadIdx.x] = v;
        if (a.in_rows && active && (!LDSV || a.store_inputs || (s.flags & GJX_SITE_CARRIED))) ch[(int64_t)(s.slot + d) * K + i] = v;
      }
      if (a.site_scores && active) a.site_scores[(int64_t)j0 * K + i] = 0.0f;
      ++j0;
      continue;
    }
    // a plate (gjx.h "Plates"; vmap.py:180-218): m consecutive body sites, n instances, ONE instance loop over the body;
    // any other site: m = n = 1
    int m = 1, ninst = 1;
    const int plate = a.sites[j0].plate;
    if (plate != 0) {
      while (j0 + m < a.n_sites && a.sites[j0 + m].plate == plate) ++m;
      ninst = a.sites[j0].plate_n;
    }
    // site numbers: the body's are consecutive (one scan tag, no run members); a drawing body site closes an open run
    uint32_t no0 = 0u, e0_single = 0u;
    bool opens_single = false, joins_single = false;
    for (int l = 0; l < m; ++l) {
      const gjx_site& sl = a.sites[j0 + l];
      uint32_t no = RNG == GJX_RNG_FLAT ? walk.next(sl.scan) : jn + (uint32_t)(l + 1);
      const bool draws_l = sl.mode == GJX_MODE_SAMPLE || sl.mode == GJX_MODE_OBS_MASK;
      const bool joins_l = plate == 0 && GJX_FLAT_JOINS(RNG, sl.kind, sl.dim, sl.mode);
      bool op = false;
      const uint32_t e = RNG == GJX_RNG_FLAT ? walk.run_elem(no, joins_l, draws_l, op) : 0u;
      if (l == 0) { no0 = no; e0_single = e; opens_single = op; joins_single = joins_l; }
    }
    key2 plate_key{0u, 0u};       // JAX32: fold_in(particle key, J), J = 1-based index of the plate's first site
    if (RNG == GJX_RNG_JAX32 && plate != 0) plate_key = fold_in(fold_in64(a.key, gidx), jn + 1u);
    jn += (RNG == GJX_RNG_JAX32 && plate != 0) ? 1u : (uint32_t)m;   // (the Vmap call is ONE traced site of its caller: static.py:349-352)
    for (int inst = 0; inst < ninst; ++inst) {
    key2 inst_key{0u, 0u};
    if (RNG == GJX_RNG_JAX32 && plate != 0) inst_key = fold_in(plate_key, (uint32_t)inst);     // split(plate key, n)[inst] (vmap.py:186)
    for (int l = 0; l < m; ++l) {
    const int j = j0 + l;
    const gjx_site& s = a.sites[j];
    // (GJX_MODE_OBS_PROPOSED: the value a proposal site of this run left in the slot — read like a per-particle constraint)
    const int kind = s.kind, mode = s.mode == GJX_MODE_OBS_PROPOSED ? GJX_MODE_OBS_SLOT : s.mode;
    const int width = (kind == GJX_CATEGORICAL_LOGITS || kind == GJX_CATEGORICAL_PROBS) ? 1 : s.dim;
    const int slot = s.slot >= 0 ? s.slot + inst * width : s.slot;
    const int obs_off = s.obs_off + inst * s.d_obs;
    const uint32_t site_no = no0 + (uint32_t)l;
    BitStreamRT<RNG> bs;
    const bool masked = mode == GJX_MODE_OBS_MASK;
    const bool draws = mode == GJX_MODE_SAMPLE || masked;           // wave-uniform
    const bool given = masked ? (val(obs_off) != 0.0f) : (mode != GJX_MODE_SAMPLE);   // per lane under a mask
    const bool joins = joins_single && plate == 0;                  // wave-uniform
    const uint32_t e0 = e0_single;
    // element at which this (site, instance) starts drawing: FLAT = the elements a vector site of n * dim elements would use
    const bool is_cat = kind == GJX_CATEGORICAL_LOGITS || kind == GJX_CATEGORICAL_PROBS;
    const uint32_t eb = (RNG == GJX_RNG_FLAT && plate != 0) ? (uint32_t)inst * (uint32_t)(is_cat ? 1 : s.dim * draws_per_elem(kind)) : 0u;
    if (joins) { if (opens_single) rs.open(walk.key, gidx, site_no); }
    else if (draws) {
      if (RNG == GJX_RNG_JAX32 && plate != 0) bs.open_site_key(fold_in(inst_key, (uint32_t)(l + 1)));   // static.py:349-352 inside the kernel
      else bs.open(RNG == GJX_RNG_FLAT ? walk.key : a.key, gidx, site_no);
    }
    float lp = 0.0f;
    if (kind == GJX_CATEGORICAL_LOGITS || kind == GJX_CATEGORICAL_PROBS) {
      const int n = s.ncat;
      const bool probs = kind == GJX_CATEGORICAL_PROBS;
      float mx = -INFINITY;
      for (int c = 0; c < n; ++c) {
        float l2 = eval_param(s.p[0], c, tab, val, inst);
        if (probs) l2 = safe_log(l2);
        mx = fmaxf(mx, l2);
      }
      float se = 0.0f;
      for (int c = 0; c < n; ++c) {
        float l2 = eval_param(s.p[0], c, tab, val, inst);
        if (probs) l2 = safe_log(l2);
        se += fast_exp(l2 - mx);
      }
      const float lse = mx + fast_log(se);
      float v;
      if (draws && RNG == GJX_RNG_FLAT) {
        // inverse CDF on one uniform (se is the float32 running total in category order)
        const float target = bits_to_unit(bs.get(eb)) * se;
        float run = 0.0f;
        int zc = n - 1;
        bool found = false;
        for (int c = 0; c < n; ++c) {
          float l2 = eval_param(s.p[0], c, tab, val, inst);
          if (probs) l2 = safe_log(l2);
          run += fast_exp(l2 - mx);
          if (!found && run > target) { zc = c; found = true; }
        }
        v = (float)zc;
      } else if (draws) {
        int best = 0;
        float bestv = -INFINITY;
        for (int c = 0; c < n; ++c) {
          float l2 = eval_param(s.p[0], c, tab, val, inst);
          if (probs) l2 = safe_log(l2);
          const float g = l2 + gumbel_from_bits(bs.get((uint32_t)c));
          if (g > bestv) { bestv = g; best = c; }
        }
        v = (float)best;
      } else if (mode == GJX_MODE_OBS_TAB) {
        v = tab[obs_off];
      } else {
        v = val(slot);
      }
      if (masked && given) v = val(slot);
      const int k = (int)v;
      if (k < 0 || k >= n) {
        lp = -INFINITY;
      } else {
        float l2 = eval_param(s.p[0], k, tab, val, inst);
        if (probs) l2 = safe_log(l2);
        lp = l2 - lse;
      }
      if (slot >= 0) {
        if (LDSV) vals_s[slot * 256 + threadIdx.x] = v;
        if (active) ch[(int64_t)slot * K + i] = v;
      }
    } else if (kind == GJX_DIRICHLET) {
      // one site, dim values on the simplex: x = softmax(log-gamma variates); density couples all elements
      const int n = s.dim;
      auto put = [&](int d, float v) {
        if (slot >= 0) {
          if (LDSV) vals_s[(slot + d) * 256 + threadIdx.x] = v;
          else if (active) ch[(int64_t)(slot + d) * K + i] = v;
        }
      };
      auto get = [&](int d) -> float {
        if (mode == GJX_MODE_OBS_TAB) return tab[obs_off + d];
        return LDSV ? vals_s[(slot + d) * 256 + threadIdx.x] : ch[(int64_t)(slot + d) * K + ii];
      };
      if (mode == GJX_MODE_SAMPLE) {
        float mx = -INFINITY;
        for (int d = 0; d < n; ++d) {
          const float lg = log_gamma_variate<RNG>(bs, (uint32_t)(d * kGammaNDraw), eval_param(s.p[0], d, tab, val));
          put(d, lg);
          mx = fmaxf(mx, lg);
        }
        float se = 0.0f;
        for (int d = 0; d < n; ++d) se += fast_exp(get(d) - mx);
        const float lse = mx + fast_log(se);
        for (int d = 0; d < n; ++d) put(d, fast_exp(get(d) - lse));
      }
      float sa = 0.0f;
      for (int d = 0; d < n; ++d) {
        const float al = eval_param(s.p[0], d, tab, val);
        const float v = get(d);
        sa += al;
        lp += ((al - 1.0f) == 0.0f ? 0.0f : (al - 1.0f) * fast_log(v)) - lgammaf(al);
        if (LDSV && slot >= 0 && mode != GJX_MODE_OBS_SLOT && active) ch[(int64_t)(slot + d) * K + i] = v;
      }
      lp += lgammaf(sa);
    } else {
      // the element loop is instantiated per distribution kind so that the sampler / density switches fold away
      auto elems = [&](auto kind_c) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int NP = kind_params(KIND);
        const int nd = draws_per_elem(KIND);
        const int dim = s.dim;
        const bool b_inv = s.p[1].op == GJX_P_CONST && s.p[1].len == 1 && s.p[1].xf == GJX_XF_NONE && s.p[1].d_off == 0;  // wave-uniform
        const float pb0 = b_inv ? tab[s.p[1].off] : 0.0f;
        for (int d = 0; d < dim; ++d) {
          const float pa = eval_param(s.p[0], d, tab, val, inst);
          const float pb = b_inv ? pb0 : eval_param(s.p[1], d, tab, val, inst);
          const float pc = NP > 2 ? eval_param(s.p[2], d, tab, val, inst) : 0.0f;
          const float pd = NP > 3 ? eval_param(s.p[3], d, tab, val, inst) : 0.0f;
          float v;
          if (KIND == GJX_NORMAL && joins) v = fmaf(pb, stream_normal<RNG>(rs, e0), pa);     // member e0 of its run
          else if (draws) v = elem_sample<RNG>(KIND, bs, eb + (uint32_t)(d * nd), pa, pb, pc, pd);
          else if (mode == GJX_MODE_OBS_TAB) v = tab[obs_off + d];
          else v = val(slot + d);
          if (masked && given) v = val(slot + d);
          lp += elem_logpdf(KIND, v, pa, pb, pc, pd);
          if (slot >= 0 && mode != GJX_MODE_OBS_SLOT) {
            if (LDSV) vals_s[(slot + d) * 256 + threadIdx.x] = v;
            if (active) ch[(int64_t)(slot + d) * K + i] = v;
          }
        }
      };
#define GJX_KIND(K_) case K_: elems(std::integral_constant<int, K_>{}); break;
      switch (kind) {
        GJX_KIND(GJX_NORMAL) GJX_KIND(GJX_MVNORMAL_DIAG) GJX_KIND(GJX_FLIP) GJX_KIND(GJX_BERNOULLI_LOGITS) GJX_KIND(GJX_BETA)
        GJX_KIND(GJX_UNIFORM) GJX_KIND(GJX_EXPONENTIAL) GJX_KIND(GJX_HALF_NORMAL) GJX_KIND(GJX_LAPLACE) GJX_KIND(GJX_LOG_NORMAL)
        GJX_KIND(GJX_CAUCHY) GJX_KIND(GJX_GAMMA) GJX_KIND(GJX_STUDENT_T) GJX_KIND(GJX_TRUNCATED_NORMAL) GJX_KIND(GJX_POISSON)
        GJX_KIND(GJX_GEOMETRIC) GJX_KIND(GJX_GUMBEL) GJX_KIND(GJX_HALF_CAUCHY) GJX_KIND(GJX_INVERSE_GAMMA) GJX_KIND(GJX_WEIBULL)
        GJX_KIND(GJX_LOGIT_NORMAL) GJX_KIND(GJX_CHI2) GJX_KIND(GJX_CHI) GJX_KIND(GJX_EXP_GAMMA) GJX_KIND(GJX_EXP_INVERSE_GAMMA)
        GJX_KIND(GJX_HALF_STUDENT_T) GJX_KIND(GJX_KUMARASWAMY) GJX_KIND(GJX_MOYAL) GJX_KIND(GJX_TRUNCATED_CAUCHY)
        GJX_KIND(GJX_DOUBLESIDED_MAXWELL) GJX_KIND(GJX_INVERSE_GAUSSIAN) GJX_KIND(GJX_NEGATIVE_BINOMIAL) GJX_KIND(GJX_VON_MISES)
        default: lp = __builtin_nanf(""); break;
      }
#undef GJX_KIND
    }
    if (s.flags & GJX_SITE_PROPOSAL) weight -= lp;      // a proposal's site: log w = log p - log q (smc.py:313), no part of the score
    else {
      score += lp;
      if (given) weight += lp;
    }
    if (a.site_scores && active) {      // a plate's body site: the sum over its instances
      float* ss = a.site_scores + (int64_t)j * K + i;
      *ss = inst == 0 ? lp : *ss + lp;
    }
    }   // body sites
    }   // instances
    j0 += m;
  }
  float lw = weight;
  if (a.logw_in) lw += a.logw_in[ii];
  if (a.sub) lw -= a.sub[ii];
  if (active) {
    if (a.score) a.score[i] = score;
    if (a.weight) a.weight[i] = weight;
    if (a.logw) a.logw[i] = lw;
  }
  if (a.partials) {
    float bm, bsum;
    block_lse_partial<256>(lw, active, red, bm, bsum);
    if (a.lse) lse_publish_and_finish<256>(bm, bsum, a.partials, a.ticket, (int)gridDim.x, a.log_k_total, a.lse, red);
    else if (threadIdx.x == 0) a.partials[blockIdx.x] = pack_f2(bm, bsum);  // consumer finishes (gjx_weight_cumsum mode 2)
  }
}

// ------------------------------------------------------------------------------------------
// fused mixture kernel
// ------------------------------------------------------------------------------------------
struct GmmArgs {
  const float* tab;
  int C;
  int logits_off, mu_off, sig_off;  // tab offsets; mu/sig tables are [C][D]
  int r_off, r_len, y_off;
  key2 key;
  int64_t K, offset;
  float* choices;  // row 0: z, rows 1..D: x
  float* score;
  float* weight;
  float* logw;
  const float* logw_in;
  const float* sub;
  unsigned long long* partials;
  unsigned* ticket;
  float* lse;
  float log_k_total;
  const float* aux;  // prepared constants (gjx_program_prepare), FLAT kernel only
  // one-launch importance step (STEP kernels): systematic comb offset, outputs, granule areas, control words
  double u;
  float* rows_out;              // [1 + D][K] resampled particles
  int32_t* ancestors;           // [K]
  unsigned long long* agg;      // 4 granule arrays of gridDim.x words each
  unsigned* ctrl;
  unsigned long long* timeline; // debug (gjx_debug_timeline): 8 realtime stamps per block
  // TILES kernels: {S_b, e_b} of every 1024-particle tile under GJX_WEIGHTS_TILE_SCALED, for gjx_resample_gather_tiled
  unsigned long long* tile_S;   // [K / 1024]
  int32_t* tile_E;              // [K / 1024]
};

template <int PPT>
struct VecStore;
template <>
struct VecStore<1> {
  static GJX_DEV void st(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct VecStore<2> {
  static GJX_DEV void st(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <>
struct VecStore<4> {
  static GJX_DEV void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// Write-through (sc1) 16-byte stores and L1-bypassing loads for data that other blocks of the SAME launch read or wrote
// (MI355X guide, G16 form R1).  Raw buffer accesses: base in a resource descriptor, the row offset in an SGPR, one
// 32-bit lane offset for all rows (the array must be smaller than 4 GiB; the launcher checks).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;   // cache-policy bits of the buffer builtins on gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1
GJX_DEV __amdgpu_buffer_rsrc_t buffer_of(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
GJX_DEV void store4_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t lane_off, uint32_t row_off, const float (&v)[4]) {
  const u32x4 x = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)lane_off, (int)row_off, kAuxSc1);
  // A buffer store of more than 8 bytes whose soffset is an SGPR reads its data late: a VALU write of the data
  // registers needs 2 wait states behind it.  The compiler's hazard recogniser inserts them, but not across the inline
  // asm / scheduling barrier that closes a pair iteration (seen as lanes 12-15 of every 16 storing the next
  // iteration's values), so they are spelled out here.
  asm volatile("s_nop 1");
}
GJX_DEV float load_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t lane_off, uint32_t row_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_off, (int)row_off, kAuxSc1));
}
GJX_DEV int32_t load_sc1(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Each lane owns PPT consecutive particles (so every SoA row is written with PPT*4-byte stores);
// a block walks tiles of THREADS*PPT particles.  K % PPT == 0 is required by the launcher when PPT > 1.
//
// VALU budget (measured on MI355X, cycles per wave-instruction): v_add_u32 / v_xor_b32 / fp32
// add,mul,fma = 2; v_alignbit_b32 and every other VOP3 integer op = 4; v_log/v_exp/v_rcp/v_sqrt = 8.
// One Threefry hash is therefore ~190 cycles per wave and yields two draws; a normal costs ~50 more.
// The kernel is bound by that integer work, not by HBM — DESIGN.md §roofline.
template <int RNG, int D, int PPT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_run_gmm(GmmArgs a) {
  constexpr int DS = D + 4;  // padded LDS row stride: rows of different z land on different 16-B slots
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C;
  float* s_mu = smem;                 // [C][DS]
  float* s_sig = s_mu + C * DS;       // [C][DS]
  float* s_logit = s_sig + C * DS;    // [C] raw logits
  float* s_race = s_logit + C;        // [C] ln2 * exp(-logit): exponential-race rate^-1
  float* s_zlp = s_race + C;          // [C] log_softmax(logits)[c] - sum_d log sigma[c][d] - D*0.5*log(2pi)
  float* s_cdf = s_zlp + C;           // [C] running sums of exp(logit - max) (FLAT: inverse-CDF draw)
  float* s_y = s_cdf + C;             // [D]
  float* s_rr = s_y + D;              // [D] 1/r
  float* s_misc = s_rr + D;           // [0]: -sum_d log r_d - D*0.5*log(2pi);  [8..]: reduction scratch
  const float* __restrict__ tab = a.tab;

  // prologue, spread over the whole block: every transcendental of the per-component constants is computed by a
  // different lane; only short LDS sums stay serial
  float* s_lsig = s_misc + 32;        // [C][DS] log sigma
  float* s_lr = s_lsig + C * DS;      // [D] log r
  for (int t = threadIdx.x; t < C * D; t += THREADS) {
    const int c = t / D, d = t % D;
    const float sg = tab[a.sig_off + t];
    s_mu[c * DS + d] = tab[a.mu_off + t];
    s_sig[c * DS + d] = sg;
    s_lsig[c * DS + d] = fast_log(sg);
  }
  for (int t = threadIdx.x; t < D; t += THREADS) {
    const float r = tab[a.r_off + (a.r_len == 1 ? 0 : t)];
    s_y[t] = tab[a.y_off + t];
    s_rr[t] = fast_rcp(r);
    s_lr[t] = fast_log(r);
  }
  for (int t = threadIdx.x; t < C; t += THREADS) {
    const float l = tab[a.logits_off + t];
    s_logit[t] = l;
    s_race[t] = kLn2 * fast_exp(-l);
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // wave 0: log-softmax and per-component constants
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 64) mx = fmaxf(mx, s_logit[c]);
    mx = wave_max(mx);
    float se = 0.0f;
    for (int c = threadIdx.x; c < C; c += 64) se += fast_exp(s_logit[c] - mx);
    se = wave_sum(se);
    const float lse = mx + fast_log(se);
    for (int c = threadIdx.x; c < C; c += 64) {
      float sl = 0.0f;
      for (int d = 0; d < D; ++d) sl += s_lsig[c * DS + d];
      s_zlp[c] = (s_logit[c] - lse) - sl - (float)D * kHalfLog2Pi;
    }
    if (threadIdx.x == 63) {
      float sl = 0.0f;
      for (int d = 0; d < D; ++d) sl += s_lr[d];
      s_misc[0] = -sl - (float)D * kHalfLog2Pi;
    }
    if (threadIdx.x == 0) {
      float run = 0.0f;  // same float32 order as the generic interpreter and the oracle
      for (int c = 0; c < C; ++c) { run += fast_exp(s_logit[c] - mx); s_cdf[c] = run; }
    }
  }
  __syncthreads();

  const int64_t K = a.K;
  const int64_t tile = (int64_t)THREADS * PPT;
  const int64_t ntiles = (K + tile - 1) / tile;
  const uint64_t goff = (uint64_t)a.offset;
  // FLAT: the launcher guarantees that all particles of this launch share the high index word
  const key2 fkey = (RNG == GJX_RNG_FLAT && (goff >> 32)) ? threefry2x32(a.key, 0xFFFFFFFFu, (uint32_t)(goff >> 32)) : a.key;
  float tmax = -INFINITY;   // running per-thread max / sum for the block's LSE partial
  float tsum = 0.0f;
  for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const int64_t i0 = tix * tile + (int64_t)threadIdx.x * PPT;
    bool act[PPT];
    uint32_t c0[PPT];          // FLAT: low word of the global particle index
    key2 sk1[PPT], sk2[PPT];   // JAX32: site keys
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      act[p] = (i0 + p) < K;
      const uint64_t gi = goff + (uint64_t)(act[p] ? i0 + p : K - 1);
      c0[p] = (uint32_t)gi;
      if (RNG == GJX_RNG_JAX32) {
        const key2 pk = fold_in64(a.key, gi);
        sk1[p] = fold_in(pk, 1u);
        sk2[p] = fold_in(pk, 2u);
      }
    }
    // ---- z ~ categorical(logits).
    //   FLAT : inverse CDF on one uniform (first c whose running sum exceeds u * total).
    //   JAX32: Gumbel-max argmax_c (l_c - log(-log u_c)), evaluated as the equivalent exponential race
    //          argmin_c (-log u_c) * exp(-l_c): one log per category.
    int z[PPT];
    float zf[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      int best = 0;
      if (RNG == GJX_RNG_FLAT) {
        const key2 h = threefry2x32(fkey, c0[p], (1u << GJX_FLAT_SITE_SHIFT));
        const float target = (__uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, h.a, 9)) - 1.0f) * s_cdf[C - 1];
        for (int c = 0; c < C - 1; ++c) best += (s_cdf[c] > target) ? 0 : 1;
      } else {
        float bestv = INFINITY;
        for (int c = 0; c < C; ++c) {
          const key2 h = threefry2x32(sk1[p], 0u, (uint32_t)c);
          // u = uniform(tiny, 1) = f + tiny (f*(1-tiny) == f in fp32; max(tiny, .) is a no-op)
          const float u = (__uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, h.a ^ h.b, 9)) - 1.0f) + kTiny;
          const float e = -__builtin_amdgcn_logf(u) * s_race[c];
          if (e < bestv) { bestv = e; best = c; }
        }
      }
      z[p] = best;
      zf[p] = (float)best;
    }
    const bool full = (i0 + PPT) <= K;
    float* ch = a.choices;
    if (full) VecStore<PPT>::st(ch + i0, zf);
    else {
#pragma unroll
      for (int p = 0; p < PPT; ++p) if (act[p]) ch[i0 + p] = zf[p];
    }
    // ---- x ~ N(mu[z], sigma[z]); y | x ~ N(x, r) observed ----
    float qx[PPT], qy[PPT];  // sums of squared z-scores
#pragma unroll
    for (int p = 0; p < PPT; ++p) { qx[p] = 0.0f; qy[p] = 0.0f; }
#pragma unroll 2
    for (int d0 = 0; d0 < D; d0 += 2) {
      float xa[PPT], xb[PPT];
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        uint32_t b0, b1 = 0u;
        if (RNG == GJX_RNG_JAX32) {
          const key2 h0 = threefry2x32(sk2[p], 0u, (uint32_t)d0);
          b0 = h0.a ^ h0.b;
          if (d0 + 1 < D) { const key2 h1 = threefry2x32(sk2[p], 0u, (uint32_t)(d0 + 1)); b1 = h1.a ^ h1.b; }
        } else {
          const key2 h = threefry2x32(fkey, c0[p], (2u << GJX_FLAT_SITE_SHIFT) | (uint32_t)(d0 >> 1));
          b0 = h.a; b1 = h.b;
        }
        const int zo = z[p] * DS + d0;
        float n0, n1;
        if (RNG == GJX_RNG_FLAT) box_muller(b0, b1, n0, n1);
        else { n0 = normal_from_bits_fast(b0); n1 = (d0 + 1 < D) ? normal_from_bits_fast(b1) : 0.0f; }
        {
          const float x = fmaf(s_sig[zo], n0, s_mu[zo]);
          qx[p] = fmaf(n0, n0, qx[p]);  // ((x - mu)/sigma)^2 up to rounding
          const float zy = (s_y[d0] - x) * s_rr[d0];
          qy[p] = fmaf(zy, zy, qy[p]);
          xa[p] = x;
        }
        if (d0 + 1 < D) {
          const float x = fmaf(s_sig[zo + 1], n1, s_mu[zo + 1]);
          qx[p] = fmaf(n1, n1, qx[p]);
          const float zy = (s_y[d0 + 1] - x) * s_rr[d0 + 1];
          qy[p] = fmaf(zy, zy, qy[p]);
          xb[p] = x;
        } else xb[p] = 0.0f;
      }
      float* r0 = ch + (int64_t)(1 + d0) * K + i0;
      if (full) {
        VecStore<PPT>::st(r0, xa);
        if (d0 + 1 < D) VecStore<PPT>::st(r0 + K, xb);
      } else {
#pragma unroll
        for (int p = 0; p < PPT; ++p) if (act[p]) { r0[p] = xa[p]; if (d0 + 1 < D) r0[K + p] = xb[p]; }
      }
    }
    float sc[PPT], wt[PPT], lw[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      wt[p] = fmaf(-0.5f, qy[p], s_misc[0]);
      sc[p] = fmaf(-0.5f, qx[p], s_zlp[z[p]]) + wt[p];
      float l = wt[p];
      const int64_t ii = act[p] ? i0 + p : K - 1;
      if (a.logw_in) l += a.logw_in[ii];
      if (a.sub) l -= a.sub[ii];
      lw[p] = l;
    }
    if (full) {
      if (a.score) VecStore<PPT>::st(a.score + i0, sc);
      if (a.weight) VecStore<PPT>::st(a.weight + i0, wt);
      if (a.logw) VecStore<PPT>::st(a.logw + i0, lw);
    } else {
#pragma unroll
      for (int p = 0; p < PPT; ++p) if (act[p]) {
        if (a.score) a.score[i0 + p] = sc[p];
        if (a.weight) a.weight[i0 + p] = wt[p];
        if (a.logw) a.logw[i0 + p] = lw[p];
      }
    }
    // online {max, sum} per thread
#pragma unroll
    for (int p = 0; p < PPT; ++p) if (act[p]) {
      const float nm = fmaxf(tmax, lw[p]);
      if (nm > -INFINITY) tsum = tsum * fast_exp(tmax - nm) + fast_exp(lw[p] - nm);
      tmax = nm;
    }
  }
  if (a.partials) {
    constexpr int NW = THREADS / 64;
    float* red = s_misc + 8;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float wm = wave_max(tmax);
    const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
    if (lane == 0) { red[wid] = wm; red[NW + wid] = ws; }
    __syncthreads();
    float bm = red[0];
    for (int w = 1; w < NW; ++w) bm = fmaxf(bm, red[w]);
    float bsum = 0.0f;
    for (int w = 0; w < NW; ++w) bsum += bm > -INFINITY ? red[NW + w] * fast_exp(red[w] - bm) : 0.0f;
    if (a.lse) lse_publish_and_finish<THREADS>(bm, bsum, a.partials, a.ticket, (int)gridDim.x, a.log_k_total, a.lse, red);
    else if (threadIdx.x == 0) a.partials[blockIdx.x] = pack_f2(bm, bsum);  // consumer finishes (gjx_weight_cumsum mode 2)
  }
}

// ------------------------------------------------------------------------------------------
// fused mixture kernel, FLAT stream (the headline kernel of BASELINE config 2)
//
// What bounds it (measured on MI355X, profiles/microbench/ub4.hip, whole-SIMD throughput in shader cycles per wave-instruction):
// v_add/xor/sub/and/or/lshr/mov and fp32 add/mul/fma 2.3; every VOP3 integer op, v_lshlrev, v_max_f32, v_cvt, v_cmp,
// v_pk_* and any VOP2 with an SGPR source 4.2; v_log/exp/sqrt/rcp/sin/cos 8.3; and a stream that alternates the two
// integer classes pays ~2.3 extra per Threefry round: one Threefry-2x32-20 hash = 272 cycles however it is ordered,
// placed or interleaved (1-8 waves per SIMD, ILP 1-8, same or different register banks).  The kernel therefore spends
// its time in the hash and everything here serves to need fewer VALU cycles per particle:
//   * the bit-packed FLAT stream (23 bits per draw): 1 + ceil((9 + 23 D) / 64) hashes per particle instead of 1 + D/2;
//   * every per-component constant comes precomputed from gjx_program_prepare (the prologue is one global -> LDS copy
//     whose latency hides behind the first tile's categorical hash);
//   * z by counting sign bits of (target - cdf[c]) (sub, lshr, add: three full-rate ops instead of cmp + cndmask/addc);
//   * x = mu + sigma n and the observed z-score a - b n (a = (y - mu) / r, b = sigma / r) are one fma each, and
//     sum_d n_d^2 of a Box-Muller pair is its squared radius -2 ln u1, which is already there.
// Aux layout (floats): T[C][4 D + 4] with T[c][4 d .. 4 d + 3] = {mu, sigma, a, b} of (c, d) — one ds_read_b128 per
// (particle, dimension), one address register per particle with the dimension in the instruction's offset field, rows
// of different z on different LDS banks — then zlp[C], cdf[C], misc[8].
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int gmm_aux_floats(int C, int D) { return C * (4 * D + 4) + 2 * C + 8; }

__global__ __launch_bounds__(256) void k_gmm_prepare(GmmArgs a, int D, float* aux) {
  const int C = a.C, RS = 4 * D + 4;
  const float* __restrict__ tab = a.tab;
  float* s_t = aux; float* s_zlp = s_t + C * RS; float* s_cdf = s_zlp + C; float* s_misc = s_cdf + C;
  for (int t = threadIdx.x; t < C * (D + 1); t += 256) {
    const int c = t / (D + 1), d = t % (D + 1);
    float mu = 0.0f, sg = 0.0f, av = 0.0f, bv = 0.0f;
    if (d < D) {
      mu = tab[a.mu_off + c * D + d];
      sg = tab[a.sig_off + c * D + d];
      const float rr = fast_rcp(tab[a.r_off + (a.r_len == 1 ? 0 : d)]);
      av = (tab[a.y_off + d] - mu) * rr;
      bv = sg * rr;
    }
    float* q = s_t + c * RS + 4 * d;
    q[0] = mu; q[1] = sg; q[2] = av; q[3] = bv;
  }
  if (threadIdx.x < 64) {  // wave 0: log-softmax, per-component log-sigma sums, running CDF (float32 category order)
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 64) mx = fmaxf(mx, tab[a.logits_off + c]);
    mx = wave_max(mx);
    float se = 0.0f;
    for (int c = threadIdx.x; c < C; c += 64) se += fast_exp(tab[a.logits_off + c] - mx);
    se = wave_sum(se);
    const float lse = mx + fast_log(se);
    for (int c = threadIdx.x; c < C; c += 64) {
      float sl = 0.0f;
      for (int d = 0; d < D; ++d) sl += fast_log(tab[a.sig_off + c * D + d]);
      s_zlp[c] = (tab[a.logits_off + c] - lse) - sl - (float)D * kHalfLog2Pi;
    }
    if (threadIdx.x == 63) {
      float sl = 0.0f;
      for (int d = 0; d < D; ++d) sl += fast_log(tab[a.r_off + (a.r_len == 1 ? 0 : d)]);
      s_misc[0] = -sl - (float)D * kHalfLog2Pi;
      for (int j = 1; j < 8; ++j) s_misc[j] = 0.0f;
    }
    if (threadIdx.x == 0) {
      float run = 0.0f;  // same float32 order as the generic interpreter and the oracle
      for (int c = 0; c < C; ++c) { run += fast_exp(tab[a.logits_off + c] - mx); s_cdf[c] = run; }
    }
  }
}

template <int D, int PPT, int THREADS, bool STEP = false, bool TILES = false>
__global__ __launch_bounds__(THREADS) void k_run_gmm_flat(GmmArgs a) {
  static_assert(!STEP || (PPT == 4 && THREADS == 256), "the one-launch step works on tiles of 256 x 4 particles");
  static_assert(!TILES || (PPT == 4 && THREADS == 256 && !STEP), "tile totals: a block's tile is one quantisation tile of 1024 particles");
  __shared__ float red_tm[4];
  __shared__ uint64_t red_tq[4];
  constexpr int RS = 4 * D + 4;       // LDS row stride of one component (floats)
  constexpr int NPAIR = (D + 1) / 2;
  unsigned epoch = 0;
  unsigned long long tag = 0;
  if (STEP && a.timeline && threadIdx.x == 0) a.timeline[blockIdx.x * 8] = __builtin_amdgcn_s_memrealtime();
  if (STEP) tag = grid_tag(a.ctrl, &epoch);   // before anything is published: block 0 bumps it once every block is known to run
  const __amdgpu_buffer_rsrc_t rs_ch = buffer_of(a.choices, STEP ? (uint32_t)((1 + D) * a.K * 4) : 0u);
  constexpr int NHASH = (9 + 23 * (D == 1 ? 2 : D) + 63) / 64;   // blocks of the x site's stream the draws touch
  constexpr bool UNROLLED = D <= 16;                 // larger D: rolled loop over pairs through BitStream (code size)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C;
  float* s_t = smem; float* s_zlp = s_t + C * RS; float* s_cdf = s_zlp + C; float* s_misc = s_cdf + C;
  const int64_t K = a.K;
  const int64_t tile = (int64_t)THREADS * PPT;
  const int64_t ntiles = (K + tile - 1) / tile;
  const uint64_t goff = (uint64_t)a.offset;
  // the launcher guarantees that all particles of this launch share the high index word
  const key2 fkey = (goff >> 32) ? threefry2x32(a.key, 0xFFFFFFFFu, (uint32_t)(goff >> 32)) : a.key;

  // ---- prologue: request the prepared table, hash while it is in flight, then fill LDS ----
  const int naux = gmm_aux_floats(C, D);
  constexpr int NPRE = 4;
  float pre[NPRE];
#pragma unroll
  for (int j = 0; j < NPRE; ++j) {
    const int t = threadIdx.x + j * THREADS;
    pre[j] = t < naux ? a.aux[t] : 0.0f;
  }
  int64_t tix = blockIdx.x;
  uint32_t hz[PPT];   // categorical draw of the next tile (element 0 of site 1)
  auto cat_bits = [&](int64_t t) {
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      int64_t i = t * tile + (int64_t)threadIdx.x * PPT + p;
      i = i < K ? i : K - 1;
      hz[p] = threefry2x32(fkey, (uint32_t)(goff + (uint64_t)i), (1u << GJX_FLAT_SITE_SHIFT)).a;
    }
  };
  if (tix < ntiles) cat_bits(tix);
#pragma unroll
  for (int j = 0; j < NPRE; ++j) {
    const int t = threadIdx.x + j * THREADS;
    if (t < naux) smem[t] = pre[j];
  }
  for (int t = threadIdx.x + NPRE * THREADS; t < naux; t += THREADS) smem[t] = a.aux[t];
  __syncthreads();

  float tmax = -INFINITY;   // running per-thread max / sum for the block's LSE partial
  float tsum = 0.0f;
  float lw_tile[PPT];       // STEP: the tile's log-weights stay in registers for the resampling phases
#pragma unroll
  for (int p = 0; p < PPT; ++p) lw_tile[p] = -INFINITY;
  const float cdf_total = s_cdf[C - 1];
  for (; tix < ntiles; tix += gridDim.x) {
    const int64_t i0 = tix * tile + (int64_t)threadIdx.x * PPT;
    // K % PPT == 0 (launcher), so a lane's PPT particles are all inside or all outside; lanes past the end are done
    // (their later tiles lie past the end too) and take no part in anything but the final reduction
    if (i0 >= K) break;
    uint32_t c0[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) c0[p] = (uint32_t)(goff + (uint64_t)(i0 + p));
    // ---- z ~ categorical(logits): inverse CDF on one uniform = number of c < C-1 with cdf[c] <= target ----
    int z[PPT];
    float zf[PPT], target[PPT];
    uint32_t neg[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      target[p] = (__uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, hz[p], 9)) - 1.0f) * cdf_total;
      neg[p] = 0u;
    }
    for (int c = 0; c < C - 1; ++c) {
      const float cd = s_cdf[c];
#pragma unroll
      for (int p = 0; p < PPT; ++p) neg[p] += __float_as_uint(target[p] - cd) >> 31;   // cdf[c] > target
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) { z[p] = (C - 1) - (int)neg[p]; zf[p] = (float)z[p]; }
    float* ch = a.choices;
    if constexpr (STEP) store4_sc1(rs_ch, (uint32_t)i0 * 4u, 0u, zf); else VecStore<PPT>::st(ch + i0, zf);
    // ---- x ~ N(mu[z], sigma[z]); y | x ~ N(x, r) observed ----
    float qx[PPT], qy[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) { qx[p] = 0.0f; qy[p] = 0.0f; }
    uint32_t w[PPT][UNROLLED ? 2 * NHASH : 2];
    BitStreamRT<GJX_RNG_FLAT> bs[UNROLLED ? 1 : PPT];
    if (!UNROLLED) {
#pragma unroll
      for (int p = 0; p < PPT; ++p) { bs[p].key = fkey; bs[p].c0 = c0[p]; bs[p].site_hi = 2u << GJX_FLAT_SITE_SHIFT; bs[p].h0 = bs[p].h1 = 0xFFFFFFFFu; }
    }
    constexpr int UF = UNROLLED ? NPAIR : 1;
#pragma unroll UF
    for (int k = 0; k < NPAIR; ++k) {            // k is a constant after unrolling when UNROLLED
      const int d0 = 2 * k;
      float xa[PPT], xb[PPT];
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        uint32_t wa, wb;
        if constexpr (UNROLLED) {
          const int last = D == 1 ? 1 : 2 * k + 1;                           // highest element this pair reads
          const int need = (((23 * last) >> 5) + ((((23 * last) & 31) != 0) ? 1 : 0)) >> 1;   // highest block it touches
          const int prev_last = 2 * (k - 1) + 1;
          const int have = k == 0 ? -1 : ((((23 * prev_last) >> 5) + ((((23 * prev_last) & 31) != 0) ? 1 : 0)) >> 1);
#pragma unroll
          for (int h = 0; h < NHASH; ++h) if (h > have && h <= need) {
            const key2 hh = threefry2x32(fkey, c0[p], (2u << GJX_FLAT_SITE_SHIFT) | (uint32_t)h);
            w[p][2 * h] = hh.a; w[p][2 * h + 1] = hh.b;
          }
          wa = GJX_FIELD(w[p], 2 * k);
          wb = GJX_FIELD(w[p], 2 * k + 1);
        } else {
          wa = bs[p].get((uint32_t)d0);
          wb = bs[p].get((uint32_t)d0 + 1u);
        }
        // Box-Muller: u1 = 2 - [1,2) in (0,1]; v_sin/v_cos take revolutions and are periodic, so [1,2) feeds them as is
        const float u1 = 2.0f - __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wa, 9));
        const float rsq = __builtin_amdgcn_logf(u1) * (-2.0f * kLn2);
        const float r = fast_sqrt(rsq);
        const float4* row = reinterpret_cast<const float4*>(s_t + z[p] * RS) + d0;   // {mu, sigma, a, b} of (z, d0)
        if (D == 1) {
          // one draw: the cosine branch of the pair whose angle is element 1 of the stream (the generic rule)
          const float u2 = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wb, 9));
          const float n0 = r * __builtin_amdgcn_cosf(u2);
          const float4 t0 = row[0];
          const float x = fmaf(t0.y, n0, t0.x);
          qx[p] = fmaf(n0, n0, qx[p]);
          const float zy = fmaf(-t0.w, n0, t0.z);
          qy[p] = fmaf(zy, zy, qy[p]);
          xa[p] = x; xb[p] = 0.0f;
        } else {
          const float u2 = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wb, 9));
          const float n0 = r * __builtin_amdgcn_cosf(u2), n1 = r * __builtin_amdgcn_sinf(u2);
          const float4 t0 = row[0], t1 = row[1];
          xa[p] = fmaf(t0.y, n0, t0.x);
          xb[p] = fmaf(t1.y, n1, t1.x);
          qx[p] += rsq;                                   // n0^2 + n1^2 = r^2
          const float zy0 = fmaf(-t0.w, n0, t0.z), zy1 = fmaf(-t1.w, n1, t1.z);
          qy[p] = fmaf(zy0, zy0, qy[p]);
          qy[p] = fmaf(zy1, zy1, qy[p]);
        }
      }
      float* r0 = ch + (int64_t)(1 + d0) * K + i0;
      if constexpr (STEP) {
        store4_sc1(rs_ch, (uint32_t)i0 * 4u, (uint32_t)(1 + d0) * (uint32_t)K * 4u, xa);
        if (D > 1) store4_sc1(rs_ch, (uint32_t)i0 * 4u, (uint32_t)(2 + d0) * (uint32_t)K * 4u, xb);
      }
      else { VecStore<PPT>::st(r0, xa); if (D > 1) VecStore<PPT>::st(r0 + K, xb); }
      // keeps the work of pair k+1 out of pair k.  Without the scheduling barrier every hash of the tile moves to the
      // top (170+ VGPRs); without pinning the two accumulators here instruction selection parks the whole
      // z-score / sum-of-squares chain (it has no side effect until the tile's last store) behind the last barrier and
      // keeps a, b and n of all D dimensions alive until then (250 VGPRs, 1-2 waves per SIMD)
#pragma unroll
      for (int p = 0; p < PPT; ++p) asm volatile("" : "+v"(qx[p]), "+v"(qy[p]));
      __builtin_amdgcn_sched_barrier(0);
    }
    float sc[PPT], wt[PPT], lw[PPT];
    const float wconst = s_misc[0];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      wt[p] = fmaf(-0.5f, qy[p], wconst);
      sc[p] = fmaf(-0.5f, qx[p], s_zlp[z[p]]) + wt[p];
      float l = wt[p];
      if (a.logw_in) l += a.logw_in[i0 + p];
      if (a.sub) l -= a.sub[i0 + p];
      lw[p] = l;
    }
    if (a.score) VecStore<PPT>::st(a.score + i0, sc);
    if (a.weight) VecStore<PPT>::st(a.weight + i0, wt);
    if (a.logw) VecStore<PPT>::st(a.logw + i0, lw);
    if (STEP) {
#pragma unroll
      for (int p = 0; p < PPT; ++p) lw_tile[p] = lw[p];
    }
    // online {max, sum} per thread: one rescale per tile
    float m4 = tmax;
#pragma unroll
    for (int p = 0; p < PPT; ++p) m4 = fmaxf(m4, lw[p]);
    if (m4 > -INFINITY) {
      float s4 = tsum * fast_exp(tmax - m4);
#pragma unroll
      for (int p = 0; p < PPT; ++p) s4 += fast_exp(lw[p] - m4);
      tsum = s4;
    }
    tmax = m4;
    if constexpr (TILES) {
      // {e_b, S_b} of this tile (GJX_WEIGHTS_TILE_SCALED, include/gjx.h; K % 1024 == 0: every wave of the block is here):
      // the resampling kernel that follows needs no grid-wide exchange of its own (gjx_resample_gather_tiled)
      const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
      float m = lw[0];
#pragma unroll
      for (int p = 1; p < PPT; ++p) m = fmaxf(m, lw[p]);
      const float wm = wave_max(m);
      if (lane == 0) red_tm[wid] = wm;
      __syncthreads();
      const int e = tile_exponent(fmaxf(fmaxf(red_tm[0], red_tm[1]), fmaxf(red_tm[2], red_tm[3])));
      uint64_t q = 0;
#pragma unroll
      for (int p = 0; p < PPT; ++p) q += tile_q(lw[p], e);
      const uint64_t wq = wave_total_u64(q);
      if (lane == 0) red_tq[wid] = wq;
      __syncthreads();
      if (threadIdx.x == 0) {
        const uint64_t tot = red_tq[0] + red_tq[1] + red_tq[2] + red_tq[3];
        a.tile_S[tix] = tot;
        a.tile_E[tix] = tot ? e : kTileDead;
      }
    }
    if (tix + gridDim.x < ntiles) cat_bits(tix + gridDim.x);
  }
  constexpr int NW = THREADS / 64;
  __shared__ float red[2 * NW + 2];
  float bm = -INFINITY, bsum = 0.0f;
  if (a.partials || STEP) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float wm = wave_max(tmax);
    const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
    if (lane == 0) { red[wid] = wm; red[NW + wid] = ws; }
    __syncthreads();
    bm = red[0];
    for (int w2 = 1; w2 < NW; ++w2) bm = fmaxf(bm, red[w2]);
    for (int w2 = 0; w2 < NW; ++w2) bsum += bm > -INFINITY ? red[NW + w2] * fast_exp(red[w2] - bm) : 0.0f;
  }
  if constexpr (!STEP) {
    if (a.partials) {
      if (a.lse) lse_publish_and_finish<THREADS>(bm, bsum, a.partials, a.ticket, (int)gridDim.x, a.log_k_total, a.lse, red);
      else if (threadIdx.x == 0) a.partials[blockIdx.x] = pack_f2(bm, bsum);  // consumer finishes (gjx_weight_cumsum mode 2)
    }
  } else {
    // =================== one-launch importance step: resample + gather without leaving the kernel ===================
    // Preconditions (launcher): one full tile per block (gridDim.x * 1024 == K), grid co-resident, N == K.
    // Every rendezvous is a tagged-granule all-gather (gjx_scan.h); bulk data that crosses blocks (the particle rows
    // written above, the ancestors) is stored write-through and read past L1 (sc1 on both sides: no fences).
    __shared__ ScanSmem sm;
#define GJX_STAMP(n) do { if (a.timeline && threadIdx.x == 0) a.timeline[blockIdx.x * 8 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    GJX_STAMP(1);
    unsigned long long* aggA0 = a.agg;
    unsigned long long* aggA1 = aggA0 + gridDim.x;
    unsigned long long* aggB = aggA1 + gridDim.x;
    unsigned long long* aggC = aggB + gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * tile + (int64_t)threadIdx.x * PPT;
    // -- A: global {max, sumexp} of the log-weights (same reduction order as block_ref_max over per-block partials)
    if (threadIdx.x == 0) {
      grid_publish(aggA0, tag, (unsigned long long)__float_as_uint(bm));
      grid_publish(aggA1, tag, (unsigned long long)__float_as_uint(bsum));
    }
    // thread t combines blocks t, t + 256, ... in that order, then waves and block as block_ref_max does (same bits as
    // the three-launch path, whose prefix-sum kernel reduces the per-block partials the same way)
    float gmax = -INFINITY, gsum = 0.0f;
    {
      float pm[8];              // gridDim.x <= 2048 (launcher)
      int np = 0;
      grid_gather(aggA0, tag, a.ctrl, [&](int, unsigned long long v) { pm[np++] = __uint_as_float((uint32_t)v); });
      np = 0;
      grid_gather(aggA1, tag, a.ctrl, [&](int, unsigned long long v) {
        const float px = pm[np++], py = __uint_as_float((uint32_t)v);
        const float nm = fmaxf(gmax, px);
        if (nm > -INFINITY) gsum = gsum * fast_exp(gmax - nm) + py * fast_exp(px - nm);
        gmax = nm;
      });
    }
    {
      const float wm = wave_max(gmax);
      const float ws = wave_sum(wm > -INFINITY ? gsum * fast_exp(gmax - wm) : 0.0f);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wm; red[NW + (threadIdx.x >> 6)] = ws; }
      __syncthreads();
      gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      gsum = 0.0f;
      for (int w2 = 0; w2 < 4; ++w2) gsum += gmax > -INFINITY ? red[NW + w2] * fast_exp(red[w2] - gmax) : 0.0f;
    }
    if (a.lse && blockIdx.x == 0 && threadIdx.x == 0) {
      const float l = gmax > -INFINITY ? gmax + logf(gsum) : -INFINITY;
      a.lse[0] = gmax; a.lse[1] = gsum; a.lse[2] = l; a.lse[3] = l - a.log_k_total;
    }
    GJX_STAMP(2);
    // -- B: fixed-point weights, tile scan, all-gather of the tile totals, ancestors of the slots this tile owns
    tile_scan_expand<PPT, true>(lw_tile, 1, gmax, i0, K, a.u, K, a.ancestors, nullptr, nullptr, aggB, tag, a.ctrl, epoch, false, sm);
    GJX_STAMP(3);
    // -- C: every block's ancestors are out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) grid_publish(aggC, tag, 1ull);
    grid_gather(aggC, tag, a.ctrl, [&](int, unsigned long long) {});
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(&a.ctrl[0], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every block has read `epoch` long ago
    GJX_STAMP(4);
    // -- D: slot-oriented copy of this block's 1024 output slots (coalesced 16-byte stores; monotone ancestors make
    //       the reads of neighbouring lanes touch neighbouring particles)
    int32_t an[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) an[p] = load_sc1(a.ancestors + i0 + p);
    constexpr int R = 1 + D;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
      float v[PPT];
#pragma unroll
      for (int p = 0; p < PPT; ++p) v[p] = load_sc1(rs_ch, (uint32_t)an[p] * 4u, (uint32_t)r * (uint32_t)K * 4u);
      VecStore<PPT>::st(a.rows_out + (int64_t)r * K + i0, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GJX_STAMP(5);
#undef GJX_STAMP
  }
}

// ------------------------------------------------------------------------------------------
// LSE: partials -> {max, sumexp, lse, lse - log K_total}
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lse_finish(const float2* partials, int n, float log_k_total, float* out) {
  __shared__ float red[8];
  float m = -INFINITY;
  for (int t = threadIdx.x; t < n; t += 256) m = fmaxf(m, partials[t].x);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.0f;
  for (int t = threadIdx.x; t < n; t += 256) {
    const float2 p = partials[t];
    if (p.x > -INFINITY) s += p.y * fast_exp(p.x - bm);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float bs = red[4] + red[5] + red[6] + red[7];
    const float lse = bm > -INFINITY ? bm + logf(bs) : -INFINITY;
    out[0] = bm; out[1] = bs; out[2] = lse; out[3] = lse - log_k_total;
  }
}

__global__ __launch_bounds__(256) void k_lse_partial(const float* x, int64_t K, float2* partials) {
  __shared__ float red[8];
  float tmax = -INFINITY, tsum = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < K; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    const float nm = fmaxf(tmax, v);
    if (nm > -INFINITY) tsum = tsum * fast_exp(tmax - nm) + fast_exp(v - nm);
    tmax = nm;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float wm = wave_max(tmax);
  const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
  if (lane == 0) { red[wid] = wm; red[4 + wid] = ws; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float bs = 0.0f;
    for (int w = 0; w < 4; ++w) bs += bm > -INFINITY ? red[4 + w] * fast_exp(red[w] - bm) : 0.0f;
    partials[blockIdx.x] = make_float2(bm, bs);
  }
}

}  // namespace gjx

using namespace gjx;

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

// engine ids reported by gjx_program_engine
enum { ENGINE_GENERIC = 0, ENGINE_GMM = 1, ENGINE_GEN = 4 };   // 2, 3: the HMC engines (gjx_hmc_engine)

struct GmmShape {
  int C, D, logits_off, mu_off, sig_off, r_off, r_len, y_off;
};

// Recognise: [categorical_logits(CONST) SAMPLE] -> [mvnormal_diag(GATHER(z), GATHER(z)) SAMPLE]
//            -> [mvnormal_diag(VALUE(x), CONST) OBS_TAB], slots z=0, x=1..D.
bool match_gmm(const gjx_program* p, GmmShape* g) {
  if (p->n_sites != 3) return false;
  for (int j = 0; j < 3; ++j) if ((p->sites[j].flags & GJX_SITE_PROPOSAL) || p->sites[j].plate != 0) return false;
  const gjx_site& s0 = p->sites[0];
  const gjx_site& s1 = p->sites[1];
  const gjx_site& s2 = p->sites[2];
  if (s0.kind != GJX_CATEGORICAL_LOGITS || s0.mode != GJX_MODE_SAMPLE || s0.slot != 0) return false;
  if (s0.p[0].op != GJX_P_CONST || s0.p[0].xf != GJX_XF_NONE || s0.p[0].len != s0.ncat) return false;
  if ((s1.kind != GJX_MVNORMAL_DIAG && s1.kind != GJX_NORMAL) || s1.mode != GJX_MODE_SAMPLE || s1.slot != 1) return false;
  const int D = s1.dim, C = s0.ncat;
  for (int k = 0; k < 2; ++k) {
    const gjx_param& q = s1.p[k];
    if (q.op != GJX_P_GATHER || q.xf != GJX_XF_NONE || q.slot != 0 || q.n != C || q.len != D) return false;
  }
  if ((s2.kind != GJX_MVNORMAL_DIAG && s2.kind != GJX_NORMAL) || s2.mode != GJX_MODE_OBS_TAB || s2.dim != D) return false;
  if (s2.p[0].op != GJX_P_VALUE || s2.p[0].xf != GJX_XF_NONE || s2.p[0].slot != 1 || s2.p[0].len != D) return false;
  if (s2.p[1].op != GJX_P_CONST || s2.p[1].xf != GJX_XF_NONE || (s2.p[1].len != 1 && s2.p[1].len != D)) return false;
  if (p->n_slots != 1 + D) return false;
  if (C < 1 || C > 64) return false;
  if (!(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32 || D == 64)) return false;
  g->C = C; g->D = D;
  g->logits_off = s0.p[0].off; g->mu_off = s1.p[0].off; g->sig_off = s1.p[1].off;
  g->r_off = s2.p[1].off; g->r_len = s2.p[1].len; g->y_off = s2.obs_off;
  return true;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// the caller's HIP events (gjx_run_opts.start_event / stop_event, gjx_importance_step_ex) are attached to the DISPATCH of the
// kernel — they take its own begin / end timestamps, not those of markers around it
struct DispatchEvents { hipEvent_t start = nullptr, stop = nullptr; };

template <class KERN>
void launch_gmm_kernel(KERN kern, const GmmArgs& a, int grid, size_t lds, hipStream_t st, DispatchEvents ev) {
  if (ev.start && ev.stop) {
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), (uint32_t)lds, st, ev.start, ev.stop, 0, a);
    return;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
}
template <int D>
void launch_gmm_d(const GmmArgs& a, bool flat, int ppt, int grid, size_t lds, hipStream_t st, DispatchEvents ev) {
  if (flat) {
    if (ppt == 4 && a.tile_S) launch_gmm_kernel(k_run_gmm_flat<D, 4, 256, false, true>, a, grid, lds, st, ev);
    else if (ppt == 4) launch_gmm_kernel(k_run_gmm_flat<D, 4, 256>, a, grid, lds, st, ev);
    else if (ppt == 2) launch_gmm_kernel(k_run_gmm_flat<D, 2, 256>, a, grid, lds, st, ev);
    else launch_gmm_kernel(k_run_gmm_flat<D, 1, 256>, a, grid, lds, st, ev);
  } else {
    if (ppt == 4) launch_gmm_kernel(k_run_gmm<GJX_RNG_JAX32, D, 4, 256>, a, grid, lds, st, ev);
    else if (ppt == 2) launch_gmm_kernel(k_run_gmm<GJX_RNG_JAX32, D, 2, 256>, a, grid, lds, st, ev);
    else launch_gmm_kernel(k_run_gmm<GJX_RNG_JAX32, D, 1, 256>, a, grid, lds, st, ev);
  }
}
void launch_gmm(const GmmArgs& a, bool flat, int D, int ppt, int grid, size_t lds, hipStream_t st, DispatchEvents ev) {
  switch (D) {
    case 1: launch_gmm_d<1>(a, flat, ppt, grid, lds, st, ev); break;
    case 2: launch_gmm_d<2>(a, flat, ppt, grid, lds, st, ev); break;
    case 4: launch_gmm_d<4>(a, flat, ppt, grid, lds, st, ev); break;
    case 8: launch_gmm_d<8>(a, flat, ppt, grid, lds, st, ev); break;
    case 16: launch_gmm_d<16>(a, flat, ppt, grid, lds, st, ev); break;
    case 32: launch_gmm_d<32>(a, flat, ppt, grid, lds, st, ev); break;
    default: launch_gmm_d<64>(a, flat, ppt, grid, lds, st, ev); break;
  }
}

void fill_gmm_args(GmmArgs& a, const gjx_program* prog, const GmmShape& g) {
  a.tab = prog->tab_dev; a.C = g.C;
  a.logits_off = g.logits_off; a.mu_off = g.mu_off; a.sig_off = g.sig_off;
  a.r_off = g.r_off; a.r_len = g.r_len; a.y_off = g.y_off;
  a.aux = prog->aux_dev;
}

// the fused mixture kernel needs the prepared constants when the stream is FLAT
bool gmm_usable(const gjx_program* p, GmmShape* g) {
  if (env_int("GJX_FORCE_GENERIC", 0) || !match_gmm(p, g)) return false;
  if (p->rng_mode == GJX_RNG_JAX32) return true;
  return p->aux_dev != nullptr && p->n_aux >= gmm_aux_floats(g->C, g->D);
}

}  // namespace

int gjx_launch_lse_finish(const void* partials, int n, int64_t K_total, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_lse_finish, dim3(1), dim3(256), 0, st, (const float2*)partials, n,
                     (float)log((double)K_total), out);
  GJX_CHECK_LAUNCH("lse_finish");
  return GJX_OK;
}

extern "C" int gjx_event_create(void** ev) {
  if (!ev) return gjx_fail(GJX_EINVAL, "gjx_event_create: bad argument");
  hipEvent_t e;
  const hipError_t r = hipEventCreate(&e);
  if (r != hipSuccess) return gjx_fail_hip(r, "gjx_event_create");
  *ev = (void*)e;
  return GJX_OK;
}
extern "C" int gjx_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return GJX_OK;
}
extern "C" int gjx_event_elapsed_us(void* start, void* stop, float* us) {
  if (!start || !stop || !us) return gjx_fail(GJX_EINVAL, "gjx_event_elapsed_us: bad argument");
  hipError_t r = hipEventSynchronize((hipEvent_t)stop);
  if (r != hipSuccess) return gjx_fail_hip(r, "gjx_event_elapsed_us");
  float ms = 0.0f;
  r = hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop);
  if (r != hipSuccess) return gjx_fail_hip(r, "gjx_event_elapsed_us");
  *us = ms * 1000.0f;
  return GJX_OK;
}


// ---- prepared constants -------------------------------------------------------------------------
extern "C" int gjx_program_aux_floats(const gjx_program* prog) {
  if (!prog || !prog->sites) return GJX_EINVAL;
  GmmShape g;
  if (prog->rng_mode == GJX_RNG_FLAT && match_gmm(prog, &g)) return gmm_aux_floats(g.C, g.D);
  return 0;
}

extern "C" int gjx_program_prepare(const gjx_program* prog, float* aux_dev, int32_t n_aux, void* stream) {
  if (!prog || !prog->sites || !prog->tab_dev) return gjx_fail(GJX_EINVAL, "gjx_program_prepare: null program");
  const int need = gjx_program_aux_floats(prog);
  if (need <= 0) return GJX_OK;
  if (!aux_dev || n_aux < need) return gjx_fail(GJX_EINVAL, "gjx_program_prepare: aux buffer too small (gjx_program_aux_floats)");
  GmmShape g;
  match_gmm(prog, &g);
  GmmArgs a = {};
  fill_gmm_args(a, prog, g);
  hipLaunchKernelGGL(k_gmm_prepare, dim3(1), dim3(256), 0, (hipStream_t)stream, a, g.D, aux_dev);
  GJX_CHECK_LAUNCH("gjx_program_prepare");
  return GJX_OK;
}

// Which engine runs a program: the hand-fused mixture kernel, the kernel generated from the site list (hipRTC), or the
// site interpreter.  GJX_ENGINE = auto (default: mixture kernel if the program has that shape, else generated, else
// interpreter) | gen (generated first) | interp; GJX_FORCE_GENERIC=1 is the older spelling of interp.
struct EnginePlan {
  int engine, ppt, grid;
  GmmShape g;
};

static EnginePlan plan_engine(const gjx_program* prog, int64_t K, int64_t particle_offset, bool want_site_scores, bool want_tiles = false) {
  EnginePlan e;
  e.engine = ENGINE_GENERIC; e.ppt = 1; e.grid = (int)((K + 255) / 256);
  const char* env = getenv("GJX_ENGINE");
  const bool interp = env_int("GJX_FORCE_GENERIC", 0) || (env && !strcmp(env, "interp"));
  if (interp) return e;
  const bool gen_first = env && !strcmp(env, "gen");
  const bool same_hi = ((uint64_t)particle_offset >> 32) == ((uint64_t)(particle_offset + K - 1) >> 32);
  bool has_proposal = false;       // proposal sites / sites scored at a proposal's draw: the generic engines only
  for (int j = 0; j < prog->n_sites; ++j) has_proposal = has_proposal || (prog->sites[j].flags & GJX_SITE_PROPOSAL) || prog->sites[j].mode == GJX_MODE_OBS_PROPOSED;
  auto try_gmm = [&]() {
    if (has_proposal || want_site_scores || !same_hi || !gmm_usable(prog, &e.g)) return false;
    int ppt = env_int("GJX_GMM_PPT", 4);
    if (ppt != 1 && ppt != 2 && ppt != 4) ppt = 4;
    if (K % ppt != 0) ppt = 1;  // row bases must stay vector-aligned
    const int64_t tile = 256 * (int64_t)ppt, ntiles = (K + tile - 1) / tile;
    const int maxgrid = env_int("GJX_GMM_GRID", 2048);
    e.engine = ENGINE_GMM; e.ppt = ppt; e.grid = (int)(ntiles < maxgrid ? ntiles : maxgrid);
    return true;
  };
  auto try_gen = [&]() {
    if (env_int("GJX_NO_CODEGEN", 0) || !same_hi) return false;     // (generated kernels take the index's high word as a launch constant)
    const int ppt = gen_pick_ppt(prog, K, want_tiles && K % 1024 == 0);     // (ppt | 256: the matrix-core flavour, gjx_codegen.hip)
    if (gen_available(prog, ppt) != GJX_OK) return false;
    // (ppt | 512: a block of 16 waves shares 64 x ppt particles — the instances of its plates are dealt to the waves)
    const int lpp = (ppt & 2048) ? 16 : ((ppt & 1024) ? 4 : 1);        // (wide flavour: lanes per particle)
    const int64_t tile = ((ppt & 512) ? 64 / lpp : 256) * (int64_t)(ppt & 255), ntiles = (K + tile - 1) / tile;
    // the per-block prologue (table copy + derived constants) is paid once per block: no more blocks than can be resident
    // at 4 per CU, each looping over its tiles
    static const int resident = [] { int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); return 4 * (cus > 0 ? cus : 256); }();
    // (the matrix-core flavour: one block per tile — two 70 KB blocks fit a CU, and 4096 short blocks fill the tail better than
    // 1024 blocks of four tiles: 520 vs 545 us on the config-5 target)
    int maxgrid = env_int("GJX_GEN_GRID", (ppt & 256) ? (1 << 20) : ((ppt & 512) ? 2 * resident : resident));
    // (a workspace has room for one {max, sumexp} pair per 256 particles: no more blocks than that, whatever a block's tile is)
    // (the wide flavour never leaves tile totals behind the pairs, so its pairs may use the workspace's 64 KB of slack: 4096 of them)
    if ((ppt & 512) && maxgrid > (int)((K + 255) / 256) && maxgrid > 4096) maxgrid = (int)((K + 255) / 256) > 4096 ? (int)((K + 255) / 256) : 4096;
    e.engine = ENGINE_GEN; e.ppt = ppt; e.grid = (int)(ntiles < maxgrid ? ntiles : maxgrid);
    return true;
  };
  if (gen_first) { if (!try_gen()) try_gmm(); }
  else if (!try_gmm()) try_gen();
  return e;
}

extern "C" int gjx_program_engine(const gjx_program* prog) {
  if (!prog || !prog->sites) return GJX_EINVAL;
  return plan_engine(prog, 1024, 0, false).engine;
}

extern "C" int gjx_run_partials_count(const gjx_program* prog, int64_t K, int64_t particle_offset) {
  if (!prog || !prog->sites || K <= 0) return GJX_EINVAL;
  return plan_engine(prog, K, particle_offset, false).grid;
}

// gjx_run_program: gjx_run_program_ex without options and without a record (the per-thread one-shot setters of ABI 6 —
// gjx_run_want_tiles, gjx_last_run_partials, gjx_last_run_tiles — are gone: options and record are arguments)
extern "C" int gjx_run_program(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K,
                               int64_t particle_offset, float* choices, float* score, float* weight,
                               float* logw, const float* logw_in, const float* sub,
                               float* site_scores, float* lse, int64_t K_total, void* workspace,
                               size_t workspace_bytes, void* stream) {
  gjx_run_opts o;
  memset(&o, 0, sizeof(o));
  return gjx_run_program_ex(prog, key0, key1, K, particle_offset, choices, score, weight, logw, logw_in, sub, site_scores, lse, K_total,
                            workspace, workspace_bytes, stream, &o, nullptr);
}

extern "C" int gjx_run_program_ex(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K,
                                  int64_t particle_offset, float* choices, float* score, float* weight,
                                  float* logw, const float* logw_in, const float* sub,
                                  float* site_scores, float* lse, int64_t K_total, void* workspace,
                                  size_t workspace_bytes, void* stream, const gjx_run_opts* opts, gjx_run_info* info_out) {
  gjx_run_opts no_opts;
  memset(&no_opts, 0, sizeof(no_opts));
  const gjx_run_opts& op = opts ? *opts : no_opts;
  gjx_run_info info = {0, 0, 0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (op.flags & GJX_RUN_TIME_DISPATCH) { ev0 = (hipEvent_t)op.start_event; ev1 = (hipEvent_t)op.stop_event; }
  if (info_out) *info_out = info;
  if (!prog || !prog->sites || !prog->sites_dev || !prog->tab_dev) return gjx_fail(GJX_EINVAL, "gjx_run_program: null program");
  if (K < 0 || prog->n_sites < 0) return gjx_fail(GJX_EINVAL, "gjx_run_program: negative size");
  if (K == 0) return GJX_OK;
  if (prog->n_slots > 0 && !choices) return gjx_fail(GJX_EINVAL, "gjx_run_program: choices is null");
  if (lse && !logw) return gjx_fail(GJX_EINVAL, "gjx_run_program: lse needs logw");
  if (prog->rng_mode == GJX_RNG_FLAT) {   // 1023 site numbers per Scan step and for the sites outside Scans
    int plain = 0, local = 0, tag = 0;
    for (int j = 0; j < prog->n_sites; ++j) {
      const int sc = prog->sites[j].scan;
      if (sc == 0) { ++plain; tag = 0; }
      else { if (sc != tag) { tag = sc; local = 0; } ++local; }
      if (plain > GJX_FLAT_MAX_SITES || local > GJX_FLAT_MAX_SITES)
        return gjx_fail(GJX_EUNSUPPORTED, "gjx_run_program: FLAT stream supports at most 1023 sites per Scan step and 1023 outside Scans");
    }
  }
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site& sj = prog->sites[j];
    if (sj.mode != GJX_MODE_OBS_MASK) continue;
    if (sj.kind == GJX_DIRICHLET) return gjx_fail(GJX_EUNSUPPORTED, "gjx_run_program: a dirichlet site cannot be masked per particle");
    if (sj.slot < 0 || sj.obs_off < 0 || sj.obs_off >= prog->n_slots) return gjx_fail(GJX_EINVAL, "gjx_run_program: OBS_MASK needs a value slot and a flag slot");
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* partials = nullptr;
  unsigned* ticket = nullptr;
  if (lse || (workspace && logw)) {  // lse == NULL with a workspace: leave the per-block partials for the consumer
    if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RUN, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_run_program: workspace too small");
    ticket = (unsigned*)workspace;
    partials = (unsigned long long*)((char*)workspace + kWsHeaderBytes);
  }
  const float log_k_total = (float)log((double)K_total);
  bool has_input = false;
  for (int j = 0; j < prog->n_sites; ++j) has_input = has_input || prog->sites[j].mode == GJX_MODE_INPUT;
  if (op.in_rows && op.in_stride <= 0) return gjx_fail(GJX_EINVAL, "gjx_run_program_ex: in_rows needs in_stride");
  const EnginePlan ep = plan_engine(prog, K, particle_offset, site_scores != nullptr, (op.flags & GJX_RUN_LEAVE_TILES) != 0 && !lse);
  info.n_partials = ep.grid; info.engine = ep.engine; info.tiles_offset = 0;
  const bool tiles_requested = (op.flags & GJX_RUN_LEAVE_TILES) != 0;
  const GmmShape& g = ep.g;
  const int ppt = ep.ppt, nblocks = ep.grid;
  if (op.resample) {
    // the resampling search in the prologue of the generated kernel (gjx_run_resample): only where a block IS a quantisation tile
    const gjx_run_resample& rs = *op.resample;
    if (!rs.logw || !rs.tile_S || !rs.tile_E || !(rs.u >= 0.0 && rs.u < 1.0) || (rs.lse_out && (!rs.lse_partials || rs.n_partials <= 0)) || !op.in_rows)
      return gjx_fail(GJX_EINVAL, "gjx_run_program_ex: resample needs logw, tile totals, u in [0, 1), in_rows (and block pairs with lse_out)");
    if (rs.logw == logw) return gjx_fail(GJX_EINVAL, "gjx_run_program_ex: resample.logw must not be this run's logw");
    if (ep.engine != ENGINE_GEN || ep.ppt != 4 || !has_input || K % 1024 != 0 || K > (1 << 20) || particle_offset != 0)
      return gjx_fail(GJX_EUNSUPPORTED, "gjx_run_program_ex: resample needs a generated kernel with 4 particles per lane, INPUT sites, K % 1024 == 0, K <= 2^20");
  }
  if (ep.engine == ENGINE_GEN) {
    GenArgs ga;
    ga.tab = prog->tab_dev; ga.key = key2{key0, key1}; ga.K = K; ga.offset = particle_offset;
    ga.choices = choices; ga.score = score; ga.weight = weight; ga.logw = logw; ga.logw_in = logw_in; ga.sub = sub;
    ga.site_scores = site_scores; ga.partials = partials; ga.ticket = ticket; ga.lse = lse; ga.log_k_total = log_k_total;
    ga.in_rows = op.in_rows; ga.in_stride = op.in_stride; ga.anc = op.in_ancestors; ga.store_inputs = (op.flags & GJX_RUN_STORE_INPUTS) ? 1 : 0;
    ga.tile_S = nullptr; ga.tile_E = nullptr;
    ga.rs_logw = nullptr; ga.rs_S = nullptr; ga.rs_E = nullptr; ga.rs_lse = nullptr; ga.rs_n_partials = 0; ga.rs_lse_out = nullptr; ga.rs_u = 0.0;
    ga.rs_anc_out = nullptr; ga.rs_ctrl = nullptr; ga.st_tag = 0ull; ga.st_rtag = 0ull;
    ga.tl = gjx::debug_timeline(128 * (size_t)nblocks);     // (profiling scripts only: gjx_debug_timeline registers the buffer)
    if (op.resample) {
      const gjx_run_resample& rs = *op.resample;
      ga.rs_logw = rs.logw; ga.rs_S = (const unsigned long long*)rs.tile_S; ga.rs_E = rs.tile_E; ga.rs_lse = rs.lse_partials; ga.rs_n_partials = rs.n_partials;
      ga.rs_lse_out = rs.lse_out; ga.rs_u = rs.u; ga.rs_anc_out = rs.ancestors_out; ga.rs_ctrl = rs.status_ws ? (unsigned*)rs.status_ws + 8 : nullptr;
    }
    if (tiles_requested && !lse && partials && K % 1024 == 0 && ppt == 4) {
      // tile totals of the tile-scaled resampler behind the block pairs: every block of the generated kernel walks whole
      // 1024-particle tiles only when 256 * ppt divides 1024 and its tile loop is tile-aligned (gjx_codegen.hip)
      const size_t off = (kWsHeaderBytes + 8 * (size_t)((K + 255) / 256) + 15) & ~(size_t)15;
      const size_t nt = (size_t)(K / 1024);
      if (off + 12 * nt <= workspace_bytes) {
        ga.tile_S = (unsigned long long*)((char*)workspace + off);
        ga.tile_E = (int32_t*)(ga.tile_S + nt);
        info.tiles_offset = (int64_t)off;
      }
    }
    if (info_out) *info_out = info;
    return gen_launch(prog, ppt, ga, nblocks, st, ev0, ev1);
  }
  if (ep.engine == ENGINE_GMM) {
    GmmArgs a;
    fill_gmm_args(a, prog, g);
    a.key = key2{key0, key1}; a.K = K; a.offset = particle_offset;
    a.choices = choices; a.score = score; a.weight = weight; a.logw = logw;
    a.logw_in = logw_in; a.sub = sub; a.partials = partials; a.ticket = ticket; a.lse = lse; a.log_k_total = log_k_total;
    const bool flat = prog->rng_mode != GJX_RNG_JAX32;
    a.tile_S = nullptr; a.tile_E = nullptr;
    // consumer-finishes mode (lse == NULL) on the 1024-particles-per-block kernel: leave the tile totals of the tile-scaled
    // resampler behind the block partials (which take at most 8 bytes per 256 particles)
    static const bool tiles_env = env_int("GJX_RUN_TILES", 0) != 0;
    const bool want_tiles = tiles_requested || tiles_env;
    if (want_tiles && flat && ppt == 4 && !lse && partials && K % 1024 == 0) {
      const size_t off = (kWsHeaderBytes + 8 * (size_t)((K + 255) / 256) + 15) & ~(size_t)15;
      const size_t nt = (size_t)(K / 1024);
      if (off + 12 * nt <= workspace_bytes) {
        a.tile_S = (unsigned long long*)((char*)workspace + off);
        a.tile_E = (int32_t*)(a.tile_S + nt);
        info.tiles_offset = (int64_t)off;
      }
    }
    const size_t lds = flat ? sizeof(float) * (size_t)gmm_aux_floats(g.C, g.D)
                            : sizeof(float) * (size_t)(3 * g.C * (g.D + 4) + 4 * g.C + 3 * g.D + 32 + 8);
    launch_gmm(a, flat, g.D, ppt, nblocks, lds, st, DispatchEvents{ev0, ev1});
  } else {
    RunArgs a;
    a.sites = prog->sites_dev; a.tab = prog->tab_dev; a.n_sites = prog->n_sites; a.n_slots = prog->n_slots;
    a.key = key2{key0, key1}; a.K = K; a.offset = particle_offset;
    a.choices = choices; a.score = score; a.weight = weight; a.logw = logw;
    a.logw_in = logw_in; a.sub = sub; a.site_scores = site_scores; a.partials = partials; a.ticket = ticket; a.lse = lse;
    a.log_k_total = log_k_total;
    a.preload = 0;
    for (int j = 0; j < prog->n_sites; ++j) a.preload |= prog->sites[j].mode == GJX_MODE_OBS_SLOT || prog->sites[j].mode == GJX_MODE_OBS_MASK;
    a.n_tab = prog->n_tab;
    a.in_rows = op.in_rows; a.in_stride = op.in_stride; a.anc = op.in_ancestors; a.store_inputs = (op.flags & GJX_RUN_STORE_INPUTS) ? 1 : 0;
    const size_t vbytes = sizeof(float) * 256 * (size_t)prog->n_slots;
    const bool ldsv = prog->n_slots > 0 && vbytes <= 48 * 1024 && !env_int("GJX_GENERIC_NO_LDS", 0);
    const size_t tbytes = sizeof(float) * (size_t)prog->n_tab;
    const bool tabl = tbytes <= 16 * 1024 && !env_int("GJX_GENERIC_NO_LDS", 0);
    const size_t lds = (ldsv ? vbytes : 0) + (tabl ? tbytes : 0);
#define GJX_GEN(R, L, T) hipLaunchKernelGGL((k_run_generic<R, L, T>), dim3(nblocks), dim3(256), lds, st, a)
    if (prog->rng_mode == GJX_RNG_JAX32) {
      if (ldsv && tabl) GJX_GEN(GJX_RNG_JAX32, true, true); else if (ldsv) GJX_GEN(GJX_RNG_JAX32, true, false);
      else if (tabl) GJX_GEN(GJX_RNG_JAX32, false, true); else GJX_GEN(GJX_RNG_JAX32, false, false);
    } else {
      if (ldsv && tabl) GJX_GEN(GJX_RNG_FLAT, true, true); else if (ldsv) GJX_GEN(GJX_RNG_FLAT, true, false);
      else if (tabl) GJX_GEN(GJX_RNG_FLAT, false, true); else GJX_GEN(GJX_RNG_FLAT, false, false);
    }
#undef GJX_GEN
  }
  GJX_CHECK_LAUNCH("gjx_run_program");
  (void)nblocks;
  if (info_out) *info_out = info;
  return GJX_OK;
}

// ---- one-launch importance step -------------------------------------------------------------------------------
namespace {
template <int D>
const void* step_kernel() { return (const void*)k_run_gmm_flat<D, 4, 256, true>; }
const void* step_kernel_for(int D) {
  switch (D) {
    case 1: return step_kernel<1>(); case 2: return step_kernel<2>(); case 4: return step_kernel<4>(); case 8: return step_kernel<8>();
    case 16: return step_kernel<16>(); default: return nullptr;
  }
}
template <int D>
void launch_step_d(const GmmArgs& a, int grid, size_t lds, hipStream_t st, DispatchEvents ev) { launch_gmm_kernel(k_run_gmm_flat<D, 4, 256, true>, a, grid, lds, st, ev); }
}  // namespace

extern "C" int gjx_importance_step(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K, int64_t particle_offset,
                                   float* choices, float* score, float* logw, float* lse, double u, float* rows_out,
                                   int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream) {
  return gjx_importance_step_ex(prog, key0, key1, K, particle_offset, choices, score, logw, lse, u, rows_out, ancestors, workspace,
                                workspace_bytes, stream, nullptr, nullptr);
}
extern "C" int gjx_importance_step_ex(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K, int64_t particle_offset,
                                      float* choices, float* score, float* logw, float* lse, double u, float* rows_out,
                                      int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream,
                                      void* start_event, void* stop_event) {
  if (!prog || !prog->sites || !prog->sites_dev || !prog->tab_dev) return gjx_fail(GJX_EINVAL, "gjx_importance_step: null program");
  if (K <= 0 || !choices || !logw || !rows_out || !ancestors || !(u >= 0.0 && u < 1.0))
    return gjx_fail(GJX_EINVAL, "gjx_importance_step: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_RUN, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_importance_step: workspace too small");
  GmmShape g;
  const bool same_hi = ((uint64_t)particle_offset >> 32) == ((uint64_t)(particle_offset + K - 1) >> 32);
  if (env_int("GJX_NO_FUSED_STEP", 0) || prog->rng_mode != GJX_RNG_FLAT || !same_hi || !gmm_usable(prog, &g) || g.D > 16 || K % 1024 != 0)
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_importance_step: no one-launch kernel for this program / size (use gjx_run_program + gjx_resample_indices + gjx_gather_rows)");
  const int64_t nblocks = K / 1024;
  const size_t lds = sizeof(float) * (size_t)gmm_aux_floats(g.C, g.D);
  const int cap = gjx_coresident_blocks(step_kernel_for(g.D), 256, lds);
  if (nblocks > cap || nblocks > 2048 || (uint64_t)(1 + g.D) * (uint64_t)K * 4ull >= (1ull << 32))
    return gjx_fail(GJX_EUNSUPPORTED, "gjx_importance_step: the grid would not be co-resident on this device");
  GmmArgs a = {};
  fill_gmm_args(a, prog, g);
  a.key = key2{key0, key1}; a.K = K; a.offset = particle_offset;
  a.choices = choices; a.score = score; a.weight = nullptr; a.logw = logw;
  a.logw_in = nullptr; a.sub = nullptr; a.partials = nullptr; a.ticket = nullptr; a.lse = lse;
  a.log_k_total = (float)log((double)K);
  a.u = u; a.rows_out = rows_out; a.ancestors = ancestors;
  a.agg = (unsigned long long*)((char*)workspace + kWsHeaderBytes);
  a.ctrl = (unsigned*)workspace + 8;
  a.timeline = gjx::debug_timeline(64 * (size_t)((K + 1023) / 1024));
  hipStream_t st = (hipStream_t)stream;
  const DispatchEvents ev{(hipEvent_t)start_event, (hipEvent_t)stop_event};
  switch (g.D) {
    case 1: launch_step_d<1>(a, (int)nblocks, lds, st, ev); break;
    case 2: launch_step_d<2>(a, (int)nblocks, lds, st, ev); break;
    case 4: launch_step_d<4>(a, (int)nblocks, lds, st, ev); break;
    case 8: launch_step_d<8>(a, (int)nblocks, lds, st, ev); break;
    default: launch_step_d<16>(a, (int)nblocks, lds, st, ev); break;
  }
  GJX_CHECK_LAUNCH("gjx_importance_step");
  return GJX_OK;
}

extern "C" int gjx_logsumexp(const float* x, int64_t K, int64_t K_total, float* out, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!x || !out || K <= 0) return gjx_fail(GJX_EINVAL, "gjx_logsumexp: bad argument");
  if (!workspace || workspace_bytes < gjx_workspace_bytes(GJX_OP_LSE, K)) return gjx_fail(GJX_EWORKSPACE, "gjx_logsumexp: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t want = (K + 1023) / 1024;
  const int nblocks = (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
  float2* parts = (float2*)((char*)workspace + kWsHeaderBytes);
  hipLaunchKernelGGL(k_lse_partial, dim3(nblocks), dim3(256), 0, st, x, K, parts);
  GJX_CHECK_LAUNCH("gjx_logsumexp/partial");
  return gjx_launch_lse_finish(parts, nblocks, K_total, out, st);
}

extern "C" int gjx_lse_combine(const float* pairs, int G, int64_t K_total, float* out, void* stream) {
  if (!pairs || !out || G <= 0) return gjx_fail(GJX_EINVAL, "gjx_lse_combine: bad argument");
  return gjx_launch_lse_finish(pairs, G, K_total, out, (hipStream_t)stream);
}
