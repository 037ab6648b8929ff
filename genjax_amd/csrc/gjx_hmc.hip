// gjx_hmc.hip — per-chain HMC move (HMC.edit, inference/requests/hmc.py:156-211) and the
// selection gradient (hmc.py:70-96) for an arbitrary site list: one chain per lane, chain state in
// the SoA rows of choices[][] (coalesced across lanes), momenta / gradients / the pre-move copy in
// caller-provided workspace rows.  The gradient of the score is a single forward sweep: every
// parameter expression reads value slots directly, so d score / d slot is a sum of per-site terms
// (what jax.grad of gen_fn.assess computes, without a tape).
#include "gjx_device.h"
#include "gjx_host.h"
#include <string.h>

namespace gjx {

struct HmcArgs {
  const gjx_site* sites;
  const float* tab;
  int n_sites, n_slots, nsel;
  key2 key;
  int64_t n, offset;
  float eps;
  int L, stale, accept;
  float* choices;
  float* score;
  float* alpha;
  float* accepted;
  float* ws_p;    // [nsel][n]    momenta
  float* ws_g;    // [n_slots][n] current gradient
  float* ws_g0;   // [n_slots][n] gradient at the initial position
  float* ws_old;  // [n_slots][n] pre-move values
};

// where a chain's gradient row lives: column i of rows g[n_slots][n] in memory, or column threadIdx.x of LDS rows
struct GlobalRows {
  float* g; int64_t n, i;
  GJX_DEV float& at(int slot) const { return g[(int64_t)slot * n + i]; }
};
struct LdsRows {
  float* g;
  GJX_DEV float& at(int slot) const { return g[slot * 256 + threadIdx.x]; }
};

// the contribution of ONE site — a plain site, or instance `inst` of a plate's body site (gjx.h "Plates": rows slot + inst * dim,
// observation obs_off + inst * d_obs, parameters through their instance strides) — to the score and the gradient rows
template <class ValFn, class Rows>
GJX_DEV float site_score_and_grad(const gjx_site& s, int inst, const float* __restrict__ tab, ValFn&& val, Rows G) {
  float score = 0.0f;
  const int kind = s.kind;
  const bool cat = kind == GJX_CATEGORICAL_LOGITS || kind == GJX_CATEGORICAL_PROBS;
  const int slot = s.slot >= 0 ? s.slot + inst * (cat ? 1 : s.dim) : -1;
  const int obs = s.obs_off + inst * s.d_obs;
  if (cat) {
    const int nc = s.ncat;
    const bool probs = kind == GJX_CATEGORICAL_PROBS;
    float mx = -INFINITY;
    for (int c = 0; c < nc; ++c) { float l = eval_param(s.p[0], c, tab, val, inst); if (probs) l = safe_log(l); mx = fmaxf(mx, l); }
    float se = 0.0f;
    for (int c = 0; c < nc; ++c) { float l = eval_param(s.p[0], c, tab, val, inst); if (probs) l = safe_log(l); se += fast_exp(l - mx); }
    const float v = slot >= 0 ? val(slot) : tab[obs];
    int k = (int)v;
    k = k < 0 ? 0 : (k > nc - 1 ? nc - 1 : k);
    float l = eval_param(s.p[0], k, tab, val, inst);
    if (probs) l = safe_log(l);
    return l - (mx + fast_log(se));   // integer site: no gradient through it (hmc.py:49-65)
  }
  if (kind == GJX_DIRICHLET) {  // simplex-valued: scored, never moved (the host refuses to select it)
    float sa = 0.0f;
    for (int d = 0; d < s.dim; ++d) {
      const float al = eval_param(s.p[0], d, tab, val, inst);
      const float x = slot >= 0 ? val(slot + d) : tab[obs + d];
      sa += al;
      score += ((al - 1.0f) == 0.0f ? 0.0f : (al - 1.0f) * fast_log(x)) - lgammaf(al);
    }
    return score + lgammaf(sa);
  }
  for (int d = 0; d < s.dim; ++d) {
    const float pa_pre = eval_param_pre(s.p[0], d, tab, val, inst);
    const float pb_pre = eval_param_pre(s.p[1], d, tab, val, inst);
    const float pa = apply_xf(s.p[0].xf, pa_pre), pb = apply_xf(s.p[1].xf, pb_pre);
    const int np = params_of(kind);
    const float pc_pre = np > 2 ? eval_param_pre(s.p[2], d, tab, val, inst) : 0.0f;
    const float pd_pre = np > 3 ? eval_param_pre(s.p[3], d, tab, val, inst) : 0.0f;
    const float pc = np > 2 ? apply_xf(s.p[2].xf, pc_pre) : 0.0f, pd = np > 3 ? apply_xf(s.p[3].xf, pd_pre) : 0.0f;
    const float x = slot >= 0 ? val(slot + d) : tab[obs + d];
    score += elem_logpdf(kind, x, pa, pb, pc, pd);
    float gx, gpar[4];
    dlogpdf(kind, x, pa, pb, pc, pd, gx, gpar);
    if (slot >= 0) G.at(slot + d) += gx;
    const float pre[4] = {pa_pre, pb_pre, pc_pre, pd_pre};
    for (int q = 0; q < np; ++q) {
      const gjx_param& p = s.p[q];
      float gp = gpar[q];
      if (gp == 0.0f || (p.op != GJX_P_VALUE && p.op != GJX_P_AFFINE && p.op != GJX_P_VGATHER && p.op != GJX_P_EXPR)) continue;
      if (p.xf != GJX_XF_NONE) gp *= xf_deriv(p.xf, pre[q]);
      const int ps = p.slot + inst * p.d_slot;
      if (p.op == GJX_P_EXPR) {          // reverse sweep through the block to its VALUE leaves (hmc.py:70-96: grad through the body)
        expr_backward(p, d, gp, tab, val, inst, G);
      } else if (p.op == GJX_P_VGATHER) {
        G.at(vgather_row(p, d, tab, val, inst)) += gp;
      } else if (p.op == GJX_P_VALUE) {
        G.at(ps + (p.len == 1 ? 0 : d % p.len)) += gp;
      } else {
        const float* row = tab + p.moff + inst * p.d_moff + d * p.n;
        for (int e = 0; e < p.n; ++e) G.at(ps + e) += gp * row[e];
      }
    }
  }
  return score;
}

// score and gradient of chain i; the gradient rows G.at(slot) are zeroed here.  A plate (the gradient of assess through a Vmap:
// hmc.py:70-96 differentiates any assess, vmap.py:363-376) is walked instance by instance
template <class ValFn, class Rows>
GJX_DEV float score_and_grad(const gjx_site* sites, int n_sites, int n_slots, const float* __restrict__ tab,
                             ValFn&& val, Rows G) {
  for (int s = 0; s < n_slots; ++s) G.at(s) = 0.0f;
  float score = 0.0f;
  for (int j = 0; j < n_sites;) {
    const gjx_site& s = sites[j];
    if (s.mode == GJX_MODE_INPUT) { ++j; continue; }      // an argument / a carry: a value that is there, no density
    if (s.plate == 0) { score += site_score_and_grad(s, 0, tab, val, G); ++j; continue; }
    int m = 1;
    while (j + m < n_sites && sites[j + m].plate == s.plate) ++m;
    for (int i = 0; i < s.plate_n; ++i)
      for (int l = 0; l < m; ++l) if (sites[j + l].mode != GJX_MODE_INPUT) score += site_score_and_grad(sites[j + l], i, tab, val, G);
    j += m;
  }
  return score;
}

// rows a site owns in choices[][]: a plate's body site has one set per instance
GJX_DEV int site_rows(const gjx_site& s) {
  const bool cat = (s.kind == GJX_CATEGORICAL_LOGITS || s.kind == GJX_CATEGORICAL_PROBS) && s.mode != GJX_MODE_INPUT;
  return (cat ? 1 : s.dim) * (s.plate ? s.plate_n : 1);
}

__global__ __launch_bounds__(256) void k_score_grad(const gjx_site* sites, const float* tab, int n_sites, int n_slots,
                                                   int64_t n, const float* choices, float* score, float* grad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  auto val = [&](int slot) -> float { return choices[(int64_t)slot * n + i]; };
  const float sc = score_and_grad(sites, n_sites, n_slots, tab, val, GlobalRows{grad, n, i});
  if (score) score[i] = sc;
  // zero rows of unselected slots (selection_gradient returns zeros for them, hmc.py:90-96)
  for (int j = 0; j < n_sites; ++j) {
    const gjx_site& s = sites[j];
    if (s.slot < 0 || (s.flags & GJX_SITE_HMC_SELECTED)) continue;
    const int rows = site_rows(s);
    for (int d = 0; d < rows; ++d) grad[(int64_t)(s.slot + d) * n + i] = 0.0f;
  }
}

// Generic per-chain HMC move over any site list.  LDSM: the chain's values and its gradient live in LDS columns
// ([slot][lane], 2 n_slots KB per block) for the whole trajectory — a site with an affine parameter reads every source
// value and adds to every source gradient once per element, which from memory costs two round trips per (element, source)
// pair (the config-5 model: 2 x 16 K per gradient).  Momenta and the stale-gradient copy stay in the workspace.
template <int RNG, bool LDSM>
__global__ __launch_bounds__(256) void k_hmc_generic(HmcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float hmc_lds[];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;      // no barriers below: every lane works on its own columns
  const int64_t n = a.n;
  float* ch = a.choices;
  float* v_s = hmc_lds;
  float* g_s = hmc_lds + (LDSM ? a.n_slots * 256 : 0);
  auto val = [&](int slot) -> float { return LDSM ? v_s[slot * 256 + threadIdx.x] : ch[(int64_t)slot * n + i]; };
  auto setv = [&](int slot, float x) { if (LDSM) v_s[slot * 256 + threadIdx.x] = x; else ch[(int64_t)slot * n + i] = x; };
  auto grad_of = [&](int slot) -> float { return LDSM ? g_s[slot * 256 + threadIdx.x] : a.ws_g[(int64_t)slot * n + i]; };
  auto gradient = [&]() -> float {
    if (LDSM) return score_and_grad(a.sites, a.n_sites, a.n_slots, a.tab, val, LdsRows{g_s});
    return score_and_grad(a.sites, a.n_sites, a.n_slots, a.tab, val, GlobalRows{a.ws_g, n, i});
  };
  const uint64_t gidx = (uint64_t)(a.offset + i);
  for (int s = 0; s < a.n_slots; ++s) {
    const float x = ch[(int64_t)s * n + i];
    a.ws_old[(int64_t)s * n + i] = x;
    if (LDSM) v_s[s * 256 + threadIdx.x] = x;
  }
  const float score0 = gradient();  // hmc.py:165-166
  if (a.stale) for (int s = 0; s < a.n_slots; ++s) a.ws_g0[(int64_t)s * n + i] = grad_of(s);
  // momenta (hmc.py:120-130): leaf l = l-th selected address in program order
  key2 knew{0u, 0u}, sub{0u, 0u};
  if (RNG == GJX_RNG_JAX32) {
    const key2 ck = fold_in64(a.key, gidx);
    knew = fold_in(ck, 0u);
    sub = fold_in(ck, 1u);  // key, sub_key = split(key)   hmc.py:167
  }
  float k0 = 0.0f;
  {
    int m = 0, leaf = 0;
    for (int j = 0; j < a.n_sites; ++j) {
      const gjx_site& s = a.sites[j];
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      BitStreamRT<RNG> bs;
      if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, (uint32_t)leaf));
      else bs.open(a.key, gidx, (uint32_t)leaf + 1u);
      const int rows = s.plate ? s.dim * s.plate_n : s.dim;     // a selected body site of a plate: one leaf, elements instance-major
      for (int d = 0; d < rows; ++d, ++m) {
        const float p = stream_normal<RNG>(bs, (uint32_t)d);
        a.ws_p[(int64_t)m * n + i] = p;
        k0 += -0.5f * p * p - kHalfLog2Pi;
      }
      ++leaf;
    }
  }
  const float he = 0.5f * a.eps;
  float sc = score0;
  for (int t = 1; t <= a.L; ++t) {  // hmc.py:170-194
    int m = 0;
    for (int j = 0; j < a.n_sites; ++j) {
      const gjx_site& s = a.sites[j];
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      const int rows = s.plate ? s.dim * s.plate_n : s.dim;
      for (int d = 0; d < rows; ++d, ++m) {
        const int sl = s.slot + d;
        // hmc.py:186 keeps the received gradient in the carry (stale): every first half-kick uses it
        const float gf = a.stale ? a.ws_g0[(int64_t)sl * n + i] : grad_of(sl);
        const float p = a.ws_p[(int64_t)m * n + i] + he * gf;
        a.ws_p[(int64_t)m * n + i] = p;
        setv(sl, val(sl) + a.eps * p);
      }
    }
    sc = gradient();
    m = 0;
    for (int j = 0; j < a.n_sites; ++j) {
      const gjx_site& s = a.sites[j];
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      const int rows = s.plate ? s.dim * s.plate_n : s.dim;
      for (int d = 0; d < rows; ++d, ++m) a.ws_p[(int64_t)m * n + i] += he * grad_of(s.slot + d);
    }
  }
  float k1 = 0.0f;
  for (int m = 0; m < a.nsel; ++m) {
    const float q = -1.0f * a.ws_p[(int64_t)m * n + i];
    k1 += -0.5f * q * q - kHalfLog2Pi;
  }
  const float al = sc - score0 + k1 - k0;  // hmc.py:196-203
  bool acc = true;
  if (a.accept) {
    BitStream<RNG> bs;
    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(knew, 0x4d48u));
    else bs.open(a.key, gidx, GJX_FLAT_MAX_SITES);
    const float lu = safe_log(bits_to_unit(bs.get(0u)));
    acc = lu < al;  // tests/inference/test_requests.py:134-137
  }
  if (!acc) sc = score0;
  if (LDSM || !acc) {
    for (int s = 0; s < a.n_slots; ++s) ch[(int64_t)s * n + i] = acc ? val(s) : a.ws_old[(int64_t)s * n + i];
  }
  if (a.score) a.score[i] = sc;
  if (a.alpha) a.alpha[i] = al;
  if (a.accepted) a.accepted[i] = acc ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------------------------------
// fused kernel for hierarchical logistic regression (BASELINE config 5):
//   log_tau ~ N(m0, s0);  beta_p ~ N(mu_p, exp(log_tau)), p < P;  y_n ~ Bernoulli(logits = b_n + X_n . beta), n < N
// The whole chain state (log_tau, beta[P], momenta, gradient) stays in VGPRs for all L leapfrog steps.  A block
// stages X (N*P floats), y and the bias ONCE into LDS; inside the trajectory there is no global-memory traffic.
// Per chain-leapfrog: 2*N*P FMA for X beta and X^T r plus N sigmoids — FP32 VALU bound, ≈0 HBM bytes; the score
// (softplus) is only needed at the two ends of the trajectory.  Random streams and results match
// k_hmc_generic on the same program.
struct LogregArgs {
  const float* tab;
  const float* tab_host;               // host copy of the same table (launcher only)
  int N;
  int x_off, b_off, b_len, y_off;      // X[N][P] row-major, bias, observed y
  int mu_off, mu_len;                  // prior mean of beta
  float m0, s0;                        // prior of log_tau
  key2 key;
  int64_t n, offset;
  float eps;
  int L, stale, accept;
  float* choices;                      // row 0: log_tau, rows 1..P: beta
  float* score;
  float* alpha;
  float* accepted;
};

// Work split: FOUR lanes per chain.  Lane (c, k) handles the observations n ≡ k (mod 4); the four partial
// gradients are summed with two quad butterflies (DPP), after which all four lanes hold the same full
// gradient and take the identical leapfrog update.  That gives 4x the waves of a lane-per-chain layout
// (2^16 chains -> 4096 waves, 4 per SIMD), which is what hides the LDS latency; each ds_read_b128 serves four
// different rows (k = 0..3) broadcast over the 16 chains of the wave — conflict-free.
// LDS image: sX[Npad][P], sY[Npad], sB[Npad]  (Npad = N rounded up to 8; padded rows are zero with y = 0.5,
// bias 0, so their residual y - sigmoid(0) is exactly 0 and they add nothing to the gradient)

// CPL chains per lane-quad member: each X row fetched from LDS feeds CPL independent chains, which halves
// (CPL = 2) the LDS traffic per FLOP and gives the two logit chains of ILP the single quad needs.
template <int P, int CPL>
GJX_DEV void logreg_grad(const LogregArgs& a, const float* __restrict__ sX, const float* __restrict__ sY,
                         const float* __restrict__ sB, int Npad, int k, const float (&lt)[CPL], const float (&beta)[CPL][P],
                         float (&g)[CPL][P], float (&glt)[CPL]) {
  const float* __restrict__ tab = a.tab;
  float gp[CPL][P];
#pragma unroll
  for (int c = 0; c < CPL; ++c)
#pragma unroll
    for (int p = 0; p < P; ++p) gp[c][p] = 0.0f;
  for (int n = k; n < Npad; n += 4) {
    float x0[P];
#pragma unroll
    for (int p = 0; p < P; ++p) x0[p] = sX[n * P + p];
    const float b0 = sB[n], y0 = sY[n];
    float s0[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) s0[c] = b0;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int c = 0; c < CPL; ++c) s0[c] = fmaf(x0[p], beta[c][p], s0[c]);
    float r0[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) r0[c] = y0 - sigmoid(s0[c]);
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int c = 0; c < CPL; ++c) gp[c][p] = fmaf(x0[p], r0[c], gp[c][p]);
  }
  const float rs0 = fast_rcp(a.s0);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const float t2i = fast_exp(-2.0f * lt[c]);  // 1 / tau^2
    float acc = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float z = beta[c][p] - tab[a.mu_off + (a.mu_len == 1 ? 0 : p)];
      g[c][p] = quad_sum(gp[c][p]) - z * t2i;
      acc = fmaf(z * z, t2i, acc);
    }
    glt[c] = -(lt[c] - a.m0) * rs0 * rs0 + acc - (float)P;
  }
}

template <int P>
GJX_DEV float logreg_score(const LogregArgs& a, const float* __restrict__ sX, const float* __restrict__ sY,
                           const float* __restrict__ sB, int k, float lt, const float (&beta)[P]) {
  const float* __restrict__ tab = a.tab;
  float part = 0.0f;
  for (int n = k; n < a.N; n += 4) {
    float sl = sB[n];
#pragma unroll
    for (int p = 0; p < P; ++p) sl = fmaf(sX[n * P + p], beta[p], sl);
    part += elem_logpdf(GJX_BERNOULLI_LOGITS, sY[n], sl, 0.0f);
  }
  float sc = quad_sum(part) + normal_logpdf(lt, a.m0, a.s0);
  const float tau = fast_exp(lt);
#pragma unroll
  for (int p = 0; p < P; ++p) sc += normal_logpdf(beta[p], tab[a.mu_off + (a.mu_len == 1 ? 0 : p)], tau);
  return sc;
}

constexpr int kLogregThreads = 512;  // (128 * CPL) chains per block; X, y, bias (~72 KB at N=1024, P=16) once per block

template <int RNG, int P, bool STALE, int CPL>
__global__ __launch_bounds__(kLogregThreads, CPL == 1 ? 4 : 2) void k_hmc_logreg(LogregArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, Npad = (N + 7) & ~7;
  float* sX = smem;
  float* sY = sX + Npad * P;
  float* sB = sY + Npad;
  for (int t = threadIdx.x; t < Npad * P; t += kLogregThreads) sX[t] = t < N * P ? a.tab[a.x_off + t] : 0.0f;
  for (int t = threadIdx.x; t < Npad; t += kLogregThreads) {
    sY[t] = t < N ? a.tab[a.y_off + t] : 0.5f;
    sB[t] = t < N ? a.tab[a.b_off + (a.b_len == 1 ? 0 : t)] : 0.0f;
  }
  __syncthreads();
  const int k = threadIdx.x & 3;
  const int64_t n = a.n;
  float* ch = a.choices;
  int64_t idx[CPL];
  bool live[CPL];
  float lt[CPL], beta[CPL][P], score0[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    idx[c] = ((int64_t)blockIdx.x * (kLogregThreads / 4) + (threadIdx.x >> 2)) * CPL + c;
    live[c] = idx[c] < n;
    if (!live[c]) idx[c] = n - 1;  // keep the quad complete for the butterflies; no stores from shadow chains
    lt[c] = ch[idx[c]];
#pragma unroll
    for (int p = 0; p < P; ++p) beta[c][p] = ch[(int64_t)(1 + p) * n + idx[c]];
    score0[c] = logreg_score<P>(a, sX, sY, sB, k, lt[c], beta[c]);
  }
  float g[CPL][P], glt[CPL], g0[STALE ? CPL : 1][STALE ? P : 1], glt0[CPL];
  logreg_grad<P, CPL>(a, sX, sY, sB, Npad, k, lt, beta, g, glt);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    glt0[c] = glt[c];
    if (STALE) {
#pragma unroll
      for (int p = 0; p < P; ++p) g0[STALE ? c : 0][STALE ? p : 0] = g[c][p];
    }
  }
  // momenta: leaf 0 = log_tau, leaf 1 = beta (same streams as k_hmc_generic; the 4 lanes of a chain agree)
  float plt[CPL], pb[CPL][P], k0[CPL];
  key2 knew[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const uint64_t gidx = (uint64_t)(a.offset + idx[c]);
    key2 sub{0u, 0u};
    knew[c] = key2{0u, 0u};
    if (RNG == GJX_RNG_JAX32) {
      const key2 ck = fold_in64(a.key, gidx);
      knew[c] = fold_in(ck, 0u);
      sub = fold_in(ck, 1u);
    }
    BitStream<RNG> bs;
    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, 0u)); else bs.open(a.key, gidx, 1u);
    plt[c] = stream_normal<RNG>(bs, 0u);
    k0[c] = -0.5f * plt[c] * plt[c] - kHalfLog2Pi;
    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, 1u)); else bs.open(a.key, gidx, 2u);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      pb[c][p] = stream_normal<RNG>(bs, (uint32_t)p);
      k0[c] += -0.5f * pb[c][p] * pb[c][p] - kHalfLog2Pi;
    }
  }
  const float he = 0.5f * a.eps;
  for (int t = 1; t <= a.L; ++t) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      plt[c] = plt[c] + he * (STALE ? glt0[c] : glt[c]);
      lt[c] = lt[c] + a.eps * plt[c];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        pb[c][p] = pb[c][p] + he * (STALE ? g0[STALE ? c : 0][STALE ? p : 0] : g[c][p]);
        beta[c][p] = beta[c][p] + a.eps * pb[c][p];
      }
    }
    logreg_grad<P, CPL>(a, sX, sY, sB, Npad, k, lt, beta, g, glt);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      plt[c] += he * glt[c];
#pragma unroll
      for (int p = 0; p < P; ++p) pb[c][p] += he * g[c][p];
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const float sc = a.L > 0 ? logreg_score<P>(a, sX, sY, sB, k, lt[c], beta[c]) : score0[c];
    float k1 = -0.5f * plt[c] * plt[c] - kHalfLog2Pi;
#pragma unroll
    for (int p = 0; p < P; ++p) k1 += -0.5f * pb[c][p] * pb[c][p] - kHalfLog2Pi;
    const float al = sc - score0[c] + k1 - k0[c];
    bool acc = true;
    if (a.accept) {
      BitStream<RNG> bs;
      if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(knew[c], 0x4d48u));
      else bs.open(a.key, (uint64_t)(a.offset + idx[c]), GJX_FLAT_MAX_SITES);
      acc = safe_log(bits_to_unit(bs.get(0u))) < al;
    }
    if (live[c] && k == 0) {
      const int64_t i = idx[c];
      if (acc) {  // rejected chains keep the values already in choices[][]
        ch[i] = lt[c];
#pragma unroll
        for (int p = 0; p < P; ++p) ch[(int64_t)(1 + p) * n + i] = beta[c][p];
      }
      if (a.score) a.score[i] = acc ? sc : score0[c];
      if (a.alpha) a.alpha[i] = al;
      if (a.accepted) a.accepted[i] = acc ? 1.0f : 0.0f;
    }
  }
}


// ------------------------------------------------------------------------------------------
// P = 16 on the matrix cores.  The two contractions of a leapfrog step are [chains x 16] x [16 x N] (logits) and
// [chains x N] x [N x 16] (gradient): f32-input MFMA (v_mfma_f32_16x16x4_f32, exact f32, the f32 VALU rate) takes
// them off the vector instructions, which keep only the sigmoids and the leapfrog update.  The f32 MFMA runs on the SIMD's
// own f32 lanes: a v_fma / v_exp issued beside it costs its full time on top of the MFMA's 32 cycles, at any number of
// waves per SIMD, and alternating MFMAs with vector instructions costs a further ~10 % over issuing each kind in groups
// of 8 (profiles/microbench/ub_mfma.hip, profiles/r03_mfma_valu_overlap_microbench.txt) — so the gradient loop issues 8
// forward MFMAs, then the 8 x 3 sigmoid instructions, then 8 backward MFMAs, and keeps every other instruction out.
//
// One wave = 16 chains.  Lane l: chain c = l & 15, group q = l >> 4.  Every per-coefficient quantity (beta, momentum,
// gradient) lives in the MFMA C layout: register i of lane (c, q) holds coefficient p = 4q + i of chain c.  With that
// choice no operand ever needs a transpose:
//   logits tile S[n0 + 4q + r][c] (C layout, r = 0..3) = bias + sum_i MFMA(A = X[n0 + (l&15)][4q + i], B = beta[i])
//       (step i contracts the coefficient set {4k + i : k = 0..3}; A[row][k] <-> X[n0 + row][4k + i])
//   gradient G[4q + i][c] += sum_r MFMA(A = X[n0 + 4k + r][p = l&15] = XT[l&15][n0 + 4q + r], B = resid[r])
// LDS holds X TRANSPOSED, XT[16][Npad + 4] (row pad 4 floats: the b128 reads of 16 rows hit 64 distinct banks; the
// forward b32 reads of 4 row groups are 16 banks apart), plus y and bias (plain and pre-scaled): 78 KB at N = 1024 -> 2 blocks per CU.
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kMfmaThreads = 512;  // 8 waves = 128 chains per block

GJX_DEV float group_sum(float v) {  // over the 4 lanes (q = 0..3) that share a chain
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

struct LogregLds {
  const float* xt;  // [16][ld]
  const float* y;   // [Npad]
  const float* b;   // [Npad]
  const float* bs;  // [Npad]  bias * -log2(e): the gradient pass accumulates -log2(e) * logits, so that
                    //         sigmoid = rcp(1 + exp2(acc)) needs no multiply per element
  int ld, Npad, N;
};
constexpr float kNegLog2e = -1.44269504f;

// gradient of the log-likelihood w.r.t. beta (C layout) for the 16 chains of the wave
GJX_DEV v4f logreg_mfma_grad(const LogregLds& s, int c16, int q, const v4f& beta_in) {
  const v4f beta = beta_in * kNegLog2e;
  v4f g0 = {0.0f, 0.0f, 0.0f, 0.0f}, g1 = g0;
  const float* xrow = s.xt + c16 * s.ld + 4 * q;   // backward A operand: XT[l&15][n0 + 4q + r]
  const float* xcol = s.xt + (4 * q) * s.ld + c16; // forward  A operand: XT[4q + i][n0 + (l&15)]
  for (int n0 = 0; n0 < s.Npad; n0 += 32) {        // two independent tiles per trip: two accumulator chains
    v4f s0 = *reinterpret_cast<const v4f*>(s.bs + n0 + 4 * q);
    v4f s1 = *reinterpret_cast<const v4f*>(s.bs + n0 + 16 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xcol[i * s.ld + n0], beta[i], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xcol[i * s.ld + n0 + 16], beta[i], s1, 0, 0, 0);
    }
    const v4f y0 = *reinterpret_cast<const v4f*>(s.y + n0 + 4 * q);
    const v4f y1 = *reinterpret_cast<const v4f*>(s.y + n0 + 16 + 4 * q);
    const v4f a0 = *reinterpret_cast<const v4f*>(xrow + n0);
    const v4f a1 = *reinterpret_cast<const v4f*>(xrow + n0 + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], y0[r] - fast_rcp(1.0f + __builtin_amdgcn_exp2f(s0[r])), g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], y1[r] - fast_rcp(1.0f + __builtin_amdgcn_exp2f(s1[r])), g1, 0, 0, 0);
    }
  }
  return g0 + g1;
}

// The same for observations that are all 0 or 1 (BIN): with s_n = 2 y_n - 1 the residual y_n - sigmoid(z_n) is
// s_n sigmoid(-s_n z_n), so LDS holds the rows X'_n = s_n X_n (and the bias s_n b_n log2(e)), the forward pass accumulates
// log2(e) s_n z_n, the residual is rcp(1 + exp2(acc)) — no y, no subtraction — and the backward pass contracts X' again.
// LD > 0: the row stride is a compile-time constant (config 5: N = 1024), every LDS address is base + immediate.
// The three phases are fenced (sched_barrier): the machine scheduler would interleave them to hide latencies, which the
// other waves of the SIMD do for free, and the interleaved order is the slow one (see the header of this section).
// ZB: the bias is zero everywhere (an affine parameter without a bias term): the accumulators start from the inline constant
// 0 and two of the eight LDS reads of a trip go away (an LDS read beside MFMAs costs the SIMD 7-10 cycles of issue, same file).
template <int LD, bool ZB>
GJX_DEV v4f logreg_mfma_grad_bin(const LogregLds& s, int c16, int q, const v4f& beta_in) {
  const int ld = LD > 0 ? LD : s.ld;
  const v4f beta = beta_in * (-kNegLog2e);
  v4f g0 = {0.0f, 0.0f, 0.0f, 0.0f}, g1 = g0;
  const float* xrow = s.xt + c16 * ld + 4 * q;
  const float* xcol = s.xt + (4 * q) * ld + c16;
  const float* bsq = s.bs + 4 * q;
  for (int n0 = 0; n0 < s.Npad; n0 += 32) {
    float xa[4], xb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xa[i] = xcol[i * ld + n0]; xb[i] = xcol[i * ld + n0 + 16]; }
    v4f s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
    if (!ZB) {
      s0 = *reinterpret_cast<const v4f*>(bsq + n0);
      s1 = *reinterpret_cast<const v4f*>(bsq + n0 + 16);
    }
    const v4f a0 = *reinterpret_cast<const v4f*>(xrow + n0);
    const v4f a1 = *reinterpret_cast<const v4f*>(xrow + n0 + 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], beta[i], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i], beta[i], s1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    v4f e0, e1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e0[r] = __builtin_amdgcn_exp2f(s0[r]); e1[r] = __builtin_amdgcn_exp2f(s1[r]); }
#pragma unroll
    for (int r = 0; r < 4; ++r) { e0[r] = 1.0f + e0[r]; e1[r] = 1.0f + e1[r]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { e0[r] = fast_rcp(e0[r]); e1[r] = fast_rcp(e1[r]); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], e0[r], g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], e1[r], g1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return g0 + g1;
}

// log-likelihood sum_n log Bernoulli(y_n; logits) of the wave's 16 chains (all 4 lanes of a chain get the total)
template <bool BIN>
GJX_DEV float logreg_mfma_loglik(const LogregLds& s, int c16, int q, const v4f& beta) {
  const float* xcol = s.xt + (4 * q) * s.ld + c16;
  float part = 0.0f;
  for (int n0 = 0; n0 < s.Npad; n0 += 16) {
    v4f sl = *reinterpret_cast<const v4f*>(s.b + n0 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) sl = __builtin_amdgcn_mfma_f32_16x16x4f32(xcol[i * s.ld + n0], beta[i], sl, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + 4 * q + r < s.N)    // BIN: the rows and the bias carry s_n, so sl = s_n z_n and log p = log sigmoid(s_n z_n)
        part += elem_logpdf(GJX_BERNOULLI_LOGITS, BIN ? 1.0f : s.y[n0 + 4 * q + r], sl[r], 0.0f);
  }
  return group_sum(part);
}

// BIN: every observation is 0 or 1 (checked by the launcher); NP > 0: Npad is this compile-time constant and the bias is zero
template <int RNG, bool STALE, bool BIN, int NP>
__global__ __launch_bounds__(kMfmaThreads, 2) void k_hmc_logreg_mfma(LogregArgs a) {
  constexpr int P = 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, Npad = NP > 0 ? NP : (N + 31) & ~31, ld = Npad + 4;
  float* sXT = smem;
  float* sY = sXT + P * ld;
  float* sB = sY + Npad;
  float* sBs = sB + Npad;
  for (int t = threadIdx.x; t < P * ld; t += kMfmaThreads) {
    const int p = t / ld, nn = t - p * ld;
    const float sgn = BIN && nn < N ? 2.0f * a.tab[a.y_off + nn] - 1.0f : 1.0f;
    sXT[t] = nn < N ? sgn * a.tab[a.x_off + nn * P + p] : 0.0f;
  }
  for (int t = threadIdx.x; t < Npad; t += kMfmaThreads) {
    sY[t] = t < N ? a.tab[a.y_off + t] : 0.5f;      // padded rows: residual y - sigmoid(0) is exactly 0 (BIN: their X' row is 0)
    const float sgn = BIN && t < N ? 2.0f * sY[t] - 1.0f : 1.0f;
    sB[t] = t < N ? sgn * a.tab[a.b_off + (a.b_len == 1 ? 0 : t)] : 0.0f;
    sBs[t] = sB[t] * (BIN ? -kNegLog2e : kNegLog2e);
  }
  __syncthreads();
  const LogregLds lds{sXT, sY, sB, sBs, ld, Npad, N};
  const float* __restrict__ tab = a.tab;
  const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4;
  const int64_t n = a.n;
  float* ch = a.choices;
  int64_t idx = ((int64_t)blockIdx.x * (kMfmaThreads / 64) + (threadIdx.x >> 6)) * 16 + c16;
  const bool live = idx < n;
  if (!live) idx = n - 1;                             // shadow chains keep the wave's matrices full; they never store
  float lt = ch[idx];
  v4f beta, mu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    beta[i] = ch[(int64_t)(1 + 4 * q + i) * n + idx];
    mu[i] = tab[a.mu_off + (a.mu_len == 1 ? 0 : 4 * q + i)];
  }
  const float rs0 = fast_rcp(a.s0);
  auto prior_score = [&](float l, const v4f& be) {
    const float tau = fast_exp(l);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += normal_logpdf(be[i], mu[i], tau);
    return normal_logpdf(l, a.m0, a.s0) + group_sum(acc);
  };
  auto full_grad = [&](float l, const v4f& be, v4f& g, float& gl) {
    g = BIN ? logreg_mfma_grad_bin<(NP > 0 ? NP + 4 : 0), (NP > 0)>(lds, c16, q, be) : logreg_mfma_grad(lds, c16, q, be);
    const float t2i = fast_exp(-2.0f * l);            // 1 / tau^2
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float z = be[i] - mu[i];
      g[i] -= z * t2i;
      acc = fmaf(z * z, t2i, acc);
    }
    gl = -(l - a.m0) * rs0 * rs0 + group_sum(acc) - (float)P;
  };
  const float score0 = logreg_mfma_loglik<BIN>(lds, c16, q, beta) + prior_score(lt, beta);
  v4f g, g0;
  float glt, glt0;
  full_grad(lt, beta, g, glt);
  g0 = g; glt0 = glt;
  // momenta: leaf 0 = log_tau, leaf 1 = beta — the streams of k_hmc_generic / k_hmc_logreg; each lane draws its 4
  const uint64_t gidx = (uint64_t)(a.offset + idx);
  key2 sub{0u, 0u}, knew{0u, 0u};
  if (RNG == GJX_RNG_JAX32) {
    const key2 ck = fold_in64(a.key, gidx);
    knew = fold_in(ck, 0u);
    sub = fold_in(ck, 1u);
  }
  BitStream<RNG> bs;
  if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, 0u)); else bs.open(a.key, gidx, 1u);
  float plt = stream_normal<RNG>(bs, 0u);
  if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, 1u)); else bs.open(a.key, gidx, 2u);
  v4f pb;
  float ksum = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pb[i] = stream_normal<RNG>(bs, (uint32_t)(4 * q + i));
    ksum += -0.5f * pb[i] * pb[i] - kHalfLog2Pi;
  }
  const float k0 = -0.5f * plt * plt - kHalfLog2Pi + group_sum(ksum);
  const float he = 0.5f * a.eps;
  for (int t = 1; t <= a.L; ++t) {
    plt += he * (STALE ? glt0 : glt);
    lt += a.eps * plt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pb[i] += he * (STALE ? g0[i] : g[i]);
      beta[i] += a.eps * pb[i];
    }
    full_grad(lt, beta, g, glt);
    plt += he * glt;
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] += he * g[i];
  }
  const float sc = a.L > 0 ? logreg_mfma_loglik<BIN>(lds, c16, q, beta) + prior_score(lt, beta) : score0;
  ksum = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) ksum += -0.5f * pb[i] * pb[i] - kHalfLog2Pi;
  const float k1 = -0.5f * plt * plt - kHalfLog2Pi + group_sum(ksum);
  const float al = sc - score0 + k1 - k0;
  bool acc = true;
  if (a.accept) {
    BitStream<RNG> bs2;
    if (RNG == GJX_RNG_JAX32) bs2.open_site_key(fold_in(knew, 0x4d48u));
    else bs2.open(a.key, gidx, GJX_FLAT_MAX_SITES);
    acc = safe_log(bits_to_unit(bs2.get(0u))) < al;
  }
  if (live) {
    if (acc) {  // rejected chains keep the values already in choices[][]
      if (q == 0) ch[idx] = lt;
#pragma unroll
      for (int i = 0; i < 4; ++i) ch[(int64_t)(1 + 4 * q + i) * n + idx] = beta[i];
    }
    if (q == 0) {
      if (a.score) a.score[idx] = acc ? sc : score0;
      if (a.alpha) a.alpha[idx] = al;
      if (a.accepted) a.accepted[idx] = acc ? 1.0f : 0.0f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same kernel for the shape the benchmark runs (0/1 observations, no bias), two steps further: one wave carries TWO groups
// of 16 chains, so every X operand fetched from LDS feeds two MFMAs, and LDS holds X' twice — row-major XR[N][16] for the
// forward A operand (the 4 coefficients a lane needs are one 16-byte read) and transposed XT[16][N + 4] for the backward one —
// so a trip of 32 observations is 4 ds_read_b128 + 32 MFMAs + 48 sigmoid instructions (before: 12 LDS reads per 32 MFMAs; an
// LDS read costs the SIMD ~9 cycles of issue).  130 KB of LDS: one 512-thread block per CU, two waves per SIMD (as fast per
// wave as four: the SIMD is busy either way).
constexpr int kMfma2Chains = kMfmaThreads / 64 * 32;   // 256 chains per block

// gradient for the wave's two chain groups; NP = Npad (compile-time) or 0
template <int NP>
GJX_DEV void logreg_mfma2_grad(const float* xr, const float* xt, int Npad_rt, int c16, int q, const v4f (&beta_in)[2], v4f (&g)[2]) {
  const int Npad = NP > 0 ? NP : Npad_rt, ld = Npad + 4;
  v4f beta[2] = {beta_in[0] * (-kNegLog2e), beta_in[1] * (-kNegLog2e)};
  v4f ga[2], gb[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) { ga[c] = v4f{0.0f, 0.0f, 0.0f, 0.0f}; gb[c] = ga[c]; }
  const float* xrow = xt + c16 * ld + 4 * q;       // backward A operand: XT[l&15][n0 + 4q + r]
  const float* xfw = xr + c16 * 16 + 4 * q;        // forward  A operand: XR[n0 + (l&15)][4q + i]
  // the operands of the next trip are fetched a phase ahead (X' for the forward MFMAs behind this trip's forward MFMAs, the
  // transposed rows behind its backward MFMAs): with two waves per SIMD nothing else hides the LDS latency
  v4f x0 = *reinterpret_cast<const v4f*>(xfw);
  v4f x1 = *reinterpret_cast<const v4f*>(xfw + 16 * 16);
  v4f a0 = *reinterpret_cast<const v4f*>(xrow);
  v4f a1 = *reinterpret_cast<const v4f*>(xrow + 16);
  for (int n0 = 0; n0 < Npad; n0 += 32) {
    const int nn = n0 + 32 < Npad ? n0 + 32 : 0;    // (the last trip fetches the first tile again: unused)
    v4f s0[2], s1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) { s0[c] = v4f{0.0f, 0.0f, 0.0f, 0.0f}; s1[c] = s0[c]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        s0[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i], beta[c][i], s0[c], 0, 0, 0);
        s1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i], beta[c][i], s1[c], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    x0 = *reinterpret_cast<const v4f*>(xfw + nn * 16);
    x1 = *reinterpret_cast<const v4f*>(xfw + (nn + 16) * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s0[c][r] = __builtin_amdgcn_exp2f(s0[c][r]); s1[c][r] = __builtin_amdgcn_exp2f(s1[c][r]); }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s0[c][r] = 1.0f + s0[c][r]; s1[c][r] = 1.0f + s1[c][r]; }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s0[c][r] = fast_rcp(s0[c][r]); s1[c][r] = fast_rcp(s1[c][r]); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        ga[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], s0[c][r], ga[c], 0, 0, 0);
        gb[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], s1[c][r], gb[c], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    a0 = *reinterpret_cast<const v4f*>(xrow + nn);
    a1 = *reinterpret_cast<const v4f*>(xrow + nn + 16);
  }
  g[0] = ga[0] + gb[0];
  g[1] = ga[1] + gb[1];
}

// log-likelihood of one chain group from the row-major copy (rows carry s_n: log p = log sigmoid(s_n z_n))
GJX_DEV float logreg_mfma2_loglik(const float* xr, int N, int Npad, int c16, int q, const v4f& beta) {
  const float* xfw = xr + c16 * 16 + 4 * q;
  float part = 0.0f;
  for (int n0 = 0; n0 < Npad; n0 += 16) {
    const v4f x0 = *reinterpret_cast<const v4f*>(xfw + n0 * 16);
    v4f sl = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) sl = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i], beta[i], sl, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + 4 * q + r < N) part += elem_logpdf(GJX_BERNOULLI_LOGITS, 1.0f, sl[r], 0.0f);
  }
  return group_sum(part);
}

template <int RNG, bool STALE, int NP>
__global__ __launch_bounds__(kMfmaThreads, 1) void k_hmc_logreg_mfma2(LogregArgs a) {
  constexpr int P = 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = a.N, Npad = NP > 0 ? NP : (N + 31) & ~31, ld = Npad + 4;
  float* sXT = smem;               // [16][ld]   X' transposed
  float* sXR = sXT + P * ld;       // [Npad][16] X' row-major
  for (int t = threadIdx.x; t < P * ld; t += kMfmaThreads) {
    const int p = t / ld, nn = t - p * ld;
    sXT[t] = nn < N ? (2.0f * a.tab[a.y_off + nn] - 1.0f) * a.tab[a.x_off + nn * P + p] : 0.0f;
  }
  for (int t = threadIdx.x; t < Npad * P; t += kMfmaThreads) {
    const int nn = t / P;
    sXR[t] = nn < N ? (2.0f * a.tab[a.y_off + nn] - 1.0f) * a.tab[a.x_off + t] : 0.0f;
  }
  __syncthreads();
  const float* __restrict__ tab = a.tab;
  const int lane = threadIdx.x & 63, c16 = lane & 15, q = lane >> 4;
  const int64_t n = a.n;
  float* ch = a.choices;
  int64_t idx[2];
  bool live[2];
  float lt[2];
  v4f beta[2], mu;
#pragma unroll
  for (int i = 0; i < 4; ++i) mu[i] = tab[a.mu_off + (a.mu_len == 1 ? 0 : 4 * q + i)];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    idx[c] = ((int64_t)blockIdx.x * (kMfmaThreads / 64) + (threadIdx.x >> 6)) * 32 + c * 16 + c16;
    live[c] = idx[c] < n;
    if (!live[c]) idx[c] = n - 1;                   // shadow chains keep the wave's matrices full; they never store
    lt[c] = ch[idx[c]];
#pragma unroll
    for (int i = 0; i < 4; ++i) beta[c][i] = ch[(int64_t)(1 + 4 * q + i) * n + idx[c]];
  }
  const float rs0 = fast_rcp(a.s0);
  auto prior_score = [&](float l, const v4f& be) {
    const float tau = fast_exp(l);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += normal_logpdf(be[i], mu[i], tau);
    return normal_logpdf(l, a.m0, a.s0) + group_sum(acc);
  };
  v4f g[2], g0[2];
  float glt[2], glt0[2];
  auto full_grad = [&]() {
    logreg_mfma2_grad<NP>(sXR, sXT, Npad, c16, q, beta, g);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float t2i = fast_exp(-2.0f * lt[c]);      // 1 / tau^2
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float z = beta[c][i] - mu[i];
        g[c][i] -= z * t2i;
        acc = fmaf(z * z, t2i, acc);
      }
      glt[c] = -(lt[c] - a.m0) * rs0 * rs0 + group_sum(acc) - (float)P;
    }
  };
  float score0[2], k0[2], plt[2];
  v4f pb[2];
  key2 knew[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) score0[c] = logreg_mfma2_loglik(sXR, N, Npad, c16, q, beta[c]) + prior_score(lt[c], beta[c]);
  full_grad();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    g0[c] = g[c]; glt0[c] = glt[c];
    // momenta: leaf 0 = log_tau, leaf 1 = beta — the streams of k_hmc_generic / k_hmc_logreg; each lane draws its 4
    const uint64_t gidx = (uint64_t)(a.offset + idx[c]);
    key2 sub{0u, 0u};
    knew[c] = key2{0u, 0u};
    if (RNG == GJX_RNG_JAX32) {
      const key2 ck = fold_in64(a.key, gidx);
      knew[c] = fold_in(ck, 0u);
      sub = fold_in(ck, 1u);
    }
    BitStream<RNG> bs;
    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, 0u)); else bs.open(a.key, gidx, 1u);
    plt[c] = stream_normal<RNG>(bs, 0u);
    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, 1u)); else bs.open(a.key, gidx, 2u);
    float ksum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pb[c][i] = stream_normal<RNG>(bs, (uint32_t)(4 * q + i));
      ksum += -0.5f * pb[c][i] * pb[c][i] - kHalfLog2Pi;
    }
    k0[c] = -0.5f * plt[c] * plt[c] - kHalfLog2Pi + group_sum(ksum);
  }
  const float he = 0.5f * a.eps;
  for (int t = 1; t <= a.L; ++t) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      plt[c] += he * (STALE ? glt0[c] : glt[c]);
      lt[c] += a.eps * plt[c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pb[c][i] += he * (STALE ? g0[c][i] : g[c][i]);
        beta[c][i] += a.eps * pb[c][i];
      }
    }
    full_grad();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      plt[c] += he * glt[c];
#pragma unroll
      for (int i = 0; i < 4; ++i) pb[c][i] += he * g[c][i];
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float sc = a.L > 0 ? logreg_mfma2_loglik(sXR, N, Npad, c16, q, beta[c]) + prior_score(lt[c], beta[c]) : score0[c];
    float ksum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ksum += -0.5f * pb[c][i] * pb[c][i] - kHalfLog2Pi;
    const float k1 = -0.5f * plt[c] * plt[c] - kHalfLog2Pi + group_sum(ksum);
    const float al = sc - score0[c] + k1 - k0[c];
    bool acc = true;
    if (a.accept) {
      BitStream<RNG> bs2;
      if (RNG == GJX_RNG_JAX32) bs2.open_site_key(fold_in(knew[c], 0x4d48u));
      else bs2.open(a.key, (uint64_t)(a.offset + idx[c]), GJX_FLAT_MAX_SITES);
      acc = safe_log(bits_to_unit(bs2.get(0u))) < al;
    }
    if (live[c]) {
      if (acc) {  // rejected chains keep the values already in choices[][]
        if (q == 0) ch[idx[c]] = lt[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) ch[(int64_t)(1 + 4 * q + i) * n + idx[c]] = beta[c][i];
      }
      if (q == 0) {
        if (a.score) a.score[idx[c]] = acc ? sc : score0[c];
        if (a.alpha) a.alpha[idx[c]] = al;
        if (a.accepted) a.accepted[idx[c]] = acc ? 1.0f : 0.0f;
      }
    }
  }
}

}  // namespace gjx

using namespace gjx;

// [normal(CONST,CONST) dim 1, selected] -> [normal(CONST, exp(VALUE slot 0)) dim P, selected]
//   -> [bernoulli_logits(AFFINE X[N][P] of slots 1..P) dim N, OBS_TAB]
static bool match_logreg(const gjx_program* p, LogregArgs* a, int* P_out) {
  if (p->n_sites != 3 || !p->tab) return false;
  for (int j = 0; j < 3; ++j) if (p->sites[j].plate != 0) return false;
  const gjx_site& s0 = p->sites[0];
  const gjx_site& s1 = p->sites[1];
  const gjx_site& s2 = p->sites[2];
  if (s0.kind != GJX_NORMAL || s0.dim != 1 || s0.slot != 0 || s0.mode != GJX_MODE_OBS_SLOT || !(s0.flags & GJX_SITE_HMC_SELECTED)) return false;
  if (s0.p[0].op != GJX_P_CONST || s0.p[1].op != GJX_P_CONST || s0.p[0].xf || s0.p[1].xf) return false;
  if ((s1.kind != GJX_NORMAL && s1.kind != GJX_MVNORMAL_DIAG) || s1.slot != 1 || s1.mode != GJX_MODE_OBS_SLOT || !(s1.flags & GJX_SITE_HMC_SELECTED)) return false;
  const int P = s1.dim;
  if (s1.p[0].op != GJX_P_CONST || s1.p[0].xf || (s1.p[0].len != 1 && s1.p[0].len != P)) return false;
  if (s1.p[1].op != GJX_P_VALUE || s1.p[1].xf != GJX_XF_EXP || s1.p[1].slot != 0 || s1.p[1].len != 1) return false;
  if (s2.kind != GJX_BERNOULLI_LOGITS || s2.mode != GJX_MODE_OBS_TAB || s2.slot >= 0) return false;
  const gjx_param& q = s2.p[0];
  if (q.op != GJX_P_AFFINE || q.xf || q.slot != 1 || q.n != P || (q.len != 1 && q.len != s2.dim)) return false;
  if (p->n_slots != 1 + P) return false;
  if (!(P == 2 || P == 4 || P == 8 || P == 16 || P == 32)) return false;
  if (sizeof(float) * ((size_t)((s2.dim + 7) & ~7) * (P + 2)) > 160 * 1024) return false;  // X, y, bias must fit the 160 KB LDS
  a->N = s2.dim; a->x_off = q.moff; a->b_off = q.off; a->b_len = q.len; a->y_off = s2.obs_off;
  a->mu_off = s1.p[0].off; a->mu_len = s1.p[0].len;
  a->m0 = p->tab[s0.p[0].off]; a->s0 = p->tab[s0.p[1].off];
  *P_out = P;
  return true;
}

// 3 = matrix-core kernel (P == 16), 2 = vector kernel
static int logreg_engine(const LogregArgs& a, int P) {
  const char* e = getenv("GJX_HMC_MFMA");
  if (e && !atoi(e)) return 2;
  const int Npad = (a.N + 31) & ~31;
  if (P == 16 && sizeof(float) * ((size_t)16 * (Npad + 4) + 3 * (size_t)Npad) <= 160 * 1024) return 3;
  return 2;
}

template <int RNG>
static int launch_logreg_mfma(const LogregArgs& a, hipStream_t st) {
  const int Npad = (a.N + 31) & ~31;
  const size_t lds = sizeof(float) * ((size_t)16 * (Npad + 4) + 3 * (size_t)Npad);
  const int chains_per_block = kMfmaThreads / 64 * 16;
  const unsigned nb = (unsigned)((a.n + chains_per_block - 1) / chains_per_block);
  // observations all 0 or 1 (what a Bernoulli site holds): the sign-folded gradient loop; anything else keeps y - sigmoid
  bool bin = !getenv("GJX_HMC_NO_BIN");
  for (int i = 0; i < a.N && bin; ++i) bin = a.tab_host[a.y_off + i] == 0.0f || a.tab_host[a.y_off + i] == 1.0f;
#define GJX_LM(ST, BN, NP)                                                                                           \
  {                                                                                                                  \
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_hmc_logreg_mfma<RNG, ST, BN, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_hmc_logreg_mfma<RNG, ST, BN, NP>), dim3(nb), dim3(kMfmaThreads), lds, st, a);              \
  }
  bool zero_bias = true;
  for (int i = 0; i < (a.b_len == 1 ? 1 : a.N) && zero_bias; ++i) zero_bias = a.tab_host[a.b_off + i] == 0.0f;
  // 0/1 observations, no bias, both copies of X' fit the LDS: the two-group kernel
  const size_t lds2 = sizeof(float) * ((size_t)16 * (Npad + 4) + (size_t)16 * Npad);
  if (bin && zero_bias && lds2 <= 160 * 1024 && !getenv("GJX_HMC_NO_MFMA2")) {
    const unsigned nb2 = (unsigned)((a.n + kMfma2Chains - 1) / kMfma2Chains);
#define GJX_LM3(ST, NP)                                                                                              \
  {                                                                                                                  \
    (void)hipFuncSetAttribute((const void*)k_hmc_logreg_mfma2<RNG, ST, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
    hipLaunchKernelGGL((k_hmc_logreg_mfma2<RNG, ST, NP>), dim3(nb2), dim3(kMfmaThreads), lds2, st, a);               \
  }
    if (a.stale) { if (Npad == 1024) GJX_LM3(true, 1024) else GJX_LM3(true, 0) }
    else { if (Npad == 1024) GJX_LM3(false, 1024) else GJX_LM3(false, 0) }
#undef GJX_LM3
    return 0;
  }
#define GJX_LM2(ST) { if (!bin) GJX_LM(ST, false, 0) else if (Npad == 1024 && zero_bias) GJX_LM(ST, true, 1024) else GJX_LM(ST, true, 0) }
  if (a.stale) GJX_LM2(true) else GJX_LM2(false)
#undef GJX_LM2
#undef GJX_LM
  return 0;
}

template <int RNG>
static int launch_logreg(const LogregArgs& a, int P, hipStream_t st) {
  if (logreg_engine(a, P) == 3) return launch_logreg_mfma<RNG>(a, st);
  // GJX_HMC_CPL=2 puts two chains on each lane (half the LDS traffic per FLOP); measured equal to 1 at 2^16 chains
  // (the kernel is bound by v_fma issue at ~2.9 cycles with three distinct VGPR sources, not by LDS), so 1 is the default
  const char* e = getenv("GJX_HMC_CPL");
  const int cpl = e ? atoi(e) : 1;
  const int chains_per_block = kLogregThreads / 4 * (cpl == 2 ? 2 : 1);
  const unsigned nb = (unsigned)((a.n + chains_per_block - 1) / chains_per_block);
  const int Npad = (a.N + 7) & ~7;
  const size_t lds = sizeof(float) * ((size_t)Npad * P + 2 * (size_t)Npad);
#define GJX_LR3(PP, ST, CP)                                                                                      \
  {                                                                                                              \
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_hmc_logreg<RNG, PP, ST, CP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_hmc_logreg<RNG, PP, ST, CP>), dim3(nb), dim3(kLogregThreads), lds, st, a);             \
  }
#define GJX_LR2(PP, ST) { if (cpl == 2) GJX_LR3(PP, ST, 2) else GJX_LR3(PP, ST, 1) }
#define GJX_LR(PP) case PP: if (a.stale) GJX_LR2(PP, true) else GJX_LR2(PP, false) break;
  switch (P) {
    GJX_LR(2) GJX_LR(4) GJX_LR(8) GJX_LR(16) GJX_LR(32)
    default: return -1;
  }
#undef GJX_LR
#undef GJX_LR2
#undef GJX_LR3
  return 0;
}

// which engines may run: GJX_HMC_ENGINE = auto (default) | fused (hand-written kernels only) | gen (generated kernel, skipping the
// hand-written match) | interp (site interpreter); GJX_FORCE_GENERIC=1 == interp
static int hmc_engine_pref() {
  const char* f = getenv("GJX_FORCE_GENERIC");
  if (f && atoi(f)) return 3;
  const char* e = getenv("GJX_HMC_ENGINE");
  if (!e || !strcmp(e, "auto")) return 0;
  if (!strcmp(e, "fused")) return 1;
  if (!strcmp(e, "gen")) return 2;
  if (!strcmp(e, "interp")) return 3;
  return 0;
}

extern "C" int gjx_hmc_engine(const gjx_program* prog) {
  if (!prog || !prog->sites) return GJX_EINVAL;
  LogregArgs a; int P;
  const int pref = hmc_engine_pref();
  if ((pref == 0 || pref == 1) && match_logreg(prog, &a, &P)) return logreg_engine(a, P);
  if ((pref == 0 || pref == 2) && hmc_gen_available(prog) == GJX_OK) return 4;
  return 0;
}

static int count_selected(const gjx_program* prog) {
  int nsel = 0;
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site& s = prog->sites[j];
    if ((s.flags & GJX_SITE_HMC_SELECTED) && s.slot >= 0) nsel += s.dim * (s.plate ? s.plate_n : 1);   // a plate's body site: every instance
  }
  return nsel;
}

extern "C" size_t gjx_hmc_workspace_bytes(const gjx_program* prog, int64_t n) {
  if (!prog || n < 0) return 0;
  return sizeof(float) * (size_t)n * (size_t)(count_selected(prog) + 3 * prog->n_slots) + 256;
}

extern "C" int gjx_hmc(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t n, int64_t chain_offset,
                       float eps, int32_t L, int32_t stale_grad_compat, int32_t accept, float* choices, float* score,
                       float* alpha, float* accepted, void* workspace, size_t workspace_bytes, void* stream) {
  if (!prog || !prog->sites || !prog->sites_dev || !prog->tab_dev || !choices || n <= 0 || L < 0)
    return gjx_fail(GJX_EINVAL, "gjx_hmc: bad argument");
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site& s = prog->sites[j];
    if (s.mode == GJX_MODE_SAMPLE || s.mode == GJX_MODE_OBS_MASK) return gjx_fail(GJX_EINVAL, "gjx_hmc: every site must be constrained (mode OBS_TAB / OBS_SLOT) or an input (GJX_MODE_INPUT)");
    // (an INPUT site — a kernel's argument, a Scan step's carry — is a per-chain value the other sites read: rows of choices[][],
    // no density, never moved)
    if (s.mode == GJX_MODE_INPUT && (s.flags & GJX_SITE_HMC_SELECTED)) return gjx_fail(GJX_EINVAL, "gjx_hmc: an INPUT site cannot be selected");
    if ((s.flags & GJX_SITE_HMC_SELECTED) && (s.kind == GJX_FLIP || s.kind == GJX_BERNOULLI_LOGITS ||
                                              s.kind == GJX_CATEGORICAL_LOGITS || s.kind == GJX_CATEGORICAL_PROBS ||
                                              s.kind == GJX_POISSON || s.kind == GJX_GEOMETRIC || s.kind == GJX_DIRICHLET || s.kind == GJX_NEGATIVE_BINOMIAL))
      return gjx_fail(GJX_EINVAL, "gjx_hmc: only unconstrained float32 sites can be selected (hmc.py:49-65)");
  }
  const int pref = hmc_engine_pref();
  {
    LogregArgs la; int P;
    if ((pref == 0 || pref == 1) && match_logreg(prog, &la, &P)) {
      la.tab = prog->tab_dev; la.tab_host = prog->tab; la.key = key2{key0, key1}; la.n = n; la.offset = chain_offset; la.eps = eps; la.L = L;
      la.stale = stale_grad_compat; la.accept = accept; la.choices = choices; la.score = score; la.alpha = alpha; la.accepted = accepted;
      if (prog->rng_mode == GJX_RNG_JAX32) launch_logreg<GJX_RNG_JAX32>(la, P, (hipStream_t)stream);
      else launch_logreg<GJX_RNG_FLAT>(la, P, (hipStream_t)stream);
      GJX_CHECK_LAUNCH("gjx_hmc/logreg");
      return GJX_OK;
    }
  }
  // a kernel generated from the site list (gjx_codegen.hip): chain state in registers, table in LDS; the workspace only for the rows of
  // selected sites inside plates
  if ((pref == 0 || pref == 2) && hmc_gen_available(prog) == GJX_OK) {
    HmcGenArgs ga;
    ga.tab = prog->tab_dev; ga.key = key2{key0, key1}; ga.n = n; ga.offset = chain_offset; ga.eps = eps; ga.L = L;
    ga.stale = stale_grad_compat; ga.accept = accept; ga.choices = choices; ga.score = score; ga.alpha = alpha; ga.accepted = accepted;
    ga.ws = workspace ? (float*)((char*)workspace + 256) : nullptr;          // (used only by programs with selected sites inside plates)
    ga.ws_floats = workspace && workspace_bytes > 256 ? (int64_t)((workspace_bytes - 256) / sizeof(float)) : 0;
    const int rc_gen = hmc_gen_launch(prog, ga, (hipStream_t)stream);
    if (rc_gen != GJX_EUNSUPPORTED) return rc_gen;      // (a mode this program's generated kernel does not carry: the site interpreter below)
  }
  if (!workspace || workspace_bytes < gjx_hmc_workspace_bytes(prog, n)) return gjx_fail(GJX_EWORKSPACE, "gjx_hmc: workspace too small");
  HmcArgs a;
  a.sites = prog->sites_dev; a.tab = prog->tab_dev; a.n_sites = prog->n_sites; a.n_slots = prog->n_slots;
  a.nsel = count_selected(prog);
  a.key = key2{key0, key1}; a.n = n; a.offset = chain_offset; a.eps = eps; a.L = L; a.stale = stale_grad_compat; a.accept = accept;
  a.choices = choices; a.score = score; a.alpha = alpha; a.accepted = accepted;
  float* w = (float*)workspace;
  a.ws_p = w; w += (size_t)a.nsel * n;
  a.ws_g = w; w += (size_t)a.n_slots * n;
  a.ws_g0 = w; w += (size_t)a.n_slots * n;
  a.ws_old = w;
  const unsigned nb = (unsigned)((n + 255) / 256);
  // values + gradient in LDS columns when they fit (2 n_slots KB per block, at most 64 KB)
  const size_t lds = (size_t)2 * a.n_slots * 256 * sizeof(float);
  const bool ldsm = lds <= 64 * 1024 && !getenv("GJX_HMC_NO_LDS");
  const bool jax = prog->rng_mode == GJX_RNG_JAX32;
  if (ldsm) {
    if (jax) hipLaunchKernelGGL((k_hmc_generic<GJX_RNG_JAX32, true>), dim3(nb), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_hmc_generic<GJX_RNG_FLAT, true>), dim3(nb), dim3(256), lds, (hipStream_t)stream, a);
  } else {
    if (jax) hipLaunchKernelGGL((k_hmc_generic<GJX_RNG_JAX32, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_hmc_generic<GJX_RNG_FLAT, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
  }
  GJX_CHECK_LAUNCH("gjx_hmc");
  return GJX_OK;
}

extern "C" int gjx_score_grad(const gjx_program* prog, int64_t n, const float* choices, float* score, float* grad,
                              void* stream) {
  if (!prog || !prog->sites_dev || !prog->tab_dev || !choices || !grad || n <= 0) return gjx_fail(GJX_EINVAL, "gjx_score_grad: bad argument");
  hipLaunchKernelGGL(k_score_grad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prog->sites_dev,
                     prog->tab_dev, prog->n_sites, prog->n_slots, n, choices, score, grad);
  GJX_CHECK_LAUNCH("gjx_score_grad");
  return GJX_OK;
}
