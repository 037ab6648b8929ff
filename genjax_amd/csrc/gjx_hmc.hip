// gjx_hmc.hip — per-chain HMC move (HMC.edit, inference/requests/hmc.py:156-211) and the
// selection gradient (hmc.py:70-96) for an arbitrary site list: one chain per lane, chain state in
// the SoA rows of choices[][] (coalesced across lanes), momenta / gradients / the pre-move copy in
// caller-provided workspace rows.  The gradient of the score is a single forward sweep: every
// parameter expression reads value slots directly, so d score / d slot is a sum of per-site terms
// (what jax.grad of gen_fn.assess computes, without a tape).
#include "gjx_device.h"
#include "gjx_host.h"

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

GJX_DEV void dlogpdf(int kind, float x, float a, float b, float& dx, float& da, float& db) {
  dx = da = db = 0.0f;
  switch (kind) {
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: {
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb;
      dx = -z * rb; da = z * rb; db = (z * z - 1.0f) * rb;
      return;
    }
    case GJX_BERNOULLI_LOGITS: da = x - sigmoid(a); return;
    case GJX_FLIP: da = (x != 0.0f ? fast_rcp(a) : 0.0f) - (x != 1.0f ? (1.0f - x) * fast_rcp(1.0f - a) : 0.0f); return;
    case GJX_HALF_NORMAL: { const float ra = fast_rcp(a); const float z = x * ra; dx = -z * ra; da = (z * z - 1.0f) * ra; return; }
    case GJX_EXPONENTIAL: dx = -a; da = fast_rcp(a) - x; return;
    case GJX_LAPLACE: { const float s = (float)((x > a) - (x < a)); const float rb = fast_rcp(b); dx = -s * rb; da = s * rb; db = fabsf(x - a) * rb * rb - rb; return; }
    case GJX_CAUCHY: { const float rb = fast_rcp(b); const float z = (x - a) * rb; const float g = 2.0f * z * fast_rcp(1.0f + z * z); dx = -g * rb; da = g * rb; db = (g * z - 1.0f) * rb; return; }
    case GJX_LOG_NORMAL: { const float lx = fast_log(x); const float rb = fast_rcp(b); const float z = (lx - a) * rb; dx = (-z * rb - 1.0f) * fast_rcp(x); da = z * rb; db = (z * z - 1.0f) * rb; return; }
    case GJX_BETA: dx = (a - 1.0f) * fast_rcp(x) - (b - 1.0f) * fast_rcp(1.0f - x); da = __builtin_nanf(""); db = __builtin_nanf(""); return;
    case GJX_GAMMA: dx = (a - 1.0f) * fast_rcp(x) - b; da = __builtin_nanf(""); db = a * fast_rcp(b) - x; return;
    case GJX_UNIFORM: { const float r = fast_rcp(b - a); da = r; db = -r; return; }
    default: return;
  }
}

GJX_DEV float xf_deriv(int xf, float pre) {
  switch (xf) {
    case GJX_XF_EXP: return fast_exp(pre);
    case GJX_XF_SOFTPLUS: return sigmoid(pre);
    case GJX_XF_SIGMOID: { const float s = sigmoid(pre); return s * (1.0f - s); }
    default: return 1.0f;
  }
}

// score and gradient rows g[n_slots][n] for chain i (rows are zeroed here)
template <class ValFn>
GJX_DEV float score_and_grad(const gjx_site* sites, int n_sites, int n_slots, const float* __restrict__ tab,
                             ValFn&& val, float* g, int64_t n, int64_t i) {
  for (int s = 0; s < n_slots; ++s) g[(int64_t)s * n + i] = 0.0f;
  float score = 0.0f;
  for (int j = 0; j < n_sites; ++j) {
    const gjx_site& s = sites[j];
    const int kind = s.kind;
    if (kind == GJX_CATEGORICAL_LOGITS || kind == GJX_CATEGORICAL_PROBS) {
      const int nc = s.ncat;
      const bool probs = kind == GJX_CATEGORICAL_PROBS;
      float mx = -INFINITY;
      for (int c = 0; c < nc; ++c) { float l = eval_param(s.p[0], c, tab, val); if (probs) l = safe_log(l); mx = fmaxf(mx, l); }
      float se = 0.0f;
      for (int c = 0; c < nc; ++c) { float l = eval_param(s.p[0], c, tab, val); if (probs) l = safe_log(l); se += fast_exp(l - mx); }
      const float v = s.slot >= 0 ? val(s.slot) : tab[s.obs_off];
      int k = (int)v;
      k = k < 0 ? 0 : (k > nc - 1 ? nc - 1 : k);
      float l = eval_param(s.p[0], k, tab, val);
      if (probs) l = safe_log(l);
      score += l - (mx + fast_log(se));
      continue;  // integer site: no gradient through it (hmc.py:49-65)
    }
    for (int d = 0; d < s.dim; ++d) {
      const float pa_pre = eval_param_pre(s.p[0], d, tab, val);
      const float pb_pre = eval_param_pre(s.p[1], d, tab, val);
      const float pa = apply_xf(s.p[0].xf, pa_pre), pb = apply_xf(s.p[1].xf, pb_pre);
      const float x = s.slot >= 0 ? val(s.slot + d) : tab[s.obs_off + d];
      score += elem_logpdf(kind, x, pa, pb);
      float gx, ga, gb;
      dlogpdf(kind, x, pa, pb, gx, ga, gb);
      if (s.slot >= 0) g[(int64_t)(s.slot + d) * n + i] += gx;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const gjx_param& p = s.p[q];
        float gp = q == 0 ? ga : gb;
        if (gp == 0.0f || (p.op != GJX_P_VALUE && p.op != GJX_P_AFFINE)) continue;
        if (p.xf != GJX_XF_NONE) gp *= xf_deriv(p.xf, q == 0 ? pa_pre : pb_pre);
        if (p.op == GJX_P_VALUE) {
          g[(int64_t)(p.slot + (p.len == 1 ? 0 : d % p.len)) * n + i] += gp;
        } else {
          const float* row = tab + p.moff + d * p.n;
          for (int e = 0; e < p.n; ++e) g[(int64_t)(p.slot + e) * n + i] += gp * row[e];
        }
      }
    }
  }
  return score;
}

__global__ __launch_bounds__(256) void k_score_grad(const gjx_site* sites, const float* tab, int n_sites, int n_slots,
                                                   int64_t n, const float* choices, float* score, float* grad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  auto val = [&](int slot) -> float { return choices[(int64_t)slot * n + i]; };
  const float sc = score_and_grad(sites, n_sites, n_slots, tab, val, grad, n, i);
  if (score) score[i] = sc;
  // zero rows of unselected slots (selection_gradient returns zeros for them, hmc.py:90-96)
  for (int j = 0; j < n_sites; ++j) {
    const gjx_site& s = sites[j];
    if (s.slot < 0 || (s.flags & GJX_SITE_HMC_SELECTED)) continue;
    for (int d = 0; d < s.dim; ++d) grad[(int64_t)(s.slot + d) * n + i] = 0.0f;
  }
}

template <int RNG>
__global__ __launch_bounds__(256) void k_hmc_generic(HmcArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int64_t n = a.n;
  float* ch = a.choices;
  auto val = [&](int slot) -> float { return ch[(int64_t)slot * n + i]; };
  const uint64_t gidx = (uint64_t)(a.offset + i);
  for (int s = 0; s < a.n_slots; ++s) a.ws_old[(int64_t)s * n + i] = ch[(int64_t)s * n + i];
  const float score0 = score_and_grad(a.sites, a.n_sites, a.n_slots, a.tab, val, a.ws_g0, n, i);  // hmc.py:165-166
  for (int s = 0; s < a.n_slots; ++s) a.ws_g[(int64_t)s * n + i] = a.ws_g0[(int64_t)s * n + i];
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
      BitStream<RNG> bs;
      if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, (uint32_t)leaf));
      else bs.open(a.key, gidx, (uint32_t)leaf + 1u);
      for (int d = 0; d < s.dim; ++d, ++m) {
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
    const float* gfirst = a.stale ? a.ws_g0 : a.ws_g;  // hmc.py:186 keeps the received gradient in the carry
    int m = 0;
    for (int j = 0; j < a.n_sites; ++j) {
      const gjx_site& s = a.sites[j];
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      for (int d = 0; d < s.dim; ++d, ++m) {
        const int64_t si = (int64_t)(s.slot + d) * n + i;
        const float p = a.ws_p[(int64_t)m * n + i] + he * gfirst[si];
        a.ws_p[(int64_t)m * n + i] = p;
        ch[si] = ch[si] + a.eps * p;
      }
    }
    sc = score_and_grad(a.sites, a.n_sites, a.n_slots, a.tab, val, a.ws_g, n, i);
    m = 0;
    for (int j = 0; j < a.n_sites; ++j) {
      const gjx_site& s = a.sites[j];
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      for (int d = 0; d < s.dim; ++d, ++m) {
        const int64_t si = (int64_t)(s.slot + d) * n + i;
        a.ws_p[(int64_t)m * n + i] += he * a.ws_g[si];
      }
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
  if (!acc) {
    for (int s = 0; s < a.n_slots; ++s) ch[(int64_t)s * n + i] = a.ws_old[(int64_t)s * n + i];
    sc = score0;
  }
  if (a.score) a.score[i] = sc;
  if (a.alpha) a.alpha[i] = al;
  if (a.accepted) a.accepted[i] = acc ? 1.0f : 0.0f;
}

}  // namespace gjx

using namespace gjx;

static int count_selected(const gjx_program* prog) {
  int nsel = 0;
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site& s = prog->sites[j];
    if ((s.flags & GJX_SITE_HMC_SELECTED) && s.slot >= 0) nsel += s.dim;
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
    if (s.mode == GJX_MODE_SAMPLE) return gjx_fail(GJX_EINVAL, "gjx_hmc: every site must be constrained (mode OBS_*)");
    if ((s.flags & GJX_SITE_HMC_SELECTED) && (s.kind == GJX_FLIP || s.kind == GJX_BERNOULLI_LOGITS ||
                                              s.kind == GJX_CATEGORICAL_LOGITS || s.kind == GJX_CATEGORICAL_PROBS))
      return gjx_fail(GJX_EINVAL, "gjx_hmc: only float32 sites can be selected (hmc.py:49-65)");
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
  if (prog->rng_mode == GJX_RNG_JAX32) hipLaunchKernelGGL(k_hmc_generic<GJX_RNG_JAX32>, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_hmc_generic<GJX_RNG_FLAT>, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
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
