/* gjx_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (genjax_amd/) never imports, links or falls back to it.
 *
 * What it restates (paths relative to /root/reference/src/genjax/_src/):
 *   - ImportanceK.run_smc's vmapped body           inference/smc.py:298-315
 *   - Target.importance                            inference/sp.py:83-87
 *   - GenerateHandler key rule / score / weight    generative_functions/static.py:340-399, 102-105
 *   - Distribution.generate_choice_map             generative_functions/distributions/distribution.py:117-147
 *   - ExactDensity.random_weighted/estimate_logpdf distribution.py:371-396 (shaped logpdf is summed)
 *   - log-ML estimate, sample_particle             inference/smc.py:96-109
 *   - ChangeTarget reweight                        inference/smc.py:378-391
 *   - HMC.edit                                     inference/requests/hmc.py:70-211
 * The arithmetic of the reference lives in un-vendored dependencies that cannot be imported in
 * the build container: jax 0.5.2 / jaxlib 0.5.1 (poetry.lock:1627-1661) and
 * tensorflow-probability 0.23.0 (poetry.lock:5015-5016).  Their published algorithms are
 * restated here: Threefry-2x32 (Salmon et al. 2011, 20 rounds; checked against the Random123
 * known-answer vectors), JAX's key derivation with jax_threefry_partitionable=True
 * (split(k,n)[i] == fold_in(k,i) == Threefry(k,(0,i))), bits -> uniform -> normal via
 * sqrt(2)*erfinv (Giles 2010 single-precision polynomial, as XLA), Gumbel-max categorical, and
 * TFP's closed-form log_prob expressions.
 *
 * PARITY STATUS: bit-level parity with the reference's sample streams is UNPINNED — the
 * reference holds no golden vectors or sampled-value assertions for this path
 * and cannot run here.  This oracle is pinned by: the Random123 KATs, scipy.stats log-pdf
 * tables (tests/golden/), the one literal value the reference's tests hold on this path
 * (tests/generative_functions/test_static_gen_fn.py:318, assess == -2.837877; tests/golden/reference_kat.json),
 * and every closed-form / tolerance check the reference's own tests hold for the path
 * (tests/inference/test_smc.py:32-87, test_requests.py:94-255, test_static_gen_fn.py:208-731, README.md:89-123).
 *
 * Plain C.  Samplers, parameter expressions and the per-particle sums run in float32 like the reference; the closed-form
 * log-densities (elem_logpdf*) and the reductions the reference leaves to XLA (logsumexp) are evaluated in double and
 * rounded once.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gjx.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------
 * Threefry-2x32, 20 rounds (Random123 threefry2x32_R(20,...)); JAX: jax/_src/prng.py
 * threefry2x32 — rotations {13,15,26,6} {17,29,16,24}, key schedule parity 0x1BD11BDA.
 * ------------------------------------------------------------------------------------- */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

void gjxo_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t out[2]) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int g = 0; g < 5; ++g) {
    const int* r = (g & 1) ? R + 4 : R;
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, r[j]);
      x1 ^= x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0;
  out[1] = x1;
}

typedef struct { uint32_t a, b; } okey;

/* ---- decision margins ---------------------------------------------------------------------------------
 * Every place where a float comparison decides a DISCRETE outcome (which category, accept / reject, which side of a
 * floor) records how close the two sides were, relative to their size.  gjxo_run_program keeps the smallest margin of
 * each particle (gjxo_set_margin_buffer).  The device uses hardware exp / log / rcp (about 1 ulp), so a device
 * particle may legitimately take the other branch ONLY where the margin is tiny: the parity tests require every
 * particle that differs from the oracle to have a margin below their stated bound, instead of allowing a blanket
 * fraction of mismatches. */
static _Thread_local float t_margin = 3.0e38f;
static float* g_margin_buf = 0;
static int64_t g_margin_n = 0;
void gjxo_set_margin_buffer(float* buf, int64_t n) { g_margin_buf = buf; g_margin_n = n; }
/* optional [n_selected_scalars][n] buffer that gjxo_hmc fills with the initial momenta it drew (tests integrate the same
 * trajectory in float64 from them to size the float32 tolerance) */
static float* g_momenta_buf = 0;
static int64_t g_momenta_n = 0;
void gjxo_set_momenta_buffer(float* buf, int64_t n) { g_momenta_buf = buf; g_momenta_n = n; }
static inline void decide(float lhs, float rhs) {
  float m = fabsf(lhs - rhs), sc = fabsf(lhs) > fabsf(rhs) ? fabsf(lhs) : fabsf(rhs);
  if (sc > 1.0f) m /= sc;
  if (!(m >= 0.0f)) m = 0.0f; /* NaN on either side: anything goes */
  if (m < t_margin) t_margin = m;
}

/* jax.random.fold_in(key, i) == jax.random.split(key, n)[i]  (partitionable threefry) */
static inline okey fold_in64(okey k, uint64_t i) {
  uint32_t o[2];
  gjxo_threefry2x32(k.a, k.b, (uint32_t)(i >> 32), (uint32_t)i, o);
  okey r = {o[0], o[1]};
  return r;
}
static inline okey fold_in(okey k, uint32_t i) { return fold_in64(k, i); }

/* A random stream = all draws of (particle idx, site) under one run key; see GJX_RNG_* in gjx.h. */
typedef struct {
  int mode;
  okey key;      /* FLAT: run key (with the high index word folded in when idx >= 2^32) */
  uint32_t c0;   /* FLAT: low 32 bits of the global particle index */
  uint32_t site; /* FLAT: 1-based site index */
  okey sk;       /* JAX32: site key */
} ostream;

static inline ostream stream_open(int mode, okey run_key, uint64_t idx, uint32_t site) {
  ostream s;
  s.mode = mode; s.key = run_key; s.c0 = (uint32_t)idx; s.site = site; s.sk = run_key;
  if (mode == GJX_RNG_JAX32) {
    okey pk = fold_in64(run_key, idx); /* jax.random.split(key, K)[idx]   smc.py:300 */
    s.sk = fold_in(pk, site);          /* fold_in(key, counter)           static.py:349-352 */
  } else if (idx >> 32) {
    uint32_t o[2];
    gjxo_threefry2x32(run_key.a, run_key.b, 0xFFFFFFFFu, (uint32_t)(idx >> 32), o);
    s.key.a = o[0]; s.key.b = o[1];
  }
  return s;
}
static inline ostream stream_from_site_key(okey sk) { /* JAX32 stream with an explicit site key */
  ostream s;
  s.mode = GJX_RNG_JAX32; s.key = sk; s.c0 = 0; s.site = 0; s.sk = sk;
  return s;
}
/* 32 random bits for element c of the stream (every consumer uses the TOP 23 of them: bits >> 9).
 *   JAX32: x0 ^ x1 of Threefry(site key, (0, c))   (_threefry_random_bits_partitionable).
 *   FLAT : the site's stream is the concatenation of the 64-bit blocks Threefry(key, (i, (site << 22) | h)),
 *          h = 0, 1, ..., read as 32-bit words (word 2h = x0, word 2h+1 = x1, little end first).  Element c is the
 *          32-bit window that starts at stream bit 23*c, so its top 23 bits are stream bits [23c + 9, 23c + 32):
 *          consecutive elements use consecutive, disjoint 23-bit fields — 64 / 23 = 2.78 draws per hash. */
static inline uint32_t flat_word(const ostream* s, uint32_t n) {
  uint32_t o[2];
  gjxo_threefry2x32(s->key.a, s->key.b, s->c0, (s->site << GJX_FLAT_SITE_SHIFT) | (n >> 1), o);
  return o[n & 1];
}
static inline uint32_t elem_bits(const ostream* s, uint32_t c) {
  uint32_t o[2];
  if (s->mode == GJX_RNG_JAX32) {
    gjxo_threefry2x32(s->sk.a, s->sk.b, 0u, c, o);
    return o[0] ^ o[1];
  }
  const uint32_t bit = 23u * c, n = bit >> 5, sh = bit & 31u;
  const uint32_t lo = flat_word(s, n);
  if (sh == 0) return lo;
  return (lo >> sh) | (flat_word(s, n + 1) << (32u - sh));
}

static inline float bits_to_unit(uint32_t bits) { /* jax _uniform: [0,1) from 23 mantissa bits */
  uint32_t u = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &u, 4);
  return f - 1.0f;
}
static inline float uniform_from_bits(uint32_t bits, float lo, float hi) {
  float f = bits_to_unit(bits);
  float v = f * (hi - lo) + lo;
  return v > lo ? v : lo; /* lax.max(minval, ...) */
}

/* Giles (2010) single-precision erfinv — the polynomial XLA's ErfInv32 uses */
static float erfinv_f32(float x) {
  float w = -log1pf(-x * x);
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = 3.43273939e-07f + p * w;
    p = -3.5233877e-06f + p * w;
    p = -4.39150654e-06f + p * w;
    p = 0.00021858087f + p * w;
    p = -0.00125372503f + p * w;
    p = -0.00417768164f + p * w;
    p = 0.246640727f + p * w;
    p = 1.50140941f + p * w;
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = 0.000100950558f + p * w;
    p = 0.00134934322f + p * w;
    p = -0.00367342844f + p * w;
    p = 0.00573950773f + p * w;
    p = -0.0076224613f + p * w;
    p = 0.00943887047f + p * w;
    p = 1.00167406f + p * w;
    p = 2.83297682f + p * w;
  }
  return p * x;
}
float gjxo_erfinv(float x) { return erfinv_f32(x); }

#define NEG1_PLUS_ULP (-0.99999994f) /* nextafter(-1, 0) */
#define F32_TINY 1.17549435e-38f
#define SQRT2_F 1.41421356f
#define HALF_LOG_2PI 0.918938533f
#define LOG_PI 1.14472989f

static inline float normal_from_bits(uint32_t bits) { /* jax _normal_real */
  float u = uniform_from_bits(bits, NEG1_PLUS_ULP, 1.0f);
  return SQRT2_F * erfinv_f32(u);
}
static inline float gumbel_from_bits(uint32_t bits) { /* jax.random.gumbel */
  float u = uniform_from_bits(bits, F32_TINY, 1.0f);
  return -logf(-logf(u));
}

/* Standard normal for element e of a stream.
 *   JAX32: sqrt(2) * erfinv(uniform(-1,1)) of the element's 32 bits (jax.random.normal).
 *   FLAT : Box-Muller on the elements (e & ~1, e | 1) of the stream: u1 = 1 - unit(even element) in (0,1],
 *          u2 = unit(odd element); even elements take r*cos(2 pi u2), odd ones r*sin(2 pi u2).  An exact
 *          sampler that needs one log, one sqrt and one sin/cos per PAIR instead of an erfinv per draw. */
static float stream_normal(const ostream* s, uint32_t e) {
  if (s->mode == GJX_RNG_JAX32) return normal_from_bits(elem_bits(s, e));
  const float u1 = 1.0f - bits_to_unit(elem_bits(s, e & ~1u));
  const float u2 = bits_to_unit(elem_bits(s, e | 1u));
  const float r = sqrtf(-2.0f * logf(u1));
  const float ang = 6.28318530718f * u2;
  return r * ((e & 1u) ? sinf(ang) : cosf(ang));
}

float gjxo_normal_from_bits(uint32_t bits) { return normal_from_bits(bits); }
float gjxo_gumbel_from_bits(uint32_t bits) { return gumbel_from_bits(bits); }
float gjxo_unit_from_bits(uint32_t bits) { return bits_to_unit(bits); }

/* Marsaglia & Tsang (2000) gamma sampler in log space; the draw budget per gamma variate is
 * fixed so that element indices are a pure function of (variate, iteration). */
#define POISSON_TRIES 16
#define VON_MISES_TRIES 16
#define GAMMA_MAXIT 32
#define GAMMA_NDRAW (4 * GAMMA_MAXIT + 2)
/* draw schedule of one gamma variate (element indices relative to `base`): iteration t takes its normal
 * from element 4t and its uniform from element 4t+2 (a FLAT normal consumes the element pair (4t, 4t+1));
 * the a < 1 boost uniform is element 4*MAXIT. */
static float log_gamma_variate(const ostream* sk, uint32_t base, float a) {
  float boost = 0.0f;
  float aa = a;
  if (a < 1.0f) {
    float u = uniform_from_bits(elem_bits(sk, base + 4 * GAMMA_MAXIT), F32_TINY, 1.0f);
    boost = logf(u) / a;
    aa = a + 1.0f;
  }
  float d = aa - (1.0f / 3.0f);
  float c = 1.0f / sqrtf(9.0f * d);
  float res = logf(d);
  for (int t = 0; t < GAMMA_MAXIT; ++t) {
    float x = stream_normal(sk, base + 4 * t);
    float u = uniform_from_bits(elem_bits(sk, base + 4 * t + 2), F32_TINY, 1.0f);
    float v = 1.0f + c * x;
    decide(v, 0.0f);
    if (v <= 0.0f) continue;
    float lv = 3.0f * logf(v);
    v = v * v * v;
    decide(logf(u), 0.5f * x * x + d - d * v + d * lv);
    if (logf(u) < 0.5f * x * x + d - d * v + d * lv) {
      res = logf(d) + lv;
      break;
    }
  }
  return res + boost;
}

/* --------------------------------------------------------------------------------------- */
static inline float softplusf(float x) { /* log(1+exp(x)), stable */
  return (x > 0.0f ? x : 0.0f) + log1pf(expf(-fabsf(x)));
}
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float xlogyf(float x, float y) { return x == 0.0f ? 0.0f : x * logf(y); }
static inline float xlog1pyf(float x, float y) { return x == 0.0f ? 0.0f : x * log1pf(y); }

/* GJX_P_VGATHER: the slot the parameter reads — row idx of an earlier vector-valued choice, idx an earlier discrete choice (or, when
 * that choice is constrained to one value for every particle, its value in the table) */
static int vgather_row(const gjx_param* p, int d, const float* tab, const float* vals) {
  int idx = (int)(p->slot >= 0 ? vals[p->slot] : tab[p->off]);
  if (idx < 0) idx = 0;
  if (idx > p->n - 1) idx = p->n - 1;
  return p->moff + idx * p->len + (d % p->len);
}

/* GJX_P_EXPR (gjx.h): the block of scalar SSA nodes tab[off + 6 i ..] = {op, a, b, c, da, db}; the reference interprets whatever JAX computes
 * between two sites in float32 (static.py:383-399, staging.py:286-298) — every node here is evaluated in double and rounded to
 * float32 once (a correctly rounded float32 operation).  `inst`: the plate instance (site_instance leaves it in pad_[0]). */
static double expr_unary_d(int op, double x) {
  switch (op) {
    case GJX_E_NEG: return -x;
    case GJX_E_EXP: return exp(x);
    case GJX_E_LOG: return log(x);
    case GJX_E_SQRT: return sqrt(x);
    case GJX_E_SQUARE: return x * x;
    case GJX_E_TANH: return tanh(x);
    case GJX_E_SIGMOID: return 1.0 / (1.0 + exp(-x));
    case GJX_E_SOFTPLUS: return (x > 0.0 ? x : 0.0) + log1p(exp(-fabs(x)));
    case GJX_E_ABS: return fabs(x);
    case GJX_E_SIN: return sin(x);
    case GJX_E_COS: return cos(x);
    case GJX_E_LOG1P: return log1p(x);
    default: return 1.0 / x; /* GJX_E_RECIP */
  }
}
static double expr_unary_deriv_d(int op, double x, double y) {
  switch (op) {
    case GJX_E_NEG: return -1.0;
    case GJX_E_EXP: return y;
    case GJX_E_LOG: return 1.0 / x;
    case GJX_E_SQRT: return 0.5 / y;
    case GJX_E_SQUARE: return 2.0 * x;
    case GJX_E_TANH: return 1.0 - y * y;
    case GJX_E_SIGMOID: return y * (1.0 - y);
    case GJX_E_SOFTPLUS: return 1.0 / (1.0 + exp(-x));
    case GJX_E_ABS: return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0);
    case GJX_E_SIN: return cos(x);
    case GJX_E_COS: return -sin(x);
    case GJX_E_LOG1P: return 1.0 / (1.0 + x);
    default: return -y * y; /* GJX_E_RECIP */
  }
}
static void expr_forward(const gjx_param* p, const float* tab, const float* vals, float* ev) {
  const float* nd = tab + p->off;
  const int inst = p->pad_[0];
  for (int i = 0; i < p->n && i < GJX_EXPR_MAX_NODES; ++i) {
    const float* q6 = nd + GJX_EXPR_NODE_FLOATS * i;
    const int op = (int)q6[0], c = (int)q6[3];
    const int a = (int)q6[1] + inst * (int)q6[4], b = (int)q6[2] + inst * (int)q6[5];
    double r;
    switch (op) {
      case GJX_E_CONST: r = tab[a]; break;
      case GJX_E_VALUE: r = vals[a]; break;
      case GJX_E_ADD: r = (double)ev[a] + ev[b]; break;
      case GJX_E_SUB: r = (double)ev[a] - ev[b]; break;
      case GJX_E_MUL: r = (double)ev[a] * ev[b]; break;
      case GJX_E_DIV: r = (double)ev[a] / ev[b]; break;
      case GJX_E_MAX: r = ev[a] >= ev[b] ? ev[a] : ev[b]; break;
      case GJX_E_MIN: r = ev[a] <= ev[b] ? ev[a] : ev[b]; break;
      case GJX_E_GT: decide(ev[a], ev[b]); r = ev[a] > ev[b] ? 1.0 : 0.0; break;   /* (a comparison: a near-tie may go the other way on the device) */
      case GJX_E_WHERE: r = ev[a] != 0.0f ? ev[b] : ev[c]; break;
      case GJX_E_LINV: r = tab[a]; for (int e = 0; e < c; ++e) r += (double)tab[a + 1 + e] * vals[b + e]; break;
      case GJX_E_LINN: r = tab[a]; for (int e = 0; e < c; ++e) r += (double)tab[a + 1 + e] * ev[b + e]; break;
      default: r = expr_unary_d(op, ev[a]); break;
    }
    ev[i] = (float)r;
  }
}
static int expr_out_node(const gjx_param* p, int d) { return p->n - p->len + (d % p->len); }
static void expr_backward(const gjx_param* p, int d, float g, const float* tab, const float* vals, float* grad) {
  float ev[GJX_EXPR_MAX_NODES];
  double ad[GJX_EXPR_MAX_NODES];
  expr_forward(p, tab, vals, ev);
  const float* nd = tab + p->off;
  const int inst = p->pad_[0];
  const int n = p->n < GJX_EXPR_MAX_NODES ? p->n : GJX_EXPR_MAX_NODES;
  for (int i = 0; i < n; ++i) ad[i] = 0.0;
  ad[expr_out_node(p, d)] = g;
  for (int i = n - 1; i >= 0; --i) {
    const double gi = ad[i];
    if (gi == 0.0) continue;
    const float* q6 = nd + GJX_EXPR_NODE_FLOATS * i;
    const int op = (int)q6[0], c = (int)q6[3];
    const int a = (int)q6[1] + inst * (int)q6[4], b = (int)q6[2] + inst * (int)q6[5];
    switch (op) {
      case GJX_E_CONST: case GJX_E_GT: break;
      case GJX_E_VALUE: grad[a] += (float)gi; break;
      case GJX_E_ADD: ad[a] += gi; ad[b] += gi; break;
      case GJX_E_SUB: ad[a] += gi; ad[b] -= gi; break;
      case GJX_E_MUL: ad[a] += gi * ev[b]; ad[b] += gi * ev[a]; break;
      case GJX_E_DIV: ad[a] += gi / ev[b]; ad[b] -= gi * ev[i] / ev[b]; break;
      case GJX_E_MAX: if (ev[a] >= ev[b]) ad[a] += gi; else ad[b] += gi; break;
      case GJX_E_MIN: if (ev[a] <= ev[b]) ad[a] += gi; else ad[b] += gi; break;
      case GJX_E_WHERE: if (ev[a] != 0.0f) ad[b] += gi; else ad[c] += gi; break;
      case GJX_E_LINV: for (int e = 0; e < c; ++e) grad[b + e] += (float)(gi * tab[a + 1 + e]); break;
      case GJX_E_LINN: for (int e = 0; e < c; ++e) ad[b + e] += gi * tab[a + 1 + e]; break;
      default: ad[a] += gi * expr_unary_deriv_d(op, ev[a], ev[i]); break;
    }
  }
}

static float eval_param(const gjx_param* p, int d, const float* tab, const float* vals) {
  float v;
  switch (p->op) {
    case GJX_P_CONST: v = tab[p->off + (d % p->len)]; break;
    case GJX_P_VALUE: v = vals[p->slot + (d % p->len)]; break;
    case GJX_P_GATHER: {
      int idx = (int)vals[p->slot];
      if (idx < 0) idx = 0;
      if (idx > p->n - 1) idx = p->n - 1;
      v = tab[p->off + idx * p->len + (d % p->len)];
      break;
    }
    case GJX_P_AFFINE: {
      float acc = tab[p->off + (d % p->len)];
      for (int e = 0; e < p->n; ++e) acc += tab[p->moff + d * p->n + e] * vals[p->slot + e];
      v = acc;
      break;
    }
    case GJX_P_VGATHER: v = vals[vgather_row(p, d, tab, vals)]; break;
    case GJX_P_EXPR: { float ev[GJX_EXPR_MAX_NODES]; expr_forward(p, tab, vals, ev); v = ev[expr_out_node(p, d)]; break; }
    default: v = NAN;
  }
  switch (p->xf) {
    case GJX_XF_EXP: v = expf(v); break;
    case GJX_XF_SOFTPLUS: v = softplusf(v); break;
    case GJX_XF_SIGMOID: v = sigmoidf_(v); break;
    default: break;
  }
  return v;
}

/* log-density of ONE scalar element (TFP 0.23 log_prob expressions) */
static float std_normal_cdf(float z) { return 0.5f * erfcf(-z / 1.41421356f); }
/* Phi(hi) - Phi(lo), through the upper tail when both bounds are positive */
static float normal_interval_mass(float lo, float hi) {
  if (lo > 0.0f) return std_normal_cdf(-lo) - std_normal_cdf(-hi);
  return std_normal_cdf(hi) - std_normal_cdf(lo);
}
static int params_of(int kind) {
  return (kind == GJX_TRUNCATED_NORMAL || kind == GJX_TRUNCATED_CAUCHY) ? 4 : ((kind == GJX_STUDENT_T || kind == GJX_HALF_STUDENT_T) ? 3 : 2);
}

/* parameter order follows the reference's constructor arguments (tfp wrappers, tensorflow_probability/__init__.py) */
/* The closed-form log-densities are evaluated in DOUBLE from the float32 inputs and rounded once: the oracle is the
 * correctly rounded value of TFP's formula at these inputs.  (A float32 evaluation has rounding of its own — a gamma log-pdf at
 * concentration 1e6 is a difference of terms of 1e7 and moves in steps of 0.25 — which would make the checker wrong in its own
 * way where the device is wrong in another; and the conditioning probe of the differential tests, which perturbs the inputs by
 * a few ulps, can only see ill-conditioning in a function that is not piecewise constant.) */
static inline double xlogy_d(double x, double y) { return x == 0.0 ? 0.0 : x * log(y); }
static inline double xlog1py_d(double x, double y) { return x == 0.0 ? 0.0 : x * log1p(y); }
static inline double softplus_d(double x) { return (x > 0.0 ? x : 0.0) + log1p(exp(-fabs(x))); }

/* exp(-x) I0(x) and I1(x) / I0(x), x >= 0: Abramowitz & Stegun 9.8.1 - 9.8.4, the polynomials the device evaluates (in double here) */
static double bessel_i0_poly(double x, int one) {
  if (x < 3.75) {
    double t2 = (x / 3.75) * (x / 3.75);
    if (one) return x * (0.5 + t2 * (0.87890594 + t2 * (0.51498869 + t2 * (0.15084934 + t2 * (0.02658733 + t2 * (0.00301532 + t2 * 0.00032411))))));
    return 1.0 + t2 * (3.5156229 + t2 * (3.0899424 + t2 * (1.2067492 + t2 * (0.2659732 + t2 * (0.0360768 + t2 * 0.0045813)))));
  }
  double u = 3.75 / x;
  if (one) return 0.39894228 + u * (-0.03988024 + u * (-0.00362018 + u * (0.00163801 + u * (-0.01031555 + u * (0.02282967 + u * (-0.02895312 + u * (0.01787654 + u * -0.00420059)))))));
  return 0.39894228 + u * (0.01328592 + u * (0.00225319 + u * (-0.00157565 + u * (0.00916281 + u * (-0.02057706 + u * (0.02635537 + u * (-0.01647633 + u * 0.00392377)))))));
}
static double bessel_i0e_d(double x) { return x < 3.75 ? bessel_i0_poly(x, 0) * exp(-x) : bessel_i0_poly(x, 0) / sqrt(x); }
static double bessel_i1_over_i0_d(double x) { return bessel_i0_poly(x, 1) / bessel_i0_poly(x, 0); }

static float elem_logpdf4(int kind, float xf, float af, float bf, float cf, float df) {
  const double x = xf, a = af, b = bf, c = cf, d = df;
  switch (kind) {
    case GJX_TRUNCATED_CAUCHY: { /* tfd.TruncatedCauchy(loc=a, scale=b, low=c, high=d) */
      if (x < c || x > d) return -INFINITY;
      double z = (x - a) / b;
      return (float)(-log(b) - log1p(z * z) - log(atan((d - a) / b) - atan((c - a) / b)));
    }
    case GJX_NEGATIVE_BINOMIAL: /* tfd.NegativeBinomial(total_count=a, logits=b): successes before a failures, success probability sigmoid(b) */
      if (x < 0.0 || x != floor(x)) return -INFINITY;
      return (float)(lgamma(x + a) - lgamma(x + 1.0) - lgamma(a) - (x == 0.0 ? 0.0 : x * softplus_d(-b)) - a * softplus_d(b));
    case GJX_VON_MISES: /* tfd.VonMises(loc=a, concentration=b) */
      return (float)(b * (cos(x - a) - 1.0) - 2.0 * (double)HALF_LOG_2PI - log(bessel_i0e_d(b)));
    case GJX_CHI: { /* tfd.Chi(df=a): the root of a chi2(a) variate */
      double h = 0.5 * a;
      return x <= 0.0 ? -INFINITY : (float)((1.0 - h) * log(2.0) + xlogy_d(a - 1.0, x) - 0.5 * x * x - lgamma(h));
    }
    case GJX_EXP_GAMMA: /* tfd.ExpGamma(concentration=a, rate=b): log of a gamma variate */
      return (float)(a * (log(b) + x) - b * exp(x) - lgamma(a));
    case GJX_EXP_INVERSE_GAMMA: /* tfd.ExpInverseGamma(concentration=a, scale=b): log of an inverse-gamma variate */
      return (float)(a * (log(b) - x) - b * exp(-x) - lgamma(a));
    case GJX_KUMARASWAMY: /* tfd.Kumaraswamy(concentration1=a, concentration0=b) */
      if (!(x > 0.0 && x < 1.0)) return -INFINITY;
      return (float)(log(a) + log(b) + xlogy_d(a - 1.0, x) + xlog1py_d(b - 1.0, -pow(x, a)));
    case GJX_MOYAL: { /* tfd.Moyal(loc=a, scale=b) */
      double z = (x - a) / b;
      return (float)(-0.5 * (z + exp(-z)) - log(b) - (double)HALF_LOG_2PI);
    }
    case GJX_DOUBLESIDED_MAXWELL: { /* tfd.DoublesidedMaxwell(loc=a, scale=b): z^2 exp(-z^2 / 2) / (b sqrt(2 pi)) */
      double z = (x - a) / b;
      return (float)(2.0 * log(fabs(z)) - 0.5 * z * z - log(b) - (double)HALF_LOG_2PI);
    }
    case GJX_INVERSE_GAUSSIAN: { /* tfd.InverseGaussian(loc=a, concentration=b) */
      if (x <= 0.0) return -INFINITY;
      double r = (x - a) / a;
      return (float)(0.5 * (log(b) - 2.0 * (double)HALF_LOG_2PI - 3.0 * log(x)) - 0.5 * b * r * r / x);
    }
    case GJX_HALF_STUDENT_T: /* tfd.HalfStudentT(df=a, loc=b, scale=c): the student-t folded at its location */
      if (x < b) return -INFINITY;
      /* fall through */
    case GJX_STUDENT_T: { /* tfd.StudentT(df=a, loc=b, scale=c) */
      double y = (x - b) / c, h = 0.5 * a, lgd;
      /* lgamma(h + 1/2) - lgamma(h) on its own (added to the small terms one lgamma at a time, a df of 1e20 absorbs them: the result
       * was exactly 0), and from the asymptotic series where the two values agree to within their own rounding */
      if (h < 1e6) lgd = lgamma(h + 0.5) - lgamma(h);
      else { double r = 1.0 / h; lgd = 0.5 * log(h) - r * (0.125 - r * r / 192.0); }
      return (float)(-0.5 * (a + 1.0) * log1p(y * y / a) - log(c) - 0.5 * log(a) - 0.5 * (double)LOG_PI + lgd + (kind == GJX_HALF_STUDENT_T ? log(2.0) : 0.0));
    }
    case GJX_TRUNCATED_NORMAL: { /* tfd.TruncatedNormal(loc=a, scale=b, low=c, high=d) */
      if (x < c || x > d) return -INFINITY;
      double z = (x - a) / b;
      return (float)(-0.5 * z * z - ((double)HALF_LOG_2PI + log(b)) - log((double)normal_interval_mass((float)((c - a) / b), (float)((d - a) / b))));
    }
    case GJX_POISSON: /* tfd.Poisson(rate=a) */
      return (x < 0.0 || x != floor(x)) ? -INFINITY : (float)(xlogy_d(x, a) - a - lgamma(x + 1.0));
    case GJX_GEOMETRIC: /* tfd.Geometric(probs=a): number of failures before the first success */
      return (x < 0.0 || x != floor(x)) ? -INFINITY : (float)(xlog1py_d(x, -a) + log(a));
    case GJX_GUMBEL: {
      double z = (x - a) / b;
      return (float)(-(z + exp(-z)) - log(b));
    }
    case GJX_HALF_CAUCHY: {
      double z = (x - a) / b;
      return x < a ? -INFINITY : (float)(log(2.0 / 3.14159265358979323846) - log(b) - log1p(z * z));
    }
    case GJX_INVERSE_GAMMA: /* concentration a, scale b */
      return x <= 0.0 ? -INFINITY : (float)(a * log(b) - lgamma(a) - (a + 1.0) * log(x) - b / x);
    case GJX_WEIBULL: { /* concentration a, scale b */
      if (x < 0.0) return -INFINITY;
      double lr = log(x / b);
      return (float)(log(a / b) + xlogy_d(a - 1.0, x / b) - exp(a * lr));
    }
    case GJX_LOGIT_NORMAL: {
      if (!(x > 0.0 && x < 1.0)) return -INFINITY;
      double lx = log(x), l1 = log1p(-x);
      double z = ((lx - l1) - a) / b;
      return (float)(-0.5 * z * z - ((double)HALF_LOG_2PI + log(b)) - lx - l1);
    }
    case GJX_CHI2: { /* df a: gamma(a/2, rate 1/2) */
      double h = 0.5 * a;
      return x <= 0.0 ? -INFINITY : (float)(xlogy_d(h - 1.0, x) - 0.5 * x - h * log(2.0) - lgamma(h));
    }
    default: return NAN;
  }
}

static float elem_logpdf(int kind, float xf, float af, float bf) {
  const double x = xf, a = af, b = bf;
  switch (kind) {
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: { /* tfd.Normal._log_prob */
      double z = x / b - a / b;
      return (float)(-0.5 * z * z - ((double)HALF_LOG_2PI + log(b)));
    }
    case GJX_FLIP: /* tfd.Bernoulli(probs): multiply_no_nan(log p, x) + multiply_no_nan(log1p(-p), 1-x) */
      return (float)((x != 0.0 ? log(a) : 0.0) + (x != 1.0 ? (1.0 - x) * log1p(-a) : 0.0));
    case GJX_BERNOULLI_LOGITS: /* -softplus(-l)*x - softplus(l)*(1-x) */
      return (float)((x != 0.0 ? -softplus_d(-a) * x : 0.0) + (x != 1.0 ? -softplus_d(a) * (1.0 - x) : 0.0));
    case GJX_BETA: /* xlogy(a-1,x) + xlog1py(b-1,-x) - lbeta(a,b) */
      return (float)(xlogy_d(a - 1.0, x) + xlog1py_d(b - 1.0, -x) - (lgamma(a) + lgamma(b) - lgamma(a + b)));
    case GJX_UNIFORM:
      return (x < a || x > b) ? -INFINITY : (float)(-log(b - a));
    case GJX_EXPONENTIAL: /* a = rate */
      return x < 0.0 ? -INFINITY : (float)(log(a) - a * x);
    case GJX_HALF_NORMAL: { /* a = scale */
      double z = x / a;
      return x < 0.0 ? -INFINITY : (float)(0.5 * log(2.0 / 3.14159265358979323846) - log(a) - 0.5 * z * z);
    }
    case GJX_LAPLACE:
      return (float)(-fabs(x - a) / b - log(2.0 * b));
    case GJX_LOG_NORMAL: {
      double lx = log(x);
      double z = lx / b - a / b;
      return (float)(-0.5 * z * z - ((double)HALF_LOG_2PI + log(b)) - lx);
    }
    case GJX_CAUCHY: {
      double z = (x - a) / b;
      return (float)(-((double)LOG_PI + log(b)) - log1p(z * z));
    }
    case GJX_GAMMA: /* a = concentration, b = rate */
      return (float)(xlogy_d(a, b) + xlogy_d(a - 1.0, x) - b * x - lgamma(a));
    default: return NAN;
  }
}

/* draws per scalar element, so that element indices are deterministic */
static int draws_per_elem(int kind) {
  switch (kind) {
    case GJX_BETA: return 2 * GAMMA_NDRAW;
    case GJX_GAMMA:
    case GJX_DIRICHLET:
    case GJX_INVERSE_GAMMA:
    case GJX_CHI2:
    case GJX_CHI:
    case GJX_EXP_GAMMA:
    case GJX_EXP_INVERSE_GAMMA: return GAMMA_NDRAW;
    case GJX_STUDENT_T:
    case GJX_HALF_STUDENT_T:
    case GJX_DOUBLESIDED_MAXWELL: return GAMMA_NDRAW + 2;
    case GJX_POISSON: return 2 * POISSON_TRIES + 2;
    case GJX_NEGATIVE_BINOMIAL: return GAMMA_NDRAW + 2 * POISSON_TRIES + 2;
    case GJX_VON_MISES: return 2 * VON_MISES_TRIES + 2;
    case GJX_INVERSE_GAUSSIAN: return 4;
    default: return 1;
  }
}

/* Poisson(lam): inversion by sequential search on one uniform below 10; Hormann's PTRS (1993) with a fixed
 * budget of tries above.  Element schedule: c for the inversion uniform, c+2+2t / c+3+2t for try t. */
static float poisson_variate(const ostream* sk, uint32_t c, float lam) {
  if (lam < 10.0f) {
    float u = bits_to_unit(elem_bits(sk, c));
    float p = expf(-lam), cdf = p;
    int k = 0;
    decide(u, cdf);
    while (u > cdf && k < 96) {
      ++k;
      p *= lam / (float)k;
      cdf += p;
      decide(u, cdf);
    }
    return (float)k;
  }
  float slam = sqrtf(lam), loglam = logf(lam);
  float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
  float inv_alpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.0f);
  for (int t = 0; t < POISSON_TRIES; ++t) {
    float U = bits_to_unit(elem_bits(sk, c + 2 + 2 * t)) - 0.5f;
    float V = uniform_from_bits(elem_bits(sk, c + 3 + 2 * t), F32_TINY, 1.0f);
    float us = 0.5f - fabsf(U);
    float k = floorf((2.0f * a / us + b) * U + lam + 0.43f);
    { const float kraw = (2.0f * a / us + b) * U + lam + 0.43f; decide(kraw, floorf(kraw)); decide(kraw, floorf(kraw) + 1.0f); }
    decide(us, 0.07f); decide(V, vr);
    if (us >= 0.07f && V <= vr) return k;
    decide(us, 0.013f); decide(V, us);
    if (k < 0.0f || (us < 0.013f && V > us)) continue;
    decide(logf(V) + logf(inv_alpha) - logf(a / (us * us) + b), -lam + k * loglam - lgammaf(k + 1.0f));
    if (logf(V) + logf(inv_alpha) - logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.0f)) return k;
  }
  return floorf(lam);
}

static float elem_sample4(int kind, const ostream* sk, uint32_t c, float a, float b, float p3, float p4) {
  switch (kind) {
    case GJX_HALF_STUDENT_T:
    case GJX_STUDENT_T: { /* z * sqrt(df / chi2_df), chi2_df = 2 Gamma(df/2, 1) */
      float z = stream_normal(sk, c);
      float lg = log_gamma_variate(sk, c + 2, 0.5f * a);
      float t = p3 * z * expf(0.5f * (logf(0.5f * a) - lg));
      return b + (kind == GJX_HALF_STUDENT_T ? fabsf(t) : t);
    }
    case GJX_TRUNCATED_CAUCHY: { /* inverse CDF on the arctangent scale */
      float lo = atanf((p3 - a) / b), hi = atanf((p4 - a) / b);
      float x = a + b * tanf(lo + bits_to_unit(elem_bits(sk, c)) * (hi - lo));
      return x < p3 ? p3 : (x > p4 ? p4 : x);
    }
    case GJX_NEGATIVE_BINOMIAL: { /* a gamma(r, rate e^-l) mixture of Poissons */
      float lg = log_gamma_variate(sk, c, a);
      return poisson_variate(sk, c + GAMMA_NDRAW, expf(lg + b));
    }
    case GJX_VON_MISES: { /* Best & Fisher (1979), fixed budget of tries (elements 2 t, 2 t + 1; the sign: the last one) */
      if (b < 1e-6f) return a + 3.14159265f * (2.0f * bits_to_unit(elem_bits(sk, c)) - 1.0f);
      double kd = b, tau = 1.0 + sqrt(1.0 + 4.0 * kd * kd), rho = (tau - sqrt(2.0 * tau)) / (2.0 * kd);
      float r = (float)((1.0 + rho * rho) / (2.0 * rho));
      float f = 1.0f;
      for (int t = 0; t < VON_MISES_TRIES; ++t) {
        float z = cosf(3.14159265f * bits_to_unit(elem_bits(sk, c + 2 * t)));
        float u2 = uniform_from_bits(elem_bits(sk, c + 2 * t + 1), F32_TINY, 1.0f);
        f = (r * z + 1.0f) / (r + z);
        float cc = b * (r - f);
        decide(u2, cc * (2.0f - cc));
        if (u2 < cc * (2.0f - cc)) break;
        decide(logf(cc / u2) + 1.0f - cc, 0.0f);
        if (logf(cc / u2) + 1.0f - cc >= 0.0f) break;
      }
      float th = acosf(f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f));
      float us = bits_to_unit(elem_bits(sk, c + 2 * VON_MISES_TRIES));
      decide(us, 0.5f);
      return a + (us < 0.5f ? -th : th);
    }
    case GJX_CHI: return expf(0.5f * (0.69314718f + log_gamma_variate(sk, c, 0.5f * a)));
    case GJX_EXP_GAMMA: return log_gamma_variate(sk, c, a) - logf(b);
    case GJX_EXP_INVERSE_GAMMA: return logf(b) - log_gamma_variate(sk, c, a);
    case GJX_KUMARASWAMY: { /* x = (1 - (1 - u)^(1 / b))^(1 / a) */
      float t = log1pf(-bits_to_unit(elem_bits(sk, c))) / b;
      float m = -expm1f(t);
      return expf(logf(m) / a);
    }
    case GJX_MOYAL: { /* -log of a chi2(1) variate: loc - scale log(n^2) */
      float n = fabsf(stream_normal(sk, c));
      return a - 2.0f * b * logf(n);
    }
    case GJX_DOUBLESIDED_MAXWELL: { /* a random sign times the root of a chi2(3) variate */
      const float u = bits_to_unit(elem_bits(sk, c));
      decide(u, 0.5f);
      float sgn = u < 0.5f ? -1.0f : 1.0f;
      float r = expf(0.5f * (0.69314718f + log_gamma_variate(sk, c + 2, 1.5f)));
      return a + b * sgn * r;
    }
    case GJX_INVERSE_GAUSSIAN: { /* Michael, Schucany & Haas (1976): a = mean, b = concentration */
      float n = stream_normal(sk, c);
      float y = n * n;
      float w = a * y / (2.0f * b);                          /* the smaller root mu (1 + w - sqrt(w (w + 2))) without its cancellation */
      float x1 = a / (1.0f + w + sqrtf(w * (w + 2.0f)));
      float u = bits_to_unit(elem_bits(sk, c + 2));
      decide(u * (a + x1), a);
      return u * (a + x1) <= a ? x1 : a * a / x1;
    }
    case GJX_TRUNCATED_NORMAL: {
      float lo = (p3 - a) / b, hi = (p4 - a) / b;
      float u = bits_to_unit(elem_bits(sk, c));
      float z;
      if (lo > 0.0f) {
        float q = std_normal_cdf(-lo) - u * (std_normal_cdf(-lo) - std_normal_cdf(-hi));
        z = -1.41421356f * erfinv_f32(2.0f * q - 1.0f);
      } else {
        float q = std_normal_cdf(lo) + u * (std_normal_cdf(hi) - std_normal_cdf(lo));
        z = 1.41421356f * erfinv_f32(2.0f * q - 1.0f);
      }
      float x = a + b * z;
      return x < p3 ? p3 : (x > p4 ? p4 : x);
    }
    case GJX_POISSON: return poisson_variate(sk, c, a);
    case GJX_GEOMETRIC: {
      const float raw = logf(uniform_from_bits(elem_bits(sk, c), F32_TINY, 1.0f)) / log1pf(-a);
      decide(raw, floorf(raw)); decide(raw, floorf(raw) + 1.0f);
      return floorf(raw);
    }
    case GJX_GUMBEL: return a - b * logf(-logf(uniform_from_bits(elem_bits(sk, c), F32_TINY, 1.0f)));
    case GJX_HALF_CAUCHY: return a + b * tanf(0.5f * 3.14159265f * bits_to_unit(elem_bits(sk, c)));
    case GJX_INVERSE_GAMMA: return b * expf(-log_gamma_variate(sk, c, a));
    case GJX_WEIBULL: return b * expf(logf(-log1pf(-bits_to_unit(elem_bits(sk, c)))) / a);
    case GJX_LOGIT_NORMAL: return sigmoidf_(a + b * stream_normal(sk, c));
    case GJX_CHI2: return 2.0f * expf(log_gamma_variate(sk, c, 0.5f * a));
    default: return NAN;
  }
}

static float elem_sample(int kind, const ostream* sk, uint32_t c, float a, float b) {
  switch (kind) {
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: return a + b * stream_normal(sk, c);
    case GJX_FLIP: decide(bits_to_unit(elem_bits(sk, c)), a); return bits_to_unit(elem_bits(sk, c)) < a ? 1.0f : 0.0f;
    case GJX_BERNOULLI_LOGITS: decide(bits_to_unit(elem_bits(sk, c)), sigmoidf_(a)); return bits_to_unit(elem_bits(sk, c)) < sigmoidf_(a) ? 1.0f : 0.0f;
    case GJX_BETA: {
      float g1 = log_gamma_variate(sk, c, a);
      float g2 = log_gamma_variate(sk, c + GAMMA_NDRAW, b);
      return sigmoidf_(g1 - g2);
    }
    case GJX_UNIFORM: return a + (b - a) * bits_to_unit(elem_bits(sk, c));
    case GJX_EXPONENTIAL: return -logf(uniform_from_bits(elem_bits(sk, c), F32_TINY, 1.0f)) / a;
    case GJX_HALF_NORMAL: return fabsf(stream_normal(sk, c)) * a;
    case GJX_LAPLACE: {
      float u = uniform_from_bits(elem_bits(sk, c), NEG1_PLUS_ULP, 1.0f);
      float s = (u > 0.0f) - (u < 0.0f);
      return a - b * s * log1pf(-fabsf(u));
    }
    case GJX_LOG_NORMAL: return expf(a + b * stream_normal(sk, c));
    case GJX_CAUCHY: return a + b * tanf(3.14159265f * (bits_to_unit(elem_bits(sk, c)) - 0.5f));
    case GJX_GAMMA: return expf(log_gamma_variate(sk, c, a)) / b;
    default: return NAN;
  }
}

/* stream key and site number of every site (gjx.h "Scan steps"): sites of a Scan step use the chained step key
 * key_t = fold_in(key_{t-1}, t) (scan.py:268) and their position within the step; the others the run key and their
 * position among the non-Scan sites.  Scalar-normal runs: gjx.h. */
typedef struct {
  okey run_key, skey;
  int32_t tag;
  uint32_t local, plain;
  uint32_t run_head, run_next; /* open scalar-normal run: head's site number, next element */
  uint32_t jn;                 /* sites seen that are not GJX_MODE_INPUT (those take no site number) */
} site_walk;

/* -> site number of site s (FLAT numbering); *e0 = element of the stream at which the site's draws start */
static uint32_t walk_next(site_walk* w, const gjx_program* prog, const gjx_site* s, int j, uint32_t* e0) {
  (void)j;
  uint32_t site_no = ++w->jn;
  *e0 = 0u;
  if (prog->rng_mode != GJX_RNG_FLAT) return site_no;
  if (s->scan == 0) { if (w->tag != 0) w->run_head = 0u; w->skey = w->run_key; w->tag = 0; site_no = ++w->plain; }
  else {
    if (s->scan != w->tag) {
      w->run_head = 0u;
      const uint32_t id = GJX_SCAN_ID(s->scan);
      const int32_t step = GJX_SCAN_STEP(s->scan);
      if (w->tag != 0 && GJX_SCAN_ID(w->tag) == id && GJX_SCAN_STEP(w->tag) == step - 1) w->skey = fold_in(w->skey, (uint32_t)step);
      else {
        w->skey = fold_in(w->run_key, 0x80000000u | id);
        for (int32_t t = 0; t <= step; ++t) w->skey = fold_in(w->skey, (uint32_t)t);
      }
      w->tag = s->scan;
      w->local = 0u;
    }
    site_no = ++w->local;
  }
  /* the STATIC mode decides membership (a masked site draws per particle: it closes the run); sites of a plate never join */
  if (s->plate == 0 && GJX_FLAT_JOINS(prog->rng_mode, s->kind, s->dim, s->mode)) {
    if (w->run_head == 0u || w->run_next >= (uint32_t)GJX_FLAT_RUN_MAX) { w->run_head = site_no; w->run_next = 0u; }
    site_no = w->run_head;
    *e0 = w->run_next++;
  } else if (s->mode == GJX_MODE_SAMPLE || s->mode == GJX_MODE_OBS_MASK) {
    w->run_head = 0u;
  }
  return site_no;
}

/* One site (instance `inst` of it when the site belongs to a plate: gjx.h "Plates") for one particle.  `sk`: the site's
 * stream; ebase: element at which this (site, instance) starts drawing.  -> log-density; *given: the value was constrained */
static float run_site(const gjx_program* prog, const gjx_site* s0, int inst, const ostream* sk, uint32_t ebase, float* vals, int* given) {
  const float* tab = prog->tab;
  gjx_site sv = *s0;                   /* the site with its per-instance offsets applied */
  gjx_site* s = &sv;
  if (s0->plate && inst) {
    const int w = (s->kind == GJX_CATEGORICAL_LOGITS || s->kind == GJX_CATEGORICAL_PROBS) ? 1 : s->dim;
    if (s->slot >= 0) s->slot += inst * w;
    s->obs_off += inst * s->d_obs;
    for (int k = 0; k < GJX_MAX_PARAMS; ++k) {
      s->p[k].off += inst * s->p[k].d_off; s->p[k].moff += inst * s->p[k].d_moff;
      if (s->p[k].slot >= 0) s->p[k].slot += inst * s->p[k].d_slot; /* (VGATHER with slot < 0: the index comes from the table) */
      if (s->p[k].op == GJX_P_EXPR) { s->p[k].off = s0->p[k].off; s->p[k].pad_[0] = inst; }   /* a block's nodes carry their own strides */
    }
  }
  /* Mask(value, flag) is a per-particle lax.cond between the constrained and the unconstrained rule
   * (distribution.py:129-143): resolve it to one of the two plain modes for THIS particle */
  int mode = s->mode;
  if (mode == GJX_MODE_OBS_PROPOSED) mode = GJX_MODE_OBS_SLOT;   /* the proposal's draw sits in the slot (vals[] is this particle's working state) */
  if (mode == GJX_MODE_OBS_MASK) mode = vals[s->obs_off] != 0.0f ? GJX_MODE_OBS_SLOT : GJX_MODE_SAMPLE;
  *given = mode != GJX_MODE_SAMPLE;
  float lp = 0.0f;
  if (s->kind == GJX_CATEGORICAL_LOGITS || s->kind == GJX_CATEGORICAL_PROBS) {
    /* logits (probs -> log p), log_softmax; sample = argmax(logits + Gumbel) */
    int n = s->ncat;
    float mx = -INFINITY;
    for (int c = 0; c < n; ++c) {
      float l = eval_param(&s->p[0], c, tab, vals);
      if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
      if (l > mx) mx = l;
    }
    double se = 0.0;
    for (int c = 0; c < n; ++c) {
      float l = eval_param(&s->p[0], c, tab, vals);
      if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
      se += exp((double)l - (double)mx);
    }
    float lse = mx + (float)log(se);
    float v;
    if (mode == GJX_MODE_SAMPLE && prog->rng_mode == GJX_RNG_FLAT) {
      /* FLAT layout: inverse CDF on ONE uniform (float32 running sum of exp(l - max), category order) */
      float tot = 0.0f;
      for (int c = 0; c < n; ++c) {
        float l = eval_param(&s->p[0], c, tab, vals);
        if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
        tot += expf(l - mx);
      }
      const float target = bits_to_unit(elem_bits(sk, ebase)) * tot;
      float run = 0.0f;
      int zc = n - 1;
      for (int c = 0; c < n; ++c) {
        float l = eval_param(&s->p[0], c, tab, vals);
        if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
        run += expf(l - mx);
        if (c < n - 1) decide(run, target);
        if (run > target) { zc = c; break; }
      }
      v = (float)zc;
    } else if (mode == GJX_MODE_SAMPLE) { /* JAX32 layout: Gumbel-max (jax.random.categorical) */
      int best = 0;
      float bestv = -INFINITY, second = -INFINITY;
      for (int c = 0; c < n; ++c) {
        float l = eval_param(&s->p[0], c, tab, vals);
        if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
        float g = l + gumbel_from_bits(elem_bits(sk, ebase + (uint32_t)c));
        if (g > bestv) { second = bestv; bestv = g; best = c; } else if (g > second) second = g;
      }
      if (n > 1) decide(bestv, second);
      v = (float)best;
    } else if (mode == GJX_MODE_OBS_TAB) {
      v = tab[s->obs_off];
    } else {
      v = vals[s->slot];
    }
    int k = (int)v;
    if (k < 0 || k >= n) {
      lp = -INFINITY;
    } else {
      float l = eval_param(&s->p[0], k, tab, vals);
      if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
      lp = l - lse;
    }
    if (s->slot >= 0) vals[s->slot] = v;
  } else if (s->kind == GJX_DIRICHLET) { /* tfd.Dirichlet(concentration): gamma variates normalised; joint density */
    int n = s->dim;
    float x[256];
    if (n > 256) n = 256;
    if (mode == GJX_MODE_SAMPLE) {
      float mx = -INFINITY, se = 0.0f;
      for (int d = 0; d < n; ++d) {
        x[d] = log_gamma_variate(sk, ebase + (uint32_t)(d * GAMMA_NDRAW), eval_param(&s->p[0], d, tab, vals));
        if (x[d] > mx) mx = x[d];
      }
      for (int d = 0; d < n; ++d) se += expf(x[d] - mx);
      for (int d = 0; d < n; ++d) x[d] = expf(x[d] - (mx + logf(se)));
    } else {
      for (int d = 0; d < n; ++d) x[d] = mode == GJX_MODE_OBS_TAB ? tab[s->obs_off + d] : vals[s->slot + d];
    }
    float sa = 0.0f;
    for (int d = 0; d < n; ++d) {
      float al = eval_param(&s->p[0], d, tab, vals);
      sa += al;
      lp += xlogyf(al - 1.0f, x[d]) - lgammaf(al);
      if (s->slot >= 0) vals[s->slot + d] = x[d];
    }
    lp += lgammaf(sa);
  } else {
    int nd = draws_per_elem(s->kind);
    int np = s->kind >= GJX_STUDENT_T ? params_of(s->kind) : 2;
    for (int d = 0; d < s->dim; ++d) {
      float a = eval_param(&s->p[0], d, tab, vals);
      float b = eval_param(&s->p[1], d, tab, vals);
      float c = np > 2 ? eval_param(&s->p[2], d, tab, vals) : 0.0f;
      float e = np > 3 ? eval_param(&s->p[3], d, tab, vals) : 0.0f;
      int wide = s->kind >= GJX_STUDENT_T;
      float v;
      if (mode == GJX_MODE_SAMPLE) v = wide ? elem_sample4(s->kind, sk, ebase + (uint32_t)(d * nd), a, b, c, e) : elem_sample(s->kind, sk, ebase + (uint32_t)(d * nd), a, b);
      else if (mode == GJX_MODE_OBS_TAB) v = tab[s->obs_off + d];
      else v = vals[s->slot + d];
      lp += wide ? elem_logpdf4(s->kind, v, a, b, c, e) : elem_logpdf(s->kind, v, a, b); /* distribution.py:392-396: summed over the event */
      if (s->slot >= 0) vals[s->slot + d] = v;
    }
  }
  return lp;
}

/* One particle through the site list.  vals[n_slots] in/out.  Returns via pointers. */
static void run_particle(const gjx_program* prog, okey run_key, uint64_t idx, float* vals, float* score_out,
                         float* weight_out, float* site_scores, int64_t ss_stride) {
  float score = 0.0f, weight = 0.0f;
  site_walk w = {run_key, run_key, 0, 0u, 0u, 0u, 0u, 0u};
  const int flat = prog->rng_mode == GJX_RNG_FLAT;
  for (int j = 0; j < prog->n_sites;) {
    const gjx_site* s = &prog->sites[j];
    if (s->mode == GJX_MODE_INPUT) { /* the carry / arguments: rows that are already there; no draw, no score, no site number */
      if (site_scores) site_scores[(int64_t)j * ss_stride] = 0.0f;
      ++j;
      continue;
    }
    if (s->plate == 0) {
      uint32_t e0;
      const uint32_t site_no = walk_next(&w, prog, s, j, &e0);
      const ostream st = stream_open(prog->rng_mode, flat ? w.skey : run_key, idx, site_no); /* counter starts at 1 */
      int given;
      const float lp = run_site(prog, s, 0, &st, e0, vals, &given);
      if (s->flags & GJX_SITE_PROPOSAL) weight -= lp; /* a proposal's site: log w = log p - log q (smc.py:313), no part of the score */
      else {
        score += lp;
        if (given) weight += lp; /* static.py:377 with distribution.py:127/147 */
      }
      if (site_scores) site_scores[(int64_t)j * ss_stride] = lp;
      ++j;
      continue;
    }
    /* a plate (gjx.h "Plates"; vmap.py:180-218): m body sites, n instances, ONE instance loop over the body */
    int m = 1;
    while (j + m < prog->n_sites && prog->sites[j + m].plate == s->plate) ++m;
    const int n = s->plate_n;
    uint32_t site_no[64];
    float acc[64];
    if (m > 64) m = 64; /* (the host never emits more) */
    for (int l = 0; l < m; ++l) { uint32_t e0; site_no[l] = walk_next(&w, prog, &prog->sites[j + l], j + l, &e0); acc[l] = 0.0f; }
    if (!flat && prog->sites[j].plate) w.jn -= (uint32_t)(m - 1);   /* JAX32: the Vmap call advances its caller's counter by ONE (static.py:349-352) */
    /* JAX32: the Vmap call is one traced site of its caller: plate key = fold_in(particle key, J), J = 1-based index of the
     * plate's first site; instance key = split(plate key, n)[i] (vmap.py:186, 201) */
    okey pkey = run_key;
    if (!flat) pkey = fold_in(fold_in64(run_key, idx), site_no[0]);
    for (int i = 0; i < n; ++i) {
      okey ikey = pkey;
      if (!flat) ikey = fold_in(pkey, (uint32_t)i);
      for (int l = 0; l < m; ++l) {
        const gjx_site* sl = &prog->sites[j + l];
        ostream st;
        uint32_t ebase = 0u;
        if (flat) {
          st = stream_open(prog->rng_mode, w.skey, idx, site_no[l]);
          const int cat = sl->kind == GJX_CATEGORICAL_LOGITS || sl->kind == GJX_CATEGORICAL_PROBS;
          ebase = cat ? (uint32_t)i : (uint32_t)i * (uint32_t)sl->dim * (uint32_t)draws_per_elem(sl->kind);
        } else {
          st = stream_from_site_key(fold_in(ikey, (uint32_t)(l + 1))); /* static.py:349-352 inside the kernel */
        }
        int given;
        const float lp = run_site(prog, sl, i, &st, ebase, vals, &given);
        if (sl->flags & GJX_SITE_PROPOSAL) weight -= lp;
        else {
          score += lp;
          if (given) weight += lp;
        }
        acc[l] += lp;
      }
    }
    if (site_scores) for (int l = 0; l < m; ++l) site_scores[(int64_t)(j + l) * ss_stride] = acc[l];
    j += m;
  }
  *score_out = score;
  *weight_out = weight;
}

static void lse4(const float* x, int64_t K, int64_t K_total, float* out) {
  float mx = -INFINITY;
  for (int64_t i = 0; i < K; ++i)
    if (x[i] > mx) mx = x[i];
  double s = 0.0;
  if (mx > -INFINITY)
    for (int64_t i = 0; i < K; ++i) s += exp((double)x[i] - (double)mx);
  out[0] = mx;
  out[1] = (float)s;
  out[2] = (mx > -INFINITY) ? (float)((double)mx + log(s)) : -INFINITY;
  out[3] = (float)((double)out[2] - log((double)K_total));
}

int gjxo_logsumexp(const float* x, int64_t K, int64_t K_total, float* out) {
  lse4(x, K, K_total, out);
  return 0;
}

int gjxo_run_program(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K,
                     int64_t particle_offset, float* choices, float* score, float* weight,
                     float* logw, const float* logw_in, const float* sub, float* site_scores,
                     float* lse, int64_t K_total) {
  if (!prog || K < 0) return GJX_EINVAL;
  const okey key = {key0, key1};
  const int ns = prog->n_slots;
#pragma omp parallel
  {
    float* vals = (float*)malloc(sizeof(float) * (size_t)(ns > 0 ? ns : 1));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < K; ++i) {
      for (int s = 0; s < ns; ++s) vals[s] = choices[(int64_t)s * K + i];
      float sc, w;
      t_margin = 3.0e38f;
      run_particle(prog, key, (uint64_t)(particle_offset + i), vals, &sc, &w, site_scores ? site_scores + i : NULL, K);
      if (g_margin_buf && i < g_margin_n) g_margin_buf[i] = t_margin;
      for (int s = 0; s < ns; ++s) choices[(int64_t)s * K + i] = vals[s];
      if (score) score[i] = sc;
      if (weight) weight[i] = w;
      if (logw) {
        float lw = w;
        if (logw_in) lw = lw + logw_in[i];
        if (sub) lw = lw - sub[i];
        logw[i] = lw;
      }
    }
    free(vals);
  }
  if (lse && logw) lse4(logw, K, K_total, lse);
  return 0;
}

/* ParticleCollection.sample_particle (smc.py:102-109) */
int gjxo_categorical_pick(const float* logw, int64_t K, int64_t particle_offset, const float* lse,
                          uint32_t key0, uint32_t key1, int32_t rng_mode, float* best_val,
                          int64_t* best_idx) {
  okey key = {key0, key1};
  float bv = -INFINITY;
  int64_t bi = 0;
  for (int64_t i = 0; i < K; ++i) {
    uint64_t gi = (uint64_t)(particle_offset + i);
    float g;
    if (rng_mode == GJX_RNG_JAX32) {
      uint32_t o[2];
      gjxo_threefry2x32(key.a, key.b, (uint32_t)(gi >> 32), (uint32_t)gi, o);
      g = gumbel_from_bits(o[0] ^ o[1]);
    } else {
      uint32_t o[2];
      uint64_t h = gi >> 1; /* FLAT: index pairs share one hash */
      gjxo_threefry2x32(key.a, key.b, (uint32_t)(h >> 32), (uint32_t)h, o);
      g = gumbel_from_bits(o[gi & 1]);
    }
    float v = (logw[i] - lse[2]) + g;
    if (v > bv) { bv = v; bi = particle_offset + i; }
  }
  *best_val = bv;
  *best_idx = bi;
  return 0;
}

/* ---- resampling: exact integer work on fixed-point weights -------------------------------- */
#define GJX_WEIGHT_SCALE 1073741824.0f /* 2^30 */

int gjxo_weight_cumsum(const float* x, int64_t K, int32_t is_log, const float* lse, uint64_t* cum,
                       uint64_t* total) {
  /* two passes over contiguous chunks, one chunk per thread: uint64 adds are associative, so the prefix sums are the
   * same bits for any number of threads (and equal to the device's, whatever its block decomposition) */
  int nt = 1;
#ifdef _OPENMP
  nt = omp_get_max_threads();
#endif
  if (nt > 256) nt = 256;
  if (K < 65536) nt = 1;
  uint64_t part[257];
  const int64_t chunk = (K + nt - 1) / nt;
#pragma omp parallel for schedule(static, 1) num_threads(nt)
  for (int t = 0; t < nt; ++t) {
    const int64_t lo = t * chunk, hi = lo + chunk < K ? lo + chunk : K;
    uint64_t acc = 0;
    for (int64_t i = lo; i < hi; ++i) {
      float w = is_log ? expf(x[i] - lse[0]) : x[i];
      if (!(w > 0.0f)) w = 0.0f;
      acc += (uint64_t)(w * GJX_WEIGHT_SCALE);
      cum[i] = acc;
    }
    part[t + 1] = acc;
  }
  part[0] = 0;
  for (int t = 1; t <= nt; ++t) part[t] += part[t - 1];
#pragma omp parallel for schedule(static, 1) num_threads(nt)
  for (int t = 1; t < nt; ++t) {
    const int64_t lo = t * chunk, hi = lo + chunk < K ? lo + chunk : K;
    const uint64_t off = part[t];
    for (int64_t i = lo; i < hi; ++i) cum[i] += off;
  }
  *total = part[nt];
  return 0;
}

/* Systematic comb: output slot j sits at p_j = (j + u) * total_all / N_total on the weight line;
 * its integer threshold is T_j = floor(p_j) (clamped below total_all) and its ancestor is the
 * first particle whose inclusive prefix sum exceeds T_j.  The double arithmetic (one add, one
 * multiply, one truncation per slot) is IEEE-exact, so a GPU evaluating the same expression gets
 * the same T_j. */
int gjxo_resample_systematic(const uint64_t* cum, int64_t K, uint64_t base, uint64_t total_all,
                             double u, int64_t N_total, int64_t out_begin, int64_t n_out,
                             int32_t* ancestors) {
  const double step = (double)total_all / (double)N_total;
  /* thresholds are non-decreasing in j: every thread walks the prefix sums once over its own stretch of slots,
   * starting from a binary search for its first threshold */
#pragma omp parallel
  {
    int nt = 1, tid = 0;
#ifdef _OPENMP
    nt = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
    const int64_t chunk = (n_out + nt - 1) / nt;
    const int64_t j0 = tid * chunk, j1 = j0 + chunk < n_out ? j0 + chunk : n_out;
    int64_t i = -1;
    for (int64_t j = j0; j < j1; ++j) {
      const double pj = ((double)(out_begin + j) + u) * step;
      uint64_t T = (uint64_t)pj;
      if (total_all > 0 && T > total_all - 1) T = total_all - 1;
      int32_t a = -1;
      if (K > 0 && T >= base && T < base + cum[K - 1]) {
        const uint64_t tl = T - base;
        if (i < 0) { /* first i with cum[i] > tl */
          int64_t lo = 0, hi = K - 1;
          while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cum[mid] > tl) hi = mid; else lo = mid + 1; }
          i = lo;
        }
        while (i < K - 1 && !(cum[i] > tl)) ++i;
        a = (int32_t)i;
      }
      ancestors[j] = a;
    }
  }
  return 0;
}

/* Systematic resampling under the TILE-SCALED fixed point (include/gjx.h GJX_WEIGHTS_TILE_SCALED; device:
 * genjax_amd/csrc/gjx_ssm.hip k_ssm_persistent<TILED> / gjx_resample_indices_tiled).  No counterpart in the reference
 * (its SMC never resamples); the comb is gjxo_resample_systematic's.  Sequential on purpose.
 *   tile b = particles [1024 b, 1024 b + 1024): e_b = ceil(max(log w) * log2 e) in float32, clamped to +-524287,
 *   q_i = floor(2^29 min(1, exp2f(fmaf(log w_i, log2 e, -e_b)))), S_b = sum q_i; a tile with S_b == 0 is dead.
 *   E = max e_b over live tiles; G_b = S_b >> (E - e_b); P = exclusive prefix of G; T_j = comb threshold on total = sum G;
 *   slot j -> tile b with P_b <= T_j < P_{b+1}; ancestor = first i in the tile with cumulative q > (T_j - P_b) << (E - e_b).
 * q_in != NULL: use these quantised weights (the device's, so that everything but the exp2 is compared bit for bit;
 * the exponents are still recomputed from logw).  q_out / e_out (optional) receive the oracle's own q and e_b.
 * All weights dead: identity ancestors (min(j, K - 1)), returns 1. */
#define GJXO_TILE_Q 1024
#define GJXO_TILE_DEAD (-524288)
int gjxo_resample_systematic_tiled(const float* logw, int64_t K, double u, int64_t N, const uint32_t* q_in,
                                   int32_t* ancestors, uint32_t* q_out, int32_t* e_out) {
  const float log2e = 1.44269504f;
  const int64_t nt = (K + GJXO_TILE_Q - 1) / GJXO_TILE_Q;
  uint64_t* S = (uint64_t*)calloc((size_t)nt + 1, sizeof(uint64_t));
  uint64_t* P = (uint64_t*)calloc((size_t)nt + 1, sizeof(uint64_t));
  int32_t* E = (int32_t*)calloc((size_t)nt, sizeof(int32_t));
  uint32_t* q = (uint32_t*)calloc((size_t)K, sizeof(uint32_t));
  int Emax = GJXO_TILE_DEAD;
  for (int64_t b = 0; b < nt; ++b) {
    const int64_t lo = b * GJXO_TILE_Q, hi = lo + GJXO_TILE_Q < K ? lo + GJXO_TILE_Q : K;
    float m = -INFINITY;
    for (int64_t i = lo; i < hi; ++i) if (logw[i] > m) m = logw[i]; /* NaN never wins, as fmaxf on the device */
    int e = GJXO_TILE_DEAD;
    if (m > -INFINITY) {
      const float t = ceilf(m * log2e);
      e = t < -524287.0f ? -524287 : (t > 524287.0f ? 524287 : (int)t);
    }
    uint64_t acc = 0;
    for (int64_t i = lo; i < hi; ++i) {
      uint32_t qi = 0;
      if (q_in) qi = q_in[i];
      else if (e != GJXO_TILE_DEAD) {
        float w = exp2f(fmaf(logw[i], log2e, -(float)e));
        if (!(w > 0.0f)) w = 0.0f;
        if (!(w < 1.0f)) w = 1.0f;
        qi = (uint32_t)(w * 536870912.0f);
      }
      q[i] = qi;
      acc += qi;
    }
    S[b] = acc;
    E[b] = acc ? e : GJXO_TILE_DEAD;
    if (acc && e > Emax) Emax = e;
    if (e_out) e_out[b] = E[b];
  }
  if (q_out) memcpy(q_out, q, (size_t)K * sizeof(uint32_t));
  for (int64_t b = 0; b < nt; ++b) {
    const int sh = Emax - E[b];
    P[b + 1] = P[b] + (sh < 64 ? S[b] >> sh : 0);
  }
  const uint64_t total = P[nt];
  int rc = 0;
  if (total == 0) {
    for (int64_t j = 0; j < N; ++j) ancestors[j] = (int32_t)(j < K ? j : K - 1);
    rc = 1;
  } else {
    const double step = (double)total / (double)N;
    int64_t b = 0;
    for (int64_t j = 0; j < N; ++j) {
      uint64_t T = (uint64_t)(((double)j + u) * step);
      if (T > total - 1) T = total - 1;
      while (!(P[b + 1] > T)) ++b;            /* thresholds are non-decreasing */
      const uint64_t r = (T - P[b]) << (Emax - E[b]);
      const int64_t lo = b * GJXO_TILE_Q, hi = lo + GJXO_TILE_Q < K ? lo + GJXO_TILE_Q : K;
      uint64_t c = 0;
      int64_t i = lo;
      for (; i < hi - 1; ++i) { c += q[i]; if (c > r) break; }
      ancestors[j] = (int32_t)i;
    }
  }
  free(S); free(P); free(E); free(q);
  return rc;
}

/* Multinomial resampling by SORTED uniforms under the tile-scaled fixed point (include/gjx.h: GJX_FILTER_MULTINOMIAL inside the
 * one-launch filter; the cookbook's jax.random.categorical over the log-weights draws N independent ancestors — their sorted order is
 * what a filter may use, the collection is exchangeable).  The j-th smallest of N uniforms = S_j / S_{N+1} with exponential spacings
 * e_i = -log2(u_i) in units of 2^-20, u_i = (2 m_i + 1) / 2^24, m_i = the top 23 bits of word 0 of Threefry(key, (0, i)); every float32
 * step below rounds as on the device (csrc/gjx_device.h exp_spacing). */
static uint64_t exp_spacing(uint32_t word) {
  const uint32_t x = ((word >> 9) << 1) | 1u;
  int k = 31 - __builtin_clz(x);
  float f = ldexpf((float)x, -k);
  if (f > 1.41421354f) { f *= 0.5f; k += 1; }
  const float t = f - 1.0f;
  float q = 0.12614846229553223f;
  q = fmaf(q, t, -0.20742103457450867f);
  q = fmaf(q, t, 0.21566985547542572f);
  q = fmaf(q, t, -0.23892034590244293f);
  q = fmaf(q, t, 0.2879183292388916f);
  q = fmaf(q, t, -0.36070483922958374f);
  q = fmaf(q, t, 0.48091059923171997f);
  q = fmaf(q, t, -0.7213473320007324f);
  q = fmaf(q, t, 1.4426950216293335f);
  const float e = fmaf(-q, t, (float)(24 - k));
  return e > 0.0f ? (uint64_t)(e * 1048576.0f) : 0ull;
}
uint64_t gjxo_exp_spacing(uint32_t word) { return exp_spacing(word); }

int gjxo_resample_sorted_multinomial_tiled(const float* logw, int64_t K, uint32_t key0, uint32_t key1, int64_t N, const uint32_t* q_in,
                                           int32_t* ancestors) {
  const float log2e = 1.44269504f;
  const int64_t nt = (K + GJXO_TILE_Q - 1) / GJXO_TILE_Q;
  uint64_t* S = (uint64_t*)calloc((size_t)nt + 1, sizeof(uint64_t));
  uint64_t* P = (uint64_t*)calloc((size_t)nt + 1, sizeof(uint64_t));
  int32_t* E = (int32_t*)calloc((size_t)nt, sizeof(int32_t));
  uint32_t* q = (uint32_t*)calloc((size_t)K, sizeof(uint32_t));
  int Emax = GJXO_TILE_DEAD;
  for (int64_t b = 0; b < nt; ++b) {
    const int64_t lo = b * GJXO_TILE_Q, hi = lo + GJXO_TILE_Q < K ? lo + GJXO_TILE_Q : K;
    float m = -INFINITY;
    for (int64_t i = lo; i < hi; ++i) if (logw[i] > m) m = logw[i];
    int e = GJXO_TILE_DEAD;
    if (m > -INFINITY) {
      const float t = ceilf(m * log2e);
      e = t < -524287.0f ? -524287 : (t > 524287.0f ? 524287 : (int)t);
    }
    uint64_t acc = 0;
    for (int64_t i = lo; i < hi; ++i) {
      uint32_t qi = 0;
      if (q_in) qi = q_in[i];
      else if (e != GJXO_TILE_DEAD) {
        float w = exp2f(fmaf(logw[i], log2e, -(float)e));
        if (!(w > 0.0f)) w = 0.0f;
        if (!(w < 1.0f)) w = 1.0f;
        qi = (uint32_t)(w * 536870912.0f);
      }
      q[i] = qi;
      acc += qi;
    }
    S[b] = acc;
    E[b] = acc ? e : GJXO_TILE_DEAD;
    if (acc && e > Emax) Emax = e;
  }
  for (int64_t b = 0; b < nt; ++b) {
    const int sh = Emax - E[b];
    P[b + 1] = P[b] + (sh < 64 ? S[b] >> sh : 0);
  }
  const uint64_t total = P[nt];
  int rc = 0;
  if (total == 0) {
    for (int64_t j = 0; j < N; ++j) ancestors[j] = (int32_t)(j < K ? j : K - 1);
    rc = 1;
  } else {
    uint64_t Sall = 0;
    for (int64_t j = 0; j <= N; ++j) {          /* N + 1 spacings: the last one closes the unit interval */
      uint32_t o[2];
      gjxo_threefry2x32(key0, key1, (uint32_t)((uint64_t)j >> 32), (uint32_t)j, o);
      Sall += exp_spacing(o[0]);
    }
    uint64_t Sj = 0;
    int64_t b = 0;
    for (int64_t j = 0; j < N; ++j) {
      uint32_t o[2];
      gjxo_threefry2x32(key0, key1, (uint32_t)((uint64_t)j >> 32), (uint32_t)j, o);
      Sj += exp_spacing(o[0]);
      uint64_t T = (uint64_t)((double)Sj * ((double)total / (double)Sall));
      if (T > total - 1) T = total - 1;
      while (!(P[b + 1] > T)) ++b;
      const uint64_t r = (T - P[b]) << (Emax - E[b]);
      const int64_t lo = b * GJXO_TILE_Q, hi = lo + GJXO_TILE_Q < K ? lo + GJXO_TILE_Q : K;
      uint64_t c = 0;
      int64_t i = lo;
      for (; i < hi - 1; ++i) { c += q[i]; if (c > r) break; }
      ancestors[j] = (int32_t)i;
    }
  }
  free(S); free(P); free(E); free(q);
  return rc;
}

/* gjx_mh_accept (include/gjx.h): the caller-side accept of tests/inference/test_requests.py:131-137.  -> accepted chains; the decision
 * margin |log u - alpha| of every chain goes to the margin buffer (gjxo_set_margin_buffer) */
int64_t gjxo_mh_accept(const float* log_alpha, int64_t K, uint32_t key0, uint32_t key1, float* rows_cur, const float* rows_prop,
                       int64_t row_stride, int32_t rows, float* accepted) {
  int64_t n = 0;
  for (int64_t i = 0; i < K; ++i) {
    uint32_t o[2];
    gjxo_threefry2x32(key0, key1, (uint32_t)((uint64_t)i >> 32), (uint32_t)i, o);
    float lu = logf(uniform_from_bits(o[0] ^ o[1], F32_TINY, 1.0f));
    int acc = lu < log_alpha[i];
    if (g_margin_buf && i < g_margin_n) g_margin_buf[i] = fabsf(lu - log_alpha[i]);
    if (acc)
      for (int r = 0; r < rows; ++r) rows_cur[(int64_t)r * row_stride + i] = rows_prop[(int64_t)r * row_stride + i];
    if (accepted) accepted[i] = (float)acc;
    n += acc;
  }
  return n;
}

int gjxo_resample_multinomial(const uint64_t* cum, int64_t K, uint64_t base, uint64_t total_all,
                              uint32_t key0, uint32_t key1, int64_t N_total, int64_t out_begin,
                              int64_t n_out, int32_t* ancestors) {
  (void)N_total;
  okey key = {key0, key1};
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n_out; ++j) {
    uint64_t gj = (uint64_t)(out_begin + j);
    uint32_t o[2];
    gjxo_threefry2x32(key.a, key.b, (uint32_t)(gj >> 32), (uint32_t)gj, o);
    /* 53-bit uniform from both words: exact integer target on the weight line */
    uint64_t r = (((uint64_t)o[0] << 32) | o[1]) >> 11;
    double uj = (double)r * (1.0 / 9007199254740992.0);
    uint64_t target = (uint64_t)(uj * (double)total_all);
    int32_t a = -1;
    if (target >= base && K > 0 && target < base + cum[K - 1]) {
      uint64_t tl = target - base;
      int64_t lo = 0, hi = K - 1; /* first i with cum[i] > tl */
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (cum[mid] > tl) hi = mid; else lo = mid + 1;
      }
      a = (int32_t)lo;
    }
    ancestors[j] = a;
  }
  return 0;
}

int gjxo_gather_rows(const float* src, int64_t src_stride, const int32_t* anc, int64_t n_out,
                     int32_t rows, float* dst, int64_t dst_stride) {
#pragma omp parallel for schedule(static) collapse(1)
  for (int64_t j = 0; j < n_out; ++j)
    if (anc[j] >= 0)
      for (int r = 0; r < rows; ++r) dst[(int64_t)r * dst_stride + j] = src[(int64_t)r * src_stride + anc[j]];
  return 0;
}

/* ---- linear-Gaussian SSM bootstrap step ---------------------------------------------------- */
int gjxo_ssm_step(int32_t dx, int32_t dy, const float* A, const float* H, float q, float r, float q0,
                  uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t t, int64_t K,
                  int64_t particle_offset, const float* x_prev, int64_t prev_stride,
                  const int32_t* anc, const float* y, float* x_out, float* logw, float* lse,
                  int64_t K_total) {
  okey key = {key0, key1};
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < K; ++i) {
    const ostream st = stream_open(rng_mode, key, (uint64_t)(particle_offset + i), 1u); /* site 1: the latent state; site 2 (y) is observed */
    const ostream* sk = &st;
    float xp[64], xn[64];
    if (t > 0) {
      int64_t a = anc ? anc[i] : i;
      for (int d = 0; d < dx; ++d) xp[d] = x_prev[(int64_t)d * prev_stride + a];
    }
    for (int d = 0; d < dx; ++d) {
      float mu = 0.0f, sd = q0;
      if (t > 0) {
        float acc = 0.0f;
        for (int e = 0; e < dx; ++e) acc += A[d * dx + e] * xp[e];
        mu = acc;
        sd = q;
      }
      xn[d] = mu + sd * stream_normal(sk, (uint32_t)d);
      x_out[(int64_t)d * K + i] = xn[d];
    }
    float lw = 0.0f;
    for (int o = 0; o < dy; ++o) {
      float m;
      if (H) {
        float acc = 0.0f;
        for (int e = 0; e < dx; ++e) acc += H[o * dx + e] * xn[e];
        m = acc;
      } else {
        m = xn[o];
      }
      float z = y[o] / r - m / r;
      lw += -0.5f * z * z - (HALF_LOG_2PI + logf(r));
    }
    logw[i] = lw;
  }
  if (lse) lse4(logw, K, K_total, lse);
  return 0;
}

/* Resample-move variant: before x_{t-1} (gathered through its ancestor) is propagated it takes
 * n_moves random-walk Metropolis steps whose invariant density is p(x_{t-1} | parent, y_{t-1}) ~ N(x; m_prev, sd^2)
 * N(y_{t-1}; H x, r^2) — the reference's Regenerate / Rejuvenate ingredients with the caller-side accept of
 * tests/inference/test_requests.py:131-137 (log u < w).  Draws come from site 2 of the step's stream: element
 * n (dx + 2) + d for the proposal of move n, n (dx + 2) + dx for its uniform. */
int gjxo_ssm_step_move(int32_t dx, int32_t dy, const float* A, const float* H, float q, float r, float q0,
                       uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t t, int64_t K, int64_t particle_offset,
                       const float* x_prev, const float* m_prev, int64_t prev_stride, const int32_t* anc,
                       const float* y_prev, const float* y, int32_t n_moves, float move_scale, float* x_out,
                       float* m_out, float* logw, float* accepted, float* lse, int64_t K_total) {
  okey key = {key0, key1};
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < K; ++i) {
    const uint64_t gi = (uint64_t)(particle_offset + i);
    const ostream st = stream_open(rng_mode, key, gi, 1u);
    float xp[64], xn[64], mp[64], xq[64];
    float nacc = 0.0f;
    t_margin = 3.0e38f;
    if (t > 0) {
      int64_t a = anc ? anc[i] : i;
      for (int d = 0; d < dx; ++d) xp[d] = x_prev[(int64_t)d * prev_stride + a];
      for (int d = 0; d < dx; ++d) mp[d] = (t > 1 && m_prev) ? m_prev[(int64_t)d * prev_stride + a] : 0.0f;
      const float sdp = t > 1 ? q : q0;
      const ostream sm = stream_open(rng_mode, key, gi, 2u);
      float cur = 0.0f;
      for (int pass = 0; pass <= n_moves; ++pass) { /* pass 0 scores the current state */
        const float* x = pass == 0 ? xp : xq;
        if (pass > 0) for (int d = 0; d < dx; ++d) xq[d] = xp[d] + move_scale * stream_normal(&sm, (uint32_t)((pass - 1) * (dx + 2) + d));
        float s2 = 0.0f;
        for (int d = 0; d < dx; ++d) { float z = (x[d] - mp[d]) / sdp; s2 += z * z; }
        for (int o = 0; o < dy; ++o) {
          float m;
          if (H) { float acc = 0.0f; for (int e = 0; e < dx; ++e) acc += H[o * dx + e] * x[e]; m = acc; } else m = x[o];
          float z = (y_prev[o] - m) / r;
          s2 += z * z;
        }
        const float lp = -0.5f * s2;
        if (pass == 0) { cur = lp; continue; }
        const float lu = logf(uniform_from_bits(elem_bits(&sm, (uint32_t)((pass - 1) * (dx + 2) + dx)), F32_TINY, 1.0f));
        decide(lu, lp - cur);
        if (lu < lp - cur) { for (int d = 0; d < dx; ++d) xp[d] = xq[d]; cur = lp; nacc += 1.0f; }
      }
    }
    for (int d = 0; d < dx; ++d) {
      float mu = 0.0f, sd = q0;
      if (t > 0) {
        float acc = 0.0f;
        for (int e = 0; e < dx; ++e) acc += A[d * dx + e] * xp[e];
        mu = acc;
        sd = q;
      }
      m_out[(int64_t)d * K + i] = mu;
      xn[d] = mu + sd * stream_normal(&st, (uint32_t)d);
      x_out[(int64_t)d * K + i] = xn[d];
    }
    float lw = 0.0f;
    for (int o = 0; o < dy; ++o) {
      float m;
      if (H) { float acc = 0.0f; for (int e = 0; e < dx; ++e) acc += H[o * dx + e] * xn[e]; m = acc; } else m = xn[o];
      float z = y[o] / r - m / r;
      lw += -0.5f * z * z - (HALF_LOG_2PI + logf(r));
    }
    logw[i] = lw;
    if (accepted) accepted[i] = nacc;
    if (g_margin_buf && i < g_margin_n) g_margin_buf[i] = t_margin;
  }
  if (lse) lse4(logw, K, K_total, lse);
  return 0;
}

/* ---- HMC.edit (hmc.py:156-211) ---------------------------------------------------------------
 * score and its gradient w.r.t. the selected slots by a reverse sweep over the site list
 * (what jax.grad of gen_fn.assess computes, hmc.py:83-94).  All sites are evaluated at their
 * current value (assess semantics: every site constrained, static.py:297-321). */
/* digamma in double precision: recurrence to x >= 10, then the asymptotic series */
static double digamma_d(double x) {
  double acc = 0.0;
  while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
  const double r = 1.0 / x, r2 = r * r;
  return acc + log(x) - 0.5 * r - r2 * (1.0 / 12 - r2 * (1.0 / 120 - r2 * (1.0 / 252 - r2 * (1.0 / 240 - r2 / 132))));
}

static void dlogpdf(int kind, float x, float a, float b, float* dx, float* da, float* db) {
  switch (kind) {
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: {
      float z = (x - a) / b;
      *dx = -z / b; *da = z / b; *db = (z * z - 1.0f) / b;
      return;
    }
    case GJX_BERNOULLI_LOGITS: *dx = 0.0f; *da = x - sigmoidf_(a); *db = 0.0f; return;
    case GJX_FLIP: *dx = 0.0f; *da = (x != 0.0f ? 1.0f / a : 0.0f) - (x != 1.0f ? (1.0f - x) / (1.0f - a) : 0.0f); *db = 0.0f; return;
    case GJX_HALF_NORMAL: { float z = x / a; *dx = -z / a; *da = (z * z - 1.0f) / a; *db = 0.0f; return; }
    case GJX_EXPONENTIAL: *dx = -a; *da = 1.0f / a - x; *db = 0.0f; return;
    case GJX_LAPLACE: { float s = (x > a) - (x < a); *dx = -s / b; *da = s / b; *db = fabsf(x - a) / (b * b) - 1.0f / b; return; }
    case GJX_CAUCHY: { float z = (x - a) / b; float g = 2.0f * z / (1.0f + z * z); *dx = -g / b; *da = g / b; *db = (g * z - 1.0f) / b; return; }
    case GJX_LOG_NORMAL: { float lx = logf(x); float z = (lx - a) / b; *dx = (-z / b - 1.0f) / x; *da = z / b; *db = (z * z - 1.0f) / b; return; }
    case GJX_BETA:
      *dx = (a - 1.0f) / x - (b - 1.0f) / (1.0f - x);
      *da = (float)(log((double)x) - digamma_d(a) + digamma_d((double)a + b));
      *db = (float)(log1p(-(double)x) - digamma_d(b) + digamma_d((double)a + b));
      return;
    case GJX_GAMMA: *dx = (a - 1.0f) / x - b; *da = (float)(log((double)b) + log((double)x) - digamma_d(a)); *db = a / b - x; return;
    case GJX_UNIFORM: *dx = 0.0f; *da = 1.0f / (b - a); *db = -1.0f / (b - a); return;
    default: *dx = *da = *db = 0.0f;
  }
}

/* gradients of the four-parameter forms; g[0..3] = d/d(a,b,c,d) */
static void dlogpdf4(int kind, float x, float a, float b, float c, float d, float* dx, float* g) {
  g[0] = g[1] = g[2] = g[3] = 0.0f;
  *dx = 0.0f;
  switch (kind) {
    case GJX_TRUNCATED_CAUCHY: {
      double z = ((double)x - a) / b, lo = ((double)c - a) / b, hi = ((double)d - a) / b;
      double rA = 1.0 / ((atan(hi) - atan(lo)) * b), wl = rA / (1.0 + lo * lo), wh = rA / (1.0 + hi * hi), w = 2.0 * z / ((1.0 + z * z) * b);
      *dx = (float)-w; g[0] = (float)(w + (wh - wl)); g[1] = (float)(w * z - 1.0 / b + (hi * wh - lo * wl)); g[2] = (float)wl; g[3] = (float)-wh;
      return;
    }
    case GJX_NEGATIVE_BINOMIAL: { double p = 1.0 / (1.0 + exp(-(double)b)); g[0] = (float)(digamma_d((double)x + a) - digamma_d(a) - softplus_d(b)); g[1] = (float)(x * (1.0 - p) - a * p); return; }
    case GJX_VON_MISES: { double sn = sin((double)x - a); *dx = (float)(-b * sn); g[0] = (float)(b * sn); g[1] = (float)(cos((double)x - a) - bessel_i1_over_i0_d(b)); return; }
    case GJX_CHI: *dx = (a - 1.0f) / x - x; g[0] = (float)(log((double)x) - 0.5 * 0.6931471805599453 - 0.5 * digamma_d(0.5 * a)); return;
    case GJX_EXP_GAMMA: { double e = exp((double)x); *dx = (float)(a - b * e); g[0] = (float)(log((double)b) + x - digamma_d(a)); g[1] = (float)(a / b - e); return; }
    case GJX_EXP_INVERSE_GAMMA: { double e = exp(-(double)x); *dx = (float)(b * e - a); g[0] = (float)(log((double)b) - x - digamma_d(a)); g[1] = (float)(a / b - e); return; }
    case GJX_KUMARASWAMY: {
      double lx = log((double)x), xa = exp(a * lx), r = xa / (1.0 - xa);
      *dx = (float)(((a - 1.0) - (b - 1.0) * a * r) / x); g[0] = (float)(1.0 / a + lx * (1.0 - (b - 1.0) * r)); g[1] = (float)(1.0 / b + log1p(-xa));
      return;
    }
    case GJX_MOYAL: { double z = ((double)x - a) / b, e1 = 0.5 * (1.0 - exp(-z)); *dx = (float)(-e1 / b); g[0] = (float)(e1 / b); g[1] = (float)((e1 * z - 1.0) / b); return; }
    case GJX_DOUBLESIDED_MAXWELL: { double z = ((double)x - a) / b, w = (2.0 / z - z) / b; *dx = (float)w; g[0] = (float)-w; g[1] = (float)((z * z - 3.0) / b); return; }
    case GJX_INVERSE_GAUSSIAN: {
      double r = ((double)x - a) / a;
      *dx = (float)(-1.5 / x - 0.5 * b * ((double)x * x - (double)a * a) / ((double)a * a * (double)x * x));
      g[0] = (float)(b * r / ((double)a * a)); g[1] = (float)(0.5 / b - 0.5 * r * r / x);
      return;
    }
    case GJX_HALF_STUDENT_T:
    case GJX_STUDENT_T: {
      float y = (x - b) / c;
      float w = (a + 1.0f) * y / (a + y * y);
      *dx = -w / c; g[1] = w / c; g[2] = (w * y - 1.0f) / c;
      g[0] = (float)(-0.5 * log1p((double)y * y / a) + 0.5 * (a + 1.0) * y * y / (a * (a + (double)y * y)) - 0.5 / a +
                     0.5 * (digamma_d(0.5 * (a + 1.0)) - digamma_d(0.5 * a)));
      return;
    }
    case GJX_TRUNCATED_NORMAL: {
      float z = (x - a) / b, lo = (c - a) / b, hi = (d - a) / b;
      float Z = normal_interval_mass(lo, hi);
      float plo = 0.39894228f * expf(-0.5f * lo * lo) / (Z * b), phi = 0.39894228f * expf(-0.5f * hi * hi) / (Z * b);
      *dx = -z / b; g[0] = z / b + (phi - plo); g[1] = (z * z - 1.0f) / b + (hi * phi - lo * plo); g[2] = plo; g[3] = -phi;
      return;
    }
    case GJX_POISSON: g[0] = x / a - 1.0f; return;
    case GJX_GEOMETRIC: g[0] = 1.0f / a - x / (1.0f - a); return;
    case GJX_GUMBEL: { float z = (x - a) / b; float e1 = 1.0f - expf(-z); *dx = -e1 / b; g[0] = e1 / b; g[1] = (e1 * z - 1.0f) / b; return; }
    case GJX_HALF_CAUCHY: { float z = (x - a) / b; float q = 2.0f * z / (1.0f + z * z); *dx = -q / b; g[0] = q / b; g[1] = (q * z - 1.0f) / b; return; }
    case GJX_INVERSE_GAMMA: *dx = -(a + 1.0f) / x + b / (x * x); g[1] = a / b - 1.0f / x; g[0] = (float)(log((double)b) - digamma_d(a) - log((double)x)); return;
    case GJX_WEIBULL: { float lr = logf(x / b); float t = expf(a * lr); *dx = ((a - 1.0f) - a * t) / x; g[1] = a * (t - 1.0f) / b; g[0] = 1.0f / a + lr * (1.0f - t); return; }
    case GJX_LOGIT_NORMAL: { float z = (logf(x) - log1pf(-x) - a) / b; *dx = -z / b / (x * (1.0f - x)) - 1.0f / x + 1.0f / (1.0f - x); g[0] = z / b; g[1] = (z * z - 1.0f) / b; return; }
    case GJX_CHI2: *dx = (0.5f * a - 1.0f) / x - 0.5f; g[0] = (float)(0.5 * (log((double)x) - 0.6931471805599453 - digamma_d(0.5 * a))); return;
    default: return;
  }
}

static float xf_deriv(int xf, float pre) { /* d xf(v) / d v at pre-transform value */
  switch (xf) {
    case GJX_XF_EXP: return expf(pre);
    case GJX_XF_SOFTPLUS: return sigmoidf_(pre);
    case GJX_XF_SIGMOID: { float s = sigmoidf_(pre); return s * (1.0f - s); }
    default: return 1.0f;
  }
}
static float eval_param_pre(const gjx_param* p, int d, const float* tab, const float* vals) {
  gjx_param q = *p;
  q.xf = GJX_XF_NONE;
  return eval_param(&q, d, tab, vals);
}
static void param_backprop(const gjx_param* p, int d, float g, const float* tab, const float* vals,
                           float* grad) {
  if (g == 0.0f) return;
  if (p->xf != GJX_XF_NONE) g *= xf_deriv(p->xf, eval_param_pre(p, d, tab, vals));
  switch (p->op) {
    case GJX_P_VALUE: grad[p->slot + (d % p->len)] += g; break;
    case GJX_P_AFFINE:
      for (int e = 0; e < p->n; ++e) grad[p->slot + e] += g * tab[p->moff + d * p->n + e];
      break;
    case GJX_P_VGATHER: grad[vgather_row(p, d, tab, vals)] += g; break;
    case GJX_P_EXPR: expr_backward(p, d, g, tab, vals, grad); break;
    default: break; /* CONST, GATHER: no float dependence */
  }
}

/* the contribution of ONE site (a plain site, or one instance of a plate's body site with its offsets applied) to the score and to
 * the gradient rows */
static float score_and_grad_site(const gjx_program* prog, const gjx_site* s, const float* vals, float* grad) {
  const float* tab = prog->tab;
  float score = 0.0f;
  {
    if (s->kind == GJX_CATEGORICAL_LOGITS || s->kind == GJX_CATEGORICAL_PROBS) {
      int n = s->ncat;
      float mx = -INFINITY;
      for (int c = 0; c < n; ++c) {
        float l = eval_param(&s->p[0], c, tab, vals);
        if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
        if (l > mx) mx = l;
      }
      double se = 0.0;
      for (int c = 0; c < n; ++c) {
        float l = eval_param(&s->p[0], c, tab, vals);
        if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
        se += exp((double)l - (double)mx);
      }
      int k = (int)(s->slot >= 0 ? vals[s->slot] : tab[s->obs_off]);
      if (k < 0) k = 0;
      if (k > n - 1) k = n - 1;
      float l = eval_param(&s->p[0], k, tab, vals);
      if (s->kind == GJX_CATEGORICAL_PROBS) l = logf(l);
      score += l - (mx + (float)log(se));
      return score; /* integer site: no gradient through it (hmc.py:49-65) */
    }
    if (s->kind == GJX_DIRICHLET) { /* scored, never differentiated (simplex-constrained value) */
      float sa = 0.0f;
      for (int d = 0; d < s->dim; ++d) {
        float al = eval_param(&s->p[0], d, tab, vals);
        float x = s->slot >= 0 ? vals[s->slot + d] : tab[s->obs_off + d];
        sa += al;
        score += xlogyf(al - 1.0f, x) - lgammaf(al);
      }
      score += lgammaf(sa);
      return score;
    }
    for (int d = 0; d < s->dim; ++d) {
      float a = eval_param(&s->p[0], d, tab, vals);
      float b = eval_param(&s->p[1], d, tab, vals);
      float x = s->slot >= 0 ? vals[s->slot + d] : tab[s->obs_off + d];
      if (s->kind >= GJX_STUDENT_T) {
        int np = params_of(s->kind);
        float c = np > 2 ? eval_param(&s->p[2], d, tab, vals) : 0.0f;
        float e = np > 3 ? eval_param(&s->p[3], d, tab, vals) : 0.0f;
        float gx, g4[4];
        score += elem_logpdf4(s->kind, x, a, b, c, e);
        dlogpdf4(s->kind, x, a, b, c, e, &gx, g4);
        if (s->slot >= 0) grad[s->slot + d] += gx;
        for (int q = 0; q < np; ++q) param_backprop(&s->p[q], d, g4[q], tab, vals, grad);
        continue;
      }
      score += elem_logpdf(s->kind, x, a, b);
      float gx, ga, gb;
      dlogpdf(s->kind, x, a, b, &gx, &ga, &gb);
      if (s->slot >= 0) grad[s->slot + d] += gx;
      param_backprop(&s->p[0], d, ga, tab, vals, grad);
      param_backprop(&s->p[1], d, gb, tab, vals, grad);
    }
  }
  return score;
}

/* instance `inst` of a plate's body site: the site with its per-instance offsets applied (gjx.h "Plates") */
static gjx_site site_instance(const gjx_site* s0, int inst) {
  gjx_site sv = *s0;
  if (s0->plate && inst) {
    const int w = (sv.kind == GJX_CATEGORICAL_LOGITS || sv.kind == GJX_CATEGORICAL_PROBS) ? 1 : sv.dim;
    if (sv.slot >= 0) sv.slot += inst * w;
    sv.obs_off += inst * sv.d_obs;
    for (int k = 0; k < GJX_MAX_PARAMS; ++k) {
      sv.p[k].off += inst * sv.p[k].d_off; sv.p[k].moff += inst * sv.p[k].d_moff;
      if (sv.p[k].slot >= 0) sv.p[k].slot += inst * sv.p[k].d_slot;
      if (sv.p[k].op == GJX_P_EXPR) { sv.p[k].off = s0->p[k].off; sv.p[k].pad_[0] = inst; }   /* the block's VALUE / CONST nodes carry their own strides (expr_forward) */
    }
  }
  return sv;
}

/* score + gradient for one chain; grad[n_slots] (all slots; caller masks by selection).  The gradient of assess through a Vmap
 * (hmc.py:70-96 differentiates any assess; vmap.py:363-376): a plate's body is walked instance by instance, every instance adding
 * to the rows of what it reads — the instance's own rows, rows outside the plate */
static float score_and_grad(const gjx_program* prog, const float* vals, float* grad) {
  float score = 0.0f;
  for (int s = 0; s < prog->n_slots; ++s) grad[s] = 0.0f;
  for (int j = 0; j < prog->n_sites;) {
    const gjx_site* s = &prog->sites[j];
    if (s->mode == GJX_MODE_INPUT) { ++j; continue; } /* an argument / a carry: a value that is there, no density (gjx.h) */
    if (s->plate == 0) { score += score_and_grad_site(prog, s, vals, grad); ++j; continue; }
    int m = 1;
    while (j + m < prog->n_sites && prog->sites[j + m].plate == s->plate) ++m;
    for (int i = 0; i < s->plate_n; ++i)
      for (int l = 0; l < m; ++l) {
        if (prog->sites[j + l].mode == GJX_MODE_INPUT) continue;
        const gjx_site sv = site_instance(&prog->sites[j + l], i);
        score += score_and_grad_site(prog, &sv, vals, grad);
      }
    j += m;
  }
  return score;
}

int gjxo_score_grad(const gjx_program* prog, int64_t n, const float* choices, float* score,
                    float* grad) {
  const int ns = prog->n_slots;
  char* sel = (char*)calloc((size_t)ns + 1, 1);
  for (int j = 0; j < prog->n_sites; ++j)
    if ((prog->sites[j].flags & GJX_SITE_HMC_SELECTED) && prog->sites[j].slot >= 0) {
      const int rows_ = prog->sites[j].dim * (prog->sites[j].plate ? prog->sites[j].plate_n : 1);   /* a plate's body site: every instance */
      for (int d = 0; d < rows_; ++d) sel[prog->sites[j].slot + d] = 1;
    }
#pragma omp parallel
  {
    float* vals = (float*)malloc(sizeof(float) * (size_t)ns);
    float* g = (float*)malloc(sizeof(float) * (size_t)ns);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      for (int s = 0; s < ns; ++s) vals[s] = choices[(int64_t)s * n + i];
      float sc = score_and_grad(prog, vals, g);
      if (score) score[i] = sc;
      for (int s = 0; s < ns; ++s) grad[(int64_t)s * n + i] = sel[s] ? g[s] : 0.0f;
    }
    free(vals);
    free(g);
  }
  free(sel);
  return 0;
}

int gjxo_hmc(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t n, int64_t chain_offset,
             float eps, int32_t L, int32_t stale_grad_compat, int32_t accept, float* choices,
             float* score, float* alpha, float* accepted) {
  const int ns = prog->n_slots;
  const okey key = {key0, key1};
  int* selslot = (int*)malloc(sizeof(int) * (size_t)(ns + 1));
  int* leaf_of = (int*)malloc(sizeof(int) * (size_t)(ns + 1));
  int* elem_of = (int*)malloc(sizeof(int) * (size_t)(ns + 1));
  int nsel = 0, leaf = 0;
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site* s = &prog->sites[j];
    if (!(s->flags & GJX_SITE_HMC_SELECTED) || s->slot < 0) continue;
    /* a selected body site of a plate is ONE leaf (the reference's leaf is the whole vmapped array): elements instance-major */
    const int rows_ = s->dim * (s->plate ? s->plate_n : 1);
    for (int d = 0; d < rows_; ++d) {
      selslot[nsel] = s->slot + d;
      leaf_of[nsel] = leaf; /* one momentum leaf per selected address, in program order */
      elem_of[nsel] = d;
      ++nsel;
    }
    ++leaf;
  }
#pragma omp parallel
  {
    float* vals = (float*)malloc(sizeof(float) * (size_t)ns);
    float* old = (float*)malloc(sizeof(float) * (size_t)ns);
    float* g = (float*)malloc(sizeof(float) * (size_t)ns);
    float* g0 = (float*)malloc(sizeof(float) * (size_t)ns);
    float* p = (float*)malloc(sizeof(float) * (size_t)(nsel + 1));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const uint64_t gidx = (uint64_t)(chain_offset + i);
      okey ck = fold_in64(key, gidx);
      okey knew = fold_in(ck, 0u), sub = fold_in(ck, 1u); /* key, sub_key = split(key)  hmc.py:167 */
      for (int s = 0; s < ns; ++s) old[s] = vals[s] = choices[(int64_t)s * n + i];
      float score0 = score_and_grad(prog, vals, g0); /* hmc.py:165-166 */
      float k0 = 0.0f;
      for (int m = 0; m < nsel; ++m) { /* sample_momenta hmc.py:120-130 */
        const ostream ms = prog->rng_mode == GJX_RNG_JAX32 ? stream_from_site_key(fold_in(sub, (uint32_t)leaf_of[m]))
                                                           : stream_open(GJX_RNG_FLAT, key, gidx, (uint32_t)leaf_of[m] + 1u);
        p[m] = stream_normal(&ms, (uint32_t)elem_of[m]);
        k0 += -0.5f * p[m] * p[m] - HALF_LOG_2PI;
        if (g_momenta_buf && i < g_momenta_n) g_momenta_buf[(int64_t)m * g_momenta_n + i] = p[m];
      }
      for (int s = 0; s < ns; ++s) g[s] = g0[s];
      float sc = score0;
      for (int t = 1; t <= L; ++t) { /* hmc.py:170-194 */
        const float* gfirst = stale_grad_compat ? g0 : g; /* hmc.py:186 carries the received gradient */
        for (int m = 0; m < nsel; ++m) p[m] = p[m] + (eps / 2.0f) * gfirst[selslot[m]];
        for (int m = 0; m < nsel; ++m) vals[selslot[m]] = vals[selslot[m]] + eps * p[m];
        sc = score_and_grad(prog, vals, g);
        for (int m = 0; m < nsel; ++m) p[m] = p[m] + (eps / 2.0f) * g[selslot[m]];
      }
      float k1 = 0.0f;
      for (int m = 0; m < nsel; ++m) { float q = -1.0f * p[m]; k1 += -0.5f * q * q - HALF_LOG_2PI; }
      float al = sc - score0 + k1 - k0; /* hmc.py:196-203 */
      int acc = 1;
      if (accept) {
        const ostream as = prog->rng_mode == GJX_RNG_JAX32 ? stream_from_site_key(fold_in(knew, 0x4d48u))
                                                           : stream_open(GJX_RNG_FLAT, key, gidx, GJX_FLAT_MAX_SITES);
        float lu = logf(bits_to_unit(elem_bits(&as, 0u)));
        acc = lu < al; /* tests/inference/test_requests.py:134-137 */
        /* decision margin of the accept (gjxo_set_margin_buffer): |log u - alpha|, absolute — alpha is a difference of
         * scores that the device's trajectory reproduces to an absolute tolerance, not a relative one */
        if (g_margin_buf && i < g_margin_n) g_margin_buf[i] = fabsf(lu - al);
      }
      if (!acc) { for (int s = 0; s < ns; ++s) vals[s] = old[s]; sc = score0; }
      for (int s = 0; s < ns; ++s) choices[(int64_t)s * n + i] = vals[s];
      if (score) score[i] = sc;
      if (alpha) alpha[i] = al;
      if (accepted) accepted[i] = (float)acc;
    }
    free(vals); free(old); free(g); free(g0); free(p);
  }
  free(selslot); free(leaf_of); free(elem_of);
  return 0;
}

int gjxo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void gjxo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* the random stream of (particle gidx, site) under a key, for restatements written above this file (oracle/moves.py: the resample-move
 * of the generic filter): n standard normals from elements e0, e0 + 1, ... (stream_normal: FLAT pairs by Box-Muller, JAX32 by erfinv)
 * or n raw 32-bit element words */
int gjxo_stream_normals(int32_t rng_mode, uint32_t key0, uint32_t key1, uint64_t gidx, uint32_t site, uint32_t e0, int32_t n, float* out) {
  const okey key = {key0, key1};
  const ostream st = stream_open(rng_mode, key, gidx, site);
  for (int32_t k = 0; k < n; ++k) out[k] = stream_normal(&st, e0 + (uint32_t)k);
  return 0;
}
int gjxo_stream_bits(int32_t rng_mode, uint32_t key0, uint32_t key1, uint64_t gidx, uint32_t site, uint32_t e0, int32_t n, uint32_t* out) {
  const okey key = {key0, key1};
  const ostream st = stream_open(rng_mode, key, gidx, site);
  for (int32_t k = 0; k < n; ++k) out[k] = elem_bits(&st, e0 + (uint32_t)k);
  return 0;
}
