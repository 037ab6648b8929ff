// gjx_device.h — device-side building blocks for the gfx950 kernels: Threefry-2x32, bits ->
// uniform/normal/Gumbel, the primitive samplers and log-densities, parameter evaluation.
//
// Written for CDNA4 (wave64, no dual issue): integer hash work is plain VALU
// (v_add_u32 / v_alignbit_b32 / v_xor_b32 — 3 ops per Threefry round), transcendentals use the
// hardware v_log_f32 / v_exp_f32 / v_rcp_f32 / v_sqrt_f32 approximations (≈1 ulp), which is why
// float parity with the CPU oracle is a tolerance, while every integer (hash words, indices,
// fixed-point weights) is bit-exact.
#pragma once

#ifndef __HIPCC_RTC__   /* hipRTC (generated kernels, gjx_codegen.hip) brings its own runtime header and integer types */
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#ifndef INFINITY
#define INFINITY __builtin_huge_valf()
#endif

#include "../../include/gjx.h"

#define GJX_DEV __device__ __forceinline__

namespace gjx {

struct key2 {
  uint32_t a, b;
};

GJX_DEV uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, r); }

// Threefry-2x32, 20 rounds.  Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (2011);
// the generator behind jax.random (jax/_src/prng.py threefry2x32).
GJX_DEV key2 threefry2x32(key2 k, uint32_t c0, uint32_t c1) {
  const uint32_t ks0 = k.a, ks1 = k.b, ks2 = k.a ^ k.b ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define GJX_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  GJX_R(13) GJX_R(15) GJX_R(26) GJX_R(6)
  x0 += ks1; x1 += ks2 + 1u;
  GJX_R(17) GJX_R(29) GJX_R(16) GJX_R(24)
  x0 += ks2; x1 += ks0 + 2u;
  GJX_R(13) GJX_R(15) GJX_R(26) GJX_R(6)
  x0 += ks0; x1 += ks1 + 3u;
  GJX_R(17) GJX_R(29) GJX_R(16) GJX_R(24)
  x0 += ks1; x1 += ks2 + 4u;
  GJX_R(13) GJX_R(15) GJX_R(26) GJX_R(6)
  x0 += ks2; x1 += ks0 + 5u;
#undef GJX_R
  return key2{x0, x1};
}

// jax.random.fold_in(k, i) == jax.random.split(k, n)[i] with jax_threefry_partitionable=True
GJX_DEV key2 fold_in(key2 k, uint32_t i) { return threefry2x32(k, 0u, i); }
GJX_DEV key2 fold_in64(key2 k, uint64_t i) { return threefry2x32(k, (uint32_t)(i >> 32), (uint32_t)i); }

// Element-bit source for one (particle, site) stream; layouts documented at GJX_RNG_* in gjx.h.
template <int RNG>
struct BitStream;

// FLAT: the (particle, site) stream is the concatenation of the 64-bit blocks Threefry(key, (i, site<<22 | h)),
// h = 0, 1, ..., read as 32-bit words; element c is the 32-bit window starting at stream bit 23*c (consumers use
// its top 23 bits), so consecutive draws use consecutive disjoint 23-bit fields: 2.78 draws per hash.
// The two most recent blocks are cached (slot = h & 1), which makes a sequential reader compute every hash once.
template <>
struct BitStream<GJX_RNG_FLAT> {
  key2 key, blk0, blk1;
  uint32_t c0, site_hi, h0, h1;   // h0 / h1: block index held in blk0 (even h) / blk1 (odd h), ~0u when empty
  float n0, n1;         // Box-Muller pair of elements (2*npair, 2*npair+1) (stream_normal): the odd element reuses the even one's work
  uint32_t npair;
  GJX_DEV BitStream() : key{0u, 0u}, blk0{0u, 0u}, blk1{0u, 0u}, c0(0u), site_hi(0u), h0(0xFFFFFFFFu), h1(0xFFFFFFFFu), n0(0.0f), n1(0.0f), npair(0xFFFFFFFFu) {}
  GJX_DEV void open(key2 run_key, uint64_t idx, uint32_t site) {
    key = (idx >> 32) ? threefry2x32(run_key, 0xFFFFFFFFu, (uint32_t)(idx >> 32)) : run_key;
    c0 = (uint32_t)idx;
    site_hi = site << GJX_FLAT_SITE_SHIFT;
    h0 = h1 = 0xFFFFFFFFu;
    npair = 0xFFFFFFFFu;
  }
  GJX_DEV void open_site_key(key2) {}
  // the same with the index in halves: `hi` wave-uniform (the launcher keeps a launch inside one 2^32 range of indices), so the
  // stream key and its key schedule stay in scalar registers
  GJX_DEV void open_hi(key2 run_key, uint32_t hi, uint32_t lo, uint32_t site) {
    key = hi ? threefry2x32(run_key, 0xFFFFFFFFu, hi) : run_key;
    c0 = lo;
    site_hi = site << GJX_FLAT_SITE_SHIFT;
    h0 = h1 = 0xFFFFFFFFu;
    npair = 0xFFFFFFFFu;
  }
  GJX_DEV uint32_t word(uint32_t n) {
    const uint32_t h = n >> 1;
    if (h & 1u) {
      if (h1 != h) { blk1 = threefry2x32(key, c0, site_hi | h); h1 = h; }
      return (n & 1u) ? blk1.b : blk1.a;
    }
    if (h0 != h) { blk0 = threefry2x32(key, c0, site_hi | h); h0 = h; }
    return (n & 1u) ? blk0.b : blk0.a;
  }
  GJX_DEV uint32_t get(uint32_t c) {
    const uint32_t bit = 23u * c, n = bit >> 5, sh = bit & 31u;
    const uint32_t lo = word(n);
    if (sh == 0u) return lo;
    return __builtin_amdgcn_alignbit(word(n + 1u), lo, sh);
  }
};

// 32-bit window of FLAT element c in the word array w[] (word 2h = x0, word 2h+1 = x1 of block h); for fused kernels
// whose c is a constant after unrolling
#define GJX_FIELD(w, c) ((((23 * (c)) & 31) == 0) ? (w)[(23 * (c)) >> 5] : __builtin_amdgcn_alignbit((w)[((23 * (c)) >> 5) + 1], (w)[(23 * (c)) >> 5], (23 * (c)) & 31))
// number of 64-bit blocks the first n elements of a FLAT stream touch
#define GJX_FLAT_BLOCKS(n) ((9 + 23 * (n) + 63) / 64)

template <>
struct BitStream<GJX_RNG_JAX32> {
  key2 sk;
  GJX_DEV BitStream() : sk{0u, 0u} {}
  GJX_DEV void open(key2 run_key, uint64_t idx, uint32_t site) { sk = fold_in(fold_in64(run_key, idx), site); }
  GJX_DEV void open_hi(key2 run_key, uint32_t hi, uint32_t lo, uint32_t site) { open(run_key, ((uint64_t)hi << 32) | lo, site); }
  GJX_DEV void open_site_key(key2 k) { sk = k; }
  GJX_DEV uint32_t get(uint32_t c) {
    const key2 h = threefry2x32(sk, 0u, c);
    return h.a ^ h.b;
  }
};

// Which key and which site number the FLAT stream of a site uses (gjx.h "Scan steps"): walked in site order.
struct SiteStreamWalk {
  key2 run_key, key;
  int32_t tag;       // scan tag of the previous site
  uint32_t local;    // sites seen in the current step
  uint32_t plain;    // non-Scan sites seen
  uint32_t run_head, run_next;   // open scalar-normal run (gjx.h "Scalar-normal runs"): its head's site number (0 = none), next element
  GJX_DEV explicit SiteStreamWalk(key2 k) : run_key(k), key(k), tag(0), local(0u), plain(0u), run_head(0u), run_next(0u) {}
  // Scalar-normal runs: call after next() with the site's number.  `joins`: the site is a sampled scalar normal
  // (GJX_FLAT_JOINS); `draws`: it consumes random bits at all.  -> the element of the stream at which the site's draws
  // start; `site` becomes the site number of the stream (the run's head for a member of a run); `opens` tells a member
  // that it is the head (the stream is to be opened).
  GJX_DEV uint32_t run_elem(uint32_t& site, bool joins, bool draws, bool& opens) {
    opens = false;
    if (joins) {
      if (run_head == 0u || run_next >= (uint32_t)GJX_FLAT_RUN_MAX) { run_head = site; run_next = 0u; opens = true; }
      site = run_head;
      return run_next++;
    }
    if (draws) run_head = 0u;
    return 0u;
  }
  // -> site number; `key` is the key to open the stream with
  GJX_DEV uint32_t next(int32_t scan) {
    if (scan == 0) { if (tag != 0) run_head = 0u; key = run_key; tag = 0; return ++plain; }
    if (scan != tag) {
      run_head = 0u;
      const uint32_t id = (uint32_t)scan >> 20;
      const int32_t step = (int32_t)((uint32_t)scan & 0xFFFFFu) - 1;
      const bool follows = tag != 0 && ((uint32_t)tag >> 20) == id && (int32_t)((uint32_t)tag & 0xFFFFFu) - 1 == step - 1;
      if (follows) key = fold_in(key, (uint32_t)step);
      else {
        key = fold_in(run_key, 0x80000000u | id);
        for (int32_t t = 0; t <= step; ++t) key = fold_in(key, (uint32_t)t);
      }
      tag = scan;
      local = 0u;
    }
    return ++local;
  }
};

// Run-time element indices (the site interpreter, rolled loops): the same stream with ONE inlined copy of the hash per
// get() instead of four — with two dozen distribution kinds instantiated in one kernel the fully inlined form is
// several hundred KB of code and thrashes the instruction cache (interpreter 147 us -> 850 us on the headline model).
template <int RNG>
struct BitStreamRT : BitStream<RNG> {};
template <>
struct BitStreamRT<GJX_RNG_FLAT> : BitStream<GJX_RNG_FLAT> {
  GJX_DEV uint32_t get(uint32_t c) {
    const uint32_t bit = 23u * c, n = bit >> 5, sh = bit & 31u;
    uint32_t lo = 0u, hi = 0u;
    const int nw = sh ? 2 : 1;
#pragma nounroll
    for (int k = 0; k < nw; ++k) {
      const uint32_t m = n + (uint32_t)k, h = m >> 1;
      const bool odd = (h & 1u) != 0u;
      if ((odd ? h1 : h0) != h) {
        const key2 b = threefry2x32(key, c0, site_hi | h);
        if (odd) { blk1 = b; h1 = h; } else { blk0 = b; h0 = h; }
      }
      const key2 b = odd ? blk1 : blk0;
      const uint32_t w = (m & 1u) ? b.b : b.a;
      if (k == 0) lo = w; else hi = w;
    }
    return sh ? __builtin_amdgcn_alignbit(hi, lo, sh) : lo;
  }
};

// ---- bits -> floats (jax/_src/random.py _uniform, _normal_real, gumbel) ------------------------
GJX_DEV float bits_to_unit(uint32_t bits) { return __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f; }

GJX_DEV float uniform_from_bits(uint32_t bits, float lo, float hi) {
  const float v = bits_to_unit(bits) * (hi - lo) + lo;
  return fmaxf(lo, v);
}

constexpr float kLn2 = 0.69314718056f;
constexpr float kHalfLog2Pi = 0.918938533f;
constexpr float kLogPi = 1.14472989f;
constexpr float kSqrt2 = 1.41421356f;
constexpr float kTiny = 1.17549435e-38f;
constexpr float kNeg1PlusUlp = -0.99999994f;
constexpr float kPi = 3.14159265f;

GJX_DEV float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }        // v_log_f32
GJX_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); } // v_exp_f32
GJX_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }               // v_rcp_f32
GJX_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }             // v_sqrt_f32

// log for arguments that can be denormal-small (uniform(tiny,1)): v_log_f32 flushes denormals,
// so scale into the normal range first.
GJX_DEV float safe_log(float x) {
  const bool small = x < 1.0e-30f;
  const float xs = small ? x * 1.8446744e19f /*2^64*/ : x;
  const float l = __builtin_amdgcn_logf(xs);
  return (small ? l - 64.0f : l) * kLn2;
}

GJX_DEV float log1p_acc(float x) {  // log(1+x), accurate near 0 (series below 2^-6)
  const float u = 1.0f + x;
  if (fabsf(x) < 0.015625f) {
    // log1p(x) = x - x^2/2 + x^3/3 - x^4/4 + x^5/5   (|x| < 2^-6 -> rel. error < 1e-9)
    return x * (1.0f + x * (-0.5f + x * (0.333333333f + x * (-0.25f + x * 0.2f))));
  }
  return fast_log(u);
}

// Giles (2010) single-precision erfinv (the polynomial of XLA's ErfInv32)
GJX_DEV float erfinv_f32(float x) {
  float w = -fast_log(fmaf(-x, x, 1.0f));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f);
    p = fmaf(p, w, -3.5233877e-06f);
    p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f);
    p = fmaf(p, w, -0.00125372503f);
    p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f);
    p = fmaf(p, w, 1.50140941f);
  } else {
    w = fast_sqrt(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f);
    p = fmaf(p, w, 0.00134934322f);
    p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f);
    p = fmaf(p, w, -0.0076224613f);
    p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f);
    p = fmaf(p, w, 2.83297682f);
  }
  return p * x;
}

// Hot-path variant of normal_from_bits: same Giles polynomial with sqrt(2) and ln(2) folded into
// the constants, (bits >> 9) | 0x3F800000 as one v_alignbit_b32, and the redundant max() dropped
// (f*2 is exact and rounding is monotone, so f*2 + lo >= lo).  Differs from normal_from_bits by
// float rounding only.
GJX_DEV float normal_from_bits_fast(uint32_t bits) {
  const float f = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, bits, 9)) - 1.0f;
  const float u = fmaf(f, 2.0f, kNeg1PlusUlp);
  const float L = __builtin_amdgcn_logf(fmaf(-u, u, 1.0f));
  float w = fmaf(L, -kLn2, -2.5f);  // (-ln(1-u^2)) - 2.5
  float p;
  if (w < 2.5f) {
    p = 3.9742602098158386e-08f;
    p = fmaf(p, w, 4.854626354244829e-07f);
    p = fmaf(p, w, -4.982822702004341e-06f);
    p = fmaf(p, w, -6.2105282268021256e-06f);
    p = fmaf(p, w, 0.0003091200196649879f);
    p = fmaf(p, w, -0.0017730349209159613f);
    p = fmaf(p, w, -0.005908133927732706f);
    p = fmaf(p, w, 0.3488026559352875f);
    p = fmaf(p, w, 2.1233136653900146f);
  } else {
    w = fast_sqrt(w + 2.5f) - 3.0f;
    p = -0.0002831457240972668f;
    p = fmaf(p, w, 0.00014276565343607217f);
    p = fmaf(p, w, 0.001908259466290474f);
    p = fmaf(p, w, -0.00519501231610775f);
    p = fmaf(p, w, 0.008116889744997025f);
    p = fmaf(p, w, -0.01077978778630495f);
    p = fmaf(p, w, 0.013348578475415707f);
    p = fmaf(p, w, 1.4165810346603394f);
    p = fmaf(p, w, 4.006434440612793f);
  }
  return p * u;
}

GJX_DEV float normal_from_bits(uint32_t bits) {
  const float u = uniform_from_bits(bits, kNeg1PlusUlp, 1.0f);
  return kSqrt2 * erfinv_f32(u);
}
// Box-Muller pair from two consecutive FLAT elements: (r cos 2πu2, r sin 2πu2), u1 in (0,1].
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so u2 feeds them directly.
GJX_DEV void box_muller(uint32_t wa, uint32_t wb, float& n0, float& n1) {
  const float u1 = 2.0f - __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wa, 9));  // 1 - unit, (0,1]
  // (the angle stays in [1, 2): v_sin / v_cos are periodic in revolutions and 1 + k 2^-23 is exact, so the -1 is an instruction for nothing)
  const float u2 = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, wb, 9));
  const float r = fast_sqrt(__builtin_amdgcn_logf(u1) * (-2.0f * kLn2));
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}

// standard normal for element e of a stream (see stream_normal in the oracle)
template <int RNG, class BS>
GJX_DEV float stream_normal(BS& bs, uint32_t e) {
  if constexpr (RNG == GJX_RNG_JAX32) {
    return normal_from_bits(bs.get(e));
  } else {
    if ((e >> 1) != bs.npair) {
      const uint32_t wa = bs.get(e & ~1u), wb = bs.get(e | 1u);
      box_muller(wa, wb, bs.n0, bs.n1);
      bs.npair = e >> 1;
    }
    return (e & 1u) ? bs.n1 : bs.n0;
  }
}

GJX_DEV float gumbel_from_bits(uint32_t bits) {
  const float u = uniform_from_bits(bits, kTiny, 1.0f);
  return -fast_log(-safe_log(u));
}

// digamma(x + 1/2) - digamma(x) without the cancellation of two digamma calls (the student-t's d/d(df) at large df: the two
// values agree to all 24 bits from df ~ 1e7 on, and the gradient would keep a spurious -1/(2 df)): recurrence up to x >= 8
// — every term 1 / (2 (x + k)(x + k + 1/2)) is positive — then 1/(2x) + 1/(8x^2) - 1/(64x^4) + 1/(128x^6); relative error < 2e-7
GJX_DEV float digamma_half_step(float x) {
  float acc = 0.0f;
  for (int k = 0; k < 8; ++k) {
    if (x < 8.0f) { acc += 0.5f * fast_rcp(x * (x + 0.5f)); x += 1.0f; }
  }
  const float r = fast_rcp(x), r2 = r * r;
  return acc + r * (0.5f + r * (0.125f + r2 * (-0.015625f + r2 * 0.0078125f)));
}

// lgamma(x + 1/2) - lgamma(x) without the cancellation of two lgamma calls (the student-t's normaliser at large df: at df = 1e4
// the two values are 3.4e4 and agree to 4 digits — float32 leaves +-2 ulp of 3.4e4 = 0.008; found by profiles/fuzz.sh as
// score -17.149 vs -15.149): the asymptotic series of log(Gamma(x + 1/2) / Gamma(x)) from x >= 8, relative error < 1e-7
GJX_DEV float lgamma_half_step(float x) {
  if (x < 8.0f) return lgammaf(x + 0.5f) - lgammaf(x);
  const float r = fast_rcp(x), r2 = r * r;
  return fmaf(0.5f, fast_log(x), r * (-0.125f + r2 * (5.2083333e-3f /*1/192*/ + r2 * (-1.5625e-3f /*1/640*/ + r2 * 1.1858259e-3f /*17/14336*/))));
}
// lgamma(a + b) - lgamma(a), a, b > 0: the log-beta normaliser with one large argument is lgamma(small) minus this, and the two
// lgamma values of the large side cancel.  Stirling from a >= 8: (a - 1/2) log1p(b / a) + b log(a + b) - b + (corrections)
GJX_DEV float lgamma_step(float a, float b) {
  if (a < 8.0f) return lgammaf(a + b) - lgammaf(a);
  const float s = a + b, ra = fast_rcp(a), rs = fast_rcp(s);
  const float corr = -b * ra * rs * 0.0833333333f - 2.7777778e-3f * (rs * rs * rs - ra * ra * ra);
  return fmaf(a - 0.5f, log1p_acc(b * ra), fmaf(b, fast_log(s), -b)) + corr;
}
GJX_DEV float lbeta_f(float a, float b) {
  const float hi = fmaxf(a, b), lo = fminf(a, b);
  return lgammaf(lo) - lgamma_step(hi, lo);
}

// ---- gamma-family log-densities at large shape parameters.  a log z - z - lgamma(a) is a difference of terms of size a log a that
// leaves O(1): in float32 the closed form is piecewise constant from a ~ 1e5 (steps of 0.25 at a = 1e7).  From a >= 8 the same
// value in the deviance form  1/2 log a - 1/2 log 2 pi - S(a) + a (log1p(d) - d),  d = (z - a) / a,  S = Stirling's correction:
// nothing large is subtracted — the caller forms z - a with ONE rounding (fma), log1p(d) - d comes from a series.
// log1p(t) - t: below |t| = 1/4, log1p(t) = 2 atanh(s) with s = t / (2 + t), and 2 s - t = -s t exactly
GJX_DEV float log1p_minus(float t) {
  if (!(fabsf(t) < 0.25f)) return log1p_acc(t) - t;
  const float s = t * fast_rcp(2.0f + t), s2 = s * s;
  return fmaf(-s, t, 2.0f * s * s2 * (0.333333333f + s2 * (0.2f + s2 * (0.142857143f + s2 * (0.111111111f + s2 * 0.0909090909f)))));
}
GJX_DEV float stirling_corr(float a) {      // lgamma(a) - ((a - 1/2) log a - a + 1/2 log 2 pi), a >= 8
  const float r = fast_rcp(a), r2 = r * r;
  return r * (0.0833333333f - r2 * (2.7777778e-3f - r2 * 7.9365079e-4f));
}
// a log z - z - lgamma(a) for a >= 8; num = z - a
GJX_DEV float gamma_kernel_big(float a, float num) {
  return fmaf(a, log1p_minus(num * fast_rcp(a)), 0.5f * fast_log(a) - kHalfLog2Pi - stirling_corr(a));
}

// digamma: recurrence up to x >= 6, then the asymptotic series (|error| < 1e-6 for x > 1e-3)
GJX_DEV float digamma_f(float x) {
  float acc = 0.0f;
  for (int k = 0; k < 6; ++k) {
    if (x < 6.0f) { acc -= fast_rcp(x); x += 1.0f; }
  }
  const float r = fast_rcp(x), r2 = r * r;
  return acc + fast_log(x) - 0.5f * r - r2 * (0.0833333333f - r2 * (0.00833333333f - r2 * 0.00396825397f));
}

GJX_DEV float softplus(float x) { return fmaxf(x, 0.0f) + log1p_acc(fast_exp(-fabsf(x))); }
GJX_DEV float sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// ---- parameter evaluation ------------------------------------------------------------------
// `val(slot)` returns the current particle's value of a slot.
// `inst`: instance of the site's plate (gjx.h "Plates": off + inst * d_off, slot + inst * d_slot, moff + inst * d_moff); 0 elsewhere
// GJX_P_VGATHER: the slot the parameter reads (gjx.h): row idx of an earlier vector-valued choice; idx an earlier discrete choice, or
// (slot < 0) the table value of a choice constrained to one value for every particle
template <class ValFn>
GJX_DEV int vgather_row(const gjx_param& p, int d, const float* __restrict__ tab, ValFn&& val, int inst = 0) {
  int idx = (int)(p.slot >= 0 ? val(p.slot + inst * p.d_slot) : tab[p.off + inst * p.d_off]);
  idx = idx < 0 ? 0 : (idx > p.n - 1 ? p.n - 1 : idx);
  return p.moff + inst * p.d_moff + idx * p.len + (p.len == 1 ? 0 : d % p.len);
}
// ---- GJX_P_EXPR (gjx.h): a block of scalar SSA nodes in the table, node i = tab[off + 6 i ..] = {op, a, b, c, da, db}.  The site
// interpreters evaluate the block into a per-lane array (dynamic indexing: scratch memory — this is the fallback engine; generated
// kernels emit the nodes as straight-line code, gjx_codegen.hip emit_expr_nodes) and sweep it backwards for gradients.
GJX_DEV float expr_tanh(float x) {
  const float e = fast_exp(-2.0f * fabsf(x));
  const float t = (1.0f - e) * fast_rcp(1.0f + e);
  return x < 0.0f ? -t : t;
}
GJX_DEV float expr_unary(int op, float x) {
  switch (op) {
    case GJX_E_NEG: return -x;
    case GJX_E_EXP: return fast_exp(x);
    case GJX_E_LOG: return fast_log(x);
    case GJX_E_SQRT: return fast_sqrt(x);
    case GJX_E_SQUARE: return x * x;
    case GJX_E_TANH: return expr_tanh(x);
    case GJX_E_SIGMOID: return fast_rcp(1.0f + fast_exp(-x));
    case GJX_E_SOFTPLUS: return fmaxf(x, 0.0f) + log1p_acc(fast_exp(-fabsf(x)));
    case GJX_E_ABS: return fabsf(x);
    case GJX_E_SIN: return sinf(x);
    case GJX_E_COS: return cosf(x);
    case GJX_E_LOG1P: return log1p_acc(x);
    default: return fast_rcp(x);   // GJX_E_RECIP
  }
}
// d unary(x) / dx given x and y = unary(x)
GJX_DEV float expr_unary_deriv(int op, float x, float y) {
  switch (op) {
    case GJX_E_NEG: return -1.0f;
    case GJX_E_EXP: return y;
    case GJX_E_LOG: return fast_rcp(x);
    case GJX_E_SQRT: return 0.5f * fast_rcp(y);
    case GJX_E_SQUARE: return 2.0f * x;
    case GJX_E_TANH: return 1.0f - y * y;
    case GJX_E_SIGMOID: return y * (1.0f - y);
    case GJX_E_SOFTPLUS: return fast_rcp(1.0f + fast_exp(-x));
    case GJX_E_ABS: return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    case GJX_E_SIN: return cosf(x);
    case GJX_E_COS: return -sinf(x);
    case GJX_E_LOG1P: return fast_rcp(1.0f + x);
    default: return -y * y;        // GJX_E_RECIP
  }
}
template <class ValFn>
GJX_DEV void expr_forward(const gjx_param& p, const float* __restrict__ tab, ValFn&& val, int inst, float* __restrict__ ev) {
  const float* __restrict__ nd = tab + p.off;
  const int n = p.n < GJX_EXPR_MAX_NODES ? p.n : GJX_EXPR_MAX_NODES;
  for (int i = 0; i < n; ++i) {
    const float* __restrict__ q6 = nd + GJX_EXPR_NODE_FLOATS * i;
    const int op = (int)q6[0], c = (int)q6[3];
    const int a = (int)q6[1] + inst * (int)q6[4], b = (int)q6[2] + inst * (int)q6[5];     // (plate strides: 0 outside plates)
    float r;
    switch (op) {
      case GJX_E_CONST: r = tab[a]; break;
      case GJX_E_VALUE: r = val(a); break;
      case GJX_E_ADD: r = ev[a] + ev[b]; break;
      case GJX_E_SUB: r = ev[a] - ev[b]; break;
      case GJX_E_MUL: r = ev[a] * ev[b]; break;
      case GJX_E_DIV: r = ev[a] * fast_rcp(ev[b]); break;
      case GJX_E_MAX: r = ev[a] >= ev[b] ? ev[a] : ev[b]; break;
      case GJX_E_MIN: r = ev[a] <= ev[b] ? ev[a] : ev[b]; break;
      case GJX_E_GT: r = ev[a] > ev[b] ? 1.0f : 0.0f; break;
      case GJX_E_WHERE: r = ev[a] != 0.0f ? ev[b] : ev[c]; break;
      case GJX_E_LINV: { r = tab[a]; for (int e = 0; e < c; ++e) r = fmaf(tab[a + 1 + e], val(b + e), r); break; }
      case GJX_E_LINN: { r = tab[a]; for (int e = 0; e < c; ++e) r = fmaf(tab[a + 1 + e], ev[b + e], r); break; }
      default: r = expr_unary(op, ev[a]); break;
    }
    ev[i] = r;
  }
}
// element d of the parameter: output node n - len + d % len
GJX_DEV int expr_out_node(const gjx_param& p, int d) { return p.n - p.len + (p.len == 1 ? 0 : d % p.len); }
template <class ValFn>
GJX_DEV float expr_eval(const gjx_param& p, int d, const float* __restrict__ tab, ValFn&& val, int inst) {
  float ev[GJX_EXPR_MAX_NODES];
  expr_forward(p, tab, val, inst, ev);
  return ev[expr_out_node(p, d)];
}
// reverse sweep: adjoint g of output element d flows to the gradient rows G.at(slot) of the block's VALUE / LINV leaves
template <class ValFn, class Rows>
GJX_DEV void expr_backward(const gjx_param& p, int d, float g, const float* __restrict__ tab, ValFn&& val, int inst, Rows G) {
  float ev[GJX_EXPR_MAX_NODES], ad[GJX_EXPR_MAX_NODES];
  expr_forward(p, tab, val, inst, ev);
  const float* __restrict__ nd = tab + p.off;
  const int n = p.n < GJX_EXPR_MAX_NODES ? p.n : GJX_EXPR_MAX_NODES;
  for (int i = 0; i < n; ++i) ad[i] = 0.0f;
  ad[expr_out_node(p, d)] = g;
  for (int i = n - 1; i >= 0; --i) {
    const float gi = ad[i];
    if (gi == 0.0f) continue;
    const float* __restrict__ q6 = nd + GJX_EXPR_NODE_FLOATS * i;
    const int op = (int)q6[0], c = (int)q6[3];
    const int a = (int)q6[1] + inst * (int)q6[4], b = (int)q6[2] + inst * (int)q6[5];
    switch (op) {
      case GJX_E_CONST: case GJX_E_GT: break;
      case GJX_E_VALUE: G.at(a) += gi; break;
      case GJX_E_ADD: ad[a] += gi; ad[b] += gi; break;
      case GJX_E_SUB: ad[a] += gi; ad[b] -= gi; break;
      case GJX_E_MUL: ad[a] = fmaf(gi, ev[b], ad[a]); ad[b] = fmaf(gi, ev[a], ad[b]); break;
      case GJX_E_DIV: { const float rb = fast_rcp(ev[b]); ad[a] = fmaf(gi, rb, ad[a]); ad[b] = fmaf(-gi * ev[i], rb, ad[b]); break; }
      case GJX_E_MAX: if (ev[a] >= ev[b]) ad[a] += gi; else ad[b] += gi; break;
      case GJX_E_MIN: if (ev[a] <= ev[b]) ad[a] += gi; else ad[b] += gi; break;
      case GJX_E_WHERE: if (ev[a] != 0.0f) ad[b] += gi; else ad[c] += gi; break;
      case GJX_E_LINV: for (int e = 0; e < c; ++e) G.at(b + e) += gi * tab[a + 1 + e]; break;
      case GJX_E_LINN: for (int e = 0; e < c; ++e) ad[b + e] = fmaf(gi, tab[a + 1 + e], ad[b + e]); break;
      default: ad[a] = fmaf(gi, expr_unary_deriv(op, ev[a], ev[i]), ad[a]); break;
    }
  }
}

template <class ValFn>
GJX_DEV float eval_param_pre(const gjx_param& p, int d, const float* __restrict__ tab, ValFn&& val, int inst = 0) {
  const int off = p.off + inst * p.d_off, slot = p.slot + inst * p.d_slot;
  switch (p.op) {
    case GJX_P_CONST: return tab[off + (p.len == 1 ? 0 : d % p.len)];
    case GJX_P_VALUE: return val(slot + (p.len == 1 ? 0 : d % p.len));
    case GJX_P_GATHER: {
      int idx = (int)val(slot);
      idx = idx < 0 ? 0 : (idx > p.n - 1 ? p.n - 1 : idx);
      return tab[off + idx * p.len + (p.len == 1 ? 0 : d % p.len)];
    }
    case GJX_P_AFFINE: {
      float acc = tab[off + (p.len == 1 ? 0 : d % p.len)];
      const float* row = tab + p.moff + inst * p.d_moff + d * p.n;
      for (int e = 0; e < p.n; ++e) acc = fmaf(row[e], val(slot + e), acc);
      return acc;
    }
    case GJX_P_VGATHER: return val(vgather_row(p, d, tab, val, inst));
    case GJX_P_EXPR: return expr_eval(p, d, tab, val, inst);
    default: return __builtin_nanf("");
  }
}
GJX_DEV float apply_xf(int xf, float v) {
  switch (xf) {
    case GJX_XF_EXP: return fast_exp(v);
    case GJX_XF_SOFTPLUS: return softplus(v);
    case GJX_XF_SIGMOID: return sigmoid(v);
    default: return v;
  }
}
template <class ValFn>
GJX_DEV float eval_param(const gjx_param& p, int d, const float* __restrict__ tab, ValFn&& val, int inst = 0) {
  return apply_xf(p.xf, eval_param_pre(p, d, tab, val, inst));
}

// ---- log-densities of one scalar element (TFP 0.23 log_prob closed forms) --------------------
GJX_DEV float normal_logpdf(float x, float mu, float sd) {
  const float rs = fast_rcp(sd);
  const float z = (x - mu) * rs;
  return fmaf(-0.5f * z, z, -(kHalfLog2Pi + fast_log(sd)));
}

// standard normal CDF differences for the truncated normal: log(Phi(hi) - Phi(lo)), evaluated in the tail that
// keeps precision (erfc of a positive argument)
GJX_DEV float normal_cdf(float z) { return 0.5f * erfcf(-z * 0.70710678f); }
GJX_DEV float normal_interval_mass(float lo, float hi) {
  if (lo > 0.0f) return normal_cdf(-lo) - normal_cdf(-hi);
  return normal_cdf(hi) - normal_cdf(lo);
}

// exp(-x) I0(x) and I1(x) / I0(x), x >= 0: Abramowitz & Stegun 9.8.1 - 9.8.4 (|relative error| < 2e-7)
GJX_DEV float bessel_i0e(float x) {
  if (x < 3.75f) {
    const float t = x * (1.0f / 3.75f), t2 = t * t;
    const float p = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(0.0045813f, t2, 0.0360768f), t2, 0.2659732f), t2, 1.2067492f), t2, 3.0899424f), t2, 3.5156229f), t2, 1.0f);
    return p * fast_exp(-x);
  }
  const float u = 3.75f * fast_rcp(x);
  const float p = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(0.00392377f, u, -0.01647633f), u, 0.02635537f), u, -0.02057706f), u, 0.00916281f), u, -0.00157565f), u, 0.00225319f), u, 0.01328592f), u, 0.39894228f);
  return p * rsqrtf(x);
}
GJX_DEV float bessel_i1_over_i0(float x) {
  if (x < 3.75f) {
    const float t = x * (1.0f / 3.75f), t2 = t * t;
    const float p1 = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(0.00032411f, t2, 0.00301532f), t2, 0.02658733f), t2, 0.15084934f), t2, 0.51498869f), t2, 0.87890594f), t2, 0.5f);
    const float p0 = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(0.0045813f, t2, 0.0360768f), t2, 0.2659732f), t2, 1.2067492f), t2, 3.0899424f), t2, 3.5156229f), t2, 1.0f);
    return x * p1 * fast_rcp(p0);
  }
  const float u = 3.75f * fast_rcp(x);
  const float p1 = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(-0.00420059f, u, 0.01787654f), u, -0.02895312f), u, 0.02282967f), u, -0.01031555f), u, 0.00163801f), u, -0.00362018f), u, -0.03988024f), u, 0.39894228f);
  const float p0 = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(0.00392377f, u, -0.01647633f), u, 0.02635537f), u, -0.02057706f), u, 0.00916281f), u, -0.00157565f), u, 0.00225319f), u, 0.01328592f), u, 0.39894228f);
  return p1 * fast_rcp(p0);
}

// parameters a (non-categorical) kind reads, and stream elements one scalar of it may draw (host and device: the emitters use them too)
constexpr int kGammaMaxIt = 32;
constexpr int kGammaNDraw = 4 * kGammaMaxIt + 2;  // draw schedule: see GAMMA_NDRAW in the oracle
constexpr int kPoissonTries = 16;
constexpr int kVonMisesTries = 16;
constexpr int kind_params(int kind) {
  return (kind == GJX_TRUNCATED_NORMAL || kind == GJX_TRUNCATED_CAUCHY) ? 4 : ((kind == GJX_STUDENT_T || kind == GJX_HALF_STUDENT_T) ? 3 : 2);
}
constexpr int kind_draws(int kind) {
  return kind == GJX_BETA ? 2 * kGammaNDraw
         : (kind == GJX_GAMMA || kind == GJX_DIRICHLET || kind == GJX_INVERSE_GAMMA || kind == GJX_CHI2 || kind == GJX_CHI || kind == GJX_EXP_GAMMA ||
            kind == GJX_EXP_INVERSE_GAMMA) ? kGammaNDraw
         : (kind == GJX_STUDENT_T || kind == GJX_HALF_STUDENT_T || kind == GJX_DOUBLESIDED_MAXWELL) ? kGammaNDraw + 2
         : kind == GJX_POISSON ? 2 * kPoissonTries + 2
         : kind == GJX_NEGATIVE_BINOMIAL ? kGammaNDraw + 2 * kPoissonTries + 2
         : kind == GJX_VON_MISES ? 2 * kVonMisesTries + 2
         : kind == GJX_INVERSE_GAUSSIAN ? 4 : 1;
}
GJX_DEV int params_of(int kind) { return kind_params(kind); }

GJX_DEV float elem_logpdf(int kind, float x, float a, float b, float c = 0.0f, float d = 0.0f) {
  switch (kind) {
    case GJX_HALF_STUDENT_T:   // a = df, b = loc, c = scale: the student-t folded at its location
      if (x < b) return -INFINITY;
      [[fallthrough]];
    case GJX_STUDENT_T: {  // a = df, b = loc, c = scale
      const float y = (x - b) * fast_rcp(c);
      // log1p(y^2 / df): a draw with df << 1 reaches 1e30 and y^2 leaves float32 — there log(y^2 / df) stands for it (the 1 is below 1e-30 of it)
      const float q = y * y * fast_rcp(a);
      const float l1p = q < 1e30f ? log1p_acc(q) : 2.0f * fast_log(fabsf(y)) - fast_log(a);
      return -0.5f * (a + 1.0f) * l1p - fast_log(c) - 0.5f * fast_log(a) - 0.5f * kLogPi + lgamma_half_step(0.5f * a) + (kind == GJX_HALF_STUDENT_T ? kLn2 : 0.0f);
    }
    case GJX_TRUNCATED_CAUCHY: {  // a = loc, b = scale, c = low, d = high: 1 / (b (1 + z^2) (atan z_high - atan z_low))
      if (x < c || x > d) return -INFINITY;
      const float rs = fast_rcp(b);
      const float z = (x - a) * rs;
      return -fast_log(b) - log1p_acc(z * z) - fast_log(atanf((d - a) * rs) - atanf((c - a) * rs));
    }
    case GJX_NEGATIVE_BINOMIAL: {  // a = total_count r, b = logits l of the success probability: C(x + r - 1, x) p^x (1 - p)^r
      if (x < 0.0f || x != floorf(x)) return -INFINITY;
      // log C(x + r - 1, x) = lgamma(hi + lo - 1) - lgamma(hi) - lgamma(lo), {hi, lo} = {r, x + 1}: the two large values cancel inside lgamma_step
      const float hi = fmaxf(a, x + 1.0f), lo = fminf(a, x + 1.0f);
      return lgamma_step(hi, lo - 1.0f) - lgammaf(lo) - (x == 0.0f ? 0.0f : x * softplus(-b)) - a * softplus(b);
    }
    case GJX_VON_MISES: {  // a = loc, b = concentration: exp(k cos(x - loc)) / (2 pi I0(k)), with I0(k) = e^k i0e(k)
      return b * (cosf(x - a) - 1.0f) - 1.83787707f - fast_log(bessel_i0e(b));
    }
    case GJX_CHI: {  // a = df: sqrt of a chi2(df) variate
      if (x <= 0.0f) return -INFINITY;
      const float h = 0.5f * a;
      if (h >= 8.0f) return gamma_kernel_big(h, 0.5f * fmaf(x, x, -a)) + kLn2 - fast_log(x);      // chi2's density at x^2, times 2 x
      return (1.0f - h) * kLn2 + ((a - 1.0f) == 0.0f ? 0.0f : (a - 1.0f) * fast_log(x)) - 0.5f * x * x - lgammaf(h);
    }
    case GJX_EXP_GAMMA: {  // a = concentration, b = rate: y = log of a gamma variate; with z = b e^y: a log z - z - lgamma(a)
      const float z = b * fast_exp(x);
      if (a >= 8.0f) return gamma_kernel_big(a, z - a);
      return a * (fast_log(b) + x) - z - lgammaf(a);
    }
    case GJX_EXP_INVERSE_GAMMA: {  // a = concentration, b = scale: y = log of an inverse-gamma variate; z = b e^-y
      const float z = b * fast_exp(-x);
      if (a >= 8.0f) return gamma_kernel_big(a, z - a);
      return a * (fast_log(b) - x) - z - lgammaf(a);
    }
    case GJX_KUMARASWAMY: {  // a = concentration1, b = concentration0
      if (!(x > 0.0f && x < 1.0f)) return -INFINITY;
      const float lx = fast_log(x);
      return fast_log(a * b) + ((a - 1.0f) == 0.0f ? 0.0f : (a - 1.0f) * lx) + ((b - 1.0f) == 0.0f ? 0.0f : (b - 1.0f) * log1p_acc(-fast_exp(a * lx)));
    }
    case GJX_MOYAL: {  // a = loc, b = scale
      const float z = (x - a) * fast_rcp(b);
      return -0.5f * (z + fast_exp(-z)) - fast_log(b) - 0.918938533f /* log sqrt(2 pi) */;
    }
    case GJX_DOUBLESIDED_MAXWELL: {  // a = loc, b = scale: z^2 exp(-z^2 / 2) / (b sqrt(2 pi))
      const float z = (x - a) * fast_rcp(b);
      return 2.0f * fast_log(fabsf(z)) - 0.5f * z * z - fast_log(b) - 0.918938533f;
    }
    case GJX_INVERSE_GAUSSIAN: {  // a = loc (mean), b = concentration
      if (x <= 0.0f) return -INFINITY;
      const float r = (x - a) * fast_rcp(a);
      return 0.5f * (fast_log(b) - 1.83787707f /* log 2 pi */ - 3.0f * fast_log(x)) - 0.5f * b * r * r * fast_rcp(x);
    }
    case GJX_TRUNCATED_NORMAL: {  // a = loc, b = scale, c = low, d = high
      if (x < c || x > d) return -INFINITY;
      const float rs = fast_rcp(b);
      return normal_logpdf(x, a, b) - fast_log(normal_interval_mass((c - a) * rs, (d - a) * rs));
    }
    case GJX_POISSON:
      if (x < 0.0f || x != floorf(x)) return -INFINITY;
      if (x >= 7.0f) return gamma_kernel_big(x + 1.0f, a - (x + 1.0f)) - fast_log(a);     // (x + 1) log a - a - lgamma(x + 1) - log a
      return (x == 0.0f ? 0.0f : x * fast_log(a)) - a - lgammaf(x + 1.0f);
    case GJX_GEOMETRIC: return (x < 0.0f || x != floorf(x)) ? -INFINITY : ((x == 0.0f ? 0.0f : x * log1p_acc(-a)) + fast_log(a));
    case GJX_GUMBEL: {
      const float z = (x - a) * fast_rcp(b);
      return -(z + fast_exp(-z)) - fast_log(b);
    }
    case GJX_HALF_CAUCHY: {
      const float z = (x - a) * fast_rcp(b);
      return x < a ? -INFINITY : (-0.451582705f /*log(2/pi)*/ - fast_log(b) - log1p_acc(z * z));
    }
    case GJX_INVERSE_GAMMA:  // a = concentration, b = scale
      if (x <= 0.0f) return -INFINITY;
      if (a >= 8.0f) return gamma_kernel_big(a, fmaf(-a, x, b) * fast_rcp(x)) - fast_log(x);   // z = b / x: a log z - z - lgamma(a) - log x
      return a * fast_log(b) - lgammaf(a) - (a + 1.0f) * fast_log(x) - b * fast_rcp(x);
    case GJX_WEIBULL: {  // a = concentration k, b = scale
      if (x < 0.0f) return -INFINITY;
      const float lr = fast_log(x * fast_rcp(b));
      return fast_log(a * fast_rcp(b)) + ((a - 1.0f) == 0.0f ? 0.0f : (a - 1.0f) * lr) - fast_exp(a * lr);
    }
    case GJX_LOGIT_NORMAL: {
      if (!(x > 0.0f && x < 1.0f)) return -INFINITY;
      const float lx = fast_log(x), l1 = log1p_acc(-x);
      return normal_logpdf(lx - l1, a, b) - lx - l1;
    }
    case GJX_CHI2: {  // a = df
      const float h = 0.5f * a;
      if (x <= 0.0f) return -INFINITY;
      if (h >= 8.0f) return gamma_kernel_big(h, 0.5f * (x - a)) - fast_log(x);                  // gamma(h, rate 1/2): z = x / 2
      return ((h - 1.0f) == 0.0f ? 0.0f : (h - 1.0f) * fast_log(x)) - 0.5f * x - h * kLn2 - lgammaf(h);
    }
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: return normal_logpdf(x, a, b);
    case GJX_FLIP:
      return (x != 0.0f ? fast_log(a) : 0.0f) + (x != 1.0f ? (1.0f - x) * log1p_acc(-a) : 0.0f);
    case GJX_BERNOULLI_LOGITS:
      return (x != 0.0f ? -softplus(-a) * x : 0.0f) + (x != 1.0f ? -softplus(a) * (1.0f - x) : 0.0f);
    case GJX_BETA: {
      if (a >= 8.0f && b >= 8.0f && x > 0.0f && x < 1.0f) {
        // both shapes large: (a - 1) log x + (b - 1) log(1 - x) and log B(a, b) are each of size n = a + b.  With the mode's
        // neighbour p = a / n:  a log(x / p) + b log((1 - x) / (1 - p)) = a L(u) + b L(w),  L(t) = log1p(t) - t,  u = t_ / a,
        // w = -t_ / b,  t_ = n x - a (the linear terms a u + b w cancel identically), and the rest of Stirling is O(log n).
        // t_ = b x - a (1 - x) from two fused multiply-adds: the rounding of n = a + b (up to 16 at n = 3e8) never enters it
        const float n = a + b, t_ = fmaf(b, x, fmaf(a, x, -a));
        const float dev = fmaf(a, log1p_minus(t_ * fast_rcp(a)), b * log1p_minus(-t_ * fast_rcp(b)));
        return dev + 0.5f * (fast_log(a) + fast_log(b) - fast_log(n)) - kHalfLog2Pi - stirling_corr(a) - stirling_corr(b) + stirling_corr(n)
               - fast_log(x) - log1p_acc(-x);
      }
      const float t1 = (a - 1.0f) == 0.0f ? 0.0f : (a - 1.0f) * fast_log(x);
      const float t2 = (b - 1.0f) == 0.0f ? 0.0f : (b - 1.0f) * log1p_acc(-x);
      return t1 + t2 - lbeta_f(a, b);
    }
    case GJX_UNIFORM: return (x < a || x > b) ? -INFINITY : -fast_log(b - a);
    case GJX_EXPONENTIAL: return x < 0.0f ? -INFINITY : fast_log(a) - a * x;
    case GJX_HALF_NORMAL: {
      const float z = x * fast_rcp(a);
      return x < 0.0f ? -INFINITY : (-0.225791353f /*0.5*log(2/pi)*/ - fast_log(a) - 0.5f * z * z);
    }
    case GJX_LAPLACE: return -fabsf(x - a) * fast_rcp(b) - fast_log(2.0f * b);
    case GJX_LOG_NORMAL: {
      const float lx = fast_log(x);
      return normal_logpdf(lx, a, b) - lx;
    }
    case GJX_CAUCHY: {
      const float z = (x - a) * fast_rcp(b);
      return -(kLogPi + fast_log(b)) - log1p_acc(z * z);
    }
    case GJX_GAMMA: {
      if (a >= 8.0f && x > 0.0f) return gamma_kernel_big(a, fmaf(b, x, -a)) - fast_log(x);     // z = b x: a log z - z - lgamma(a) - log x
      const float t0 = a == 0.0f ? 0.0f : a * fast_log(b);
      const float t1 = (a - 1.0f) == 0.0f ? 0.0f : (a - 1.0f) * fast_log(x);
      return t0 + t1 - b * x - lgammaf(a);
    }
    default: return __builtin_nanf("");
  }
}

// ---- samplers ----------------------------------------------------------------------------------
GJX_DEV int draws_per_elem(int kind) { return kind_draws(kind); }

// Marsaglia & Tsang (2000), log space, fixed draw budget (same element schedule as the oracle)
template <int RNG, class BS>
GJX_DEV float log_gamma_variate(BS& bs, uint32_t base, float a) {
  float boost = 0.0f, aa = a;
  if (a < 1.0f) {
    const float u = uniform_from_bits(bs.get(base + 4 * kGammaMaxIt), kTiny, 1.0f);
    boost = safe_log(u) / a;
    aa = a + 1.0f;
  }
  const float d = aa - (1.0f / 3.0f);
  const float c = 1.0f / sqrtf(9.0f * d);
  float res = fast_log(d);
  for (int t = 0; t < kGammaMaxIt; ++t) {
    const float x = stream_normal<RNG>(bs, base + 4 * t);
    const float u = uniform_from_bits(bs.get(base + 4 * t + 2), kTiny, 1.0f);
    float v = fmaf(c, x, 1.0f);
    if (v <= 0.0f) continue;
    const float lv = 3.0f * fast_log(v);
    v = v * v * v;
    if (safe_log(u) < 0.5f * x * x + d - d * v + d * lv) {
      res = fast_log(d) + lv;
      break;
    }
  }
  return res + boost;
}

// Poisson: inversion by sequential search on one uniform for rate < 10, Hörmann's transformed rejection (PTRS,
// 1993) with a fixed budget of tries above (same element schedule as the oracle: try t uses elements c+2+2t, +1)
template <int RNG, class BS>
GJX_DEV float poisson_variate(BS& bs, uint32_t c, float lam) {
  if (lam < 10.0f) {
    const float u = bits_to_unit(bs.get(c));
    float p = fast_exp(-lam), cdf = p;
    int k = 0;
    while (u > cdf && k < 96) {
      ++k;
      p *= lam / (float)k;
      cdf += p;
    }
    return (float)k;
  }
  const float slam = sqrtf(lam), loglam = fast_log(lam);
  const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
  const float inv_alpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.0f);
  for (int t = 0; t < kPoissonTries; ++t) {
    const float U = bits_to_unit(bs.get(c + 2 + 2 * t)) - 0.5f;
    const float V = uniform_from_bits(bs.get(c + 3 + 2 * t), kTiny, 1.0f);
    const float us = 0.5f - fabsf(U);
    const float k = floorf((2.0f * a / us + b) * U + lam + 0.43f);
    if (us >= 0.07f && V <= vr) return k;
    if (k < 0.0f || (us < 0.013f && V > us)) continue;
    if (fast_log(V) + fast_log(inv_alpha) - fast_log(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.0f)) return k;
  }
  return floorf(lam);
}

template <int RNG, class BS>
GJX_DEV float elem_sample(int kind, BS& bs, uint32_t c, float a, float b, float p3 = 0.0f, float p4 = 0.0f) {
  switch (kind) {
    case GJX_HALF_STUDENT_T:
    case GJX_STUDENT_T: {  // a = df, b = loc, p3 = scale: z * sqrt(df / chi2_df), chi2_df = 2 * Gamma(df/2)
      const float z = stream_normal<RNG>(bs, c);
      const float lg = log_gamma_variate<RNG>(bs, c + 2, 0.5f * a);
      const float t = p3 * z * fast_exp(0.5f * (fast_log(0.5f * a) - lg));
      return b + (kind == GJX_HALF_STUDENT_T ? fabsf(t) : t);
    }
    case GJX_TRUNCATED_CAUCHY: {  // inverse CDF on the arctangent scale
      const float rs = fast_rcp(b);
      const float lo = atanf((p3 - a) * rs), hi = atanf((p4 - a) * rs);
      return fminf(fmaxf(fmaf(b, tanf(fmaf(bits_to_unit(bs.get(c)), hi - lo, lo)), a), p3), p4);
    }
    case GJX_NEGATIVE_BINOMIAL: {  // a gamma(r, rate e^-l) mixture of Poissons: rate = exp(log gamma(r, 1) + l)
      const float lg = log_gamma_variate<RNG>(bs, c, a);
      return poisson_variate<RNG>(bs, c + kGammaNDraw, fast_exp(lg + b));
    }
    case GJX_VON_MISES: {  // Best & Fisher (1979): a wrapped-Cauchy envelope, fixed budget of tries (elements 2 t, 2 t + 1; the sign: the last one)
      if (b < 1e-6f) return fmaf(kPi, fmaf(2.0f, bits_to_unit(bs.get(c)), -1.0f), a);          // (no concentration: uniform on the circle)
      const double kd = (double)b;
      const double tau = 1.0 + sqrt(1.0 + 4.0 * kd * kd);
      const double rho = (tau - sqrt(2.0 * tau)) / (2.0 * kd);
      const float r = (float)((1.0 + rho * rho) / (2.0 * rho));
      float f = 1.0f;
      for (int t = 0; t < kVonMisesTries; ++t) {
        const float z = cosf(kPi * bits_to_unit(bs.get(c + 2 * t)));
        const float u2 = uniform_from_bits(bs.get(c + 2 * t + 1), kTiny, 1.0f);
        f = fmaf(r, z, 1.0f) * fast_rcp(r + z);
        const float cc = b * (r - f);
        if (u2 < cc * (2.0f - cc) || fast_log(cc * fast_rcp(u2)) + 1.0f - cc >= 0.0f) break;
      }
      const float th = acosf(fminf(fmaxf(f, -1.0f), 1.0f));
      return a + (bits_to_unit(bs.get(c + 2 * kVonMisesTries)) < 0.5f ? -th : th);
    }
    case GJX_CHI: return fast_exp(0.5f * (kLn2 + log_gamma_variate<RNG>(bs, c, 0.5f * a)));
    case GJX_EXP_GAMMA: return log_gamma_variate<RNG>(bs, c, a) - fast_log(b);
    case GJX_EXP_INVERSE_GAMMA: return fast_log(b) - log_gamma_variate<RNG>(bs, c, a);
    case GJX_KUMARASWAMY: {  // x = (1 - (1 - u)^(1 / b))^(1 / a)
      const float t = log1p_acc(-bits_to_unit(bs.get(c))) * fast_rcp(b);
      return fast_exp(safe_log(-expm1f(t)) * fast_rcp(a));
    }
    case GJX_MOYAL: {  // -log of a chi2(1) variate: loc - scale log(n^2)
      const float n = stream_normal<RNG>(bs, c);
      return fmaf(-2.0f * b, safe_log(fabsf(n)), a);
    }
    case GJX_DOUBLESIDED_MAXWELL: {  // a random sign times the root of a chi2(3) variate
      const float sgn = bits_to_unit(bs.get(c)) < 0.5f ? -1.0f : 1.0f;
      const float r = fast_exp(0.5f * (kLn2 + log_gamma_variate<RNG>(bs, c + 2, 1.5f)));
      return fmaf(b * sgn, r, a);
    }
    case GJX_INVERSE_GAUSSIAN: {  // Michael, Schucany & Haas (1976): a = mean, b = concentration
      const float n = stream_normal<RNG>(bs, c);
      const float y = n * n;
      // the smaller root mu (1 + w - sqrt(w (w + 2))), w = mu y / (2 lambda), in the form without the cancellation: (1 + w)^2 - w (w + 2) = 1
      const float w = a * y * fast_rcp(2.0f * b);
      const float x1 = a * fast_rcp(1.0f + w + sqrtf(w * (w + 2.0f)));
      const float u = bits_to_unit(bs.get(c + 2));
      return u * (a + x1) <= a ? x1 : a * a * fast_rcp(x1);
    }
    case GJX_TRUNCATED_NORMAL: {  // inverse CDF inside [low, high], in the tail that keeps precision
      const float rs = fast_rcp(b);
      const float lo = (p3 - a) * rs, hi = (p4 - a) * rs;
      const float u = bits_to_unit(bs.get(c));
      float z;
      if (lo > 0.0f) {
        const float q = fmaf(-u, normal_cdf(-lo) - normal_cdf(-hi), normal_cdf(-lo));   // upper-tail mass
        z = -kSqrt2 * erfinv_f32(fmaf(2.0f, q, -1.0f));
      } else {
        const float q = fmaf(u, normal_cdf(hi) - normal_cdf(lo), normal_cdf(lo));
        z = kSqrt2 * erfinv_f32(fmaf(2.0f, q, -1.0f));
      }
      return fminf(fmaxf(fmaf(b, z, a), p3), p4);
    }
    case GJX_POISSON: return poisson_variate<RNG>(bs, c, a);
    case GJX_GEOMETRIC: return floorf(safe_log(uniform_from_bits(bs.get(c), kTiny, 1.0f)) / log1p_acc(-a));
    case GJX_GUMBEL: return a - b * safe_log(-safe_log(uniform_from_bits(bs.get(c), kTiny, 1.0f)));
    case GJX_HALF_CAUCHY: return fmaf(b, tanf(0.5f * kPi * bits_to_unit(bs.get(c))), a);
    case GJX_INVERSE_GAMMA: return b * fast_exp(-log_gamma_variate<RNG>(bs, c, a));
    case GJX_WEIBULL: return b * fast_exp(safe_log(-log1p_acc(-bits_to_unit(bs.get(c)))) / a);
    case GJX_LOGIT_NORMAL: return sigmoid(fmaf(b, stream_normal<RNG>(bs, c), a));
    case GJX_CHI2: return 2.0f * fast_exp(log_gamma_variate<RNG>(bs, c, 0.5f * a));
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: return fmaf(b, stream_normal<RNG>(bs, c), a);
    case GJX_FLIP: return bits_to_unit(bs.get(c)) < a ? 1.0f : 0.0f;
    case GJX_BERNOULLI_LOGITS: return bits_to_unit(bs.get(c)) < sigmoid(a) ? 1.0f : 0.0f;
    case GJX_BETA: {
      const float g1 = log_gamma_variate<RNG>(bs, c, a);
      const float g2 = log_gamma_variate<RNG>(bs, c + kGammaNDraw, b);
      return sigmoid(g1 - g2);
    }
    case GJX_UNIFORM: return fmaf(b - a, bits_to_unit(bs.get(c)), a);
    case GJX_EXPONENTIAL: return -safe_log(uniform_from_bits(bs.get(c), kTiny, 1.0f)) / a;
    case GJX_HALF_NORMAL: return fabsf(stream_normal<RNG>(bs, c)) * a;
    case GJX_LAPLACE: {
      const float u = uniform_from_bits(bs.get(c), kNeg1PlusUlp, 1.0f);
      const float s = (float)((u > 0.0f) - (u < 0.0f));
      return a - b * s * log1p_acc(-fabsf(u));
    }
    case GJX_LOG_NORMAL: return fast_exp(fmaf(b, stream_normal<RNG>(bs, c), a));
    case GJX_CAUCHY: return fmaf(b, tanf(kPi * (bits_to_unit(bs.get(c)) - 0.5f)), a);
    case GJX_GAMMA: return fast_exp(log_gamma_variate<RNG>(bs, c, a)) / b;
    default: return __builtin_nanf("");
  }
}

// ---- reductions --------------------------------------------------------------------------------
// `__shfl_*` compiles to ds_bpermute_b32 (an LDS-crossbar round trip, ~100+ cycles each): a 64-lane u64 scan is 12 of them
// in a dependent chain, and the latency-bound co-resident kernels run half a dozen such chains per step.  The same
// scan as 6 DPP steps (MI355X guide, DPP idioms): row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into rows 2 and 3; lanes without a source keep the identity (`old`).
template <unsigned CTRL, unsigned ROW_MASK>
GJX_DEV uint32_t dpp_mov(uint32_t identity, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
constexpr unsigned kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppBcast15 = 0x142, kDppBcast31 = 0x143;

template <unsigned CTRL, unsigned ROW_MASK>
GJX_DEV float dpp_max_f32(float v) {
  return fmaxf(v, __uint_as_float(dpp_mov<CTRL, ROW_MASK>(0xFF800000u /* -inf */, __float_as_uint(v))));
}
template <unsigned CTRL, unsigned ROW_MASK>
GJX_DEV float dpp_add_f32(float v) { return v + __uint_as_float(dpp_mov<CTRL, ROW_MASK>(0u, __float_as_uint(v))); }
GJX_DEV float wave_max_dpp(float v) {        // maximum over the wave, in every lane (NaN never wins, as fmaxf)
  v = dpp_max_f32<kDppShr1, 0xf>(v);
  v = dpp_max_f32<kDppShr2, 0xf>(v);
  v = dpp_max_f32<kDppShr4, 0xf>(v);
  v = dpp_max_f32<kDppShr8, 0xf>(v);
  v = dpp_max_f32<kDppBcast15, 0xa>(v);
  v = dpp_max_f32<kDppBcast31, 0xc>(v);
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
GJX_DEV float wave_sum_dpp(float v) {        // sum over the wave, in every lane (a fixed order, not wave_sum's butterfly)
  v = dpp_add_f32<kDppShr1, 0xf>(v);
  v = dpp_add_f32<kDppShr2, 0xf>(v);
  v = dpp_add_f32<kDppShr4, 0xf>(v);
  v = dpp_add_f32<kDppShr8, 0xf>(v);
  v = dpp_add_f32<kDppBcast15, 0xa>(v);
  v = dpp_add_f32<kDppBcast31, 0xc>(v);
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
GJX_DEV float row_sum_to_lane15(float v) {
  v = dpp_add_f32<kDppShr1, 0xf>(v);
  v = dpp_add_f32<kDppShr2, 0xf>(v);
  v = dpp_add_f32<kDppShr4, 0xf>(v);
  v = dpp_add_f32<kDppShr8, 0xf>(v);
  return v;
}

GJX_DEV float wave_max(float v) { return wave_max_dpp(v); }
GJX_DEV float wave_sum(float v) { return wave_sum_dpp(v); }

// Block-wide {max, sum exp(x - max)} of one value per thread; result valid in thread 0.
// `red` is LDS scratch of >= 2 * (blockDim/64) floats.
template <int THREADS>
GJX_DEV void block_lse_partial(float x, bool valid, float* red, float& out_max, float& out_sum) {
  constexpr int NW = THREADS / 64;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float xm = valid ? x : -INFINITY;
  float m = wave_max(xm);
  if (lane == 0) red[wid] = m;
  __syncthreads();
  float bm = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) bm = fmaxf(bm, red[w]);
  float e = (valid && bm > -INFINITY) ? fast_exp(x - bm) : 0.0f;
  float s = wave_sum(e);
  if (lane == 0) red[NW + wid] = s;
  __syncthreads();
  float bs = 0.0f;
#pragma unroll
  for (int w = 0; w < NW; ++w) bs += red[NW + w];
  out_max = bm;
  out_sum = bs;
}

// ---- single-launch log-sum-exp: block partials + "last block finishes" ---------------------------
// Every block publishes its {max, sumexp} as ONE 8-byte agent-scope (write-through, sc1) store, drains
// it, and takes a ticket; the block that draws the last ticket re-reads all granules with agent-scope
// loads (L1-bypassing) and writes out[4] = {max, sumexp, lse, lse - log K_total}.  No fences: the
// payload is a single naturally aligned granule per producer (MI355X guide, G16 form R2), and the
// ticket's atomic RMW orders after the drained store.  The ticket word lives at workspace[0], must be
// zero when the kernel starts and is reset to zero by the finishing block.
constexpr int kWsHeaderBytes = 256;  // control block in front of every workspace

GJX_DEV unsigned long long pack_f2(float a, float b) {
  return ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
}

template <int THREADS>
GJX_DEV void lse_publish_and_finish(float bm, float bsum, unsigned long long* partials, unsigned* ticket, int nblocks,
                                    float log_k_total, float* out, float* red /* >= 2*THREADS/64 + 1 floats of LDS */) {
  constexpr int NW = THREADS / 64;
  __syncthreads();  // red[] may still be in use by the caller's block reduction
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], pack_f2(bm, bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[2 * NW] = (t == (unsigned)(nblocks - 1)) ? 1.0f : 0.0f;
  }
  __syncthreads();
  if (red[2 * NW] == 0.0f) return;
  float tmax = -INFINITY, tsum = 0.0f;
  for (int t = threadIdx.x; t < nblocks; t += THREADS) {
    const unsigned long long v = __hip_atomic_load(&partials[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float m = __uint_as_float((unsigned)v), sm = __uint_as_float((unsigned)(v >> 32));
    const float nm = fmaxf(tmax, m);
    if (nm > -INFINITY) tsum = tsum * fast_exp(tmax - nm) + sm * fast_exp(m - nm);
    tmax = nm;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float wm = wave_max(tmax);
  const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);
  if (lane == 0) { red[wid] = wm; red[NW + wid] = ws; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
    float sm = 0.0f;
    for (int w = 0; w < NW; ++w) sm += m > -INFINITY ? red[NW + w] * fast_exp(red[w] - m) : 0.0f;
    const float lse = m > -INFINITY ? m + logf(sm) : -INFINITY;
    out[0] = m; out[1] = sm; out[2] = lse; out[3] = lse - log_k_total;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- arguments of a generated per-program kernel (gjx_codegen.hip emits `extern "C" __global__ void gjx_gen(GenArgs)`) ----
struct GenArgs {
  const float* tab;
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
  // GJX_MODE_INPUT sites (gjx_run_program_ex): in_rows[(obs_off + d) * in_stride + ancestor(i)], or their rows of choices
  const float* in_rows;
  int64_t in_stride;
  const int32_t* anc;
  int store_inputs;
  // with lse == NULL: {S_b, e_b} of every 1024-particle tile of logw (tile-scaled fixed point) for gjx_resample_gather_tiled
  unsigned long long* tile_S;
  int32_t* tile_E;
  // gjx_run_resample: the tile-scaled systematic search over the PREVIOUS collection in the prologue (rs_logw != NULL)
  const float* rs_logw;
  const unsigned long long* rs_S;
  const int32_t* rs_E;
  const float* rs_lse;
  int rs_n_partials;
  float* rs_lse_out;
  double rs_u;
  int32_t* rs_anc_out;
  unsigned* rs_ctrl;
  // steps kernel (gjx_gen_steps: every step of a filter run in ONE launch).  st_tag != 0: rows, log-weights and the block pair
  // are stored write-through at agent scope and the tile's {e_b, S_b} goes out LAST, as a granule tagged st_tag in tile_S[tix]
  // (tile_E unused); the resampling side polls rs_S for granules tagged st_rtag and reads the previous step at agent scope
  unsigned long long st_tag, st_rtag;
  unsigned long long* tl;    // debug (gjx_debug_timeline): 16 realtime stamps per tile of one step of the steps kernel, or NULL
};

// ---- arguments of `gjx_gen_steps`: steps T0 .. T-1 of a filter whose step programs share one structure (a periodic Scan) ----
struct GenStepsArgs {
  GenArgs base;                       // what every step shares: K, log_k_total, ... (per-step fields are filled in by the kernel)
  int T0, T;
  const float* const* tabs;           // [T] the step programs' tables (they differ in their observations)
  const uint32_t* keys;               // [T][2]
  const double* us;                   // [T]
  float* rows_a; float* rows_b;       // choices of even / odd steps ...
  float* rows_all; int64_t rows_step; // ... or, when the run is recorded, step t at rows_all + t * rows_step
  int64_t in_row0_first, in_row0;     // floats in front of the OWN rows of step T0-1 / of the later steps in their buffer
  float* logw_a; float* logw_b;       // log-weights of step t in logw_b when (T - 1 - t) is odd
  unsigned long long* gran_a; unsigned long long* gran_b;   // [K / 1024] granules of even / odd steps
  unsigned long long* part_a; unsigned long long* part_b;   // [K / 1024] block pairs {max, sumexp} of even / odd steps
  float* lse_steps;                   // [T][4]
  int32_t* anc; int32_t* anc_all;     // ancestors of the last resampling / of every resampling [T-1][K]
  unsigned* ctrl;                     // control block of a workspace (status word at [2])
  unsigned long long* timeline;       // debug: stamps of step T / 2 (profiles/microbench/scan_steps_timeline.py)
  unsigned epoch;                     // granule tags: (epoch + t) % 15 + 1
};

// ---- arguments of a generated per-program HMC kernel (gjx_codegen.hip emits `extern "C" __global__ void gjx_hmc_gen(HmcGenArgs)`) ----
struct HmcGenArgs {
  const float* tab;
  key2 key;
  int64_t n, offset;
  float eps;
  int L, stale, accept;
  float* choices;   // [n_slots][n] in / out
  float* score;     // [n] or NULL
  float* alpha;     // [n] or NULL
  float* accepted;  // [n] or NULL
  float* ws;        // [4][PROWS][n]: positions, momenta, gradients, first gradients of the selected sites INSIDE plates (else unused)
  int64_t ws_floats;
};

// ---- analytic gradients of the element log-densities (gjx_hmc.hip, generated HMC kernels) ----
// gradient of elem_logpdf w.r.t. the value and the parameters; g[0..3] = d/d(a, b, c, d).  Shape parameters (gamma / beta
// concentrations, student-t / chi2 degrees of freedom, inverse-gamma concentration) go through digamma, so HMC over
// hierarchical shape parameters works as it does under jax.grad (hmc.py:70-96).
GJX_DEV void dlogpdf(int kind, float x, float a, float b, float c, float d, float& dx, float* gpar) {
  float da = 0.0f, db = 0.0f, dc = 0.0f, dd = 0.0f;
  dx = 0.0f;
  auto done = [&]() { gpar[0] = da; gpar[1] = db; gpar[2] = dc; gpar[3] = dd; };
  switch (kind) {
    case GJX_TRUNCATED_CAUCHY: {  // a = loc, b = scale, c = low, d = high
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb, lo = (c - a) * rb, hi = (d - a) * rb;
      const float rA = fast_rcp(atanf(hi) - atanf(lo)) * rb;
      const float wl = fast_rcp(fmaf(lo, lo, 1.0f)) * rA, wh = fast_rcp(fmaf(hi, hi, 1.0f)) * rA;
      const float w = 2.0f * z * fast_rcp(fmaf(z, z, 1.0f)) * rb;
      dx = -w; da = w + (wh - wl); db = fmaf(w, z, -rb) + (hi * wh - lo * wl); dc = wl; dd = -wh;
      done(); return;
    }
    case GJX_NEGATIVE_BINOMIAL: {
      const float p = sigmoid(b);
      da = digamma_f(x + a) - digamma_f(a) - softplus(b); db = x * (1.0f - p) - a * p;
      done(); return;
    }
    case GJX_VON_MISES: {
      const float sn = sinf(x - a);
      dx = -b * sn; da = b * sn; db = cosf(x - a) - bessel_i1_over_i0(b);
      done(); return;
    }
    case GJX_CHI: dx = (a - 1.0f) * fast_rcp(x) - x; da = fast_log(x) - 0.5f * kLn2 - 0.5f * digamma_f(0.5f * a); done(); return;
    case GJX_EXP_GAMMA: {
      const float e = fast_exp(x);
      dx = fmaf(-b, e, a); da = fast_log(b) + x - digamma_f(a); db = a * fast_rcp(b) - e;
      done(); return;
    }
    case GJX_EXP_INVERSE_GAMMA: {
      const float e = fast_exp(-x);
      dx = fmaf(b, e, -a); da = fast_log(b) - x - digamma_f(a); db = a * fast_rcp(b) - e;
      done(); return;
    }
    case GJX_KUMARASWAMY: {
      const float lx = fast_log(x), xa = fast_exp(a * lx);
      const float r = xa * fast_rcp(1.0f - xa);                                  // x^a / (1 - x^a)
      dx = ((a - 1.0f) - (b - 1.0f) * a * r) * fast_rcp(x);
      da = fast_rcp(a) + lx * (1.0f - (b - 1.0f) * r);
      db = fast_rcp(b) + log1p_acc(-xa);
      done(); return;
    }
    case GJX_MOYAL: {
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb;
      const float e1 = 0.5f * (1.0f - fast_exp(-z));
      dx = -e1 * rb; da = e1 * rb; db = (e1 * z - 1.0f) * rb;
      done(); return;
    }
    case GJX_DOUBLESIDED_MAXWELL: {
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb;
      const float w = (2.0f * fast_rcp(z) - z) * rb;
      dx = w; da = -w; db = (z * z - 3.0f) * rb;
      done(); return;
    }
    case GJX_INVERSE_GAUSSIAN: {
      const float ra = fast_rcp(a), rx = fast_rcp(x);
      const float r = (x - a) * ra;
      dx = -1.5f * rx - 0.5f * b * (x * x - a * a) * ra * ra * rx * rx;
      da = b * r * ra * ra;
      db = 0.5f * fast_rcp(b) - 0.5f * r * r * rx;
      done(); return;
    }
    case GJX_HALF_STUDENT_T:
    case GJX_STUDENT_T: {  // a = df, b = loc, c = scale
      const float rc = fast_rcp(c);
      const float y = (x - b) * rc;
      // a draw with df << 1 reaches 1e30 and y^2 leaves float32 (inf / inf below): there y^2 / (df + y^2) is 1 to every bit and
      // log1p(y^2 / df) is log(y^2 / df), as in the log-density above (found by the differential test: a NaN where the oracle has 1.2)
      const bool big = fabsf(y) > 1e18f;
      const float yy = y * y;
      const float w = big ? (a + 1.0f) * fast_rcp(y) : (a + 1.0f) * y * fast_rcp(a + yy);
      const float wy = big ? a + 1.0f : w * y;
      dx = -w * rc; db = w * rc; dc = (wy - 1.0f) * rc;
      const float l1p = big ? 2.0f * fast_log(fabsf(y)) - fast_log(a) : log1p_acc(yy * fast_rcp(a));
      const float t2 = big ? 0.5f * (a + 1.0f) * fast_rcp(a) : 0.5f * (a + 1.0f) * yy * fast_rcp(a * (a + yy));
      da = -0.5f * l1p + t2 - 0.5f * fast_rcp(a) + 0.5f * digamma_half_step(0.5f * a);
      done(); return;
    }
    case GJX_TRUNCATED_NORMAL: {  // a = loc, b = scale, c = low, d = high
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb, lo = (c - a) * rb, hi = (d - a) * rb;
      const float rZ = fast_rcp(normal_interval_mass(lo, hi));
      const float plo = 0.39894228f * fast_exp(-0.5f * lo * lo) * rZ * rb, phi = 0.39894228f * fast_exp(-0.5f * hi * hi) * rZ * rb;
      dx = -z * rb;
      da = z * rb + (phi - plo);
      db = (z * z - 1.0f) * rb + (hi * phi - lo * plo);
      dc = plo; dd = -phi;
      done(); return;
    }
    case GJX_POISSON: da = x * fast_rcp(a) - 1.0f; done(); return;
    case GJX_GEOMETRIC: da = fast_rcp(a) - x * fast_rcp(1.0f - a); done(); return;
    case GJX_GUMBEL: {
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb;
      const float e1 = 1.0f - fast_exp(-z);
      dx = -e1 * rb; da = e1 * rb; db = (e1 * z - 1.0f) * rb;
      done(); return;
    }
    case GJX_HALF_CAUCHY: {
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb;
      const float gq = 2.0f * z * fast_rcp(1.0f + z * z);
      dx = -gq * rb; da = gq * rb; db = (gq * z - 1.0f) * rb;
      done(); return;
    }
    case GJX_INVERSE_GAMMA: {
      const float rx = fast_rcp(x);
      dx = -(a + 1.0f) * rx + b * rx * rx; db = a * fast_rcp(b) - rx; da = fast_log(b) - digamma_f(a) - fast_log(x);
      done(); return;
    }
    case GJX_WEIBULL: {
      const float lr = fast_log(x * fast_rcp(b));
      const float t = fast_exp(a * lr);
      dx = ((a - 1.0f) - a * t) * fast_rcp(x); db = a * (t - 1.0f) * fast_rcp(b); da = fast_rcp(a) + lr * (1.0f - t);
      done(); return;
    }
    case GJX_LOGIT_NORMAL: {
      const float rb = fast_rcp(b);
      const float z = (fast_log(x) - log1p_acc(-x) - a) * rb;
      dx = -z * rb * fast_rcp(x * (1.0f - x)) - fast_rcp(x) + fast_rcp(1.0f - x); da = z * rb; db = (z * z - 1.0f) * rb;
      done(); return;
    }
    case GJX_CHI2: dx = (0.5f * a - 1.0f) * fast_rcp(x) - 0.5f; da = 0.5f * (fast_log(x) - kLn2 - digamma_f(0.5f * a)); done(); return;
    default: break;
  }
  switch (kind) {
    case GJX_NORMAL:
    case GJX_MVNORMAL_DIAG: {
      const float rb = fast_rcp(b);
      const float z = (x - a) * rb;
      dx = -z * rb; da = z * rb; db = (z * z - 1.0f) * rb;
      break;
    }
    case GJX_BERNOULLI_LOGITS: da = x - sigmoid(a); break;
    case GJX_FLIP: da = (x != 0.0f ? fast_rcp(a) : 0.0f) - (x != 1.0f ? (1.0f - x) * fast_rcp(1.0f - a) : 0.0f); break;
    case GJX_HALF_NORMAL: { const float ra = fast_rcp(a); const float z = x * ra; dx = -z * ra; da = (z * z - 1.0f) * ra; break; }
    case GJX_EXPONENTIAL: dx = -a; da = fast_rcp(a) - x; break;
    case GJX_LAPLACE: { const float s = (float)((x > a) - (x < a)); const float rb = fast_rcp(b); dx = -s * rb; da = s * rb; db = fabsf(x - a) * rb * rb - rb; break; }
    case GJX_CAUCHY: { const float rb = fast_rcp(b); const float z = (x - a) * rb; const float g = 2.0f * z * fast_rcp(1.0f + z * z); dx = -g * rb; da = g * rb; db = (g * z - 1.0f) * rb; break; }
    case GJX_LOG_NORMAL: { const float lx = fast_log(x); const float rb = fast_rcp(b); const float z = (lx - a) * rb; dx = (-z * rb - 1.0f) * fast_rcp(x); da = z * rb; db = (z * z - 1.0f) * rb; break; }
    case GJX_BETA: {
      const float pab = digamma_f(a + b);
      dx = (a - 1.0f) * fast_rcp(x) - (b - 1.0f) * fast_rcp(1.0f - x);
      da = fast_log(x) - digamma_f(a) + pab; db = log1p_acc(-x) - digamma_f(b) + pab;
      break;
    }
    case GJX_GAMMA: dx = (a - 1.0f) * fast_rcp(x) - b; da = fast_log(b) + fast_log(x) - digamma_f(a); db = a * fast_rcp(b) - x; break;
    case GJX_UNIFORM: { const float r = fast_rcp(b - a); da = r; db = -r; break; }
    default: break;
  }
  done();
}

GJX_DEV float xf_deriv(int xf, float pre) {
  switch (xf) {
    case GJX_XF_EXP: return fast_exp(pre);
    case GJX_XF_SOFTPLUS: return sigmoid(pre);
    case GJX_XF_SIGMOID: { const float s = sigmoid(pre); return s * (1.0f - s); }
    default: return 1.0f;
  }
}

// sum over the 4 lanes of an aligned lane quad, in every lane of it
GJX_DEV float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// ---- systematic-resampling comb on the fixed-point weight line (shared by gjx_resample.hip / gjx_shard.hip) ----
// comb threshold of output slot j (identical double arithmetic in the oracle)
GJX_DEV uint64_t comb_threshold(int64_t j, double u, double step, uint64_t total) {
  uint64_t T = (uint64_t)(((double)j + u) * step);
  if (total > 0 && T > total - 1) T = total - 1;
  return T;
}

// ---- multinomial resampling by SORTED uniforms (include/gjx.h GJX_FILTER_MULTINOMIAL in the one-launch filter): the j-th smallest of N
// independent uniforms is S_j / S_{N+1}, S_j = e_0 + ... + e_j, for independent exponential spacings e_i.  A spacing is -log2(u_i) in
// units of 2^-20, u_i = (2 m + 1) / 2^24 with m the top 23 bits of the slot's word — computed in float32 steps that round identically
// on the device and in the oracle (integer -> float, a power-of-two scale, a Horner chain of fmaf, float -> integer): the sums are
// exact integers, so the sorted uniforms and with them the ancestors do not depend on launch geometry, sharding or the math library.
GJX_DEV uint64_t exp_spacing(uint32_t word) {
  const uint32_t x = ((word >> 9) << 1) | 1u;                    // odd, < 2^24: exact in float32
  int k = 31 - __builtin_clz(x);                                 // x = 2^k f, f in [1, 2)
  float f = ldexpf((float)x, -k);
  if (f > 1.41421354f) { f *= 0.5f; k += 1; }                    // f in (sqrt 1/2, sqrt 2]
  const float t = f - 1.0f;
  float q = 0.12614846229553223f;                                // log2(1 + t) = t Q(t), |error| < 7e-8 on the interval
  q = fmaf(q, t, -0.20742103457450867f);
  q = fmaf(q, t, 0.21566985547542572f);
  q = fmaf(q, t, -0.23892034590244293f);
  q = fmaf(q, t, 0.2879183292388916f);
  q = fmaf(q, t, -0.36070483922958374f);
  q = fmaf(q, t, 0.48091059923171997f);
  q = fmaf(q, t, -0.7213473320007324f);
  q = fmaf(q, t, 1.4426950216293335f);
  const float e = fmaf(-q, t, (float)(24 - k));                  // -log2(x / 2^24) >= 0
  return e > 0.0f ? (uint64_t)(e * 1048576.0f) : 0ull;
}
// threshold of the slot whose inclusive spacing sum is Sj: floor(U_(j) total) on the weight line, non-decreasing in j
GJX_DEV uint64_t sorted_threshold(uint64_t Sj, uint64_t Sall, uint64_t total) {
  uint64_t T = (uint64_t)((double)Sj * ((double)total / (double)Sall));
  if (total > 0 && T > total - 1) T = total - 1;
  return T;
}

// number of output slots whose comb threshold lies strictly below c (0 <= c <= total):
// J(c) = #{j in [0, N) : T_j < c}.  T_j is non-decreasing in j, so J is found from the real-valued guess
// ceil(c/step - u) and corrected with the EXACT integer predicate (the same T_j the per-slot search uses).
GJX_DEV int64_t slots_below(uint64_t c, double u, double step, double inv_step, uint64_t total, int64_t N) {
  if (c == 0) return 0;
  if (c >= total) return N;
  double gd = ceil((double)c * inv_step - u);  // any guess works: the loops below make the result exact
  int64_t g = gd < 0.0 ? 0 : (gd > (double)N ? N : (int64_t)gd);
  while (g > 0 && comb_threshold(g - 1, u, step, total) >= c) --g;
  while (g < N && comb_threshold(g, u, step, total) < c) ++g;
  return g;
}

}  // namespace gjx
