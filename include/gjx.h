/* gjx.h — C ABI of the MI355X-native inference-kernel backend (libgjx_hip.so).
 *
 * This is the drop-in boundary for the vmapped hot path of genjax.inference.smc and the
 * per-chain MCMC step.  The reference (/root/reference, pure Python on JAX) has no FFI; the
 * seams these entry points sit behind are its abstract Python interfaces, cited per function
 * as  <file>:<line>  relative to  /root/reference/src/genjax/_src/ .
 *
 * Conventions
 *   - return 0 (GJX_OK) on success, a negative gjx_status otherwise; gjx_last_error() gives the
 *     thread-local message of the last failure.
 *   - the library never allocates or frees caller-visible memory: the caller owns every buffer
 *     (torch tensors in the Python host layer); scratch is sized by gjx_workspace_bytes().  The only
 *     device memory the library itself holds belongs to the two opaque contexts (gjx_shard_ctx,
 *     gjx_peer_ctx): allocated by their create call, released by their destroy call, never in between.
 *   - every `*_dev` / output pointer is a DEVICE pointer; every call is asynchronous on the
 *     given hipStream_t (passed as void*) and re-entrant.  What the library keeps between calls:
 *     the thread-local error string, the caches of generated kernels (compiled code, keyed by
 *     program structure) and the buffer registered by gjx_debug_timeline (profiling scripts).
 *     Options travel as arguments (gjx_run_opts, gjx_filter_opts, the flags of
 *     gjx_peer_ctx_create_ex); the environment variables that remain select engines for
 *     experiments (GJX_ENGINE, GJX_HMC_ENGINE, GJX_GEN_*, GJX_HMC_GEN_*) and are read per call.
 *   - particle/chain state is SoA: choices[slot][K] — one contiguous f32[K] row per scalar of a
 *     random choice, so a 64-lane wavefront reads/writes 256 contiguous bytes per row.
 *   - all floating values are float32 (the reference's default, core/typing.py:42); integer and
 *     boolean choices (categorical index, flip) are stored as exact float32 values.
 *   - PRNG keys are Threefry-2x32 keys (two uint32 words), jax.random.key(seed) == (0, seed).
 */
#ifndef GJX_H
#define GJX_H

#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>
#else   /* hipRTC has no libc headers: its runtime header keeps the fixed-width types in a namespace */
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::int64_t int64_t;
typedef __hip_internal::uint64_t uint64_t;
typedef __hip_internal::uint8_t uint8_t;
typedef unsigned long size_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GJX_ABI_VERSION 10

typedef enum gjx_status {
  GJX_OK = 0,
  GJX_EINVAL = -1,       /* bad argument (null pointer, negative size, malformed program) */
  GJX_EUNSUPPORTED = -2, /* program uses a form this entry point cannot run */
  GJX_EHIP = -3,         /* HIP runtime error (launch failure, no device) */
  GJX_EWORKSPACE = -4    /* workspace too small */
} gjx_status;

/* ---- model program ---------------------------------------------------------------------
 * A generative function in the reference is a Python function traced to a Jaxpr and
 * interpreted site by site (generative_functions/static.py:340-399).  Here it is an explicit
 * straight-line list of sites; "user code between sites" is one of the parameter
 * expression forms below: five closed forms the engines have fast paths for, and
 * GJX_P_EXPR, a general elementwise expression block.
 */

/* primitive distribution kinds — generative_functions/distributions/tensorflow_probability/__init__.py */
enum {
  GJX_NORMAL = 1,             /* normal(loc, scale)            :259  (scale = std-dev)      */
  GJX_FLIP = 2,               /* flip(p)  Bernoulli(probs=p)   :155                         */
  GJX_BERNOULLI_LOGITS = 3,   /* bernoulli(logits)             :72                          */
  GJX_BETA = 4,               /* beta(a, b)                    :82                          */
  GJX_CATEGORICAL_LOGITS = 5, /* categorical(logits)           :102-104                     */
  GJX_CATEGORICAL_PROBS = 6,  /* categorical(probs=...)        :102-104                     */
  GJX_UNIFORM = 7,            /* uniform(low, high)            :294                         */
  GJX_MVNORMAL_DIAG = 8,      /* mv_normal_diag(loc, scale)    :239  (one site, dim scalars)*/
  GJX_EXPONENTIAL = 9,        /* exponential(rate)             :150                         */
  GJX_HALF_NORMAL = 10,       /* half_normal(scale)            :184                         */
  GJX_LAPLACE = 11,           /* laplace(loc, scale)           :214                         */
  GJX_LOG_NORMAL = 12,        /* log_normal(loc, scale)        :219                         */
  GJX_CAUCHY = 13,            /* cauchy(loc, scale)            :110                         */
  GJX_GAMMA = 14,             /* gamma(concentration, rate)    :164                         */
  GJX_STUDENT_T = 15,         /* student_t(df, loc, scale)     :279                         */
  GJX_TRUNCATED_NORMAL = 16,  /* truncated_normal(loc, scale, low, high) :289               */
  GJX_POISSON = 17,           /* poisson(rate)                 :264  (value stored as float) */
  GJX_GEOMETRIC = 18,         /* geometric(probs)              :169  (failures before success)*/
  GJX_DIRICHLET = 19,         /* dirichlet(concentration[dim]) :125  (one site, dim scalars) */
  GJX_GUMBEL = 20,            /* gumbel(loc, scale)            :174                         */
  GJX_HALF_CAUCHY = 21,       /* half_cauchy(loc, scale)       :179                         */
  GJX_INVERSE_GAMMA = 22,     /* inverse_gamma(concentration, scale) :194                   */
  GJX_WEIBULL = 23,           /* weibull(concentration, scale) :309                         */
  GJX_LOGIT_NORMAL = 24,      /* logit_normal(loc, scale)      :224                         */
  GJX_CHI2 = 25,              /* chi2(df)                      :120                         */
  GJX_CHI = 26,               /* chi(df)                       :115                         */
  GJX_EXP_GAMMA = 27,         /* exp_gamma(concentration, rate): log of a gamma variate :140  */
  GJX_EXP_INVERSE_GAMMA = 28, /* exp_inverse_gamma(concentration, scale): log of an inverse-gamma variate :145 */
  GJX_HALF_STUDENT_T = 29,    /* half_student_t(df, loc, scale) :189                        */
  GJX_KUMARASWAMY = 30,       /* kumaraswamy(concentration1, concentration0) :204           */
  GJX_MOYAL = 31,             /* moyal(loc, scale)             :229                         */
  GJX_TRUNCATED_CAUCHY = 32,  /* truncated_cauchy(loc, scale, low, high) :284               */
  GJX_DOUBLESIDED_MAXWELL = 33, /* double_sided_maxwell(loc, scale) :135                    */
  GJX_INVERSE_GAUSSIAN = 34,  /* inverse_gaussian(loc, concentration) :199                  */
  GJX_NEGATIVE_BINOMIAL = 35, /* negative_binomial(total_count, logits) :249  (successes before total_count failures; value stored as float) */
  GJX_VON_MISES = 36,         /* von_mises(loc, concentration) :299                         */
  GJX_KIND_MAX = 37
};

/* parameter expression forms (what the model body computes between sites) */
enum {
  GJX_P_CONST = 0,  /* tab[off + (d % len)]                                               */
  GJX_P_VALUE = 1,  /* choices[slot + (d % len)][i]                                       */
  GJX_P_GATHER = 2, /* tab[off + clamp((int)choices[slot][i], 0, n-1) * len + (d % len)]  */
  GJX_P_AFFINE = 3, /* tab[off + (d % len)] + sum_{e<n} tab[moff + d*n + e] * choices[slot+e][i] */
  GJX_P_VGATHER = 4,/* choices[moff + clamp(idx, 0, n-1) * len + (d % len)][i]: a row of an EARLIER vector-valued choice (first slot
                       `moff`, n rows of `len` values: the component means of a mixture with latent means, mu[z]) picked by a discrete
                       choice — idx = (int)choices[slot][i], or, slot < 0 (the index site is constrained to one value for every
                       particle and owns no storage), idx = (int)tab[off].  Differentiable in the picked row (gjx_hmc,
                       gjx_score_grad: the gradient goes to row moff + idx * len + d % len).  Plate strides: slot + i d_slot (or
                       off + i d_off), moff + i d_moff */
  GJX_P_EXPR = 5    /* an expression block in the table: see "GJX_P_EXPR" below */
};
/* GJX_P_EXPR (ABI 9) — ANY elementwise computation between sites.  The reference stages the whole @gen body and interprets whatever
 * JAX computes between two trace sites (generative_functions/static.py:383-399, core/compiler/staging.py:286-298), and
 * selection_gradient differentiates through it (inference/requests/hmc.py:70-96).  Here such a computation is a small SSA BLOCK of
 * scalar nodes stored in the program's float table: node i is the GJX_EXPR_NODE_FLOATS = 6 floats tab[off + 6 i ..] = {op, a, b, c, da,
 * db} (small integers held exactly in float32; a / b / c are node indices WITHIN the block — always smaller than i —, slots, table
 * offsets or counts, by op; da / db are the plate strides of a / b, 0 outside plates).  gjx_param: op = GJX_P_EXPR, off = the block, n = its number of nodes (1 .. GJX_EXPR_MAX_NODES), len = number of
 * outputs: the LAST len nodes (len = 1: one value for every element; else len = the site's dim and element d reads output d % len).
 * xf still applies on top.  The node list is part of the program's STRUCTURE (generated kernels bake it in and the structure key
 * hashes it); the constants and weights it refers to are ordinary table entries and may change like any table value.
 * Plate instance i (gjx.h "Plates"): VALUE nodes read slot a + i da, CONST nodes tab[a + i da]; LINV reads its [bias, weights] at
 * tab[a + i da ...] and its slots from b + i db; LINN its [bias, weights] at tab[a + i da ...] — per-instance data (the covariates of a
 * regression with a nonlinear link, vmapped over the observations) and per-instance latents, with ONE node list for the plate.
 * Reverse mode (gjx_hmc, gjx_score_grad): the adjoint of the output flows back through the block to its VALUE leaves; CONST leaves,
 * comparisons and the condition of a WHERE carry no gradient (jax.grad of jnp.where / lax.select, comparisons). */
enum {
  GJX_E_CONST = 0,    /* tab[a + inst * da]                                                             */
  GJX_E_VALUE = 1,    /* choices[a + inst * da][i]                                                      */
  GJX_E_ADD = 2,      /* node a + node b                                                                */
  GJX_E_SUB = 3,      /* node a - node b                                                                */
  GJX_E_MUL = 4,      /* node a * node b                                                                */
  GJX_E_DIV = 5,      /* node a / node b                                                                */
  GJX_E_NEG = 6,      /* -node a                                                                        */
  GJX_E_EXP = 7, GJX_E_LOG = 8, GJX_E_SQRT = 9, GJX_E_SQUARE = 10, GJX_E_TANH = 11, GJX_E_SIGMOID = 12, GJX_E_SOFTPLUS = 13,
  GJX_E_ABS = 14, GJX_E_SIN = 15, GJX_E_COS = 16, GJX_E_LOG1P = 17, GJX_E_RECIP = 18,   /* unary, of node a                */
  GJX_E_MAX = 19,     /* max(node a, node b)   (gradient to the larger; to a at a tie)                   */
  GJX_E_MIN = 20,     /* min(node a, node b)                                                            */
  GJX_E_GT = 21,      /* node a > node b ? 1 : 0                                                        */
  GJX_E_WHERE = 22,   /* node a != 0 ? node b : node c                                                  */
  GJX_E_LINV = 23,    /* tab[a] + sum_{e < c} tab[a + 1 + e] * choices[b + e][i]: bias and weights contiguous in the table */
  GJX_E_LINN = 24,    /* tab[a] + sum_{e < c} tab[a + 1 + e] * node (b + e)                              */
  GJX_E_OP_MAX = 25
};
#define GJX_EXPR_MAX_NODES 96
#define GJX_EXPR_NODE_FLOATS 6
/* unary transform applied to the evaluated parameter */
enum { GJX_XF_NONE = 0, GJX_XF_EXP = 1, GJX_XF_SOFTPLUS = 2, GJX_XF_SIGMOID = 3 };

/* how a site obtains its value in one run of the program
 * (distribution.py:117-147 generate_choice_map; static.py:340-380 GenerateHandler) */
enum {
  GJX_MODE_SAMPLE = 0,   /* unconstrained: v ~ dist, score += logpdf(v), weight += 0           */
  GJX_MODE_OBS_TAB = 1,  /* constrained, same value for every particle: v = tab[obs_off + d]   */
  GJX_MODE_OBS_SLOT = 2, /* constrained per particle: v = choices[slot + d][i] (already there) */
  GJX_MODE_OBS_MASK = 3, /* Mask(value, flag) per particle (distribution.py:129-143): flag = choices[obs_off][i];
                            flag != 0: as OBS_SLOT; flag == 0: as SAMPLE (the draw overwrites the slot).
                            Not accepted by dirichlet sites, gjx_hmc or gjx_score_grad. */
  GJX_MODE_OBS_PROPOSED = 5, /* constrained, per particle, to the value an EARLIER site of this run left in the same slot — a draw of a
                            proposal site (GJX_SITE_PROPOSAL) that this model site is then scored at (smc.py:302-313: the importance
                            step with a custom proposal q, inside ONE program).  The value is read from the slot as it stands;
                            scored like OBS_SLOT (score and weight); nothing is stored (the proposal site stored the row) */
  GJX_MODE_INPUT = 4     /* not a random choice: `dim` rows that hold a per-particle INPUT of the program — the carry a Scan step
                            receives from the step before it (scan.py:237-294), the arguments of a kernel.  The value is read
                            (from choices[slot + d][i], or through the ancestor gather of gjx_run_program_ex:
                            in_rows[obs_off + d][ancestor(i)]), never drawn and never scored; the site takes NO site number in
                            either stream layout (the sites behind it are numbered as if it were not there); kind and
                            parameters are ignored.  gjx_hmc / gjx_score_grad read it from choices[slot + d][i] like any per-chain
                            value (an HMC move of a kernel's latents given its arguments / of a Scan step given its carry): no
                            density, no gradient row, never selected. */
};

enum { GJX_SITE_HMC_SELECTED = 1, /* gjx_site.flags: site is moved by gjx_hmc (hmc.py:70-96) */
       /* the site belongs to a PROPOSAL q, not to the model (smc.py:302-313; the SMC step's proposal, scan.py:325-416 extended by one
        * step): mode GJX_MODE_SAMPLE; its draw stays in its slot (the model's site of the same slot follows with GJX_MODE_OBS_PROPOSED),
        * its log-density is SUBTRACTED from the weight and is no part of the score: log w += log p - log q — proper weighting
        * (the reference's Marginal.random_weighted returns 0 here).  site_scores row: log q.  It takes a site number
        * like any site.  Accepted by gjx_run_program[_ex] and the filters built on it; not by gjx_hmc / gjx_score_grad. */
       GJX_SITE_PROPOSAL = 2,
       /* a GJX_MODE_INPUT site whose gathered value is STORED into the site's own rows whenever it was read through in_rows
        * (as GJX_RUN_STORE_INPUTS does for every input): a value that travels with the particle from step to step of a filter — a
        * static parameter drawn in front of the Scan (GJX_FILTER_ABSOLUTE_INPUTS) */
       GJX_SITE_CARRIED = 4 };

typedef struct gjx_param {
  int32_t op;   /* GJX_P_*  */
  int32_t xf;   /* GJX_XF_* */
  int32_t off;  /* CONST: values; GATHER: table base; AFFINE: bias; VGATHER with slot < 0: the index (into tab); EXPR: the node block */
  int32_t len;  /* CONST/VALUE: vector length (1 = broadcast); GATHER/VGATHER: row length; AFFINE: bias length; EXPR: outputs */
  int32_t slot; /* VALUE/AFFINE: first source slot; GATHER/VGATHER: slot holding the index     */
  int32_t n;    /* AFFINE: inner length; GATHER/VGATHER: number of rows; EXPR: number of nodes */
  int32_t moff; /* AFFINE: matrix [dim][n] row-major (into tab); VGATHER: first SLOT of the indexed choice */
  /* plate strides (sites with gjx_site.plate != 0, see "Plates" below): instance i evaluates the parameter with
   * off + i * d_off, slot + i * d_slot, moff + i * d_moff.  All 0 outside plates and for what the instances share. */
  int32_t d_off, d_slot, d_moff;
  int32_t pad_[2];
} gjx_param; /* 48 bytes */

#define GJX_MAX_PARAMS 4

typedef struct gjx_site {
  int32_t kind;    /* GJX_NORMAL ...                                                        */
  int32_t dim;     /* event size: number of scalar slots this site owns (1 for scalars)     */
  int32_t slot;    /* first slot of the value in choices[n_slots][K]                        */
  int32_t mode;    /* GJX_MODE_*                                                            */
  int32_t obs_off; /* OBS_TAB: offset of the observed value in tab                          */
  int32_t ncat;    /* categorical: number of categories (length of params[0])               */
  int32_t flags;   /* GJX_SITE_*                                                            */
  int32_t scan;    /* 0: not inside a Scan; else (scan_id << 20) | (step + 1): see "Scan steps" below */
  int32_t plate;   /* 0: not inside a plate; else the plate's id (from 1): see "Plates" below          */
  int32_t plate_n; /* number of instances of the plate                                                */
  int32_t d_obs;   /* plate stride of obs_off: OBS_TAB dim (one observed value per instance), OBS_MASK 1 (one flag row per
                      instance)                                                                       */
  int32_t pad_;
  gjx_param p[GJX_MAX_PARAMS];
} gjx_site; /* 240 bytes */

/* Plates (combinators/vmap.py:180-218: Vmap.simulate / generate over n instances of a kernel generative function).
 * The m sites of a kernel body that is vmapped over n instances are m CONSECUTIVE sites that carry the same `plate` id and
 * `plate_n` = n; the engines run ONE instance loop over them:
 *     for i in [0, n):  for each body site, in order:  the site with
 *         value rows      slot + i * dim .. slot + (i + 1) * dim - 1     (site-major: a body site owns n * dim rows;
 *                                                                        categorical sites: dim = 1)
 *         observed value  tab[obs_off + i * d_obs ...]   (OBS_TAB)  /  flag row obs_off + i * d_obs   (OBS_MASK)
 *         parameters      off + i * d_off,  slot + i * d_slot,  moff + i * d_moff
 * A parameter reads an earlier site of the SAME instance with d_slot = that site's dim, a site outside the plate with
 * d_slot = 0, instance i of an EARLIER plate with that plate's dim; tables are shared (d_off = 0) or stacked per instance.
 * The score of a body site is the sum over its instances (vmap.py:206-218 sums the instances' weights); site_scores holds
 * one row per body site.  The body counts m — not n m — towards the site numbers of the FLAT layout.
 * Streams.  GJX_RNG_FLAT: a body site has ONE site number (its position, like any other site) and instance i draws at the
 *     elements a vector site of n * dim elements would use: element (i * dim + c) * draws_per_element + k; categorical
 *     sites: element i.  Body sites never join scalar-normal runs (a drawing body site closes an open run).
 *   GJX_RNG_JAX32 — the reference's key rule:  plate key P = fold_in(particle key, J), J = the caller's site counter at
 *     the Vmap call: the number of sites traced before it, plus one, where a whole plate counts as ONE site (the Vmap call
 *     is one traced site of its caller, static.py:349-352: the site behind a plate of m body sites has counter J + 1);
 *     instance key = Threefry(P, (0, i)) = jax.random.split(P, n)[i] (vmap.py:186, 201);  site key = fold_in(instance key,
 *     l), l = 1-based position of the site in the body (static.py:349-352 inside the kernel); elements from 0 per instance.
 * A plate may sit inside a Scan step (its sites carry both tags); plates do not nest on the device (the host unrolls the
 * outer one). */

/* Scan steps (combinators/scan.py:237-294).  The sites of step t of a Scan are consecutive and carry the same `scan`
 * tag.  Their random streams follow the reference's CHAINED key rule (scan.py:268, key_t = fold_in(key_{t-1}, t)):
 *   base   = fold_in(run key, 0x80000000 | scan_id)       (the key the Scan call itself receives)
 *   key_t  = fold_in(key_{t-1}, t),  key_{-1} = base       (fold_in(k, i) = Threefry(k, (0, i)))
 * and site number = position of the site within its step, from 1 — so a Scan of any length never runs out of the
 * 1023 site numbers of the FLAT layout; the site loop is a device loop over the steps' descriptors.  Sites outside
 * any Scan use the run key and their position among the non-Scan sites, from 1.  (GJX_RNG_JAX32 keeps
 * fold_in(particle key, global site index) for every site.) */
#define GJX_SCAN_TAG(scan_id, step) ((int32_t)(((uint32_t)(scan_id) << 20) | (uint32_t)((step) + 1)))
#define GJX_SCAN_ID(tag) ((uint32_t)(tag) >> 20)
#define GJX_SCAN_STEP(tag) ((int32_t)((uint32_t)(tag) & 0xFFFFFu) - 1)

/* random-stream layouts.  Both are Threefry-2x32-20 counter streams and both give results that are
 * independent of how particles are sharded over GPUs (the counter carries the GLOBAL particle index).
 *   GJX_RNG_FLAT  (default, the MI355X-first layout): no per-particle or per-site key derivation, and no
 *       random bit is thrown away.  The stream of (particle i, site j (1-based)) is the concatenation of the
 *       64-bit blocks Threefry(key, (i, (j << 22) | h)), h = 0, 1, ... read as 32-bit words (x0 first); every
 *       draw consumes 23 bits (a float32 mantissa): element c is the 32-bit window starting at stream bit
 *       23*c, of which the consumers use the top 23 bits, i.e. stream bits [23c + 9, 23c + 32).
 *                                                              [i < 2^32, j < 1024, h < 2^22]
 *       -> 1 hash per 2.78 draws (16 normals = 6 hashes), key schedule wave-uniform.
 *   GJX_RNG_JAX32 (the reference's layout, jax 0.5.2 with jax_threefry_partitionable=True):
 *       particle key = Threefry(key, (0, i))   (jax.random.split(key, K)[i], smc.py:300)
 *       site key     = Threefry(particle key, (0, j))          (fold_in(key, counter), static.py:349-352)
 *       bits(c)      = x0 ^ x1 of Threefry(site key, (0, c))   (prng._threefry_random_bits_partitionable)
 *       -> 1 hash per 32-bit draw plus 1 + n_sites hashes per particle. */
enum { GJX_RNG_FLAT = 0, GJX_RNG_JAX32 = 1 };
#define GJX_FLAT_SITE_SHIFT 22
#define GJX_FLAT_MAX_SITES 1023
/* Scalar-normal runs (GJX_RNG_FLAT only).  A model written as a chain of scalar normal sites (x_t ~ normal(x_{t-1}, s),
 * one address per step) would spend one hash and one half-used Box-Muller pair per site.  Instead, walking the sites in
 * order, the sampled scalar normal sites (kind GJX_NORMAL, dim 1, mode GJX_MODE_SAMPLE) form RUNS: a run starts at such
 * a site when none is open; every following such site joins it; sites that draw nothing (OBS_TAB, OBS_SLOT) are
 * transparent; any other drawing site, a change of the Scan tag, or GJX_FLAT_RUN_MAX members close it.  All members of
 * a run read the stream of the run's HEAD (its key and site number): member number e (from 0) takes stream element e,
 * so members 2k and 2k+1 are the two outputs of ONE Box-Muller evaluation and 16 members cost 6 hashes — exactly the
 * stream of a 16-wide mv_normal_diag site at the head's position.  A run of one member is the plain per-site stream.
 * The site numbers of the members are not reused.  (Interpreter, generated kernels and the oracle walk the same rule:
 * SiteStreamWalk::run_elem in csrc/gjx_device.h, run_particle in oracle/gjx_oracle.c.) */
#define GJX_FLAT_RUN_MAX 32
#define GJX_FLAT_JOINS(rng_mode, kind, dim, mode) ((rng_mode) == GJX_RNG_FLAT && (kind) == GJX_NORMAL && (dim) == 1 && (mode) == GJX_MODE_SAMPLE)

typedef struct gjx_program {
  int32_t n_sites;
  int32_t n_slots;          /* rows of choices[][] */
  int32_t n_tab;            /* floats in tab */
  int32_t rng_mode;         /* GJX_RNG_* */
  const gjx_site* sites;    /* HOST copy of the site list (launcher inspects it) */
  const gjx_site* sites_dev;/* DEVICE copy of the same bytes */
  const float* tab;         /* HOST copy of the float table (constants, args, observations) */
  const float* tab_dev;     /* DEVICE copy of the same floats */
  const float* aux_dev;     /* DEVICE: constants derived from tab by gjx_program_prepare (or NULL) */
  int32_t n_aux;            /* floats in aux_dev */
  int32_t uid;              /* 0, or a caller-chosen id that names THIS site list (and n_tab, rng_mode) for the lifetime of the
                             * process: lets the library cache its per-program analysis (engine choice, kernel lookup) instead of
                             * walking the site list on every call — matters for long Scans.  Table VALUES may change under one id. */
} gjx_program;

/* ---- library ---------------------------------------------------------------------------- */
int gjx_version(void);
/* Threefry-2x32-20 of one counter on the HOST (no device work): (x0 << 32) | x1.  jax.random.fold_in(k, i) ==
 * split(k, n)[i] == gjx_host_threefry2x32(k0, k1, i >> 32, i) (jax/_src/prng.py, jax_threefry_partitionable=True) — the
 * scalar key operations of the inference drivers (smc.py:154,299; scan.py:268). */
uint64_t gjx_host_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1);
const char* gjx_last_error(void);
/* which engine a program will run on: 0 = generic site interpreter, >0 = id of a fused kernel */
int gjx_program_engine(const gjx_program* prog);
/* Constants a fused kernel would otherwise recompute in every block (log-softmax of constant logits, running CDF,
 * sums of log sigma, reciprocal scales, ...) are computed ONCE per program: the reference gets the same effect from
 * XLA's constant folding of the traced model body (core/compiler/staging.py:286-298).  gjx_program_aux_floats
 * returns how many floats the program's engine wants (0: none); gjx_program_prepare fills a caller-owned device
 * buffer of that size from tab_dev (call it again after changing tab_dev); the program then carries the buffer in
 * aux_dev / n_aux.  A program without aux_dev still runs (generic interpreter). */
/* Generated kernels (gjx_codegen.hip): the HIP source emitted for a program (returns its length; copies at most
 * cap - 1 characters), and compilation without a launch — hipRTC cross-compiles for gfx950 without a GPU, so a build
 * step can fill the on-disk cache (GJX_JIT_CACHE, default jit_cache/ next to the library).  ppt = particles per lane. */
int64_t gjx_program_source(const gjx_program* prog, int32_t ppt, char* out, int64_t cap);
int gjx_program_precompile(const gjx_program* prog, int32_t ppt);
/* the same for the HMC kernel generated from the program's site list (gjx_hmc engine 4: chain values, gradient and momenta
 * in registers, analytic gradient sweep, L leapfrog steps and the accept in one launch): its HIP source, and a compile
 * without launch.  GJX_EUNSUPPORTED when the emitter does not cover the program (the site interpreter runs it). */
int64_t gjx_program_hmc_source(const gjx_program* prog, char* out, int64_t cap);
int gjx_program_hmc_precompile(const gjx_program* prog);
/* what the generated-kernel caches did in this process so far: out4 = {kernels compiled by hipRTC, code objects read from the on-disk
 * cache, microseconds spent in hipRTC, kernel structures known}.  A deployment (and bench.py's line: jit_compiles_at_runtime) wants the
 * first to stay 0: build() precompiles the programs of genjax_amd/jit_manifest.py into the cache that ships with the library. */
int gjx_jit_stats(int64_t* out4);
int gjx_program_aux_floats(const gjx_program* prog);
int gjx_program_prepare(const gjx_program* prog, float* aux_dev, int32_t n_aux, void* stream);

/* ---- counter-based RNG (jax.random.{split,fold_in,bits}; call sites smc.py:299-300,
 *      static.py:349-352, scan.py:213,268) ------------------------------------------------ */
/* out[i] = Threefry2x32(key, (ctr_hi, ctr_lo0 + i)), 2 words each: device uint32[n][2] */
int gjx_threefry2x32(uint32_t key0, uint32_t key1, uint32_t ctr_hi, uint32_t ctr_lo0, int64_t n,
                     uint32_t* out_dev, void* stream);

/* ---- particle propagate + reweight ------------------------------------------------------
 * Runs the program once per particle: the vmapped body of ImportanceK.run_smc
 * (inference/smc.py:298-315  ->  sp.py:83-87  ->  static.py:340-399  ->  distribution.py:117-147).
 * Particle i has GLOBAL index particle_offset + i; its draws follow prog->rng_mode (above).
 *   choices   f32[n_slots][K]  in/out (OBS_SLOT sites read their value; all sites write it back)
 *   score     f32[K]  out   sum of every site's logpdf            (static.py:102-105)
 *   weight    f32[K]  out   sum of constrained sites' logpdf      (static.py:377)
 *   logw      f32[K]  out   weight + (logw_in ? logw_in[i] : 0) - (sub ? sub[i] : 0)
 *                           smc.py:313 (sub = proposal log-density) and smc.py:383
 *                           (ChangeTarget: sub = previous score, logw_in = previous weight)
 *   site_scores f32[n_sites][K] out or NULL
 *   lse       f32[4]  out or NULL: {max, sum exp(logw-max), logsumexp, logsumexp - log(K_total)}
 *                           over THIS call's K particles (smc.py:96-97 when K_total == K)
 * `weight` or `logw` may be NULL.  workspace: gjx_workspace_bytes(GJX_OP_RUN, K).
 * With lse == NULL and a workspace, the kernel still leaves its per-block {max, sumexp} pairs at workspace + 256
 * (gjx_run_partials_count() of them) for a consumer that finishes the reduction itself (gjx_weight_cumsum, is_log 2).
 */
int gjx_run_program(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K,
                    int64_t particle_offset, float* choices, float* score, float* weight,
                    float* logw, const float* logw_in, const float* sub, float* site_scores,
                    float* lse, int64_t K_total, void* workspace, size_t workspace_bytes,
                    void* stream);

/* The same call with its options and its record as explicit arguments — no per-thread state (the one-shot setters and getters of
 * ABI 6, gjx_run_want_tiles / gjx_last_run_partials / gjx_last_run_tiles, are gone, and so is ABI 7's gjx_profile_next_run:
 * the one-launch step takes its events as arguments, gjx_importance_step_ex).
 *   opts (or NULL):
 *     flags  GJX_RUN_LEAVE_TILES   with lse == NULL: leave the {S_b, e_b} of every 1024-particle tile of logw (tile-scaled
 *                                  fixed point, below) behind the block pairs for gjx_resample_gather_tiled
 *            GJX_RUN_TIME_DISPATCH attach start_event / stop_event (gjx_event_create) to the dispatch of the propagate kernel
 *            GJX_RUN_STORE_INPUTS  with in_rows: also write the gathered inputs into their rows of choices[][]
 *     in_rows / in_stride / in_ancestors   the program's GJX_MODE_INPUT sites read in_rows[(obs_off + d) * in_stride + a(i)],
 *            a(i) = in_ancestors ? in_ancestors[i] : i — the particle gather of a resampling step (smc.py:90-91 applied to the
 *            carry) fused into the read side of the next propagate step: the resampled collection is never materialised.
 *     resample (or NULL)   the tile-scaled systematic resampling of the PREVIOUS collection (GJX_WEIGHTS_TILE_SCALED, N = K) in
 *            the prologue of this run's kernel: every block searches the ancestors of its own 1024 particles from the previous
 *            log-weights and their tile totals (what gjx_resample_gather_tiled does with rows = 0 — one body, the same
 *            ancestors bit for bit) and reads its GJX_MODE_INPUT rows through them — resampling, gather, propagate and reweight
 *            of an SMC step in ONE plain launch.  logw / tile_S / tile_E / lse_partials: what the previous run left (its logw,
 *            the tile totals and block pairs in ITS workspace — which must not be this run's workspace, nor logw this run's
 *            logw: blocks of this launch still read them while others write); u: comb offset in [0, 1); lse_out f32[4] or NULL:
 *            the previous run's LSE record, finished by one block; ancestors_out int32[K] or NULL; status_ws: a workspace whose
 *            status word takes GJX_STATUS_ZERO_TOTAL.  Needs a generated kernel that gives a lane 4 particles, INPUT sites,
 *            K % 1024 == 0, K <= 2^20 and in_rows (in_ancestors ignored); otherwise GJX_EUNSUPPORTED and nothing is launched.
 *   info_out (or NULL): n_partials = the block pairs left at workspace + 256 (the grid launched); tiles_offset = byte offset of
 *            the tile totals in the workspace (uint64 S[nt] then int32 E[nt], nt = K / 1024), 0 when none were left;
 *            engine = 0 interpreter, 1 hand-fused mixture kernel, 4 generated kernel. */
enum { GJX_RUN_LEAVE_TILES = 1, GJX_RUN_TIME_DISPATCH = 2, GJX_RUN_STORE_INPUTS = 4 };
typedef struct gjx_run_resample {
  const float* logw;               /* f32[K]: log-weights of the collection being resampled */
  const uint64_t* tile_S;          /* [K / 1024] */
  const int32_t* tile_E;           /* [K / 1024] */
  const float* lse_partials;       /* the producing run's block pairs (its workspace + 256) */
  int32_t n_partials, pad_;
  float* lse_out;                  /* f32[4] or NULL */
  double u;
  int32_t* ancestors_out;          /* int32[K] or NULL */
  void* status_ws;                 /* workspace whose status word is written, or NULL */
} gjx_run_resample;
typedef struct gjx_run_opts {
  int32_t flags, pad_;
  void* start_event;
  void* stop_event;
  const float* in_rows;
  int64_t in_stride;
  const int32_t* in_ancestors;
  const gjx_run_resample* resample;
} gjx_run_opts;
typedef struct gjx_run_info {
  int32_t n_partials, engine;
  int64_t tiles_offset;
} gjx_run_info;
int gjx_run_program_ex(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K, int64_t particle_offset, float* choices,
                       float* score, float* weight, float* logw, const float* logw_in, const float* sub, float* site_scores,
                       float* lse, int64_t K_total, void* workspace, size_t workspace_bytes, void* stream,
                       const gjx_run_opts* opts, gjx_run_info* info_out);

enum { GJX_OP_RUN = 1, GJX_OP_LSE = 2, GJX_OP_PICK = 3, GJX_OP_RESAMPLE = 4, GJX_OP_HMC = 5,
       GJX_OP_SSM = 6 };
/* Scratch for one call.  The first 256 bytes are a control block (completion tickets of the fused
 * log-sum-exp): the caller zero-fills a workspace ONCE after allocating it; every entry point leaves the
 * control block zeroed again, so a workspace can be reused by any sequence of calls on one stream. */
size_t gjx_workspace_bytes(int op, int64_t K);
/* status word of a workspace: 0, or GJX_STATUS_* bits left by the kernels that synchronise their blocks through
 * memory (one-launch resampling, one-launch importance / filter steps).  Synchronises the stream; clears the word. */
enum { GJX_STATUS_POLL_TIMEOUT = 1, GJX_STATUS_ZERO_TOTAL = 2, GJX_STATUS_VERIFY_MISMATCH = 4 /* gjx_peer_ctx, GJX_PEER_VERIFY=1 */ };
int gjx_workspace_status(void* workspace, int32_t* status_host, void* stream);
/* profiling hook of profiles/microbench/: registers a device buffer into which the co-resident kernels write per-block
 * phase stamps (s_memrealtime) while it is registered and large enough for their grid (8 or 16 u64 per block);
 * (NULL, 0) unregisters.  Not for production runs. */
int gjx_debug_timeline(void* device_buffer, size_t bytes);

/* One SMC importance step in ONE launch on one GPU (smc.py:298-315 + 96-97 + the cookbook's resample-and-gather):
 * propagate + reweight every particle, global log-sum-exp, fixed-point prefix sums of the weights, systematic
 * ancestors (comb offset u) and the gather of the resampled particles — the blocks of one co-resident grid meet at
 * three tagged-granule all-gathers instead of three kernel boundaries; log-weights never leave registers between
 * propagate and the prefix sum, ancestors live only as scratch.  Outputs: choices f32[n_slots][K], score, logw f32[K]
 * (the collection BEFORE resampling), lse f32[4], rows_out f32[n_slots][K] (the resampled collection), ancestors i32[K].
 * Same bits as gjx_run_program -> gjx_resample_indices -> gjx_gather_rows.  Returns GJX_EUNSUPPORTED when the program
 * has no fused engine, K is not a multiple of 1024, or K/1024 blocks would not be co-resident on the device: callers
 * then use the three calls.  workspace: gjx_workspace_bytes(GJX_OP_RUN, K), zero-filled once; check
 * gjx_workspace_status after a batch of steps. */
int gjx_importance_step(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K, int64_t particle_offset,
                        float* choices, float* score, float* logw, float* lse, double u, float* rows_out,
                        int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream);
/* the same with a pair of HIP events (or NULLs) attached to the dispatch of the step's kernel: its own begin and end */
int gjx_importance_step_ex(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t K, int64_t particle_offset,
                           float* choices, float* score, float* logw, float* lse, double u, float* rows_out,
                           int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream,
                           void* start_event, void* stop_event);

/* ---- log-sum-exp (smc.py:97,107,464) -----------------------------------------------------
 * out[4] = {max, sum exp(x-max), logsumexp, logsumexp - log(K_total)} */
int gjx_logsumexp(const float* x, int64_t K, int64_t K_total, float* out, void* workspace,
                  size_t workspace_bytes, void* stream);
/* combine G partial {max, sumexp} pairs (one per rank, gathered by the caller over RCCL) */
int gjx_lse_combine(const float* pairs /*[G][2]*/, int G, int64_t K_total, float* out,
                    void* stream);

/* ---- independent trials in one launch: jax.vmap(alg.run_smc / alg.random_weighted)(jax.random.split(key, n)) of the
 * reference's README (README.md:108-113).  Trial t owns the global particle indices [particle_offset + t K,
 * particle_offset + (t + 1) K) of ONE n K-particle run of gjx_run_program (so trial t is exactly the K_local = K shard at
 * that offset of the sharded run, normalised on its own): lse_out f32[n_trials][4] = {max, sumexp, lse, lse - log K} per
 * trial; pick_out (or NULL) int32[n_trials] = the trial's 1-of-K draw as a GLOBAL index, by the rule of
 * gjx_categorical_pick (smc.py:102-109) with that trial's offset (a trial whose weights are all zero: lse = -inf, and the
 * index of its first particle). */
int gjx_trials_lse_pick(const float* logw, int64_t n_trials, int64_t K, int64_t particle_offset, uint32_t key0,
                        uint32_t key1, int32_t rng_mode, float* lse_out, int32_t* pick_out, void* stream);

/* ---- measurement aid (bench.py's roofline figure): HIP events that gjx_run_program_ex (GJX_RUN_TIME_DISPATCH) and
 * gjx_importance_step_ex attach to the dispatch of their kernel, so that the kernel's own begin and end are timed inside a
 * running loop (an event pair recorded around the call also times the dispatch hand-offs on both sides).  Events are
 * created / read / destroyed through the library so that the caller needs no HIP; the library keeps no reference to them. */
int gjx_event_create(void** event_out);
int gjx_event_destroy(void* event);
int gjx_event_elapsed_us(void* start_event, void* stop_event, float* us_out); /* waits for stop_event */

/* ---- 1-of-K categorical draw over the weights: ParticleCollection.sample_particle
 * (smc.py:102-109): idx = argmax_i (logw[i] - lse) + Gumbel(bits(key, i)).
 * out_dev: {float best_value, int32 idx (global index = particle_offset + i)} as 2 words */
int gjx_categorical_pick(const float* logw, int64_t K, int64_t particle_offset, const float* lse,
                         uint32_t key0, uint32_t key1, int32_t rng_mode, void* out_dev,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- N-of-K resampling (not in the reference library; the cookbook idiom is K independent
 * categorical draws + gather, docs/cookbook/inactive/inference/importance_sampling.ipynb) ----
 * Weights are turned into exact fixed-point integers q_i = (uint64)(w_i * 2^30),
 * w_i = is_log ? exp(x_i - max) : x_i, so prefix sums and comb searches are exact integer work.
 *   cum            u64[K] out : inclusive prefix sums of q
 *   base_total_dev u64[2] out : {0, sum q} — ready to pass to the resamplers on one GPU; with several ranks the
 *                               caller overwrites it with {sum of lower ranks' totals, sum over all ranks}
 */
int gjx_weight_cumsum(const float* x, int64_t K, int32_t is_log, const float* lse, int32_t n_partials,
                      uint64_t* cum, uint64_t* base_total_dev, float* lse_out, int64_t K_total,
                      void* workspace, size_t workspace_bytes, void* stream);
/*   is_log = 0 : x are linear weights.
 *   is_log = 1 : x are log-weights, lse[0] holds their maximum (a finished LSE record).
 *   is_log = 2 : x are log-weights and `lse` points at the n_partials per-block {max, sumexp} pairs that
 *                gjx_run_program leaves at workspace + 256 when called with lse == NULL; the reduction of the
 *                pairs rides in this call's prologue (no serial tail in the producing kernel) and, if lse_out is
 *                not NULL, the finished record {max, sumexp, lse, lse - log K_total} is written there. */
int gjx_run_partials_count(const gjx_program* prog, int64_t K, int64_t particle_offset);
/* (the block pairs a run left, and the byte offset of its tile totals — uint64 S[nt] followed by int32 E[nt], nt = K / 1024, left
 * when GJX_RUN_LEAVE_TILES is set, lse == NULL, K % 1024 == 0 and a block of the kernel covers whole 1024-particle tiles — are
 * reported in gjx_run_info by gjx_run_program_ex; gjx_run_partials_count re-derives a plan and is for sizing only) */
/* systematic comb over the GLOBAL weight line [0, total_all): local particles cover
 * [base, base + cum[K-1]).  Output slot j (global, 0..N_total-1) sits at (j + u) * total_all / N_total.
 * Writes ancestors for output slots [out_begin, out_begin + n_out) that fall on local particles:
 * ancestors[j - out_begin] = local index i; slots that belong to another rank's particles are LEFT UNTOUCHED
 * (a caller that needs markers pre-fills the buffer with -1; on one GPU every slot is written).
 * base_total_dev: u64[2] = {base, total_all} on the device. */
int gjx_resample_systematic(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev,
                            double u, int64_t N_total, int64_t out_begin, int64_t n_out,
                            int32_t* ancestors, void* stream);
/* Single-GPU resampling indices in ONE launch: fixed-point weights (is_log / lse / n_partials as in
 * gjx_weight_cumsum), their prefix sums and the systematic ancestors ancestors[j], j < N (every slot is written).
 * cum u64[K] and base_total_dev u64[2] are optional outputs (NULL = not materialised; they are required only when K
 * is too large for the co-resident fused kernel, K > 2^22, and the call falls back to the three-launch path).
 * Results are bit-identical to gjx_weight_cumsum + gjx_resample_systematic.
 * The fused kernel's blocks exchange their totals inside the launch, so its whole grid (<= 1024 blocks) must be
 * resident: do not run two of these launches concurrently on one device (two streams).  A block that cannot collect
 * the totals within its poll budget (~1 s) sets bit 0 of workspace word 10 and the output of that call is undefined;
 * a zero grand total (every weight -inf, NaN or 0) sets bit 1 and yields the identity ancestors (in bounds for any
 * later gather).  Read and clear the word with gjx_workspace_status after a batch of calls. */
int gjx_resample_indices(const float* x, int64_t K, int32_t is_log, const float* lse, int32_t n_partials, double u,
                         int64_t N, int32_t* ancestors, uint64_t* cum, uint64_t* base_total_dev, float* lse_out,
                         int64_t K_total, void* workspace, size_t workspace_bytes, void* stream);
/* Resampling AND the row gather in ONE launch on one GPU (N = K): dst[r][j] = src[r][ancestor(j)], r < rows, j < K,
 * with the ancestors of gjx_resample_indices (bit-identical), which are written to `ancestors` (int32[K]) only when it
 * is not NULL.  Weights as in gjx_resample_indices (is_log / lse / n_partials / lse_out / K_total).  Every block produces
 * the slots of its own index range and finds their ancestors by binary search (prefix of the tile totals, then the
 * re-scanned source tile), so the work is balanced for any weights.  Co-residency, status bits and the workspace are
 * those of gjx_resample_indices; returns GJX_EUNSUPPORTED when K / 1024 blocks would not be co-resident on the device
 * (K > 2^20 on a full MI355X): use gjx_resample_indices + gjx_gather_rows then.
 * Replaces the resample-and-index idiom of the reference's SMC cookbook (docs/cookbook/inactive/inference/
 * importance_sampling.ipynb: jax.random.categorical over the log-weights + jtu.tree_map(lambda v: v[idx], ...)). */
int gjx_resample_gather(const float* x, int64_t K, int32_t is_log, const float* lse, int32_t n_partials, double u,
                        const float* src, int64_t src_stride, int32_t rows, float* dst, int64_t dst_stride,
                        int32_t* ancestors, float* lse_out, int64_t K_total, void* workspace, size_t workspace_bytes,
                        void* stream);
/* Resampling and the row gather (N = K) under GJX_WEIGHTS_TILE_SCALED (below, at gjx_ssm_filter_scheme) as a PLAIN launch:
 * no block waits for another one, so there is no co-residency requirement, no poll budget and no limit from the device's
 * capacity (K <= 2^22: 4096 tiles).  tile_S / tile_E: the {S_b, e_b} of every 1024-particle tile of logw, as the producing
 * gjx_run_program_ex left them (gjx_run_info.tiles_offset) — both NULL: computed here by one extra small launch into the workspace.
 * lse_mode 2: `lse` points at n_partials block pairs of the producing run and lse_out receives the finished record
 * (reduced by block 0, off the critical path); lse_mode 0: no record.  Ancestors (written to `ancestors` when not NULL)
 * are those of gjx_resample_indices_tiled bit for bit (oracle: gjxo_resample_systematic_tiled).  A dead collection
 * yields the identity and bit 1 of the status word.  workspace: gjx_workspace_bytes(GJX_OP_RESAMPLE, K), zeroed once.
 * Same reference idiom as gjx_resample_gather (categorical draw over the weights + gather of every leaf). */
int gjx_resample_gather_tiled(const float* logw, int64_t K, const uint64_t* tile_S, const int32_t* tile_E, int32_t lse_mode,
                              const float* lse, int32_t n_partials, double u, const float* src, int64_t src_stride,
                              int32_t rows, float* dst, int64_t dst_stride, int32_t* ancestors, float* lse_out,
                              int64_t K_total, void* workspace, size_t workspace_bytes, void* stream);
/* the same search fused with the row gather: dst[r][j - out_begin] = src[r][ancestor(j)] for r < rows
 * (slots owned by another rank are left untouched in dst and in ancestors); ancestors int32[n_out] is scratch/output */
int gjx_resample_gather_systematic(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev, double u,
                                   int64_t N_total, int64_t out_begin, int64_t n_out, const float* src,
                                   int64_t src_stride, int32_t rows, float* dst, int64_t dst_stride,
                                   int32_t* ancestors, void* stream);
/* multinomial: slot j draws u_j = uniform(bits(key, j)), ancestor = first i with cum_i > u_j * total */
int gjx_resample_multinomial(const uint64_t* cum, int64_t K, const uint64_t* base_total_dev,
                             uint32_t key0, uint32_t key1, int64_t N_total, int64_t out_begin,
                             int64_t n_out, int32_t* ancestors, void* stream);
/* dst[r][j] = src[r][anc[j]]  for r < rows, j < n_out (anc[j] < 0 leaves dst untouched);
 * particle gather of smc.py:90-91 applied to every SoA row */
int gjx_gather_rows(const float* src, int64_t src_stride, const int32_t* anc, int64_t n_out,
                    int32_t rows, float* dst, int64_t dst_stride, void* stream);

/* ---- sharded collections: one rank's part of a global systematic resampling -----------------
 * The reference has no multi-device path; this is the build's own extension of
 * smc.py:90-101 to a collection split over ranks in contiguous particle ranges.  Every rank all-gathers
 * the per-rank fixed-point totals (gjx_weight_cumsum's base_total[1]) and builds the same plan. */
#define GJX_MAX_RANKS 64
typedef struct gjx_shard_plan {
  uint64_t base, total;       /* weight mass on lower ranks; mass on all ranks                       */
  int64_t slot0, n_valid;     /* this rank's particles produce the output slots [slot0, slot0+n_valid) */
  int64_t own_lo, own_n;      /* this rank stores the output slots [own_lo, own_lo+own_n)             */
  int64_t keep_lo, keep_hi;   /* produced AND stored here (no traffic)                                */
  int64_t n_ranks, status;    /* status 1: all weights are zero                                       */
  int64_t seq, reserved;      /* caller's sequence number, stored LAST (system-scope release)         */
  int64_t bounds[GJX_MAX_RANKS + 1]; /* rank r produces [bounds[r], bounds[r+1])                      */
} gjx_shard_plan;
/* one tiny launch: totals_dev u64[n_ranks] -> plan_dev, and (optional) the same bytes into pinned host
 * memory so the host can size the exchange without draining the stream: it polls plan_host_pinned->seq
 * until it reads the seq of this call (written after all other words), while the kernels queued behind
 * the plan keep running */
int gjx_shard_plan_build(const uint64_t* totals_dev, int32_t n_ranks, int32_t rank, double u, int64_t N_total,
                         int64_t seq, gjx_shard_plan* plan_dev, gjx_shard_plan* plan_host_pinned, void* stream);
/* ancestors[j - slot0] = local index of slot j's ancestor for this rank's run (capacity anc_capacity >= n_valid;
 * N_total always suffices), then the children that stay: dst[r][j - own_lo] = src[r][ancestor(j)],
 * keep_lo <= j < keep_hi.  Slot ranges are read from plan_dev; own_n (host-known) sizes the launch. */
int gjx_shard_resample(const uint64_t* cum, int64_t K, const gjx_shard_plan* plan_dev, double u, int64_t N_total,
                       int32_t* ancestors, int64_t anc_capacity, const float* src, int64_t src_stride, int32_t rows,
                       float* dst, int64_t dst_stride, int64_t own_n, void* stream);
/* surplus children as [n_pre + n_suf][rows] messages: message j < n_pre is the child of ancestors[j] (slots below
 * own_lo, bound for lower ranks), the others of ancestors[n_valid - n_suf + (j - n_pre)] (higher ranks) */
int gjx_shard_pack(const float* src, int64_t src_stride, int32_t rows, const int32_t* ancestors, int64_t n_valid,
                   int64_t n_pre, int64_t n_suf, float* msg, void* stream);
/* received [n_lo + n_hi][rows] messages into the head and the tail of this rank's slot range:
 * dst[r][j] = msg[j][r] (j < n_lo), dst[r][own_n - n_hi + t] = msg[n_lo + t][r] */
int gjx_shard_unpack(const float* msg, int64_t n_lo, int64_t n_hi, int32_t rows, float* dst, int64_t dst_stride,
                     int64_t own_n, void* stream);
/* The same exchange as ONE host call over RCCL (xGMI): all-gather of the per-rank {max, sumexp} (local_lse[0..1]),
 * fixed-point prefix sums, all-gather of the totals, plan, ancestors, children that stay gathered in place,
 * the surplus as grouped ncclSend/ncclRecv, unpack — everything on `stream`.  The context owns the
 * communicator and all scratch; RCCL is dlopen'ed from rccl_library_path (pass the library the process already
 * uses, e.g. torch/lib/librccl.so) and rank 0's gjx_rccl_unique_id bytes must be handed to every rank.
 *   rows_in  f32[rows][K_local] (stride in_stride), rows_out f32[rows][own_n] (must not alias rows_in), where
 *   own_n = this rank's share of N_total output slots (N_total/n_ranks, remainder to the low ranks);
 *   lse_out device f32[4] = global {max, sumexp, lse, lse - log N_total};
 *   info_host (optional) int64[4] = {children sent, received, first slot produced, slots produced}.
 * Blocks the host only until the plan kernel has run (the G+1 slot bounds size the messages). */
typedef struct gjx_shard_ctx gjx_shard_ctx;
int gjx_rccl_unique_id(const char* rccl_library_path, uint8_t* out128);
int gjx_shard_ctx_create(const char* rccl_library_path, const uint8_t* unique_id128, int32_t n_ranks, int32_t rank,
                         int64_t K_local, int32_t rows, int64_t N_total, gjx_shard_ctx** out);
int gjx_shard_ctx_destroy(gjx_shard_ctx* ctx);
int gjx_shard_resample_step(gjx_shard_ctx* ctx, const float* logw, const float* local_lse, const float* rows_in,
                            int64_t in_stride, float* rows_out, int64_t out_stride, double u, float* lse_out,
                            int64_t* info_host, void* stream);
/* host arithmetic only: one rank's message sizes from the plan.  send_counts / recv_counts int64[n_ranks];
 * parts4 (optional) = {children sent to lower ranks, to higher ranks, received from lower, from higher}.
 * send_counts[d] on rank r == recv_counts[r] on rank d for every pair, because all ranks hold the same bounds. */
int gjx_shard_message_counts(const gjx_shard_plan* plan, int32_t rank, int64_t N_total, int64_t* send_counts,
                             int64_t* recv_counts, int64_t* parts4);
/* dst[r*dst_row_stride + j*dst_col_stride] = src[r*src_row_stride + idx(j)*src_col_stride], idx = anc[j] or j
 * when anc is NULL: packs children into [n][rows] messages and unpacks received ones */
int gjx_gather_rows_strided(const float* src, int64_t src_row_stride, int64_t src_col_stride, const int32_t* anc,
                            int64_t n, int32_t rows, float* dst, int64_t dst_row_stride, int64_t dst_col_stride,
                            void* stream);

/* ---- linear-Gaussian state-space bootstrap-filter step (BASELINE config 3/4) --------------
 * x_t ~ N(A x_{t-1}[anc], q), weight = log N(y_t; H x_t, r); fused ancestor gather + propagate
 * + reweight + LSE partials.  Semantics are those of Scan.generate (combinators/scan.py:237-294)
 * applied one step at a time with resampling in between; the step key is chained by the caller
 * (key_t = fold_in(key_{t-1}, t), scan.py:268).
 *   A f32[dx][dx], H f32[dy][dx] row-major, y f32[dy], on the device.
 *   x_prev f32[dx][K_prev_stride], anc int32[K] or NULL (identity); t == 0 samples x_0 ~ N(0, I*q0)
 *   x_out f32[dx][K], logw f32[K] (incremental weight), lse f32[4] as above; lse == NULL with a workspace leaves
 *   ceil(K/256) per-block {max, sumexp} pairs at workspace + 256 (finished by gjx_weight_cumsum, is_log 2).
 */
typedef struct gjx_ssm {
  int32_t dx, dy;
  const float* A_dev;
  const float* H_dev; /* NULL = identity (dy == dx) */
  float q, r, q0;
} gjx_ssm;
int gjx_ssm_step(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t t,
                 int64_t K, int64_t particle_offset, const float* x_prev, int64_t prev_stride,
                 const int32_t* anc, const float* y_dev, float* x_out, float* logw, float* lse,
                 int64_t K_total, void* workspace, size_t workspace_bytes, void* stream);
/* The same step with a resample-move rejuvenation in front of the propagation (the reference's
 * ingredients are Regenerate / Rejuvenate, distribution.py:258-300, rejuvenate.py:70-94, and the caller-side accept of
 * tests/inference/test_requests.py:131-137): the resampled x_{t-1} takes n_moves random-walk Metropolis steps (scale
 * move_scale) that leave p(x_{t-1} | its parent, y_{t-1}) invariant — proposal, both densities and the accept fused into
 * the step kernel — and is then propagated.  m_prev f32[dx][K] = E[x_{t-1} | parent] written by the previous call as its
 * m_out (ignored at t <= 1: the prior mean is 0); accepted f32[K] = accepted moves per particle, or NULL;
 * x_moved_out f32[dx][K] (or NULL) = the moved x_{t-1} every slot was propagated from, in slot order (t >= 1) — the
 * parents a trajectory store must keep instead of the resampled, un-moved ones. */
int gjx_ssm_step_move(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t t, int64_t K,
                      int64_t particle_offset, const float* x_prev, const float* m_prev, int64_t prev_stride,
                      const int32_t* anc, const float* y_prev_dev, const float* y_dev, int32_t n_moves, float move_scale,
                      float* x_out, float* m_out, float* logw, float* accepted, float* x_moved_out, float* lse,
                      int64_t K_total, void* workspace, size_t workspace_bytes, void* stream);

/* The whole T-step bootstrap filter on ONE GPU: step 0, then steps 1 .. T-1 in ONE launch when the grid of K / 1024 blocks
 * is co-resident (otherwise one or two launches per step, looped in C++; no host round trip either way):
 * step keys k_t = fold_in(k_{t-1}, t), (k_prop, k_res) = split(k_t), systematic resampling before every
 * propagate step with comb offset uniform(k_res) — identical to issuing the per-step calls from the host.
 *   ys_dev f32[T][dy]; x_a, x_b f32[dx][K] (step t writes x_a for even t, x_b for odd t); logw f32[K];
 *   cum u64[K], ancestors i32[K] scratch; lse_steps f32[T][4] out (log-ML estimate = sum_t lse_steps[t][3]).
 *   workspace: 2 * gjx_workspace_bytes(GJX_OP_SSM, K) + 64 bytes, zero-filled once. */
int gjx_ssm_filter(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                   const float* ys_dev, float* x_a, float* x_b, float* logw, uint64_t* cum, int32_t* ancestors,
                   float* lse_steps, void* workspace, size_t workspace_bytes, void* stream);

/* Fixed-point weight schemes of the systematic resampler inside the filter loop.
 *   GJX_WEIGHTS_GLOBAL_MAX (what gjx_resample_indices does): q_i = floor(2^30 exp(log w_i - max_all log w)).  The exact
 *     global maximum is a grid-wide dependency of its own: two rendezvous per step in the one-launch filter.
 *   GJX_WEIGHTS_TILE_SCALED: every tile of 1024 consecutive particles is quantised against its own power of two,
 *     e_b = ceil(max_tile(log w) * log2 e) (float32 multiply, clamped to +-524287, i.e. |log w| < 3.6e5; a tile without a
 *     finite positive weight is dead), q_i = floor(2^29 min(1, exp2(fma(log w_i, log2 e, -e_b)))), S_b = sum of the tile's q_i.  With E = max e_b over
 *     the live tiles, tile b covers G_b = S_b >> (E - e_b) units of the global weight line (0 when the shift is >= 64),
 *     the comb thresholds T_j = floor((j + u) total / N) are taken on the prefix P of the G_b exactly as before, and
 *     inside its source tile b slot j takes the first particle whose tile-local cumulative q exceeds
 *     (T_j - P_b) << (E - e_b).  One exchange of {e_b, S_b} per step instead of two; a tile loses less than one unit of
 *     2^(E-29) (under 2^-28 of the largest weight), nothing else is approximated; the result does not depend on the
 *     launch geometry (the tile is part of the scheme).  Same systematic comb, same key discipline, same propagation
 *     streams; the ancestors differ from GLOBAL_MAX's only where a threshold falls within the quantisation step of a
 *     particle boundary.  Oracle: gjxo_resample_systematic_tiled.
 * gjx_ssm_filter == gjx_ssm_filter_scheme(..., GJX_WEIGHTS_GLOBAL_MAX, ...). */
enum { GJX_WEIGHTS_GLOBAL_MAX = 0, GJX_WEIGHTS_TILE_SCALED = 1,
       /* OR-ed into weight_scheme: run the filter as plain launches only — no kernel whose blocks wait for each other.  Same
        * keys, bit-identical results; what a caller passes when it repeats a run whose status word shows GJX_STATUS_POLL_TIMEOUT */
       GJX_WEIGHTS_PLAIN_LAUNCHES = 256 };
int gjx_ssm_filter_scheme(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                          const float* ys_dev, float* x_a, float* x_b, float* logw, uint64_t* cum, int32_t* ancestors,
                          float* lse_steps, int32_t weight_scheme, void* workspace, size_t workspace_bytes, void* stream);
/* The same filter WITH resample-move rejuvenation inside the one launch (requests/rejuvenate.py:70-94 fused into the
 * filter, as gjx_ssm_step_move does per step): after every resampling each particle takes n_moves random-walk Metropolis
 * steps of scale move_scale that leave p(x_{t-1} | parent, y_{t-1}) invariant, then propagates.  GJX_WEIGHTS_TILE_SCALED
 * resampler; the same draws and arithmetic as the per-step calls (gjx_resample_indices_tiled + gjx_ssm_step_move), so
 * particles and weights are bit-identical to that loop.  m_a, m_b f32[dx][K]: the transition means (ping-pong like x_a,
 * x_b); logw_alt f32[K]: the second log-weight buffer; accepted_total u64[1] (or NULL): accepted moves over the run.
 * GJX_EUNSUPPORTED when the shape (dx not in 2, 4, 8, 16; dy > 32) or the size (grid not co-resident) is outside the
 * one-launch kernel: loop over gjx_ssm_step_move then.  workspace as gjx_ssm_filter. */
int gjx_ssm_filter_move(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, int64_t K,
                        const float* ys_dev, float* x_a, float* x_b, float* m_a, float* m_b, float* logw, float* logw_alt,
                        int32_t* ancestors, float* lse_steps, int32_t n_moves, float move_scale, uint64_t* accepted_total,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The tile-scaled systematic resampler on its own (three plain launches; what the filter's one-launch form computes
 * between two steps, bit for bit): log-weights f32[K] -> ancestors i32[N].  cum u64[K] scratch (tile-local cumulative q);
 * q_out u32[K] / e_out i32[ceil(K/1024)] receive the quantised weights and tile exponents when not NULL (the oracle
 * checks the integer logic from them, the exp2 separately).  A dead collection yields identity ancestors and
 * GJX_STATUS_ZERO_TOTAL.  workspace: gjx_workspace_bytes(GJX_OP_RESAMPLE, K), zero-filled once.  There is no counterpart
 * in the reference (its SMC does not resample); the comb is that of gjx_resample_systematic. */
int gjx_resample_indices_tiled(const float* logw, int64_t K, double u, int64_t N, int32_t* ancestors, uint64_t* cum,
                               uint32_t* q_out, int32_t* e_out, void* workspace, size_t workspace_bytes, void* stream);
/* MULTINOMIAL resampling on the same tile-scaled weight line, by SORTED uniforms (what the generic filter's GJX_FILTER_MULTINOMIAL
 * computes between two steps, bit for bit; five plain launches here).  The reference's SMC cookbook resamples with
 * jax.random.categorical(key, log_weights, shape=(N,)) (docs/cookbook/inactive/inference/importance_sampling.ipynb): N iid draws
 * from the normalised weights.  The N draws as a SET are the images of N iid uniforms under the inverse CDF; this call produces the
 * uniforms already SORTED — U_(j) = S_j / S_{N+1}, S_j = e_0 + ... + e_j the running sum of N + 1 iid standard exponentials
 * (Devroye 1986, V.3.1) — so that slot j's ancestor needs no sort and no second pass, the ancestors come out non-decreasing (the
 * gather that follows reads forward), and the resampled collection is the same multinomial draw as the cookbook's up to the order of
 * its members, which no estimator over the collection depends on.
 *   e_j = floor(2^20 * -log2(x_j / 2^24)), x_j = 2 (w_j >> 9) + 1 for w_j the first word of Threefry(key, (j >> 32, j)) (the common
 *   scale 2^20 / ln 2 cancels in S_j / S_{N+1}) — log2 by a fixed float32 polynomial (integer -> float, a power-of-two scale, a Horner chain of fmaf), the same integers on
 *   the device and in the oracle (gjxo_exp_spacing); S_j are exact integer sums (independent of the launch geometry);
 *   T_j = min(floor((double)S_j * ((double)total / (double)S_{N+1})), total - 1) on the tile-scaled weight line; ancestor(j) = the
 *   particle under T_j exactly as in gjx_resample_indices_tiled.
 * cum, q_out, e_out, the dead collection and the workspace as gjx_resample_indices_tiled; the workspace also holds
 * 16 ceil((N + 1) / 1024) + 8 bytes of slot-tile sums behind the 288 + 24 ceil(K / 1024) bytes of the weight tiles (GJX_EWORKSPACE
 * when N is so much larger than K that gjx_workspace_bytes(GJX_OP_RESAMPLE, K) does not hold them).  Oracle:
 * gjxo_resample_sorted_multinomial_tiled. */
int gjx_resample_sorted_multinomial_tiled(const float* logw, int64_t K, uint32_t key0, uint32_t key1, int64_t N, int32_t* ancestors,
                                          uint64_t* cum, uint32_t* q_out, int32_t* e_out, void* workspace, size_t workspace_bytes,
                                          void* stream);

/* ---- bootstrap filter for ANY Scan kernel (SURVEY.md §8 R-2 beyond the linear-Gaussian model).  The reference's
 * ingredients: Scan.generate's step recursion (combinators/scan.py:237-294 — step t receives the carry of step t-1, weights
 * add over steps) and the cookbook's resample-and-index idiom; there is no filter in the reference.
 *   steps[T]: one program per step (host array).  steps[0] is step 0 (it reads no carry); the sites of steps[t], t >= 1, that
 *   stand for the choices of step t-1 have mode GJX_MODE_INPUT, come FIRST, and their obs_off numbers the rows the OWN
 *   (non-INPUT) sites of steps[t-1] wrote, in order.  Periodic Scans give T - 1 programs of one structure (one generated
 *   kernel) that differ in their tables (the step's observation).  Keys: k_t = fold_in(k_{t-1}, t), (k_prop, k_res) =
 *   split(k_t); step t runs under k_prop with its sites numbered from 1 (INPUT sites take no number); systematic
 *   resampling (GJX_WEIGHTS_TILE_SCALED) with comb offset uniform(k_res) in front of every step t >= 1.
 * Forms, fastest first (the library takes the first one the run allows; opts->flags can rule forms out; info_out says which ran):
 *   GJX_FILTER_FORM_WIDE   steps 1 .. T-1 in ONE launch of the filter kernel GENERATED for the step program (gjx_gen_pf): the
 *       step's sites as the model of the skeleton the hand-written linear-Gaussian filter runs on (csrc/gjx_pfcore.h) — a
 *       1024-particle tile is a block of 16 waves, one particle per lane; one tagged granule {e_b, S_b} per tile is the step's only
 *       rendezvous; the step's table is staged and its standard-normal draws are taken while the granules travel; the carry is
 *       read through the ancestors.  Needs: the step programs 1 .. T-1 are one kernel (same sites: a periodic Scan — tables, keys
 *       and comb offsets are per-step data), their sites are SAMPLE / OBS_TAB / INPUT, T >= 2, K <= 2^22 (any K: the last tile
 *       may be partial), a grid of ceil(K / 1024 / tiles-per-block) co-resident blocks, and the workspace room below.
 *   GJX_FILTER_FORM_STEPS  steps 2 .. T-1 in one launch of the 256-thread steps kernel (gjx_gen_steps; 4 particles per lane,
 *       K % 1024 == 0, K <= 2^20 and a quarter of the tiles co-resident); kept for comparison (GJX_FILTER_NO_WIDE).
 *   GJX_FILTER_FORM_PER_STEP  one plain launch per step: the step's kernel searches the ancestors of its own tile in its prologue
 *       (gjx_run_resample) and reads its carry THROUGH them.  K % 1024 == 0, K <= 2^20.  No co-resident grid, nothing to time out.
 *   GJX_FILTER_FORM_TWO_LAUNCH  two plain launches per step: the resampler's search (log-weights -> ancestors), then the step's
 *       kernel.  Any K <= 2^26, any program an engine runs.
 * All forms give the same ancestors, states and weights bit for bit (LSE records to float summation order).  A one-launch form whose
 * grid turns out not to be co-resident (another kernel holds compute units) sets GJX_STATUS_POLL_TIMEOUT in the status word of the
 * OP_RESAMPLE part of the workspace (workspace + gjx_workspace_bytes(GJX_OP_RUN, K)) and ends the launch: the caller repeats the
 * run with GJX_FILTER_NO_ONE_LAUNCH.  The status word is CLEARED at the start of every call (it describes this call only).
 *   rows_a / rows_b f32[max_t n_slots][K]: choices of even / odd steps (the last step's end up in rows_[(T-1)&1]);
 *   logw f32[K] the last step's incremental log-weights; ancestors int32[K] scratch / the last resampling's ancestors;
 *   ancestors_all (or NULL) int32[T-1][K]: the ancestors of every resampling (trajectory reconstruction);
 *   lse_steps f32[T][4]: log-ML estimate = sum_t lse_steps[t][3];
 *   workspace: gjx_workspace_bytes(GJX_OP_RUN, K) + gjx_workspace_bytes(GJX_OP_RESAMPLE, K), zero-filled once; with
 *   2 * OP_RUN + OP_RESAMPLE + 4 K + 512 bytes the per-step form is possible (a second run workspace and log-weight buffer); with
 *   192 ceil(K / 1024) + 32 T + 2048 bytes more, the one-launch forms.
 *   opts (or NULL = defaults), info_out (or NULL): below.  The call keeps no state between calls and reads no environment
 *   variable: every choice is an argument. */
enum { GJX_FILTER_NO_WIDE = 1,          /* not the 16-wave filter kernel                                   */
       GJX_FILTER_NO_STEPS = 2,         /* not the 256-thread steps kernel                                 */
       GJX_FILTER_NO_ONE_LAUNCH = 3,    /* neither: plain launches only (the repeat after a poll time-out) */
       GJX_FILTER_TWO_LAUNCH = 4,       /* two launches per step (search, then step)                       */
       /* MULTINOMIAL resampling instead of systematic: the resampling in front of step t is gjx_resample_sorted_multinomial_tiled
        * under k_res_t (the key whose first word feeds the systematic comb offset) — N = K sorted uniforms from exponential spacings,
        * on the tile-scaled weight line.  Runs INSIDE the one-launch filter kernel (GJX_FILTER_FORM_WIDE: the spacing sums ride the
        * tile granules of the step's one rendezvous as a second word, no extra barrier) when the run has no rejuvenation move; the
        * other forms resample with the standalone call between two step launches — the same ancestors bit for bit.  The workspace
        * needs 8 K + 256 bytes beyond OP_RUN + OP_RESAMPLE (the plain-launch forms' cumulative weights). */
       GJX_FILTER_MULTINOMIAL = 8,
       /* A model that is MORE than the Scan: latent sites in front of it (static parameters, `phi ~ beta(...)` before the state-space
        * Scan; Scan.generate is a callee of any @gen body, scan.py:237-294) are drawn by step 0 and travel with the particle: every
        * later step receives them as GJX_MODE_INPUT sites flagged GJX_SITE_CARRIED — gathered through the ancestors like the carry,
        * STORED into the step's own INPUT rows, from where the next step gathers them again.  With this flag an INPUT site's obs_off
        * is the ABSOLUTE row of the previous step's buffer it reads (without it: the row among the previous step's OWN rows, which sit
        * behind its INPUT rows).  Forms: the wide filter kernel, one launch per step, two launches per step (not the 256-thread
        * steps kernel, no moves, one GPU). */
       GJX_FILTER_ABSOLUTE_INPUTS = 16 };
enum { GJX_FILTER_FORM_TWO_LAUNCH = 0, GJX_FILTER_FORM_PER_STEP = 1, GJX_FILTER_FORM_STEPS = 2, GJX_FILTER_FORM_WIDE = 3 };
typedef struct gjx_filter_opts {
  int32_t flags;                 /* GJX_FILTER_* */
  int32_t coresident_blocks;     /* 0: ask the device (occupancy query); > 0: assume so many blocks of a one-launch form are
                                    resident together (tests of the time-out path) */
  void* timeline;                /* profiling (profiles/microbench): device buffer for 16 u64 phase stamps per block of the step T / 2
                                    of the wide form, or NULL */
  int64_t timeline_bytes;
  /* resample-move (the reference's ingredients: Rejuvenate, requests/rejuvenate.py:70-94, and the caller-side accept of
   * tests/inference/test_requests.py:131-137; SURVEY.md §8 f-2 for ANY Scan kernel): behind every resampling from the second on, each
   * particle's gathered carry x_{t-1} takes n_moves random-walk Metropolis steps of scale move_scale (continuous rows; discrete rows
   * stay) that leave p(x_{t-1} | its ancestor's own inputs, the observations of step t-1) invariant — the density of the step
   * program itself, re-scored by code generated from it (under step t-1's table), proposal, both densities and the accept fused into
   * the filter kernel.  Stream: site 1022 of the step's propagation key; with R' = the number R of continuous carry rows rounded up
   * to even, move n draws elements n (R' + 2) + c for the c-th continuous row and n (R' + 2) + R' for the accept's uniform (FLAT
   * normals come in Box-Muller pairs of elements: the accept must not share an element with a normal's partner).  The (moved) inputs of every step are stored in its INPUT rows
   * (rows [0, n_in) of the step's choices).  GJX_FILTER_FORM_WIDE only (GJX_EUNSUPPORTED otherwise): the step's latent choices must
   * be exactly the carry of the next step, no plates.  accepted_total u64[1] on the device (or NULL): accepted moves of the run. */
  int32_t n_moves;
  float move_scale;
  void* accepted_total;
  /* (ABI 10) an HMC move behind every resampling — the reference's HMC edit request used as a rejuvenation move (hmc.py:138-211) with
   * the caller-side accept of tests/inference/test_requests.py:134-137 fused (gjx_hmc(..., accept = 1)) — in the plain-launch form of
   * the filter.  hmc_targets[t] (t = 0 .. T-2), or NULL: step t's program in ASSESS form — its GJX_MODE_INPUT sites as they are, its
   * latent choices GJX_MODE_OBS_SLOT with the moved ones flagged GJX_SITE_SELECTED, its observations in the table, the SAME rows as
   * step t.  Behind the resampling in front of step t + 1 the particle [its ancestor's own inputs | its latent choices] is gathered
   * into hmc_rows, moved by ONE gjx_hmc launch over all K particles (L leapfrog steps of size hmc_eps; key
   * fold_in(fold_in(k_prop of step t + 1, 0x6d6f7665), 0)), and step t + 1 propagates from the moved rows, which it stores as its
   * inputs.  The target is p(x_t | inputs) p(y_t | x_t), the law of the resampled particle: the filter stays proper.
   * hmc_rows f32[max n_slots][K], hmc_out f32[3][K] (score, alpha, accepted flags of the last move), hmc_workspace /
   * hmc_workspace_bytes (>= gjx_hmc_workspace_bytes of every target): device scratch of the caller; accepted_total counts the accepted
   * chains of the run.  Forces GJX_FILTER_FORM_TWO_LAUNCH; GJX_EUNSUPPORTED with n_moves > 0 or GJX_FILTER_ABSOLUTE_INPUTS. */
  const gjx_program* hmc_targets;
  float hmc_eps;
  int32_t hmc_L;
  float* hmc_rows;
  float* hmc_out;
  void* hmc_workspace;
  size_t hmc_workspace_bytes;
} gjx_filter_opts;
typedef struct gjx_filter_info {
  int32_t form;                  /* GJX_FILTER_FORM_* of the steps from the third on (the form that dominates the run) */
  int32_t launches;              /* kernel launches this call issued (without the few tiny argument uploads of a one-launch form) */
  int32_t grid;                  /* blocks of the one-launch kernel (0: none ran) */
  int32_t tiles_per_block;       /* of the one-launch kernel */
} gjx_filter_info;
int gjx_scan_filter(const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, int64_t K, float* rows_a, float* rows_b,
                    float* logw, int32_t* ancestors, int32_t* ancestors_all, float* lse_steps, void* workspace, size_t workspace_bytes,
                    void* stream, const gjx_filter_opts* opts, gjx_filter_info* info_out);
/* The same run with the choices of EVERY step kept: rows_all f32[T][rows_per_step][K] (step t's program writes its rows into
 * block t, rows_per_step >= every step's n_slots) and ancestors_all int32[T-1][K] (required for T > 1).  The trajectory that
 * ends in particle i of the last step is read back by following the ancestors: i_{t-1} = ancestors_all[t-1][i_t] — one row
 * gather per step instead of the reference's stacked per-particle trace (ScanTrace, scan.py:56-97, re-gathered whole at every
 * resampling).  Everything else as gjx_scan_filter. */
int gjx_scan_filter_history(const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, int64_t K, float* rows_all,
                            int32_t rows_per_step, float* logw, int32_t* ancestors_all, float* lse_steps, void* workspace,
                            size_t workspace_bytes, void* stream, const gjx_filter_opts* opts, gjx_filter_info* info_out);
/* the filter kernel generated for a step program (GJX_FILTER_FORM_WIDE) with `tiles_per_block` in {1, 2, 4, 8, 16} (| 256: the flavour
 * that runs on a collection sharded over peer-mapped windows — system-scope accesses and the verify mode decided at run time; | 512: with
 * the rejuvenation move; | 1024, on its own: multinomial resampling by sorted uniforms): its HIP source
 * (returns the length; copies at most cap - 1 characters) and compilation without a launch (hipRTC cross-compiles for gfx950
 * without a GPU: a build step fills the on-disk cache).  GJX_EUNSUPPORTED when the emitter does not cover the program. */
int64_t gjx_program_filter_source(const gjx_program* step, int32_t tiles_per_block, char* out, int64_t cap);
int gjx_program_filter_precompile(const gjx_program* step, int32_t tiles_per_block);

/* The same filter on a collection sharded over the ranks of a shard context, BASELINE config 4: every rank
 * runs this loop with the same key and ys; per step one propagate+reweight launch on its K_local particles
 * (streams indexed by the global particle index particle_offset + i, so results do not depend on the number
 * of ranks) and one gjx_shard_resample_step.  ctx must have been created for (K_local, rows = dx,
 * N_total = n_ranks * K_local).  x_a (propagated) and x_b (resampled) f32[dx][K_local]; lse_steps f32[T][4]
 * receives the GLOBAL record of every step; workspace: gjx_workspace_bytes(GJX_OP_SSM, K_local) + 64 bytes,
 * zero-filled once. */
int gjx_ssm_filter_sharded(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T,
                           gjx_shard_ctx* ctx, int64_t particle_offset, const float* ys_dev, float* x_a, float* x_b,
                           float* logw, float* lse_steps, void* workspace, size_t workspace_bytes, void* stream);
/* shape of a context {K_local, rows, N_total, n_ranks, rank}; global LSE record from this rank's local one
 * (8-byte all-gather + combine) */
/* Multinomial resampling of the sharded collection (the all-to-all of north_star): output slot j draws its uniform
 * from the hash of its GLOBAL index (as gjx_resample_multinomial), every rank keeps the slots whose threshold lands
 * on its particles and ships each child, tagged with its slot, to the slot's owner.  Collectives per call: the two
 * 8-byte all-gathers of the systematic step, an all-gather of G counts per rank, grouped send/recv between every
 * pair of ranks with children to exchange.  Result == gjx_resample_multinomial + gjx_gather_rows on the unsharded
 * collection, bit for bit, for any number of ranks.  info_host[4] = {sent, received, kept, N_total}. */
int gjx_shard_resample_multinomial_step(gjx_shard_ctx* ctx, const float* logw, const float* local_lse, const float* rows_in,
                                        int64_t in_stride, float* rows_out, int64_t out_stride, uint32_t key0, uint32_t key1,
                                        float* lse_out, int64_t* info_host, void* stream);
/* counters since creation: out4 = {resampling steps, children sent, children received, ranks of the communicator} */
int gjx_shard_ctx_stats(const gjx_shard_ctx* ctx, int64_t* out4);
int gjx_shard_ctx_shape(const gjx_shard_ctx* ctx, int64_t out5[5]);
int gjx_shard_global_lse(gjx_shard_ctx* ctx, const float* local_lse, float* lse_out, void* stream);

/* ---- sharded collections without the host in the loop: peer-mapped exchange windows --------------------------
 * (build's own design for north_star's config 4; nothing in the reference.)  One process per GPU.
 * Every rank owns two windows — DATA (particle rows x2 and log-weights x2: what other ranks READ) and FLAG (tagged
 * granules, the LSE ring, `ready` words: what other ranks WRITE) — exports them as hipIpc handles and maps everybody
 * else's (xGMI peer access between GPUs).  The kernels below then exchange everything themselves: granules are pushed
 * with system-scope stores, source tiles and ancestors' rows are pulled with system-scope loads.  No collective call,
 * no host synchronisation and no allocation inside any loop.
 *   create:  K_local particles and `rows` SoA rows per rank; n_ranks > 1 needs K_local % 1024 == 0;
 *            ranks_on_this_device > 1 declares that so many ranks share ONE device (dry runs): grids are sized so
 *            that all of them stay co-resident.
 *   export:  out128 = the two 64-byte hipIpcMemHandle_t (data, flag); hand every rank's 128 bytes to
 *   connect: in rank order (n_ranks x 128 bytes; a rank's own entry is ignored).  One rank: connected from the start.
 *   buffers: out6 = device pointers of this rank's rows[0], rows[1] (f32[rows][K_local]), logw[0], logw[1] (f32[K_local])
 *            and the byte sizes of the two windows.  The caller's kernels write their particles THERE.
 *   status:  bit 0 a rendezvous timed out (a peer is missing: results undefined), bit 1 a collection had zero total
 *            weight, bit 2 (verify mode) a pulled row or source tile did not match its owner's check; read and cleared,
 *            synchronises the stream.
 *   Environment, read by create (the same on every rank of a run):
 *     GJX_PEER_VERIFY=1   every kernel that writes particle rows other ranks read leaves a 32-bit check word per row beside
 *                         it (a hash of the row's floats, of the STEP / call it belongs to and of the particle's global index),
 *                         every reader recomputes it from what it pulled; the fixed-point total of every re-scanned source
 *                         tile is compared with the total in the tile's granule.  A stale, torn or misdirected read raises
 *                         GJX_STATUS_VERIFY_MISMATCH instead of going unnoticed.  Results are unchanged bit for bit.
 *     GJX_PEER_DATA=fine  the DATA window is fine-grained device memory (hipDeviceMallocFinegrained) instead of ordinary
 *                         device memory: coherent between agents by memory type — the fallback if a fabric shows mismatches
 *                         with the default (coarse) window, whose visibility rests on sc0 sc1 accesses.
 *   destroy: only after every rank has finished using the context (the caller's barrier). */
typedef struct gjx_peer_ctx gjx_peer_ctx;
int gjx_peer_ctx_create(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device,
                        gjx_peer_ctx** out);
/* the same with the switches as an argument instead of the environment (a library user that must not touch the process environment;
 * the self-check of the host layer, which always runs with the check words on) */
enum { GJX_PEER_VERIFY_ON = 1, GJX_PEER_DATA_FINE = 2, GJX_PEER_VERIFY_FAULTY = 4 /* test hook: THIS rank publishes wrong check words */ };
int gjx_peer_ctx_create_ex(int32_t n_ranks, int32_t rank, int64_t K_local, int32_t rows, int32_t ranks_on_this_device, int32_t flags,
                           gjx_peer_ctx** out);
int gjx_peer_ctx_export(gjx_peer_ctx* ctx, uint8_t* out128);
int gjx_peer_ctx_connect(gjx_peer_ctx* ctx, const uint8_t* handles);
int gjx_peer_ctx_buffers(gjx_peer_ctx* ctx, uint64_t* out6);
int gjx_peer_ctx_status(gjx_peer_ctx* ctx, int32_t* status_host, void* stream);
int gjx_peer_ctx_destroy(gjx_peer_ctx* ctx);
/* BASELINE config 4 — the bootstrap filter of gjx_ssm_filter_scheme(GJX_WEIGHTS_TILE_SCALED) on a collection sharded
 * over the ranks of a peer context created with rows == dx; every rank calls this with the same key and ys (T >= 2).
 * TWO launches per rank whatever T is: step 0, then steps 1 .. T-1 in one launch in which the ranks meet once per step
 * through their flag windows (Scan.generate's step recursion, combinators/scan.py:237-294, with systematic resampling
 * in front of every step).  Streams are indexed by the global particle index and every integer of the resampling comes
 * from the same K_total / 1024 granules on every rank, so particles, weights and log-ML do not depend on n_ranks.
 * The particles of the last step are left in rows[(T - 1) & 1], their log-weights in logw[0]; lse_steps f32[T][4] =
 * the GLOBAL record of every step (on every rank); ancestors (or NULL) int32[K_local] = global index of every slot's
 * ancestor at the last resampling.  T <= GJX_PEER_MAX_STEPS per call (the context's per-step scratch; nothing is allocated after
 * gjx_peer_ctx_create): a longer run is cut by the caller. */
#define GJX_PEER_MAX_STEPS 4096
int gjx_ssm_filter_peer(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* ctx,
                        const float* ys_dev, float* lse_steps, int32_t* ancestors, void* stream);
/* the sharded filter with resample-move rejuvenation: gjx_ssm_filter_move on a collection sharded over the ranks of a peer
 * context (the transition means travel through the DATA windows like the states).  accepted_total u64[1] (or NULL): accepted
 * moves of THIS rank's particles over the run.  Results do not depend on the number of ranks. */
int gjx_ssm_filter_peer_move(const gjx_ssm* m, uint32_t key0, uint32_t key1, int32_t rng_mode, int32_t T, gjx_peer_ctx* ctx,
                             const float* ys_dev, float* lse_steps, int32_t* ancestors, int32_t n_moves, float move_scale,
                             uint64_t* accepted_total, void* stream);
/* The bootstrap filter for ANY Scan kernel (gjx_scan_filter, form GJX_FILTER_FORM_WIDE) on a collection sharded over the ranks of a
 * peer context (north_star: particles shard across the GPUs of a node; the recursion is Scan.generate's, combinators/scan.py:237-294).
 * Every rank calls this with ITS copies of the step programs (same structure and tables on every rank) and the same key; the context
 * must have rows >= every step's n_slots.  TWO launches per rank whatever T is: step 0 by its program's kernel, then steps 1 .. T-1
 * in one launch of the filter kernel generated for the step program — the model of the skeleton gjx_ssm_filter_peer's kernel runs on,
 * hence the same exchange: granules pushed into every rank's flag window, source tiles' log-weights and the ancestors' carry rows
 * pulled through the peer mappings, one rendezvous per step.  Streams are indexed by the global particle index and every integer of
 * the resampling comes from the same K_total / 1024 granules on every rank: results do not depend on n_ranks (bit-identical to
 * gjx_scan_filter on the unsharded collection).  GJX_PEER_VERIFY / GJX_PEER_DATA of the context apply (check words cover the carry
 * rows a step hands to the next).  The choices of the last step are left in rows[(T - 1) & 1] of the context (row r of the step
 * program = row r of the window), their log-weights in logw[0]; lse_steps f32[T][4] = the GLOBAL record of every step (every
 * rank); ancestors (or NULL) int32[K_local] = global index of every slot's ancestor at the last resampling.
 * workspace: gjx_workspace_bytes(GJX_OP_RUN, K_local) + 8 T + 256 bytes, zero-filled once.  GJX_EUNSUPPORTED: the step programs are
 * not one kernel the filter emitter covers, or no co-resident grid exists for the size. */
/* what takes unpredictable HOST time in the call below — generating, compiling and loading the kernels of the step programs, the
 * occupancy queries behind the choice of tiles per block — and no launch: the ranks of a job call it, meet at a host barrier, and only
 * then enter the filter together (a rank that waits for a peer that is still compiling would run out of its poll budget).  info_out
 * (or NULL): the grid and tiles per block the run will use. */
int gjx_scan_filter_peer_prepare(gjx_peer_ctx* ctx, const gjx_program* steps, int32_t T, gjx_filter_info* info_out);
int gjx_scan_filter_peer(gjx_peer_ctx* ctx, const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, float* lse_steps,
                         int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream, gjx_filter_info* info_out);
/* the two calls above with the options of gjx_scan_filter that the sharded kernel carries (opts NULL: the calls above):
 * GJX_FILTER_MULTINOMIAL — multinomial resampling by sorted uniforms over the WHOLE sharded collection ("multinomial
 * resampling uses the same exchange with sorted uniforms"): slot j's spacing comes from its GLOBAL index, every tile's spacing total
 * travels to every rank as the second word of the tile's granule, so the ranks exchange nothing beyond the systematic filter's granules
 * and pulled carry rows; ancestors == gjx_resample_sorted_multinomial_tiled on the unsharded collection, bit for bit.  Any other flag,
 * or n_moves > 0: GJX_EUNSUPPORTED.  prepare_opts must be given the opts of the run (the flavour is another kernel). */
int gjx_scan_filter_peer_prepare_opts(gjx_peer_ctx* ctx, const gjx_program* steps, int32_t T, const gjx_filter_opts* opts, gjx_filter_info* info_out);
int gjx_scan_filter_peer_opts(gjx_peer_ctx* ctx, const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, float* lse_steps,
                              int32_t* ancestors, void* workspace, size_t workspace_bytes, void* stream, const gjx_filter_opts* opts,
                              gjx_filter_info* info_out);
/* BASELINE configs 2 / 4 on a sharded collection — one systematic resampling step over the WHOLE collection in ONE
 * launch per rank (ParticleCollection resampling, the N-of-K form of smc.py:102-109): this rank's log-weights logw[parity]
 * and rows rows[parity] (the buffers of the DATA window the producing kernel wrote) -> the children of this rank's
 * K_local output slots in rows_out f32[rows][out_stride] (caller's memory), pulled from whichever rank holds the
 * ancestor.  Tile-scaled fixed point; the ranks meet twice through their flag windows (largest tile exponent, then the
 * rank totals at that exponent: G words each).  Ancestors == gjx_resample_indices_tiled on the unsharded collection, bit
 * for bit, for any number of ranks.  Alternate `parity` from call to call (a rank may be one call ahead of another).
 *   partials / n_partials: the per-block {max, sumexp} pairs gjx_run_program(lse == NULL) left at its workspace + 256:
 *   lse_out f32[4] then receives the GLOBAL record on every rank; partials == NULL: no record.
 *   ancestors (or NULL) int32[K_local]: global ancestor index of every slot.
 * GJX_EUNSUPPORTED when K_local / 1024 blocks are not co-resident (K_local <= 2^20 on a full MI355X). */
int gjx_peer_resample_gather(gjx_peer_ctx* ctx, int32_t parity, const float* partials, int32_t n_partials, double u,
                             float* rows_out, int64_t out_stride, int32_t* ancestors, float* lse_out, void* stream);

/* ---- HMC move: HMC.edit (inference/requests/hmc.py:156-211) -------------------------------
 * One chain per particle column.  Moves the slots of sites flagged GJX_SITE_HMC_SELECTED (float
 * sites only, hmc.py:49-65); every other site keeps its value (mode OBS_*).
 *   JAX32 stream: chain key = Threefry(key, (0, chain_offset + i)); (key', sub) = split(chain key);
 *   momentum leaf l ~ N(0,1) from fold_in(sub, l) (hmc.py:120-130).  FLAT stream: leaf l is site l+1.
 *   choices f32[n_slots][n] in/out, score f32[n] in/out, alpha f32[n] out (hmc.py:196-203).
 *   stale_grad_compat != 0 reproduces hmc.py:186 (first half-kick always uses the INITIAL gradient).
 *   workspace: gjx_hmc_workspace_bytes(prog, n) — the site interpreter's chain state; a generated kernel uses it for the trajectory
 *   rows (position, momentum, gradient) of selected sites inside plates and of a rolled Scan's steps, nothing else (its other state
 *   lives in registers).  Plate-tagged programs and long periodic Scans run on generated kernels: a selected body site of a plate is
 *   ONE momentum leaf with elements i * dim + d; a Scan's sites are one leaf each, in program order.
 *   accept != 0 additionally applies the caller-side MH rule of tests/inference/test_requests.py:134-137
 *   with log U drawn from fold_in(key', 0x4d48) (FLAT: site 1023) and reverts rejected chains; accepted f32[n] out or NULL.
 */
size_t gjx_hmc_workspace_bytes(const gjx_program* prog, int64_t n);
/* which engine gjx_hmc will use: 0 = generic site interpreter, 2 / 3 = fused hierarchical-logistic-regression kernels (vector /
 * matrix-core), 4 = a kernel generated from the site list; GJX_HMC_ENGINE = auto | fused | gen | interp restricts the choice.
 * A generated kernel runs both contractions of a rolled site with an affine parameter (n = 16 .. 64 inputs, rows a multiple of
 * 16: the likelihood of a regression) on the matrix cores, and folds 0 / 1 observations of a bernoulli-logits site into its
 * copies of the matrix; profiling variants of the emitter: GJX_HMC_GEN_NO_MFMA=1 (scalar rolled loop), GJX_HMC_GEN_NO_FOLD=1,
 * GJX_HMC_GEN_BT=256|512|1024 (threads per block).  GJX_MODE_INPUT sites are read as given values (never selected). */
int gjx_hmc_engine(const gjx_program* prog);
int gjx_hmc(const gjx_program* prog, uint32_t key0, uint32_t key1, int64_t n, int64_t chain_offset,
            float eps, int32_t L, int32_t stale_grad_compat, int32_t accept, float* choices,
            float* score, float* alpha, float* accepted, void* workspace, size_t workspace_bytes,
            void* stream);
/* The caller-side Metropolis-Hastings accept of the reference's move requests — `log(uniform(key)) < w` behind Rejuvenate.edit / HMC.edit
 * (tests/inference/test_requests.py:131-137) — as a device call: chain i draws log u_i from the bits x0 ^ x1 of Threefry(key, (i >> 32, i))
 * (23 bits, u in [tiny, 1)), and where log u_i < log_alpha[i] it takes the proposal: rows_cur[r][i] = rows_prop[r][i] for r < rows
 * (row_stride in floats, both arrays).  A NaN log_alpha never accepts.  accepted f32[K] receives 1 / 0 (or NULL); accepted_total, a u64
 * on the device, is incremented by the number of accepted chains (or NULL). */
int gjx_mh_accept(const float* log_alpha, int64_t K, uint32_t key0, uint32_t key1, float* rows_cur, const float* rows_prop,
                  int64_t row_stride, int32_t rows, float* accepted, void* accepted_total, void* stream);
/* d score / d choices for the selected slots: selection_gradient (hmc.py:70-96).
 * grad f32[n_slots][n] (rows of unselected slots are written as 0). */
int gjx_score_grad(const gjx_program* prog, int64_t n, const float* choices, float* score,
                   float* grad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GJX_H */
