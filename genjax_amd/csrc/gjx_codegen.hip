// gjx_codegen.hip — per-program fused propagate+reweight kernels.
//
// The reference stages ANY @gen body into a Jaxpr and lets XLA fuse it (generative_functions/static.py:383-399,
// core/compiler/staging.py:286-298).  The counterpart here: the site list of a gjx_program is turned into the HIP
// source of ONE straight-line kernel — every site's kind, mode, event size, parameter forms, table offsets and slot
// numbers are literals, every element loop is unrolled, particle values live in registers, tables in LDS, and whatever
// depends on the float table only (log-softmax / running CDF of constant logits, log and reciprocal of constant
// scales) is computed once per block in the prologue instead of once per particle — compiled with hipRTC for gfx950
// and cached per structure (in memory, and as a code object next to this library so that a build step can pre-populate
// the cache; the table VALUES are run-time data, so new observations do not recompile).  The site interpreter
// (k_run_generic) stays as the fallback for what the emitter does not cover (dirichlet sites, vector values wider than
// 32 that later sites read, non-table categorical logits).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hiprtc.h>
#include <unistd.h>
#include <stdarg.h>
#include <string.h>
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <unordered_map>
#include <mutex>
#include <string>
#include <vector>

#include "gjx_device.h"
#include "gjx_host.h"
#include "gjx_pfcore.h"

namespace {

const char* kDeviceHeader =
#include "build/gjx_device_h.inc"
    ;
const char* kApiHeader =
#include "build/gjx_h.inc"
    ;
const char* kScanHeader =
#include "build/gjx_scan_h.inc"
    ;
const char* kTileHeader =
#include "build/gjx_tile_h.inc"
    ;
const char* kPfCoreHeader =
#include "build/gjx_pfcore_h.inc"
    ;

// ---------------------------------------------------------------------------------------------------------
// emitter
// ---------------------------------------------------------------------------------------------------------
struct Emit {
  std::string s;
  void f(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    char buf[2048];
    va_list ap, ap2;
    va_start(ap, fmt);
    va_copy(ap2, ap);
    const int n = vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (n >= 0 && (size_t)n < sizeof(buf)) s += buf;
    else if (n > 0) {                       // (longer than the stack buffer: format again into the string itself)
      const size_t at = s.size();
      s.resize(at + (size_t)n + 1);
      vsnprintf(&s[at], (size_t)n + 1, fmt, ap2);
      s.resize(at + (size_t)n);
    }
    va_end(ap2);
  }
};

struct Companion {   // derived constants in LDS behind the table
  int kind;          // 0: log of tab[off .. off+n)   1: reciprocal   2: categorical {cdf[n], lse}
                     // 3: per-row normaliser of a diagonal normal, sum_e (log sqrt(2 pi) + log tab[off + r len + e % len]), r < n
                     // 4: observed value over scale, tab[dim + e] / tab[off + e % len], e < n (dim = offset of the observation)
  int off, n, at;    // table range; offset of the result in comp[]
  int len = 0, dim = 0;
};

struct SiteStream {
  std::string key_var; unsigned site_no;   // "" = the run key
  int run = -1;                            // scalar-normal run the site belongs to (gjx.h "Scalar-normal runs"), -1: none
  unsigned elem = 0;                       // its element in the run's stream
  bool opens = false;                      // the run's head: declares and opens the stream
};

// Scalar-normal runs over a segment of emitted sites [j0, j1) that share one stream key and one numbering
// (SiteStreamWalk::run_elem in gjx_device.h walks the same rule): fills run / elem / opens and points the members'
// site_no at their head's
void assign_runs(const gjx_program* prog, std::vector<SiteStream>& st, int j0, int j1, int& n_runs) {
  if (prog->rng_mode != GJX_RNG_FLAT) return;
  int head = -1;
  unsigned next = 0;
  int32_t tag = j0 < j1 ? prog->sites[j0].scan : 0;
  for (int j = j0; j < j1; ++j) {
    const gjx_site& s = prog->sites[j];
    if (s.scan != tag) { head = -1; tag = s.scan; }
    const bool draws = s.mode == GJX_MODE_SAMPLE || s.mode == GJX_MODE_OBS_MASK;
    if (s.plate == 0 && GJX_FLAT_JOINS(prog->rng_mode, s.kind, s.dim, s.mode)) {
      if (head < 0 || next >= (unsigned)GJX_FLAT_RUN_MAX) { head = j; next = 0; st[j].opens = true; st[j].run = n_runs++; }
      else { st[j].run = st[head].run; st[j].site_no = st[head].site_no; }
      st[j].elem = next++;
    } else if (draws) {
      head = -1;
    }
  }
}

// Where an emitted site's rows and table entries really are.  Outside a rolled Scan everything is static (the site's own
// fields); inside the loop over t_ (steps 1 .. T-1 of a rolled Scan) the site is emitted ONCE, from step 1's descriptor,
// and every step-dependent quantity is `value at step 1 + (t_ - 1) * stride`.
struct RollInfo {
  int row = -1, d_row = 0;              // first row of the site's value in choices[][]
  int score_row = 0, d_score_row = 0;   // row of the site in site_scores[][]
  int d_obs = 0;                        // stride of obs_off
  bool load_here = false;               // the site's own rows (OBS_SLOT / OBS_MASK values, the mask flags) are loaded in front of
                                        // the site instead of at the top of the kernel: their rows move with t_, or the
                                        // register that holds them is not the row number (rolled programs)
  int flag_row = -1, d_flag_row = 0;    // OBS_MASK: row of the site's flags in choices[][] (the register is gjx_site.obs_off)
  int d_off[4] = {0, 0, 0, 0}, d_moff[4] = {0, 0, 0, 0};   // strides of the parameters' table offsets
  // plates (gjx.h "Plates"): the site is emitted once inside `for (i_ ...)`; its loop variable is i_ instead of (t_ - 1)
  bool plate = false;
  int plate_n = 0, plate_l = 0, plate_j0 = 0;   // instances, position in the body, program index of the plate's first site
  // a parameter whose source rows are not in registers (instances of ANOTHER plate, one instance of a plate read from outside
  // it): read from choices[][] — rows mem_slot + i_ * d_mem + element; -1: a register source
  int mem_slot[4] = {-1, -1, -1, -1}, d_mem[4] = {0, 0, 0, 0};
  // GJX_P_EXPR parameter k of a PLATE program: where each leaf of its block really is (plate_program resolves them like the sources
  // of the closed forms): eleaf[k][eleaf_at[k][node] + e] for element e of the node's leaf span (VALUE: 1, LINV: its count)
  struct ExprLeaf { int reg = -1; int mem_row = -1, mem_stride = 0; };
  std::vector<ExprLeaf> eleaf[4];
  std::vector<int> eleaf_at[4];
  // ... of a ROLLED Scan's loop body: how far the table entries a node names (CONST; the [bias, weights] of LINV / LINN) advance per
  // step (detect_roll checked that every step's block is step 1's with these strides); empty: the nodes' own plate strides
  std::vector<int> enode_dtab[4];
};

// loop variable of the site being emitted: steps of a rolled Scan, or instances of a plate
thread_local const char* g_loop_var = "(t_ - 1)";

// "off" or "(off + <loop variable> * stride)"
std::string toff(int off, int stride) {
  if (stride == 0) return std::to_string(off);
  return "(" + std::to_string(off) + " + " + g_loop_var + " * " + std::to_string(stride) + ")";
}

// value of source element `elem` (an expression) of parameter k: a register, or a row of choices[][] (RollInfo::mem_slot)
std::string src_val(const gjx_param& q, const RollInfo& ri, int k, const std::string& elem) {
  if (ri.mem_slot[k] < 0) return "v[" + std::to_string(q.slot) + " + (" + elem + ")][p]";
  return "a.choices[(int64_t)(" + toff(ri.mem_slot[k], ri.d_mem[k]) + " + (" + elem + ")) * K + i0 + p]";
}

struct Plan {
  const gjx_program* prog;
  int ppt;
  // an observation of the site in front of it, element by element (y ~ normal(x, sigma) right behind x: every state-space and mixture
  // model's last site) is scored INSIDE that site's element loop — the value is used while it is in a register and its row is stored
  // at once, so that the rows of x are not held until the next site (fused_into[j] = the site whose loop scores site j, or -1)
  std::vector<int> fused_into;
  // straight-line programs (no plates, no rolled Scan, no filter flavour) store their rows through ONE running pointer, advanced by
  // K from row to row — a base address per row is a pair of scalar registers each, hoisted out of the tile loop, and a program of
  // 17 rows spills the scalar file.  rp_row: the row the pointer stands at (emitter's bookkeeping), -1 = plain addressing
  bool seq_rows = false;
  int rp_row = 0;
  std::string row_store(const char* macro, const std::string& ind, int row, const std::string& val) {
    char b[512];
    if (!seq_rows) { snprintf(b, sizeof(b), "%s%s(a.choices + (int64_t)%d * K + i0, %s);\n", ind.c_str(), macro, row, val.c_str()); return b; }
    std::string s;
    const int delta = row - rp_row;
    rp_row = row;
    if (delta) { snprintf(b, sizeof(b), "%srp_ += (int64_t)%d * K; asm volatile(\"\" : \"+v\"(rp_));\n", ind.c_str(), delta); s += b; }
    snprintf(b, sizeof(b), "%s%s(rp_, %s);\n", ind.c_str(), macro, val.c_str());
    return s + b;
  }
  bool mfma = false;                // flavour: big affine sites on the matrix cores (static LDS, PPT = 1, K % 256 == 0)
  int mfma_floats = 0;              // floats of the transpose patches (4 waves x 64 particles x inner length)
  bool tab_lds;
  std::vector<SiteStream> stream;   // per emitted site
  std::vector<RollInfo> info;       // per emitted site
  std::string key_decls;            // definitions of the chained step keys, in order
  std::vector<Companion> comps;
  int comp_floats = 0;
  // filter flavour (generate_pf): standard-normal draws of sampled normal sites are taken ahead of the site — while the step's
  // granules travel — into Draws::nz; hoist_at[j] = the site's first entry there, or -1
  std::vector<int> hoist_at;
  int n_hoist = 0;
  bool pf = false;
  std::vector<char> skip;           // sites emit_body leaves out (the assess form of a step program: inputs and proposal sites)
  // plate flavour "wide" (ppt code | 512): a block is 16 waves that ALL hold the same 64 x PPT particles; the instances of a plate
  // are dealt to the waves in contiguous chunks and the waves' partial sums meet in LDS (fixed order: deterministic), so a program
  // with few particles and many instances still fills the SIMDs.  Sites outside plates are computed by every wave, stored by wave 0
  bool wide = false;
  int lpp = 1;                      // wide flavour: lanes per particle (ppt code | 1024: 4, | 2048: 16) — FEW particles over very many instances:
                                    // a wave holds 64 / lpp particles, the instances are dealt to 16 x lpp chunks (wave, lane within the particle)
  int find(int kind, int off, int n, int len = 0, int dim = 0) {
    for (auto& c : comps) if (c.kind == kind && c.off == off && c.n == n && c.len == len && c.dim == dim) return c.at;
    Companion c{kind, off, n, comp_floats, len, dim};
    comp_floats += kind == 2 ? n + 1 : n;
    comps.push_back(c);
    return c.at;
  }
};

constexpr int kMaxExpandDim = 32;
constexpr int kMaxLdsTab = 12288;   // floats of the table copied to LDS (48 KB)

bool is_normal(int k) { return k == GJX_NORMAL || k == GJX_MVNORMAL_DIAG; }
bool is_categorical(int k) { return k == GJX_CATEGORICAL_LOGITS || k == GJX_CATEGORICAL_PROBS; }
bool table_param(const gjx_param& p) { return (p.op == GJX_P_CONST || p.op == GJX_P_GATHER) && p.xf == GJX_XF_NONE; }
int table_range(const gjx_param& p) { return p.op == GJX_P_CONST ? p.len : p.n * p.len; }

int n_params(int kind) { return is_categorical(kind) ? 1 : gjx::kind_params(kind); }


// ---------------------------------------------------------------------------------------------------------
// Rolled Scans.  A program whose tail is a Scan of T >= 4 steps (gjx_site.scan tags, include/gjx.h "Scan steps") whose
// step descriptors are PERIODIC from step 1 on — same kinds / shapes / modes, and every slot, row and table offset of
// step t equal to step 1's plus (t - 1) times a fixed stride — is emitted as: the sites before the Scan, step 0, then
// ONE loop over t_ = 1 .. T-1 around step 1's sites.  Values of the previous and of the current step live in registers
// (slots n_pre .. n_pre + S - 1 and n_pre + S .. n_pre + 2 S - 1 of the kernel's value array); rows and table offsets
// advance with t_; the step key chains, skt = fold_in(skt, t_) (scan.py:268).  The emitted program `sites` holds the
// pre-Scan sites, step 0 and step 1 with their slots remapped to those registers.
// ---------------------------------------------------------------------------------------------------------
struct Roll {
  bool ok = false;
  int i0 = 0, m = 0, T = 0, S = 0, n_pre = 0;
  int n_post = 0;                   // sites behind the Scan (emitted after the loop)
  int n_regs = 0;                   // value registers of the emitted kernel
  unsigned scan_id = 0;
  std::vector<gjx_site> sites;      // emitted sites: [0, i0) pre, [i0, i0 + m) step 0, [i0 + m, i0 + 2 m) step 1 (loop body)
  std::vector<RollInfo> info;
};

// GJX_P_EXPR (gjx.h): node i of the block of parameter q = {op, a, b, c}, read from the HOST copy of the table at codegen time (the
// node list is program STRUCTURE: baked into the kernel and hashed into its key; the constants / weights it names are run-time table reads)
struct ExprNode { int op, a, b, c, da, db; };
ExprNode expr_node(const gjx_program* p, const gjx_param& q, int i) {
  const float* nd = p->tab + q.off + GJX_EXPR_NODE_FLOATS * i;
  return ExprNode{(int)nd[0], (int)nd[1], (int)nd[2], (int)nd[3], (int)nd[4], (int)nd[5]};
}
bool has_expr(const gjx_site* sites, int n) {
  for (int j = 0; j < n; ++j) if (sites[j].mode != GJX_MODE_INPUT) for (int k = 0; k < GJX_MAX_PARAMS; ++k) if (sites[j].p[k].op == GJX_P_EXPR) return true;
  return false;
}
// a block the emitters take: in range, SSA order, no plate strides (plates with blocks run on the interpreter), leaves in [0, n_slots)
bool expr_block_ok(const gjx_program* p, const gjx_param& q, int n_slots, int dim, bool plate = false) {   // plate: strides allowed, leaves resolved by plate_program
  if (!p->tab || q.n < 1 || q.n > GJX_EXPR_MAX_NODES || q.len < 1 || q.len > q.n || q.off < 0 || q.off + GJX_EXPR_NODE_FLOATS * q.n > p->n_tab) return false;
  if (q.len != 1 && (q.len != dim || dim > 32)) return false;
  for (int i = 0; i < q.n; ++i) {
    const ExprNode e = expr_node(p, q, i);
    auto nodeok = [&](int x) { return x >= 0 && x < i; };
    switch (e.op) {
      case GJX_E_CONST: if (e.a < 0 || e.a >= p->n_tab || (e.da != 0 && !plate)) return false; break;
      case GJX_E_VALUE: if (e.a < 0 || (!plate && (e.a >= n_slots || e.da != 0))) return false; break;
      case GJX_E_ADD: case GJX_E_SUB: case GJX_E_MUL: case GJX_E_DIV: case GJX_E_MAX: case GJX_E_MIN: case GJX_E_GT:
        if (!nodeok(e.a) || !nodeok(e.b)) return false; break;
      case GJX_E_WHERE: if (!nodeok(e.a) || !nodeok(e.b) || !nodeok(e.c)) return false; break;
      case GJX_E_LINV: if (e.c < 1 || e.c > 64 || e.a < 0 || e.a + 1 + e.c > p->n_tab || e.b < 0 || (!plate && (e.b + e.c > n_slots || e.da || e.db))) return false; break;
      case GJX_E_LINN: if (e.c < 1 || e.c > 64 || e.a < 0 || e.a + 1 + e.c > p->n_tab || e.b < 0 || e.b + e.c > i || (e.da && !plate)) return false; break;
      default: if (e.op < GJX_E_NEG || e.op > GJX_E_RECIP || !nodeok(e.a)) return false; break;
    }
  }
  return true;
}
thread_local const gjx_program* g_expr_prog = nullptr;   // the program whose table holds the blocks of the sites being emitted

bool slot_op(int op) { return op == GJX_P_VALUE || op == GJX_P_GATHER || op == GJX_P_AFFINE || op == GJX_P_VGATHER; }
bool has_vgather(const gjx_site* sites, int n) {
  for (int j = 0; j < n; ++j) for (int k = 0; k < GJX_MAX_PARAMS; ++k) if (sites[j].p[k].op == GJX_P_VGATHER) return true;
  return false;
}
// "7" / "(7)" -> 7; -1 when the element index is not a literal (a rolled site's loop variable)
int literal_index(const std::string& dx) {
  size_t a = 0, b = dx.size();
  while (a < b && (dx[a] == '(' || dx[a] == ' ')) ++a;
  while (b > a && (dx[b - 1] == ')' || dx[b - 1] == ' ')) --b;
  if (a == b) return -1;
  int v = 0;
  for (size_t i = a; i < b; ++i) { if (dx[i] < '0' || dx[i] > '9') return -1; v = v * 10 + (dx[i] - '0'); }
  return v;
}
int ref_span(const gjx_param& q) { return q.op == GJX_P_AFFINE ? q.n : (q.op == GJX_P_VALUE ? q.len : 1); }

Roll detect_roll(const gjx_program* p, bool any_stream = false) {     // any_stream: the HMC emitter (its sweep draws nothing)
  Roll r;
  if ((p->rng_mode != GJX_RNG_FLAT && !any_stream) || getenv("GJX_GEN_NO_ROLL")) return r;
  const int n = p->n_sites;
  for (int j = 0; j < n; ++j) if (p->sites[j].plate != 0) return r;      // (a rolled Scan and a plate loop in one kernel: not emitted)
  int i0 = 0;
  while (i0 < n && p->sites[i0].scan == 0) ++i0;
  if (i0 == n || GJX_SCAN_STEP(p->sites[i0].scan) != 0) return r;
  const unsigned id = GJX_SCAN_ID(p->sites[i0].scan);
  int m = 0;
  while (i0 + m < n && p->sites[i0 + m].scan == p->sites[i0].scan) ++m;
  int i1 = i0;                                              // end of the Scan: the sites behind it carry no Scan tag
  while (i1 < n && p->sites[i1].scan != 0 && GJX_SCAN_ID(p->sites[i1].scan) == id) ++i1;
  for (int j = i1; j < n; ++j) if (p->sites[j].scan != 0) return r;     // one Scan per rolled program
  if ((i1 - i0) % m != 0) return r;
  const int T = (i1 - i0) / m, n_post = n - i1;
  if (T < 4) return r;
  auto at = [&](int t, int l) -> const gjx_site& { return p->sites[i0 + t * m + l]; };
  auto mode_ok = [](int md) { return md == GJX_MODE_SAMPLE || md == GJX_MODE_OBS_TAB || md == GJX_MODE_OBS_SLOT || md == GJX_MODE_OBS_MASK; };
  for (int t = 0; t < T; ++t)
    for (int l = 0; l < m; ++l) {
      const gjx_site& s = at(t, l);
      if (s.scan != GJX_SCAN_TAG(id, t)) return r;
      if (!mode_ok(s.mode)) return r;
    }
  auto width = [&](const gjx_site& s) { return is_categorical(s.kind) ? 1 : s.dim; };
  // value slots: steps are laid out back to back, S slots each, step 0 right behind the pre-Scan slots; then the slots of
  // the sites behind the Scan; then one flag row per OBS_MASK site, in site order (program.py)
  int S = 0, base0 = -1;
  for (int l = 0; l < m; ++l) if (at(0, l).slot >= 0) { S += width(at(0, l)); if (base0 < 0 || at(0, l).slot < base0) base0 = at(0, l).slot; }
  if (S <= 0 || base0 < 0) return r;
  int post_slots = 0, nf_pre = 0, mk = 0, nf_post = 0;
  for (int j = i1; j < n; ++j) if (p->sites[j].slot >= 0) post_slots += width(p->sites[j]);
  for (int j = 0; j < i0; ++j) nf_pre += p->sites[j].mode == GJX_MODE_OBS_MASK;
  for (int l = 0; l < m; ++l) mk += at(0, l).mode == GJX_MODE_OBS_MASK;
  for (int j = i1; j < n; ++j) nf_post += p->sites[j].mode == GJX_MODE_OBS_MASK;
  const int post0 = base0 + T * S;                  // first slot of the sites behind the Scan
  const int F0 = post0 + post_slots;                // first flag row
  if (p->n_slots != F0 + nf_pre + T * mk + nf_post) return r;
  for (int j = 0; j < i0; ++j) if (p->sites[j].slot >= 0 && p->sites[j].slot + width(p->sites[j]) > base0) return r;
  const int n_pre = base0;
  auto base = [&](int t) { return base0 + t * S; };
  // every step has step 1's shape; step 0 may differ in its parameters only (the initial carry is constant)
  for (int t = 0; t < T; ++t) {
    int fi = 0;
    for (int l = 0; l < m; ++l) {
      const gjx_site &a = at(t, l), &b = at(1, l);
      if (a.kind != b.kind || a.dim != b.dim || a.mode != b.mode || a.ncat != b.ncat || a.flags != b.flags) return r;
      if ((a.slot < 0) != (b.slot < 0) || (a.slot >= 0 && a.slot - base(t) != b.slot - base(1))) return r;
      if (a.slot >= 0 && (a.slot < base(t) || a.slot + width(a) > base(t) + S)) return r;
      if (a.mode == GJX_MODE_OBS_MASK && a.obs_off != F0 + nf_pre + t * mk + fi++) return r;
    }
  }
  {   // flags of the sites in front of and behind the Scan
    int fi = 0;
    for (int j = 0; j < i0; ++j) if (p->sites[j].mode == GJX_MODE_OBS_MASK && p->sites[j].obs_off != F0 + fi++) return r;
    fi = 0;
    for (int j = i1; j < n; ++j) if (p->sites[j].mode == GJX_MODE_OBS_MASK && p->sites[j].obs_off != F0 + nf_pre + T * mk + fi++) return r;
  }
  // strides from steps 1 and 2, checked on every later step
  std::vector<RollInfo> st(m);
  for (int l = 0; l < m; ++l) {
    const gjx_site &a = at(1, l), &b = at(2, l);
    st[l].d_obs = a.mode == GJX_MODE_OBS_TAB ? b.obs_off - a.obs_off : 0;
    for (int k = 0; k < n_params(a.kind); ++k) { st[l].d_off[k] = b.p[k].off - a.p[k].off; st[l].d_moff[k] = b.p[k].moff - a.p[k].moff; }
  }
  for (int t = 1; t < T; ++t)
    for (int l = 0; l < m; ++l) {
      const gjx_site &a = at(1, l), &b = at(t, l);
      if (a.mode == GJX_MODE_OBS_TAB && b.obs_off != a.obs_off + (t - 1) * st[l].d_obs) return r;
      for (int k = 0; k < n_params(a.kind); ++k) {
        const gjx_param &qa = a.p[k], &qb = b.p[k];
        if (qa.op != qb.op || qa.xf != qb.xf || qa.len != qb.len || qa.n != qb.n) return r;
        if (qa.op == GJX_P_EXPR) {
          // an expression block: step t's node list is step 1's with the leaves that read Scan slots moved by (t - 1) S and the table
          // entries moved by a fixed stride per node (taken from steps 1 and 2)
          if (!p->tab || qa.n < 1 || qa.n > GJX_EXPR_MAX_NODES || qa.off < 0 || qa.off + GJX_EXPR_NODE_FLOATS * qa.n > p->n_tab ||
              qb.off < 0 || qb.off + GJX_EXPR_NODE_FLOATS * qb.n > p->n_tab) return r;
          if (t == 1) {
            const gjx_param& q2 = at(2, l).p[k];
            if (q2.op != GJX_P_EXPR || q2.n != qa.n || q2.off < 0 || q2.off + GJX_EXPR_NODE_FLOATS * q2.n > p->n_tab) return r;
            st[l].enode_dtab[k].assign(qa.n, 0);
            for (int i = 0; i < qa.n; ++i) {
              const ExprNode ea = expr_node(p, qa, i), e2 = expr_node(p, q2, i);
              if (ea.op == GJX_E_CONST || ea.op == GJX_E_LINV || ea.op == GJX_E_LINN) st[l].enode_dtab[k][i] = e2.a - ea.a;
            }
          }
          for (int i = 0; i < qa.n; ++i) {
            const ExprNode ea = expr_node(p, qa, i), eb = expr_node(p, qb, i);
            if (ea.op != eb.op || ea.c != eb.c || ea.da || ea.db || eb.da || eb.db) return r;
            auto moves_ok = [&](int sa, int sb, int span) {     // a slot reference of step 1 (sa) and of step t (sb)
              const bool pre = sa + span <= n_pre;
              if (pre ? sb != sa : sb != sa + (t - 1) * S) return false;
              if (pre) return true;
              return (sa >= base(0) && sa + span <= base(0) + S) || (sa >= base(1) && sa + span <= base(1) + S);
            };
            switch (ea.op) {
              case GJX_E_CONST: if (eb.a != ea.a + (t - 1) * st[l].enode_dtab[k][i]) return r; break;
              case GJX_E_VALUE: if (!moves_ok(ea.a, eb.a, 1)) return r; break;
              case GJX_E_LINV: if (eb.a != ea.a + (t - 1) * st[l].enode_dtab[k][i] || !moves_ok(ea.b, eb.b, ea.c)) return r; break;
              case GJX_E_LINN: if (eb.a != ea.a + (t - 1) * st[l].enode_dtab[k][i] || eb.b != ea.b) return r; break;
              default: if (eb.a != ea.a || eb.b != ea.b) return r; break;
            }
          }
          continue;
        }
        if (qa.op != GJX_P_VALUE && qb.off != qa.off + (t - 1) * st[l].d_off[k]) return r;
        if (qa.op == GJX_P_AFFINE && qb.moff != qa.moff + (t - 1) * st[l].d_moff[k]) return r;
        // a row gather of a latent vector: the vector lives in front of the Scan (registers that do not move with the step)
        if (qa.op == GJX_P_VGATHER && (qb.moff != qa.moff || qa.moff < 0 || qa.moff + qa.n * qa.len > n_pre)) return r;
        if (slot_op(qa.op)) {
          const bool pre = qa.slot + ref_span(qa) <= n_pre;
          if (pre ? qb.slot != qa.slot : qb.slot != qa.slot + (t - 1) * S) return r;
          // a reference that moves with the step stays inside the previous or inside the current step
          if (!pre) {
            const bool in_prev = qa.slot >= base(0) && qa.slot + ref_span(qa) <= base(0) + S;
            const bool in_cur = qa.slot >= base(1) && qa.slot + ref_span(qa) <= base(1) + S;
            if (!in_prev && !in_cur) return r;
          }
        }
      }
    }
  // step 0's references: pre-Scan slots or its own step
  for (int l = 0; l < m; ++l)
    for (int k = 0; k < n_params(at(0, l).kind); ++k) {
      const gjx_param& q = at(0, l).p[k];
      if (q.op == GJX_P_EXPR) {
        if (!p->tab || q.n < 1 || q.n > GJX_EXPR_MAX_NODES || q.off < 0 || q.off + GJX_EXPR_NODE_FLOATS * q.n > p->n_tab) return r;
        for (int i = 0; i < q.n; ++i) {
          const ExprNode e = expr_node(p, q, i);
          if (e.da || e.db) return r;
          if (e.op != GJX_E_VALUE && e.op != GJX_E_LINV) continue;
          const int sa = e.op == GJX_E_VALUE ? e.a : e.b, span = e.op == GJX_E_VALUE ? 1 : e.c;
          if (!(sa + span <= n_pre) && !(sa >= base(0) && sa + span <= base(0) + S)) return r;
        }
        continue;
      }
      if (q.op == GJX_P_VGATHER && (q.moff < 0 || q.moff + q.n * q.len > n_pre)) return r;
      if (!slot_op(q.op)) continue;
      const bool pre = q.slot + ref_span(q) <= n_pre;
      const bool own = q.slot >= base(0) && q.slot + ref_span(q) <= base(0) + S;
      if (!pre && !own) return r;
    }
  // the sites behind the Scan: their own slots follow the Scan's; they read pre-Scan slots, the LAST step (still in
  // registers when the loop ends) or each other
  for (int j = i1; j < n; ++j) {
    const gjx_site& sj = p->sites[j];
    if (!mode_ok(sj.mode)) return r;
    if (sj.slot >= 0 && (sj.slot < post0 || sj.slot + width(sj) > F0)) return r;
    for (int k = 0; k < n_params(sj.kind); ++k) {
      const gjx_param& q = sj.p[k];
      if (q.op == GJX_P_EXPR) {
        if (!p->tab || q.n < 1 || q.n > GJX_EXPR_MAX_NODES || q.off < 0 || q.off + GJX_EXPR_NODE_FLOATS * q.n > p->n_tab) return r;
        for (int i = 0; i < q.n; ++i) {
          const ExprNode e = expr_node(p, q, i);
          if (e.da || e.db) return r;
          if (e.op != GJX_E_VALUE && e.op != GJX_E_LINV) continue;
          const int sa = e.op == GJX_E_VALUE ? e.a : e.b, span = e.op == GJX_E_VALUE ? 1 : e.c;
          if (!(sa + span <= n_pre) && !(sa >= base(T - 1) && sa + span <= base(T - 1) + S) && !(sa >= post0 && sa + span <= F0)) return r;
        }
        continue;
      }
      if (q.op == GJX_P_VGATHER && (q.moff < 0 || q.moff + q.n * q.len > n_pre)) return r;
      if (!slot_op(q.op)) continue;
      const bool pre = q.slot + ref_span(q) <= n_pre;
      const bool last = q.slot >= base(T - 1) && q.slot + ref_span(q) <= base(T - 1) + S;
      const bool post = q.slot >= post0 && q.slot + ref_span(q) <= F0;
      if (!pre && !last && !post) return r;
    }
  }
  // ---- the emitted program.  Registers: [0, n_pre) pre-Scan values | [n_pre, n_pre + S) previous step | [.., n_pre + 2 S)
  //      current step | post-Scan values | flags: pre-Scan sites, the current step's, post-Scan sites
  const int reg_post0 = n_pre + 2 * S, reg_f0 = reg_post0 + post_slots;
  auto remap = [&](int ref, int tau) -> int {     // a slot referenced from step tau -> register slot
    if (ref < n_pre) return ref;
    if (ref >= base(tau) && ref < base(tau) + S) return n_pre + S + (ref - base(tau));   // current step
    return n_pre + (ref - base(tau - 1));                                                 // previous step
  };
  auto remap_post = [&](int ref) -> int {         // a slot referenced from behind the Scan
    if (ref < n_pre) return ref;
    if (ref >= post0) return reg_post0 + (ref - post0);
    return n_pre + (ref - base(T - 1));           // the last step: the loop's final carry left it in the "previous" registers
  };
  // the leaves of a site's expression blocks -> the registers of the emitted program (RollInfo::eleaf)
  auto expr_leaves = [&](const gjx_site& s, RollInfo& ri, const std::function<int(int)>& map) {
    if (s.mode == GJX_MODE_INPUT) return;
    for (int k = 0; k < n_params(s.kind); ++k) {
      const gjx_param& q = s.p[k];
      if (q.op != GJX_P_EXPR) continue;
      ri.eleaf[k].clear();
      ri.eleaf_at[k].assign(q.n, -1);
      for (int i = 0; i < q.n; ++i) {
        const ExprNode e = expr_node(p, q, i);
        if (e.op != GJX_E_VALUE && e.op != GJX_E_LINV) continue;
        ri.eleaf_at[k][i] = (int)ri.eleaf[k].size();
        for (int t2 = 0; t2 < (e.op == GJX_E_VALUE ? 1 : e.c); ++t2) { RollInfo::ExprLeaf lf; lf.reg = map((e.op == GJX_E_VALUE ? e.a : e.b) + t2); ri.eleaf[k].push_back(lf); }
      }
    }
  };
  {
    int fi = 0;
    for (int j = 0; j < i0; ++j) {
      gjx_site s = p->sites[j];
      RollInfo ri; ri.row = s.slot; ri.score_row = j;
      if (s.mode == GJX_MODE_OBS_MASK) { ri.flag_row = s.obs_off; s.obs_off = reg_f0 + fi++; ri.load_here = true; }
      expr_leaves(s, ri, [](int ref) { return ref; });
      r.sites.push_back(s);
      r.info.push_back(ri);
    }
  }
  for (int tau = 0; tau < 2; ++tau) {
    int fi = 0;
    for (int l = 0; l < m; ++l) {
      gjx_site s = at(tau, l);
      RollInfo ri = tau ? st[l] : RollInfo();
      ri.row = s.slot; ri.score_row = i0 + tau * m + l;
      if (tau) { ri.d_row = S; ri.d_score_row = m; }
      if (s.mode == GJX_MODE_OBS_SLOT || s.mode == GJX_MODE_OBS_MASK) ri.load_here = true;
      if (s.mode == GJX_MODE_OBS_MASK) { ri.flag_row = s.obs_off; ri.d_flag_row = tau ? mk : 0; s.obs_off = reg_f0 + nf_pre + fi++; }
      if (s.slot >= 0) s.slot = remap(s.slot, tau);
      for (int k = 0; k < n_params(s.kind); ++k) if (slot_op(s.p[k].op)) s.p[k].slot = remap(s.p[k].slot, tau);
      expr_leaves(s, ri, [&](int ref) { return remap(ref, tau); });
      if (!tau) for (int k = 0; k < 4; ++k) ri.enode_dtab[k].clear();        // (step 0 is emitted outside the loop: nothing strides)
      r.sites.push_back(s);
      r.info.push_back(ri);
    }
  }
  {
    int fi = 0;
    for (int j = i1; j < n; ++j) {
      gjx_site s = p->sites[j];
      RollInfo ri; ri.row = s.slot; ri.score_row = j;
      if (s.mode == GJX_MODE_OBS_SLOT || s.mode == GJX_MODE_OBS_MASK) ri.load_here = true;
      if (s.mode == GJX_MODE_OBS_MASK) { ri.flag_row = s.obs_off; s.obs_off = reg_f0 + nf_pre + mk + fi++; }
      if (s.slot >= 0) s.slot = remap_post(s.slot);
      for (int k = 0; k < n_params(s.kind); ++k) if (slot_op(s.p[k].op)) s.p[k].slot = remap_post(s.p[k].slot);
      expr_leaves(s, ri, [&](int ref) { return remap_post(ref); });
      r.sites.push_back(s);
      r.info.push_back(ri);
    }
  }
  r.ok = true; r.i0 = i0; r.m = m; r.T = T; r.S = S; r.n_pre = n_pre; r.scan_id = id; r.n_post = n_post;
  r.n_regs = reg_f0 + nf_pre + mk + nf_post;
  return r;
}

// ---------------------------------------------------------------------------------------------------------
// Plates (gjx.h "Plates").  The m body sites of a plate are emitted ONCE, inside `for (i_ = 0; i_ < n; ++i_)`: the values of
// the CURRENT instance live in registers, their rows in choices[][] / offsets in the table advance with i_.  The emitted
// program holds every site with its slots remapped to registers (a body site owns one instance's worth); a parameter whose
// source is not in a register — instances of an earlier plate, one instance of a plate read from outside it — reads
// choices[][] (written earlier by the same lane).
// ---------------------------------------------------------------------------------------------------------
struct PlateXf {
  bool any = false, ok = false;
  std::vector<gjx_site> sites;
  std::vector<RollInfo> info;
  int n_regs = 0;
};

PlateXf plate_program(const gjx_program* p) {
  PlateXf x;
  const int n = p->n_sites;
  for (int j = 0; j < n; ++j) x.any = x.any || p->sites[j].plate != 0;
  if (!x.any) return x;
  auto width = [&](const gjx_site& s) { return (is_categorical(s.kind) && s.mode != GJX_MODE_INPUT) ? 1 : s.dim; };
  auto rows = [&](const gjx_site& s) { return width(s) * (s.plate ? s.plate_n : 1); };
  std::vector<int> reg(n, -1), flag(n, -1), first(n, 0), pos(n, 0);
  int next = 0;
  for (int j = 0; j < n; ++j) {
    const gjx_site& s = p->sites[j];
    if (s.plate && (s.plate_n < 1 || s.kind == GJX_DIRICHLET)) return x;
    if (s.plate) { first[j] = (j > 0 && p->sites[j - 1].plate == s.plate) ? first[j - 1] : j; pos[j] = j - first[j]; }
    if (s.slot >= 0) { reg[j] = next; next += width(s); }
  }
  for (int j = 0; j < n; ++j) if (p->sites[j].mode == GJX_MODE_OBS_MASK) flag[j] = next++;
  x.n_regs = next;
  auto owner_of = [&](int slot, int before) {
    for (int o = 0; o < before; ++o) { const gjx_site& so = p->sites[o]; if (so.slot >= 0 && slot >= so.slot && slot < so.slot + rows(so)) return o; }
    return -1;
  };
  for (int j = 0; j < n; ++j) {
    gjx_site s = p->sites[j];
    RollInfo ri;
    ri.row = s.slot; ri.score_row = j;
    if (s.plate) {
      ri.plate = true; ri.plate_n = s.plate_n; ri.plate_l = pos[j]; ri.plate_j0 = first[j];
      ri.d_row = width(s); ri.d_obs = s.mode == GJX_MODE_OBS_TAB ? s.d_obs : 0;
      for (int k = 0; k < n_params(s.kind); ++k) { ri.d_off[k] = s.p[k].d_off; ri.d_moff[k] = s.p[k].d_moff; }
    }
    if (s.mode == GJX_MODE_OBS_SLOT || s.mode == GJX_MODE_OBS_MASK) ri.load_here = true;   // (registers are not row numbers here)
    if (s.mode == GJX_MODE_OBS_MASK) { ri.flag_row = s.obs_off; ri.d_flag_row = s.plate ? s.d_obs : 0; s.obs_off = flag[j]; }
    for (int k = 0; k < (s.mode == GJX_MODE_INPUT ? 0 : n_params(s.kind)); ++k) {
      gjx_param& q = s.p[k];
      if (q.op == GJX_P_EXPR) {
        // every leaf of the block like a source of the closed forms: a register (a site outside the plates; an earlier site of the
        // SAME instance) or rows of choices[][] (instances of another plate, elements of a vector site picked by the instance)
        if (!p->tab || q.n < 1 || q.n > GJX_EXPR_MAX_NODES || q.off < 0 || q.off + GJX_EXPR_NODE_FLOATS * q.n > p->n_tab) return x;
        ri.eleaf_at[k].assign(q.n, -1);
        for (int i = 0; i < q.n; ++i) {
          const ExprNode e = expr_node(p, q, i);
          if (e.op != GJX_E_VALUE && e.op != GJX_E_LINV) continue;
          const int base = e.op == GJX_E_VALUE ? e.a : e.b, dsl = e.op == GJX_E_VALUE ? e.da : e.db, cnt = e.op == GJX_E_VALUE ? 1 : e.c;
          if (!s.plate && dsl != 0) return x;
          ri.eleaf_at[k][i] = (int)ri.eleaf[k].size();
          for (int t = 0; t < cnt; ++t) {
            RollInfo::ExprLeaf lf;
            const int o = owner_of(base + t, j);
            if (o < 0) return x;
            const gjx_site& so = p->sites[o];
            const int off = base + t - so.slot;
            if (so.plate == 0 && dsl == 0) lf.reg = reg[o] + off;
            else if (so.plate != 0 && s.plate == so.plate && dsl == width(so) && off < width(so)) lf.reg = reg[o] + off;
            else { lf.mem_row = base + t; lf.mem_stride = s.plate ? dsl : 0; }
            ri.eleaf[k].push_back(lf);
          }
        }
        continue;
      }
      if (q.op == GJX_P_VGATHER) {
        // the indexed choice: all its rows in the registers of ONE site outside the plates, the same for every instance
        const int ov = owner_of(q.moff, j);
        if (ov < 0 || p->sites[ov].plate != 0 || q.d_moff != 0 || q.moff + q.n * q.len > p->sites[ov].slot + rows(p->sites[ov])) return x;
        q.moff = reg[ov] + (q.moff - p->sites[ov].slot);
        if (q.slot < 0) continue;        // (the index comes from the table: RollInfo::d_off is its stride)
      }
      if (!slot_op(q.op)) continue;
      const int span = ref_span(q);
      const int o = owner_of(q.slot, j);
      if (o < 0) return x;
      const gjx_site& so = p->sites[o];
      const int off = q.slot - so.slot;
      if (so.plate == 0 && q.d_slot != 0) {
        // element i_ of a VECTOR site outside the plate (an earlier plate in its vector form): registers cannot be indexed by
        // i_, the rows in choices[][] can (the site stored them before this plate started)
        ri.mem_slot[k] = q.slot; ri.d_mem[k] = q.d_slot;
        q.slot = 0;
      } else if (so.plate == 0) {
        // a span may run over several consecutive non-plate sites (an affine form over several sites): registers must follow
        for (int e = 0; e < span; ++e) {
          const int oe = owner_of(q.slot + e, j);
          if (oe < 0 || p->sites[oe].plate != 0 || reg[oe] + (q.slot + e - p->sites[oe].slot) != reg[o] + off + e) return x;
        }
        q.slot = reg[o] + off;
      } else if (s.plate == so.plate && q.d_slot == width(so) && off + span <= width(so)) {
        q.slot = reg[o] + off;                     // an earlier site of the SAME instance
      } else {
        if (off % width(so) + span > width(so) && q.op != GJX_P_AFFINE) return x;
        ri.mem_slot[k] = q.slot; ri.d_mem[k] = s.plate ? q.d_slot : 0;
        q.slot = 0;
      }
    }
    if (s.slot >= 0) s.slot = reg[j];
    x.sites.push_back(s);
    x.info.push_back(ri);
  }
  x.ok = true;
  return x;
}

// what the emitter covers; everything else runs on the interpreter
bool supported_sites(const gjx_site* sites, int n_sites, int n_slots, const gjx_program* prog = nullptr, bool plate_prog = false) {   // prog: the program whose table holds expression blocks
  if (n_sites < 1 || n_sites > 48 || n_slots > 160) return false;
  int total = 0;
  for (int j = 0; j < n_sites; ++j) {
    const gjx_site& s = sites[j];
    if (s.mode == GJX_MODE_INPUT) { if (s.dim < 1 || s.dim > kMaxExpandDim || s.slot < 0) return false; total += s.dim; continue; }
    if (s.kind == GJX_DIRICHLET || s.kind < 1 || s.kind >= GJX_KIND_MAX) return false;
    if (is_categorical(s.kind)) {
      if (s.p[0].op != GJX_P_CONST && s.p[0].op != GJX_P_GATHER) return false;
      if (s.ncat < 1 || s.ncat > 64) return false;
      total += s.ncat;
      continue;
    }
    if (s.dim < 1) return false;
    if (s.dim > kMaxExpandDim && s.mode != GJX_MODE_OBS_TAB) return false;
    for (int k = 0; k < n_params(s.kind); ++k) {
      const gjx_param& q = s.p[k];
      if (q.op == GJX_P_EXPR) {      // an expression block: emitted inline, one straight-line copy per element that reads it
        if (!prog || !expr_block_ok(prog, q, n_slots, s.dim, plate_prog) || (s.dim > kMaxExpandDim && q.len != 1)) return false;
        total += q.n / 2;
        continue;
      }
      if (q.op < GJX_P_CONST || q.op > GJX_P_VGATHER) return false;
      if (q.op == GJX_P_AFFINE && (q.n < 1 || q.n > 64)) return false;
      if (s.dim > kMaxExpandDim && q.op == GJX_P_VALUE && q.len != 1) return false;
      // a row of a choice in registers picked by a select chain: few rows, and a literal element index unless rows are scalars
      if (q.op == GJX_P_VGATHER && (q.n < 1 || q.n * q.len > 32 || q.moff < 0 || q.moff + q.n * q.len > n_slots || (q.len != 1 && s.dim > kMaxExpandDim))) return false;
    }
    total += s.dim > kMaxExpandDim ? 8 : s.dim;
  }
  return total <= 320;
}

bool supported_uncached(const gjx_program* p) {
  const PlateXf px = plate_program(p);
  if (px.any) return px.ok && supported_sites(px.sites.data(), (int)px.sites.size(), px.n_regs, p, true);
  if (p->n_sites >= 1 && supported_sites(p->sites, p->n_sites, p->n_slots, p)) return true;
  const Roll r = detect_roll(p);      // a long periodic Scan is emitted as a loop
  return r.ok && supported_sites(r.sites.data(), (int)r.sites.size(), r.n_regs, p, true);
}

// index into the float table of parameter q at element `d` (a C expression; `dx` is the element index expression)
std::string tab_index(const gjx_param& q, const std::string& dx, int site, int k, int d_off = 0) {
  char b[256];
  const std::string e = q.len == 1 ? "0" : ("(" + dx + ") % " + std::to_string(q.len));
  const std::string off = toff(q.off, d_off);
  if (q.op == GJX_P_CONST) snprintf(b, sizeof(b), "%s + %s", off.c_str(), e.c_str());
  else snprintf(b, sizeof(b), "%s + gi_%d_%d[p] * %d + %s", off.c_str(), site, k, q.len, e.c_str());
  return b;
}

// value of parameter q at element dx for particle p (before the transform)
std::string param_expr(const gjx_param& q, const std::string& dx, int site, int k, const RollInfo& ri) {
  char b[512];
  switch (q.op) {
    case GJX_P_CONST:
    case GJX_P_GATHER: return "TAB(" + tab_index(q, dx, site, k, ri.d_off[k]) + ")";
    case GJX_P_VALUE:
      if (q.len == 1) return src_val(q, ri, k, "0");
      return src_val(q, ri, k, "(" + dx + ") % " + std::to_string(q.len));
    case GJX_P_VGATHER: snprintf(b, sizeof(b), "vg_%d_%d", site, k); return b;   // (select chain emitted by emit_param_pre)
    case GJX_P_EXPR: snprintf(b, sizeof(b), "ex_%d_%d", site, k); return b;      // (the block's nodes emitted by emit_param_pre)
    default: snprintf(b, sizeof(b), "aff_%d_%d", site, k); return b;   // computed into a local just before use
  }
}
std::string xf_wrap(int xf, const std::string& e) {
  switch (xf) {
    case GJX_XF_EXP: return "fast_exp(" + e + ")";
    case GJX_XF_SOFTPLUS: return "softplus(" + e + ")";
    case GJX_XF_SIGMOID: return "sigmoid(" + e + ")";
    default: return e;
  }
}

// GJX_P_EXPR: the nodes the output `out` depends on, as straight-line statements `const float <pfx><i> = ...;` (the compiler
// schedules and merges them with the site's own arithmetic; values are in registers: val(slot) is `vfmt` with the slot number).
// -> the set of emitted nodes.  The unary forms use the device header's helpers (the interpreter's expr_unary computes the same).
std::vector<char> emit_expr_nodes(Emit& o, const gjx_program* prog, const gjx_param& q, int out, const std::string& pfx, const char* ind,
                                  const std::function<std::string(int, int)>& val,      // val(node, element of the node's leaf span)
                                  const std::vector<int>* dtab = nullptr) {              // a rolled Scan's per-node table strides (RollInfo::enode_dtab)
  std::vector<char> need(q.n, 0);
  need[out] = 1;
  for (int i = out; i >= 0; --i) {
    if (!need[i]) continue;
    const ExprNode e = expr_node(prog, q, i);
    switch (e.op) {
      case GJX_E_CONST: case GJX_E_VALUE: case GJX_E_LINV: break;
      case GJX_E_ADD: case GJX_E_SUB: case GJX_E_MUL: case GJX_E_DIV: case GJX_E_MAX: case GJX_E_MIN: case GJX_E_GT: need[e.a] = need[e.b] = 1; break;
      case GJX_E_WHERE: need[e.a] = need[e.b] = need[e.c] = 1; break;
      case GJX_E_LINN: for (int t = 0; t < e.c; ++t) need[e.b + t] = 1; break;
      default: need[e.a] = 1; break;
    }
  }
  auto N = [&](int i) { return pfx + std::to_string(i); };
  // table entry `at` of a node whose table base advances by `stride` per plate instance / rolled step (toff: the loop variable)
  auto T = [&](int base, int stride, int at) { return "TAB(" + toff(base, stride) + (at ? " + " + std::to_string(at) : "") + ")"; };
  for (int i = 0; i <= out; ++i) {
    if (!need[i]) continue;
    ExprNode e = expr_node(prog, q, i);
    if (dtab && !dtab->empty()) e.da = (*dtab)[i];
    std::string r;
    const std::string A = e.op >= GJX_E_ADD && e.op != GJX_E_LINV && e.op != GJX_E_LINN ? N(e.a) : "", B = N(e.b);
    switch (e.op) {
      case GJX_E_CONST: r = T(e.a, e.da, 0); break;
      case GJX_E_VALUE: r = val(i, 0); break;
      case GJX_E_ADD: r = A + " + " + B; break;
      case GJX_E_SUB: r = A + " - " + B; break;
      case GJX_E_MUL: r = A + " * " + B; break;
      case GJX_E_DIV: r = A + " * fast_rcp(" + B + ")"; break;
      case GJX_E_MAX: r = A + " >= " + B + " ? " + A + " : " + B; break;
      case GJX_E_MIN: r = A + " <= " + B + " ? " + A + " : " + B; break;
      case GJX_E_GT: r = A + " > " + B + " ? 1.0f : 0.0f"; break;
      case GJX_E_WHERE: r = A + " != 0.0f ? " + B + " : " + N(e.c); break;
      case GJX_E_LINV: {
        r = T(e.a, e.da, 0);
        for (int t = 0; t < e.c; ++t) r = "fmaf(" + T(e.a, e.da, 1 + t) + ", " + val(i, t) + ", " + r + ")";
        break;
      }
      case GJX_E_LINN: {
        r = T(e.a, e.da, 0);
        for (int t = 0; t < e.c; ++t) r = "fmaf(" + T(e.a, e.da, 1 + t) + ", " + N(e.b + t) + ", " + r + ")";
        break;
      }
      default: r = "expr_unary(" + std::to_string(e.op) + ", " + A + ")"; break;
    }
    o.f("%sconst float %s = %s;\n", ind, N(i).c_str(), r.c_str());
  }
  return need;
}

// register that holds element t of the leaf span of `node` (VALUE: t = 0; LINV: t < count) of parameter k's block: the slot number, or —
// a plate program — what plate_program resolved; -1: the leaf lives in memory (rows of another plate)
int expr_leaf_reg(const gjx_program* prog, const gjx_param& q, const RollInfo& ri, int k, int node, int t) {
  const ExprNode e = expr_node(prog, q, node);
  if (ri.eleaf_at[k].empty()) return (e.op == GJX_E_VALUE ? e.a : e.b) + t;
  return ri.eleaf[k][ri.eleaf_at[k][node] + t].reg;
}

// the leaf reader of parameter k's block for the propagate emitters: registers by slot number, or — a plate program — what
// plate_program resolved (RollInfo::eleaf): a register, or rows of choices[][] that advance with the instance
std::function<std::string(int, int)> expr_leaf_reader(const gjx_program* prog, const gjx_param& q, const RollInfo& ri, int k) {
  return [prog, &q, &ri, k](int node, int t) -> std::string {
    const ExprNode e = expr_node(prog, q, node);
    if (ri.eleaf_at[k].empty()) return "v[" + std::to_string((e.op == GJX_E_VALUE ? e.a : e.b) + t) + "][p]";
    const RollInfo::ExprLeaf& lf = ri.eleaf[k][ri.eleaf_at[k][node] + t];
    if (lf.reg >= 0) return "v[" + std::to_string(lf.reg) + "][p]";
    return "a.choices[(int64_t)" + toff(lf.mem_row, lf.mem_stride) + " * K + i0 + p]";
  };
}

// statements that must precede the use of param_expr for element dx (affine accumulations)
void emit_param_pre(Emit& o, const gjx_param& q, const std::string& dx, int site, int k, const char* ind, const RollInfo& ri) {
  if (q.op == GJX_P_EXPR) {
    // (supported_sites: a block with several outputs only under literal element indices; no plate / roll remapping of its slots)
    const int out = q.n - q.len + (q.len == 1 ? 0 : literal_index(dx) % q.len);
    const std::string pfx = "en_" + std::to_string(site) + "_" + std::to_string(k) + "_";
    emit_expr_nodes(o, g_expr_prog, q, out, pfx, ind, expr_leaf_reader(g_expr_prog, q, ri, k), &ri.enode_dtab[k]);
    o.f("%sconst float ex_%d_%d = %s%d;\n", ind, site, k, pfx.c_str(), out);
    return;
  }
  if (q.op == GJX_P_VGATHER) {
    // row gi_ of a choice whose rows live in registers: registers cannot be indexed, n - 1 selects can (supported_sites: n len <= 32)
    const int e = q.len == 1 ? 0 : literal_index(dx) % q.len;
    o.f("%sfloat vg_%d_%d = v[%d][p];\n", ind, site, k, q.moff + e);
    for (int c = 1; c < q.n; ++c) o.f("%svg_%d_%d = gi_%d_%d[p] == %d ? v[%d][p] : vg_%d_%d;\n", ind, site, k, site, k, c, q.moff + c * q.len + e, site, k);
    return;
  }
  if (q.op != GJX_P_AFFINE) return;
  const std::string e = q.len == 1 ? "0" : ("(" + dx + ") % " + std::to_string(q.len));
  o.f("%sfloat aff_%d_%d = TAB(%s + %s);\n", ind, site, k, toff(q.off, ri.d_off[k]).c_str(), e.c_str());
  o.f("%s{ const int r_ = %s + (%s) * %d;\n", ind, toff(q.moff, ri.d_moff[k]).c_str(), dx.c_str(), q.n);
  o.f("%s  _Pragma(\"unroll\") for (int e_ = 0; e_ < %d; ++e_) aff_%d_%d = fmaf(TAB(r_ + e_), %s, aff_%d_%d); }\n", ind, q.n, site,
      k, src_val(q, ri, k, "e_").c_str(), site, k);
}

void emit_gather_index(Emit& o, const gjx_site& s, int site, int np, const RollInfo& ri) {
  for (int k = 0; k < np; ++k) {
    const gjx_param& q = s.p[k];
    if (q.op != GJX_P_GATHER && q.op != GJX_P_VGATHER) continue;
    o.f("      int gi_%d_%d[PPT];\n", site, k);
    const std::string from = (q.op == GJX_P_VGATHER && q.slot < 0) ? "TAB(" + toff(q.off, ri.d_off[k]) + ")" : src_val(q, ri, k, "0");
    o.f("      PLOOP { int g_ = (int)%s; gi_%d_%d[p] = g_ < 0 ? 0 : (g_ > %d ? %d : g_); }\n", from.c_str(), site, k, q.n - 1, q.n - 1);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Big affine sites on the matrix cores.  An observed site with more than kMaxExpandDim elements whose first parameter is an
// AFFINE form over n = 16, 32, 48 or 64 values (the N rows of a regression likelihood: logits = X beta) is a [dim x n] x
// [n x particles] contraction: emitted on v_mfma_f32_16x16x4_f32 with the particles of a wave as 4 column groups of 16.
// The reference gets any model's contraction fused by XLA (inference/requests/hmc.py:70-96, smc.py:298-315 under jit).
//   layout (as the hand-written k_hmc_logreg_mfma2, gjx_hmc.hip): lane l = (c = l & 15, q = l >> 4); for group g the B operand
//   of step s is coefficient (n/4) q + s of particle 16 g + c — fetched once per site from a per-wave LDS patch into which
//   every lane wrote its particle's n values; the A operand of step s is X[n0 + c][(n/4) q + s], one b128 LDS read per four
//   steps, SHARED by the four groups; the result tile (C layout) is logits[n0 + 4 q + r][particle 16 g + c], r = 0..3.
//   The elementwise log-density is then taken in that layout and the four lanes of a column add up their partial sums.
// ---------------------------------------------------------------------------------------------------------
bool mfma_site_ok(const gjx_site& s, const RollInfo& ri) {
  if (s.mode != GJX_MODE_OBS_TAB || s.dim <= kMaxExpandDim || s.dim < 64 || is_categorical(s.kind) || s.kind == GJX_DIRICHLET) return false;
  if (ri.plate || ri.d_obs || ri.d_row || ri.mem_slot[0] >= 0) return false;
  const gjx_param& q = s.p[0];
  if (q.op != GJX_P_AFFINE || q.xf != GJX_XF_NONE || q.n < 16 || q.n > 64 || (q.n & 15) || (q.moff & 3) || ri.d_off[0] || ri.d_moff[0]) return false;
  if (q.len != 1 && q.len != s.dim) return false;
  for (int k = 1; k < n_params(s.kind); ++k) if (s.p[k].op != GJX_P_CONST || ri.d_off[k]) return false;
  return true;
}

bool has_mfma_site(const gjx_program* p) {
  if (getenv("GJX_GEN_NO_MFMA") || p->n_tab > 30000) return false;      // table + patches in static LDS (<= 150 KB)
  for (int j = 0; j < p->n_sites; ++j) { RollInfo ri; if (p->sites[j].plate == 0 && mfma_site_ok(p->sites[j], ri)) return true; }
  return false;
}

void emit_mfma_site(Emit& o, Plan& pl, int j) {
  const gjx_site& s = pl.prog->sites[j];
  const RollInfo& ri = pl.info[j];
  const gjx_param& q = s.p[0];
  const int n = q.n, M = n / 4, dim = s.dim, np = n_params(s.kind);
  o.f("    { // ---- site %d: kind %d, %d rows x %d coefficients on the matrix cores (v_mfma_f32_16x16x4_f32)\n", j, s.kind, dim, n);
  o.f("      float lp[PPT];\n      const int lane_ = threadIdx.x & 63, c_ = lane_ & 15, q_ = lane_ >> 4;\n");
  // (1) lane = particle -> C layout: group g, step s needs coefficient M q + s of particle 16 g + c, i.e. register (M q + s) of lane
  //     16 g + c — one cross-lane read (ds_bpermute: no LDS allocation, the table needs the space) per candidate q, the lane keeps
  //     its own q's.  Once per particle and site: 4 n reads against dim / 16 tiles of 4 n / 4 matrix instructions each.
  o.f("      float bg_[4][%d];\n", M);
  const bool bern = s.kind == GJX_BERNOULLI_LOGITS;
  for (int g = 0; g < 4; ++g)
    for (int st = 0; st < M; ++st) {
      o.f("      { float t_ = 0.0f;");
      for (int qq = 0; qq < 4; ++qq) o.f(" { const float u_ = __shfl(v[%d][0], %d + c_, 64); if (q_ == %d) t_ = u_; }", q.slot + M * qq + st, 16 * g, qq);
      o.f(" bg_[%d][%d] = t_%s; }\n", g, st, bern ? " * 1.44269504f" : "");     // (bernoulli: the tile is accumulated in log2 units)
    }
  o.f("      float part_[4] = {0.0f, 0.0f, 0.0f, 0.0f};\n");
  std::string pc[4] = {"0.0f", "0.0f", "0.0f", "0.0f"};
  for (int k = 1; k < np; ++k) {
    const gjx_param& qk = s.p[k];
    pc[k] = xf_wrap(qk.xf, "TAB(" + std::to_string(qk.off) + (qk.len == 1 ? "" : " + row_ % " + std::to_string(qk.len)) + ")");
  }
  const bool y128 = (s.obs_off & 3) == 0 && (dim & 15) == 0;
  const bool b128 = q.len == dim && (q.off & 3) == 0 && (dim & 15) == 0;
  // profiling variants (GJX_GEN_MFMA_DEBUG, part of the cache key): 1 = no matrix instructions, 2 = no elementwise phase,
  // 4 = no scheduling barrier behind the matrix phase, 8 = dependent matrix-instruction order, 16 / 32 = one / two row tiles per trip
  const int dbg_all = getenv("GJX_GEN_MFMA_DEBUG") ? atoi(getenv("GJX_GEN_MFMA_DEBUG")) : 0;
  const int dbg = dbg_all & 3;
  const char* bsc = bern ? " * 1.44269504f" : "";
  // TB row tiles of 16 per trip: their matrix instructions are issued as ONE batch (4 TB accumulators), then their elementwise
  // phase — the f32 matrix unit and the vector ALU of a SIMD do not overlap, and every switch between the two kinds costs
  // (profiles/r03_mfma_valu_overlap_microbench.txt), so the batches are as long as the registers allow
  const int TB = (dbg_all & 16) ? 1 : (((dbg_all & 32) || (dim % 64)) ? ((dim % 32) == 0 ? 2 : 1) : 4);
  o.f("      for (int n0_ = 0; n0_ < %d; n0_ += %d) {\n", (dim + 15) & ~15, 16 * TB);
  o.f("        float xa_[%d][%d];\n        v4f_ bias_[%d], y_[%d];\n", TB, M, TB, TB);
  for (int tb = 0; tb < TB; ++tb) {
    const std::string n0 = tb ? "(n0_ + " + std::to_string(16 * tb) + ")" : "n0_";
    o.f("        { const int ra_ = %s + c_ < %d ? %s + c_ : %d;\n", n0.c_str(), dim, n0.c_str(), dim - 1);
    o.f("          _Pragma(\"unroll\") for (int s4_ = 0; s4_ < %d; ++s4_) {\n            const v4f_ t_ = *(const v4f_*)&TAB(%d + ra_ * %d + %d * q_ + 4 * s4_);\n"
        "            xa_[%d][4 * s4_] = t_[0]; xa_[%d][4 * s4_ + 1] = t_[1]; xa_[%d][4 * s4_ + 2] = t_[2]; xa_[%d][4 * s4_ + 3] = t_[3];\n          }\n",
        M / 4, q.moff, n, M, tb, tb, tb, tb);
    if (q.len == 1) o.f("          { const float b0_ = TAB(%d)%s; bias_[%d] = v4f_{b0_, b0_, b0_, b0_}; }\n", q.off, bsc, tb);
    else if (b128) o.f("          bias_[%d] = *(const v4f_*)&TAB(%d + %s + 4 * q_)%s;\n", tb, q.off, n0.c_str(), bsc);
    else o.f("          _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) { const int row_ = %s + 4 * q_ + r_; bias_[%d][r_] = TAB(%d + (row_ < %d ? row_ : %d))%s; }\n",
             n0.c_str(), tb, q.off, dim, dim - 1, bsc);
    if (y128) o.f("          y_[%d] = *(const v4f_*)&TAB(%d + %s + 4 * q_);\n", tb, s.obs_off, n0.c_str());
    else o.f("          _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) { const int row_ = %s + 4 * q_ + r_; y_[%d][r_] = TAB(%d + (row_ < %d ? row_ : %d)); }\n",
             n0.c_str(), tb, s.obs_off, dim, dim - 1);
    o.f("        }\n");
  }
  if (bern) o.f("        v4f_ yh_[%d];\n        _Pragma(\"unroll\") for (int tb_ = 0; tb_ < %d; ++tb_) yh_[tb_] = y_[tb_] - 0.5f;\n", TB, TB);
  o.f("        v4f_ acc_[%d][4];\n        _Pragma(\"unroll\") for (int tb_ = 0; tb_ < %d; ++tb_) _Pragma(\"unroll\") for (int g_ = 0; g_ < 4; ++g_) acc_[tb_][g_] = bias_[tb_];\n"
      "        __builtin_amdgcn_sched_barrier(0);\n", TB, TB);
  if (dbg == 1)
    o.f("        _Pragma(\"unroll\") for (int tb_ = 0; tb_ < %d; ++tb_) _Pragma(\"unroll\") for (int g_ = 0; g_ < 4; ++g_) acc_[tb_][g_] += xa_[tb_][g_ %% %d] * bg_[g_][0];\n"
        "        __builtin_amdgcn_sched_barrier(0);\n", TB, M);
  else if (dbg_all & 8)
    o.f("        _Pragma(\"unroll\") for (int tb_ = 0; tb_ < %d; ++tb_) _Pragma(\"unroll\") for (int g_ = 0; g_ < 4; ++g_) _Pragma(\"unroll\") for (int s_ = 0; s_ < %d; ++s_)\n"
        "          acc_[tb_][g_] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa_[tb_][s_], bg_[g_][s_], acc_[tb_][g_], 0, 0, 0);\n        __builtin_amdgcn_sched_barrier(0);\n", TB, M);
  else
    o.f("        _Pragma(\"unroll\") for (int s_ = 0; s_ < %d; ++s_) _Pragma(\"unroll\") for (int tb_ = 0; tb_ < %d; ++tb_) _Pragma(\"unroll\") for (int g_ = 0; g_ < 4; ++g_)\n"
        "          acc_[tb_][g_] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa_[tb_][s_], bg_[g_][s_], acc_[tb_][g_], 0, 0, 0);\n        %s\n", M, TB,
        (dbg_all & 4) ? "" : "__builtin_amdgcn_sched_barrier(0);");
  o.f("        _Pragma(\"unroll\") for (int tb_ = 0; tb_ < %d; ++tb_) _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) {\n"
      "          const int row_ = n0_ + 16 * tb_ + 4 * q_ + r_; (void)row_;\n", TB);
  o.f("          const float pb_ = %s, pc_ = %s, pd_ = %s;\n", pc[1].c_str(), pc[2].c_str(), pc[3].c_str());
  o.f("          _Pragma(\"unroll\") for (int g_ = 0; g_ < 4; ++g_) {\n            const float a_ = acc_[tb_][g_][r_];\n");
  if (dbg == 2) o.f("            const float e_ = a_ + y_[tb_][r_];\n");
  else if (bern)   // y a - softplus(a) — the value of TFP's -softplus(-a) y - softplus(a) (1 - y) for any y — in log2 units (a_ = a log2 e)
    // with max(a_, 0) = (a_ + |a_|) / 2:  (y - 1/2) a_ - |a_| / 2 - log2(1 + exp2(-|a_|))   (two fused multiply-adds beside the exp2 / log2);
    // the factor ln 2 is applied once to the particle's sum
    o.f("            const float e_ = fmaf(yh_[tb_][r_], a_, fmaf(-0.5f, fabsf(a_), -__builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a_)))));\n");
  else
    o.f("            const float e_ = elem_logpdf(%d, y_[tb_][r_], a_, pb_, pc_, pd_);\n", s.kind);
  if (dim & 15) o.f("            part_[g_] += row_ < %d ? e_ : 0.0f;\n", dim);
  else o.f("            part_[g_] += e_;\n");
  o.f("          }\n        }\n        __builtin_amdgcn_sched_barrier(0);\n      }\n");
  // (3) the four lanes of a column hold partial sums of ONE particle; lane q == g keeps group g's total (it IS that particle's lane)
  o.f("      float tot_ = 0.0f;\n      _Pragma(\"unroll\") for (int g_ = 0; g_ < 4; ++g_) {\n        float t_ = part_[g_];\n"
      "        t_ += __shfl_xor(t_, 16, 64);\n        t_ += __shfl_xor(t_, 32, 64);\n        if (q_ == g_) tot_ = t_;\n      }\n      lp[0] = tot_%s;\n", bern ? " * kLn2" : "");
  o.f("      PLOOP { score[p] += lp[p]; weight[p] += lp[p]; }\n");
  o.f("      if (a.site_scores && OWN_) { float ss_[PPT]; PLOOP ss_[p] = lp[p]; VecStore<PPT>::st(a.site_scores + (int64_t)%d * K + i0, ss_); }\n", ri.score_row);
  o.f("      PLOOP asm volatile(\"\" : \"+v\"(score[p]), \"+v\"(weight[p]));\n      __builtin_amdgcn_sched_barrier(0);\n    }\n");
}

// may site j + 1 be scored in the element loop of site j?  (see Plan::fused_into)
bool fusable_observation(const Plan& pl, int j) {
  const gjx_program* prog = pl.prog;
  if (getenv("GJX_GEN_NO_FUSE") || pl.pf || pl.mfma || pl.wide || j + 1 >= prog->n_sites) return false;
  const gjx_site& a = prog->sites[j];
  const gjx_site& b = prog->sites[j + 1];
  const RollInfo& ra = pl.info[j];
  const RollInfo& rb = pl.info[j + 1];
  if (ra.plate || rb.plate || ra.load_here || rb.load_here || ra.d_row || rb.d_obs || rb.d_off[0] || rb.d_off[1] || rb.mem_slot[0] >= 0 || rb.mem_slot[1] >= 0) return false;
  if (a.scan != b.scan) return false;
  if (is_categorical(a.kind) || a.kind == GJX_DIRICHLET || a.slot < 0 || a.dim > kMaxExpandDim || a.mode != GJX_MODE_SAMPLE || (a.flags & GJX_SITE_PROPOSAL)) return false;
  if (!is_normal(b.kind) || b.mode != GJX_MODE_OBS_TAB || b.dim != a.dim || (b.flags & GJX_SITE_PROPOSAL)) return false;
  const gjx_param& m = b.p[0];
  const gjx_param& sc = b.p[1];
  if (m.op != GJX_P_VALUE || m.xf != GJX_XF_NONE || m.slot != a.slot || m.len != (a.dim == 1 ? 1 : a.dim)) return false;
  return sc.op == GJX_P_CONST && sc.xf == GJX_XF_NONE;
}

void emit_site(Emit& o, Plan& pl, int j) {
  const gjx_program* prog = pl.prog;
  const gjx_site& s = prog->sites[j];
  const RollInfo& ri = pl.info[j];
  g_loop_var = ri.plate ? "i_" : "(t_ - 1)";
  if (pl.fused_into.size() != (size_t)prog->n_sites) pl.fused_into.assign(prog->n_sites, -1);
  const bool fuse_next = fusable_observation(pl, j);
  const bool fused_here = pl.fused_into[j] >= 0;       // this site's elements were scored by the site in front of it
  if (s.mode == GJX_MODE_INPUT) {
    // the carry of a Scan step / an argument: rows that are already there, or the rows of the ancestor the resampling step
    // picked for this slot (gjx_run_program_ex: the particle gather fused into the read side); no draw, no score
    o.f("    { // ---- site %d: INPUT, %d rows, slot %d\n", j, s.dim, s.slot);
    for (int d = 0; d < s.dim; ++d) {
      if (pl.pf) o.f("      PLOOP v[%d][p] = LDIN(a.in_rows + (int64_t)%d * a.in_stride + src_[p]);\n", s.slot + d, s.obs_off + d);   // (always through the ancestor)
      else o.f("      PLOOP v[%d][p] = a.in_rows ? LDIN(a.in_rows + (int64_t)%d * a.in_stride + src_[p]) : a.choices[(int64_t)%d * K + i0 + p];\n", s.slot + d,
               s.obs_off + d, ri.row + d);
    }
    if (pl.pf) {       // (the filter flavour stores its inputs behind the rejuvenation move, if any: generate_pf) — a CARRIED input here
      if (s.flags & GJX_SITE_CARRIED) for (int d = 0; d < s.dim; ++d) o.f("      VSTORE(a.choices + (int64_t)%d * K + i0, v[%d]);\n", ri.row + d, s.slot + d);
    } else {
      o.f("      if (a.in_rows && (a.store_inputs || %d)) {\n", (s.flags & GJX_SITE_CARRIED) ? 1 : 0);
      for (int d = 0; d < s.dim; ++d) o.f("        VSTORE1(a.choices + (int64_t)%d * K + i0, v[%d]);\n", ri.row + d, s.slot + d);
      o.f("      }\n");
    }
    o.f("      if (a.site_scores && OWN_) { float ss_[PPT]; PLOOP ss_[p] = 0.0f; VecStore<PPT>::st(a.site_scores + (int64_t)%d * K + i0, ss_); }\n    }\n", ri.score_row);
    return;
  }
  if (pl.mfma && mfma_site_ok(s, ri)) { emit_mfma_site(o, pl, j); return; }
  // (GJX_MODE_OBS_PROPOSED: scored at the value a proposal site of this run left in the slot's registers — an OBS_SLOT without a load)
  const int mode = s.mode == GJX_MODE_OBS_PROPOSED ? GJX_MODE_OBS_SLOT : s.mode, kind = s.kind;
  const bool masked = mode == GJX_MODE_OBS_MASK;
  const bool draws = mode == GJX_MODE_SAMPLE || masked;
  const bool is_proposal = (s.flags & GJX_SITE_PROPOSAL) != 0;    // log w -= log q; no part of the score
  // the filter flavour keeps weights only (no score, no per-site scores): the log-density of a SAMPLED model site is never looked at
  const bool need_lp = !(pl.pf && mode == GJX_MODE_SAMPLE && !is_proposal);
  const int np = n_params(kind);
  const int hoist = (!pl.hoist_at.empty() && pl.hoist_at[j] >= 0) ? pl.hoist_at[j] : -1;
  if (pl.stream[j].opens && hoist < 0) {
    const SiteStream& hs = pl.stream[j];
    o.f("    BitStream<RNG> rs%d[PPT];\n    PLOOP rs%d[p].open_hi(%s, GHI_, (uint32_t)gidx[p], %du);\n", hs.run, hs.run,
        hs.key_var.empty() ? "a.key" : hs.key_var.c_str(), hs.site_no);
  }
  int fz_rcp = -1, fz_ys = -1;       // companions of the observation scored in this site's loop: 1 / sigma and y / sigma
  if (fuse_next) {
    const gjx_site& b = prog->sites[j + 1];
    pl.fused_into[j + 1] = j;
    fz_rcp = pl.find(1, b.p[1].off, table_range(b.p[1]));
    fz_ys = pl.find(4, b.p[1].off, b.dim, b.p[1].len, b.obs_off);
    o.f("    float q2f%d[PPT];   // squared z-scores of site %d, taken in the element loop of site %d\n    PLOOP q2f%d[p] = 0.0f;\n", j + 1, j + 1, j, j + 1);
  }
  o.f("    { // ---- site %d: kind %d, dim %d, mode %d, slot %d%s\n", j, kind, is_categorical(kind) ? s.ncat : s.dim, mode, s.slot, fused_here ? " (scored in the loop of the site in front)" : "");
  o.f("      float lp[PPT];\n      PLOOP lp[p] = 0.0f;\n");
  // stream key and site number (gjx.h "Scan steps"): chained step keys are wave-uniform locals emitted on first use
  const SiteStream& ss = pl.stream[j];
  // element at which this (site, instance) starts drawing (gjx.h "Plates": the elements of a vector site of n * dim elements)
  std::string ebase = "0u";
  if (ri.plate && prog->rng_mode == GJX_RNG_FLAT) {
    const int per = is_categorical(kind) ? 1 : s.dim * gjx::kind_draws(kind);
    ebase = "(uint32_t)(i_ * " + std::to_string(per) + ")";
  }
  if (draws && ri.plate) {
    // FLAT: the body site's stream lives outside the instance loop (its hash blocks are shared by consecutive instances);
    // JAX32: site key = fold_in(instance key, position in the body) (static.py:349-352 inside the vmapped kernel)
    if (prog->rng_mode == GJX_RNG_FLAT) o.f("      BitStream<RNG> (&bs)[PPT] = ps%d;\n", j);
    else o.f("      BitStream<RNG> bs[PPT];\n      PLOOP bs[p].open_site_key(fold_in(ik_[p], %du));\n", ri.plate_l + 1);
  } else if (draws && hoist >= 0) {
    // (filter flavour: the site's standard normals were drawn ahead, dr_->nz)
  } else if (draws && ss.run >= 0) {
    // member of a scalar-normal run: the run's stream lives OUTSIDE the site's block (declared by the head, in the
    // enclosing scope), members 2k and 2k+1 share one Box-Muller evaluation through its pair cache
    o.f("      BitStream<RNG> (&bs)[PPT] = rs%d;\n", ss.run);
  } else if (draws) {
    if (prog->rng_mode != GJX_RNG_FLAT || ss.key_var.empty()) o.f("      BitStream<RNG> bs[PPT];\n      PLOOP bs[p].open_hi(a.key, GHI_, (uint32_t)gidx[p], %du);\n", ss.site_no);
    else o.f("      BitStream<RNG> bs[PPT];\n      PLOOP bs[p].open_hi(%s, GHI_, (uint32_t)gidx[p], %du);\n", ss.key_var.c_str(), ss.site_no);
  }
  if (ri.load_here) {   // per-particle constraint rows / mask flags whose row moves with t_ or differs from the register number
    const int nrow_ = is_categorical(kind) ? 1 : s.dim;
    if (mode == GJX_MODE_OBS_SLOT || masked)
      for (int d = 0; d < nrow_; ++d) o.f("      PLOOP v[%d][p] = a.choices[(int64_t)%s * K + i0 + p];\n", s.slot + d, toff(ri.row + d, ri.d_row).c_str());
    if (masked) o.f("      PLOOP v[%d][p] = a.choices[(int64_t)%s * K + i0 + p];\n", s.obs_off, toff(ri.flag_row, ri.d_flag_row).c_str());
  }
  if (masked) o.f("      bool given[PPT];\n      PLOOP given[p] = v[%d][p] != 0.0f;\n", s.obs_off);
  emit_gather_index(o, s, j, np, ri);
  if (is_categorical(kind)) {
    const gjx_param& q = s.p[0];
    const int n = s.ncat;
    const bool probs = kind == GJX_CATEGORICAL_PROBS;
    const bool fast = q.op == GJX_P_CONST && q.xf == GJX_XF_NONE && !probs && q.len == n && ri.d_off[0] == 0;
    // L(c): logit of category c for particle p
    std::string L = xf_wrap(q.xf, param_expr(q, "c_", j, 0, ri));
    if (probs) L = "safe_log(" + L + ")";
    if (fast) {
      const int at = pl.find(2, q.off, n);
      o.f("      PLOOP {\n        float val;\n");
      if (draws) {
        o.f("        if (RNG == GJX_RNG_FLAT) {\n");
        o.f("          const float target = bits_to_unit(bs[p].get(%s)) * COMP(%d);\n          unsigned neg = 0u;\n", ebase.c_str(), at + n - 1);
        o.f("          _Pragma(\"unroll\") for (int c_ = 0; c_ < %d; ++c_) neg += __float_as_uint(target - COMP(%d + c_)) >> 31;\n", n - 1, at);
        o.f("          val = (float)(%d - (int)neg);\n        } else {\n", n - 1);
        o.f("          int best = 0; float bestv = -INFINITY;\n");
        o.f("          for (int c_ = 0; c_ < %d; ++c_) { const float g_ = %s + gumbel_from_bits(bs[p].get((uint32_t)c_)); if (g_ > bestv) { bestv = g_; best = c_; } }\n", n, L.c_str());
        o.f("          val = (float)best;\n        }\n");
      }
      if (mode == GJX_MODE_OBS_TAB) o.f("        val = TAB(%s);\n", toff(s.obs_off, ri.d_obs).c_str());
      if (mode == GJX_MODE_OBS_SLOT) o.f("        val = v[%d][p];\n", s.slot);
      if (masked) o.f("        if (given[p]) val = v[%d][p];\n", s.slot);
      o.f("        const int k_ = (int)val;\n");
      o.f("        lp[p] = (k_ < 0 || k_ >= %d) ? -INFINITY : TAB(%d + k_) - COMP(%d);\n", n, q.off, at + n);
      if (s.slot >= 0 && mode != GJX_MODE_OBS_SLOT) o.f("        v[%d][p] = val;\n", s.slot);
      o.f("      }\n");
    } else {
      o.f("      PLOOP {\n        float mx = -INFINITY;\n");
      o.f("        for (int c_ = 0; c_ < %d; ++c_) mx = fmaxf(mx, %s);\n", n, L.c_str());
      o.f("        float se = 0.0f;\n        for (int c_ = 0; c_ < %d; ++c_) se += fast_exp(%s - mx);\n", n, L.c_str());
      o.f("        const float lse_ = mx + fast_log(se);\n        float val;\n");
      if (draws) {
        o.f("        if (RNG == GJX_RNG_FLAT) {\n          const float target = bits_to_unit(bs[p].get(%s)) * se; float run = 0.0f; int zc = %d; bool found = false;\n", ebase.c_str(), n - 1);
        o.f("          for (int c_ = 0; c_ < %d; ++c_) { run += fast_exp(%s - mx); if (!found && run > target) { zc = c_; found = true; } }\n", n, L.c_str());
        o.f("          val = (float)zc;\n        } else {\n          int best = 0; float bestv = -INFINITY;\n");
        o.f("          for (int c_ = 0; c_ < %d; ++c_) { const float g_ = %s + gumbel_from_bits(bs[p].get((uint32_t)c_)); if (g_ > bestv) { bestv = g_; best = c_; } }\n", n, L.c_str());
        o.f("          val = (float)best;\n        }\n");
      }
      if (mode == GJX_MODE_OBS_TAB) o.f("        val = TAB(%s);\n", toff(s.obs_off, ri.d_obs).c_str());
      if (mode == GJX_MODE_OBS_SLOT) o.f("        val = v[%d][p];\n", s.slot);
      if (masked) o.f("        if (given[p]) val = v[%d][p];\n", s.slot);
      o.f("        const int k_ = (int)val;\n");
      o.f("        if (k_ < 0 || k_ >= %d) lp[p] = -INFINITY; else { const int c_ = k_; lp[p] = %s - lse_; }\n", n, L.c_str());
      if (s.slot >= 0 && mode != GJX_MODE_OBS_SLOT) o.f("        v[%d][p] = val;\n", s.slot);
      o.f("      }\n");
    }
  } else {
    const int dim = s.dim;
    const bool expand = dim <= kMaxExpandDim;
    const int nd = gjx::kind_draws(kind);
    // the scale of a normal that comes straight from the table has its log and reciprocal in the prologue's companions
    const bool norm = is_normal(kind);
    const gjx_param& qb = s.p[1];
    const bool comp_scale = norm && table_param(qb) && ri.d_off[1] == 0;   // companions are built once: time-invariant scales only
    int at_rcp = -1, at_sum = -1;
    if (comp_scale) {
      if (mode != GJX_MODE_SAMPLE) at_rcp = pl.find(1, qb.off, table_range(qb));
      at_sum = pl.find(3, qb.off, qb.op == GJX_P_CONST ? 1 : qb.n, qb.len, dim);
    }
    // observed in the table under a constant scale: z = y/sigma - mean/sigma with y/sigma precomputed
    const int at_ys = (comp_scale && mode == GJX_MODE_OBS_TAB && qb.op == GJX_P_CONST && ri.d_obs == 0) ? pl.find(4, qb.off, dim, qb.len, s.obs_off) : -1;
    // diagonal normals accumulate the squared z-scores; the normaliser is one constant per (gathered) row of the
    // scale table, or summed alongside when the scale is not a plain table entry
    if (norm && need_lp) o.f("      float q2[PPT], ls[PPT];\n      PLOOP { q2[p] = 0.0f; ls[p] = 0.0f; }\n");
    const bool store_early = expand && s.slot >= 0 && mode != GJX_MODE_OBS_SLOT && !ri.plate && !pl.pf && !pl.wide && !getenv("GJX_GEN_NO_EARLY_STORE");
    auto element = [&](const std::string& dx, const char* ind) {
      o.f("%sPLOOP {\n", ind);
      std::string in2 = std::string(ind) + "  ";
      std::string pe[4];
      for (int k = 0; k < np; ++k) {
        emit_param_pre(o, s.p[k], dx, j, k, in2.c_str(), ri);
        pe[k] = xf_wrap(s.p[k].xf, param_expr(s.p[k], dx, j, k, ri));
      }
      for (int k = np; k < 4; ++k) pe[k] = "0.0f";
      const std::string vslot = "v[" + std::to_string(s.slot) + " + (" + dx + ")][p]";
      o.f("%sconst float pa = %s;\n", in2.c_str(), pe[0].c_str());
      if (norm) {
        std::string logb, rcpb;
        if (comp_scale) {
          const std::string idx = "(" + tab_index(qb, dx, j, 1) + ") - " + std::to_string(qb.off);   // d_off[1] == 0 here
          if (at_rcp >= 0) rcpb = "COMP(" + std::to_string(at_rcp) + " + " + idx + ")";
        }
        o.f("%sconst float pb = %s;\n", in2.c_str(), pe[1].c_str());
        if (logb.empty()) logb = "fast_log(pb)";
        if (rcpb.empty()) rcpb = "fast_rcp(pb)";
        o.f("%sfloat val;\n", in2.c_str());
        if (mode == GJX_MODE_SAMPLE && hoist >= 0) {
          // the site's standard normal from the words taken ahead (Plan::hoist_at: the base of the site's / the run's words)
          const int e_ = (pl.stream[j].run >= 0 ? (int)pl.stream[j].elem : 0) + literal_index(dx);
          if (prog->rng_mode != GJX_RNG_JAX32)
            o.f("%sfloat hn0_, hn1_;\n%sbox_muller(__float_as_uint(dr_->nz[%d]), __float_as_uint(dr_->nz[%d]), hn0_, hn1_);\n%sconst float n_ = %s;\n", in2.c_str(), in2.c_str(),
                hoist + (e_ & ~1), hoist + (e_ | 1), in2.c_str(), (e_ & 1) ? "hn1_" : "hn0_");
          else
            o.f("%sconst float n_ = normal_from_bits(__float_as_uint(dr_->nz[%d]));\n", in2.c_str(), hoist + e_);
          o.f("%sval = fmaf(pb, n_, pa);\n", in2.c_str());
          if (need_lp) o.f("%sq2[p] = fmaf(n_, n_, q2[p]);\n", in2.c_str());
        } else if (mode == GJX_MODE_SAMPLE) {
          o.f("%sconst float n_ = stream_normal<RNG>(bs[p], %s + (uint32_t)(%u + (%s)));\n", in2.c_str(), ebase.c_str(), pl.stream[j].run >= 0 ? pl.stream[j].elem : 0u, dx.c_str());
          o.f("%sval = fmaf(pb, n_, pa);\n", in2.c_str());
          if (need_lp) o.f("%sq2[p] = fmaf(n_, n_, q2[p]);\n", in2.c_str());
        } else {
          if (masked) o.f("%sval = given[p] ? %s : fmaf(pb, stream_normal<RNG>(bs[p], %s + (uint32_t)(%s)), pa);\n", in2.c_str(), vslot.c_str(), ebase.c_str(), dx.c_str());
          else if (mode == GJX_MODE_OBS_TAB) o.f("%sval = TAB(%s + (%s));\n", in2.c_str(), toff(s.obs_off, ri.d_obs).c_str(), dx.c_str());
          else o.f("%sval = %s;\n", in2.c_str(), vslot.c_str());
          if (at_ys >= 0) o.f("%s{ const float z_ = fmaf(-%s, pa, COMP(%d + (%s))); q2[p] = fmaf(z_, z_, q2[p]); }\n", in2.c_str(), rcpb.c_str(), at_ys, dx.c_str());
          else o.f("%s{ const float z_ = (val - pa) * %s; q2[p] = fmaf(z_, z_, q2[p]); }\n", in2.c_str(), rcpb.c_str());
        }
        if (at_sum < 0 && need_lp) o.f("%sls[p] += kHalfLog2Pi + %s;\n", in2.c_str(), logb.c_str());
      } else {
        o.f("%sconst float pb = %s, pc = %s, pd = %s;\n%sfloat val;\n", in2.c_str(), pe[1].c_str(), pe[2].c_str(), pe[3].c_str(), in2.c_str());
        const std::string smp = "elem_sample<RNG>(" + std::to_string(kind) + ", bs[p], " + ebase + " + (uint32_t)((" + dx + ") * " + std::to_string(nd) + "), pa, pb, pc, pd)";
        if (mode == GJX_MODE_SAMPLE) o.f("%sval = %s;\n", in2.c_str(), smp.c_str());
        else if (masked) o.f("%s{ const float smp_ = %s; val = given[p] ? %s : smp_; }\n", in2.c_str(), smp.c_str(), vslot.c_str());
        else if (mode == GJX_MODE_OBS_TAB) o.f("%sval = TAB(%s + (%s));\n", in2.c_str(), toff(s.obs_off, ri.d_obs).c_str(), dx.c_str());
        else o.f("%sval = %s;\n", in2.c_str(), vslot.c_str());
        if (need_lp) o.f("%slp[p] += elem_logpdf(%d, val, pa, pb, pc, pd);\n", in2.c_str(), kind);
      }
      if (s.slot >= 0 && mode != GJX_MODE_OBS_SLOT) o.f("%s%s = val;\n", in2.c_str(), vslot.c_str());
      if (fuse_next) {
        const gjx_param& sb = prog->sites[j + 1].p[1];
        const std::string ib = sb.len == 1 ? "0" : "(" + dx + ") % " + std::to_string(sb.len);
        o.f("%s{ const float z_ = fmaf(-COMP(%d + %s), val, COMP(%d + (%s))); q2f%d[p] = fmaf(z_, z_, q2f%d[p]); }\n", in2.c_str(), fz_rcp, ib.c_str(), fz_ys, dx.c_str(), j + 1, j + 1);
      }
      o.f("%s}\n", ind);
      // the row leaves as soon as it exists (registers are not held to the end of the site)
      if (store_early && pl.seq_rows && literal_index(dx) >= 0) o.f("%s", pl.row_store("VSTORE1", ind, ri.row + literal_index(dx), "v[" + std::to_string(s.slot + literal_index(dx)) + "]").c_str());
      else if (store_early) o.f("%sVSTORE1(a.choices + (int64_t)(%s + (%s)) * K + i0, v[%d + (%s)]);\n", ind, toff(ri.row, ri.d_row).c_str(), dx.c_str(), s.slot, dx.c_str());
    };
    if (fused_here) {
      o.f("      PLOOP q2[p] = q2f%d[p];\n", j);
    } else if (expand) {
      for (int d = 0; d < dim; ++d) {
        element(std::to_string(d), "      ");
        if ((d & 1) == 1 && d + 1 < dim && need_lp) o.f("      PLOOP asm volatile(\"\" : \"+v\"(%s[p]));\n      __builtin_amdgcn_sched_barrier(0);\n", norm ? "q2" : "lp");
      }
    } else {
      o.f("      for (int d_ = 0; d_ < %d; ++d_) {\n", dim);
      element("d_", "        ");
      o.f("      }\n");
    }
    if (norm && need_lp) {
      if (at_sum < 0) o.f("      PLOOP lp[p] = fmaf(-0.5f, q2[p], -ls[p]);\n");
      else if (qb.op == GJX_P_CONST) o.f("      PLOOP lp[p] = fmaf(-0.5f, q2[p], -COMP(%d));\n", at_sum);
      else o.f("      PLOOP lp[p] = fmaf(-0.5f, q2[p], -COMP(%d + gi_%d_1[p]));\n", at_sum, j);
    }
  }
  // bookkeeping: score, weight, per-site scores, store the site's rows
  if (is_proposal) o.f("      PLOOP weight[p] -= lp[p];\n");
  else if (pl.pf) { if (mode != GJX_MODE_SAMPLE) o.f("      PLOOP weight[p] += lp[p];\n"); }
  else o.f("      PLOOP { score[p] += lp[p];%s }\n", masked ? " if (given[p]) weight[p] += lp[p];" : (mode != GJX_MODE_SAMPLE ? " weight[p] += lp[p];" : ""));
  if (ri.plate) o.f("      PLOOP pacc%d[p] += lp[p];\n", j);      // a plate's body site: the sum over its instances
  else o.f("      if (a.site_scores && OWN_) { float ss_[PPT]; PLOOP ss_[p] = lp[p]; VecStore<PPT>::st(a.site_scores + (int64_t)%s * K + i0, ss_); }\n",
           toff(ri.score_row, ri.d_score_row).c_str());
  const bool stored_early = !is_categorical(kind) && s.dim <= kMaxExpandDim && s.slot >= 0 && mode != GJX_MODE_OBS_SLOT && !ri.plate && !pl.pf && !pl.wide && !getenv("GJX_GEN_NO_EARLY_STORE");
  if (s.slot >= 0 && mode != GJX_MODE_OBS_SLOT && !stored_early) {
    const int nrow = is_categorical(kind) ? 1 : s.dim;
    // (a plate's instance is stored by the wave that produced it; anything else by the block's first wave: VSTORE1)
    for (int d = 0; d < nrow; ++d) {
      if (pl.seq_rows && !ri.plate) o.f("%s", pl.row_store("VSTORE1", "      ", ri.row + d, "v[" + std::to_string(s.slot + d) + "]").c_str());
      else o.f("      %s(a.choices + (int64_t)%s * K + i0, v[%d]);\n", ri.plate ? "VSTORE" : "VSTORE1", toff(ri.row + d, ri.d_row).c_str(), s.slot + d);
    }
  }
  if (pl.pf) o.f("      PLOOP asm volatile(\"\" : \"+v\"(weight[p]));\n      __builtin_amdgcn_sched_barrier(0);\n    }\n");
  else o.f("      PLOOP asm volatile(\"\" : \"+v\"(score[p]), \"+v\"(weight[p]));\n      __builtin_amdgcn_sched_barrier(0);\n    }\n");
}

bool want_roll() { const char* e = getenv("GJX_GEN_ROLL"); return e && atoi(e) != 0; }

// everything the emitter derives from a program before it writes a line: the emitted site list (plates / a rolled Scan remapped to
// registers), stream keys, site numbers, scalar-normal runs
struct GenCtx {
  Roll roll;
  PlateXf px;
  gjx_program eprog;
  Plan pl;
};

void plan_program(const gjx_program* prog_in, int ppt_code, GenCtx& g, bool allow_roll = true) {
  const int ppt = ppt_code & 255;
  const bool mfma = ((ppt_code >> 8) & 1) != 0 && ppt == 1;
  Roll& roll = g.roll;
  Plan& pl = g.pl;
  g_expr_prog = prog_in;
  // a Scan too long to unroll (or GJX_GEN_ROLL=1) is emitted as a loop over its steps when its descriptors are periodic
  g.px = plate_program(prog_in);
  const PlateXf& px = g.px;
  if (allow_roll && !px.any && (want_roll() || !supported_sites(prog_in->sites, prog_in->n_sites, prog_in->n_slots, prog_in))) roll = detect_roll(prog_in);
  g.eprog = *prog_in;
  gjx_program& eprog = g.eprog;
  if (roll.ok) { eprog.sites = roll.sites.data(); eprog.n_sites = (int)roll.sites.size(); eprog.n_slots = roll.n_regs; }
  if (px.any) { eprog.sites = px.sites.data(); eprog.n_sites = (int)px.sites.size(); eprog.n_slots = px.n_regs; }
  const gjx_program* prog = &eprog;
  pl.prog = prog;
  pl.ppt = ppt;
  pl.mfma = mfma;
  pl.wide = (ppt_code & 512) != 0 && px.any && !mfma;
  pl.lpp = pl.wide ? ((ppt_code & 2048) ? 16 : ((ppt_code & 1024) ? 4 : 1)) : 1;
  pl.seq_rows = !px.any && !roll.ok && !mfma && !getenv("GJX_GEN_NO_SEQ_ROWS");      // (generate_pf clears it: its rows move with the step)
  pl.tab_lds = mfma || (prog->n_tab <= kMaxLdsTab && !getenv("GJX_GEN_TAB_GLOBAL"));   // (the variable: profiling variant, part of the cache key)
  if (roll.ok) {
    pl.info = roll.info;
    unsigned plain = 0;
    for (int j = 0; j < roll.i0; ++j) pl.stream.push_back({"", ++plain});
    for (int l = 0; l < roll.m; ++l) pl.stream.push_back({"sk0", (unsigned)(l + 1)});
    for (int l = 0; l < roll.m; ++l) pl.stream.push_back({"skt", (unsigned)(l + 1)});
    for (int j = 0; j < roll.n_post; ++j) pl.stream.push_back({"", ++plain});      // behind the Scan: the run key again, numbering continues
    int n_runs = 0;   // runs never cross the segments (the Scan tag changes at each boundary)
    assign_runs(prog, pl.stream, 0, roll.i0, n_runs);
    assign_runs(prog, pl.stream, roll.i0, roll.i0 + roll.m, n_runs);
    assign_runs(prog, pl.stream, roll.i0 + roll.m, roll.i0 + 2 * roll.m, n_runs);
    assign_runs(prog, pl.stream, roll.i0 + 2 * roll.m, roll.i0 + 2 * roll.m + roll.n_post, n_runs);
    char b[256];
    snprintf(b, sizeof(b), "  const key2 sk0 = fold_in(fold_in(a.key, %uu), 0u);\n", 0x80000000u | roll.scan_id);
    pl.key_decls = b;
  } else {   // stream keys and site numbers, as SiteStreamWalk (gjx_device.h) walks them
    int tag = 0, nkey = 0;
    unsigned local = 0, plain = 0, jn = 0;
    std::string cur;
    for (int j = 0; j < prog->n_sites; ++j) {
      RollInfo ri; ri.row = prog->sites[j].slot; ri.score_row = j;
      pl.info.push_back(px.any ? px.info[j] : ri);
      const int sc = prog->sites[j].scan;
      if (prog->sites[j].mode == GJX_MODE_INPUT) { pl.stream.push_back({"", 0u}); continue; }      // (takes no site number)
      if (prog->rng_mode != GJX_RNG_FLAT) {   // (a plate is ONE traced site of its caller: its body sites share the first one's number)
        const bool inner = prog->sites[j].plate != 0 && j > 0 && prog->sites[j - 1].plate == prog->sites[j].plate;
        pl.stream.push_back({"", inner ? jn : ++jn});
        continue;
      }
      if (sc == 0) { tag = 0; pl.stream.push_back({"", ++plain}); continue; }
      if (sc != tag) {
        const unsigned id = (unsigned)sc >> 20;
        const int step = (int)((unsigned)sc & 0xFFFFFu) - 1;
        const bool follows = tag != 0 && ((unsigned)tag >> 20) == id && (int)((unsigned)tag & 0xFFFFFu) - 1 == step - 1;
        char b[256];
        const std::string nm = "sk" + std::to_string(nkey++);
        if (follows) snprintf(b, sizeof(b), "  const key2 %s = fold_in(%s, %du);\n", nm.c_str(), cur.c_str(), (unsigned)step);
        else snprintf(b, sizeof(b), "  key2 %s = fold_in(a.key, %uu); for (int t_ = 0; t_ <= %d; ++t_) %s = fold_in(%s, (uint32_t)t_);\n", nm.c_str(),
                      0x80000000u | id, step, nm.c_str(), nm.c_str());
        pl.key_decls += b;
        cur = nm;
        tag = sc;
        local = 0;
      }
      pl.stream.push_back({cur, ++local});
    }
    int n_runs = 0;
    assign_runs(prog, pl.stream, 0, prog->n_sites, n_runs);
  }
}

// the sites of the planned program, in order (a rolled Scan as a loop, a plate as an instance loop)
std::string emit_body(GenCtx& g, int j_lo = 0, int j_hi = -1) {
  Roll& roll = g.roll;
  Plan& pl = g.pl;
  const gjx_program* prog = pl.prog;
  if (j_hi < 0) j_hi = prog->n_sites;
  auto skipped = [&](int j) { return !pl.skip.empty() && pl.skip[j]; };
  Emit body;
  if (roll.ok) {
    auto carry = [&]() {   // the step just produced becomes the previous step
      body.f("      PLOOP { ");
      for (int l = 0; l < roll.S; ++l) body.f("v[%d][p] = v[%d][p]; ", roll.n_pre + l, roll.n_pre + roll.S + l);
      body.f("}\n");
    };
    for (int j = 0; j < roll.i0 + roll.m; ++j) emit_site(body, pl, j);
    carry();
    body.f("    key2 skt = sk0;\n    for (int t_ = 1; t_ < %d; ++t_) {   // steps 1 .. T-1: one emitted body, rows and table offsets advance with t_\n"
           "      skt = fold_in(skt, (uint32_t)t_);\n", roll.T);
    for (int j = roll.i0 + roll.m; j < roll.i0 + 2 * roll.m; ++j) emit_site(body, pl, j);
    carry();
    body.f("    }\n");
    for (int j = roll.i0 + 2 * roll.m; j < roll.i0 + 2 * roll.m + roll.n_post; ++j) emit_site(body, pl, j);   // the last step sits in the "previous" registers
  } else {
    for (int j = j_lo; j < j_hi;) {
      if (skipped(j)) { ++j; continue; }
      if (!pl.info[j].plate) { emit_site(body, pl, j++); continue; }
      // ---- a plate: prologue, ONE instance loop over the body, epilogue (gjx.h "Plates"; vmap.py:180-218)
      int m = 1;
      while (j + m < prog->n_sites && pl.info[j + m].plate && pl.info[j + m].plate_j0 == pl.info[j].plate_j0) ++m;
      const bool flat = prog->rng_mode == GJX_RNG_FLAT;
      body.f("    { // ---- plate of %d instances x %d sites\n", pl.info[j].plate_n, m);
      for (int l = 0; l < m; ++l) {
        const gjx_site& sl = prog->sites[j + l];
        body.f("    float pacc%d[PPT];\n    PLOOP pacc%d[p] = 0.0f;\n", j + l, j + l);
        const bool draws = sl.mode == GJX_MODE_SAMPLE || sl.mode == GJX_MODE_OBS_MASK;
        if (flat && draws) {
          const SiteStream& ss = pl.stream[j + l];
          body.f("    BitStream<RNG> ps%d[PPT];\n    PLOOP ps%d[p].open_hi(%s, GHI_, (uint32_t)gidx[p], %du);\n", j + l, j + l, ss.key_var.empty() ? "a.key" : ss.key_var.c_str(), ss.site_no);
        }
      }
      // JAX32: the Vmap call is one traced site of its caller: plate key = fold_in(particle key, J); instance key = split(plate key, n)[i]
      if (!flat) body.f("    key2 pk_[PPT];\n    PLOOP pk_[p] = fold_in(fold_in64(a.key, gidx[p]), %uu);\n", pl.stream[j].site_no);
      if (pl.wide) {
        // the instances in 16 contiguous chunks, one per wave of the block (consecutive instances share hash blocks of the FLAT
        // streams); the plate's contribution to score / weight is summed apart and joined across the waves below
        body.f("    const int pc_ = (%d + NCH_ - 1) / NCH_, plo_ = pw_ * pc_, phi_ = plo_ + pc_ < %d ? plo_ + pc_ : %d;\n"
               "    float psc_[PPT], pwt_[PPT];\n    PLOOP { psc_[p] = score[p]; pwt_[p] = weight[p]; score[p] = 0.0f; weight[p] = 0.0f; }\n"
               "    _Pragma(\"nounroll\") for (int i_ = plo_; i_ < phi_; ++i_) {\n", pl.info[j].plate_n, pl.info[j].plate_n, pl.info[j].plate_n);
      } else
      body.f("    _Pragma(\"nounroll\") for (int i_ = 0; i_ < %d; ++i_) {\n", pl.info[j].plate_n);
      if (!flat) body.f("    key2 ik_[PPT];\n    PLOOP ik_[p] = fold_in(pk_[p], (uint32_t)i_);\n");
      for (int l = 0; l < m; ++l) emit_site(body, pl, j + l);
      body.f("    }\n");
      if (pl.wide) {
        // join: every wave leaves its partial in LDS, every lane adds the 16 partials of its particles in wave order
        body.f("    PRED_(score, psc_);\n    PRED_(weight, pwt_);\n    if (a.site_scores) {\n      float pz_[PPT];\n      PLOOP pz_[p] = 0.0f;\n");
        for (int l = 0; l < m; ++l) body.f("      PRED_(pacc%d, pz_);\n", j + l);
        body.f("    }\n");
      }
      body.f("    if (a.site_scores && OWN_) {\n");
      for (int l = 0; l < m; ++l) body.f("      VecStore<PPT>::st(a.site_scores + (int64_t)%d * K + i0, pacc%d);\n", j + l, j + l);
      body.f("    }\n    }\n");
      j += m;
    }
  }
  return body.s;
}

// derived constants (Companion) from the float table: one pass per entry, spread over the block.  The emitted code reads the table
// through TSRC(i) (the LDS copy, or the step's table in global memory) and strides by BT_ threads
void emit_companions(Emit& o, Plan& pl) {
  for (auto& c : pl.comps) {
    if (c.kind == 0) o.f("  for (int t = threadIdx.x; t < %d; t += BT_) COMP(%d + t) = fast_log(TSRC(%d + t));\n", c.n, c.at, c.off);
    else if (c.kind == 1) o.f("  for (int t = threadIdx.x; t < %d; t += BT_) COMP(%d + t) = fast_rcp(TSRC(%d + t));\n", c.n, c.at, c.off);
    else if (c.kind == 4)
      o.f("  for (int t = threadIdx.x; t < %d; t += BT_) COMP(%d + t) = TSRC(%d + t) * fast_rcp(TSRC(%d + t %% %d));\n", c.n, c.at, c.dim, c.off, c.len);
    else if (c.kind == 3) {
      int p2 = 1;
      while (p2 < c.dim) p2 <<= 1;
      if (p2 <= 64) {   // one (row, element) per lane, rows are aligned groups of p2 lanes: sum by xor-shuffles
        o.f("  for (int t0 = 0; t0 < %d; t0 += BT_) {\n    const int t = t0 + threadIdx.x, r_ = t / %d, e_ = t %% %d;\n"
            "    float s_ = (r_ < %d && e_ < %d) ? kHalfLog2Pi + fast_log(TSRC(%d + r_ * %d + e_ %% %d)) : 0.0f;\n",
            c.n * p2, p2, p2, c.n, c.dim, c.off, c.len, c.len);
        for (int o2 = p2 >> 1; o2 >= 1; o2 >>= 1) o.f("    s_ += __shfl_xor(s_, %d, 64);\n", o2);
        o.f("    if (r_ < %d && e_ == 0) COMP(%d + r_) = s_;\n  }\n", c.n, c.at);
      } else {
        o.f("  for (int t = threadIdx.x; t < %d; t += BT_) { float s_ = 0.0f; for (int e_ = 0; e_ < %d; ++e_) s_ += kHalfLog2Pi + fast_log(TSRC(%d + t * %d + e_ %% %d)); COMP(%d + t) = s_; }\n",
            c.n, c.dim, c.off, c.len, c.len, c.at);
      }
    }
    else {
      o.f("  if (threadIdx.x == 0) {   // running CDF (float32, category order) and log-sum-exp of constant logits\n"
          "    float mx = -INFINITY;\n    for (int c_ = 0; c_ < %d; ++c_) mx = fmaxf(mx, TSRC(%d + c_));\n"
          "    float run = 0.0f;\n    for (int c_ = 0; c_ < %d; ++c_) { run += fast_exp(TSRC(%d + c_) - mx); COMP(%d + c_) = run; }\n"
          "    COMP(%d) = mx + fast_log(run);\n  }\n", c.n, c.off, c.n, c.off, c.at, c.at + c.n);
    }
  }
}

std::string generate(const gjx_program* prog_in, int ppt_code) {
  const int ppt = ppt_code & 255;
  GenCtx g;
  plan_program(prog_in, ppt_code, g);
  Plan& pl = g.pl;
  const gjx_program* prog = pl.prog;
  const bool mfma = pl.mfma;
  const bool wide = pl.wide;
  const int BT = wide ? 1024 : 256;
  const std::string body_s = emit_body(g);
  Emit o;
  // GHI_: the high word of every particle index of the launch (plan_engine keeps a launch inside one 2^32 range): wave-uniform
  o.f("#define GHI_ ((uint32_t)((uint64_t)a.offset >> 32))\n");
  o.f("#include \"gjx_device.h\"\n#include \"gjx_tile.h\"\nusing namespace gjx;\n#define RNG %d\n#define PPT %d\n#define PLOOP _Pragma(\"unroll\") for (int p = 0; p < PPT; ++p)\n",
      prog->rng_mode == GJX_RNG_JAX32 ? GJX_RNG_JAX32 : GJX_RNG_FLAT, ppt);
  if (wide)
    // LPP_ lanes per particle: PT_ = 64 / LPP_ particle lanes per wave, NCH_ = 16 LPP_ chunks of instances (wave, lane within the particle);
    // the chunks' partial sums meet in LDS and are added in chunk order
    o.f("#define LPP_ %d\n#define PT_ (64 / LPP_)\n#define NCH_ (16 * LPP_)\n"
        "#define OWN_ (threadIdx.x < 64u && (threadIdx.x %% LPP_) == 0u)\n"
        "#define PRED_(acc_, base_) do { __syncthreads(); PLOOP pred_[(pw_ * PT_ + pl_) * PPT + p] = acc_[p]; __syncthreads(); \\\n"
        "    PLOOP { float t_ = 0.0f; _Pragma(\"nounroll\") for (int w_ = 0; w_ < NCH_; ++w_) t_ += pred_[(w_ * PT_ + pl_) * PPT + p]; acc_[p] = base_[p] + t_; } } while (0)\n", pl.lpp);
  else o.f("#define OWN_ true\n");
  o.f("#define NTAB %d\n#define NCOMP %d\n", prog->n_tab, pl.comp_floats);
  if (pl.tab_lds) o.f("#define TAB(i) tab_s[i]\n#define COMP(i) tab_s[NTAB + (i)]\n");
  else o.f("#define TAB(i) a.tab[i]\n#define COMP(i) tab_s[i]\n");
  o.f("template <int N> struct VecStore;\n"
      "template <> struct VecStore<1> { static GJX_DEV void st(float* q, const float (&x)[1]) { *q = x[0]; } };\n"
      "template <> struct VecStore<2> { static GJX_DEV void st(float* q, const float (&x)[2]) { *reinterpret_cast<float2*>(q) = make_float2(x[0], x[1]); } };\n"
      "template <> struct VecStore<4> { static GJX_DEV void st(float* q, const float (&x)[4]) { *reinterpret_cast<float4*>(q) = make_float4(x[0], x[1], x[2], x[3]); } };\n");
  o.f("typedef float v4f_ __attribute__((ext_vector_type(4)));\n");
  // steps kernel (gjx_gen_steps): what other blocks of the launch read next step leaves at agent scope, write-through
  o.f("template <int N> GJX_DEV void vec_store_live(float* q, const float (&x)[N]) { _Pragma(\"unroll\") for (int k = 0; k < N; ++k) store_agent(q + k, x[k]); }\n"
      "template <> GJX_DEV void vec_store_live<4>(float* q, const float (&x)[4]) { store_agent_x4(q, x); }\n"
      "#define VSTORE(q, x) do { if (live_) vec_store_live<PPT>(q, x); else VecStore<PPT>::st(q, x); } while (0)\n"
      "#define VSTORE1(q, x) do { if (OWN_) VSTORE(q, x); } while (0)\n"
      "#define LDIN(q) (live_ ? load_agent(q) : *(q))\n"
      "#define TSTAMP(n) do { if (a.tl && threadIdx.x == 0) a.tl[(size_t)blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n");
  if (mfma)    // static LDS: the table (with its companions) and the per-wave transpose patches may exceed the 64 KB a launch can ask for dynamically
    // (two blocks per CU by launch bounds: with at most 256 VGPRs per lane the compiler keeps the matrix-core accumulators in
    // VGPRs — the elementwise phase reads them there; the AGPR form it picks otherwise ran the loop at HALF the matrix rate)
    o.f("template <bool LIVE_>\nstatic __device__ __forceinline__ void gjx_step_(const GenArgs& a) {\n"
        "  __shared__ __attribute__((aligned(16))) float tab_s[%d];\n  __shared__ __attribute__((aligned(16))) float mfma_s[%d];\n"
        "  __shared__ float red[16];\n  __shared__ uint64_t red_q[4];\n", ((prog->n_tab + pl.comp_floats + 3) & ~3) + 4, pl.mfma_floats > 0 ? pl.mfma_floats : 4);
  else
  o.f("template <bool LIVE_>\nstatic __device__ __forceinline__ void gjx_step_(const GenArgs& a) {\n"
      "  extern __shared__ __attribute__((aligned(16))) float tab_s[];\n  __shared__ float red[16];\n  __shared__ uint64_t red_q[4];\n");
  if (wide) o.f("  __shared__ float pred_[1024 * PPT];   // the chunks' partial sums of a plate\n"
                "  const int pw_ = (int)(threadIdx.x >> 6) * LPP_ + (int)((threadIdx.x & 63u) %% LPP_), pl_ = (int)((threadIdx.x & 63u) / LPP_);\n");
  o.f("  constexpr bool live_ = LIVE_;   // steps kernel: agent-scope traffic, granules (compiled out of the one-step kernel)\n  TSTAMP(0);\n");
  if (pl.tab_lds) o.f("  for (int t = threadIdx.x; t < NTAB; t += %d) tab_s[t] = a.tab[t];\n  __syncthreads();\n", BT);
  o.f("#define TSRC(i) TAB(i)\n#define BT_ %d\n", BT);
  emit_companions(o, pl);
  o.f("#undef TSRC\n#undef BT_\n");
  if (!pl.comps.empty()) o.f("  __syncthreads();\n");
  o.f("  TSTAMP(1);\n%s", pl.key_decls.c_str());
  o.f("  const int64_t K = a.K;\n  const int64_t tile = %d * (int64_t)PPT;\n  const int64_t ntiles = (K + tile - 1) / tile;\n"
      "  float tmax = -INFINITY, tsum = 0.0f;\n"
      "  for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {\n"
      "    const int64_t i0 = tix * tile + (int64_t)%s * PPT;\n    if (i0 >= K) break;   // K %% PPT == 0 (launcher)\n"
      "    uint64_t gidx[PPT];\n    PLOOP gidx[p] = (uint64_t)(a.offset + i0 + p);\n"
      "    float score[PPT], weight[PPT];\n    PLOOP { score[p] = 0.0f; weight[p] = 0.0f; }\n", wide ? 64 / pl.lpp : 256, wide ? "pl_" : "threadIdx.x");
  o.f("    float v[%d][PPT];\n", prog->n_slots > 0 ? prog->n_slots : 1);
  if (pl.seq_rows) { o.f("    float* rp_ = a.choices + i0;   // the running row pointer (Plan::seq_rows)\n"); pl.rp_row = 0; }
  {
    bool has_input = false;
    for (int j = 0; j < prog->n_sites; ++j) has_input = has_input || prog->sites[j].mode == GJX_MODE_INPUT;
    if (has_input) o.f("    int64_t src_[PPT];\n    PLOOP src_[p] = a.anc ? (int64_t)a.anc[i0 + p] : i0 + p;\n");
    if (has_input && ppt == 4)
      // gjx_run_resample: the search of the tile-scaled systematic resampler for this block's tile, in front of the reads of the
      // carry (tiled_search_tile, gjx_tile.h: the body of k_resample_gather_tiled) — resample + gather + propagate + reweight in one launch
      o.f("    if (a.rs_logw) {\n      __shared__ TiledSearchShared rs_sh_;\n      __shared__ uint64_t rs_pl_[1026];\n      __shared__ int32_t rs_eb_[1024];\n"
          "      int32_t anc_[4];\n      __syncthreads();\n"
          "      if (live_) tiled_search_tile<false, true>(a.rs_logw, K, (const uint64_t*)a.rs_S, a.rs_E, nullptr, nullptr, (int)ntiles, (int)tix, rs_pl_, rs_eb_, rs_sh_,\n"
          "                               a.rs_lse_out ? 2 : 0, a.rs_lse, a.rs_n_partials, a.rs_lse_out, a.log_k_total, a.rs_u, a.rs_ctrl, live_ && a.tl ? a.tl + (size_t)blockIdx.x * 16 + 8 - (size_t)tix * 8 : nullptr, anc_, a.st_rtag);\n"
          "      else tiled_search_tile<false>(a.rs_logw, K, (const uint64_t*)a.rs_S, a.rs_E, nullptr, nullptr, (int)ntiles, (int)tix, rs_pl_, rs_eb_, rs_sh_,\n"
          "                               a.rs_lse_out ? 2 : 0, a.rs_lse, a.rs_n_partials, a.rs_lse_out, a.log_k_total, a.rs_u, a.rs_ctrl, nullptr, anc_);\n"
          "      TSTAMP(2);\n      PLOOP src_[p] = (int64_t)anc_[p];\n"
          "      if (a.rs_anc_out) *reinterpret_cast<int4*>(a.rs_anc_out + i0) = make_int4(anc_[0], anc_[1], anc_[2], anc_[3]);\n    }\n");
  }
  // rows that already hold values (per-particle constraints, mask flags)
  std::vector<char> pre(prog->n_slots > 0 ? prog->n_slots : 1, 0);
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site& s = prog->sites[j];
    if (pl.info[j].load_here) continue;
    if (s.mode == GJX_MODE_OBS_SLOT || s.mode == GJX_MODE_OBS_MASK) {
      const int n = is_categorical(s.kind) ? 1 : s.dim;
      for (int d = 0; d < n; ++d) pre[s.slot + d] = 1;
      if (s.mode == GJX_MODE_OBS_MASK) pre[s.obs_off] = 1;
    }
  }
  for (int r = 0; r < prog->n_slots; ++r)
    if (pre[r]) o.f("    PLOOP v[%d][p] = a.choices[(int64_t)%d * K + i0 + p];\n", r, r);
  o.s += body_s;
  o.f("    TSTAMP(3);\n    float lw[PPT];\n    PLOOP { float l = weight[p]; if (a.logw_in) l += a.logw_in[i0 + p]; if (a.sub) l -= a.sub[i0 + p]; lw[p] = l; }\n"
      "    if (a.score && OWN_) VecStore<PPT>::st(a.score + i0, score);\n    if (a.weight && OWN_) VecStore<PPT>::st(a.weight + i0, weight);\n"
      "    if (a.logw) VSTORE1(a.logw + i0, lw);\n"
      "    if (!OWN_) { PLOOP lw[p] = -INFINITY; }   // (wide flavour: the other waves hold copies — one contribution per particle to the LSE)\n"
      "    float m4 = tmax;\n    PLOOP m4 = fmaxf(m4, lw[p]);\n"
      "    if (m4 > -INFINITY) { float s4 = tsum * fast_exp(tmax - m4); PLOOP s4 += fast_exp(lw[p] - m4); tsum = s4; }\n    tmax = m4;\n");
  // {e_b, S_b} of this block-tile under GJX_WEIGHTS_TILE_SCALED (include/gjx.h) when it IS a quantisation tile (PPT == 4,
  // K % 1024 == 0: the launcher checks): the resampling kernel that follows needs no pass of its own over the weights
  o.f("    if (PPT == 4 && a.tile_S) {\n      const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;\n"
      "      float m_ = lw[0];\n      PLOOP m_ = fmaxf(m_, lw[p]);\n      const float wm_ = wave_max(m_);\n"
      "      __syncthreads();\n      if (lane == 0) red[8 + wid] = wm_;\n      __syncthreads();\n"
      "      const int e_ = tile_exponent(fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11])));\n"
      "      uint64_t q_ = 0;\n      PLOOP q_ += tile_q(lw[p], e_);\n      const uint64_t wq_ = wave_total_u64(q_);\n"
      "      if (lane == 0) red_q[wid] = wq_;\n"
      "      if (live_) {\n"
      "        // steps kernel: this tile's block pair and then — once every wave's write-through stores of the step have completed —\n"
      "        // its granule {tag, e_b, S_b}: the next step of every block waits for it (one tile per block: the launcher checks)\n"
      "        const float tm_ = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));\n"
      "        float se_ = 0.0f;\n        PLOOP se_ += tm_ > -INFINITY ? fast_exp(lw[p] - tm_) : 0.0f;\n        const float wse_ = wave_sum(se_);\n"
      "        if (lane == 0) red[12 + wid] = wse_;\n        TSTAMP(4);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __syncthreads();\n        TSTAMP(5);\n"
      "        if (threadIdx.x == 0) {\n          const uint64_t tot_ = red_q[0] + red_q[1] + red_q[2] + red_q[3];\n"
      "          __hip_atomic_store(&a.partials[tix], pack_f2(tm_, red[12] + red[13] + red[14] + red[15]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
      "          asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n"
      "          __hip_atomic_store(&a.tile_S[tix * kLiveGranulePad], tile_granule(a.st_tag, tot_ ? e_ : kTileDead, tot_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
      "        }\n      } else {\n      __syncthreads();\n"
      "      if (threadIdx.x == 0) { const uint64_t tot_ = red_q[0] + red_q[1] + red_q[2] + red_q[3]; a.tile_S[tix] = tot_; a.tile_E[tix] = tot_ ? e_ : kTileDead; }\n"
      "      }\n    }\n  }\n");
  if (wide) o.f("  if ((threadIdx.x >> 6) >= 4u) return;   // (the block's LSE pair is made by 256 threads; waves 1 .. 3 contribute nothing, the rest leave)\n");
  o.f("  if (a.partials && !live_) {\n    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;\n"
      "    const float wm = wave_max(tmax);\n    const float ws = wave_sum(wm > -INFINITY ? tsum * fast_exp(tmax - wm) : 0.0f);\n"
      "    if (lane == 0) { red[wid] = wm; red[4 + wid] = ws; }\n    __syncthreads();\n"
      "    float bm = red[0];\n    for (int w = 1; w < 4; ++w) bm = fmaxf(bm, red[w]);\n    float bsum = 0.0f;\n"
      "    for (int w = 0; w < 4; ++w) bsum += bm > -INFINITY ? red[4 + w] * fast_exp(red[w] - bm) : 0.0f;\n"
      "    if (a.lse) lse_publish_and_finish<256>(bm, bsum, a.partials, a.ticket, (int)gridDim.x, a.log_k_total, a.lse, red);\n"
      "    else if (threadIdx.x == 0) a.partials[blockIdx.x] = pack_f2(bm, bsum);\n  }\n  TSTAMP(7);\n}\n");
  {
    // GJX_GEN_WAVES_PER_EU=n: the register budget of n waves per SIMD (512 / n VGPRs) instead of the compiler's own choice
    const char* we = getenv("GJX_GEN_WAVES_PER_EU");
    const int wpe = we ? atoi(we) : 0;
    std::string attr;
    if (wpe >= 1 && wpe <= 8 && !mfma) attr = "__attribute__((amdgpu_waves_per_eu(" + std::to_string(wpe) + ", " + std::to_string(wpe) + "))) ";
    o.f("extern \"C\" __global__ %s__launch_bounds__(%d%s) void gjx_gen(GenArgs a) { gjx_step_<false>(a); }\n", attr.c_str(), BT, mfma ? ", 2" : "");
  }
  {
    bool has_input = false;
    for (int j = 0; j < prog->n_sites; ++j) has_input = has_input || prog->sites[j].mode == GJX_MODE_INPUT;
    if (has_input && ppt == 4 && !mfma && !wide)
      // every step t = T0 .. T-1 of a filter in ONE launch: the step programs share this structure and differ in their tables, keys and
      // comb offsets; a step's kernel boundary is replaced by the granules its blocks publish (co-resident grid: the launcher checks)
      o.f("extern \"C\" __global__ __launch_bounds__(256) void gjx_gen_steps(GenStepsArgs s) {\n"
          "  const int64_t K = s.base.K;\n"
          "  for (int t = s.T0; t < s.T; ++t) {\n"
          "    if (__hip_atomic_load(&s.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusPollTimeout) break;   // a rendezvous timed out: the host repeats the run step by step\n"
          "    GenArgs a = s.base;\n"
          "    const bool odd = (t & 1) != 0, lodd = ((s.T - 1 - t) & 1) != 0;\n"
          "    a.tab = s.tabs[t];\n    a.key = key2{s.keys[2 * t], s.keys[2 * t + 1]};\n"
          "    a.choices = s.rows_all ? s.rows_all + (int64_t)t * s.rows_step : (odd ? s.rows_b : s.rows_a);\n"
          "    a.in_rows = (s.rows_all ? s.rows_all + (int64_t)(t - 1) * s.rows_step : (odd ? s.rows_a : s.rows_b)) + (t == s.T0 ? s.in_row0_first : s.in_row0);\n"
          "    a.in_stride = K;\n    a.anc = nullptr;\n"
          "    a.logw = lodd ? s.logw_b : s.logw_a;\n    a.rs_logw = lodd ? s.logw_a : s.logw_b;\n"
          "    a.tile_S = odd ? s.gran_b : s.gran_a;\n    a.tile_E = nullptr;\n    a.partials = odd ? s.part_b : s.part_a;\n"
          "    a.rs_S = odd ? s.gran_a : s.gran_b;\n    a.rs_E = nullptr;\n    a.rs_lse = (const float*)(odd ? s.part_a : s.part_b);\n"
          "    a.rs_n_partials = (int)(K >> 10);\n    a.rs_lse_out = s.lse_steps + 4 * (int64_t)(t - 1);\n    a.rs_u = s.us[t];\n"
          "    a.rs_anc_out = s.anc_all ? s.anc_all + (int64_t)(t - 1) * K : s.anc;\n    a.rs_ctrl = s.ctrl;\n"
          "    a.tl = (s.timeline && t == s.T / 2) ? s.timeline : nullptr;\n"
          "    a.st_tag = (unsigned long long)((s.epoch + (unsigned)t) %% 15u) + 1ull;\n    a.st_rtag = (unsigned long long)((s.epoch + (unsigned)t - 1u) %% 15u) + 1ull;\n"
          "    gjx_step_<true>(a);\n  }\n}\n");
  }
  // LDS bytes the kernel needs, as a trailing comment the host parses back (keeps one source of truth)
  o.f("// LDS_FLOATS %d\n// BT %d\n", mfma ? 0 : (pl.tab_lds ? prog->n_tab : 0) + pl.comp_floats, BT);
  return o.s;
}

// ---------------------------------------------------------------------------------------------------------
// Generated filter kernels (`gjx_gen_pf`): the step program of a Scan kernel as the MODEL of the one-launch filter skeleton
// pf_core (gjx_pfcore.h) — the skeleton the hand-written linear-Gaussian filter k_pf_persistent runs on.  A tile of 1024
// particles is one block of 16 waves, ONE particle per lane; the step's table is staged in LDS (with its derived constants)
// while the granules of the rendezvous travel, the standard-normal draws of the step's sampled normal sites are taken in the same
// wait (they depend on the key and the particle index only: same streams, same bits), and behind the search a slot gathers
// its carry from its ancestor's row — on this device or, for a collection sharded over the GPUs of a node, through a peer mapping
// — and runs the emitted sites.  Reference: Scan.generate's step recursion (combinators/scan.py:237-294); the filter itself is
// the build's.  SPL = tiles per block (the launcher picks the smallest whose grid is co-resident).
// ---------------------------------------------------------------------------------------------------------
constexpr int kPfMaxHoist = 32;        // standard normals a lane keeps across the search

bool pf_supported(const gjx_program* p) {
  if (p->n_sites < 1 || p->n_tab > 16384) return false;
  bool has_input = false;
  for (int j = 0; j < p->n_sites; ++j) {
    const int md = p->sites[j].mode;
    if (md != GJX_MODE_SAMPLE && md != GJX_MODE_OBS_TAB && md != GJX_MODE_INPUT && md != GJX_MODE_OBS_PROPOSED) return false;
    if (p->sites[j].scan != 0) return false;
    has_input = has_input || md == GJX_MODE_INPUT;
  }
  return has_input && supported_uncached(p);
}

bool discrete_kind(int k) {
  return is_categorical(k) || k == GJX_FLIP || k == GJX_BERNOULLI_LOGITS || k == GJX_POISSON || k == GJX_GEOMETRIC || k == GJX_NEGATIVE_BINOMIAL;
}

// may the filter kernel of this step program carry a rejuvenation move (generate_pf, | 512)?  The move re-scores the PREVIOUS step —
// the same sites under the previous step's table — at candidate values of its latent choices: every own (non-INPUT) site with rows
// must be one the next step reads (the latents ARE the carry), no plates, at least one continuous carry row
bool pf_moves_supported(const gjx_program* p) {
  if (!pf_supported(p)) return false;
  int n_in = 0, own_rows = 0, cont = 0;
  for (int j = 0; j < p->n_sites; ++j) {
    const gjx_site& s = p->sites[j];
    if (s.plate != 0 || s.kind == GJX_DIRICHLET) return false;
    const int w = (is_categorical(s.kind) && s.mode != GJX_MODE_INPUT) ? 1 : s.dim;
    if (s.mode == GJX_MODE_INPUT) { n_in += s.dim; if (!discrete_kind(s.kind)) cont += s.dim; }
    else if (s.slot >= 0 && s.mode != GJX_MODE_OBS_PROPOSED) own_rows += w;
  }
  return cont > 0 && own_rows == n_in && p->n_slots == 2 * n_in;      // a periodic step: as many carry rows out as in
}

// spl_code: tiles per block | 256 for the flavour that runs on a collection sharded over peer-mapped windows (gjx_peer.hip)
//           | 1024 multinomial resampling by sorted uniforms instead of the systematic comb (pf_core's MULTI flavour)
//           | 512 with a rejuvenation move behind every resampling (GenPfArgs::n_moves random-walk Metropolis steps per particle)
std::string generate_pf(const gjx_program* prog_in, int spl_code) {
  if (!pf_supported(prog_in)) return "";
  const int spl = spl_code & 255;
  const bool sharded = (spl_code & 256) != 0;
  const bool moves = (spl_code & 512) != 0;
  const bool multi = (spl_code & 1024) != 0;         // multinomial resampling by sorted uniforms (pf_core<..., MULTI>)
  if (moves && !pf_moves_supported(prog_in)) return "";
  GenCtx g;
  plan_program(prog_in, 1, g, false);
  Plan& pl = g.pl;
  const gjx_program* prog = pl.prog;
  pl.pf = true;
  pl.seq_rows = false;
  pl.tab_lds = true;
  // ---- which draws are taken ahead: sampled normal sites outside plates, whole scalar-normal runs or none of a run ----
  const int ns = prog->n_sites;
  pl.hoist_at.assign(ns, -1);
  if (!getenv("GJX_GEN_NO_HOIST")) {
    for (int j = 0; j < ns; ++j) {
      const gjx_site& s = prog->sites[j];
      const RollInfo& ri = pl.info[j];
      if (ri.plate || s.mode != GJX_MODE_SAMPLE || !is_normal(s.kind) || s.dim > kMaxExpandDim || pl.hoist_at[j] >= 0) continue;
      // what is taken ahead are the stream's 32-bit WORDS (the hashes: arithmetic that hides nothing); the normals are finished
      // inside the slot, behind the loads of the carry, as the hand-written model does (its timeline: 1.1 us for this phase
      // against 2.3 with Box-Muller in it).  FLAT pairs elements (2 k, 2 k + 1): the words of whole pairs are kept
      const bool flat_ = prog->rng_mode != GJX_RNG_JAX32;
      if (pl.stream[j].run >= 0) {
        if (!pl.stream[j].opens) continue;                       // (a member is decided with its head)
        int members = 0;
        for (int l = j; l < ns; ++l) members += pl.stream[l].run == pl.stream[j].run;
        const int words = flat_ ? 2 * ((members + 1) / 2) : members;
        if (pl.n_hoist + words > kPfMaxHoist) continue;
        for (int l = j; l < ns; ++l) if (pl.stream[l].run == pl.stream[j].run) pl.hoist_at[l] = pl.n_hoist;      // (the run's base; a member's element picks its word)
        pl.n_hoist += words;
      } else {
        const int words = flat_ ? 2 * ((s.dim + 1) / 2) : s.dim;
        if (pl.n_hoist + words > kPfMaxHoist) continue;
        pl.hoist_at[j] = pl.n_hoist;
        pl.n_hoist += words;
      }
    }
  }
  int n_input_sites = 0;
  while (n_input_sites < ns && prog->sites[n_input_sites].mode == GJX_MODE_INPUT) ++n_input_sites;
  for (int j = n_input_sites; j < ns; ++j) if (prog->sites[j].mode == GJX_MODE_INPUT) return "";      // (INPUT sites come first: include/gjx.h)
  const std::string inputs_s = emit_body(g, 0, n_input_sites);
  const std::string body_s = emit_body(g, n_input_sites, ns);
  int n_in = 0;
  for (int j = 0; j < ns; ++j) if (prog->sites[j].mode == GJX_MODE_INPUT) n_in += prog->sites[j].dim;
  // ---- the rejuvenation move's target: the ASSESS form of this step program — inputs and latents given (registers), the model's
  //      sites scored, proposal sites left out — evaluated under the PREVIOUS step's table (second LDS copy)
  GenCtx ga;
  std::vector<gjx_site> asites;
  gjx_program aprog = *prog_in;
  std::string assess_s;
  if (moves) {
    asites.assign(prog_in->sites, prog_in->sites + prog_in->n_sites);
    for (auto& sa : asites)
      if (sa.mode == GJX_MODE_SAMPLE && !(sa.flags & GJX_SITE_PROPOSAL)) sa.mode = GJX_MODE_OBS_PROPOSED;   // "given, in its registers"
    aprog.sites = asites.data();
    plan_program(&aprog, 1, ga, false);
    ga.pl.pf = true;
    ga.pl.seq_rows = false;
    ga.pl.tab_lds = true;
    ga.pl.skip.assign(ns, 0);
    for (int j = 0; j < ns; ++j) ga.pl.skip[j] = asites[j].mode == GJX_MODE_INPUT || (asites[j].flags & GJX_SITE_PROPOSAL);
    assess_s = emit_body(ga);
  }
  Emit o;
  o.f("#define GHI_ 0u   /* a filter's collection: K_total <= 2^25 slots */\n");
  o.f("#include \"gjx_device.h\"\n#include \"gjx_pfcore.h\"\nusing namespace gjx;\n#define RNG %d\n#define PPT 1\n#define PLOOP _Pragma(\"unroll\") for (int p = 0; p < PPT; ++p)\n",
      prog->rng_mode == GJX_RNG_JAX32 ? GJX_RNG_JAX32 : GJX_RNG_FLAT);
  o.f("#define NTAB %d\n#define NCOMP %d\n#define SPL %d\n#define NHOIST %d\n#define TAB(i) tab_s[i]\n#define COMP(i) tab_s[NTAB + (i)]\n", prog->n_tab, pl.comp_floats, spl, pl.n_hoist);
  o.f("#define NCOMPA %d\n#define MOVES %d\n", moves ? ga.pl.comp_floats : 0, moves ? 1 : 0);
  o.f("template <int N> struct VecStore;\n"
      "template <> struct VecStore<1> { static GJX_DEV void st(float* q, const float (&x)[1]) { *q = x[0]; } };\n"
      "typedef float v4f_ __attribute__((ext_vector_type(4)));\n"
      "#define VSTORE(q, x) do { if (act_) store_scoped((q), (x)[0], sys_); } while (0)\n"
      "#define VSTORE1(q, x) VSTORE(q, x)\n#define OWN_ true\n"
      "#define LDIN(q) load_scoped((q), sys_)\n");
  o.f("struct GenPfModel {\n  const GenPfArgs& f;\n  float* const tab_s;\n"
      "  struct Draws { float nz[NHOIST > 0 ? NHOIST : 1]; };\n"
      "  float* cur_;            // rows of the step being produced, and the OWN rows of the step before it: chosen ONCE per step (stage)\n"
      "  const float* in_;\n"
      "  const float* pp_;       // MOVES: the INPUT rows the previous step stored — what the ancestor itself was propagated from\n"
      "  float* const tabp_s;    // MOVES: the previous step's table (+ the derived constants of the assess form)\n"
      "  unsigned acc_lane;      // MOVES: accepted moves of this lane's slots\n"
      "  const float* tbn_;      // the NEXT step's table: its address is read a step ahead (stage() then waits for ONE dependent load, not two)\n"
      "  GJX_DEV GenPfModel(const GenPfArgs& a, float* t, float* tp) : f(a), tab_s(t), cur_(nullptr), in_(nullptr), pp_(nullptr), tabp_s(tp), acc_lane(0u), tbn_(nullptr) {}\n"
      "  GJX_DEV float* rows(int t) const { return f.rows_all ? f.rows_all + (int64_t)t * f.rows_step : ((t & 1) ? f.rows_b : f.rows_a); }\n"
      "  GJX_DEV const float* in_rows(int t) const { return rows(t - 1) + (t == 1 ? f.in_row0_first : f.in_row0); }   // the OWN rows of step t - 1\n"
      "  GJX_DEV void prologue(int) { tbn_ = f.tabs[1]; }\n"
      "  GJX_DEV void epilogue(int lane) {\n    if (MOVES && f.acc_total) {\n      const unsigned wacc = wave_scan_u32(acc_lane);\n"
      "      if (lane == 63 && wacc) atomicAdd(f.acc_total, (unsigned long long)wacc);\n    }\n  }\n");
  // ---- stage: the step's table and what derives from it, while the granules travel ----
  o.f("  GJX_DEV void stage(int t, int tid) {\n    cur_ = rows(t);\n    in_ = in_rows(t);\n    const float* __restrict__ tb_ = tbn_;\n"
      "    if (t + 1 < f.core.T) tbn_ = f.tabs[t + 1];\n"
      "    for (int e = tid; e < NTAB; e += %d) tab_s[e] = tb_[e];\n#define TSRC(i) tb_[i]\n#define BT_ %d\n", 1024, 1024);
  {
    Emit c;
    emit_companions(c, pl);
    o.s += c.s;
  }
  o.f("#undef TSRC\n#undef BT_\n");
  if (moves) {
    o.f("    if (t >= 2) {      // the rejuvenation move re-scores step t - 1: its table, the constants derived from it, its stored inputs\n"
        "      pp_ = rows(t - 1);\n      const float* __restrict__ tq_ = f.tabs[t - 1];\n"
        "      for (int e = tid; e < NTAB; e += 1024) tabp_s[e] = tq_[e];\n#define TSRC(i) tq_[i]\n#define BT_ 1024\n#undef COMP\n#define COMP(i) tabp_s[NTAB + (i)]\n");
    Emit c;
    emit_companions(c, ga.pl);
    o.s += c.s;
    o.f("#undef TSRC\n#undef BT_\n#undef COMP\n#define COMP(i) tab_s[NTAB + (i)]\n    }\n");
  }
  o.f("  }\n");
  // ---- draw: the standard normals of the hoisted sites (the statements the sites themselves would run) ----
  o.f("  GJX_DEV void draw(int, key2 key, uint64_t gidx_, Draws& d) const {\n    (void)key; (void)gidx_; (void)d;\n");
  for (int j = 0; j < ns; ++j) {
    if (pl.hoist_at[j] < 0) continue;
    const SiteStream& ss = pl.stream[j];
    const gjx_site& s = prog->sites[j];
    if (ss.run >= 0) {
      if (!ss.opens) continue;
      int members = 0;
      for (int l = j; l < ns; ++l) members += pl.stream[l].run == ss.run;
      const int words = prog->rng_mode != GJX_RNG_JAX32 ? 2 * ((members + 1) / 2) : members;
      o.f("    { BitStream<RNG> bs; bs.open(key, gidx_, %du);   // scalar-normal run %d: the words of its %d elements\n", ss.site_no, ss.run, members);
      for (int e = 0; e < words; ++e) o.f("      d.nz[%d] = __uint_as_float(bs.get(%du));\n", pl.hoist_at[j] + e, e);
      o.f("    }\n");
    } else {
      const int words = prog->rng_mode != GJX_RNG_JAX32 ? 2 * ((s.dim + 1) / 2) : s.dim;
      o.f("    { BitStream<RNG> bs; bs.open(key, gidx_, %du);   // site %d: the words of its %d elements\n", ss.site_no, j, s.dim);
      for (int e = 0; e < words; ++e) o.f("      d.nz[%d] = __uint_as_float(bs.get(%du));\n", pl.hoist_at[j] + e, e);
      o.f("    }\n");
    }
  }
  o.f("  }\n");
  // ---- verify mode (GJX_PEER_VERIFY, gjx_peer.hip): the check word of the rows the previous LAUNCH (step 0) wrote ----
  o.f("  GJX_DEV void seal_prev(int t_prev, int j, uint32_t gslot, const PfSlotCtx& cx) const {\n"
      "    const float* r_ = in_rows(t_prev + 1);\n    uint32_t h = row_check_init(t_prev, gslot);\n");
  for (int j = 0; j < ns; ++j) {
    const gjx_site& s = prog->sites[j];
    if (s.mode != GJX_MODE_INPUT) continue;
    for (int d = 0; d < s.dim; ++d) o.f("    h = row_check_mix(h, r_[(int64_t)%d * cx.K + j]);\n", s.obs_off + d);
  }
  o.f("    store_scoped_u32(cx.chk_cur + j, cx.verify == 2 ? h ^ 1u : h, cx.sys);\n  }\n");
  // ---- MOVES: log-density of the previous step at candidate values of its latent choices (the assess form, previous table) ----
  struct InRow { int slot, row, obs; bool cont; };
  std::vector<InRow> inrows;
  for (int j = 0; j < ns; ++j) {
    const gjx_site& s = prog->sites[j];
    if (s.mode != GJX_MODE_INPUT) continue;
    for (int d = 0; d < s.dim; ++d) inrows.push_back({s.slot + d, pl.info[j].row + d, s.obs_off + d, !discrete_kind(s.kind)});
  }
  int n_cont = 0;
  for (auto& r : inrows) n_cont += r.cont ? 1 : 0;
  // (NCPAD: the continuous rows rounded up to an even count — FLAT normals come in Box-Muller PAIRS of elements (2k, 2k + 1), and the
  // accept's uniform must not share its element with the partner of the last normal: that correlates the proposal with its own accept)
  o.f("#define NIN %d\n#define NCONT %d\n#define NCPAD %d\n", (int)inrows.size() > 0 ? (int)inrows.size() : 1, n_cont, n_cont + (n_cont & 1));
  if (moves) {
    o.f("#undef TAB\n#undef COMP\n#define TAB(i) tabp_s[i]\n#define COMP(i) tabp_s[NTAB + (i)]\n"
        "  GJX_DEV float logpi(const float (&pin_)[NIN], const float (&xc_)[NIN]) const {\n"
        "    constexpr bool live_ = true; (void)live_;\n    const int64_t K = 0, i0 = 0; (void)K; (void)i0;\n"
        "    struct { float* site_scores; float* choices; } a; a.site_scores = nullptr; a.choices = nullptr; (void)a;\n"
        "    float score[PPT] = {0.0f}, weight[PPT] = {0.0f};\n    float v[%d][PPT];\n", prog->n_slots > 0 ? prog->n_slots : 1);
    for (size_t r = 0; r < inrows.size(); ++r)
      o.f("    v[%d][0] = pin_[%d]; v[%d][0] = xc_[%d];\n", inrows[r].slot, (int)r, n_in + inrows[r].obs, (int)r);
    o.s += assess_s;
    o.f("    (void)score;\n    return weight[0];\n  }\n#undef TAB\n#undef COMP\n#define TAB(i) tab_s[i]\n#define COMP(i) tab_s[NTAB + (i)]\n");
  }
  // ---- slot: gather the carry through the ancestor, [move it,] run the sites, store the rows; -> the incremental log-weight ----
  o.f("  GJX_DEV float slot(int t, key2 key_, int j, bool act_, int sg, int sl, uint64_t gidx_, const Draws* hoisted, const PfSlotCtx& cx) {\n"
      "    constexpr bool live_ = true; (void)live_;\n    const int64_t K = cx.K, i0 = j;\n    const bool sys_ = cx.sys;\n"
      "    struct { key2 key; float* choices; const float* in_rows; int64_t in_stride; int store_inputs; float* site_scores; const float* tab; } a;\n"
      "    a.key = key_; a.choices = cur_; a.in_rows = peer_ptr(in_, cx.sPD[sg]); a.in_stride = K; a.store_inputs = 0; a.site_scores = nullptr; a.tab = nullptr;\n"
      "    uint64_t gidx[PPT] = {gidx_};\n    int64_t src_[PPT] = {(int64_t)sl};\n"
      "    float score[PPT] = {0.0f}, weight[PPT] = {0.0f};\n    float v[%d][PPT];\n"
      "    Draws late_;\n    const Draws* dr_ = hoisted;\n    if (!dr_) { draw(t, key_, gidx_, late_); dr_ = &late_; }\n    (void)dr_;\n",
      prog->n_slots > 0 ? prog->n_slots : 1);
  o.s += inputs_s;
  // verify (reader side): the pulled carry rows against their owner's check word — before anything moves them
  o.f("    if (cx.verify) {\n      const unsigned want = load_scoped_u32(peer_ptr((const unsigned*)cx.chk_prev, cx.sPD[sg]) + sl, sys_);\n"
      "      uint32_t h = row_check_init(t - 1, (uint32_t)((int64_t)sg * K + sl));\n");
  for (auto& r : inrows) o.f("      h = row_check_mix(h, v[%d][0]);\n", r.slot);
  o.f("      if (act_ && cx.live && h != want) cx.mismatch();\n    }\n");
  if (moves) {
    // resample-move (requests/rejuvenate.py:70-94 with a symmetric random-walk proposal; the caller-side accept of
    // tests/inference/test_requests.py:131-137, fused): the gathered carry x_{t-1} takes n_moves Metropolis steps that leave
    // p(x_{t-1} | the ancestor's own inputs, the observations of step t-1) invariant.  Stream: site 1022 of the step's key, move n
    // uses elements n (NCPAD + 2) + c for the c-th continuous row and n (NCPAD + 2) + NCPAD for the accept's uniform.
    o.f("    if (f.n_moves > 0 && t >= 2) {\n      float pin_[NIN], xc_[NIN];\n      const float* ppr_ = peer_ptr(pp_, cx.sPD[sg]) + sl;\n");
    for (size_t r = 0; r < inrows.size(); ++r)
      o.f("      pin_[%d] = LDIN(ppr_ + (int64_t)%d * K); xc_[%d] = v[%d][0];\n", (int)r, inrows[r].row, (int)r, inrows[r].slot);
    o.f("      float curlp_ = logpi(pin_, xc_), nacc_ = 0.0f;\n      BitStreamRT<RNG> bm_;\n      bm_.open(key_, gidx_, %du);\n"
        "      for (int n_ = 0; n_ < f.n_moves; ++n_) {\n        float xq_[NIN];\n", (unsigned)(GJX_FLAT_MAX_SITES - 1));
    {
      int c = 0;
      for (size_t r = 0; r < inrows.size(); ++r) {
        if (inrows[r].cont) o.f("        xq_[%d] = fmaf(f.move_scale, stream_normal<RNG>(bm_, (uint32_t)(n_ * (NCPAD + 2) + %d)), xc_[%d]);\n", (int)r, c++, (int)r);
        else o.f("        xq_[%d] = xc_[%d];\n", (int)r, (int)r);
      }
    }
    o.f("        const float prop_ = logpi(pin_, xq_);\n"
        "        const float lu_ = safe_log(uniform_from_bits(bm_.get((uint32_t)(n_ * (NCPAD + 2) + NCPAD)), kTiny, 1.0f));\n"
        "        if (lu_ < prop_ - curlp_) {\n          _Pragma(\"unroll\") for (int r_ = 0; r_ < NIN; ++r_) xc_[r_] = xq_[r_];\n          curlp_ = prop_; nacc_ += 1.0f;\n        }\n      }\n");
    for (size_t r = 0; r < inrows.size(); ++r) o.f("      v[%d][0] = xc_[%d];\n", inrows[r].slot, (int)r);
    o.f("      if (act_) acc_lane += (unsigned)nacc_;\n    }\n");
    // the (moved) inputs are kept: the next step's move conditions on them as ITS ancestor's inputs
    o.f("    if (f.n_moves > 0) {\n");
    for (auto& r : inrows) o.f("      VSTORE(a.choices + (int64_t)%d * K + i0, v[%d]);\n", r.row, r.slot);
    o.f("    }\n");
  }
  o.s += body_s;
  // verify (writer side): this slot's own rows — what the next step pulls — get their check word
  o.f("    if (cx.verify && act_) {\n      uint32_t g_ = row_check_init(t, (uint32_t)gidx_);\n");
  for (auto& r : inrows)
    // (the rows the NEXT step reads: this step's own rows, numbered like the rows this step read of the step before it; read back
    // from this lane's own stores — with plates the values of all instances are not in registers any more)
    o.f("      g_ = row_check_mix(g_, load_scoped(a.choices + (int64_t)%d * K + i0, sys_));\n", n_in + r.obs);
  o.f("      store_scoped_u32(cx.chk_cur + j, cx.verify == 2 ? g_ ^ 1u : g_, sys_);\n    }\n");
  o.f("    (void)score;\n    return weight[0];\n  }\n};\n");
  // (one rank: agent-scope accesses and no verify mode compiled in; GJX_PF_SHARDED: the peer-sharded flavour decides both at run time)
  o.f("extern \"C\" __global__ __launch_bounds__(1024) void gjx_gen_pf(GenPfArgs a) {\n"
      "  extern __shared__ __attribute__((aligned(16))) unsigned char pf_dyn[];\n"
      "  __shared__ __attribute__((aligned(16))) float tab_s[%d];\n"
      "  __shared__ __attribute__((aligned(16))) float tabp_s[%d];\n"
      "  GenPfModel m(a, tab_s, tabp_s);\n  pf_core<GenPfModel, SPL, %s, %s>(a.core, m, pf_dyn);\n}\n", ((prog->n_tab + pl.comp_floats + 3) & ~3) + 4,
      moves ? ((prog->n_tab + ga.pl.comp_floats + 3) & ~3) + 4 : 4, sharded ? "2, 2" : "0, 0", multi ? "true" : "false");
  o.f("// LDS_FLOATS 0\n");
  return o.s;
}

// ---------------------------------------------------------------------------------------------------------
// Generated HMC kernels.  The reference differentiates ANY model with jax.grad over gen_fn.assess and lets XLA fuse the
// leapfrog loop (inference/requests/hmc.py:70-96, 156-211).  Here the site list becomes ONE straight-line kernel
// `gjx_hmc_gen`: the chain's values, gradient, momenta and (compat mode) initial gradient live in registers for the
// whole trajectory; a sweep over the sites computes the score and its ANALYTIC gradient in one forward pass (every
// parameter form reads value slots directly, so d score / d slot is a sum of per-site terms: dlogpdf x the parameter's
// transform derivative x the affine row) and is emitted twice — with and without the score, which only the two ends of the
// trajectory need; the float table sits in LDS; L leapfrog steps and the MH accept run inside the kernel.  A site with
// more than kMaxExpandDim elements (an observed plate: the N rows of a regression likelihood) is a rolled loop whose
// elements are dealt round-robin to the CPL = 4 lanes that share a chain — each lane accumulates its part of the score and
// of the gradient rows, joined by two DPP butterflies per accumulator at the end of the site — so a program with a big
// likelihood runs 4 waves per SIMD instead of 1.  Streams, semantics and results: k_hmc_generic's (gjx_hmc.hip), to
// float summation order.
// ---------------------------------------------------------------------------------------------------------
constexpr int kHmcMaxSel = 48, kHmcMaxSlots = 96, kHmcMaxTab = 36864;   // 144 KB of LDS for the table
constexpr int kHmcBigSlots = 512;   // chain values of the LDS-state flavour (HmcPlan::big)

struct HmcPlan {
  // the EMITTED site list: the program's, or — with plates — plate_program's (every slot a register; a body site owns one instance's
  // worth, its rows in choices[][] are RollInfo::row + i_ * d_row); n_regs = registers of the chain state
  std::vector<gjx_site> sites;
  std::vector<RollInfo> info;
  int n_regs = 0;
  bool plates = false;
  int nsel = 0;
  std::vector<int> sel_of_slot;    // slot -> index among the selected slots, or -1
  std::vector<int> slot_of_sel;
  bool looped = false;             // some site is a rolled loop (CPL = 4)
  // selected sites INSIDE plates (a latent per instance that HMC moves too): their positions, momenta and gradients are rows of the
  // caller's workspace — [4][prows][n]: q, p, g, g0 — that the lane which owns the instance reads and writes in the sweep and in the
  // leapfrog updates; the selected indices m >= nout name one instance's worth of them (registers of the plate loop's body)
  struct PlateSel { int j, reg, dim, row, d_row, plate_n, prow0, leaf, m0; };
  std::vector<PlateSel> psel;
  int prows = 0, nout = 0;
  std::vector<int> leaf_of_site;   // momentum leaf (hmc.py:120-130: one per selected address, in program order) or -1
  // a long periodic Scan (detect_roll): the steps are dealt to the lanes of a chain in contiguous chunks; the values of the previous and
  // of the current step live in registers, the trajectory state of the steps' selected values in workspace rows t * ssel + k
  bool rolled = false;
  Roll roll;
  int ssel = 0, sl_step = 0, leaf0_scan = 0;
  std::vector<int> step_ls;        // k -> the selected value's slot within its step (0 .. S - 1)
  // rolled sites whose AFFINE parameter runs on the matrix cores (hmc_emit_mfma_site): parameter index or -1 per site, offset
  // of the site's transposed matrix in the second LDS array, and what the layout needs from the launch
  bool mfma = false;
  std::vector<int> mf_k, xt_off;
  int xt_floats = 0, block = 256;
  // chain state beyond the register budget (more than kHmcMaxSlots values or kHmcMaxSel selected ones: a network's weights, a model
  // written as a hundred scalar sites): values, gradient, momenta (and the first gradient when it fits: nostale otherwise) live in
  // LDS COLUMNS of a one-wave block — slot s of lane l at st[s * 64 + l]: conflict-free, indexed like the register arrays by the
  // same emitted code —, the table is read from memory (the lanes of a wave read the same entry)
  bool big = false, nostale = false;
  // bernoulli(logits = X v + b) with every observation 0 or 1: rows of X (both copies) and the bias are folded with the sign
  // s = 2 y - 1 and log2 e in the kernel's prologue — log p = log sigmoid(s a), d/da = s sigmoid(-s a): the element costs
  // exp2, add, rcp and needs neither y nor a subtraction (what the hand-written k_hmc_logreg_mfma2 does for its one shape)
  std::vector<char> fold;
  std::vector<int> bs_off;         // the folded bias vector of a fold site, behind its transposed matrix
};

// may site j's matrix (parameter k) be folded with the observations' signs IN the kernel's LDS copy of the table?  A bernoulli-logits
// site without a transform, observed from the table with values 0 / 1 only (the host copy of the table is read: the answer is part
// of the kernel's cache key, hmc_variant), and a matrix no other parameter or observation of the program shares storage with
bool hmc_fold_ok(const gjx_program* p, int j, int k) {
  const gjx_site& s = p->sites[j];
  if (getenv("GJX_HMC_GEN_NO_FOLD") || s.kind != GJX_BERNOULLI_LOGITS || k != 0 || s.p[0].xf != GJX_XF_NONE || s.slot >= 0 || s.mode != GJX_MODE_OBS_TAB || !p->tab) return false;
  for (int d = 0; d < s.dim; ++d) { const float y = p->tab[s.obs_off + d]; if (y != 0.0f && y != 1.0f) return false; }
  const int lo = s.p[0].moff, hi = lo + s.dim * s.p[0].n;
  auto hits = [&](int a, int n) { return n > 0 && a < hi && a + n > lo; };
  for (int jj = 0; jj < p->n_sites; ++jj) {
    const gjx_site& t = p->sites[jj];
    const int rows = is_categorical(t.kind) ? t.ncat : t.dim;
    if (t.mode == GJX_MODE_OBS_TAB && hits(t.obs_off, is_categorical(t.kind) ? 1 : t.dim)) return false;
    for (int kk = 0; kk < (is_categorical(t.kind) ? 1 : n_params(t.kind)); ++kk) {
      const gjx_param& q = t.p[kk];
      if (q.op == GJX_P_EXPR) return false;      // (a block may read any table entry: no folding beside it)
      if (q.op == GJX_P_CONST && hits(q.off, q.len)) return false;
      if (q.op == GJX_P_GATHER && hits(q.off, q.n * q.len)) return false;
      if (q.op == GJX_P_AFFINE) {
        if (hits(q.off, q.len)) return false;
        if (!(jj == j && kk == k) && hits(q.moff, rows * q.n)) return false;
      }
    }
  }
  return true;
}

// a rolled site for the matrix cores: exactly one AFFINE parameter over n = 16, 32, 48 or 64 values with 16-byte aligned rows,
// rows in multiples of 16
int hmc_mfma_param(const gjx_site& s) {
  if (s.dim <= kMaxExpandDim || (s.dim & 15) || is_categorical(s.kind) || s.kind == GJX_DIRICHLET) return -1;
  int k_aff = -1;
  for (int k = 0; k < n_params(s.kind); ++k) {
    const gjx_param& q = s.p[k];
    if (q.op != GJX_P_AFFINE) continue;
    if (k_aff >= 0 || q.n < 16 || q.n > 64 || (q.n & 15) || (q.moff & 3) || (q.len != 1 && q.len != s.dim)) return -1;
    k_aff = k;
  }
  return k_aff;
}

bool hmc_elementwise(int kind) { return !is_categorical(kind) && kind != GJX_DIRICHLET; }

// roll: plan the program as a rolled Scan (tried when the straight-line plan does not fit)
bool hmc_plan_form(const gjx_program* p, HmcPlan* out, bool roll) {
  if (p->n_sites < 1 || p->n_slots < 1) return false;
  g_expr_prog = p;
  HmcPlan pl;
  if (roll) {
    const Roll r = detect_roll(p, true);
    if (!r.ok || r.n_post != 0 || getenv("GJX_HMC_GEN_NO_ROLL")) return false;
    pl.sites = r.sites; pl.info = r.info; pl.n_regs = r.n_regs; pl.rolled = true; pl.roll = r; pl.looped = true;
  } else {
    const PlateXf px = plate_program(p);
    if (px.any) {
      if (!px.ok) return false;
      pl.sites = px.sites; pl.info = px.info; pl.n_regs = px.n_regs; pl.plates = true;
    } else {
      pl.sites.assign(p->sites, p->sites + p->n_sites);
      for (int j = 0; j < p->n_sites; ++j) { RollInfo ri; ri.row = p->sites[j].slot; ri.score_row = j; pl.info.push_back(ri); }
      pl.n_regs = p->n_slots;
    }
  }
  const int ns = (int)pl.sites.size();
  if (ns > 64 || pl.n_regs < 1) return false;
  if (pl.n_regs > kHmcMaxSlots) { if (pl.rolled || pl.n_regs > kHmcBigSlots) return false; pl.big = true; }
  pl.sel_of_slot.assign(pl.n_regs, -1);
  int unrolled = 0;
  for (int j = 0; j < ns; ++j) {
    const gjx_site& s = pl.sites[j];
    const RollInfo& ri = pl.info[j];
    if (s.mode == GJX_MODE_INPUT) {      // a per-chain value the other sites read (registers, loaded with the chain's values): no density
      if (ri.plate || (pl.rolled && j >= pl.roll.i0) || (s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) return false;
      continue;
    }
    if (s.kind < 1 || s.kind >= GJX_KIND_MAX || s.kind == GJX_DIRICHLET) return false;
    if (s.mode != GJX_MODE_OBS_TAB && s.mode != GJX_MODE_OBS_SLOT) return false;
    for (int k = 0; k < 4; ++k) if (ri.mem_slot[k] >= 0) return false;     // (instances of another plate, one instance read from outside: the interpreter)
    if (ri.plate) {
      // a plate's body: scored and differentiated instance by instance inside the sweep; what HMC moves lives outside the plate
      // (a selected per-instance latent would be plate_n x dim chain registers: the site interpreter keeps that state in memory)
      if (!is_categorical(s.kind) && s.dim > kMaxExpandDim) return false;
      if ((s.flags & GJX_SITE_HMC_SELECTED) && (is_categorical(s.kind) || s.slot < 0 || getenv("GJX_HMC_GEN_NO_PLATE_SEL"))) return false;
      pl.looped = true;
    }
    if (is_categorical(s.kind)) {
      if (s.p[0].op != GJX_P_CONST && s.p[0].op != GJX_P_GATHER) return false;
      if (s.ncat < 1 || s.ncat > 64 || (s.flags & GJX_SITE_HMC_SELECTED)) return false;
      continue;
    }
    if (s.dim < 1) return false;
    const bool big = s.dim > kMaxExpandDim;
    if (big && (s.slot >= 0 || (s.flags & GJX_SITE_HMC_SELECTED))) return false;   // a rolled site reads its values from the table
    for (int k = 0; k < n_params(s.kind); ++k) {
      const gjx_param& q = s.p[k];
      if (q.op == GJX_P_EXPR) {      // forward nodes and the reverse sweep as straight-line code; straight-line programs only
        if (!expr_block_ok(p, q, pl.n_regs, s.dim, pl.plates || pl.rolled) || (big && q.len != 1)) return false;
        for (const auto& lf : ri.eleaf[k]) if (lf.reg < 0) return false;      // (a leaf in another plate's rows: the interpreter)
        unrolled += q.n / 4;
        continue;
      }
      if (q.op < GJX_P_CONST || q.op > GJX_P_VGATHER) return false;
      if (q.op == GJX_P_AFFINE && (q.n < 1 || q.n > 64)) return false;
      if (big && q.op == GJX_P_VALUE && q.len != 1) return false;
      // a row of a (selected) choice picked by a discrete choice: select chains forwards and backwards, few rows
      if (q.op == GJX_P_VGATHER && (q.n < 1 || q.n * q.len > 32 || q.moff < 0 || q.moff + q.n * q.len > pl.n_regs || (big && q.len != 1))) return false;
    }
    if (big && pl.rolled) return false;                 // (a rolled site inside a rolled Scan: not emitted)
    if (big) pl.looped = true; else unrolled += s.dim;
    const bool step_site = pl.rolled && j >= pl.roll.i0;      // the Scan's sites: their selected values get indices behind the others'
    if ((s.flags & GJX_SITE_HMC_SELECTED) && s.slot >= 0 && !ri.plate && !step_site)
      for (int d = 0; d < s.dim; ++d) { pl.sel_of_slot[s.slot + d] = pl.nsel++; pl.slot_of_sel.push_back(s.slot + d); }
  }
  pl.nout = pl.nsel;
  pl.leaf_of_site.assign(ns, -1);
  if (pl.rolled) {
    // selected values of a step: k = 0 .. ssel - 1 in site order; indices NOUT + k for the PREVIOUS step's registers, NOUT + ssel + k
    // for the current step's (step 0's sites live in the current step's registers too: detect_roll)
    const Roll& r = pl.roll;
    int leaf = 0;
    for (int j = 0; j < r.i0; ++j) if ((pl.sites[j].flags & GJX_SITE_HMC_SELECTED) && pl.sites[j].slot >= 0) pl.leaf_of_site[j] = leaf++;
    pl.leaf0_scan = leaf;
    for (int l = 0; l < r.m; ++l) {
      const gjx_site& s = pl.sites[r.i0 + r.m + l];          // the loop body's site (registers of the current step)
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      if (is_categorical(s.kind)) return false;
      pl.leaf_of_site[r.i0 + l] = pl.leaf_of_site[r.i0 + r.m + l] = pl.sl_step++;       // (leaf within the step)
      for (int d = 0; d < s.dim; ++d) pl.step_ls.push_back(s.slot + d - (r.n_pre + r.S));
    }
    pl.ssel = (int)pl.step_ls.size();                        // (0: nothing of the Scan is moved — the loop then only feeds the gradient rows in front of it)
    for (int k = 0; k < pl.ssel; ++k) { pl.sel_of_slot[r.n_pre + pl.step_ls[k]] = pl.nsel++; pl.slot_of_sel.push_back(r.n_pre + pl.step_ls[k]); }
    for (int k = 0; k < pl.ssel; ++k) { pl.sel_of_slot[r.n_pre + r.S + pl.step_ls[k]] = pl.nsel++; pl.slot_of_sel.push_back(r.n_pre + r.S + pl.step_ls[k]); }
    pl.prows = r.T * pl.ssel;
  } else {
    int leaf = 0;
    for (int j = 0; j < ns; ++j) {
      const gjx_site& s = pl.sites[j];
      const RollInfo& ri = pl.info[j];
      if (!(s.flags & GJX_SITE_HMC_SELECTED) || s.slot < 0) continue;
      pl.leaf_of_site[j] = leaf++;
      if (!ri.plate) continue;
      HmcPlan::PlateSel ps{j, s.slot, s.dim, ri.row, ri.d_row, ri.plate_n, pl.prows, pl.leaf_of_site[j], pl.nsel};
      for (int d = 0; d < s.dim; ++d) { pl.sel_of_slot[s.slot + d] = pl.nsel++; pl.slot_of_sel.push_back(s.slot + d); }
      pl.prows += ri.plate_n * s.dim;
      pl.psel.push_back(ps);
    }
  }
  if (pl.nsel < 1) return false;
  if (pl.nsel > kHmcMaxSel || unrolled > 256 || p->n_tab > kHmcMaxTab) { if (pl.rolled) return false; pl.big = true; }   // (a table beyond the LDS: read from memory)
  if (pl.big) {
    // LDS columns of a 64-lane block: v, g, p (+ g0 for the stale-carry compatibility mode when it still fits 152 KB)
    if (unrolled > 1024 || getenv("GJX_HMC_GEN_NO_BIG")) return false;
    const int need3 = pl.n_regs + 2 * pl.nsel, need4 = need3 + pl.nsel;
    if (need3 * 256 > 152 * 1024) return false;
    pl.nostale = need4 * 256 > 152 * 1024;
    pl.block = 64;
  }
  pl.mf_k.assign(ns, -1);
  pl.xt_off.assign(ns, 0);
  pl.fold.assign(ns, 0);
  pl.bs_off.assign(ns, 0);
  if (pl.looped && !pl.rolled && !pl.big && !getenv("GJX_HMC_GEN_NO_MFMA")) {
    const int tab_pad = (p->n_tab + 3) & ~3;
    for (int j = 0; j < ns; ++j) {
      const gjx_site& s = pl.sites[j];
      const int k = pl.info[j].plate ? -1 : hmc_mfma_param(s);
      if (k < 0) continue;
      const bool fold = !pl.plates && hmc_fold_ok(p, j, k);
      const int need = s.p[k].n * (s.dim + 4) + (fold ? s.dim : 0);
      if (tab_pad + pl.xt_floats + need > 40000) continue;           // 160 KB of LDS: table + transposed matrices (+ nothing else)
      pl.mf_k[j] = k;
      pl.xt_off[j] = pl.xt_floats;
      pl.fold[j] = fold ? 1 : 0;
      pl.bs_off[j] = pl.xt_floats + s.p[k].n * (s.dim + 4);
      pl.xt_floats += need;
      pl.mfma = true;
    }
    if (pl.mfma) {
      // the LDS footprint decides how many blocks a CU holds; the block is sized so that a CU still runs 8 waves — two per
      // SIMD with 256 registers each: chain state (4 NSEL + NS values) and four row tiles of operands do not fit 128
      const int bytes = 4 * (tab_pad + pl.xt_floats), per_cu = bytes > 0 ? (160 * 1024) / bytes : 8;
      pl.block = per_cu >= 2 ? 256 : 512;
      if (const char* e = getenv("GJX_HMC_GEN_BT")) { const int b = atoi(e); if (b == 256 || b == 512 || b == 1024) pl.block = b; }
    }
  }
  *out = pl;
  return true;
}

// the straight-line / plate forms first (chain state in registers); a program they do not fit — a long Scan — as a rolled loop
bool hmc_plan(const gjx_program* p, HmcPlan* out) { return hmc_plan_form(p, out, false) || hmc_plan_form(p, out, true); }

// one element of site j: parameters, (score,) gradient terms.  dx: element index expression (a literal, or "d_" in a rolled
// site); acc: name of the gradient accumulator array indexed by SELECTED-slot index ("g" or the site's partial "ga")
// mf (matrix-core sites): parameter mf->k arrives finished in `pre` (bias + contraction, the tile's C layout) and its
// gradient weight leaves through `wout` for the backward contraction; `x` (optional) names the element's value
struct HmcMf { int k; std::string pre, wout, x; };

void hmc_emit_element(Emit& o, const gjx_program* prog, const HmcPlan& hp, int j, const std::string& dx, bool dyn, const char* acc,
                      const char* sc, const char* ind, const HmcMf* mf = nullptr) {
  const gjx_site& s = prog->sites[j];
  const RollInfo& ri = hp.info[j];
  g_loop_var = hp.rolled ? "(t_ - 1)" : "i_";   // (a plate's body site: table offsets advance with the instance; a rolled Scan's: with the step)
  const int np = n_params(s.kind);
  o.f("%s{\n", ind);
  for (int k = 0; k < 4; ++k) {
    if (k >= np) { o.f("%s  const float par_%d = 0.0f;\n", ind, k); continue; }
    const gjx_param& q = s.p[k];
    const std::string e = q.len == 1 ? "0" : ("(" + dx + ") % " + std::to_string(q.len));
    if (mf && k == mf->k) {
      o.f("%s  const float pre_%d = %s;\n%s  const float par_%d = %s;\n", ind, k, mf->pre.c_str(), ind, k, xf_wrap(q.xf, "pre_" + std::to_string(k)).c_str());
      continue;
    }
    switch (q.op) {
      case GJX_P_CONST: o.f("%s  const float pre_%d = TAB(%s + %s);\n", ind, k, toff(q.off, ri.d_off[k]).c_str(), e.c_str()); break;
      case GJX_P_GATHER: o.f("%s  const float pre_%d = TAB(%s + gi_%d_%d * %d + %s);\n", ind, k, toff(q.off, ri.d_off[k]).c_str(), j, k, q.len, e.c_str()); break;
      case GJX_P_VALUE:
        if (q.len == 1) o.f("%s  const float pre_%d = v[%d];\n", ind, k, q.slot);
        else o.f("%s  const float pre_%d = v[%d + %s];\n", ind, k, q.slot, e.c_str());   // (unrolled sites only: literal index)
        break;
      case GJX_P_EXPR: {      // the block's nodes (kept: the reverse sweep below reads them)
        const int out = q.n - q.len + (q.len == 1 ? 0 : literal_index(dx) % q.len);
        const std::string pfx = "en_" + std::to_string(k) + "_";
        emit_expr_nodes(o, g_expr_prog, q, out, pfx, (std::string(ind) + "  ").c_str(),
                        [&q, &ri, k](int node, int t) { return "v[" + std::to_string(expr_leaf_reg(g_expr_prog, q, ri, k, node, t)) + "]"; }, &ri.enode_dtab[k]);
        o.f("%s  const float pre_%d = %s%d;\n", ind, k, pfx.c_str(), out);
        break;
      }
      case GJX_P_VGATHER: {   // row gi_ of a choice in registers: n - 1 selects
        const int e0 = q.len == 1 ? 0 : literal_index(dx) % q.len;
        o.f("%s  float pre_%d = v[%d];\n", ind, k, q.moff + e0);
        for (int c = 1; c < q.n; ++c) o.f("%s  pre_%d = gi_%d_%d == %d ? v[%d] : pre_%d;\n", ind, k, j, k, c, q.moff + c * q.len + e0, k);
        break;
      }
      default: {   // AFFINE: the row stays in registers for the gradient
        o.f("%s  float row_%d[%d];\n", ind, k, q.n);
        if (q.moff % 4 == 0 && q.n % 4 == 0 && ri.d_moff[k] % 4 == 0)
          o.f("%s  { const float4* r4_ = (const float4*)&TAB(%s + (%s) * %d); _Pragma(\"unroll\") for (int e_ = 0; e_ < %d; ++e_) { const float4 t_ = r4_[e_]; "
              "row_%d[4 * e_] = t_.x; row_%d[4 * e_ + 1] = t_.y; row_%d[4 * e_ + 2] = t_.z; row_%d[4 * e_ + 3] = t_.w; } }\n",
              ind, toff(q.moff, ri.d_moff[k]).c_str(), dx.c_str(), q.n, q.n / 4, k, k, k, k);
        else
          o.f("%s  _Pragma(\"unroll\") for (int e_ = 0; e_ < %d; ++e_) row_%d[e_] = TAB(%s + (%s) * %d + e_);\n", ind, q.n, k, toff(q.moff, ri.d_moff[k]).c_str(), dx.c_str(), q.n);
        o.f("%s  float pre_%d = TAB(%s + %s);\n", ind, k, toff(q.off, ri.d_off[k]).c_str(), e.c_str());
        o.f("%s  _Pragma(\"unroll\") for (int e_ = 0; e_ < %d; ++e_) pre_%d = fmaf(row_%d[e_], v[%d + e_], pre_%d);\n", ind, q.n, k, k, q.slot, k);
      }
    }
    o.f("%s  const float par_%d = %s;\n", ind, k, xf_wrap(q.xf, "pre_" + std::to_string(k)).c_str());
  }
  if (mf && !mf->x.empty()) o.f("%s  const float x_ = %s;\n", ind, mf->x.c_str());
  else if (s.slot >= 0) o.f("%s  const float x_ = v[%d + %s];\n", ind, s.slot, dx.c_str());
  else o.f("%s  const float x_ = TAB(%s + %s);\n", ind, toff(s.obs_off, ri.d_obs).c_str(), dx.c_str());
  o.f("%s  if (SC) %s += elem_logpdf(%d, x_, par_0, par_1, par_2, par_3);\n", ind, sc, s.kind);
  o.f("%s  float gx_, gp_[4];\n%s  dlogpdf(%d, x_, par_0, par_1, par_2, par_3, gx_, gp_);\n", ind, ind, s.kind);
  if (s.slot >= 0 && !dyn) {
    const int m = hp.sel_of_slot[s.slot + atoi(dx.c_str())];
    if (m >= 0) o.f("%s  %s[%d] += gx_;\n", ind, acc, m);
  }
  for (int k = 0; k < np; ++k) {
    const gjx_param& q = s.p[k];
    if (q.op != GJX_P_VALUE && q.op != GJX_P_AFFINE && q.op != GJX_P_VGATHER && q.op != GJX_P_EXPR) continue;
    const std::string w = q.xf == GJX_XF_NONE ? "gp_[" + std::to_string(k) + "]"
                                              : "(gp_[" + std::to_string(k) + "] * xf_deriv(" + std::to_string(q.xf) + ", pre_" + std::to_string(k) + "))";
    if (q.op == GJX_P_EXPR) {
      // reverse sweep through the block (hmc.py:70-96: selection_gradient differentiates whatever the body computes): one adjoint
      // per node on a path from a SELECTED leaf to the output, everything else is never emitted
      const int out = q.n - q.len + (q.len == 1 ? 0 : literal_index(dx) % q.len);
      const std::string pfx = "en_" + std::to_string(k) + "_", ad = "ad_" + std::to_string(k) + "_";
      std::vector<char> need(q.n, 0), live(q.n, 0);     // need: feeds the output; live: depends on a selected value
      need[out] = 1;
      for (int i = out; i >= 0; --i) {
        if (!need[i]) continue;
        const ExprNode e = expr_node(g_expr_prog, q, i);
        switch (e.op) {
          case GJX_E_CONST: case GJX_E_VALUE: case GJX_E_LINV: break;
          case GJX_E_ADD: case GJX_E_SUB: case GJX_E_MUL: case GJX_E_DIV: case GJX_E_MAX: case GJX_E_MIN: case GJX_E_GT: need[e.a] = need[e.b] = 1; break;
          case GJX_E_WHERE: need[e.a] = need[e.b] = need[e.c] = 1; break;
          case GJX_E_LINN: for (int t = 0; t < e.c; ++t) need[e.b + t] = 1; break;
          default: need[e.a] = 1; break;
        }
      }
      for (int i = 0; i <= out; ++i) {
        if (!need[i]) continue;
        const ExprNode e = expr_node(g_expr_prog, q, i);
        switch (e.op) {
          case GJX_E_CONST: case GJX_E_GT: break;
          case GJX_E_VALUE: live[i] = hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, 0)] >= 0; break;
          case GJX_E_LINV: for (int t = 0; t < e.c; ++t) live[i] = live[i] || hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, t)] >= 0; break;
          case GJX_E_ADD: case GJX_E_SUB: case GJX_E_MUL: case GJX_E_DIV: case GJX_E_MAX: case GJX_E_MIN: live[i] = live[e.a] || live[e.b]; break;
          case GJX_E_WHERE: live[i] = live[e.b] || live[e.c]; break;
          case GJX_E_LINN: for (int t = 0; t < e.c; ++t) live[i] = live[i] || live[e.b + t]; break;
          default: live[i] = live[e.a]; break;
        }
      }
      if (!live[out]) continue;
      auto N = [&](int i) { return pfx + std::to_string(i); };
      auto AD = [&](int i) { return ad + std::to_string(i); };
      o.f("%s  {\n", ind);
      for (int i = 0; i <= out; ++i) if (need[i] && live[i]) o.f("%s    float %s = %s;\n", ind, AD(i).c_str(), i == out ? w.c_str() : "0.0f");
      for (int i = out; i >= 0; --i) {
        if (!need[i] || !live[i]) continue;
        ExprNode e = expr_node(g_expr_prog, q, i);
        if (!ri.enode_dtab[k].empty()) e.da = ri.enode_dtab[k][i];
        const std::string gi = AD(i);
        auto add = [&](int to, const std::string& term) { if (live[to]) o.f("%s    %s += %s;\n", ind, AD(to).c_str(), term.c_str()); };
        switch (e.op) {
          case GJX_E_VALUE: o.f("%s    %s[%d] += %s;\n", ind, acc, hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, 0)], gi.c_str()); break;
          case GJX_E_LINV:
            for (int t = 0; t < e.c; ++t) {
              const int m_ = hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, t)];
              if (m_ >= 0) o.f("%s    %s[%d] = fmaf(%s, TAB(%s + %d), %s[%d]);\n", ind, acc, m_, gi.c_str(), toff(e.a, e.da).c_str(), 1 + t, acc, m_);
            }
            break;
          case GJX_E_ADD: add(e.a, gi); add(e.b, gi); break;
          case GJX_E_SUB: add(e.a, gi); add(e.b, "-" + gi); break;
          case GJX_E_MUL: add(e.a, gi + " * " + N(e.b)); add(e.b, gi + " * " + N(e.a)); break;
          case GJX_E_DIV: add(e.a, gi + " * fast_rcp(" + N(e.b) + ")"); add(e.b, "-" + gi + " * " + N(i) + " * fast_rcp(" + N(e.b) + ")"); break;
          case GJX_E_MAX: add(e.a, N(e.a) + " >= " + N(e.b) + " ? " + gi + " : 0.0f"); add(e.b, N(e.a) + " >= " + N(e.b) + " ? 0.0f : " + gi); break;
          case GJX_E_MIN: add(e.a, N(e.a) + " <= " + N(e.b) + " ? " + gi + " : 0.0f"); add(e.b, N(e.a) + " <= " + N(e.b) + " ? 0.0f : " + gi); break;
          case GJX_E_WHERE: add(e.b, N(e.a) + " != 0.0f ? " + gi + " : 0.0f"); add(e.c, N(e.a) + " != 0.0f ? 0.0f : " + gi); break;
          case GJX_E_LINN: for (int t = 0; t < e.c; ++t) add(e.b + t, gi + " * TAB(" + toff(e.a, e.da) + " + " + std::to_string(1 + t) + ")"); break;
          case GJX_E_CONST: case GJX_E_GT: break;
          default: add(e.a, gi + " * expr_unary_deriv(" + std::to_string(e.op) + ", " + N(e.a) + ", " + N(i) + ")"); break;
        }
      }
      o.f("%s  }\n", ind);
      continue;
    }
    if (mf && k == mf->k) {
      o.f("%s  %s = %s;\n", ind, mf->wout.c_str(), w.c_str());
    } else if (q.op == GJX_P_VGATHER) {   // the gradient goes to the picked row
      const int e0 = q.len == 1 ? 0 : literal_index(dx) % q.len;
      bool any = false;
      for (int c = 0; c < q.n; ++c) any = any || hp.sel_of_slot[q.moff + c * q.len + e0] >= 0;
      if (!any) continue;
      o.f("%s  { const float w_ = %s;\n", ind, w.c_str());
      for (int c = 0; c < q.n; ++c) {
        const int m = hp.sel_of_slot[q.moff + c * q.len + e0];
        if (m >= 0) o.f("%s    %s[%d] += gi_%d_%d == %d ? w_ : 0.0f;\n", ind, acc, m, j, k, c);
      }
      o.f("%s  }\n", ind);
    } else if (q.op == GJX_P_VALUE) {
      const int src = q.slot + (q.len == 1 ? 0 : atoi(dx.c_str()) % q.len);
      const int m = hp.sel_of_slot[src];
      if (m >= 0) o.f("%s  %s[%d] += %s;\n", ind, acc, m, w.c_str());
    } else {
      bool any = false;
      for (int e = 0; e < q.n; ++e) any = any || hp.sel_of_slot[q.slot + e] >= 0;
      if (!any) continue;
      o.f("%s  { const float w_ = %s;\n", ind, w.c_str());
      for (int e = 0; e < q.n; ++e) {
        const int m = hp.sel_of_slot[q.slot + e];
        if (m >= 0) o.f("%s    %s[%d] = fmaf(w_, row_%d[%d], %s[%d]);\n", ind, acc, m, k, e, acc, m);
      }
      o.f("%s  }\n", ind);
    }
  }
  o.f("%s}\n", ind);
}

// A rolled site with an AFFINE parameter on v_mfma_f32_16x16x4_f32 (the layout of the hand-written k_hmc_logreg_mfma2,
// gjx_hmc.hip, for any such site): a wave holds 16 chains, lane l = (c = l & 15, q = l >> 4) is one of the four lanes of chain c.
//   forward   pre[n0 + 4 q + r][c] = bias + sum_k X[n0 + ..][k] v[k][c]:  A = X[n0 + c][16 b + 4 q + s] (one b128 LDS read of the
//             row-major table per four instructions), B = v[16 b + 4 q + s] of chain c (registers: every lane keeps its chain's values);
//   the element log-densities' derivatives are taken in the result layout — 4 elements per lane, no lane idle or redundant —
//   backward  d/dv[16 b + 4 q + r][c] = sum_rows X[row][..] w[row][c]:  A = XT[16 b + c][n0 + 4 q + r] (one b128 read of the
//             transposed copy made in the kernel's prologue), B = w of row n0 + 4 q + r, which is what this lane just computed;
//   after the site the four lanes of a chain exchange their quarters of the gradient (n lane permutes).
void hmc_emit_mfma_site(Emit& o, const gjx_program* prog, const HmcPlan& hp, int j) {
  const gjx_site& s = prog->sites[j];
  const int k = hp.mf_k[j], np = n_params(s.kind);
  const gjx_param& q = s.p[k];
  const int n = q.n, NB = n / 16, dim = s.dim, LD = dim + 4;
  const int TB = dim % 64 == 0 ? 4 : (dim % 32 == 0 ? 2 : 1);
  const bool fold = hp.fold[j] != 0;
  bool any_in = false;
  for (int e = 0; e < n; ++e) any_in = any_in || hp.sel_of_slot[q.slot + e] >= 0;
  o.f("    // %d rows x %d inputs on the matrix cores, %d row tiles per trip%s\n", dim, n, TB,
      fold ? "; rows and bias carry the observation's sign and log2 e (folded in the prologue)" : "");
  o.f("    float bq_[%d][4];\n", NB);
  for (int b = 0; b < NB; ++b)
    for (int st = 0; st < 4; ++st) {
      const int v0 = q.slot + 16 * b + st;
      o.f("    bq_[%d][%d] = q_ == 0 ? v[%d] : (q_ == 1 ? v[%d] : (q_ == 2 ? v[%d] : v[%d]));\n", b, st, v0, v0 + 4, v0 + 8, v0 + 12);
    }
  o.f("    v4f_ gacc_[%d][%d];\n    _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int b_ = 0; b_ < %d; ++b_) gacc_[t_][b_] = v4f_{0.0f, 0.0f, 0.0f, 0.0f};\n",
      TB, NB, TB, NB);
  o.f("    float ga[NSEL];\n    _Pragma(\"unroll\") for (int m_ = 0; m_ < NSEL; ++m_) ga[m_] = 0.0f;\n    float scp_ = 0.0f;\n");
  o.f("    const float* xf_ = &TAB(%d) + c16_ * %d + 4 * q_;\n    const float* xb_ = xt_s + %d + c16_ * %d + 4 * q_;\n", q.moff, n, hp.xt_off[j], LD);
  o.f("    _Pragma(\"nounroll\") for (int n0_ = 0; n0_ < %d; n0_ += %d) {\n", dim, 16 * TB);
  o.f("      v4f_ s_[%d], xa_[%d][%d];\n", TB, TB, NB);
  o.f("      _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) {\n", TB);
  if (fold) o.f("        s_[t_] = *(const v4f_*)(xt_s + %d + n0_ + 16 * t_ + 4 * q_);\n", hp.bs_off[j]);
  else if (q.len == 1) o.f("        { const float b0_ = TAB(%d); s_[t_] = v4f_{b0_, b0_, b0_, b0_}; }\n", q.off);
  else if ((q.off & 3) == 0) o.f("        s_[t_] = *(const v4f_*)&TAB(%d + n0_ + 16 * t_ + 4 * q_);\n", q.off);
  else o.f("        _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) s_[t_][r_] = TAB(%d + n0_ + 16 * t_ + 4 * q_ + r_);\n", q.off);
  o.f("        _Pragma(\"unroll\") for (int b_ = 0; b_ < %d; ++b_) xa_[t_][b_] = *(const v4f_*)(xf_ + (n0_ + 16 * t_) * %d + 16 * b_);\n      }\n", NB, n);
  o.f("      __builtin_amdgcn_sched_barrier(0);\n");
  o.f("      _Pragma(\"unroll\") for (int b_ = 0; b_ < %d; ++b_) _Pragma(\"unroll\") for (int st_ = 0; st_ < 4; ++st_) _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_)\n"
      "        s_[t_] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa_[t_][b_][st_], bq_[b_][st_], s_[t_], 0, 0, 0);\n      __builtin_amdgcn_sched_barrier(0);\n", NB, TB);
  // the transposed rows of the backward phase are fetched here, behind the forward instructions
  if (any_in) o.f("      v4f_ xc_[%d][%d];\n      _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int b_ = 0; b_ < %d; ++b_) xc_[t_][b_] = *(const v4f_*)(xb_ + b_ * %d + n0_ + 16 * t_);\n",
                  TB, NB, TB, NB, 16 * LD);
  const bool y128 = s.slot < 0 && (s.obs_off & 3) == 0;
  if (fold) {
    // s_ = log2 e * s a: log p = -ln 2 * log2(1 + 2^(-s_)) (accumulated in log2 units), d log p / d(s a) = 1 / (1 + 2^(s_))
    o.f("      if (SC) {\n        _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) {\n"
        "          const float m_ = -s_[t_][r_];\n          scp_ += fmaxf(m_, 0.0f) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(m_)));\n        }\n      }\n", TB);
    o.f("      _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) s_[t_][r_] = __builtin_amdgcn_exp2f(s_[t_][r_]);\n"
        "      _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) s_[t_][r_] = 1.0f + s_[t_][r_];\n"
        "      _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) s_[t_][r_] = fast_rcp(s_[t_][r_]);\n"
        "      __builtin_amdgcn_sched_barrier(0);\n", TB, TB, TB);
  } else {
    o.f("      _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) {\n", TB);
    if (y128) o.f("        const v4f_ y4_ = *(const v4f_*)&TAB(%d + n0_ + 16 * t_ + 4 * q_);\n", s.obs_off);
    o.f("        _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) {\n          const int d_ = n0_ + 16 * t_ + 4 * q_ + r_; (void)d_;\n");
    const HmcMf mf{k, "s_[t_][r_]", "s_[t_][r_]", y128 ? "y4_[r_]" : ""};
    hmc_emit_element(o, prog, hp, j, "d_", true, "ga", "scp_", "          ", &mf);
    o.f("        }\n      }\n      __builtin_amdgcn_sched_barrier(0);\n");
  }
  if (any_in)
    o.f("      _Pragma(\"unroll\") for (int r_ = 0; r_ < 4; ++r_) _Pragma(\"unroll\") for (int t_ = 0; t_ < %d; ++t_) _Pragma(\"unroll\") for (int b_ = 0; b_ < %d; ++b_)\n"
        "        gacc_[t_][b_] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc_[t_][b_][r_], s_[t_][r_], gacc_[t_][b_], 0, 0, 0);\n      __builtin_amdgcn_sched_barrier(0);\n", TB, NB);
  o.f("    }\n");
  // the lane (c, qq) holds d/dv[16 b + 4 qq + r] of chain c in gacc_[.][b][r]: every lane of the chain fetches all of them
  if (any_in) {
    o.f("    _Pragma(\"unroll\") for (int t_ = 1; t_ < %d; ++t_) _Pragma(\"unroll\") for (int b_ = 0; b_ < %d; ++b_) gacc_[0][b_] += gacc_[t_][b_];\n", TB, NB);
    for (int e = 0; e < n; ++e) {
      const int m = hp.sel_of_slot[q.slot + e];
      if (m >= 0) o.f("    g[%d] += %s__shfl(gacc_[0][%d][%d], c16_ + %d, 64);\n", m, fold ? "kLn2 * " : "", e / 16, e % 4, 16 * ((e % 16) / 4));
    }
  }
  std::vector<char> touched(hp.nsel, 0);
  for (int kk = 0; kk < np; ++kk) {
    const gjx_param& qq = s.p[kk];
    if (kk != k && qq.op == GJX_P_VALUE && hp.sel_of_slot[qq.slot] >= 0) touched[hp.sel_of_slot[qq.slot]] = 1;
  }
  for (int m = 0; m < hp.nsel; ++m) if (touched[m] && !fold) o.f("    g[%d] += QSUM(ga[%d]);\n", m, m);
  o.f("    if (SC) sc_ += %sQSUM(scp_);\n", fold ? "-kLn2 * " : "");
}

// cpl_code: lanes per chain of a kernel with loops (plates, rolled sites): 0 = 4; 16 or 64 for FEW chains over LONG loops — the loop's
// trips are dealt to that many lanes, every lane keeps a copy of the chain's register state (hmc_gen_launch picks it by the number of
// chains; matrix-core kernels have their own layout)
std::string generate_hmc(const gjx_program* prog_in, int cpl_code = 0) {
  HmcPlan hp;
  if (!hmc_plan(prog_in, &hp)) return "";
  g_expr_prog = prog_in;
  // the emitted program: the plan's site list (plates: every slot a register, a body site owns one instance's worth)
  gjx_program eprog = *prog_in;
  eprog.sites = hp.sites.data();
  eprog.n_sites = (int)hp.sites.size();
  eprog.n_slots = hp.n_regs;
  const gjx_program* prog = &eprog;
  int longest = hp.rolled ? hp.roll.T : 0;       // the longest loop of the kernel: what more lanes per chain can share
  for (int j = 0; j < prog->n_sites; ++j) {
    const gjx_site& s = hp.sites[j];
    const int len = hp.info[j].plate ? hp.info[j].plate_n : (!is_categorical(s.kind) && s.dim > kMaxExpandDim ? s.dim : 0);
    longest = len > longest ? len : longest;
  }
  const int cpl_max = (!hp.looped || hp.mfma) ? (hp.looped ? 4 : 1) : (longest >= 256 ? 64 : (longest >= 64 ? 16 : 4));
  const int cpl = !hp.looped ? 1 : ((cpl_code == 16 || cpl_code == 64) && cpl_code <= cpl_max ? cpl_code : 4);
  const int NS = prog->n_slots, NSEL = hp.nsel;
  Emit o;
  o.f("#include \"gjx_device.h\"\nusing namespace gjx;\n#define RNG %d\n#define CPL %d\n#define NS %d\n#define NSEL %d\n#define NTAB %d\n",
      prog->rng_mode == GJX_RNG_JAX32 ? GJX_RNG_JAX32 : GJX_RNG_FLAT, cpl, NS, NSEL, prog->n_tab);
  o.f("#define TAB(i) tab_s[i]\n#define BT %d\n#define XTF %d\ntypedef float v4f_ __attribute__((ext_vector_type(4)));\n", hp.block, hp.xt_floats);
  // NOUT: selected values outside plates (chain state in registers); PROWS: rows of the selected sites inside plates (state in the workspace)
  o.f("#define NOUT %d\n#define PROWS %d\n", hp.nout, hp.prows);
  // the lanes of a chain: an aligned quad, or with a matrix-core site the four 16-lane rows of the wave (lane & 15 = chain)
  if (hp.mfma) o.f("GJX_DEV float QSUM(float x) { x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64); return x; }\n");
  else if (cpl <= 4) o.f("#define QSUM(x) quad_sum(x)\n");
  else o.f("GJX_DEV float QSUM(float x) { _Pragma(\"unroll\") for (int o_ = 1; o_ < CPL; o_ <<= 1) x += __shfl_xor(x, o_, 64); return x; }\n");
  // ---- the sweep: score (SC) and gradient of the selected slots.  ch_ / n_ / ic_: the chain's column of choices[][] — a plate's
  //      body sites with per-chain values (OBS_SLOT) read their instance's rows from there, sweep after sweep
  // (the chain state as register arrays, or — HmcPlan::big — as LDS columns behind the same indexing syntax)
  o.f("struct LdsCol { float* b; GJX_DEV float& operator[](int i) const { return b[i * BT]; } };\n"
      "template <bool SC, class VA, class GA>\nGJX_DEV float sweep(VA& v, GA& g, const float* __restrict__ tab_s, const float* __restrict__ xt_s, const int q_,\n"
      "                  const float* __restrict__ ch_, const int64_t n_, const int64_t ic_, const float* wq_, float* wg_, const bool live_) {\n"
      "  float sc_ = 0.0f;\n  const int c16_ = (int)(threadIdx.x & 15u); (void)c16_; (void)xt_s; (void)ch_; (void)n_; (void)ic_; (void)wq_; (void)wg_; (void)live_;\n"
      "  _Pragma(\"unroll\") for (int m_ = 0; m_ < NSEL; ++m_) g[m_] = 0.0f;\n");
  // one site: `acc` / `sc` name the gradient and score accumulators (a plate's body adds to per-lane partials)
  auto emit_one = [&](int j, const char* acc, const char* sc) {
    const gjx_site& s = prog->sites[j];
    const RollInfo& ri = hp.info[j];
    g_loop_var = hp.rolled ? "(t_ - 1)" : "i_";
    if (s.mode == GJX_MODE_INPUT) { o.f("  // ---- site %d: INPUT, %d rows from slot %d (a value, no density)\n", j, s.dim, s.slot); return; }
    const int np = n_params(s.kind);
    o.f("  { // ---- site %d: kind %d, dim %d, slot %d%s\n", j, s.kind, is_categorical(s.kind) ? s.ncat : s.dim, s.slot, ri.plate ? " (plate body)" : "");
    for (int k = 0; k < np; ++k) {
      const gjx_param& q = s.p[k];
      if (q.op == GJX_P_GATHER || (q.op == GJX_P_VGATHER && q.slot >= 0))
        o.f("    int gi_%d_%d; { const int g_ = (int)v[%d]; gi_%d_%d = g_ < 0 ? 0 : (g_ > %d ? %d : g_); }\n", j, k, q.slot, j, k, q.n - 1, q.n - 1);
      else if (q.op == GJX_P_VGATHER)     // the index choice is constrained to one value for every chain: read from the table
        o.f("    int gi_%d_%d; { const int g_ = (int)TAB(%s); gi_%d_%d = g_ < 0 ? 0 : (g_ > %d ? %d : g_); }\n", j, k, toff(q.off, ri.d_off[k]).c_str(), j, k, q.n - 1, q.n - 1);
    }
    if (is_categorical(s.kind)) {   // an integer site: scored, no gradient through it (hmc.py:49-65; k_hmc_generic)
      const gjx_param& q = s.p[0];
      const bool probs = s.kind == GJX_CATEGORICAL_PROBS;
      std::string L = "TAB(" + toff(q.off, ri.d_off[0]) + (q.op == GJX_P_GATHER ? " + gi_" + std::to_string(j) + "_0 * " + std::to_string(q.len) : "") +
                      " + (c_) % " + std::to_string(q.len) + ")";
      L = xf_wrap(q.xf, L);
      if (probs) L = "safe_log(" + L + ")";
      o.f("    if (SC) {\n      float mx = -INFINITY;\n      for (int c_ = 0; c_ < %d; ++c_) mx = fmaxf(mx, %s);\n", s.ncat, L.c_str());
      o.f("      float se = 0.0f;\n      for (int c_ = 0; c_ < %d; ++c_) se += fast_exp(%s - mx);\n", s.ncat, L.c_str());
      if (s.slot >= 0) o.f("      const float val_ = v[%d];\n", s.slot);
      else o.f("      const float val_ = TAB(%s);\n", toff(s.obs_off, ri.d_obs).c_str());
      o.f("      int k_ = (int)val_;\n      k_ = k_ < 0 ? 0 : (k_ > %d ? %d : k_);\n      { const int c_ = k_; %s += %s - (mx + fast_log(se)); }\n    }\n",
          s.ncat - 1, s.ncat - 1, sc, L.c_str());
    } else if (s.dim <= kMaxExpandDim) {
      for (int d = 0; d < s.dim; ++d) hmc_emit_element(o, prog, hp, j, std::to_string(d), false, acc, sc, "    ");
    } else if (hp.mf_k[j] >= 0) {
      hmc_emit_mfma_site(o, prog, hp, j);
    } else {
      // a rolled site: elements dealt round-robin to the CPL lanes of the chain, partial score and gradient rows joined at the end
      o.f("    float ga[NSEL];\n    _Pragma(\"unroll\") for (int m_ = 0; m_ < NSEL; ++m_) ga[m_] = 0.0f;\n    float scp_ = 0.0f;\n");
      o.f("    _Pragma(\"unroll 2\") for (int d_ = q_; d_ < %d; d_ += CPL) {\n", s.dim);
      hmc_emit_element(o, prog, hp, j, "d_", true, "ga", "scp_", "      ");
      o.f("    }\n");
      // which accumulators the site touches
      std::vector<char> touched(NSEL, 0);
      for (int k = 0; k < np; ++k) {
        const gjx_param& q = s.p[k];
        if (q.op == GJX_P_VALUE && hp.sel_of_slot[q.slot] >= 0) touched[hp.sel_of_slot[q.slot]] = 1;
        if (q.op == GJX_P_AFFINE) for (int e = 0; e < q.n; ++e) if (hp.sel_of_slot[q.slot + e] >= 0) touched[hp.sel_of_slot[q.slot + e]] = 1;
        if (q.op == GJX_P_VGATHER) for (int e = 0; e < q.n * q.len; ++e) if (hp.sel_of_slot[q.moff + e] >= 0) touched[hp.sel_of_slot[q.moff + e]] = 1;
        if (q.op == GJX_P_EXPR)
          for (int i = 0; i < q.n; ++i) {
            const ExprNode e = expr_node(g_expr_prog, q, i);
            if (e.op == GJX_E_VALUE && hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, 0)] >= 0) touched[hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, 0)]] = 1;
            if (e.op == GJX_E_LINV) for (int t = 0; t < e.c; ++t) { const int m_ = hp.sel_of_slot[expr_leaf_reg(g_expr_prog, q, ri, k, i, t)]; if (m_ >= 0) touched[m_] = 1; }
          }
      }
      for (int m = 0; m < NSEL; ++m) if (touched[m]) o.f("    g[%d] += CPL > 1 ? QSUM(ga[%d]) : ga[%d];\n", m, m, m);
      o.f("    if (SC) sc_ += CPL > 1 ? QSUM(scp_) : scp_;\n");
    }
    o.f("  }\n");
  };
  if (hp.rolled) {
    // ---- a rolled Scan (scan.py:237-294 differentiated by hmc.py:70-96): the T steps are dealt to the CPL lanes of the chain in
    //      contiguous chunks [a_, b_).  A lane walks its chunk with the previous and the current step's values in registers; a term of
    //      step t contributes to the gradient rows of steps t - 1 and t, so the lane carries step t's own part (gcar_) into step
    //      t + 1's iteration, where the row is complete and stored — and runs ONE extra iteration, t = b_, for its last row only
    //      (score and outside gradients of that iteration belong to the next lane and are dropped)
    const Roll& r = hp.roll;
    const int SS = hp.ssel, S = r.S, np0 = r.n_pre;
    for (int j = 0; j < r.i0; ++j) emit_one(j, "g", "sc_");        // the sites in front of the Scan: every lane, registers
    o.f("  { // ---- rolled Scan: %d steps x %d sites, %d selected values per step\n"
        "    float ga[NSEL];\n    _Pragma(\"unroll\") for (int m_ = 0; m_ < NSEL; ++m_) ga[m_] = 0.0f;\n    float scp_ = 0.0f, gcar_[%d];\n"
        "    _Pragma(\"unroll\") for (int k_ = 0; k_ < %d; ++k_) gcar_[k_] = 0.0f;\n"
        "    const int L_ = (%d + CPL - 1) / CPL, a_ = q_ * L_, b_ = a_ + L_ < %d ? a_ + L_ : %d;\n"
        "    if (a_ < %d) {\n", r.T, r.m, SS, SS > 0 ? SS : 1, SS, r.T, r.T, r.T, r.T);
    // loads of one step's values into the CURRENT step's registers: selected -> the trajectory's position (workspace), else the chain's column
    auto load_step = [&](const char* t, const char* ind) {
      for (int l = 0; l < r.m; ++l) {
        const gjx_site& sl = prog->sites[r.i0 + r.m + l];
        if (sl.slot < 0) continue;
        const int w = is_categorical(sl.kind) ? 1 : sl.dim;
        for (int d = 0; d < w; ++d) {
          const int ls = sl.slot + d - (np0 + S), m = hp.sel_of_slot[sl.slot + d];
          if (m >= 0) o.f("%sv[%d] = wq_[((int64_t)(%s) * %d + %d) * n_ + ic_];\n", ind, sl.slot + d, t, SS, m - hp.nout - SS);
          else o.f("%sv[%d] = ch_[((int64_t)%d + (int64_t)(%s) * %d + %d) * n_ + ic_];\n", ind, sl.slot + d, np0, t, S, ls);
        }
      }
    };
    o.f("    int t0_;\n    if (q_ == 0) {   // step 0 (its sites' own parameters: the initial carry)\n");
    load_step("0", "      ");
    for (int l = 0; l < r.m; ++l) emit_one(r.i0 + l, "ga", "scp_");
    o.f("      _Pragma(\"unroll\") for (int k_ = 0; k_ < %d; ++k_) gcar_[k_] = ga[NOUT + %d + k_];\n      t0_ = 1;\n    } else {\n", SS, SS);
    load_step("a_ - 1", "      ");
    o.f("      t0_ = a_;\n    }\n"
        "    _Pragma(\"nounroll\") for (int t_ = t0_; t_ <= b_ && t_ < %d; ++t_) {\n"
        "      const bool real_ = t_ < b_;\n", r.T);
    for (int ls = 0; ls < S; ++ls) o.f("      v[%d] = v[%d];\n", np0 + ls, np0 + S + ls);      // the current step becomes the previous one
    load_step("t_", "      ");
    o.f("      float sv_[NOUT > 0 ? NOUT : 1]; const float svs_ = scp_;\n      _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) sv_[m_] = ga[m_];\n"
        "      _Pragma(\"unroll\") for (int m_ = NOUT; m_ < NSEL; ++m_) ga[m_] = 0.0f;\n");
    for (int l = 0; l < r.m; ++l) emit_one(r.i0 + r.m + l, "ga", "scp_");
    o.f("      if (!real_) { scp_ = svs_; _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) ga[m_] = sv_[m_]; }\n"
        "      if (t_ - 1 >= a_ && live_) { _Pragma(\"unroll\") for (int k_ = 0; k_ < %d; ++k_) wg_[((int64_t)(t_ - 1) * %d + k_) * n_ + ic_] = gcar_[k_] + ga[NOUT + k_]; }\n"
        "      _Pragma(\"unroll\") for (int k_ = 0; k_ < %d; ++k_) gcar_[k_] = ga[NOUT + %d + k_];\n    }\n"
        "    if (b_ == %d && live_) { _Pragma(\"unroll\") for (int k_ = 0; k_ < %d; ++k_) wg_[((int64_t)%d * %d + k_) * n_ + ic_] = gcar_[k_]; }\n"
        "    }\n", SS, SS, SS, SS, r.T, SS, r.T - 1, SS);
    o.f("    _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) g[m_] += CPL > 1 ? QSUM(ga[m_]) : ga[m_];\n    if (SC) sc_ += CPL > 1 ? QSUM(scp_) : scp_;\n  }\n");
  }
  for (int j = 0; j < (hp.rolled ? 0 : prog->n_sites);) {
    if (!hp.info[j].plate) { emit_one(j, "g", "sc_"); ++j; continue; }
    // ---- a plate (gjx.h "Plates"; the gradient of assess through a Vmap: hmc.py:70-96, vmap.py:363-376): the instances are dealt
    //      round-robin to the CPL lanes of the chain; every instance adds its body's score and its gradient terms — into the rows of
    //      the selected sites OUTSIDE the plate — to per-lane partials, joined across the chain's lanes behind the loop
    int m = 1;
    while (j + m < prog->n_sites && hp.info[j + m].plate && hp.info[j + m].plate_j0 == hp.info[j].plate_j0) ++m;
    o.f("  { // ---- plate of %d instances x %d sites\n    float ga[NSEL];\n    _Pragma(\"unroll\") for (int m_ = 0; m_ < NSEL; ++m_) ga[m_] = 0.0f;\n"
        "    float scp_ = 0.0f;\n    _Pragma(\"nounroll\") for (int i_ = q_; i_ < %d; i_ += CPL) {\n", hp.info[j].plate_n, m, hp.info[j].plate_n);
    for (int l = 0; l < m; ++l) {          // the instance's per-chain values (constrained per chain: OBS_SLOT), from the chain's column
      const gjx_site& sl = prog->sites[j + l];
      const RollInfo& rl = hp.info[j + l];
      if (sl.slot < 0) continue;
      const int w = is_categorical(sl.kind) ? 1 : sl.dim;
      const HmcPlan::PlateSel* ps = nullptr;
      for (const auto& c : hp.psel) if (c.j == j + l) ps = &c;
      for (int d = 0; d < w; ++d) {
        // a selected body site: the trajectory's current position of this instance (workspace row), and its gradient row starts at 0
        if (ps) o.f("    v[%d] = wq_[(int64_t)(%d + i_ * %d + %d) * n_ + ic_];\n    ga[%d] = 0.0f;\n", sl.slot + d, ps->prow0, ps->dim, d, ps->m0 + d);
        else o.f("    v[%d] = ch_[(int64_t)(%d + i_ * %d + %d) * n_ + ic_];\n", sl.slot + d, rl.row, rl.d_row, d);
      }
    }
    for (int l = 0; l < m; ++l) emit_one(j + l, "ga", "scp_");
    for (const auto& c : hp.psel)
      if (c.j >= j && c.j < j + m)
        for (int d = 0; d < c.dim; ++d) o.f("    if (live_) wg_[(int64_t)(%d + i_ * %d + %d) * n_ + ic_] = ga[%d];\n", c.prow0, c.dim, d, c.m0 + d);
    o.f("    }\n    _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) g[m_] += CPL > 1 ? QSUM(ga[m_]) : ga[m_];\n    if (SC) sc_ += CPL > 1 ? QSUM(scp_) : scp_;\n  }\n");
    j += m;
  }
  o.f("  return sc_;\n}\n\n");
  // ---- the kernel
  if (hp.big)
    o.f("extern \"C\" __global__ __launch_bounds__(BT) void gjx_hmc_gen(HmcGenArgs a) {\n"
        "  const float* __restrict__ tab_s = a.tab;     // (the table stays in memory: the LDS holds the chains' state)\n"
        "  __shared__ __attribute__((aligned(16))) float xt_s[4];\n"
        "  __shared__ __attribute__((aligned(16))) float st_s[(NS + %d * NSEL) * BT];\n", hp.nostale ? 2 : 3);
  else
  o.f("extern \"C\" __global__ __launch_bounds__(BT) void gjx_hmc_gen(HmcGenArgs a) {\n"
      "  __shared__ __attribute__((aligned(16))) float tab_s[NTAB > 0 ? ((NTAB + 3) & ~3) : 4];\n"
      "  __shared__ __attribute__((aligned(16))) float xt_s[XTF > 0 ? XTF : 4];\n"
      "  for (int t = threadIdx.x; t < NTAB; t += BT) tab_s[t] = a.tab[t];\n");
  for (int j = 0; j < prog->n_sites; ++j) {
    if (hp.mf_k[j] < 0) continue;
    const gjx_site& s = prog->sites[j];
    const gjx_param& q = s.p[hp.mf_k[j]];
    if (hp.fold[j]) {
      // the LDS copy of the matrix takes the sign of its row's observation and log2 e IN PLACE (no other parameter shares the
      // storage: hmc_fold_ok), then the transposed copy and the bias vector are made from it
      o.f("  __syncthreads();\n  for (int t = threadIdx.x; t < %d; t += BT) tab_s[%d + t] *= (2.0f * tab_s[%d + t / %d] - 1.0f) * 1.44269504f;   // site %d: rows folded\n",
          q.n * s.dim, q.moff, s.obs_off, q.n, j);
      o.f("  for (int t = threadIdx.x; t < %d; t += BT) xt_s[%d + t] = (2.0f * tab_s[%d + t] - 1.0f) * 1.44269504f * tab_s[%d + %s];\n",
          s.dim, hp.bs_off[j], s.obs_off, q.off, q.len == 1 ? "0" : "t");
      o.f("  __syncthreads();\n  for (int t = threadIdx.x; t < %d; t += BT) { const int k_ = t / %d, r_ = t - k_ * %d; xt_s[%d + k_ * %d + r_] = tab_s[%d + r_ * %d + k_]; }\n",
          q.n * s.dim, s.dim, s.dim, hp.xt_off[j], s.dim + 4, q.moff, q.n);
    } else
    o.f("  for (int t = threadIdx.x; t < %d; t += BT) { const int k_ = t / %d, r_ = t - k_ * %d; xt_s[%d + k_ * %d + r_] = a.tab[%d + r_ * %d + k_]; }   // site %d: X transposed\n",
        q.n * s.dim, s.dim, s.dim, hp.xt_off[j], s.dim + 4, q.moff, q.n, j);
  }
  o.f("  __syncthreads();\n");
  if (hp.mfma)
    o.f("  const int q_ = (int)(threadIdx.x & 63u) >> 4;\n"
        "  const int64_t i_raw = ((((int64_t)blockIdx.x * BT + threadIdx.x) >> 6) << 4) + (int64_t)(threadIdx.x & 15u);\n");
  else
    o.f("  const int q_ = (int)threadIdx.x %% CPL;\n"
        "  const int64_t i_raw = ((int64_t)blockIdx.x * BT + threadIdx.x) / CPL;\n");
  o.f(
      "  const bool live = i_raw < a.n;\n  const int64_t n = a.n, i = live ? i_raw : a.n - 1;   // (every lane runs: the quads reduce across lanes)\n"
      "  const uint64_t gidx = (uint64_t)(a.offset + i);\n"
      "%s"
      "  _Pragma(\"unroll\") for (int s_ = 0; s_ < NS; ++s_) v[s_] = 0.0f;\n",
      !hp.big ? "  float v[NS], g[NSEL], g0[NSEL], p[NSEL];\n"
      : (hp.nostale ? "  LdsCol v{st_s + threadIdx.x}, g{st_s + NS * BT + threadIdx.x}, p{st_s + (NS + NSEL) * BT + threadIdx.x}, g0{st_s + NS * BT + threadIdx.x};   // (no room for the first gradient: the launcher refuses the stale-carry mode)\n"
                    : "  LdsCol v{st_s + threadIdx.x}, g{st_s + NS * BT + threadIdx.x}, p{st_s + (NS + NSEL) * BT + threadIdx.x}, g0{st_s + (NS + 2 * NSEL) * BT + threadIdx.x};\n"));
  for (int j = 0; j < (hp.rolled ? hp.roll.i0 : prog->n_sites); ++j) {       // the chain's values: rows of choices[][] -> registers (a plate's body, a rolled Scan's steps: in the sweep)
    const gjx_site& sj = prog->sites[j];
    if (sj.slot < 0 || hp.info[j].plate) continue;
    const int w = (is_categorical(sj.kind) && sj.mode != GJX_MODE_INPUT) ? 1 : sj.dim;
    for (int d = 0; d < w; ++d) o.f("  v[%d] = a.choices[(int64_t)%d * n + i];\n", sj.slot + d, hp.info[j].row + d);
  }
  // the rows of the selected sites inside plates: working positions, momenta, gradients (and the first gradient for the stale-carry
  // compatibility mode) in the caller's workspace, [4][PROWS][n]; a lane owns the instances q_, q_ + CPL, ... of its chain
  auto plate_rows = [&](const char* body) {     // `body` sees idx_ (row * n + chain) and e_ (the element's index within its leaf)
    if (hp.rolled) {
      // a rolled Scan's rows: the lane's chunk of steps (the chunks of the sweep: a row is only ever touched by the lane that owns its step)
      std::string ls = "{";
      for (int k = 0; k < hp.ssel; ++k) ls += (k ? ", " : "") + std::to_string(hp.step_ls[k]);
      ls += "}";
      o.f("  { const int lsk_[%d] = %s; const int L_ = (%d + CPL - 1) / CPL, a_ = q_ * L_, b_ = a_ + L_ < %d ? a_ + L_ : %d;\n"
          "  _Pragma(\"nounroll\") for (int t_ = a_; t_ < b_; ++t_) _Pragma(\"unroll\") for (int k_ = 0; k_ < %d; ++k_) {\n"
          "    const int64_t idx_ = ((int64_t)t_ * %d + k_) * n + i, src_ = ((int64_t)%d + (int64_t)t_ * %d + lsk_[k_]) * n + i; (void)src_;\n"
          "    %s\n  } }\n", hp.ssel, ls.c_str(), hp.roll.T, hp.roll.T, hp.roll.T, hp.ssel, hp.ssel, hp.roll.n_pre, hp.roll.S, body);
      return;
    }
    for (const auto& c : hp.psel)
      o.f("  _Pragma(\"nounroll\") for (int i_ = q_; i_ < %d; i_ += CPL) _Pragma(\"unroll\") for (int d_ = 0; d_ < %d; ++d_) {\n"
          "    const int64_t idx_ = (int64_t)(%d + i_ * %d + d_) * n + i, src_ = (int64_t)(%d + i_ * %d + d_) * n + i; const uint32_t e_ = (uint32_t)(i_ * %d + d_); (void)src_; (void)e_;\n"
          "    %s\n  }\n", c.plate_n, c.dim, c.prow0, c.dim, c.row, c.d_row, c.dim, body);
  };
  if (hp.prows) {
    o.f("  float* const wq_ = a.ws; float* const wp_ = a.ws + (int64_t)PROWS * n; float* const wg_ = a.ws + 2 * (int64_t)PROWS * n; float* const wg0_ = a.ws + 3 * (int64_t)PROWS * n;\n");
    plate_rows("if (live) wq_[idx_] = a.choices[src_];");
  } else {
    o.f("  float* const wq_ = nullptr; float* const wg_ = nullptr;\n");
  }
  o.f("  const float score0 = sweep<true>(v, g, tab_s, xt_s, q_, a.choices, n, i, wq_, wg_, live);   // hmc.py:165-166\n"
      "  _Pragma(\"unroll\") for (int m_ = 0; m_ < NSEL; ++m_) g0[m_] = g[m_];\n"
      "  key2 knew{0u, 0u}, sub{0u, 0u};\n"
      "  if (RNG == GJX_RNG_JAX32) { const key2 ck = fold_in64(a.key, gidx); knew = fold_in(ck, 0u); sub = fold_in(ck, 1u); }   // hmc.py:167\n"
      "  float k0 = 0.0f;\n");
  if (hp.prows) { o.f("  if (a.stale) {\n"); plate_rows("if (live) wg0_[idx_] = wg_[idx_];"); o.f("  }\n"); }
  {   // momenta (hmc.py:120-130): leaf l = l-th selected address in program order
    int m = 0;
    bool scan_momenta_done = false;
    for (int j = 0; j < prog->n_sites; ++j) {
      const gjx_site& s = prog->sites[j];
      const int leaf = hp.leaf_of_site[j];
      if (leaf < 0) continue;
      if (hp.rolled && j >= hp.roll.i0) {
        if (scan_momenta_done) continue;     // (emitted once: every selected site of every step of the lane's chunk)
        scan_momenta_done = true;
        // the momenta of the Scan's selected sites: leaf = (leaves in front of the Scan) + t * (selected sites per step) + position in the step
        o.f("  { float kp_ = 0.0f; const int L_ = (%d + CPL - 1) / CPL, a_ = q_ * L_, b_ = a_ + L_ < %d ? a_ + L_ : %d;\n"
            "  _Pragma(\"nounroll\") for (int t_ = a_; t_ < b_; ++t_) {\n", hp.roll.T, hp.roll.T, hp.roll.T);
        int k = 0;
        for (int l = 0; l < hp.roll.m; ++l) {
          const int lf = hp.leaf_of_site[hp.roll.i0 + l];
          if (lf < 0) continue;
          const gjx_site& sl = prog->sites[hp.roll.i0 + hp.roll.m + l];
          o.f("    { BitStreamRT<RNG> bs; const uint32_t lf_ = (uint32_t)(%d + t_ * %d + %d);\n"
              "      if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, lf_)); else bs.open(a.key, gidx, lf_ + 1u);\n", hp.leaf0_scan, hp.sl_step, lf);
          for (int d = 0; d < sl.dim; ++d, ++k)
            o.f("      { const float pm_ = stream_normal<RNG>(bs, %du); if (live) wp_[((int64_t)t_ * %d + %d) * n + i] = pm_; kp_ += -0.5f * pm_ * pm_ - kHalfLog2Pi; }\n", d, hp.ssel, k);
          o.f("    }\n");
        }
        o.f("  }\n  k0 += CPL > 1 ? QSUM(kp_) : kp_; }\n");
        continue;
      }
      if (hp.info[j].plate) {
        // a selected body site is ONE leaf: element i dim + d for instance i — drawn by the lane that owns the instance
        for (const auto& c : hp.psel) {
          if (c.j != j) continue;
          o.f("  { float kp_ = 0.0f;\n  _Pragma(\"nounroll\") for (int i_ = q_; i_ < %d; i_ += CPL) {\n"
              "    BitStreamRT<RNG> bs;\n    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, %du)); else bs.open(a.key, gidx, %du);\n", c.plate_n, leaf, leaf + 1);
          o.f("    _Pragma(\"unroll\") for (int d_ = 0; d_ < %d; ++d_) {\n      const float pm_ = stream_normal<RNG>(bs, (uint32_t)(i_ * %d + d_));\n"
              "      if (live) wp_[(int64_t)(%d + i_ * %d + d_) * n + i] = pm_;\n      kp_ += -0.5f * pm_ * pm_ - kHalfLog2Pi;\n    }\n  }\n"
              "  k0 += CPL > 1 ? QSUM(kp_) : kp_; }\n", c.dim, c.dim, c.prow0, c.dim);
        }
        continue;
      }
      o.f("  { BitStreamRT<RNG> bs;\n    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(sub, %du)); else bs.open(a.key, gidx, %du);\n", leaf, leaf + 1);
      for (int d = 0; d < s.dim; ++d, ++m)
        o.f("    p[%d] = stream_normal<RNG>(bs, %du); k0 += -0.5f * p[%d] * p[%d] - kHalfLog2Pi;\n", m, d, m, m);
      o.f("  }\n");
    }
  }
  o.f("  const float he = 0.5f * a.eps;\n  float sc = score0;\n"
      "  for (int t = 1; t <= a.L; ++t) {   // hmc.py:170-194\n"
      "    _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) p[m_] += he * (a.stale ? g0[m_] : g[m_]);   // hmc.py:186: the carry keeps the received gradient\n");
  for (int m = 0; m < hp.nout; ++m) o.f("    v[%d] += a.eps * p[%d];\n", hp.slot_of_sel[m], m);
  if (hp.prows) plate_rows("{ const float pp_ = wp_[idx_] + he * (a.stale ? wg0_[idx_] : wg_[idx_]); if (live) { wp_[idx_] = pp_; wq_[idx_] += a.eps * pp_; } }");
  o.f("    if (t == a.L) sc = sweep<true>(v, g, tab_s, xt_s, q_, a.choices, n, i, wq_, wg_, live); else (void)sweep<false>(v, g, tab_s, xt_s, q_, a.choices, n, i, wq_, wg_, live);\n"
      "    _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) p[m_] += he * g[m_];\n");
  if (hp.prows) plate_rows("if (live) wp_[idx_] += he * wg_[idx_];");
  o.f("  }\n"
      "  float k1 = 0.0f;\n  _Pragma(\"unroll\") for (int m_ = 0; m_ < NOUT; ++m_) { const float q2_ = -1.0f * p[m_]; k1 += -0.5f * q2_ * q2_ - kHalfLog2Pi; }\n");
  if (hp.prows) {
    o.f("  { float kp_ = 0.0f;\n");
    plate_rows("{ const float q2_ = -1.0f * wp_[idx_]; kp_ += -0.5f * q2_ * q2_ - kHalfLog2Pi; }");
    o.f("  k1 += CPL > 1 ? QSUM(kp_) : kp_; }\n");
  }
  o.f("  const float al = sc - score0 + k1 - k0;   // hmc.py:196-203\n"
      "  bool acc = true;\n"
      "  if (a.accept) {\n    BitStream<RNG> bs;\n    if (RNG == GJX_RNG_JAX32) bs.open_site_key(fold_in(knew, 0x4d48u)); else bs.open(a.key, gidx, GJX_FLAT_MAX_SITES);\n"
      "    acc = safe_log(bits_to_unit(bs.get(0u))) < al;   // tests/inference/test_requests.py:134-137\n  }\n"
      "  if (!acc) sc = score0;\n"
      "  if (live && q_ == 0) {\n    if (acc) {\n");
  {
    std::vector<int> row_of_reg(NS > 0 ? NS : 1, -1);
    for (int j = 0; j < (hp.rolled ? hp.roll.i0 : prog->n_sites); ++j) {
      const gjx_site& sj = prog->sites[j];
      if (sj.slot < 0 || hp.info[j].plate) continue;
      const int w = (is_categorical(sj.kind) && sj.mode != GJX_MODE_INPUT) ? 1 : sj.dim;
      for (int d = 0; d < w; ++d) row_of_reg[sj.slot + d] = hp.info[j].row + d;
    }
    for (int m = 0; m < hp.nout; ++m) o.f("      a.choices[(int64_t)%d * n + i] = v[%d];\n", row_of_reg[hp.slot_of_sel[m]], hp.slot_of_sel[m]);
  }
  o.f("    }\n    if (a.score) a.score[i] = sc;\n    if (a.alpha) a.alpha[i] = al;\n    if (a.accepted) a.accepted[i] = acc ? 1.0f : 0.0f;\n  }\n");
  if (hp.prows) { o.f("  if (acc) {\n"); plate_rows("if (live) a.choices[src_] = wq_[idx_];"); o.f("  }\n"); }
  o.f("}\n");
  o.f("// PROWS %d\n// CPLMAX %d\n// NOSTALE %d\n", hp.prows, cpl_max, hp.nostale ? 1 : 0);
  o.f("// CPL %d\n// BT %d\n// LDS_FLOATS 0\n", cpl, hp.block);
  return o.s;
}

// ---------------------------------------------------------------------------------------------------------
// hipRTC (resolved at run time: the library must not need it when no program is ever generated)
// ---------------------------------------------------------------------------------------------------------
struct Rtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) Create = nullptr;
  decltype(&hiprtcCompileProgram) Compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) LogSize = nullptr;
  decltype(&hiprtcGetProgramLog) Log = nullptr;
  decltype(&hiprtcGetCodeSize) CodeSize = nullptr;
  decltype(&hiprtcGetCode) Code = nullptr;
  decltype(&hiprtcDestroyProgram) Destroy = nullptr;
  bool ok = false;
};

Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("GJX_HIPRTC"), "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
      if (!n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); ok = ok && p; return p; };
    r.Create = (decltype(r.Create))sym("hiprtcCreateProgram");
    r.Compile = (decltype(r.Compile))sym("hiprtcCompileProgram");
    r.LogSize = (decltype(r.LogSize))sym("hiprtcGetProgramLogSize");
    r.Log = (decltype(r.Log))sym("hiprtcGetProgramLog");
    r.CodeSize = (decltype(r.CodeSize))sym("hiprtcGetCodeSize");
    r.Code = (decltype(r.Code))sym("hiprtcGetCode");
    r.Destroy = (decltype(r.Destroy))sym("hiprtcDestroyProgram");
    r.ok = ok;
  });
  return r;
}

uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

std::string cache_dir() {
  if (const char* e = getenv("GJX_JIT_CACHE")) return e;
  Dl_info info;
  if (dladdr((void*)&fnv1a, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/jit_cache";
  }
  return "/tmp/gjx_jit_cache";
}

struct Compiled {
  std::vector<char> code;   // code object
  int lds_floats = 0;
  int cpl = 1, block = 256; // generated HMC kernels: lanes per chain, threads per block
  int prows = 0;            // generated HMC kernels: workspace rows (selected sites inside plates), x 4 x n floats
  int cpl_max = 1;          // generated HMC kernels: the most lanes per chain the program's loops can use
  int nostale = 0;          // generated HMC kernels, LDS-state flavour without room for the first gradient: no stale-carry mode
  std::string error;        // non-empty: this structure cannot be generated / compiled
};

std::mutex g_mu;
std::map<uint64_t, Compiled> g_compiled;                                 // by structure key
// gjx_jit_stats: kernels compiled by hipRTC in this process, code objects taken from the on-disk cache, time spent compiling (us)
std::atomic<int64_t> g_rtc_compiles{0}, g_disk_hits{0}, g_rtc_us{0};
std::map<std::pair<uint64_t, int>, std::pair<hipModule_t, hipFunction_t>> g_loaded;   // by (key, device)

// per-program analysis cached under gjx_program.uid (0 = no caching): the site-list hash, the emitter's verdict and the
// register footprint that decides PPT — each of them a walk over the whole site list
struct ProgMeta { bool roll_pref; uint64_t sites_hash; int supported; int slots; };   // supported / slots: -1 = not computed yet
std::mutex g_meta_mu;
std::unordered_map<int32_t, ProgMeta> g_meta;

// the site list AND the expression blocks its GJX_P_EXPR parameters name (node lists are structure: gjx.h)
uint64_t sites_hash_uncached(const gjx_program* p) {
  uint64_t h = fnv1a(p->sites, sizeof(gjx_site) * (size_t)p->n_sites);
  if (p->tab)
    for (int j = 0; j < p->n_sites; ++j)
      for (int k = 0; k < GJX_MAX_PARAMS; ++k) {
        const gjx_param& q = p->sites[j].p[k];
        if (p->sites[j].mode != GJX_MODE_INPUT && q.op == GJX_P_EXPR && q.off >= 0 && q.n > 0 && q.off + GJX_EXPR_NODE_FLOATS * q.n <= p->n_tab)
          h = fnv1a(p->tab + q.off, sizeof(float) * GJX_EXPR_NODE_FLOATS * (size_t)q.n, h);
      }
  return h | 1ull;
}

ProgMeta* meta_of(const gjx_program* p) {      // call with g_meta_mu held; nullptr when the program has no uid
  if (p->uid == 0) return nullptr;
  ProgMeta& m = g_meta[p->uid];
  if (m.sites_hash == 0 || m.roll_pref != want_roll()) m = ProgMeta{want_roll(), sites_hash_uncached(p), -1, -1};
  return &m;
}

uint64_t sites_hash(const gjx_program* p) {
  std::lock_guard<std::mutex> lock(g_meta_mu);
  if (ProgMeta* m = meta_of(p)) return m->sites_hash;
  return sites_hash_uncached(p);
}

bool supported(const gjx_program* p) {
  std::lock_guard<std::mutex> lock(g_meta_mu);
  ProgMeta* m = meta_of(p);
  if (!m) return supported_uncached(p);
  if (m->supported < 0) m->supported = supported_uncached(p) ? 1 : 0;
  return m->supported == 1;
}

// values a lane keeps in registers: all slots, or two steps' worth (+ the pre-Scan slots) when the program is rolled
int register_slots(const gjx_program* p) {
  std::lock_guard<std::mutex> lock(g_meta_mu);
  ProgMeta* m = meta_of(p);
  if (m && m->slots >= 0) return m->slots;
  int slots = p->n_slots;
  const PlateXf px = plate_program(p);
  if (px.any) slots = px.n_regs;
  else if (want_roll() || !supported_sites(p->sites, p->n_sites, p->n_slots, p)) {
    const Roll r = detect_roll(p);
    if (r.ok) slots = r.n_pre + 2 * r.S;
  }
  if (m) m->slots = slots;
  return slots;
}

uint64_t structure_key(const gjx_program* p, int ppt, int flavour = 0) {   // flavour 0: propagate+reweight kernel, 1: HMC kernel
  uint64_t h = sites_hash(p);
  if (flavour == 1) {   // the HMC emitter's data-dependent choice (hmc_fold_ok reads the observations): part of the kernel's identity
    HmcPlan hp;
    if (hmc_plan(p, &hp)) h = fnv1a(hp.fold.data(), hp.fold.size(), h);
  }
  const int32_t extra[8] = {p->n_sites, p->n_slots, p->n_tab, p->rng_mode, ppt, want_roll() ? 1 : 0, flavour,
                            (getenv("GJX_GEN_MFMA_DEBUG") ? atoi(getenv("GJX_GEN_MFMA_DEBUG")) : 0) ^
                                (getenv("GJX_HMC_GEN_BT") ? atoi(getenv("GJX_HMC_GEN_BT")) << 8 : 0) ^ (getenv("GJX_HMC_GEN_NO_MFMA") ? 1 << 20 : 0) ^
                                (getenv("GJX_HMC_GEN_DEBUG") ? atoi(getenv("GJX_HMC_GEN_DEBUG")) << 21 : 0) ^ (getenv("GJX_GEN_TAB_GLOBAL") ? 1 << 24 : 0) ^
                                (getenv("GJX_GEN_NO_HOIST") ? 1 << 25 : 0) ^ (getenv("GJX_GEN_NO_FUSE") ? 1 << 26 : 0) ^ (getenv("GJX_GEN_NO_EARLY_STORE") ? 1 << 27 : 0) ^
                                (getenv("GJX_GEN_NO_SEQ_ROWS") ? 1 << 28 : 0) ^ (getenv("GJX_JIT_FP_CONTRACT") ? 1 << 29 : 0) ^ (getenv("GJX_GEN_WAVES_PER_EU") ? atoi(getenv("GJX_GEN_WAVES_PER_EU")) << 12 : 0)};
  h = fnv1a(extra, sizeof(extra), h);
  static const uint64_t header_hash = fnv1a(kDeviceHeader, strlen(kDeviceHeader));   // a new device header invalidates the caches
  return h ^ header_hash ^ (0x9E3779B97F4A7C15ull * GJX_ABI_VERSION);
}

const Compiled& compile(const gjx_program* prog, int ppt, int flavour = 0) {
  const uint64_t key = structure_key(prog, ppt, flavour);
  auto it = g_compiled.find(key);
  if (it != g_compiled.end()) return it->second;
  Compiled& c = g_compiled[key];
  const std::string src = flavour == 1 ? generate_hmc(prog, ppt) : (flavour == 2 ? generate_pf(prog, ppt) : generate(prog, ppt));
  if (src.empty()) { c.error = "codegen: program outside the emitter's coverage"; return c; }
  const size_t m = src.rfind("// LDS_FLOATS ");
  c.lds_floats = atoi(src.c_str() + m + 14);
  const size_t mc = src.rfind("// CPL ");
  if (mc != std::string::npos) c.cpl = atoi(src.c_str() + mc + 7);
  const size_t mb = src.rfind("// BT ");
  if (mb != std::string::npos) c.block = atoi(src.c_str() + mb + 6);
  const size_t mp = src.rfind("// PROWS ");
  if (mp != std::string::npos) c.prows = atoi(src.c_str() + mp + 9);
  const size_t mx = src.rfind("// CPLMAX ");
  if (mx != std::string::npos) c.cpl_max = atoi(src.c_str() + mx + 10);
  const size_t mn = src.rfind("// NOSTALE ");
  if (mn != std::string::npos) c.nostale = atoi(src.c_str() + mn + 11);
  if ((size_t)c.lds_floats * 4 + 256 > 64 * 1024) { c.error = "the program's table does not fit the LDS budget"; return c; }
  // the code object on disk is named by the SOURCE it was compiled from (and the headers): a changed emitter or header
  // can never pick up a stale file
  char name[64];
  // generated kernels are compiled WITHOUT implicit fused multiply-adds (the explicit fmaf of the emitters and of gjx_device.h stay):
  // the same site then rounds the same way in every kernel it is compiled into — the filter kernel with one particle per lane and
  // gjx_gen with four gave a student-t draw that differed in the last bit — at no measurable cost (mixture kernel 31.6 -> 31.9 us,
  // filter steps unchanged); GJX_JIT_FP_CONTRACT=fast restores the compiler's default for experiments
  // — for the propagate and filter kernels; the HMC kernels (one kernel per program: nothing to agree with) keep the default, which is
  // worth 13 - 25 % on gradient sweeps written without explicit fmaf
  const bool no_contract = flavour != 1 && !(getenv("GJX_JIT_FP_CONTRACT") && !strcmp(getenv("GJX_JIT_FP_CONTRACT"), "fast"));
  snprintf(name, sizeof(name), "%016llx", (unsigned long long)((no_contract ? 0x5bd1e995ull : 0ull) ^ fnv1a(src.data(), src.size()) ^ fnv1a(kDeviceHeader, strlen(kDeviceHeader)) ^ fnv1a(kApiHeader, strlen(kApiHeader)) ^
                                                               fnv1a(kScanHeader, strlen(kScanHeader)) ^ (fnv1a(kTileHeader, strlen(kTileHeader)) << 1) ^
                                                               (flavour == 2 ? fnv1a(kPfCoreHeader, strlen(kPfCoreHeader)) << 2 : 0ull)));
  const std::string dir = cache_dir(), path = dir + "/" + name + ".hsaco";
  if (!getenv("GJX_JIT_NO_DISK")) {
    if (FILE* f = fopen(path.c_str(), "rb")) {
      fseek(f, 0, SEEK_END);
      const long n = ftell(f);
      fseek(f, 0, SEEK_SET);
      c.code.resize((size_t)n);
      const size_t got = fread(c.code.data(), 1, (size_t)n, f);
      fclose(f);
      if (got == (size_t)n && n > 0) { g_disk_hits++; return c; }
      c.code.clear();
    }
  }
  if (getenv("GJX_JIT_DUMP")) {
    if (FILE* f = fopen((std::string(getenv("GJX_JIT_DUMP")) + "/" + name + ".hip").c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
  }
  Rtc& r = rtc();
  if (!r.ok) { c.error = "hipRTC is not available (libhiprtc.so)"; return c; }
  hiprtcProgram p;
  const char* hn[] = {"gjx_device.h", "../../include/gjx.h", "gjx_scan.h", "gjx_tile.h", "gjx_pfcore.h"};
  const char* hs[] = {kDeviceHeader, kApiHeader, kScanHeader, kTileHeader, kPfCoreHeader};
  if (r.Create(&p, src.c_str(), flavour == 1 ? "gjx_hmc_gen.hip" : (flavour == 2 ? "gjx_gen_pf.hip" : "gjx_gen.hip"), 5, hs, hn) != HIPRTC_SUCCESS) { c.error = "hiprtcCreateProgram failed"; return c; }
  // (offline clang takes -mllvm -amdgpu-mfma-vgpr-form=1, which would keep matrix-core results out of the AGPRs; this hipRTC's LLVM
  // does not know the option, so the generated kernels pay 16 v_accvgpr_read per tile: about 3 %)
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
  const auto t_rtc = std::chrono::steady_clock::now();
  const hiprtcResult rc = r.Compile(p, no_contract ? 4 : 3, opts);
  g_rtc_compiles++;
  g_rtc_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_rtc).count();
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    r.LogSize(p, &ls);
    std::string log(ls, 0);
    if (ls) r.Log(p, &log[0]);
    c.error = "hipRTC: " + log.substr(0, 1500);
    r.Destroy(&p);
    return c;
  }
  size_t cs = 0;
  r.CodeSize(p, &cs);
  c.code.resize(cs);
  r.Code(p, c.code.data());
  r.Destroy(&p);
  if (!getenv("GJX_JIT_NO_DISK")) {
    mkdir(dir.c_str(), 0755);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    if (FILE* f = fopen(tmp.c_str(), "wb")) {
      fwrite(c.code.data(), 1, c.code.size(), f);
      fclose(f);
      rename(tmp.c_str(), path.c_str());
    }
  }
  return c;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// interface used by gjx_run.hip
// ---------------------------------------------------------------------------------------------------------
extern "C" int gjx_jit_stats(int64_t* out4) {
  if (!out4) return GJX_EINVAL;
  out4[0] = g_rtc_compiles.load();
  out4[1] = g_disk_hits.load();
  out4[2] = g_rtc_us.load();
  { std::lock_guard<std::mutex> lock(g_mu); out4[3] = (int64_t)g_compiled.size(); }
  return GJX_OK;
}

namespace gjx {

int gen_pick_ppt(const gjx_program* prog, int64_t K, bool prefer4) {
  const int slots = register_slots(prog);
  int ppt = slots <= 6 ? 4 : (slots <= 24 ? 2 : 1);
  if (prefer4 && slots <= 40) ppt = 4;     // a block-tile of 1024 particles = one quantisation tile (tile totals, GJX_RUN_LEAVE_TILES)
  // a big affine site goes to the matrix cores: one particle per lane, whole waves (code = ppt | 256: see generate())
  if (K % 256 == 0 && !getenv("GJX_GEN_PPT") && has_mfma_site(prog)) return 1 | 256;
  // a long plate: the instances dealt to the 16 waves of a block (code = ppt | 512), unless the particles alone fill the machine
  // many times over (then the plain form's single pass per particle has less overhead); GJX_GEN_WIDE = 0 / 1 forces the choice
  {
    int longest = 0;
    for (int j = 0; j < prog->n_sites; ++j) if (prog->sites[j].plate && prog->sites[j].plate_n > longest) longest = prog->sites[j].plate_n;
    const char* e = getenv("GJX_GEN_WIDE");
    // (measured, vmapped mixture: N = 4096 x K = 2^17 — 256 plain blocks, one wave per SIMD — 4.1x faster wide; N = 1024 x K = 2^20 —
    // 2048 plain blocks — 8 % slower wide: the plain form wins once the particles alone give every SIMD four waves)
    const bool want = e ? atoi(e) != 0 : (longest >= 64 && K / (256 * (int64_t)ppt) < 1024);
    if (want && longest >= 16 && !prefer4) {
      int wp = slots <= 4 ? 2 : 1;
      if (const char* pe = getenv("GJX_GEN_PPT")) { const int q = atoi(pe); if (q == 1 || q == 2) wp = q; }
      while (wp > 1 && K % wp != 0) wp >>= 1;
      // few particles, very many instances: 4 or 16 lanes per particle (| 1024, | 2048) until the launch has two blocks per CU
      // (measured, N = 2^16 x K = 2^12: one lane per particle, 64 blocks, 5.36 ms; 4 lanes, 256 blocks, 1.58 ms; 16 lanes, 1024 blocks, 1.28 ms)
      int lpp = 1;
      while (lpp < 16 && (K * lpp) / (64 * (int64_t)wp) < 512 && longest >= 16 * (lpp * 4) * 16 && K % (64 * wp / (lpp * 4)) == 0) lpp *= 4;
      if (const char* le = getenv("GJX_GEN_LPP")) { const int q = atoi(le); if (q == 1 || q == 4 || q == 16) lpp = q; }
      return wp | 512 | (lpp == 4 ? 1024 : (lpp == 16 ? 2048 : 0));
    }
  }
  if (const char* e = getenv("GJX_GEN_PPT")) ppt = atoi(e);
  if (ppt != 1 && ppt != 2 && ppt != 4) ppt = 1;
  while (ppt > 1 && K % ppt != 0) ppt >>= 1;
  return ppt;
}

// 0: a generated kernel exists (compiled now if need be); otherwise the reason is in gjx_last_error
int gen_available(const gjx_program* prog, int ppt) {
  if (!supported(prog)) return gjx_fail(GJX_EUNSUPPORTED, "codegen: program outside the emitter's coverage");
  std::lock_guard<std::mutex> lock(g_mu);
  const Compiled& c = compile(prog, ppt);
  if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
  return GJX_OK;
}

int gen_launch(const gjx_program* prog, int ppt, const GenArgs& args, int grid, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  hipFunction_t fn = nullptr;
  int lds_floats = 0;
  unsigned block = 256;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    const Compiled& c = compile(prog, ppt);
    if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
    lds_floats = c.lds_floats;
    block = (unsigned)c.block;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gjx_fail(GJX_EHIP, "codegen: no device");
    const auto lk = std::make_pair(structure_key(prog, ppt), dev);
    auto it = g_loaded.find(lk);
    if (it == g_loaded.end()) {
      hipModule_t mod;
      hipError_t e = hipModuleLoadData(&mod, c.code.data());
      if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleLoadData");
      e = hipModuleGetFunction(&fn, mod, "gjx_gen");
      if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleGetFunction");
      g_loaded[lk] = std::make_pair(mod, fn);
    } else {
      fn = it->second.second;
    }
  }
  GenArgs a = args;
  size_t sz = sizeof(a);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  hipError_t e;
  if (ev0 && ev1) e = hipExtModuleLaunchKernel(fn, (uint32_t)grid * block, 1, 1, block, 1, 1, (size_t)lds_floats * 4, st, nullptr, config, ev0, ev1, 0);
  else e = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, block, 1, 1, (unsigned)(lds_floats * 4), st, nullptr, config);
  if (e != hipSuccess) return gjx_fail_hip(e, "codegen: launch");
  return GJX_OK;
}

// ---- the steps kernel of a generated filter (gjx_scanfilter.hip): gjx_gen_steps of the module generated for `prog` ----
// two programs run through one steps kernel only if they ARE one kernel: same structure key (sites, table size, stream layout)
bool gen_same_kernel(const gjx_program* p, const gjx_program* q, int ppt) { return structure_key(p, ppt) == structure_key(q, ppt); }

// blocks of gjx_gen_steps that are resident at the same time on the current device, or 0 (no such kernel / query failed)
static int gen_steps_function(const gjx_program* prog, int ppt, hipFunction_t* fn_out, int* lds_floats) {
  std::lock_guard<std::mutex> lock(g_mu);
  const Compiled& c = compile(prog, ppt);
  if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
  *lds_floats = c.lds_floats;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return gjx_fail(GJX_EHIP, "codegen: no device");
  static std::map<std::pair<uint64_t, int>, hipFunction_t> steps_fn;
  const auto lk = std::make_pair(structure_key(prog, ppt), dev);
  auto sit = steps_fn.find(lk);
  if (sit != steps_fn.end()) { *fn_out = sit->second; return *fn_out ? GJX_OK : GJX_EUNSUPPORTED; }
  hipModule_t mod;
  auto it = g_loaded.find(lk);
  if (it == g_loaded.end()) {
    hipFunction_t fn;
    hipError_t e = hipModuleLoadData(&mod, c.code.data());
    if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleLoadData");
    e = hipModuleGetFunction(&fn, mod, "gjx_gen");
    if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleGetFunction");
    g_loaded[lk] = std::make_pair(mod, fn);
  } else {
    mod = it->second.first;
  }
  hipFunction_t sf = nullptr;
  if (hipModuleGetFunction(&sf, mod, "gjx_gen_steps") != hipSuccess) { (void)hipGetLastError(); sf = nullptr; }
  steps_fn[lk] = sf;
  *fn_out = sf;
  return sf ? GJX_OK : GJX_EUNSUPPORTED;
}

int gen_steps_resident_blocks(const gjx_program* prog, int ppt) {
  hipFunction_t fn = nullptr;
  int lds_floats = 0;
  if (gen_steps_function(prog, ppt, &fn, &lds_floats) != GJX_OK) return 0;
  int per_cu = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, (size_t)lds_floats * 4) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (gjx_plain_launches_forced()) return 0;
  return (per_cu > 6 ? 6 : per_cu) * cus;        // (as gjx_coresident_blocks: answers above 6 per CU are not exact; tests override through gjx_filter_opts)
}

int gen_steps_launch(const gjx_program* prog, int ppt, const GenStepsArgs& args, int grid, hipStream_t st) {
  hipFunction_t fn = nullptr;
  int lds_floats = 0;
  const int rc = gen_steps_function(prog, ppt, &fn, &lds_floats);
  if (rc) return rc;
  GenStepsArgs a = args;
  size_t sz = sizeof(a);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, (unsigned)(lds_floats * 4), st, nullptr, config);
  if (e != hipSuccess) return gjx_fail_hip(e, "codegen: launch (steps kernel)");
  return GJX_OK;
}

// ---- generated filter kernels (gjx_scanfilter.hip, gjx_peer.hip): gjx_gen_pf of the module generated for a step program ----
bool gen_pf_supported(const gjx_program* p) { return pf_supported(p); }
bool gen_pf_moves_supported(const gjx_program* p) { return pf_moves_supported(p); }
// two step programs run as steps of one launch only if they ARE one kernel
bool gen_pf_same_kernel(const gjx_program* p, const gjx_program* q) { return structure_key(p, 1, 2) == structure_key(q, 1, 2); }

static int gen_pf_function(const gjx_program* prog, int spl, hipFunction_t* fn_out) {
  std::lock_guard<std::mutex> lock(g_mu);
  const Compiled& c = compile(prog, spl, 2);
  if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return gjx_fail(GJX_EHIP, "codegen: no device");
  const auto lk = std::make_pair(structure_key(prog, spl, 2), dev);
  auto it = g_loaded.find(lk);
  if (it == g_loaded.end()) {
    hipModule_t mod;
    hipFunction_t fn;
    hipError_t e = hipModuleLoadData(&mod, c.code.data());
    if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleLoadData");
    e = hipModuleGetFunction(&fn, mod, "gjx_gen_pf");
    if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleGetFunction");
    // (static + dynamic LDS is above the 64 KB default once a run has more than ~2000 tiles)
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess) (void)hipGetLastError();
    g_loaded[lk] = std::make_pair(mod, fn);
    *fn_out = fn;
  } else {
    *fn_out = it->second.second;
  }
  return GJX_OK;
}

int gen_pf_precompile(const gjx_program* prog, int spl) {
  if (!pf_supported(prog)) return gjx_fail(GJX_EUNSUPPORTED, "codegen: step program outside the filter emitter's coverage");
  std::lock_guard<std::mutex> lock(g_mu);
  const Compiled& c = compile(prog, spl, 2);
  if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
  return GJX_OK;
}

// blocks of gjx_gen_pf<spl> that are resident at the same time on the current device, or 0
int gen_pf_resident_blocks(const gjx_program* prog, int spl, size_t dyn_lds) {
  if (gjx_plain_launches_forced()) return 0;
  hipFunction_t fn = nullptr;
  if (gen_pf_function(prog, spl, &fn) != GJX_OK) return 0;
  int per_cu = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 1024, dyn_lds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return (per_cu > 2 ? 2 : per_cu) * cus;
}

int gen_pf_launch(const gjx_program* prog, int spl, const GenPfArgs& args, int grid, size_t dyn_lds, hipStream_t st) {
  hipFunction_t fn = nullptr;
  const int rc = gen_pf_function(prog, spl, &fn);
  if (rc) return rc;
  GenPfArgs a = args;
  size_t sz = sizeof(a);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 1024, 1, 1, (unsigned)dyn_lds, st, nullptr, config);
  if (e != hipSuccess) return gjx_fail_hip(e, "codegen: launch (filter kernel)");
  return GJX_OK;
}

// ---- generated HMC kernels (gjx_hmc.hip) ----
int hmc_gen_available(const gjx_program* prog) {
  HmcPlan hp;
  if (!hmc_plan(prog, &hp)) return gjx_fail(GJX_EUNSUPPORTED, "codegen: program outside the HMC emitter's coverage");
  std::lock_guard<std::mutex> lock(g_mu);
  const Compiled& c = compile(prog, 0, 1);
  if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
  return GJX_OK;
}

int hmc_gen_launch(const gjx_program* prog, const HmcGenArgs& args, hipStream_t st) {
  hipFunction_t fn = nullptr;
  int cpl = 1, block = 256;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    const Compiled& c0 = compile(prog, 0, 1);
    if (!c0.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c0.error.c_str());
    // few chains over long loops (the usual shape of HMC: hundreds of chains, thousands of data): more lanes per chain, until the
    // launch has about two waves per SIMD (GJX_HMC_GEN_CPL forces 4, 16 or 64)
    int variant = 0;
    {
      int want = 4;
      while (want < c0.cpl_max && args.n * want < 131072) want *= 4;
      if (const char* e = getenv("GJX_HMC_GEN_CPL")) want = atoi(e);
      if ((want == 16 || want == 64) && want <= c0.cpl_max) variant = want;
    }
    const Compiled& c = variant ? compile(prog, variant, 1) : c0;
    if (!c.error.empty()) return gjx_fail(GJX_EUNSUPPORTED, c.error.c_str());
    cpl = c.cpl;
    block = c.block;
    if (c.nostale && args.stale) return gjx_fail(GJX_EUNSUPPORTED, "codegen: the LDS-state HMC kernel of this program has no room for the stale-carry compatibility mode");
    if (c.prows > 0 && (!args.ws || args.ws_floats < 4 * (int64_t)c.prows * args.n))
      return gjx_fail(GJX_EWORKSPACE, "gjx_hmc: workspace too small (selected sites inside a plate keep their trajectory state there)");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gjx_fail(GJX_EHIP, "codegen: no device");
    const auto lk = std::make_pair(structure_key(prog, variant, 1), dev);
    auto it = g_loaded.find(lk);
    if (it == g_loaded.end()) {
      hipModule_t mod;
      hipError_t e = hipModuleLoadData(&mod, c.code.data());
      if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleLoadData");
      e = hipModuleGetFunction(&fn, mod, "gjx_hmc_gen");
      if (e != hipSuccess) return gjx_fail_hip(e, "codegen: hipModuleGetFunction");
      g_loaded[lk] = std::make_pair(mod, fn);
    } else {
      fn = it->second.second;
    }
  }
  HmcGenArgs a = args;
  size_t sz = sizeof(a);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  const int64_t threads = a.n * cpl;
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  const hipError_t e = hipModuleLaunchKernel(fn, grid, 1, 1, (unsigned)block, 1, 1, 0, st, nullptr, config);
  if (e != hipSuccess) return gjx_fail_hip(e, "codegen: launch (HMC kernel)");
  return GJX_OK;
}

}  // namespace gjx

// source of the generated HMC kernel of a program (GJX_EUNSUPPORTED when the emitter does not cover it)
extern "C" int64_t gjx_program_hmc_source(const gjx_program* prog, char* out, int64_t cap) {
  if (!prog || !prog->sites) return GJX_EINVAL;
  const std::string src = generate_hmc(prog);
  if (src.empty()) return gjx_fail(GJX_EUNSUPPORTED, "codegen: program outside the HMC emitter's coverage");
  if (out && cap > 0) {
    const size_t n = src.size() < (size_t)cap - 1 ? src.size() : (size_t)cap - 1;
    memcpy(out, src.data(), n);
    out[n] = 0;
  }
  return (int64_t)src.size();
}

// compile (or load from the disk cache) the generated HMC kernel of a program without launching it
extern "C" int gjx_program_hmc_precompile(const gjx_program* prog) {
  if (!prog || !prog->sites) return gjx_fail(GJX_EINVAL, "gjx_program_hmc_precompile: null program");
  return gjx::hmc_gen_available(prog);
}

// the generated source of a program (debugging, tests, docs): returns the length, copies at most cap - 1 characters
extern "C" int64_t gjx_program_source(const gjx_program* prog, int32_t ppt, char* out, int64_t cap) {
  if (!prog || !prog->sites) return GJX_EINVAL;
  if (!supported(prog)) return gjx_fail(GJX_EUNSUPPORTED, "codegen: program outside the emitter's coverage");
  {
    const int base = ppt & ~(1024 | 2048);
    if ((base != 1 && base != 2 && base != 4 && base != (1 | 256) && base != (1 | 512) && base != (2 | 512)) || ((ppt & (1024 | 2048)) && !(base & 512)))
      ppt = gjx::gen_pick_ppt(prog, 4);
  }
  const std::string src = generate(prog, ppt);
  if (out && cap > 0) {
    const size_t n = src.size() < (size_t)cap - 1 ? src.size() : (size_t)cap - 1;
    memcpy(out, src.data(), n);
    out[n] = 0;
  }
  return (int64_t)src.size();
}

// compile (or load from the disk cache) the kernel of a program without launching it — build steps pre-populate the
// cache with this on machines without a GPU (hipRTC cross-compiles)
extern "C" int gjx_program_precompile(const gjx_program* prog, int32_t ppt) {
  if (!prog || !prog->sites) return gjx_fail(GJX_EINVAL, "gjx_program_precompile: null program");
  const int wide_lanes = ppt & (1024 | 2048);
  const int base = ppt & ~(1024 | 2048);
  if ((base != 1 && base != 2 && base != 4 && base != (1 | 256) && base != (1 | 512) && base != (2 | 512)) || (wide_lanes && !(base & 512)) || wide_lanes == (1024 | 2048))
    return gjx_fail(GJX_EINVAL, "gjx_program_precompile: ppt must be 1, 2, 4, 257 (1 | 256: big affine sites on the matrix cores) or 513 / 514 (| 512: the instances of a plate dealt to the 16 waves of a block; | 1024 / | 2048: and to 4 / 16 lanes per particle)");
  return gjx::gen_available(prog, ppt);
}

// the filter kernel generated for a step program (GJX_FILTER_FORM_WIDE of gjx_scan_filter): source, and compile without launch
extern "C" int64_t gjx_program_filter_source(const gjx_program* step, int32_t tiles_per_block, char* out, int64_t cap) {
  if (!step || !step->sites) return GJX_EINVAL;
  const std::string src = generate_pf(step, tiles_per_block);
  if (src.empty()) return gjx_fail(GJX_EUNSUPPORTED, "codegen: step program outside the filter emitter's coverage");
  if (out && cap > 0) {
    const size_t n = src.size() < (size_t)cap - 1 ? src.size() : (size_t)cap - 1;
    memcpy(out, src.data(), n);
    out[n] = 0;
  }
  return (int64_t)src.size();
}

extern "C" int gjx_program_filter_precompile(const gjx_program* step, int32_t tiles_per_block) {
  if (!step || !step->sites) return gjx_fail(GJX_EINVAL, "gjx_program_filter_precompile: null program");
  const int tpb = tiles_per_block & 255;
  if ((tiles_per_block & ~(255 | 256 | 512 | 1024)) || (tpb != 1 && tpb != 2 && tpb != 4 && tpb != 8 && tpb != 16) || (tiles_per_block & (256 | 512 | 1024)) > 1024)
    return gjx_fail(GJX_EINVAL, "gjx_program_filter_precompile: tiles_per_block must be 1, 2, 4, 8 or 16 (| 256: the flavour for sharded collections, | 512: with the rejuvenation move, | 1024: multinomial resampling by sorted uniforms, on its own)");
  return gjx::gen_pf_precompile(step, tiles_per_block);
}
