// gjx_scanfilter.hip — bootstrap filter for ANY Scan kernel (gjx_scan_filter): the step recursion of Scan.generate
// (combinators/scan.py:237-294: step t receives the carry of step t-1, weights add over steps) with systematic resampling in
// front of every step.  Host code only.  Per step ONE plain launch where the step's generated kernel can resample in its
// prologue (gjx_run_resample: 4 particles per lane, K a multiple of 1024 up to 2^20 — every block searches the ancestors of
// its own tile from the previous step's log-weights and tile totals, which therefore alternate between two buffers / two
// run workspaces), otherwise TWO, issued back to back from this loop —
//   1. the tile-scaled systematic resampler's search (gjx_resample_gather_tiled with no rows to copy): log-weights of step
//      t-1 -> ancestors; the block pairs and tile totals come from the producing kernel, so this launch reads 4 B per particle
//      and writes 4 B; it also finishes the LSE record of step t-1;
//   2. the step's generated propagate + reweight kernel (gjx_run_program_ex) whose GJX_MODE_INPUT sites read the carry through
//      the ancestors — the particle gather is fused into the read side, the resampled collection is never materialised —
//      and which leaves the {max, sumexp} block pairs and the tile totals of ITS log-weights for the next search.
// The hand-written linear-Gaussian filters (gjx_ssm.hip, gjx_pfilter.inl) stay the fast path for that one model.
#include <math.h>
#include <string.h>

#include <vector>

#include "gjx_device.h"
#include "gjx_host.h"
#include "gjx_tile.h"
#include "gjx_pfcore.h"
#include "gjx_pfilter_host.h"

using namespace gjx;

// tile totals {S_b, e_b} and block pairs of a step that ran as its own launch -> the tagged granules and the pair array the steps
// kernel's first step polls / reads (gjx_gen_steps)
static __global__ void k_clear_status(unsigned* ctrl) { if (threadIdx.x == 0) ctrl[2] = 0u; }
// accepted chains of one HMC move (gjx_hmc's flags f32[K]) added to the run's counter
// (a grid-stride loop over at most 64 blocks, ONE atomic per block: a wave-level atomic per 64 chains — 1024 of them on one address at
// K = 2^16 — took 13 us, four times the HMC kernel it counts for; rocprofv3, profiles/r06_moves_kernel_stats.csv)
static __global__ __launch_bounds__(256) void k_count_flags(const float* flags, int64_t K, unsigned long long* total) {
  __shared__ unsigned wcount[4];
  unsigned n = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < K; i += (int64_t)gridDim.x * 256) n += flags[i] > 0.5f ? 1u : 0u;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
  if ((threadIdx.x & 63) == 0) wcount[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    if (t) atomicAdd(total, (unsigned long long)t);
  }
}

static __global__ void k_merge_status(unsigned* from, unsigned* to) { if (threadIdx.x == 0 && from[2]) { atomicOr(&to[2], from[2]); from[2] = 0u; } }

__global__ void k_tiles_to_granules(const uint64_t* __restrict__ S, const int32_t* __restrict__ E, const unsigned long long* __restrict__ pairs,
                                    unsigned long long* gran, unsigned long long* part, int nt, unsigned long long tag) {
  const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (b >= nt) return;
  const uint64_t sv = S[b];
  gran[(size_t)b * kLiveGranulePad] = tile_granule(tag, sv ? E[b] : kTileDead, sv);
  part[b] = pairs[b];
}

// rows_of(t): the buffer step t writes its choices into (two alternating buffers, or one per step when the run is recorded)
template <class RowsOf>
static int scan_filter_impl(const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, int64_t K, RowsOf&& rows_of,
                            float* logw, int32_t* ancestors, int32_t* ancestors_all, float* lse_steps, void* workspace,
                            size_t workspace_bytes, void* stream, const gjx_filter_opts* opts, gjx_filter_info* info_out) {
  const int32_t fflags = opts ? opts->flags : 0;
  const int n_moves = opts ? opts->n_moves : 0;
  if (n_moves < 0) return gjx_fail(GJX_EINVAL, "gjx_scan_filter: n_moves < 0");
  gjx_filter_info finfo = {GJX_FILTER_FORM_TWO_LAUNCH, 0, 0, 0};
  auto report = [&](int rc_) { if (info_out) *info_out = finfo; return rc_; };
  if (info_out) *info_out = finfo;
  const size_t need_run = gjx_workspace_bytes(GJX_OP_RUN, K), need_res = gjx_workspace_bytes(GJX_OP_RESAMPLE, K);
  if (!workspace || workspace_bytes < need_run + need_res) return gjx_fail(GJX_EWORKSPACE, "gjx_scan_filter: workspace too small (OP_RUN + OP_RESAMPLE)");
  char* ws_run = (char*)workspace;
  char* ws_res = ws_run + need_run;
  // the one-launch step needs a second run workspace and a second log-weight buffer behind the two the call must have
  const size_t logw_off = (need_run + need_res + need_run + 255) & ~(size_t)255;
  const bool room = workspace_bytes >= logw_off + sizeof(float) * (size_t)K;
  char* ws_run2 = room ? ws_res + need_res : nullptr;
  float* logw2 = room ? (float*)((char*)workspace + logw_off) : nullptr;
  // ... and the steps kernel (every step from the third in one launch) an area for its granules, pair arrays and per-step arguments
  const int64_t nt = K / 1024;
  const size_t steps_off = (logw_off + sizeof(float) * (size_t)K + 255) & ~(size_t)255;
  const size_t steps_bytes = (16 * (size_t)kLiveGranulePad + 16) * (size_t)nt + 24 * (size_t)T + 64;
  char* steps_area = (room && K % 1024 == 0 && workspace_bytes >= steps_off + steps_bytes) ? (char*)workspace + steps_off : nullptr;
  // an HMC move behind every resampling (gjx_filter_opts::hmc_targets): the plain two-launch step with a gather and one gjx_hmc between them
  const gjx_program* hmc_t = opts ? opts->hmc_targets : nullptr;
  if (hmc_t) {
    if (n_moves > 0 || (fflags & GJX_FILTER_ABSOLUTE_INPUTS)) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter: the HMC move runs without n_moves and without carried static inputs");
    if (!opts->hmc_rows || !opts->hmc_out || !opts->hmc_workspace || opts->hmc_L < 1 || !(opts->hmc_eps > 0.0f))
      return gjx_fail(GJX_EINVAL, "gjx_scan_filter: hmc_targets needs hmc_rows, hmc_out, hmc_workspace, hmc_L >= 1, hmc_eps > 0");
    for (int t = 0; t + 1 < T; ++t)
      if (hmc_t[t].n_slots != steps[t].n_slots || opts->hmc_workspace_bytes < gjx_hmc_workspace_bytes(&hmc_t[t], K))
        return gjx_fail(GJX_EINVAL, "gjx_scan_filter: hmc_targets[t] must have the rows of step t, and hmc_workspace must hold gjx_hmc_workspace_bytes of every target");
  }
  const bool no_fuse = (fflags & GJX_FILTER_TWO_LAUNCH) != 0 || hmc_t != nullptr;
  bool fused = room && !no_fuse && K % 1024 == 0 && K <= (1 << 20);
  hipStream_t st0 = (hipStream_t)stream;
  // the status word describes THIS call (a stale time-out bit would end a one-launch form at its first step)
  hipLaunchKernelGGL(k_clear_status, dim3(1), dim3(64), 0, st0, (unsigned*)ws_res + 8);
  GJX_CHECK_LAUNCH("gjx_scan_filter(status word)");
  if (fused && T > 1) {
    // the second run workspace's control block must be zero like the first one's (the caller zero-fills the workspace once; be safe)
    const hipError_t e = hipMemsetAsync(ws_run2, 0, kWsHeaderBytes, (hipStream_t)stream);
    if (e != hipSuccess) return gjx_fail_hip(e, "gjx_scan_filter(workspace)");
  }
  // key discipline of inference/pf.py: k_t = fold_in(k_{t-1}, t) (scan.py:268); (k_prop, k_res) = split(k_t); comb offset = uniform(k_res)
  std::vector<uint32_t> keys, res_keys;
  std::vector<double> us;
  pf_step_keys_res(key0, key1, T, keys, us, res_keys);
  // multinomial resampling (GJX_FILTER_MULTINOMIAL): plain launches, the prefix sums of the fixed-point weights behind the workspace
  const bool multinomial = (fflags & GJX_FILTER_MULTINOMIAL) != 0;
  uint64_t* mn_cum = nullptr;
  if (multinomial) {
    if (n_moves > 0) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter: the rejuvenation move runs with systematic resampling (the one-launch filter kernel)");
    const size_t off = (need_run + need_res + 255) & ~(size_t)255;
    if (workspace_bytes < off + 8 * (size_t)K + 256) return gjx_fail(GJX_EWORKSPACE, "gjx_scan_filter: multinomial resampling needs 8 K + 256 bytes beyond OP_RUN + OP_RESAMPLE");
    mn_cum = (uint64_t*)((char*)workspace + off);
    fused = false;
  }
  auto input_rows = [](const gjx_program& p) {   // rows of the program's INPUT sites (they come first and in order)
    int n = 0;
    for (int j = 0; j < p.n_sites; ++j) if (p.sites[j].mode == GJX_MODE_INPUT) n += p.sites[j].dim;
    return n;
  };
  // GJX_FILTER_ABSOLUTE_INPUTS: an INPUT site's obs_off is the row of the previous step's WHOLE buffer (carried statics are read from
  // its INPUT rows, the carry from its own rows); otherwise the row among the previous step's own rows, which sit behind its inputs
  const bool abs_in = (fflags & GJX_FILTER_ABSOLUTE_INPUTS) != 0;
  auto in_base = [&](const gjx_program& p) { return abs_in ? 0 : input_rows(p); };
  if (abs_in && n_moves > 0) return gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter: no rejuvenation move with carried static inputs (GJX_FILTER_ABSOLUTE_INPUTS)");
  gjx_run_opts o;
  gjx_run_resample rs;
  gjx_run_info info = {0, 0, 0}, prev = {0, 0, 0};
  // ---- GJX_FILTER_FORM_WIDE: step 0 as a plain launch, then steps 1 .. T-1 in ONE launch of the filter kernel generated for the step
  //      program (gjx_gen_pf: the skeleton of gjx_pfcore.h, 16 waves per tile) ----
  const int64_t ntw = (K + 1023) / 1024;
  const size_t wide_off = (logw_off + sizeof(float) * (size_t)K + 255) & ~(size_t)255;
  const size_t wide_bytes = 256 + (16 * (size_t)kPfCorePad + 24) * (size_t)ntw + 8 * (size_t)ntw + 24 * (size_t)T + 64;
  if (room && T >= 2 && !(multinomial && n_moves > 0) && !(fflags & GJX_FILTER_NO_WIDE) && !no_fuse && ntw <= kPfHostMaxTiles && workspace_bytes >= wide_off + wide_bytes &&
      !gjx_plain_launches_forced() && steps[1].tab_dev && gen_pf_supported(&steps[1]) && (n_moves == 0 || gen_pf_moves_supported(&steps[1]))) {
    // the kernel flavour: | 512 with the rejuvenation move, | 1024 multinomial resampling by sorted uniforms (pf_core's MULTI)
    const int mv = (n_moves > 0 ? 512 : 0) | (multinomial ? 1024 : 0);
    bool same = true;
    for (int u = 2; u < T && same; ++u)
      same = steps[u].n_tab == steps[1].n_tab && steps[u].n_slots == steps[1].n_slots && input_rows(steps[u]) == input_rows(steps[1]) &&
             steps[u].tab_dev != nullptr && gen_pf_same_kernel(&steps[1], &steps[u]);
    if (!abs_in && input_rows(steps[1]) > steps[0].n_slots - input_rows(steps[0])) same = false;
    const size_t dyn = pf_core_dyn_lds((int)ntw, multinomial);
    int spl = 0, grid = 0;
    const int spls[5] = {1, 2, 4, 8, 16};
    for (int i = 0; i < 5 && same && !spl; ++i) {
      const int64_t g = (ntw + spls[i] - 1) / spls[i];
      // (ask for the cheapest geometry first: a kernel is compiled — hipRTC, cached on disk — only for a geometry that could fit)
      if (g > 2 * 1024) continue;
      const int cap = (opts && opts->coresident_blocks > 0) ? opts->coresident_blocks : gen_pf_resident_blocks(&steps[1], spls[i] | mv, dyn);
      if (cap <= 0) break;                                   // no such kernel (compile failure: the reason is in gjx_last_error)
      if (g <= cap) { spl = spls[i]; grid = (int)g; }
    }
    if (spl) {
      // step 0 (no carry to read): its program's own kernel; log-weights where the skeleton expects those of step 0
      float* lw_even = logw; float* lw_odd = logw2;
      float* lw0 = ((T - 1) & 1) ? lw_odd : lw_even;
      memset(&o, 0, sizeof(o));
      int rc = gjx_run_program_ex(&steps[0], keys[0], keys[1], K, 0, rows_of(0), nullptr, nullptr, lw0, nullptr, nullptr, nullptr, nullptr, K,
                                  ws_run, need_run, stream, &o, &info);
      if (rc) return report(rc);
      char* wa = (char*)workspace + wide_off;
      // [256 B control][aggA 64 nt][aggB 64 nt][bsum 12 nt][bmax 12 nt][ready 4 grid, padded][us 8 T][keys 8 T][tabs 8 T]
      unsigned long long* aggA = (unsigned long long*)(wa + 256);
      unsigned long long* aggB = aggA + (size_t)ntw * kPfCorePad;
      float* bsum = (float*)(aggB + (size_t)ntw * kPfCorePad);
      float* bmax = bsum + 3 * (size_t)ntw;
      unsigned* ready = (unsigned*)(bmax + 3 * (size_t)ntw);
      double* us_dev = (double*)(ready + 2 * (((size_t)grid + 1) / 2));
      uint32_t* keys_dev = (uint32_t*)(us_dev + T);
      const float** tabs_dev = (const float**)(keys_dev + 2 * (size_t)T);
      hipError_t e = hipMemsetAsync(wa, 0, 256 + (16 * (size_t)kPfCorePad + 24) * (size_t)ntw + 8 * (((size_t)grid + 1) / 2), st0);   // no stale granule may pass
      if (e != hipSuccess) return report(gjx_fail_hip(e, "gjx_scan_filter(workspace)"));
      std::vector<const float*> h_tabs((size_t)T, nullptr);
      for (int u = 0; u < T; ++u) h_tabs[u] = steps[u].tab_dev;
      // (kernel arguments carry these small arrays: nothing on the host has to outlive the call)
      if (multinomial) {
        // (the sorted-uniform resampler takes the resampling KEY of every step where the comb takes its offset: f.us[t] = the key's
        // two words as one 64-bit pattern)
        std::vector<double> kb((size_t)T, 0.0);
        for (int u = 0; u < T; ++u) { const uint64_t w = (uint64_t)res_keys[2 * u] | ((uint64_t)res_keys[2 * u + 1] << 32); memcpy(&kb[u], &w, 8); }
        if (int rcu = upload_words(us_dev, kb.data(), (size_t)T, st0)) return report(rcu);
      } else
      if (int rcu = upload_words(us_dev, us.data(), (size_t)T, st0)) return report(rcu);
      if (int rcu = upload_words(keys_dev, keys.data(), (size_t)T, st0)) return report(rcu);
      if (int rcu = upload_words(tabs_dev, h_tabs.data(), (size_t)T, st0)) return report(rcu);
      GenPfArgs ga;
      memset(&ga, 0, sizeof(ga));
      PfCoreArgs& c = ga.core;
      c.T = T; c.K = K; c.K_total = K; c.offset = 0; c.G = 1; c.rank = 0; c.nt = (int)ntw; c.NT = (int)ntw;
      c.lw_even = lw_even; c.lw_odd = lw_odd; c.aggA = aggA; c.aggB = aggB; c.bsum = bsum; c.bmax = bmax; c.ready = ready;
      c.peer_data = nullptr; c.peer_flag = nullptr; c.keys = keys_dev; c.us = us_dev; c.lse_steps = lse_steps;
      c.ancestors = ancestors_all ? nullptr : ancestors; c.ancestors_all = ancestors_all;
      c.ctrl = (unsigned*)wa + 8; c.log_k = (float)log((double)K); c.first_budget = kPollBudget; c.zero_ptr = nullptr; c.zero_n = 0;
      c.verify = 0; c.chk_a = nullptr; c.chk_b = nullptr;
      c.timeline = (opts && opts->timeline && opts->timeline_bytes >= (int64_t)(128 * (size_t)grid)) ? (unsigned long long*)opts->timeline : nullptr;
      ga.tabs = tabs_dev;
      float* r0 = rows_of(0); float* r1 = rows_of(1);
      if (T > 2 && rows_of(2) == r0) { ga.rows_a = r0; ga.rows_b = r1; ga.rows_all = nullptr; ga.rows_step = 0; }
      else if (T == 2) { ga.rows_a = r0; ga.rows_b = r1; ga.rows_all = nullptr; ga.rows_step = 0; }
      else { ga.rows_a = nullptr; ga.rows_b = nullptr; ga.rows_all = r0; ga.rows_step = (int64_t)(r1 - r0); }
      ga.in_row0_first = (int64_t)in_base(steps[0]) * K;
      ga.in_row0 = (int64_t)in_base(steps[1]) * K;
      ga.n_moves = n_moves; ga.move_scale = opts ? opts->move_scale : 0.0f; ga.acc_total = (opts && n_moves > 0) ? (unsigned long long*)opts->accepted_total : nullptr;
      if (ga.acc_total) {
        const hipError_t ez = hipMemsetAsync(ga.acc_total, 0, sizeof(unsigned long long), st0);
        if (ez != hipSuccess) return report(gjx_fail_hip(ez, "gjx_scan_filter(accept counter)"));
      }
      rc = gen_pf_launch(&steps[1], spl | mv, ga, grid, dyn, st0);
      if (rc == GJX_OK) {
        // the skeleton's status bits live in ITS control block: fold them into the word the caller reads
        hipLaunchKernelGGL(k_merge_status, dim3(1), dim3(64), 0, st0, (unsigned*)wa + 8, (unsigned*)ws_res + 8);
        GJX_CHECK_LAUNCH("gjx_scan_filter(status)");
        finfo.form = GJX_FILTER_FORM_WIDE; finfo.launches = 2; finfo.grid = grid; finfo.tiles_per_block = spl;
        return report(GJX_OK);
      }
      if (rc != GJX_EUNSUPPORTED) return report(rc);
      // (the kernel could not be launched: the forms below start again from step 0)
    }
  }
  if (n_moves > 0 && T >= 2)     // (T < 2: there is no resampling, hence nothing to move: the plain forms below are the whole run)
    return report(gjx_fail(GJX_EUNSUPPORTED, "gjx_scan_filter: the rejuvenation move runs inside the filter kernel on the shared skeleton only "
                                             "(GJX_FILTER_FORM_WIDE: periodic step programs whose latent choices are the carry, no plates, a co-resident "
                                             "grid, the workspace room of the one-launch forms)"));
  // fused form: step t writes the log-weights / block pairs / tile totals of parity (T - 1 - t) & 1, so that the last step's land
  // in `logw` and in the first run workspace; the two-launch form uses one buffer throughout
  auto lw_of = [&](int t) { return (fused && ((T - 1 - t) & 1)) ? logw2 : logw; };
  auto ws_of = [&](int t) { return (fused && ((T - 1 - t) & 1)) ? ws_run2 : ws_run; };
  for (int t = 0; t < T; ++t) {
    const gjx_program& pr = steps[t];
    float* out = rows_of(t);
    const float* in = t > 0 ? rows_of(t - 1) : nullptr;
    memset(&o, 0, sizeof(o));
    o.flags = GJX_RUN_LEAVE_TILES;
    int rc = GJX_EUNSUPPORTED;
    bool ran = false;
    if (t > 0) {
      if (!abs_in && input_rows(pr) > steps[t - 1].n_slots - input_rows(steps[t - 1]))
        return gjx_fail(GJX_EINVAL, "gjx_scan_filter: a step reads more carry rows than the step before it produced");
      const char* pws = ws_of(t - 1);
      const float* lw_prev = lw_of(t - 1);
      const bool tiles = prev.tiles_offset != 0;
      const uint64_t* tS = tiles ? (const uint64_t*)(pws + prev.tiles_offset) : nullptr;
      const int32_t* tE = tiles ? (const int32_t*)(tS + (K / 1024)) : nullptr;
      int32_t* anc_t = ancestors_all ? ancestors_all + (size_t)(t - 1) * (size_t)K : ancestors;
      o.in_rows = in + (size_t)in_base(steps[t - 1]) * (size_t)K;      // the rows the previous step's OWN sites wrote (abs_in: its whole buffer)
      o.in_stride = K;
      if (fused && tiles) {
        memset(&rs, 0, sizeof(rs));
        rs.logw = lw_prev; rs.tile_S = tS; rs.tile_E = tE; rs.lse_partials = (const float*)(pws + kWsHeaderBytes); rs.n_partials = prev.n_partials;
        rs.lse_out = lse_steps + 4 * (size_t)(t - 1); rs.u = us[t]; rs.ancestors_out = anc_t; rs.status_ws = ws_res;
        o.resample = &rs;
        rc = gjx_run_program_ex(&pr, keys[2 * t], keys[2 * t + 1], K, 0, out, nullptr, nullptr, lw_of(t), nullptr, nullptr, nullptr, nullptr, K,
                                ws_of(t), need_run, stream, &o, &info);
        o.resample = nullptr;
        if (rc != GJX_OK && rc != GJX_EUNSUPPORTED) return rc;
        ran = rc == GJX_OK;
      }
      if (!ran) {
        // a step whose kernel cannot resample in its prologue: this step and the rest in the two-launch form, which writes one
        // buffer throughout (this step still reads what step t - 1 left where it left it)
        fused = false;
        if (multinomial) {
          uint64_t* bt = mn_cum + K;               // {0, total}
          rc = gjx_weight_cumsum(lw_prev, K, 2, (const float*)(pws + kWsHeaderBytes), prev.n_partials, mn_cum, bt, lse_steps + 4 * (size_t)(t - 1), K,
                                 ws_res, need_res, stream);
          if (rc) return rc;
          // (the call above: the finished LSE record of step t - 1 from the run's block pairs; its prefix sums are overwritten)
          rc = gjx_resample_sorted_multinomial_tiled(lw_prev, K, res_keys[2 * t], res_keys[2 * t + 1], K, anc_t, mn_cum, nullptr, nullptr, ws_res, need_res, stream);
          finfo.launches += 5;
        } else
        rc = gjx_resample_gather_tiled(lw_prev, K, tS, tE, 2, (const float*)(pws + kWsHeaderBytes), prev.n_partials, us[t], nullptr, 0, 0, nullptr, 0,
                                       anc_t, lse_steps + 4 * (size_t)(t - 1), K, ws_res, need_res, stream);
        if (rc) return rc;
        o.in_ancestors = anc_t;
        if (hmc_t) {
          // the resampled particle of step t - 1 — [its ancestor's inputs | its latent choices] — gathered, moved, and handed to the step
          const gjx_program& tg = hmc_t[t - 1];
          rc = gjx_gather_rows(in, K, anc_t, K, tg.n_slots, opts->hmc_rows, K, stream);
          if (rc) return rc;
          uint32_t k1[2], k2[2];
          host_threefry2x32(keys[2 * t], keys[2 * t + 1], 0u, 0x6d6f7665u, k1);
          host_threefry2x32(k1[0], k1[1], 0u, 0u, k2);
          rc = gjx_hmc(&tg, k2[0], k2[1], K, 0, opts->hmc_eps, opts->hmc_L, 0, 1, opts->hmc_rows, opts->hmc_out, opts->hmc_out + K, opts->hmc_out + 2 * (size_t)K,
                       opts->hmc_workspace, opts->hmc_workspace_bytes, stream);
          if (rc) return rc;
          if (opts->accepted_total) {
            hipLaunchKernelGGL(k_count_flags, dim3((unsigned)((K + 255) / 256 < 64 ? (K + 255) / 256 : 64)), dim3(256), 0, st0, (const float*)(opts->hmc_out + 2 * (size_t)K), K,
                               (unsigned long long*)opts->accepted_total);
            GJX_CHECK_LAUNCH("gjx_scan_filter(accepted chains)");
          }
          o.in_rows = opts->hmc_rows + (size_t)input_rows(steps[t - 1]) * (size_t)K;
          o.in_ancestors = nullptr;
          o.flags |= GJX_RUN_STORE_INPUTS;
          finfo.launches += 3;
        }
      }
    }
    if (!ran) {
      rc = gjx_run_program_ex(&pr, keys[2 * t], keys[2 * t + 1], K, 0, out, nullptr, nullptr, lw_of(t), nullptr, nullptr, nullptr, nullptr, K,
                              ws_of(t), need_run, stream, &o, &info);
      if (rc) return rc;
    }
    prev = info;
    // ---- steps 2 .. T-1 in ONE launch (gjx_gen_steps) when step 1 ran with the search in its prologue, the remaining step programs are
    //      the same kernel (a periodic Scan: they differ in tables, keys, comb offsets), the grid of K / 1024 blocks is co-resident
    //      and the workspace has the room (granules, pair arrays, the per-step arguments) ----
    finfo.launches += ran ? 1 : (t > 0 ? 2 : 1);
    if (t == 1) finfo.form = ran ? GJX_FILTER_FORM_PER_STEP : GJX_FILTER_FORM_TWO_LAUNCH;
    if (t == 1 && ran && T >= 4 && steps_area && info.engine == 4 && prev.tiles_offset != 0 && prev.n_partials == (int)nt &&
        !(fflags & GJX_FILTER_NO_STEPS) && !abs_in && !gjx_plain_launches_forced()) {
      bool same = true;
      for (int u = 2; u < T && same; ++u)
        same = steps[u].n_tab == steps[1].n_tab && steps[u].n_slots == steps[1].n_slots && input_rows(steps[u]) == input_rows(steps[1]) &&
               steps[u].tab_dev != nullptr && gen_same_kernel(&steps[1], &steps[u], 4);
      // (a grid the device cannot hold at once runs with as many blocks as are resident, each taking several tiles of a step in turn:
      // a block waits only at the top of a step, for granules every block publishes before it waits itself)
      const int64_t resident = !same ? 0 : ((opts && opts->coresident_blocks > 0) ? (int64_t)opts->coresident_blocks : (int64_t)gen_steps_resident_blocks(&steps[1], 4));
      const int64_t grid_steps = resident >= nt ? nt : resident;
      if (grid_steps > 0 && nt <= 4 * grid_steps) {
        hipStream_t st = (hipStream_t)stream;
        unsigned long long* gran_a = (unsigned long long*)steps_area;           // even steps
        unsigned long long* gran_b = gran_a + nt * kLiveGranulePad;
        unsigned long long* part_a = gran_b + nt * kLiveGranulePad;
        unsigned long long* part_b = part_a + nt;
        const float** tabs_dev = (const float**)(part_b + nt);
        uint32_t* keys_dev = (uint32_t*)(tabs_dev + T);
        double* us_dev = (double*)(keys_dev + 2 * (size_t)T);
        // (the per-step arguments travel as kernel arguments: nothing on the host has to outlive the call, no per-thread staging)
        std::vector<const float*> h_tabs((size_t)T, nullptr);
        for (int u = 0; u < T; ++u) h_tabs[u] = steps[u].tab_dev;
        hipError_t e = hipMemsetAsync(gran_a, 0, 16 * (size_t)kLiveGranulePad * (size_t)nt, st);
        if (e != hipSuccess) return report(gjx_fail_hip(e, "gjx_scan_filter(step arguments)"));
        if (int rcu = upload_words(tabs_dev, h_tabs.data(), (size_t)T, st)) return report(rcu);
        if (int rcu = upload_words(keys_dev, keys.data(), (size_t)T, st)) return report(rcu);
        if (int rcu = upload_words(us_dev, us.data(), (size_t)T, st)) return report(rcu);
        const char* w1 = ws_of(1);
        const uint64_t* tS = (const uint64_t*)(w1 + prev.tiles_offset);
        hipLaunchKernelGGL(k_tiles_to_granules, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, tS, (const int32_t*)(tS + nt),
                           (const unsigned long long*)(w1 + kWsHeaderBytes), gran_b, part_b, (int)nt, (unsigned long long)(1u % 15u) + 1ull);
        GJX_CHECK_LAUNCH("gjx_scan_filter(granules of step 1)");
        GenStepsArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.base.K = K; sa.base.offset = 0; sa.base.log_k_total = (float)log((double)K);
        sa.T0 = 2; sa.T = T; sa.tabs = tabs_dev; sa.keys = keys_dev; sa.us = us_dev;
        float* r0 = rows_of(0); float* r1 = rows_of(1); float* r2 = rows_of(2);
        if (r2 == r0) { sa.rows_a = r0; sa.rows_b = r1; sa.rows_all = nullptr; sa.rows_step = 0; }
        else { sa.rows_a = nullptr; sa.rows_b = nullptr; sa.rows_all = r0; sa.rows_step = (int64_t)(r1 - r0); }
        sa.in_row0_first = sa.in_row0 = (int64_t)input_rows(steps[1]) * K;
        sa.logw_a = logw; sa.logw_b = logw2;
        sa.gran_a = gran_a; sa.gran_b = gran_b; sa.part_a = part_a; sa.part_b = part_b;
        sa.lse_steps = lse_steps; sa.anc = ancestors; sa.anc_all = ancestors_all; sa.ctrl = (unsigned*)ws_res + 8; sa.epoch = 0u;
        sa.timeline = gjx::debug_timeline(128 * (size_t)grid_steps);
        const int rc2 = gen_steps_launch(&steps[1], 4, sa, (int)grid_steps, st);
        if (rc2 == GJX_OK) {
          finfo.form = GJX_FILTER_FORM_STEPS; finfo.launches += 3; finfo.grid = (int)grid_steps; finfo.tiles_per_block = (int)((nt + grid_steps - 1) / grid_steps);
          return report(gjx_launch_lse_finish(((T - 1) & 1) ? part_b : part_a, (int)nt, K, lse_steps + 4 * (size_t)(T - 1), st));
        }
        if (rc2 != GJX_EUNSUPPORTED) return report(rc2);
      }
    }
  }
  // the record of the last step: its block pairs are still in its run workspace
  finfo.launches += 1;
  return report(gjx_launch_lse_finish(ws_of(T - 1) + kWsHeaderBytes, prev.n_partials, K, lse_steps + 4 * (size_t)(T - 1), (hipStream_t)stream));
}

extern "C" int gjx_scan_filter(const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, int64_t K, float* rows_a, float* rows_b,
                               float* logw, int32_t* ancestors, int32_t* ancestors_all, float* lse_steps, void* workspace,
                               size_t workspace_bytes, void* stream, const gjx_filter_opts* opts, gjx_filter_info* info_out) {
  if (!steps || T < 1 || K <= 0 || !rows_a || !rows_b || !logw || !ancestors || !lse_steps)
    return gjx_fail(GJX_EINVAL, "gjx_scan_filter: bad argument");
  const gjx_plain_launch_scope plain_scope(opts && (opts->flags & GJX_FILTER_NO_ONE_LAUNCH) == GJX_FILTER_NO_ONE_LAUNCH);
  return scan_filter_impl(steps, T, key0, key1, K, [&](int t) { return (t & 1) ? rows_b : rows_a; }, logw, ancestors, ancestors_all, lse_steps,
                          workspace, workspace_bytes, stream, opts, info_out);
}

// the same run with the choices of EVERY step kept (rows_all f32[T][rows_per_step][K]) and every resampling's ancestors: what a
// trajectory reconstruction needs (the reference's ScanTrace stacks the whole trace per particle, scan.py:56-97)
extern "C" int gjx_scan_filter_history(const gjx_program* steps, int32_t T, uint32_t key0, uint32_t key1, int64_t K, float* rows_all,
                                       int32_t rows_per_step, float* logw, int32_t* ancestors_all, float* lse_steps, void* workspace,
                                       size_t workspace_bytes, void* stream, const gjx_filter_opts* opts, gjx_filter_info* info_out) {
  if (!steps || T < 1 || K <= 0 || !rows_all || rows_per_step < 1 || !logw || (T > 1 && !ancestors_all) || !lse_steps)
    return gjx_fail(GJX_EINVAL, "gjx_scan_filter_history: bad argument");
  for (int t = 0; t < T; ++t)
    if (steps[t].n_slots > rows_per_step) return gjx_fail(GJX_EINVAL, "gjx_scan_filter_history: a step has more rows than rows_per_step");
  int32_t* anc = ancestors_all ? ancestors_all : (int32_t*)rows_all;     // (T == 1: never written)
  const gjx_plain_launch_scope plain_scope(opts && (opts->flags & GJX_FILTER_NO_ONE_LAUNCH) == GJX_FILTER_NO_ONE_LAUNCH);
  return scan_filter_impl(steps, T, key0, key1, K, [&](int t) { return rows_all + (size_t)t * (size_t)rows_per_step * (size_t)K; }, logw, anc,
                          ancestors_all, lse_steps, workspace, workspace_bytes, stream, opts, info_out);
}
