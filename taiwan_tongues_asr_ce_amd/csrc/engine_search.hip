// libttasr: logits-processor rules, kernel-selection options, greedy / sampled generation and beam search (one of the engine
// translation units, see engine_ctx.hpp).
#include "engine_ctx.hpp"
#include <chrono>

namespace ttasr_detail {

int check_ready(ttasr_ctx* c, int B) {
  if (!c) return TTASR_E_INVALID;
  if (!c->finalized) return fail(c, TTASR_E_INVALID, "weights not finalized (call ttasr_finalize_weights first)");
  if (B < 1 || B > c->maxB) return fail(c, TTASR_E_INVALID, "batch %d outside [1, max_batch=%d]", B, c->maxB);
  HIPCHK(c, hipSetDevice(c->device));
  return 0;
}

int upload_rules(ttasr_ctx* c, const ttasr_gen_opts* o, int max_prompt) {
  if (!o) return fail(c, TTASR_E_INVALID, "opts is NULL");
  if (o->max_new_tokens < 1 || o->max_new_tokens > c->max_new_alloc)
    return fail(c, TTASR_E_INVALID, "max_new_tokens %d outside [1, %d]", o->max_new_tokens, c->max_new_alloc);
  std::vector<uint8_t> mask(c->V, 0);
  for (int i = 0; i < o->n_suppress; ++i) {
    int t = o->suppress[i];
    if (t < 0 || t >= c->V) return fail(c, TTASR_E_INVALID, "suppress id %d outside vocabulary", t);
    mask[t] |= 1;
  }
  for (int i = 0; i < o->n_begin_suppress; ++i) {
    int t = o->begin_suppress[i];
    if (t < 0 || t >= c->V) return fail(c, TTASR_E_INVALID, "begin_suppress id %d outside vocabulary", t);
    mask[t] |= 2;
  }
  HIPCHK(c, hipMemcpyAsync(c->mask_dev, mask.data(), c->V, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  RuleParams& rp = c->rp;
  rp.V = c->V; rp.ldv = c->ldv; rp.max_prompt = max_prompt; rp.max_new = o->max_new_tokens;
  rp.eot = o->eot; rp.no_timestamps = o->no_timestamps; rp.timestamp_begin = o->timestamp_begin;
  rp.no_speech = o->no_speech; rp.sot_index = o->sot_index; rp.timestamps = o->timestamps;
  rp.max_initial = o->max_initial_timestamp_index; rp.suppress_eot = o->suppress_eot;
  rp.temperature = 0.f; rp.seed = 0;
  if (rp.eot < 0 || rp.eot >= c->V || rp.timestamp_begin < 0 || rp.timestamp_begin > c->V)
    return fail(c, TTASR_E_INVALID, "special token ids outside vocabulary");
  return 0;
}

// After the rules of a call are known: the per-window scalars go to device memory (stream-ordered in front of the decode
// launches); the captured mode 0 / 2 graphs are dropped only when a BAKED scalar changed.
int commit_rules(ttasr_ctx* c, const RuleParams& old) {
  c->rule_dyn_host = RuleDyn{c->rp.max_prompt, c->rp.max_new, c->rp.sot_index, c->rp.seed};
  HIPCHK(c, hipMemcpyAsync(c->rule_dyn_dev, &c->rule_dyn_host, sizeof(RuleDyn), hipMemcpyHostToDevice, c->stream));
  RuleParams a = old;
  a.max_prompt = c->rp.max_prompt; a.max_new = c->rp.max_new; a.sot_index = c->rp.sot_index; a.seed = c->rp.seed;
  if (memcmp(&a, &c->rp, sizeof a) != 0) drop_rule_graphs(c);
  return 0;
}

// Kernel-selection overrides (ttasr_set_option).  Everything defaults to the measured configuration; an override changes
// which kernels the captured decode graphs hold, so the graphs are dropped.
int set_option(ttasr_ctx* c, const std::string& key, int v) {
  const bool on = v != 0;
  if (key == "enc_kernel_timing") { c->enc_timing = on; return 0; }   // measurement only: the captured decode graphs stay
  if (key == "flash") c->no_flash = !on;
  else if (key == "prefill") c->no_prefill = !on;
  else if (key == "vocab_persistent") c->vocab_persistent = on;
  else if (key == "xsplit") c->no_xsplit = !on;
  else if (key == "graph") c->use_graph = on;
  else if (key == "multi_step_graph") c->multi_step = on;
  else if (key == "generic_kernels") c->force_basic = on;
  else if (key == "prefill_tiled") c->prefill_tiled = on;
  else if (key == "prefill_ns_min") { if (v < 0) return 1; c->prefill_ns_min = v; }
  else if (key == "enc_residual_epilogue") c->enc_res_epilogue = on;
  else if (key == "enc_gemm") { if (v < 0 || v > 4) return 1; c->gemm_force = v; }
  else if (key == "enc_gemm_persistent") c->gemm_persistent = on;
  else if (key == "xkv_grouped") c->xkv_grouped = on;
  else if (key == "enc_gemm_tail") c->gemm_tail = on;
  else if (key == "enc_ln_defer") c->enc_ln_defer = on;
  else if (key == "dec_narrow_blocks") { if (c->weights_packed && on != c->dec_narrow) return 1; c->dec_narrow = on; }   // a layout choice: before the first weight arrives
  else if (key == "ksplit_out") { if (v < 0 || v > 16) return 1; c->ks_want[0] = v; }
  else if (key == "ksplit_q") { if (v < 0 || v > 16) return 1; c->ks_want[1] = v; }
  else if (key == "ksplit_qkv") { if (v < 0 || v > 16) return 1; c->ks_want[2] = v; }
  else if (key == "ksplit_fc2") { if (v < 0 || v > 16) return 1; c->ks_want[3] = v; }
  else if (key == "xattn_nontemporal") c->xattn_nt = on ? 1 : 0;   // per context (kernel template choice)
  else if (key == "xattn_pipeline") c->xattn_pipe = on ? 1 : 0;
  else if (key == "xkv_fp8") {
    if (on && !c->lowp) return 1;   // 16-bit engines only
    if (on && (!c->xkv8 || !c->xkv8_scale)) {   // the small allocation first; the mode turns on only when BOTH exist
      const size_t n = (size_t)c->cfg.dec_layers * c->xkv_layer_elems;
      if (!c->xkv8_scale && dalloc(c, &c->xkv8_scale, (size_t)c->cfg.dec_layers * 2 * c->maxB * c->H * sizeof(float)) != 0) return 2;
      if (!c->xkv8 && dalloc(c, &c->xkv8, n, false) != 0) return 2;   // 2: dalloc's error text (out of memory) stands
    }
    c->xkv_fp8 = on; c->xkv8_valid = false;   // the e4m3 copy is (re)built by the next encode
  }
  else if (key == "weights_nontemporal") c->weights_nt = on ? 1 : 0;
  else if (key == "dec_x_lds") c->dec_x_lds = on;
  else if (key == "ragged_exit") c->ragged_exit = on;
  else if (key == "flash_qw") { if (v < 1 || v > 2) return 1; c->flash_qw = v; }
  else if (key == "xattn_mq_slices") { if (v < 0 || v > 8) return 1; c->xattn_mq_slices = v; }
  else if (key == "xattn_deep_items") { if (v < 0 || v > 1 << 20) return 1; c->xattn_deep_items = v; }
  else return 1;
  g_xattn_variant = c->xattn_nt | (c->xattn_pipe << 1); g_skinny_nt = c->weights_nt; g_skinny_narrow = c->dec_narrow ? 1 : 0;
  g_skinny_x_lds = c->dec_x_lds ? 1 : 0;
  g_xattn_deep_items = c->xattn_deep_items; g_xattn_mq_slices = c->xattn_mq_slices; g_flash_qw = c->flash_qw;
  drop_graphs(c);
  return 0;
}

int reset_search(ttasr_ctx* c, int B) {
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemsetAsync(c->st.step, 0, 16, s));
  HIPCHK(c, hipMemsetAsync(c->st.n_sampled, 0, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->st.last_tok, 0xff, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->st.pen_tok, 0xff, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->st.last_ts, 0xff, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->st.done, 0, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->st.n_done, 0, 16, s));
  HIPCHK(c, hipMemsetAsync(c->st.sum_logprob, 0, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->st.no_speech, 0, B * 4, s));
  HIPCHK(c, hipMemsetAsync(c->row_cap_dev, 0x7f, B * 4, s));   // no per-row token budget (ttasr_generate_capped uploads its own)
  return 0;
}

// shared by ttasr_generate and ttasr_generate_sample: R rows, row r uses prompt (r / rows_per_clip)
int generate_rows(ttasr_ctx* c, int R, int rows_per_clip, const int32_t* prompt, const int32_t* prompt_len, int max_prompt,
                  const ttasr_gen_opts* o, float temperature, uint32_t seed, int32_t* out_tokens, int32_t* out_len, float* out_lp,
                  float* out_ns, const int32_t* row_cap) {
  int min_plen = 1 << 30, max_plen = 0;
  const int A = R / rows_per_clip;
  for (int a = 0; a < A; ++a) {
    if (prompt_len[a] < 1 || prompt_len[a] > max_prompt) return fail(c, TTASR_E_INVALID, "prompt_len[%d]=%d", a, prompt_len[a]);
    if (prompt_len[a] >= c->cfg.n_text_ctx)
      return fail(c, TTASR_E_INVALID, "prompt_len[%d]=%d leaves no room in the %d-token context", a, prompt_len[a], c->cfg.n_text_ctx);
    min_plen = std::min(min_plen, prompt_len[a]); max_plen = std::max(max_plen, prompt_len[a]);
    for (int j = 0; j < prompt_len[a]; ++j)
      if (prompt[a * max_prompt + j] < 0 || prompt[a * max_prompt + j] >= c->V)
        return fail(c, TTASR_E_INVALID, "prompt token outside vocabulary");
  }
  RuleParams old = c->rp;
  TRY(upload_rules(c, o, max_prompt));
  c->rp.temperature = temperature; c->rp.seed = seed;
  TRY(commit_rules(c, old));
  TRY(reset_search(c, R));
  hipStream_t s = c->stream;
  std::vector<int32_t> pr((size_t)R * max_prompt, 0), pl(R);
  for (int r = 0; r < R; ++r) {
    const int a = r / rows_per_clip;
    pl[r] = prompt_len[a];
    memcpy(&pr[(size_t)r * max_prompt], &prompt[(size_t)a * max_prompt], (size_t)max_prompt * 4);
  }
  HIPCHK(c, hipMemcpyAsync(c->prompt_dev, pr.data(), pr.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->plen_dev, pl.data(), R * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpy2DAsync(c->st.cur_tok, 4, c->prompt_dev, (size_t)max_prompt * 4, 4, R, hipMemcpyDeviceToDevice, s));
  int max_cap = o->max_new_tokens;
  if (row_cap) {   // per-row token budgets: row r is finished after min(row_cap[r], max_new_tokens) sampled tokens (or EOT)
    max_cap = 1;
    for (int r = 0; r < R; ++r) {
      if (row_cap[r] < 1 || row_cap[r] > o->max_new_tokens)
        return fail(c, TTASR_E_INVALID, "row_max_new[%d]=%d outside [1, max_new_tokens=%d]", r, row_cap[r], o->max_new_tokens);
      max_cap = std::max(max_cap, (int)row_cap[r]);
    }
    HIPCHK(c, hipMemcpyAsync(c->row_cap_dev, row_cap, (size_t)R * 4, hipMemcpyHostToDevice, s));
  }
  HIPCHK(c, hipStreamSynchronize(s));  // pr / pl (and the caller's row_cap) are read by the copies above
  c->st.prompt = c->prompt_dev; c->st.prompt_len = c->plen_dev;
  c->B_dec = R;
  c->kv_div = rows_per_clip;
  struct Restore { ttasr_ctx* c; ~Restore() { c->kv_div = 1; } } restore{c};
  const int interval = std::max(1, o->check_interval);
  // exclusive; prompt + sampled tokens never exceed n_text_ctx (the reference's max_length = 448: the token sampled
  // from position n_text_ctx - 2 is the last one, position n_text_ctx - 1 is never fed)
  // (with per-row budgets no row samples past the largest of them: the loop ends there, no host poll needed to find out)
  const int last_step = std::min(c->cfg.n_text_ctx - 1, max_plen - 1 + max_cap);
  hipEventRecord(c->ev[5], s);
  // A prefill pass runs the encoder-side GEMM kernels on rows x positions; for a handful of positions that costs more
  // than the decode steps it replaces (measured at large-v3, 3 positions x 32 rows: +5 ms), so the
  // <|startoftranscript|> position is only folded into the prefill when the prompt is long (previous-text prompts)
  const int pre = prefill_positions(c, min_plen, o, /*ns_from_prefill=*/min_plen - 1 >= c->prefill_ns_min);
  if (pre > 0) {  // positions 0..pre-1 of every row in one batched pass; the step loop resumes at position `pre`
    sched_prefill(c, R, pre, rows_per_clip, max_prompt);
    if (o->no_speech >= 0 && o->sot_index < pre) {  // the <|startoftranscript|> position was prefilled: its logits come from here
      TRY(sched_prefill_no_speech(c, R, pre, o->sot_index, o->no_speech));
    }
    c->pinned_i32[1] = pre;
    HIPCHK(c, hipMemcpyAsync(c->st.step, &c->pinned_i32[1], 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpy2DAsync(c->st.cur_tok, 4, c->prompt_dev + pre, (size_t)max_prompt * 4, 4, R, hipMemcpyDeviceToDevice, s));
  }
  for (int step = pre; step < last_step; ++step) {
    const bool all_forced = step + 1 < min_plen;
    const bool need_logits = !all_forced || (o->no_speech >= 0 && step == o->sot_index);
    // runs of sampled steps up to (and including) the next host poll replay as ONE multi-step graph of 8 or 4 steps
    int run = 1;
    if (c->multi_step && need_logits && step + 1 >= min_plen) {
      int until_poll = last_step - step;                       // steps left
      if (!o->suppress_eot) until_poll = std::min(until_poll, interval - (step + 1 - min_plen) % interval);
      run = until_poll >= 8 ? 8 : (until_poll >= 4 ? 4 : 1);
    }
    TRY(step_graph(c, R, need_logits ? 0 : 2, run));
    step += run - 1;
    if (!o->suppress_eot && step + 1 >= min_plen && ((step + 1 - min_plen) % interval == interval - 1)) {
      HIPCHK(c, hipMemcpyAsync(c->pinned_i32, c->st.n_done, 4, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
      if (c->pinned_i32[0] >= R) break;
    }
  }
  hipEventRecord(c->ev[6], s);
  HIPCHK(c, hipMemcpyAsync(out_tokens, c->st.out_tokens, (size_t)R * c->rp.max_new * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(out_len, c->st.n_sampled, R * 4, hipMemcpyDeviceToHost, s));
  if (out_lp) HIPCHK(c, hipMemcpyAsync(out_lp, c->st.sum_logprob, R * 4, hipMemcpyDeviceToHost, s));
  if (out_ns) HIPCHK(c, hipMemcpyAsync(out_ns, c->st.no_speech, R * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemsetAsync(c->st.done, 0, (size_t)R * 4, s));   // the flags belong to THIS search: later step-API calls see live rows
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  hipEventElapsedTime(&c->phase_ms[3], c->ev[5], c->ev[6]);
  for (int r = 0; r < R; ++r) out_len[r] = std::min(out_len[r], c->rp.max_new);
  return TTASR_OK;
}

// Beam search over A clips x `beam` rows.  Prompts may be ragged: clip a has plens[a] tokens at prompt + a * max_prompt and
// its <|startoftranscript|> at sots[a]; the step loop is position-synchronous, so at a given position some clips are
// still being forced through their prompt while others already search.
int beam_search_impl(ttasr_ctx* c, int32_t A, int32_t beam, const int32_t* prompt, int32_t max_prompt, const int32_t* plens,
                            const int32_t* sots, const ttasr_gen_opts* o, float patience, int32_t* out_tokens, int32_t* out_len,
                            float* out_lp, float* out_ns) {
  if (!c) return TTASR_E_INVALID;
  if (beam < 1 || beam > 7 || A < 1) return fail(c, TTASR_E_INVALID, "beam must be 1..7 and n_audio >= 1");
  const int R = A * beam;
  TRY(check_ready(c, R));
  if (!prompt || !plens || !out_tokens || !out_len || !o) return fail(c, TTASR_E_INVALID, "NULL argument");
  if (A > c->B_enc) return fail(c, TTASR_E_INVALID, "encoder state holds %d clips, %d requested", c->B_enc, A);
  if (max_prompt < 1 || max_prompt > c->max_prompt_alloc) return fail(c, TTASR_E_INVALID, "max_prompt %d", max_prompt);
  int min_plen = 1 << 30, min_sot = 1 << 30;
  for (int a = 0; a < A; ++a) {
    if (plens[a] < 1 || plens[a] > max_prompt || plens[a] >= c->cfg.n_text_ctx) return fail(c, TTASR_E_INVALID, "prompt_len[%d]=%d", a, plens[a]);
    const int sot = sots ? sots[a] : o->sot_index;
    if (o->no_speech >= 0 && out_ns && (sot < 0 || sot >= plens[a])) return fail(c, TTASR_E_INVALID, "sot_index[%d]=%d outside the prompt", a, sot);
    min_plen = std::min(min_plen, (int)plens[a]); min_sot = std::min(min_sot, sot);
    for (int i = 0; i < plens[a]; ++i)
      if (prompt[(size_t)a * max_prompt + i] < 0 || prompt[(size_t)a * max_prompt + i] >= c->V)
        return fail(c, TTASR_E_INVALID, "prompt token outside vocabulary");
  }
  auto sot_of = [&](int a) { return sots ? sots[a] : o->sot_index; };
  RuleParams old_rp = c->rp;
  TRY(upload_rules(c, o, max_prompt));
  TRY(commit_rules(c, old_rp));
  TRY(reset_search(c, R));
  c->st.prompt = nullptr; c->st.prompt_len = nullptr;
  c->B_dec = R;
  c->kv_div = beam; c->identity_pages = 0;
  struct Restore { ttasr_ctx* c; ~Restore() { c->kv_div = 1; c->identity_pages = 1; } } restore{c};
  hipStream_t s = c->stream;
  const int pps = c->pages_per_seq, n_pages = c->maxB * pps, max_new = c->rp.max_new, K = beam + 1;
  const int max_cand = std::max(1, (int)std::lround(beam * patience));
  std::vector<int32_t> tbl((size_t)R * pps, -1), refcnt(n_pages, 0), free_pages, cur_tok(R), pairs;
  std::vector<std::vector<int>> seqs(R);
  std::vector<double> sums(R, 0.0);
  std::vector<std::map<std::vector<int>, double>> finished(A);
  // the pinned exchange block (engine_ctx.hpp): [page tables R x pps | fed tokens R | row histories 4 R | page pairs 2 R | done R]
  // out, [log-probs R x K | ids R x K | no-speech R] back
  const size_t n_up = (size_t)R * pps, o_tok = n_up, o_state = o_tok + R, o_pairs = o_state + 4 * (size_t)R, o_done = o_pairs + 2 * (size_t)R,
               o_lp = o_done + R, o_id = o_lp + (size_t)R * K, o_ns = o_id + (size_t)R * K, n_words = o_ns + R;
  if (c->pinned_beam_bytes < n_words * 4) {
    if (c->pinned_beam) hipHostFree(c->pinned_beam);
    c->pinned_beam = nullptr; c->pinned_beam_bytes = 0;
    const size_t want = (size_t)c->maxB * (pps + 8 + 2 * 8 + 1) * 4 + 4096;     // the largest search this context can run
    HIPCHK(c, hipHostMalloc((void**)&c->pinned_beam, std::max(want, n_words * 4)));
    c->pinned_beam_bytes = std::max(want, n_words * 4);
  }
  int32_t* const pb = (int32_t*)c->pinned_beam;
  int32_t *const p_up = pb, *const p_tok = pb + o_tok, *const h_state = pb + o_state, *const p_pairs = pb + o_pairs, *const p_done = pb + o_done,
          *const h_id = pb + o_id;
  float *const h_lp = (float*)(pb + o_lp), *const h_ns = (float*)(pb + o_ns);
  for (int r = 0; r < R; ++r) h_ns[r] = 0.f;
  using clk = std::chrono::steady_clock;
  double t_enq = 0, t_wait = 0, t_sel = 0; int n_pos = 0;
  auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
  auto rebuild_free = [&](int upto_idx) {
    std::fill(refcnt.begin(), refcnt.end(), 0);
    for (int r = 0; r < R; ++r)
      for (int j = 0; j <= upto_idx && j < pps; ++j)
        if (tbl[(size_t)r * pps + j] >= 0) refcnt[tbl[(size_t)r * pps + j]]++;
    free_pages.clear();
    for (int p = n_pages - 1; p >= 0; --p) if (refcnt[p] == 0) free_pages.push_back(p);
  };
  rebuild_free(-1);
  for (int r = 0; r < R; ++r) cur_tok[r] = prompt[(size_t)(r / beam) * max_prompt];
  std::vector<char> done(A, 0);
  std::vector<int32_t> done_rows(R, 0);   // device copy of `done`, one flag per row: finished clips leave the attention kernels (round 6)
  bool done_dirty = false;
  std::vector<float> ns_final(A, 0.f);
  hipEventRecord(c->ev[5], s);
  // Batched prompt prefill: the beam rows of a clip share one prompt, so its positions are computed ONCE per clip
  // into pages that all `beam` page tables then reference (the copy-on-write below splits the last, partially
  // filled page on the first private write).
  // every clip must still be inside its prompt (and before its <|startoftranscript|> when no-speech is wanted)
  ttasr_gen_opts o_pre = *o; o_pre.sot_index = min_sot;
  const int pre = prefill_positions(c, min_plen, &o_pre);
  if (pre > 0) {
    const int n_pg = (pre + 15) / 16;
    std::vector<int32_t> ptab((size_t)A * pps, 0);
    for (int a = 0; a < A; ++a)
      for (int q = 0; q < n_pg; ++q) {
        if (free_pages.empty()) return fail(c, TTASR_E_NOMEM, "KV page pool exhausted");
        const int32_t pg = free_pages.back(); free_pages.pop_back();
        for (int b = 0; b < beam; ++b) tbl[(size_t)(a * beam + b) * pps + q] = pg;
        ptab[(size_t)a * pps + q] = pg;
      }
    HIPCHK(c, hipMemcpyAsync(c->page_table, ptab.data(), ptab.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->prompt_dev, prompt, (size_t)A * max_prompt * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));  // ptab is a stack temporary
    sched_prefill(c, A, pre, 1, max_prompt);
    c->pinned_i32[1] = pre;
    HIPCHK(c, hipMemcpyAsync(c->st.step, &c->pinned_i32[1], 4, hipMemcpyHostToDevice, s));
    rebuild_free(n_pg - 1);
    for (int r = 0; r < R; ++r) cur_tok[r] = prompt[(size_t)(r / beam) * max_prompt + pre];
  }
  bool stop = false;
  for (int pos = pre; pos < c->cfg.n_text_ctx - 1 && !stop; ++pos) {
    // 1. the page this step writes must exist and be private to the row (copy-on-write after a re-index)
    auto t0 = clk::now();
    ++n_pos;
    const int j = pos / 16;
    pairs.clear();
    for (int r = 0; r < R; ++r) {
      int32_t& pg = tbl[(size_t)r * pps + j];
      if (pos % 16 == 0 || pg < 0) {
        if (free_pages.empty()) return fail(c, TTASR_E_NOMEM, "KV page pool exhausted");
        pg = free_pages.back(); free_pages.pop_back(); refcnt[pg] = 1;
      } else if (refcnt[pg] > 1) {
        if (free_pages.empty()) return fail(c, TTASR_E_NOMEM, "KV page pool exhausted");
        const int32_t np = free_pages.back(); free_pages.pop_back();
        pairs.push_back(pg); pairs.push_back(np);
        refcnt[pg]--; refcnt[np] = 1; pg = np;
      }
    }
    // per clip: still forced through its prompt, searching, or finished
    auto forced_next = [&](int a) { return prompt[(size_t)a * max_prompt + pos + 1]; };
    bool any_sampling = false, any_ns = false;
    for (int a = 0; a < A; ++a) {
      any_sampling |= !done[a] && pos + 1 >= plens[a];
      any_ns |= o->no_speech >= 0 && out_ns && pos == sot_of(a);
    }
    // everything the device needs for this position, staged in the pinned block and copied asynchronously: page tables ([row][pps],
    // unused entries clamped to a valid page id), the fed tokens, the copy-on-write pairs, finished flags, and - when a clip
    // searches at this position - the row histories the candidate kernel applies the rules from (host-known before the step)
    for (size_t i = 0; i < n_up; ++i) p_up[i] = tbl[i] < 0 ? 0 : tbl[i];
    memcpy(p_tok, cur_tok.data(), (size_t)R * 4);
    HIPCHK(c, hipMemcpyAsync(c->page_table, p_up, n_up * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->st.cur_tok, p_tok, (size_t)R * 4, hipMemcpyHostToDevice, s));
    if (!pairs.empty()) {
      memcpy(p_pairs, pairs.data(), pairs.size() * 4);
      HIPCHK(c, hipMemcpyAsync(c->pairs_dev, p_pairs, pairs.size() * 4, hipMemcpyHostToDevice, s));
      TT_DISPATCH(c, launch_copy_pages<T>((T*)c->pool, c->pairs_dev, (int)pairs.size() / 2, c->cfg.dec_layers, c->H, c->pool_layer_elems, s));
    }
    if (done_dirty) {
      memcpy(p_done, done_rows.data(), (size_t)R * 4);
      HIPCHK(c, hipMemcpyAsync(c->st.done, p_done, (size_t)R * 4, hipMemcpyHostToDevice, s));
      done_dirty = false;
    }
    if (any_sampling || any_ns) {
      for (int r = 0; r < R; ++r) {
        int last = -1, pen = -1, lts = -1;
        for (int t : seqs[r]) { pen = last; last = t; if (t >= o->timestamp_begin) lts = t; }
        h_state[r] = (int)seqs[r].size(); h_state[R + r] = last; h_state[2 * R + r] = pen; h_state[3 * R + r] = lts;
      }
      HIPCHK(c, hipMemcpyAsync(c->row_state, h_state, (size_t)4 * R * 4, hipMemcpyHostToDevice, s));
    }
    // 2. one decoder step over the R rows (logits only; the search itself runs on the host) and, behind it, the candidates
    TRY(step_graph(c, R, 1));
    if (any_sampling || any_ns) {
      BeamRowState bs{c->row_state, c->row_state + R, c->row_state + 2 * R, c->row_state + 3 * R, c->mask_dev};
      launch_beam_topk(c->logits, bs, c->rp, R, K, c->topk_lp, c->topk_id, any_ns ? c->st.no_speech : nullptr, s);
      HIPCHK(c, hipMemcpyAsync(h_lp, c->topk_lp, (size_t)R * K * 4, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipMemcpyAsync(h_id, c->topk_id, (size_t)R * K * 4, hipMemcpyDeviceToHost, s));
      if (any_ns) HIPCHK(c, hipMemcpyAsync(h_ns, c->st.no_speech, (size_t)R * 4, hipMemcpyDeviceToHost, s));
    }
    t_enq += ms_since(t0); t0 = clk::now();
    HIPCHK(c, hipStreamSynchronize(s));   // the ONE synchronisation of the position (also frees the pinned block for the next one)
    t_wait += ms_since(t0); t0 = clk::now();
    struct Sel { double& acc; clk::time_point t0; ~Sel() { acc += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); } } sel_timer{t_sel, t0};
    if (!any_sampling && !any_ns) {
      for (int r = 0; r < R; ++r) cur_tok[r] = done[r / beam] ? o->eot : forced_next(r / beam);
      continue;
    }
    for (int a = 0; a < A; ++a)
      if (o->no_speech >= 0 && out_ns && pos == sot_of(a)) ns_final[a] = h_ns[a * beam];
    if (!any_sampling) {
      for (int r = 0; r < R; ++r) cur_tok[r] = done[r / beam] ? o->eot : forced_next(r / beam);
      continue;
    }
    // 3. candidate selection per clip (Whisper BeamSearchDecoder semantics; identical sequences collapse)
    std::vector<std::vector<int>> nseq; std::vector<double> nsum; std::vector<int> src;
    for (int a = 0; a < A; ++a) {
      if (done[a] || pos + 1 < plens[a]) {  // not searching at this position: hypotheses and page lists carry over unchanged
        for (int b = 0; b < beam; ++b) { nseq.push_back(seqs[a * beam + b]); nsum.push_back(sums[a * beam + b]); src.push_back(a * beam + b); }
        continue;
      }
      std::map<std::vector<int>, std::pair<double, int>> cand;
      for (int b = 0; b < beam; ++b) {
        const int r = a * beam + b;
        for (int q = 0; q < K; ++q) {
          const int tok = h_id[(size_t)r * K + q];
          if (tok < 0) continue;
          std::vector<int> key(seqs[r]); key.push_back(tok);
          const double val = sums[r] + (double)h_lp[(size_t)r * K + q];
          auto it = cand.find(key);
          if (it == cand.end() || val > it->second.first) cand[key] = {val, r};
        }
      }
      std::vector<std::pair<double, const std::vector<int>*>> order;
      for (auto& kv : cand) order.push_back({kv.second.first, &kv.first});
      std::sort(order.begin(), order.end(), [](auto& x, auto& y) { return x.first != y.first ? x.first > y.first : *x.second < *y.second; });
      int saved = 0;
      std::vector<std::pair<double, const std::vector<int>*>> fin_new;
      for (auto& e : order) {
        if (e.second->back() == o->eot) { fin_new.push_back(e); continue; }
        nseq.push_back(*e.second); nsum.push_back(e.first); src.push_back(cand[*e.second].second);
        if (++saved == beam) break;
      }
      for (auto& e : fin_new) { if ((int)finished[a].size() >= max_cand) break; finished[a][*e.second] = e.first; }
      if (saved == 0) return fail(c, TTASR_E_INVALID, "beam search: no live candidate (every token masked)");
      while (saved < beam) { nseq.push_back(nseq.back()); nsum.push_back(-1e30); src.push_back(src.back()); ++saved; }
    }
    // 4. re-index: hypotheses inherit their parent's page list (shared pages; refcounts rebuilt)
    std::vector<int32_t> ntbl((size_t)R * pps, -1);
    for (int r = 0; r < R; ++r)
      for (int q = 0; q <= j; ++q) ntbl[(size_t)r * pps + q] = tbl[(size_t)src[r] * pps + q];
    tbl.swap(ntbl);
    rebuild_free(j);
    seqs.swap(nseq); sums.swap(nsum);
    bool all_done = true;
    for (int a = 0; a < A; ++a) {
      const bool searching = !done[a] && pos + 1 >= plens[a];
      if (searching && ((int)finished[a].size() >= max_cand || (int)seqs[a * beam].size() >= max_new)) {
        done[a] = 1; done_dirty = true;
        for (int b = 0; b < beam; ++b) done_rows[a * beam + b] = 1;
      }
      for (int b = 0; b < beam; ++b) {
        const int r = a * beam + b;
        cur_tok[r] = done[a] ? o->eot : (searching ? seqs[r].back() : forced_next(a));
      }
      all_done &= (bool)done[a];
    }
    if (all_done) stop = true;
  }
  hipEventRecord(c->ev[6], s);
  c->beam_prof_ms[0] = (float)t_enq; c->beam_prof_ms[1] = (float)t_wait; c->beam_prof_ms[2] = (float)t_sel; c->beam_prof_ms[3] = (float)n_pos;
  HIPCHK(c, hipMemsetAsync(c->st.done, 0, (size_t)R * 4, s));   // the flags belong to THIS search
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  hipEventElapsedTime(&c->phase_ms[3], c->ev[5], c->ev[6]);
  for (int a = 0; a < A; ++a) {
    std::map<std::vector<int>, double> pool(finished[a]);
    if ((int)pool.size() < beam) {
      std::vector<int> idx(beam);
      for (int b = 0; b < beam; ++b) idx[b] = b;
      std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return sums[a * beam + x] > sums[a * beam + y]; });
      for (int b : idx) { if ((int)pool.size() >= beam) break; pool.insert({seqs[a * beam + b], sums[a * beam + b]}); }
    }
    const std::vector<int>* best = nullptr; double best_v = -1e300, best_sum = 0;
    for (auto& kv : pool) {
      const double v = kv.second / std::max<size_t>(kv.first.size(), 1);
      if (!best || v > best_v) { best = &kv.first; best_v = v; best_sum = kv.second; }
    }
    int n = 0;
    for (int t : *best) if (t != o->eot && n < max_new) out_tokens[(size_t)a * max_new + n++] = t;
    out_len[a] = n;
    if (out_lp) out_lp[a] = (float)best_sum;
    if (out_ns) out_ns[a] = ns_final[a];
  }
  return TTASR_OK;
}

}  // namespace ttasr_detail
