// libttasr: the typed launch schedules - encoder, cross-KV, decode step, prompt prefill - and the captured decode-step graphs
// (one of the engine translation units, see engine_ctx.hpp).
#include "engine_ctx.hpp"

namespace ttasr_detail {

// ---- typed schedules ------------------------------------------------------------------------------
// 256 x 256 tiles of a 16-bit GEMM (all its batches and groups)
static int64_t gemm_tiles_256(const GemmArgs& g) {
  return ((int64_t)(g.M + 255) / 256) * (g.N / 256) * std::max(1, g.batch) * std::max(1, g.groups);
}
// THE place that decides whether a 16-bit GEMM runs as persistent 256 x 256 workgroups (round 4): it pays once a workgroup has
// several tiles to walk (>= 2 per CU).  Shared by gemm() and the grouped cross-KV launch of run_cross_kv, which used to carry
// its own copy of this predicate.
static bool gemm_takes_persistent(const ttasr_ctx* c, const GemmArgs& g) {
  if (c->force_basic || g.M < 256) return false;
  const int v = c->gemm_force;  // option enc_gemm: force 1 = 128x128 two-stage, 2 = 256x128 three-stage, 3 = 256x256 four-stage, 4 = 3 as persistent workgroups
  return (v ? v == 4 : (c->gemm_persistent && gemm_tiles_256(g) >= 512)) && gemm_bf16_v4_ok(g);
}

template <typename T>
void gemm(ttasr_ctx* c, const GemmArgs& g) {
  if constexpr (sizeof(T) == 2) {
    if (!c->force_basic && g.M >= 256) {
      const int v = c->gemm_force;
      // 256x256 tiles need >= ~half the CUs' worth of tiles to pay; below that (one or two clips, short audio windows,
      // prefill) the 256x128 kernel's twice-as-many workgroups win (B = 1 encoder: 9.45 -> 6.6 ms)
      const int64_t tiles_v3 = gemm_tiles_256(g);
      if (gemm_takes_persistent(c, g)) {
        // round 5: the same persistent kernel with the last partial round of workgroups re-tiled into shorter tiles where that pays
        if (c->gemm_tail && v != 4 && launch_gemm_bf16_v5<T>(g, c->cur)) return;
        launch_gemm_bf16_v4<T>(g, c->cur);
        return;
      }
      if ((v ? v == 3 || v == 4 : tiles_v3 >= 128) && gemm_bf16_v3_ok(g)) { launch_gemm_bf16_v3<T>(g, c->cur); return; }
      if (v != 1 && gemm_bf16_v2_ok(g)) { launch_gemm_bf16_v2<T>(g, c->cur); return; }
      if (gemm_bf16_fast_ok(g)) { launch_gemm_bf16_fast<T>(g, c->cur); return; }
    }
  }
  launch_gemm_basic<T>(g, c->cur);
}

// decode-step GEMM: B rows against a streamed weight; bf16 uses the fragment-packed skinny kernel
template <typename T>
void dec_gemm(ttasr_ctx* c, const GemmArgs& g, const void* Wsh) {
  if (c->skip_mask & 2) return;
  if constexpr (sizeof(T) == 2) {
    if (!c->force_basic && Wsh) {
      // the vocabulary projection (f32 logits, nothing else in the epilogue): persistent workgroups, activation rows in registers
      const GemmEpi& e = g.epi;
      if (c->vocab_persistent && e.out_f32 && !e.out_t && !e.bias && !e.residual && e.act == 0 &&
          launch_gemm_vocab<T>((const T*)Wsh, (const T*)g.A, g.M, g.N, g.K, e.out_f32, e.ldc, c->cur, c->device)) return;
      if (launch_gemm_skinny<T>((const T*)Wsh, (const T*)g.A, g.M, g.N, g.K, g.epi, c->cur)) return;
    }
  }
  launch_gemm_basic<T>(g, c->cur);
}




void enc_mark(ttasr_ctx* c, int cls) {   // cls < 0: the start mark
  if (!c->enc_timing) return;
  const size_t i = c->enc_ev_class.size();
  if (i >= c->enc_ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->enc_ev.push_back(e); }
  hipEventRecord(c->enc_ev[i], c->cur);
  c->enc_ev_class.push_back(cls);
}

template <typename T>
int run_cross_kv(ttasr_ctx* c, int B) {
  const int d = c->d, T_ = c->T, L = c->cfg.dec_layers;
  auto layer_args = [&](int l) {
    GemmArgs g = lin_args<T>(c->enc_out, c->dec[l].wkvx, B * T_, 2 * d, d);
    g.epi.bias = c->dec[l].bkvx;
    g.epi.out_t = (char*)c->xkv + (size_t)l * c->xkv_layer_elems * c->esz;
    g.epi.headsplit = 1; g.epi.hs_T = T_; g.epi.hs_H = c->H; g.epi.hs_d = d; g.epi.hs_which = c->xkv_which_elems;
    return g;
  };
  // Round 5: the L projections share A (the encoder output) and their weights / biases / outputs are a fixed stride apart
  // (build_weights), so the persistent 256 x 256 kernel walks all L x tiles as ONE launch: at large-v3, B = 32 that is 60 160
  // tiles = 235.0 rounds of 256 workgroups instead of 32 launches x 8 rounds (7.34 rounded up).  Same tiles, same arithmetic
  // per tile: bit-identical to the per-layer launches (option xkv_grouped = 0).
  bool grouped = false;
  if constexpr (sizeof(T) == 2) {
    GemmArgs g = layer_args(0);
    g.groups = L; g.group_stride_w = (int64_t)2 * d * d; g.group_stride_out = c->xkv_layer_elems;
    if (c->xkv_grouped && L > 1 && gemm_takes_persistent(c, g)) {
      launch_gemm_bf16_v4<T>(g, c->cur);
      grouped = true;
      enc_mark(c, EC_XKV);
    }
  }
  for (int l = 0; l < L; ++l) {
    if (!grouped) gemm<T>(c, layer_args(l));
    if constexpr (sizeof(T) == 2) {
      if (c->xkv_fp8 && c->xkv8) {   // quantise this layer's K and V blocks of the B clips (one workgroup per (clip, head) block)
        for (int which = 0; which < 2; ++which) {
          const int64_t off = (int64_t)l * c->xkv_layer_elems + which * c->xkv_which_elems;
          launch_xkv_quant<T>((const T*)c->xkv + off, c->xkv8 + off, c->xkv8_scale + ((size_t)l * 2 + which) * c->maxB * c->H,
                              (int64_t)B * c->H, T_, c->cur);
        }
      }
    }
    if (!grouped || (c->xkv_fp8 && c->xkv8)) enc_mark(c, EC_XKV);
  }
  c->xkv8_valid = c->xkv_fp8 && c->xkv8 != nullptr && sizeof(T) == 2;
  return 0;
}

template <typename T>
int run_encoder(ttasr_ctx* c, int B) {
  const int d = c->d, T_ = c->T, F = c->F, M = c->M, ffn = c->ffn;
  hipStream_t s = c->cur;
  hipEventRecord(c->ev[2], s);
  c->enc_ev_class.clear();
  enc_mark(c, -1);
  {  // conv1 as GEMM over the zero-padded time-major mel image: row t of A = rows t..t+2 of the image
    GemmArgs g; g.A = c->mel_t; g.W = c->conv1_w; g.M = F; g.N = d; g.K = 3 * M; g.lda = M; g.ldw = 3 * M;
    g.batch = B; g.batch_stride_a = (int64_t)(F + 2) * M;
    g.epi.bias = c->conv1_b; g.epi.act = 1; g.epi.out_t = (char*)c->c1 + (size_t)d * c->esz; g.epi.ldc = d;
    g.epi.batch_stride_c = (int64_t)(F + 2) * d;
    gemm<T>(c, g);
  }
  {  // conv2 (stride 2): row t of A starts at padded row 2t; epilogue adds the sinusoid positions
    GemmArgs g; g.A = c->c1; g.W = c->conv2_w; g.M = T_; g.N = d; g.K = 3 * d; g.lda = 2 * d; g.ldw = 3 * d;
    g.batch = B; g.batch_stride_a = (int64_t)(F + 2) * d;
    g.epi.bias = c->conv2_b; g.epi.act = 1; g.epi.rowtab = c->epos; g.epi.rowmod = T_; g.epi.out_f32 = c->x;
    g.epi.ldc = d; g.epi.batch_stride_c = (int64_t)T_ * d;
    gemm<T>(c, g);
    enc_mark(c, EC_CONV);
  }
  const int R = B * T_;
  // bf16 mode: the out-proj / fc2 GEMMs write their result (bias added) as a T "delta" into the h buffer (dead at that
  // point: its consumer GEMM has run) with the plain wide-store epilogue, and the LayerNorm that follows adds it to the
  // f32 residual stream while normalising (kernels_misc.hip layernorm_kernel ADD).  The f32 residual read-modify-write in
  // those GEMMs' epilogues - exposed at one workgroup per CU - was what held out-proj at 0.66 PF/s.
  // f32 parity mode keeps the residual epilogue.
  const bool delta = sizeof(T) == 2 && !c->force_basic && !c->enc_res_epilogue;
  // Round 5 (option enc_ln_defer): the out-projection's delta goes to the (dead) qkv buffer and is NOT folded into x by the
  // LayerNorm that follows it (which only "peeks": normalises x + delta); the next LayerNorm - after fc2, whose delta sits in h -
  // folds both in, x = (x + delta_attn) + delta_ffn: one f32 read-modify-write of the residual stream per layer instead of two
  // (LayerNorm traffic per layer and element 22 B instead of 24), bit-identical.
  const bool defer = delta && c->enc_ln_defer;
  bool pending = false;       // a delta sits in h and has not been added to x yet
  bool pending_attn = false;  // defer: the out-projection's delta sits in the qkv buffer and has not been added to x yet
  auto ln = [&](const float* g_, const float* b_, void* out, bool after_attn = false) {
    if (pending_attn && after_attn) launch_layernorm_peek<T>(c->x, (const T*)c->qkv, g_, b_, (T*)out, R, d, s);
    else if (pending_attn && pending) { launch_layernorm_add2<T>(c->x, (const T*)c->qkv, (const T*)c->h, g_, b_, (T*)out, R, d, s); pending_attn = false; }
    else if (pending) launch_layernorm_add<T>(c->x, (const T*)c->h, g_, b_, (T*)out, R, d, s);
    else launch_layernorm<T>(c->x, g_, b_, (T*)out, R, d, s);
    if (!after_attn || !pending_attn) pending = false;
    enc_mark(c, EC_LN);
  };
  auto residual_gemm = [&](const void* A, const void* W, const float* bias, int K, int cls) {
    GemmArgs g = lin_args<T>(A, W, R, d, K); g.epi.bias = bias;
    if (delta && defer && cls == EC_OUT) { g.epi.out_t = c->qkv; pending_attn = true; }   // q / k / v are dead once attention has run
    else if (delta) { g.epi.out_t = c->h; pending = true; }
    else { g.epi.residual = c->x; g.epi.out_f32 = c->x; }
    gemm<T>(c, g);
    enc_mark(c, cls);
  };
  for (int l = 0; l < c->cfg.enc_layers; ++l) {
    const EncLayerW& L = c->enc[l];
    ln(L.ln1g, L.ln1b, c->h);
    { GemmArgs g = lin_args<T>(c->h, L.wqkv, R, 3 * d, d); g.epi.bias = L.bqkv; g.epi.out_t = c->qkv; gemm<T>(c, g); }
    enc_mark(c, EC_QKV);
    bool flash = false;
    if constexpr (sizeof(T) == 2) {
      if (!c->force_basic && !c->no_flash) { launch_enc_attn_flash_bf16<T>((const T*)c->qkv, (T*)c->att, B, T_, c->H, s); flash = true; }
    }
    if (!flash) launch_enc_attn_simple<T>((const T*)c->qkv, (T*)c->att, B, T_, c->H, s);
    enc_mark(c, EC_ATTN);
    residual_gemm(c->att, L.wo, L.bo, d, EC_OUT);
    ln(L.ln2g, L.ln2b, c->h, true);
    { GemmArgs g = lin_args<T>(c->h, L.w1, R, ffn, d); g.epi.bias = L.b1; g.epi.act = 1; g.epi.out_t = c->mid; gemm<T>(c, g); }
    enc_mark(c, EC_FC1);
    residual_gemm(c->mid, L.w2, L.b2, ffn, EC_FC2);
  }
  ln(c->elnf_g, c->elnf_b, c->enc_out);
  hipEventRecord(c->ev[3], s);
  run_cross_kv<T>(c, B);
  hipEventRecord(c->ev[4], s);
  return 0;
}

// One decoder step for rows [row0, row0 + n) at position *st.step, enqueued on c->cur.
// mode 0: through logits + select; 1: logits only (test API / beam search); 2: no logits (all rows forced by
// the prompt), select just advances the forced token.  `total_rows` = rows of the whole step (both half-batch chains):
// the select launch that finishes last advances the position counter.
//
// bf16 launch plan per layer (the measured mode; no float atomics anywhere, every launch bit-reproducible):
//   LN1 -> qkv GEMM (K-split, f32 slabs) -> self-attention (sums the q, k, v slabs) -> out-proj (K-split slabs) ->
//   LN2 (x += bias + slabs, then normalise) -> q GEMM (K-split slabs) -> cross-attention (sums the q slabs) ->
//   out-proj (slabs) -> LN3 (sums) -> fc1 + GELU (unsplit: the activation needs the full sum) -> fc2 (slabs) -> next LN1 (sums)
// Splitting K spreads every weight matrix over >= 160 workgroups in pieces of <= 20-40 KB (a CU takes in only ~25 GB/s
// of HBM-cold bytes).  LN1 of layer 0 creates the row from the token + position embedding itself.
// The f32 parity mode runs the generic kernels: LayerNorm, gemm_basic with the residual epilogue in place.
template <typename T>
void run_decode_rows(ttasr_ctx* c, int row0, int n, int mode, int total_rows) {
  const int d = c->d, ffn = c->ffn;
  hipStream_t s = c->cur;
  const size_t e = c->esz;
  auto tp = [&](void* base, int64_t width) { return (void*)((char*)base + (size_t)row0 * width * e); };  // T rows
  float* dx = c->dx + (size_t)row0 * d;
  void *dh = tp(c->dh, d), *dqkv = tp(c->dqkv, 3 * d), *dq = tp(c->dq, d), *datt = tp(c->datt, d), *dmid = tp(c->dmid, ffn);
  float* logits = c->logits + (size_t)row0 * c->ldv;
  const bool skinny = sizeof(T) == 2 && !c->force_basic;
  // round 6: rows whose search has FINISHED (st.done, set by select_kernel / uploaded by the beam search) leave the attention
  // kernels of the step - the per-row cross-KV and self-KV streams are the bytes of a decode step that scale with the rows
  const int32_t* done = c->ragged_exit ? c->st.done + row0 : nullptr;
  float* slab_base = c->slab;
  // K slices per GEMM kind (0 out-proj, 1 q, 2 qkv, 3 fc2); attention consumers sum at most 4 slabs
  auto slices = [&](int kind, int N, int K) {
    if (!skinny) return 1;
    int want = c->ks_want[kind];
    // Round 6: with the activation tile staged through LDS (kernels_skinny.hip) the K-split optimum of two GEMMs moved - re-swept
    // with `tools/decode_variants.py` (large-v3, 32 rows, decode of 128 tokens, two interleaved rounds): qkv UNSPLIT (120 n-blocks x
    // 8 waves, the self-attention reads q, k, v directly) 354.7-355.5 ms against 356.9-357.7 with the 2 slices that rounds 3-5
    // used; out-proj at the automatic 4 slices 355.4-356.1 against the 5 of round 5; both 352.7-352.9.  Wider batches (beam
    // search, streaming: 33-128 rows) keep the automatic choice, whose k-steps per wave fit the straight-line form.
    // (A different K split is a different summation order: the 16-bit token CRC of the benchmark was re-recorded with this change.)
    if (kind == 2 && want == 0 && n <= 32 && (N + 31) / 32 >= 96) want = 1;
    int ks = gemm_skinny_ksplit(n, N, K, want);
    if ((kind == 1 || kind == 2) && ks > 4) ks = gemm_skinny_ksplit(n, N, K, 4);
    return ks;
  };
  // what the next LayerNorm still has to add to the residual rows (K-split residual GEMM) or to create (embedding)
  struct { const float* bias = nullptr; int n_slab = 0; bool embed = true; } pend;

  // K-split GEMM into slabs [ks][maxB rows][N]; returns the slab descriptor for the consumer (n == 0: not split, `g` ran whole)
  auto split_gemm = [&](const GemmArgs& g, const void* Wsh, const float* bias, int ks) -> SlabIn {
    SlabIn si;
    if constexpr (sizeof(T) == 2) {
      if (ks > 1 && Wsh) {
        GemmEpi ep; ep.ldc = g.N;
        float* slab = slab_base;  // rows are local to this chain's region: [ks][maxB][N]
        const int64_t stride = (int64_t)c->maxB * g.N;
        if (launch_gemm_skinny<T>((const T*)Wsh, (const T*)g.A, n, g.N, g.K, ep, s, ks, slab, stride)) {
          si.slab = slab; si.bias = bias; si.n = ks; si.stride = stride; si.ld = g.N;
        }
      }
    }
    return si;
  };
  // x += W a + b
  auto residual_gemm = [&](const void* A, const void* W, const void* Wsh, const float* bias, int K, int kind) {
    if (c->skip_mask & 2) return;
    GemmArgs g = lin_args<T>(A, W, n, d, K);
    const SlabIn si = split_gemm(g, Wsh, bias, slices(kind, d, K));
    if (si.n) { pend.bias = bias; pend.n_slab = si.n; return; }
    g.epi.bias = bias; g.epi.residual = dx; g.epi.out_f32 = dx;
    dec_gemm<T>(c, g, Wsh);
  };
  auto ln = [&](const float* g_, const float* b_) {
    if (c->skip_mask & 1) return;
    LnPre pre;
    pre.x_out = dx;
    if (pend.embed) { pre.tok = c->st.cur_tok + row0; pre.step = c->st.step; pre.emb = c->emb; pre.pos = c->dpos; }
    else if (pend.n_slab) { pre.bias = pend.bias; pre.slab = slab_base; pre.n_slab = pend.n_slab; pre.slab_stride = (int64_t)c->maxB * d; }
    launch_layernorm_rows<T>(dx, g_, b_, (T*)dh, n, d, pre, s);
    pend.bias = nullptr; pend.n_slab = 0; pend.embed = false;
  };
  for (int l = 0; l < c->cfg.dec_layers; ++l) {
    const DecLayerW& L = c->dec[l];
    ln(L.ln1g, L.ln1b);
    SlabIn sqkv;
    { GemmArgs g = lin_args<T>(dh, L.wqkv, n, 3 * d, d);
      sqkv = split_gemm(g, L.wqkv_sh, L.bqkv, slices(2, 3 * d, d));
      if (!sqkv.n) { g.epi.bias = L.bqkv; g.epi.out_t = dqkv; dec_gemm<T>(c, g, L.wqkv_sh); } }
    if (!(c->skip_mask & 4))
      launch_self_attn_decode<T>((const T*)dqkv, (T*)c->pool, c->page_table, c->pages_per_seq, (int64_t)l * c->pool_layer_elems,
                                 c->identity_pages, row0, c->st.step, (T*)datt, n, c->H, s, sqkv, done);
    residual_gemm(datt, L.wo, L.wo_sh, L.bo, d, 0);
    ln(L.ln2g, L.ln2b);
    SlabIn sq;
    { GemmArgs g = lin_args<T>(dh, L.wqx, n, d, d);
      sq = split_gemm(g, L.wqx_sh, L.bqx, slices(1, d, d));
      if (!sq.n) { g.epi.bias = L.bqx; g.epi.out_t = dq; dec_gemm<T>(c, g, L.wqx_sh); } }
    // cross-KV of clip (row / kv_div); a half-batch offset is only used with kv_div == 1
    const T* Kx = (const T*)c->xkv + (int64_t)l * c->xkv_layer_elems + (int64_t)(row0 / c->kv_div) * c->H * c->T * 64;
    bool fp8_done = false;
    if constexpr (sizeof(T) == 2) {   // opt-in: the e4m3 copy of the cache, unshared rows that fill the chip (the single-pass kernel's case)
      if (c->xkv_fp8 && c->xkv8_valid && c->kv_div == 1 && n * c->H >= 256 && skinny && !(c->skip_mask & 8)) {
        const int64_t off = (int64_t)l * c->xkv_layer_elems + (int64_t)row0 * c->H * c->T * 64;
        const float* ksc = c->xkv8_scale + ((size_t)l * 2) * c->maxB * c->H + (size_t)row0 * c->H;
        fp8_done = launch_cross_attn_fp8<T>((const T*)dq, c->xkv8 + off, c->xkv8 + off + c->xkv_which_elems, ksc, ksc + (size_t)c->maxB * c->H,
                                            (T*)datt, n, c->H, c->T, s, sq, done);
      }
    }
    if (!fp8_done && !(c->skip_mask & 8))
      launch_cross_attn_decode<T>((const T*)dq, Kx, Kx + c->xkv_which_elems, (T*)datt, n, c->H, c->T, c->kv_div, s,
                                  c->no_xsplit ? nullptr : c->xsplit_ws + (size_t)row0 * c->H * 8 * 66, sq, c->maxB - row0, QProj{}, done);
    residual_gemm(datt, L.wox, L.wox_sh, L.box, d, 0);
    ln(L.ln3g, L.ln3b);
    { GemmArgs g = lin_args<T>(dh, L.w1, n, ffn, d); g.epi.bias = L.b1; g.epi.act = 1; g.epi.out_t = dmid; dec_gemm<T>(c, g, L.w1_sh); }
    residual_gemm(dmid, L.w2, L.w2_sh, L.b2, ffn, 3);
  }
  if (mode != 2) {
    ln(c->dlnf_g, c->dlnf_b);
    GemmArgs g = lin_args<T>(dh, c->emb, n, c->V, d);  // proj_out tied to embed_tokens (modeling_whisper.py:965)
    g.epi.out_f32 = logits; g.epi.ldc = c->ldv;
    dec_gemm<T>(c, g, c->emb_sh);
  }
  if (mode != 1 && !(c->skip_mask & 16)) {
    DecState st = c->st;  // row-offset view of the search state
    st.cur_tok += row0; st.n_sampled += row0; st.last_tok += row0; st.pen_tok += row0; st.last_ts += row0; st.done += row0;
    st.sum_logprob += row0; st.no_speech += row0; st.out_tokens += (size_t)row0 * c->rp.max_new; st.row_cap += row0;
    if (st.prompt) { st.prompt += (size_t)row0 * c->rp.max_prompt; st.prompt_len += row0; }
    launch_select(logits, st, c->rp, n, nullptr, s, c->st.step + 1, total_rows);
  }
}


template <typename T>
void run_prefill(ttasr_ctx* c, int n_seq, int npos, int seq_per_clip, int max_prompt, const AlignOut* al) {
  const int d = c->d, ffn = c->ffn, n = n_seq * npos;
  hipStream_t s = c->cur = c->stream;
  float* x = c->x;
  void *h = c->h, *qkv = c->qkv, *att = c->att, *mid = c->mid;
  launch_embed_prefill<T>(c->prompt_dev, max_prompt, 1, n_seq, npos, (const T*)c->emb, (const T*)c->dpos, x, d, s);
  // up to 128 rows (short prompts: a handful of positions x the clips of a pass) the fragment-packed decode GEMM streams each
  // weight once for all rows; beyond that the rows are a real M dimension for the tiled encoder GEMMs
  const bool small = n <= 128 && !c->force_basic && !c->prefill_tiled;
  // Round 3: the small pass runs the DECODE-STEP launch plan - every GEMM whose consumer can add partial results is cut into K
  // slices (160-320 workgroups instead of 40 of them pulling 164-656 KB each: a CU takes in ~25 GB/s of cold bytes), the
  // partial tiles go to the f32 slabs, and the per-row LayerNorm / the cross-attention kernel sum them in slab order (no
  // atomics: bit-reproducible).  A 3-position prompt of 32 clips then costs about 1.3 decode steps instead of 3.
  const bool slabbed = small && sizeof(T) == 2;
  const int64_t slab_cap = (int64_t)16 * c->maxB * 3 * d;   // floats in c->slab
  struct { const float* bias = nullptr; int n_slab = 0; int64_t stride = 0; } pend;
  auto pgemm = [&](const GemmArgs& g, const void* Wsh) {
    if constexpr (sizeof(T) == 2) {
      if (small && Wsh && launch_gemm_skinny<T>((const T*)Wsh, (const T*)g.A, g.M, g.N, g.K, g.epi, s)) return;
    }
    gemm<T>(c, g);
  };
  // K-split GEMM into slabs [ks][n][N]; returns the slab descriptor (n == 0: not split, the caller runs the GEMM whole)
  auto split_gemm = [&](const GemmArgs& g, const void* Wsh, const float* bias, int max_ks) -> SlabIn {
    SlabIn si;
    if constexpr (sizeof(T) == 2) {
      if (slabbed && Wsh) {
        int ks = gemm_skinny_ksplit(n, g.N, g.K, 0);
        if (ks > max_ks) ks = gemm_skinny_ksplit(n, g.N, g.K, max_ks);
        const int64_t stride = (int64_t)n * g.N;
        GemmEpi ep; ep.ldc = g.N;
        if (ks > 1 && ks * stride <= slab_cap &&
            launch_gemm_skinny<T>((const T*)Wsh, (const T*)g.A, n, g.N, g.K, ep, s, ks, c->slab, stride)) {
          si.slab = c->slab; si.bias = bias; si.n = ks; si.stride = stride; si.ld = g.N;
        }
      }
    }
    return si;
  };
  auto ln = [&](const float* g_, const float* b_) {
    if (slabbed) {
      LnPre pre; pre.x_out = x;
      if (pend.n_slab) { pre.bias = pend.bias; pre.slab = c->slab; pre.n_slab = pend.n_slab; pre.slab_stride = pend.stride; }
      launch_layernorm_rows<T>(x, g_, b_, (T*)h, n, d, pre, s);
      pend.bias = nullptr; pend.n_slab = 0;
    } else {
      launch_layernorm<T>(x, g_, b_, (T*)h, n, d, s);
    }
  };
  auto residual_gemm = [&](const void* A, const void* W, const void* Wsh, const float* bias, int K) {   // x += W a + b
    GemmArgs g = lin_args<T>(A, W, n, d, K);
    const SlabIn si = split_gemm(g, Wsh, bias, 16);
    if (si.n) { pend.bias = bias; pend.n_slab = si.n; pend.stride = si.stride; return; }
    g.epi.bias = bias; g.epi.residual = x; g.epi.out_f32 = x;
    pgemm(g, Wsh);
  };
  for (int l = 0; l < c->cfg.dec_layers; ++l) {
    const DecLayerW& L = c->dec[l];
    ln(L.ln1g, L.ln1b);
    { GemmArgs g = lin_args<T>(h, L.wqkv, n, 3 * d, d); g.epi.bias = L.bqkv; g.epi.out_t = qkv; pgemm(g, L.wqkv_sh); }
    launch_self_attn_prefill<T>((const T*)qkv, (T*)c->pool, c->page_table, c->pages_per_seq, (int64_t)l * c->pool_layer_elems,
                                c->identity_pages, (T*)att, n_seq, npos, c->H, s);
    residual_gemm(att, L.wo, L.wo_sh, L.bo, d);
    ln(L.ln2g, L.ln2b);
    SlabIn sq;   // the query of the cross-attention: K-split too when its consumer can sum slabs (not the alignment pass)
    { GemmArgs g = lin_args<T>(h, L.wqx, n, d, d);
      if (!al && npos * seq_per_clip < 32) sq = split_gemm(g, L.wqx_sh, L.bqx, 4);   // >= 32 rows per clip: the MFMA flash pass reads T rows
      if (!sq.n) { g.epi.bias = L.bqx; g.epi.out_t = qkv; pgemm(g, L.wqx_sh); } }  // q reuses the qkv buffer
    const T* Kx = (const T*)c->xkv + (int64_t)l * c->xkv_layer_elems;
    if (al) {
      const T* Kc = Kx + (int64_t)al->clip * c->H * c->T * 64;
      launch_cross_attn_probs<T>((const T*)qkv, Kc, Kc + c->xkv_which_elems, (T*)att, n, c->H, c->T, al->sel + (size_t)l * c->H,
                                 al->probs, s);
    } else {
      launch_cross_attn_decode<T>((const T*)qkv, Kx, Kx + c->xkv_which_elems, (T*)att, n, c->H, c->T, npos * seq_per_clip, s,
                                  c->no_xsplit ? nullptr : c->xsplit_ws, sq, c->maxB);
    }
    residual_gemm(att, L.wox, L.wox_sh, L.box, d);
    ln(L.ln3g, L.ln3b);
    { GemmArgs g = lin_args<T>(h, L.w1, n, ffn, d); g.epi.bias = L.b1; g.epi.act = 1; g.epi.out_t = mid; pgemm(g, L.w1_sh); }
    residual_gemm(mid, L.w2, L.w2_sh, L.b2, ffn);
  }
  // callers read the finished residual rows from c->x (no-speech probability, token log-probs of the alignment pass): fold the
  // last fc2's partial tiles in (the final decoder LayerNorm does it; its normalised output lands in h and is not used here)
  if (pend.n_slab) ln(c->dlnf_g, c->dlnf_b);
}

// How many leading prompt positions can be prefilled: every row must still have a forced token after them and the rows
// must fit the borrowed encoder workspaces.  Below 2 positions the pass does not pay.  `ns_from_prefill` = the caller can
// take the no-speech probability from the prefilled <|startoftranscript|> position (prefill_no_speech); otherwise that
// position needs a real decode step and bounds the prefill.
int prefill_positions(const ttasr_ctx* c, int min_plen, const ttasr_gen_opts* o, bool ns_from_prefill) {
  if (c->no_prefill) return 0;
  int p = min_plen - 1;
  if (o->no_speech >= 0 && !ns_from_prefill) p = std::min(p, o->sot_index);
  p = std::min(p, c->cfg.n_audio_ctx);
  return p >= 2 ? p : 0;
}

// No-speech probability from a prefill pass: the residual rows of position `sot` of every sequence (left in c->x by
// run_prefill, rows [sequence][position]) -> final LayerNorm -> vocabulary projection -> softmax(raw logits)[no_speech].
template <typename T>
int prefill_no_speech(ttasr_ctx* c, int n_seq, int npos, int sot, int no_speech_tok) {
  hipStream_t s = c->stream;
  const int d = c->d;
  c->cur = s;
  HIPCHK(c, hipMemcpy2DAsync(c->dx, (size_t)d * 4, c->x + (size_t)sot * d, (size_t)npos * d * 4, (size_t)d * 4, n_seq,
                             hipMemcpyDeviceToDevice, s));
  launch_layernorm_rows<T>(c->dx, c->dlnf_g, c->dlnf_b, (T*)c->dh, n_seq, d, LnPre{}, s);
  GemmArgs g = lin_args<T>(c->dh, c->emb, n_seq, c->V, d);
  g.epi.out_f32 = c->logits; g.epi.ldc = c->ldv;
  dec_gemm<T>(c, g, c->emb_sh);
  launch_token_prob(c->logits, c->ldv, c->V, no_speech_tok, c->st.no_speech, n_seq, s);
  return 0;
}

// One decode step = one dependent chain of ~355 launches on the context's stream, captured as a hipGraph.  Splitting the
// batch into two half-batch chains on two streams inside the graph (round 1's dual-chain experiment: +4 % then) doubles the
// launch count and, with the round-2 kernels, measures 3 % SLOWER (3.15 vs 3.05 ms per step): removed.
template <typename T>
void run_decode_step(ttasr_ctx* c, int B, int mode) {
  c->cur = c->stream;
  run_decode_rows<T>(c, 0, B, mode, B);
  // modes 0 and 2 end with select_kernel, whose last workgroup advances the position; mode 1 has no select
  if (mode == 1 || (c->skip_mask & 16)) launch_advance(c->st.step, c->stream);
}

// `nsteps` consecutive steps of the same mode as ONE graph (round 4): the search state is device-resident, so a run of greedy
// steps between two host polls needs no host involvement at all; one replay instead of nsteps saves the graph-launch gap
// (~8 us on the device, 10-16 us of host time per replay) per step.
int step_graph(ttasr_ctx* c, int B, int mode, int nsteps) {
  if (!c->use_graph) {
    for (int i = 0; i < nsteps; ++i) TT_DISPATCH(c, run_decode_step<T>(c, B, mode));
    return 0;
  }
  // everything that decides WHICH kernels a captured step holds is part of the key: whether the e4m3 cross-KV copy is live
  // changes with encode (run_cross_kv builds it), not only with set_option (a graph captured between
  // set_option(xkv_fp8) and the next encode held the 16-bit kernel and kept replaying after the copy existed)
  const int fp8_live = (c->xkv_fp8 && c->xkv8_valid) ? 1 : 0;
  const int variant = ((c->kv_div * 2 + c->identity_pages) * 2 + fp8_live) * 64 + nsteps;
  for (size_t i = 0; i < c->graphs.size(); ++i) {
    if (c->graphs[i].B == B && c->graphs[i].mode == mode && c->graphs[i].variant == variant) {
      // most recently used at the back: the cache is bounded (the streaming micro-batcher varies B from 1 to max_batch rows)
      if (i + 1 != c->graphs.size()) std::rotate(c->graphs.begin() + i, c->graphs.begin() + i + 1, c->graphs.end());
      HIPCHK(c, hipGraphLaunch(c->graphs.back().exec, c->stream));
      return 0;
    }
  }
  hipGraph_t graph = nullptr;
  HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nsteps; ++i) TT_DISPATCH(c, run_decode_step<T>(c, B, mode));
  HIPCHK(c, hipStreamEndCapture(c->stream, &graph));
  hipGraphExec_t exec = nullptr;
  const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);   // on the failure path too
  if (ie != hipSuccess) return fail(c, TTASR_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
  if (c->graphs.size() >= ttasr_ctx::kMaxGraphs) {   // evict the least recently used executable
    hipGraphExecDestroy(c->graphs.front().exec);
    c->graphs.erase(c->graphs.begin());
  }
  c->graphs.push_back({B, mode, variant, exec});
  HIPCHK(c, hipGraphLaunch(exec, c->stream));
  return 0;
}

void drop_graphs(ttasr_ctx* c) {
  for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
  c->graphs.clear();
}

// The rule scalars (RuleParams) are baked into the select launch of the mode 0 / 2 graphs only; the logits-only graphs
// (mode 1: step API, beam search) never launch select_kernel and survive a change of rules - with
// condition_on_previous_text the prompt geometry changes on nearly every window of a file.
void drop_rule_graphs(ttasr_ctx* c) {
  size_t k = 0;
  for (auto& g : c->graphs) {
    if (g.mode == 1) c->graphs[k++] = g; else hipGraphExecDestroy(g.exec);
  }
  c->graphs.resize(k);
}

// ---- entry points for the other translation units: T is dispatched here ----
void sched_encoder(ttasr_ctx* c, int B) { TT_DISPATCH(c, run_encoder<T>(c, B)); }
void sched_cross_kv(ttasr_ctx* c, int B) { TT_DISPATCH(c, run_cross_kv<T>(c, B)); }
void sched_prefill(ttasr_ctx* c, int n_seq, int npos, int seq_per_clip, int max_prompt, const AlignOut* al) {
  TT_DISPATCH(c, run_prefill<T>(c, n_seq, npos, seq_per_clip, max_prompt, al));
}
int sched_prefill_no_speech(ttasr_ctx* c, int n_seq, int npos, int sot, int no_speech_tok) {
  TT_DISPATCH(c, return prefill_no_speech<T>(c, n_seq, npos, sot, no_speech_tok));
  return 0;
}
void sched_gemm(ttasr_ctx* c, const GemmArgs& g) { TT_DISPATCH(c, gemm<T>(c, g)); }
void sched_dec_gemm(ttasr_ctx* c, const GemmArgs& g, const void* Wsh) { TT_DISPATCH(c, dec_gemm<T>(c, g, Wsh)); }

}  // namespace ttasr_detail
