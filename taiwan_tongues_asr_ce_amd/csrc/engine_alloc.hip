// libttasr: device arenas, weight slots / workspaces, weight intake (one of the engine translation units, see engine_ctx.hpp).
#include "engine_ctx.hpp"

namespace ttasr_detail {

thread_local std::string g_create_error;

int fail(ttasr_ctx* c, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

// slaney mel filter bank, same construction as the oracle's mel_filter_bank (float64, cast to f32)
double hz2mel(double f) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel2hz(double m) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
std::vector<float> mel_filters(int n_mels) {
  const int nf = 201;
  std::vector<double> hz(n_mels + 2);
  const double m0 = hz2mel(0.0), m1 = hz2mel(8000.0);
  for (int i = 0; i < n_mels + 2; ++i) hz[i] = mel2hz(m0 + (m1 - m0) * i / (n_mels + 1));
  std::vector<float> fb((size_t)nf * n_mels);
  for (int k = 0; k < nf; ++k) {
    double f = 8000.0 * k / (nf - 1);
    for (int m = 0; m < n_mels; ++m) {
      double down = -(hz[m] - f) / (hz[m + 1] - hz[m]);
      double up = (hz[m + 2] - f) / (hz[m + 2] - hz[m + 1]);
      double w = std::max(0.0, std::min(down, up)) * 2.0 / (hz[m + 2] - hz[m]);
      fb[(size_t)k * n_mels + m] = (float)w;
    }
  }
  return fb;
}

void add_slot(ttasr_ctx* c, const std::string& name, void* dst, int64_t rows, int64_t cols, int kind, float scale = 1.f) {
  Slot s; s.dst = dst; s.rows = rows; s.cols = cols; s.kind = kind; s.scale = scale;
  c->slots[name] = s;
}

int alloc_mat(ttasr_ctx* c, void** p, int64_t elems) { return dalloc(c, (char**)p, (size_t)elems * c->esz); }
int alloc_vec(ttasr_ctx* c, float** p, int64_t elems) { return dalloc(c, p, (size_t)elems * 4); }

int build_weights(ttasr_ctx* c) {
  const int d = c->d, F = c->ffn, M = c->M, V = c->V;
  const size_t e = c->esz;
  auto off = [&](void* base, int64_t elems) { return (void*)((char*)base + (size_t)elems * e); };
  TRY(alloc_mat(c, &c->conv1_w, (int64_t)d * 3 * M)); TRY(alloc_vec(c, &c->conv1_b, d));
  TRY(alloc_mat(c, &c->conv2_w, (int64_t)d * 3 * d)); TRY(alloc_vec(c, &c->conv2_b, d));
  TRY(alloc_vec(c, &c->epos, (int64_t)c->T * d));
  add_slot(c, "model.encoder.conv1.weight", c->conv1_w, d, 3 * M, 2);
  add_slot(c, "model.encoder.conv1.bias", c->conv1_b, d, 1, 1);
  add_slot(c, "model.encoder.conv2.weight", c->conv2_w, d, 3 * d, 2);
  add_slot(c, "model.encoder.conv2.bias", c->conv2_b, d, 1, 1);
  add_slot(c, "model.encoder.embed_positions.weight", c->epos, c->T, d, 3);
  auto ln = [&](const std::string& p, float** g, float** b) -> int {
    TRY(alloc_vec(c, g, d)); TRY(alloc_vec(c, b, d));
    add_slot(c, p + ".weight", *g, d, 1, 1); add_slot(c, p + ".bias", *b, d, 1, 1);
    return 0;
  };
  // fused q|k|v: rows [0,d) = q (pre-scaled by head_dim^-0.5 = 1/8, exact in f32 and bf16;
  // HF modeling_whisper.py:309 scales the q_proj output), [d,2d) = k (no bias, :279), [2d,3d) = v
  auto attn_fused = [&](const std::string& p, void** wqkv, float** bqkv) -> int {
    TRY(alloc_mat(c, wqkv, (int64_t)3 * d * d)); TRY(alloc_vec(c, bqkv, 3 * d));
    add_slot(c, p + ".q_proj.weight", *wqkv, d, d, 0, 0.125f);
    add_slot(c, p + ".q_proj.bias", *bqkv, d, 1, 1, 0.125f);
    add_slot(c, p + ".k_proj.weight", off(*wqkv, (int64_t)d * d), d, d, 0);
    add_slot(c, p + ".v_proj.weight", off(*wqkv, (int64_t)2 * d * d), d, d, 0);
    add_slot(c, p + ".v_proj.bias", *bqkv + 2 * d, d, 1, 1);
    return 0;
  };
  auto lin = [&](const std::string& p, void** w, float** b, int64_t n_out, int64_t n_in) -> int {
    TRY(alloc_mat(c, w, n_out * n_in)); TRY(alloc_vec(c, b, n_out));
    add_slot(c, p + ".weight", *w, n_out, n_in, 0); add_slot(c, p + ".bias", *b, n_out, 1, 1);
    return 0;
  };
  c->enc.resize(c->cfg.enc_layers);
  for (int i = 0; i < c->cfg.enc_layers; ++i) {
    std::string p = "model.encoder.layers." + std::to_string(i);
    EncLayerW& L = c->enc[i];
    TRY(ln(p + ".self_attn_layer_norm", &L.ln1g, &L.ln1b));
    TRY(attn_fused(p + ".self_attn", &L.wqkv, &L.bqkv));
    TRY(lin(p + ".self_attn.out_proj", &L.wo, &L.bo, d, d));
    TRY(ln(p + ".final_layer_norm", &L.ln2g, &L.ln2b));
    TRY(lin(p + ".fc1", &L.w1, &L.b1, F, d));
    TRY(lin(p + ".fc2", &L.w2, &L.b2, d, F));
  }
  TRY(ln("model.encoder.layer_norm", &c->elnf_g, &c->elnf_b));
  TRY(alloc_mat(c, &c->emb, (int64_t)V * d));
  TRY(alloc_mat(c, &c->dpos, (int64_t)c->cfg.n_text_ctx * d));
  add_slot(c, "model.decoder.embed_tokens.weight", c->emb, V, d, 0);
  add_slot(c, "model.decoder.embed_positions.weight", c->dpos, c->cfg.n_text_ctx, d, 0);
  c->dec.resize(c->cfg.dec_layers);
  // the cross-attention K / V projections of ALL decoder layers as one [layers][2 d][d] block (+ biases [layers][2 d]): the
  // cross-KV build walks them as one grouped GEMM (run_cross_kv), whose groups are a fixed stride apart
  void* xkv_w_all = nullptr; float* xkv_b_all = nullptr;
  TRY(alloc_mat(c, &xkv_w_all, (int64_t)c->cfg.dec_layers * 2 * d * d));
  TRY(alloc_vec(c, &xkv_b_all, (int64_t)c->cfg.dec_layers * 2 * d));
  for (int i = 0; i < c->cfg.dec_layers; ++i) {
    std::string p = "model.decoder.layers." + std::to_string(i);
    DecLayerW& L = c->dec[i];
    TRY(ln(p + ".self_attn_layer_norm", &L.ln1g, &L.ln1b));
    TRY(attn_fused(p + ".self_attn", &L.wqkv, &L.bqkv));
    TRY(lin(p + ".self_attn.out_proj", &L.wo, &L.bo, d, d));
    TRY(ln(p + ".encoder_attn_layer_norm", &L.ln2g, &L.ln2b));
    TRY(alloc_mat(c, &L.wqx, (int64_t)d * d)); TRY(alloc_vec(c, &L.bqx, d));
    add_slot(c, p + ".encoder_attn.q_proj.weight", L.wqx, d, d, 0, 0.125f);
    add_slot(c, p + ".encoder_attn.q_proj.bias", L.bqx, d, 1, 1, 0.125f);
    L.wkvx = off(xkv_w_all, (int64_t)i * 2 * d * d); L.bkvx = xkv_b_all + (int64_t)i * 2 * d;
    add_slot(c, p + ".encoder_attn.k_proj.weight", L.wkvx, d, d, 0);
    add_slot(c, p + ".encoder_attn.v_proj.weight", off(L.wkvx, (int64_t)d * d), d, d, 0);
    add_slot(c, p + ".encoder_attn.v_proj.bias", L.bkvx + d, d, 1, 1);
    TRY(lin(p + ".encoder_attn.out_proj", &L.wox, &L.box, d, d));
    TRY(ln(p + ".final_layer_norm", &L.ln3g, &L.ln3b));
    TRY(lin(p + ".fc1", &L.w1, &L.b1, F, d));
    TRY(lin(p + ".fc2", &L.w2, &L.b2, d, F));
  }
  TRY(ln("model.decoder.layer_norm", &c->dlnf_g, &c->dlnf_b));
  if (c->lowp) {  // fragment-packed copies of every matrix the decode step streams
    auto packed = [&](const std::string& name, void** base, int64_t rows_total, int64_t K, int row_off) -> int {
      if (!*base) TRY(alloc_mat(c, base, (rows_total + 31) / 32 * 32 * K));
      Slot& s = c->slots[name];
      s.sh_base = *base; s.sh_row_off = row_off; s.sh_rows_total = (int)rows_total;
      return 0;
    };
    TRY(packed("model.decoder.embed_tokens.weight", &c->emb_sh, V, d, 0));
    for (int i = 0; i < c->cfg.dec_layers; ++i) {
      std::string p = "model.decoder.layers." + std::to_string(i);
      DecLayerW& L = c->dec[i];
      TRY(packed(p + ".self_attn.q_proj.weight", &L.wqkv_sh, 3 * d, d, 0));
      TRY(packed(p + ".self_attn.k_proj.weight", &L.wqkv_sh, 3 * d, d, d));
      TRY(packed(p + ".self_attn.v_proj.weight", &L.wqkv_sh, 3 * d, d, 2 * d));
      TRY(packed(p + ".self_attn.out_proj.weight", &L.wo_sh, d, d, 0));
      TRY(packed(p + ".encoder_attn.q_proj.weight", &L.wqx_sh, d, d, 0));
      TRY(packed(p + ".encoder_attn.out_proj.weight", &L.wox_sh, d, d, 0));
      TRY(packed(p + ".fc1.weight", &L.w1_sh, F, d, 0));
      TRY(packed(p + ".fc2.weight", &L.w2_sh, d, F, 0));
    }
  }
  c->stage_elems = 0;   // every host upload is staged in the source layout: the largest registered tensor decides
  for (auto& kv : c->slots) c->stage_elems = std::max<size_t>(c->stage_elems, (size_t)(kv.second.rows * kv.second.cols));
  TRY(dalloc(c, &c->stage_f32, c->stage_elems * 4, false));
  TRY(dalloc(c, &c->stage_raw, c->stage_elems * 4, false));
  return 0;
}

int build_workspaces(ttasr_ctx* c) {
  const int64_t B = c->maxB, T = c->T, F = c->F, d = c->d, M = c->M, H = c->H;
  TRY(dalloc(c, &c->pcm_dev, (size_t)B * (c->n_samples + 512) * 4));  // + context samples of file windows
  TRY(dalloc(c, &c->mel_geom, (size_t)B * 3 * 8));
  TRY(dalloc(c, &c->nsamp_dev, (size_t)B * 8));
  TRY(dalloc(c, &c->clip_max, (size_t)B * 4));
  TRY(dalloc(c, &c->mel, (size_t)B * M * F * 4));
  TRY(alloc_mat(c, &c->mel_t, B * (F + 2) * M));
  TRY(alloc_mat(c, &c->c1, B * (F + 2) * d));
  TRY(dalloc(c, &c->x, (size_t)B * T * d * 4));
  TRY(alloc_mat(c, &c->h, B * T * d));
  TRY(alloc_mat(c, &c->qkv, B * T * 3 * d));
  TRY(alloc_mat(c, &c->att, B * T * d));
  TRY(alloc_mat(c, &c->mid, B * T * c->ffn));
  TRY(alloc_mat(c, &c->enc_out, B * T * d));
  c->xkv_which_elems = B * H * T * 64;
  c->xkv_layer_elems = 2 * c->xkv_which_elems;
  TRY(alloc_mat(c, &c->xkv, c->xkv_layer_elems * c->cfg.dec_layers));
  TRY(dalloc(c, &c->xsplit_ws, (size_t)B * H * 8 * 66 * 4));
  c->pages_per_seq = (c->cfg.n_text_ctx + 15) / 16;
  const int64_t n_pages = B * c->pages_per_seq;
  c->pool_layer_elems = n_pages * 2 * H * 16 * 64;
  TRY(alloc_mat(c, &c->pool, c->pool_layer_elems * c->cfg.dec_layers));
  TRY(dalloc(c, &c->page_table, (size_t)n_pages * 4));
  std::vector<int32_t> pt(n_pages);
  for (int64_t i = 0; i < n_pages; ++i) pt[i] = (int32_t)i;  // identity: row b owns pages [b*pps, (b+1)*pps)
  HIPCHK(c, hipMemcpyAsync(c->page_table, pt.data(), n_pages * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  TRY(dalloc(c, &c->dx, (size_t)B * d * 4));
  TRY(alloc_mat(c, &c->dh, B * d));
  TRY(alloc_mat(c, &c->dqkv, B * 3 * d));
  TRY(alloc_mat(c, &c->dq, B * d));
  TRY(alloc_mat(c, &c->datt, B * d));
  TRY(alloc_mat(c, &c->dmid, B * c->ffn));
  TRY(dalloc(c, &c->logits, (size_t)B * c->ldv * 4));
  TRY(dalloc(c, &c->slab, (size_t)16 * B * 3 * d * 4));
  c->max_new_alloc = c->cfg.n_text_ctx;
  c->max_prompt_alloc = c->cfg.n_text_ctx;
  TRY(dalloc(c, &c->st.cur_tok, B * 4)); TRY(dalloc(c, &c->st.step, 16)); TRY(dalloc(c, &c->st.n_sampled, B * 4));
  TRY(dalloc(c, &c->st.last_tok, B * 4)); TRY(dalloc(c, &c->st.pen_tok, B * 4)); TRY(dalloc(c, &c->st.last_ts, B * 4));
  TRY(dalloc(c, &c->st.done, B * 4)); TRY(dalloc(c, &c->st.n_done, 16)); TRY(dalloc(c, &c->st.sum_logprob, B * 4));
  TRY(dalloc(c, &c->st.no_speech, B * 4)); TRY(dalloc(c, &c->st.out_tokens, (size_t)B * c->max_new_alloc * 4));
  TRY(dalloc(c, &c->prompt_dev, (size_t)B * c->max_prompt_alloc * 4)); TRY(dalloc(c, &c->plen_dev, B * 4));
  TRY(dalloc(c, &c->mask_dev, (size_t)c->V + 16));
  TRY(dalloc(c, &c->rule_dyn_dev, sizeof(RuleDyn)));
  c->st.dyn = c->rule_dyn_dev;
  TRY(dalloc(c, &c->row_cap_dev, B * 4, false));
  HIPCHK(c, hipMemsetAsync(c->row_cap_dev, 0x7f, B * 4, c->stream));
  c->st.row_cap = c->row_cap_dev;
  TRY(dalloc(c, &c->pairs_dev, (size_t)B * 2 * 4));
  TRY(dalloc(c, &c->topk_lp, (size_t)B * 8 * 4)); TRY(dalloc(c, &c->topk_id, (size_t)B * 8 * 4));
  TRY(dalloc(c, &c->row_state, (size_t)B * 4 * 4));
  c->st.mask = c->mask_dev;
  HIPCHK(c, hipHostMalloc((void**)&c->pinned_i32, 4096));
  // mel constants
  std::vector<float> fb = mel_filters(c->M), cs(400), sn(400), wn(400);
  for (int i = 0; i < 400; ++i) {
    cs[i] = (float)std::cos(2.0 * M_PI * i / 400.0);
    sn[i] = (float)std::sin(2.0 * M_PI * i / 400.0);
    wn[i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / 400.0));
  }
  TRY(dalloc(c, &c->filters, fb.size() * 4)); TRY(dalloc(c, &c->dcos, 1600)); TRY(dalloc(c, &c->dsin, 1600));
  TRY(dalloc(c, &c->window, 1600));
  HIPCHK(c, hipMemcpyAsync(c->filters, fb.data(), fb.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->dcos, cs.data(), 1600, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->dsin, sn.data(), 1600, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->window, wn.data(), 1600, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& e : c->ev) HIPCHK(c, hipEventCreate(&e));
  return 0;
}

// Shared by the host and device entry points: `src` is a DEVICE pointer to the tensor in its source layout (float32 or
// bf16 bits); everything from here on - conv tap re-ordering, q pre-scaling, bf16 cast, MFMA-fragment packing - runs on
// the device.
int ingest_tensor(ttasr_ctx* c, const char* name, const void* src, int src_type, const int64_t* dims, int32_t ndim) {
  if (std::string(name) == "proj_out.weight") return TTASR_OK;  // tied to embed_tokens
  auto it = c->slots.find(name);
  if (it == c->slots.end()) return fail(c, TTASR_E_WEIGHTS, "unknown tensor '%s'", name);
  Slot& s = it->second;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= dims[i];
  if (n != s.rows * s.cols) return fail(c, TTASR_E_WEIGHTS, "tensor '%s': %lld elements, expected %lld", name, (long long)n,
                                        (long long)(s.rows * s.cols));
  int64_t conv_in = 0;
  if (s.kind == 2) {  // [out][in][3] -> [out][3][in]: tap-major rows so conv == GEMM over a sliding window
    if (ndim != 3 || dims[2] != 3) return fail(c, TTASR_E_WEIGHTS, "tensor '%s': expected [out][in][3]", name);
    conv_in = dims[1];
  }
  const bool to_f32 = s.kind == 1 || s.kind == 3 || !c->lowp;
  if (!to_f32 && (size_t)n > c->stage_elems) return fail(c, TTASR_E_WEIGHTS, "tensor '%s' larger than staging", name);
  float* f32_dst = to_f32 ? (float*)s.dst : c->stage_f32;
  launch_prep_weight(src, src_type, f32_dst, n, conv_in, s.scale, c->stream);
  if (!to_f32) {
    if (c->f16) {
      launch_cast<f16_t>(c->stage_f32, (f16_t*)s.dst, n, c->stream);
      if (s.sh_base) { launch_shuffle_cast<f16_t>(c->stage_f32, (f16_t*)s.sh_base, (int)s.rows, (int)s.cols, s.sh_row_off, c->stream, s.sh_rows_total); c->weights_packed = true; }
    } else {
      launch_cast<bf16_t>(c->stage_f32, (bf16_t*)s.dst, n, c->stream);
      if (s.sh_base) { launch_shuffle_cast<bf16_t>(c->stage_f32, (bf16_t*)s.sh_base, (int)s.rows, (int)s.cols, s.sh_row_off, c->stream, s.sh_rows_total); c->weights_packed = true; }
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffers are reused by the next tensor
  HIPCHK(c, hipGetLastError());
  s.loaded = true;
  return TTASR_OK;
}

}  // namespace ttasr_detail
