// libttasr: the C ABI declared in include/ttasr.h.  One context = one GPU = one HIP stream; no hidden CPU fallback: every compute
// entry point launches the HIP kernels of this directory or fails with an error code.  The context, the schedules and the search
// loops live in engine_ctx.hpp / engine_alloc.hip / engine_sched.hip / engine_search.hip.
#include "engine_ctx.hpp"
// =====================================================================================================
// C ABI
// =====================================================================================================
// No C++ exception may cross the C ABI (std::bad_alloc from a host vector would otherwise terminate the caller's process)
template <class F>
static int guarded(ttasr_ctx* c, F&& f) {
  // a context is not re-entrant (ttasr.h): a call that arrives while another is in flight on the same context is
  // refused instead of corrupting the search state (its error text is not stored: the other call owns c->err)
  struct Busy {
    ttasr_ctx* c; bool own;
    explicit Busy(ttasr_ctx* c_) : c(c_), own(c_ == nullptr || !c_->busy.test_and_set(std::memory_order_acquire)) {}
    ~Busy() { if (c && own) c->busy.clear(std::memory_order_release); }
  } busy(c);
  if (!busy.own) return TTASR_E_INVALID;
  if (c) { g_xattn_variant = c->xattn_nt | (c->xattn_pipe << 1); g_skinny_nt = c->weights_nt; g_skinny_narrow = c->dec_narrow ? 1 : 0; g_skinny_x_lds = c->dec_x_lds ? 1 : 0; g_xattn_deep_items = c->xattn_deep_items; g_xattn_mq_slices = c->xattn_mq_slices; g_flash_qw = c->flash_qw; }   // this context's kernel variants for everything f launches
  g_launch_fault[0] = 0;
  try {
    const int rc = f();
    // a launcher that had no kernel for what it was asked (common.hpp launch_fault) launched nothing: the call's output is invalid
    if (g_launch_fault[0] && rc == TTASR_OK) return fail(c, TTASR_E_INVALID, "launcher refused: %s", g_launch_fault);
    return rc;
  } catch (const std::bad_alloc&) {
    return fail(c, TTASR_E_NOMEM, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(c, TTASR_E_INVALID, "C++ exception: %s", e.what());
  } catch (...) {
    return fail(c, TTASR_E_INVALID, "unknown C++ exception");
  }
}

extern "C" {

const char* ttasr_version(void) { return "ttasr 0.4 (gfx950, HIP; f32 | bf16 | fp16)"; }

const char* ttasr_last_error(const ttasr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int create_impl(const ttasr_config* cfg, int device_id, ttasr_ctx* owner, ttasr_ctx** out_ctx);
// `sharers` / `destroy_pending` of every context change under this lock: an owner and its last sharer may be destroyed from two
// threads at once, and exactly one of them must free the owner
static std::mutex g_share_mu;

int ttasr_create(const ttasr_config* cfg, int device_id, ttasr_ctx** out_ctx) {
  return guarded(nullptr, [&]() -> int { return create_impl(cfg, device_id, nullptr, out_ctx); });
}

int ttasr_create_shared(ttasr_ctx* owner, int32_t max_batch, ttasr_ctx** out_ctx) {
  return guarded(nullptr, [&]() -> int {
  if (!owner || !out_ctx) return fail(nullptr, TTASR_E_INVALID, "owner/out_ctx is NULL");
  *out_ctx = nullptr;
  if (owner->weight_owner) owner = owner->weight_owner;   // sharing with a sharer = sharing with its owner
  if (!owner->finalized) return fail(nullptr, TTASR_E_INVALID, "the owner's weights are not finalized (ttasr_finalize_weights first)");
  { std::lock_guard<std::mutex> lk(g_share_mu);
    if (owner->destroy_pending) return fail(nullptr, TTASR_E_INVALID, "the owner context was destroyed"); }
  ttasr_config cfg = owner->cfg;
  if (max_batch > 0) cfg.max_batch = max_batch;
  return create_impl(&cfg, owner->device, owner, out_ctx);
  });
}

// the weight pointers of `o` (and what the kernels need to know about their layout) into `c`: nothing is copied on the device
static void adopt_weights(ttasr_ctx* c, ttasr_ctx* o) {
  c->conv1_w = o->conv1_w; c->conv2_w = o->conv2_w; c->emb = o->emb; c->dpos = o->dpos; c->emb_sh = o->emb_sh;
  c->conv1_b = o->conv1_b; c->conv2_b = o->conv2_b; c->epos = o->epos; c->elnf_g = o->elnf_g; c->elnf_b = o->elnf_b;
  c->dlnf_g = o->dlnf_g; c->dlnf_b = o->dlnf_b;
  c->enc = o->enc; c->dec = o->dec;
  c->dec_narrow = o->dec_narrow; c->weights_packed = o->weights_packed;
  c->finalized = true;
  c->weight_owner = o;
  std::lock_guard<std::mutex> lk(g_share_mu);
  o->sharers.fetch_add(1);
}

static int create_impl(const ttasr_config* cfg, int device_id, ttasr_ctx* owner, ttasr_ctx** out_ctx) {
  {
  if (!cfg || !out_ctx) return fail(nullptr, TTASR_E_INVALID, "cfg/out_ctx is NULL");
  *out_ctx = nullptr;
  if (cfg->d_model <= 0 || cfg->n_heads <= 0 || cfg->d_model != cfg->n_heads * 64)
    return fail(nullptr, TTASR_E_INVALID, "head_dim must be 64 (d_model=%d, n_heads=%d)", cfg->d_model, cfg->n_heads);
  if (cfg->d_model > 1280) return fail(nullptr, TTASR_E_INVALID, "d_model %d > 1280 (LayerNorm keeps a row in registers)", cfg->d_model);
  if (cfg->vocab > 13 * 4096)  // kernels_decode.hip LOGIT_NIT: Whisper vocabularies are 51 864 .. 51 866
    return fail(nullptr, TTASR_E_INVALID, "vocab %d > 53248 (the token-selection kernels keep a logits row in registers)", cfg->vocab);
  if (cfg->n_mels % 8 || cfg->n_mels <= 0 || cfg->ffn_dim % 64 || cfg->n_audio_ctx < 1 || cfg->vocab < 2 ||
      cfg->n_text_ctx < 2 || cfg->n_text_ctx > 448 || cfg->enc_layers < 1 || cfg->dec_layers < 1 || cfg->max_batch < 1)
    return fail(nullptr, TTASR_E_INVALID, "unsupported geometry");
  if (cfg->compute_type != TTASR_COMPUTE_F32 && cfg->compute_type != TTASR_COMPUTE_BF16 && cfg->compute_type != TTASR_COMPUTE_F16)
    return fail(nullptr, TTASR_E_INVALID, "compute_type must be 0 (f32), 1 (bf16) or 2 (fp16)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, TTASR_E_HIP, "no HIP device available (%s); libttasr has no CPU fallback", hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, TTASR_E_INVALID, "device %d of %d", device_id, ndev);
  std::unique_ptr<ttasr_ctx> c(new ttasr_ctx());
  c->cfg = *cfg; c->device = device_id;
  c->lowp = cfg->compute_type != TTASR_COMPUTE_F32; c->f16 = cfg->compute_type == TTASR_COMPUTE_F16; c->esz = c->lowp ? 2 : 4;
  c->T = cfg->n_audio_ctx; c->F = 2 * c->T; c->d = cfg->d_model; c->H = cfg->n_heads; c->ffn = cfg->ffn_dim;
  c->V = cfg->vocab; c->ldv = (cfg->vocab + 63) / 64 * 64; c->M = cfg->n_mels; c->maxB = cfg->max_batch;
  c->n_samples = c->F * 160;
  // The release library reads NO environment variable: every kernel-selection override goes through ttasr_set_option (an
  // explicit call a test or a measurement script makes).  Experiment builds (-DTTASR_EXPERIMENTS) additionally map the old
  // TTASR_* variables onto the same options, after the context exists (below).
  ttasr_ctx* p = c.get();
  auto die = [&](int rc) { g_create_error = p->err; ttasr_destroy(c.release()); return rc; };
  if (hipSetDevice(device_id) != hipSuccess) return die(fail(p, TTASR_E_HIP, "hipSetDevice(%d) failed", device_id));
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess)
    return die(fail(p, TTASR_E_HIP, "hipStreamCreate failed"));
  p->cur = p->stream;
  gemm_vocab_init(device_id);
  gemm_tiles_init(device_id);
  {  // weights (+ packed decoder copies) + encoder workspaces + cross-KV + self-KV pool, in elements of the compute type
    const size_t d = p->d, ffn = p->ffn, T = p->T, B = p->maxB;
    const size_t w = owner ? 0 : ((size_t)cfg->enc_layers * (4 * d * d + 2 * d * ffn) + (size_t)cfg->dec_layers * (8 * d * d + 2 * d * ffn) * 2 + 2 * (size_t)p->V * d);
    const size_t act = B * T * (4 * d + 3 * d + ffn + 2 * d) + (size_t)cfg->dec_layers * 2 * B * T * d + (size_t)cfg->dec_layers * 2 * B * cfg->n_text_ctx * d;
    p->arena_hint = (w + act) * p->esz;
  }
  int rc = 0;
  if (owner) adopt_weights(p, owner);   // before anything can fail below: ttasr_destroy (die) releases the share again
  else rc = build_weights(p);
  if (rc) return die(rc);
  rc = build_workspaces(p);
  if (rc) return die(rc);
#ifdef TTASR_EXPERIMENTS
  {  // experiment builds only: the environment switches of the lab notebook (tools/microbench/README.md)
    static const struct { const char* env; const char* key; int on; } flags[] = {
        {"TTASR_FORCE_BASIC", "generic_kernels", 1}, {"TTASR_NO_GRAPH", "graph", 0},          {"TTASR_NO_PREFILL", "prefill", 0},
        {"TTASR_PREFILL_TILED", "prefill_tiled", 1}, {"TTASR_NO_XSPLIT", "xsplit", 0},        {"TTASR_NO_FLASH", "flash", 0},
        {"TTASR_ENC_RES_EPI", "enc_residual_epilogue", 1}};
    for (auto& f : flags) if (getenv(f.env)) set_option(p, f.key, f.on);
    if (getenv("TTASR_SKIP")) p->skip_mask = atoi(getenv("TTASR_SKIP"));
    if (const char* v = getenv("TTASR_KS")) {
      int ks[4] = {0, 0, 0, 0};
      sscanf(v, "%d,%d,%d,%d", &ks[0], &ks[1], &ks[2], &ks[3]);
      set_option(p, "ksplit_out", ks[0]); set_option(p, "ksplit_q", ks[1]); set_option(p, "ksplit_qkv", ks[2]); set_option(p, "ksplit_fc2", ks[3]);
    }
    if (const char* v = getenv("TTASR_PREFILL_NS_MIN")) set_option(p, "prefill_ns_min", atoi(v));
    if (const char* v = getenv("TTASR_GEMM")) set_option(p, "enc_gemm", (v[0] == 'v' && v[1] >= '1' && v[1] <= '3') ? v[1] - '0' : 0);
    if (const char* v = getenv("TTASR_XATTN")) set_option(p, "xattn_nontemporal", atoi(v) & 1);
    if (const char* v = getenv("TTASR_W_NT")) set_option(p, "weights_nontemporal", atoi(v));
  }
#endif
  *out_ctx = c.release();
  return TTASR_OK;
  }
}

int ttasr_set_option(ttasr_ctx* c, const char* key, int32_t value) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (!key) return fail(c, TTASR_E_INVALID, "key is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int rc = set_option(c, key, value);
  if (rc == 2) return TTASR_E_NOMEM;   // an allocation of the option failed: dalloc's message is the error text
  if (rc) return fail(c, TTASR_E_INVALID, "unknown option '%s' (or value %d out of range)", key, value);
  return TTASR_OK;
  });
}

void ttasr_destroy(ttasr_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  { std::lock_guard<std::mutex> lk(g_share_mu);
    if (c->sharers.load() > 0 && !c->destroy_pending) {   // other contexts still read these weights: the last of them frees this one
      c->destroy_pending = true;
      return;
    } }
  ttasr_ctx* owner = c->weight_owner;
  drop_graphs(c);
  for (auto& e : c->ev) if (e) hipEventDestroy(e);
  for (auto& e : c->enc_ev) hipEventDestroy(e);
  for (void* p : c->allocs) hipFree(p);
  if (c->pinned_i32) hipHostFree(c->pinned_i32);
  if (c->pinned_beam) hipHostFree(c->pinned_beam);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  if (owner) {
    bool last;
    { std::lock_guard<std::mutex> lk(g_share_mu); last = owner->sharers.fetch_sub(1) == 1 && owner->destroy_pending; }
    if (last) ttasr_destroy(owner);   // we were the last sharer of an owner that is already gone for its user
  }
}



static int weights_writable(ttasr_ctx* c) {
  if (c->weight_owner) return fail(c, TTASR_E_INVALID, "this context shares another context's weights (ttasr_create_shared): they are read-only");
  if (c->sharers.load() > 0) return fail(c, TTASR_E_INVALID, "%d context(s) share these weights: they are read-only", c->sharers.load());
  return 0;
}

int ttasr_load_tensor(ttasr_ctx* c, const char* name, const float* data, const int64_t* dims, int32_t ndim) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  TRY(weights_writable(c));
  if (!name || !data || !dims) return fail(c, TTASR_E_INVALID, "NULL argument");
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= dims[i];
  if (n < 0 || (size_t)n > c->stage_elems) return fail(c, TTASR_E_WEIGHTS, "tensor '%s': %lld elements exceed the staging buffer", name, (long long)n);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->stage_raw, data, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  return ingest_tensor(c, name, c->stage_raw, 0, dims, ndim);
  });
}

int ttasr_load_tensor_device(ttasr_ctx* c, const char* name, const void* data_dev, int32_t dtype, const int64_t* dims, int32_t ndim) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  TRY(weights_writable(c));
  if (!name || !data_dev || !dims) return fail(c, TTASR_E_INVALID, "NULL argument");
  if (dtype != TTASR_DTYPE_F32 && dtype != TTASR_DTYPE_BF16 && dtype != TTASR_DTYPE_F16)
    return fail(c, TTASR_E_INVALID, "dtype must be 0 (float32), 1 (bfloat16 bits) or 2 (float16 bits)");
  HIPCHK(c, hipSetDevice(c->device));
  return ingest_tensor(c, name, data_dev, dtype, dims, ndim);
  });
}

int ttasr_finalize_weights(ttasr_ctx* c) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  for (auto& kv : c->slots)
    if (!kv.second.loaded) return fail(c, TTASR_E_WEIGHTS, "tensor '%s' was never loaded", kv.first.c_str());
  c->finalized = true;
  return TTASR_OK;
  });
}

int ttasr_log_mel(ttasr_ctx* c, const float* pcm, int64_t pcm_stride, const int64_t* n_samples, int32_t B,
                  int32_t on_device, float* out_mel) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (B < 1 || B > c->maxB) return fail(c, TTASR_E_INVALID, "batch %d outside [1, %d]", B, c->maxB);
  if (!pcm || !n_samples) return fail(c, TTASR_E_INVALID, "pcm / n_samples is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const float* src = pcm;
  int64_t stride = pcm_stride;
  std::vector<int64_t> ns(n_samples, n_samples + B);
  for (int b = 0; b < B; ++b) {
    if (ns[b] < 0 || ns[b] > pcm_stride) return fail(c, TTASR_E_INVALID, "n_samples[%d]=%lld outside [0, stride]", b, (long long)ns[b]);
    if (ns[b] > c->n_samples) ns[b] = c->n_samples;  // trim to one window
  }
  hipEventRecord(c->ev[0], s);
  if (!on_device) {
    bool whole = pcm_stride == c->n_samples;   // full windows back to back (the batch path): ONE copy instead of B
    for (int b = 0; b < B && whole; ++b) whole = ns[b] == c->n_samples;
    // (Round 5 also built the copy in four chunks on a second stream with the log-mel of chunk i under the copy of chunk i + 1:
    // mel phase 2.3 -> 1.7 ms - and removed it: a context that owns a SECOND hardware queue makes two PROCESSES sharing one GPU
    // time-slice instead of overlapping; their decode chains ran 3.2x slower, 565 -> 1 805 ms per 131 steps each.  DESIGN 4.11.)
    if (whole) {
      HIPCHK(c, hipMemcpyAsync(c->pcm_dev, pcm, (size_t)B * c->n_samples * 4, hipMemcpyHostToDevice, s));
    } else {
      for (int b = 0; b < B; ++b)
        if (ns[b] > 0)
          HIPCHK(c, hipMemcpyAsync(c->pcm_dev + (int64_t)b * c->n_samples, pcm + b * pcm_stride, ns[b] * 4, hipMemcpyHostToDevice, s));
    }
    src = c->pcm_dev; stride = c->n_samples;
  }
  HIPCHK(c, hipMemcpyAsync(c->nsamp_dev, ns.data(), B * 8, hipMemcpyHostToDevice, s));
  launch_mel(src, stride, c->nsamp_dev, B, c->M, c->F, c->filters, c->dcos, c->dsin, c->window, c->mel, c->clip_max, s);
  TT_DISPATCH(c, launch_mel_finish<T>(c->mel, c->clip_max, (T*)c->mel_t, B, c->M, c->F, s));
  hipEventRecord(c->ev[1], s);
  if (out_mel) HIPCHK(c, hipMemcpyAsync(out_mel, c->mel, (size_t)B * c->M * c->F * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  hipEventElapsedTime(&c->phase_ms[0], c->ev[0], c->ev[1]);
  c->B_mel = B;
  return TTASR_OK;
  });
}

int ttasr_log_mel_windows(ttasr_ctx* c, const float* const* file_pcm_of, const int64_t* file_samples_of, const int64_t* seek_frames,
                          int32_t B, const float* floor_max, float* out_window_max, float* out_mel) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (B < 1 || B > c->maxB) return fail(c, TTASR_E_INVALID, "batch %d outside [1, %d]", B, c->maxB);
  if (!file_pcm_of || !file_samples_of || !seek_frames) return fail(c, TTASR_E_INVALID, "NULL argument");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int64_t stride = c->n_samples + 512;
  std::vector<int64_t> ns(B), geom((size_t)3 * B);
  hipEventRecord(c->ev[0], s);
  for (int b = 0; b < B; ++b) {
    const float* file_pcm = file_pcm_of[b];
    const int64_t file_samples = file_samples_of[b];
    if (!file_pcm || file_samples < 0) return fail(c, TTASR_E_INVALID, "window %d: recording is NULL / negative length", b);
    const int64_t file_frames = file_samples / 160;   // the whole-file STFT drops its last frame (HF feature extractor :154)
    const int64_t seek = seek_frames[b];
    if (seek < 0 || seek > file_frames) return fail(c, TTASR_E_INVALID, "seek_frames[%d]=%lld outside the recording (%lld frames)", b, (long long)seek, (long long)file_frames);
    const int64_t start = seek * 160, lead = std::min<int64_t>(200, start);
    const int64_t avail = std::min<int64_t>(file_samples - (start - lead), lead + c->n_samples + 200);  // samples from x[0]
    ns[b] = std::max<int64_t>(avail, 0);
    geom[3 * b] = lead;
    // reflect where the FILE ends if that is inside the span this window's frames touch; otherwise never
    geom[3 * b + 1] = (file_samples - (start - lead) < lead + c->n_samples + 200) ? file_samples - (start - lead) : ((int64_t)1 << 40);
    geom[3 * b + 2] = std::min<int64_t>(c->F, file_frames - seek);
    if (ns[b] > 0) HIPCHK(c, hipMemcpyAsync(c->pcm_dev + (int64_t)b * stride, file_pcm + (start - lead), ns[b] * 4, hipMemcpyHostToDevice, s));
  }
  HIPCHK(c, hipMemcpyAsync(c->nsamp_dev, ns.data(), B * 8, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->mel_geom, geom.data(), (size_t)B * 24, hipMemcpyHostToDevice, s));
  launch_mel(c->pcm_dev, stride, c->nsamp_dev, B, c->M, c->F, c->filters, c->dcos, c->dsin, c->window, c->mel, c->clip_max, s, c->mel_geom);
  std::vector<unsigned> mx(B);
  if (out_window_max) {
    HIPCHK(c, hipMemcpyAsync(mx.data(), c->clip_max, B * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b) out_window_max[b] = mel_max_from_ordered(mx[b]);
  }
  if (floor_max) {  // the whole-file maximum decides the dynamic-range floor of every window
    for (int b = 0; b < B; ++b) mx[b] = mel_max_to_ordered(floor_max[b]);
    HIPCHK(c, hipMemcpyAsync(c->clip_max, mx.data(), B * 4, hipMemcpyHostToDevice, s));
  }
  TT_DISPATCH(c, launch_mel_finish<T>(c->mel, c->clip_max, (T*)c->mel_t, B, c->M, c->F, s, c->mel_geom));
  hipEventRecord(c->ev[1], s);
  if (out_mel) HIPCHK(c, hipMemcpyAsync(out_mel, c->mel, (size_t)B * c->M * c->F * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));  // mx / ns / geom are stack temporaries
  HIPCHK(c, hipGetLastError());
  hipEventElapsedTime(&c->phase_ms[0], c->ev[0], c->ev[1]);
  c->B_mel = B;
  return TTASR_OK;
  });
}

int ttasr_set_mel(ttasr_ctx* c, const float* mel, int32_t B) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (B < 1 || B > c->maxB || !mel) return fail(c, TTASR_E_INVALID, "bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->mel, mel, (size_t)B * c->M * c->F * 4, hipMemcpyHostToDevice, c->stream));
  TT_DISPATCH(c, launch_mel_transpose<T>(c->mel, (T*)c->mel_t, B, c->M, c->F, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->B_mel = B;
  return TTASR_OK;
  });
}

int ttasr_encode(ttasr_ctx* c, int32_t B, float* out_enc) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (c->B_mel < B) return fail(c, TTASR_E_INVALID, "mel for %d clips requested but only %d resident", B, c->B_mel);
  sched_encoder(c, B);
  if (out_enc) {
    const int64_t n = (int64_t)B * c->T * c->d;
    if (c->lowp) { TT_DISPATCH(c, launch_uncast<T>((const T*)c->enc_out, c->x, n, c->stream));
                   HIPCHK(c, hipMemcpyAsync(out_enc, c->x, n * 4, hipMemcpyDeviceToHost, c->stream)); }
    else HIPCHK(c, hipMemcpyAsync(out_enc, c->enc_out, n * 4, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  hipEventElapsedTime(&c->phase_ms[1], c->ev[2], c->ev[3]);
  hipEventElapsedTime(&c->phase_ms[2], c->ev[3], c->ev[4]);
  if (c->enc_timing) {
    for (float& v : c->enc_class_ms) v = 0.f;
    for (size_t i = 1; i < c->enc_ev_class.size(); ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, c->enc_ev[i - 1], c->enc_ev[i]);
      if (c->enc_ev_class[i] >= 0 && c->enc_ev_class[i] < 8) c->enc_class_ms[c->enc_ev_class[i]] += ms;
    }
  }
  c->B_enc = B;
  return TTASR_OK;
  });
}

int ttasr_encoder_kernel_ms(ttasr_ctx* c, float out[8]) {
  return guarded(c, [&]() -> int {
  if (!c || !out) return TTASR_E_INVALID;
  if (!c->enc_timing) return fail(c, TTASR_E_INVALID, "set option enc_kernel_timing = 1 and run ttasr_encode first");
  for (int i = 0; i < 8; ++i) out[i] = c->enc_class_ms[i];
  return TTASR_OK;
  });
}

int ttasr_set_encoder_output(ttasr_ctx* c, const float* enc, int32_t B) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (!enc) return fail(c, TTASR_E_INVALID, "enc is NULL");
  const int64_t n = (int64_t)B * c->T * c->d;
  if (c->lowp) { HIPCHK(c, hipMemcpyAsync(c->x, enc, n * 4, hipMemcpyHostToDevice, c->stream));
                 TT_DISPATCH(c, launch_cast<T>(c->x, (T*)c->enc_out, n, c->stream)); }
  else HIPCHK(c, hipMemcpyAsync(c->enc_out, enc, n * 4, hipMemcpyHostToDevice, c->stream));
  c->cur = c->stream;
  c->enc_ev_class.clear();   // in-situ timing marks belong to ONE pass: start a fresh list (they used to pile up here)
  enc_mark(c, -1);
  sched_cross_kv(c, B);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  c->B_enc = B;
  return TTASR_OK;
  });
}

int ttasr_get_cross_kv(ttasr_ctx* c, int32_t layer, int32_t which, int32_t B, float* out) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (layer < 0 || layer >= c->cfg.dec_layers || which < 0 || which > 1 || !out || B > c->B_enc)
    return fail(c, TTASR_E_INVALID, "bad arguments");
  const int64_t n = (int64_t)B * c->H * c->T * 64;
  const char* src = (const char*)c->xkv + ((size_t)layer * c->xkv_layer_elems + (size_t)which * c->xkv_which_elems) * c->esz;
  if (c->lowp) { TT_DISPATCH(c, launch_uncast<T>((const T*)src, c->x, n, c->stream));
                 HIPCHK(c, hipMemcpyAsync(out, c->x, n * 4, hipMemcpyDeviceToHost, c->stream)); }
  else HIPCHK(c, hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TTASR_OK;
  });
}

int ttasr_decode_reset(ttasr_ctx* c, int32_t B) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  TRY(reset_search(c, B));
  c->st.prompt = nullptr; c->st.prompt_len = nullptr;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B_dec = B;
  return TTASR_OK;
  });
}

int ttasr_decode_step(ttasr_ctx* c, const int32_t* tokens, int32_t B, float* logits) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (!tokens) return fail(c, TTASR_E_INVALID, "tokens is NULL");
  if (B != c->B_dec || B > c->B_enc) return fail(c, TTASR_E_INVALID, "call ttasr_encode and ttasr_decode_reset(B) first");
  for (int b = 0; b < B; ++b)
    if (tokens[b] < 0 || tokens[b] >= c->V) return fail(c, TTASR_E_INVALID, "token %d outside vocabulary", tokens[b]);
  HIPCHK(c, hipMemcpyAsync(c->pinned_i32, c->st.step, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->pinned_i32[0] >= c->cfg.n_text_ctx) return fail(c, TTASR_E_INVALID, "decoder context (%d) exhausted", c->cfg.n_text_ctx);
  HIPCHK(c, hipMemcpyAsync(c->st.cur_tok, tokens, B * 4, hipMemcpyHostToDevice, c->stream));
  TRY(step_graph(c, B, 1));
  if (logits) HIPCHK(c, hipMemcpy2DAsync(logits, (size_t)c->V * 4, c->logits, (size_t)c->ldv * 4, (size_t)c->V * 4, B,
                                         hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return TTASR_OK;
  });
}

int ttasr_generate(ttasr_ctx* c, int32_t B, const int32_t* prompt, const int32_t* prompt_len, int32_t max_prompt,
                   const ttasr_gen_opts* o, int32_t* out_tokens, int32_t* out_len, float* out_lp, float* out_ns);



int ttasr_generate(ttasr_ctx* c, int32_t B, const int32_t* prompt, const int32_t* prompt_len, int32_t max_prompt,
                   const ttasr_gen_opts* o, int32_t* out_tokens, int32_t* out_len, float* out_lp, float* out_ns) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (!prompt || !prompt_len || !out_tokens || !out_len || !o) return fail(c, TTASR_E_INVALID, "NULL argument");
  if (B > c->B_enc) return fail(c, TTASR_E_INVALID, "encoder state holds %d clips, %d requested", c->B_enc, B);
  if (max_prompt < 1 || max_prompt > c->max_prompt_alloc) return fail(c, TTASR_E_INVALID, "max_prompt %d", max_prompt);
  return generate_rows(c, B, 1, prompt, prompt_len, max_prompt, o, 0.f, 0, out_tokens, out_len, out_lp, out_ns);
  });
}

int ttasr_generate_capped(ttasr_ctx* c, int32_t B, const int32_t* prompt, const int32_t* prompt_len, int32_t max_prompt,
                          const ttasr_gen_opts* o, const int32_t* row_max_new, int32_t* out_tokens, int32_t* out_len, float* out_lp,
                          float* out_ns) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (!prompt || !prompt_len || !out_tokens || !out_len || !o || !row_max_new) return fail(c, TTASR_E_INVALID, "NULL argument");
  if (B > c->B_enc) return fail(c, TTASR_E_INVALID, "encoder state holds %d clips, %d requested", c->B_enc, B);
  if (max_prompt < 1 || max_prompt > c->max_prompt_alloc) return fail(c, TTASR_E_INVALID, "max_prompt %d", max_prompt);
  return generate_rows(c, B, 1, prompt, prompt_len, max_prompt, o, 0.f, 0, out_tokens, out_len, out_lp, out_ns, row_max_new);
  });
}

int ttasr_generate_sample(ttasr_ctx* c, int32_t A, int32_t best_of, const int32_t* prompt, int32_t plen, const ttasr_gen_opts* o,
                          float temperature, uint32_t seed, int32_t* out_tokens, int32_t* out_len, float* out_lp, float* out_ns) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (A < 1 || best_of < 1 || !(temperature > 0.f)) return fail(c, TTASR_E_INVALID, "n_audio, best_of >= 1 and temperature > 0 required");
  const int R = A * best_of;
  TRY(check_ready(c, R));
  if (!prompt || !out_tokens || !out_len || !o) return fail(c, TTASR_E_INVALID, "NULL argument");
  if (A > c->B_enc) return fail(c, TTASR_E_INVALID, "encoder state holds %d clips, %d requested", c->B_enc, A);
  if (plen < 1 || plen > c->max_prompt_alloc) return fail(c, TTASR_E_INVALID, "prompt_len %d", plen);
  const int max_new = o->max_new_tokens;
  std::vector<int32_t> toks((size_t)R * std::max(max_new, 1)), lens(R), plens(A, plen);
  std::vector<float> lp(R), ns(R);
  TRY(generate_rows(c, R, best_of, prompt, plens.data(), plen, o, temperature, seed, toks.data(), lens.data(), lp.data(), ns.data()));
  for (int a = 0; a < A; ++a) {
    int best = a * best_of;
    for (int r = a * best_of; r < (a + 1) * best_of; ++r)
      if (lp[r] / std::max(lens[r], 1) > lp[best] / std::max(lens[best], 1)) best = r;
    memcpy(out_tokens + (size_t)a * max_new, &toks[(size_t)best * max_new], (size_t)max_new * 4);
    out_len[a] = lens[best];
    if (out_lp) out_lp[a] = lp[best];
    if (out_ns) out_ns[a] = ns[a * best_of];
  }
  return TTASR_OK;
  });
}



int ttasr_generate_beam(ttasr_ctx* c, int32_t A, int32_t beam, const int32_t* prompt, int32_t plen, const ttasr_gen_opts* o,
                        float patience, int32_t* out_tokens, int32_t* out_len, float* out_lp, float* out_ns) {
  return guarded(c, [&]() -> int {
    if (A < 1) return fail(c, TTASR_E_INVALID, "n_audio >= 1 required");
    std::vector<int32_t> plens(A, plen);
    return beam_search_impl(c, A, beam, prompt, plen, plens.data(), nullptr, o, patience, out_tokens, out_len, out_lp, out_ns);
  });
}

int ttasr_generate_beam_ragged(ttasr_ctx* c, int32_t A, int32_t beam, const int32_t* prompt, const int32_t* prompt_len,
                               const int32_t* sot_index, int32_t max_prompt, const ttasr_gen_opts* o, float patience,
                               int32_t* out_tokens, int32_t* out_len, float* out_lp, float* out_ns) {
  return guarded(c, [&]() -> int {
    return beam_search_impl(c, A, beam, prompt, max_prompt, prompt_len, sot_index, o, patience, out_tokens, out_len, out_lp, out_ns);
  });
}

int ttasr_apply_rules(ttasr_ctx* c, const float* rows, const int32_t* hist, int32_t hist_stride, int32_t n,
                      const ttasr_gen_opts* o, float* out_rows, int32_t* out_choice) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (n < 1 || n > c->maxB || !rows || !hist || !out_rows) return fail(c, TTASR_E_INVALID, "bad arguments (n <= max_batch)");
  HIPCHK(c, hipSetDevice(c->device));
  RuleParams old = c->rp;
  ttasr_gen_opts oo = *o;
  oo.max_new_tokens = std::max(1, std::min(oo.max_new_tokens, c->max_new_alloc));
  TRY(upload_rules(c, &oo, 1));
  TRY(commit_rules(c, old));
  TRY(reset_search(c, n));
  std::vector<int32_t> ns(n), last(n, -1), pen(n, -1), lts(n, -1);
  for (int r = 0; r < n; ++r) {
    int k = 0;
    for (; k < hist_stride && hist[r * hist_stride + k] >= 0; ++k) {
      int t = hist[r * hist_stride + k];
      pen[r] = last[r]; last[r] = t;
      if (t >= o->timestamp_begin) lts[r] = t;
    }
    ns[r] = k;
  }
  hipStream_t s = c->stream;
  HIPCHK(c, hipMemcpyAsync(c->st.n_sampled, ns.data(), n * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->st.last_tok, last.data(), n * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->st.pen_tok, pen.data(), n * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->st.last_ts, lts.data(), n * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpy2DAsync(c->logits, (size_t)c->ldv * 4, rows, (size_t)c->V * 4, (size_t)c->V * 4, n, hipMemcpyHostToDevice, s));
  if (!c->rows_out) TRY(dalloc(c, &c->rows_out, (size_t)c->maxB * c->V * 4));
  DecState st = c->st; st.prompt = nullptr; st.prompt_len = nullptr;
  RuleParams rp = c->rp; rp.max_new = c->max_new_alloc;  // histories may be longer than opts.max_new_tokens
  c->rule_dyn_host.max_new = rp.max_new;
  HIPCHK(c, hipMemcpyAsync(c->rule_dyn_dev, &c->rule_dyn_host, sizeof(RuleDyn), hipMemcpyHostToDevice, s));
  launch_select(c->logits, st, rp, n, c->rows_out, s, nullptr, 0);
  HIPCHK(c, hipMemcpyAsync(out_rows, c->rows_out, (size_t)n * c->V * 4, hipMemcpyDeviceToHost, s));
  if (out_choice) HIPCHK(c, hipMemcpyAsync(out_choice, c->st.cur_tok, n * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return TTASR_OK;
  });
}

int ttasr_beam_profile(ttasr_ctx* c, float out[4]) {
  return guarded(c, [&]() -> int {
  if (!c || !out) return TTASR_E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = c->beam_prof_ms[i];
  return TTASR_OK;
  });
}

int ttasr_phase_ms(ttasr_ctx* c, float out[4]) {
  return guarded(c, [&]() -> int {
  if (!c || !out) return TTASR_E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = c->phase_ms[i];
  return TTASR_OK;
  });
}

int ttasr_sync(ttasr_ctx* c) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TTASR_OK;
  });
}

int ttasr_align(ttasr_ctx* c, int32_t clip, const int32_t* tokens, int32_t n_tok, const int32_t* pairs, int32_t n_pairs,
                float* out_weights, float* out_logprob) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, 1));
  if (!tokens || !pairs || !out_weights) return fail(c, TTASR_E_INVALID, "NULL argument");
  if (clip < 0 || clip >= c->B_enc) return fail(c, TTASR_E_INVALID, "clip %d but the encoder state holds %d", clip, c->B_enc);
  if (n_tok < 2 || n_tok > c->cfg.n_text_ctx || n_tok > c->cfg.n_audio_ctx)
    return fail(c, TTASR_E_INVALID, "n_tokens %d outside [2, min(n_text_ctx, n_audio_ctx)]", n_tok);
  if (n_pairs < 1 || n_pairs > c->cfg.dec_layers * c->H) return fail(c, TTASR_E_INVALID, "n_pairs %d", n_pairs);
  for (int i = 0; i < n_tok; ++i)
    if (tokens[i] < 0 || tokens[i] >= c->V) return fail(c, TTASR_E_INVALID, "token outside vocabulary");
  std::vector<int32_t> sel((size_t)c->cfg.dec_layers * c->H, -1);
  for (int i = 0; i < n_pairs; ++i) {
    const int l = pairs[2 * i], h = pairs[2 * i + 1];
    if (l < 0 || l >= c->cfg.dec_layers || h < 0 || h >= c->H) return fail(c, TTASR_E_INVALID, "alignment head (%d, %d)", l, h);
    if (sel[(size_t)l * c->H + h] >= 0) return fail(c, TTASR_E_INVALID, "alignment head (%d, %d) listed twice", l, h);
    sel[(size_t)l * c->H + h] = i;
  }
  hipStream_t s = c->stream;
  const size_t n_w = (size_t)n_pairs * n_tok * c->T;
  float* probs = nullptr; int32_t* sel_dev = nullptr; float* lp_dev = nullptr;
  struct Free { void** p; ~Free() { if (*p) hipFree(*p); } } f1{(void**)&probs}, f2{(void**)&sel_dev}, f3{(void**)&lp_dev};
  if (hipMalloc(&probs, n_w * 4) != hipSuccess || hipMalloc(&sel_dev, sel.size() * 4) != hipSuccess ||
      hipMalloc(&lp_dev, (size_t)n_tok * 4) != hipSuccess)
    return fail(c, TTASR_E_NOMEM, "alignment buffers (%zu bytes)", n_w * 4);
  HIPCHK(c, hipMemcpyAsync(sel_dev, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->prompt_dev, tokens, (size_t)n_tok * 4, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipStreamSynchronize(s));  // sel is a stack temporary
  c->B_dec = 0;  // the pass reuses sequence 0's self-attention pages: any step-level decode state is gone
  AlignOut al{clip, sel_dev, probs};
  sched_prefill(c, 1, n_tok, 1, n_tok, &al);
  if (out_logprob) {
    // raw log p(tokens[i + 1] | tokens[0..i]): final LayerNorm + vocabulary projection, max_batch rows at a time
    for (int r0 = 0; r0 < n_tok - 1; r0 += c->maxB) {
      const int n = std::min(c->maxB, n_tok - 1 - r0);
      c->cur = s;
      TT_DISPATCH(c, {
        launch_layernorm<T>(c->x + (size_t)r0 * c->d, c->dlnf_g, c->dlnf_b, (T*)c->dh, n, c->d, s);
        GemmArgs g = lin_args<T>(c->dh, c->emb, n, c->V, c->d); g.epi.out_f32 = c->logits; g.epi.ldc = c->ldv;
        sched_dec_gemm(c, g, c->emb_sh);
      });
      launch_token_logprob(c->logits, c->ldv, c->V, c->prompt_dev + r0 + 1, lp_dev + r0, n, s);
    }
    HIPCHK(c, hipMemcpyAsync(out_logprob, lp_dev, (size_t)(n_tok - 1) * 4, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(c, hipMemcpyAsync(out_weights, probs, n_w * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return TTASR_OK;
  });
}

int ttasr_dtw(const float* cost, int32_t n_rows, int32_t n_cols, int32_t* out_row, int32_t* out_col, int32_t* out_len) {
  return guarded(nullptr, [&]() -> int {
  if (!cost || !out_row || !out_col || !out_len || n_rows < 1 || n_cols < 1) return TTASR_E_INVALID;
  const size_t W_ = (size_t)n_cols + 1;
  std::vector<float> acc(((size_t)n_rows + 1) * W_, INFINITY);
  std::vector<int8_t> trace(((size_t)n_rows + 1) * W_, -1);
  acc[0] = 0.f;
  for (int j = 1; j <= n_cols; ++j)
    for (int i = 1; i <= n_rows; ++i) {
      const float c0 = acc[(size_t)(i - 1) * W_ + j - 1], c1 = acc[(size_t)(i - 1) * W_ + j], c2 = acc[(size_t)i * W_ + j - 1];
      float best; int8_t t;
      if (c0 < c1 && c0 < c2) { best = c0; t = 0; } else if (c1 < c0 && c1 < c2) { best = c1; t = 1; } else { best = c2; t = 2; }
      acc[(size_t)i * W_ + j] = cost[(size_t)(i - 1) * n_cols + j - 1] + best;
      trace[(size_t)i * W_ + j] = t;
    }
  for (int j = 0; j <= n_cols; ++j) trace[j] = 2;
  for (int i = 0; i <= n_rows; ++i) trace[(size_t)i * W_] = 1;
  int i = n_rows, j = n_cols, n = 0;
  while (i > 0 || j > 0) {  // at most n_rows + n_cols entries, written back to front then reversed
    out_row[n] = i - 1; out_col[n] = j - 1; ++n;
    const int8_t t = trace[(size_t)i * W_ + j];
    if (t == 0) { --i; --j; } else if (t == 1) --i; else --j;
  }
  std::reverse(out_row, out_row + n);
  std::reverse(out_col, out_col + n);
  *out_len = n;
  return TTASR_OK;
  });
}

int ttasr_set_audio_ctx(ttasr_ctx* c, int32_t n_ctx) {
  return guarded(c, [&]() -> int {
  if (!c) return TTASR_E_INVALID;
  if (n_ctx == 0) n_ctx = c->cfg.n_audio_ctx;
  if (n_ctx < 4 || n_ctx > c->cfg.n_audio_ctx || (n_ctx & 1))
    return fail(c, TTASR_E_INVALID, "audio context %d outside [4, %d] or odd", n_ctx, c->cfg.n_audio_ctx);
  if (n_ctx == c->T) return TTASR_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // every buffer is sized for the model's full context; a shorter window only changes the strides inside them.
  // Captured decode graphs hold the old strides, resident mel / encoder state is for the old window: drop both.
  drop_graphs(c);
  c->T = n_ctx; c->F = 2 * n_ctx; c->n_samples = c->F * 160;
  c->B_mel = c->B_enc = c->B_dec = 0;
  // the conv stem reads one zero row before each clip's first frame; those rows sit at clip stride (F + 2), so
  // they move with the window: clear the padded images (conv1 only ever writes rows 1..F of each clip)
  const size_t Fmax = 2 * (size_t)c->cfg.n_audio_ctx + 2;
  HIPCHK(c, hipMemsetAsync(c->c1, 0, (size_t)c->maxB * Fmax * c->d * c->esz, c->stream));
  HIPCHK(c, hipMemsetAsync(c->mel_t, 0, (size_t)c->maxB * Fmax * c->M * c->esz, c->stream));
  return TTASR_OK;
  });
}

int ttasr_bench_kernel(ttasr_ctx* c, const char* name, int32_t B, int32_t iters, float* out_ms, double* out_bytes,
                       double* out_flops) {
  return guarded(c, [&]() -> int {
  TRY(check_ready(c, B));
  if (!name || iters < 1 || !out_ms) return fail(c, TTASR_E_INVALID, "bad arguments");
  const std::string k(name);
  hipStream_t s = c->stream;
  const double e = (double)c->esz, d = c->d, T_ = c->T, ffn = c->ffn;
  double bytes = 0, flops = 0;
  auto once = [&](void) -> int {
    if (k == "xattn") {
      static int layer_rr = 0;  // walk the layers: one layer's K/V (246 MB at B=32) would sit in the Infinity Cache
#ifdef TTASR_EXPERIMENTS
      static const bool same_layer = getenv("TTASR_BENCH_XATTN_SAME_LAYER") != nullptr;  // (that case, for comparison)
#else
      constexpr bool same_layer = false;
#endif
      const char* Kx = (const char*)c->xkv + (size_t)(same_layer ? 0 : layer_rr++ % c->cfg.dec_layers) * c->xkv_layer_elems * c->esz;
      // the instantiation the decode step launches: 16-bit engines read the query from the q GEMM's K-split partial tiles
      SlabIn sqb;
      if (c->lowp && !c->force_basic && c->slab) {
        const int ks = gemm_skinny_ksplit(B, c->d, c->d, c->ks_want[1] ? c->ks_want[1] : 4);
        if (ks > 1) { sqb.slab = c->slab; sqb.bias = c->dec[0].bqx; sqb.n = ks; sqb.stride = (int64_t)c->maxB * c->d; sqb.ld = c->d; }
      }
      TT_DISPATCH(c, launch_cross_attn_decode<T>((const T*)c->dq, (const T*)Kx, (const T*)Kx + c->xkv_which_elems, (T*)c->datt, B, c->H,
                                                 c->T, 1, s, nullptr, sqb));
      bytes = (double)B * (2.0 * T_ * d + 2.0 * d) * e; flops = (double)B * 4.0 * T_ * d;
    } else if (k == "xattn_beam5" || k == "xattn_beam5_rows") {
      // B rows = B / 5 clips x 5 hypotheses sharing their clip's cross-KV: one stream per clip ("xattn_beam5") or the
      // one-workgroup-per-row kernel ("xattn_beam5_rows")
      if (B % 5) return fail(c, TTASR_E_INVALID, "xattn_beam5 needs a multiple of 5 rows");
      static int layer_rr2 = 0;
      const char* Kx = (const char*)c->xkv + (size_t)(layer_rr2++ % c->cfg.dec_layers) * c->xkv_layer_elems * c->esz;
      float* ws = k == "xattn_beam5" ? c->xsplit_ws : nullptr;
      TT_DISPATCH(c, launch_cross_attn_decode<T>((const T*)c->dq, (const T*)Kx, (const T*)Kx + c->xkv_which_elems, (T*)c->datt, B, c->H,
                                                 c->T, 5, s, ws, SlabIn{}, c->maxB));
      bytes = (double)(B / 5) * 2.0 * T_ * d * e + (double)B * 2.0 * d * e; flops = (double)B * 4.0 * T_ * d;
    } else if (k == "enc_gemm_fc1") {
      GemmArgs g; g.A = c->h; g.W = c->enc[0].w1; g.M = B * c->T; g.N = c->ffn; g.K = c->d; g.lda = c->d; g.ldw = c->d;
      g.epi.ldc = c->ffn; g.epi.bias = c->enc[0].b1; g.epi.act = 1; g.epi.out_t = c->mid;
      sched_gemm(c, g);
      bytes = ((double)B * T_ * (d + ffn) + ffn * d) * e; flops = 2.0 * B * T_ * d * ffn;
    } else if (k == "enc_gemm_qkv") {
      GemmArgs g; g.A = c->h; g.W = c->enc[0].wqkv; g.M = B * c->T; g.N = 3 * c->d; g.K = c->d; g.lda = c->d; g.ldw = c->d;
      g.epi.ldc = 3 * c->d; g.epi.bias = c->enc[0].bqkv; g.epi.out_t = c->qkv;
      sched_gemm(c, g);
      bytes = ((double)B * T_ * 4 * d + 3 * d * d) * e; flops = 2.0 * B * T_ * d * 3 * d;
    } else if (k == "enc_gemm_out") {
      GemmArgs g; g.A = c->att; g.W = c->enc[0].wo; g.M = B * c->T; g.N = c->d; g.K = c->d; g.lda = c->d; g.ldw = c->d;
      g.epi.ldc = c->d; g.epi.bias = c->enc[0].bo;
      const bool delta = c->lowp && !c->force_basic && !c->enc_res_epilogue;  // the epilogue run_encoder uses
      if (delta) g.epi.out_t = c->h; else { g.epi.residual = c->x; g.epi.out_f32 = c->x; }
      sched_gemm(c, g);
      bytes = ((double)B * T_ * d + d * d) * e + (delta ? e : 8.0) * B * T_ * d; flops = 2.0 * B * T_ * d * d;
    } else if (k == "enc_gemm_fc2") {
      GemmArgs g; g.A = c->mid; g.W = c->enc[0].w2; g.M = B * c->T; g.N = c->d; g.K = c->ffn; g.lda = c->ffn; g.ldw = c->ffn;
      g.epi.ldc = c->d; g.epi.bias = c->enc[0].b2;
      const bool delta = c->lowp && !c->force_basic && !c->enc_res_epilogue;
      if (delta) g.epi.out_t = c->h; else { g.epi.residual = c->x; g.epi.out_f32 = c->x; }
      sched_gemm(c, g);
      bytes = ((double)B * T_ * ffn + ffn * d) * e + (delta ? e : 8.0) * B * T_ * d; flops = 2.0 * B * T_ * d * ffn;
    } else if (k == "enc_attn") {
      TT_DISPATCH(c, {
        bool flash = false;
        if constexpr (sizeof(T) == 2) {
          if (!c->force_basic && !c->no_flash) { launch_enc_attn_flash_bf16<T>((const T*)c->qkv, (T*)c->att, B, c->T, c->H, s); flash = true; }
        }
        if (!flash) launch_enc_attn_simple<T>((const T*)c->qkv, (T*)c->att, B, c->T, c->H, s);
      });
      bytes = (double)B * T_ * 4.0 * d * e; flops = 4.0 * B * T_ * T_ * d;
    } else if (k == "dec_gemm_fc1") {
      GemmArgs g; g.A = c->dh; g.W = c->dec[0].w1; g.M = B; g.N = c->ffn; g.K = c->d; g.lda = c->d; g.ldw = c->d;
      g.epi.ldc = c->ffn; g.epi.bias = c->dec[0].b1; g.epi.act = 1; g.epi.out_t = c->dmid;
      sched_dec_gemm(c, g, c->lowp ? c->dec[0].w1_sh : nullptr);
      bytes = (ffn * d + (double)B * (d + ffn)) * e; flops = 2.0 * B * d * ffn;
    } else if (k == "logits_gemm") {
      GemmArgs g; g.A = c->dh; g.W = c->emb; g.M = B; g.N = c->V; g.K = c->d; g.lda = c->d; g.ldw = c->d;
      g.epi.ldc = c->ldv; g.epi.out_f32 = c->logits;
      sched_dec_gemm(c, g, c->lowp ? c->emb_sh : nullptr);
      bytes = (double)c->V * d * e + (double)B * c->V * 4.0; flops = 2.0 * B * d * c->V;
    } else {
      return fail(c, TTASR_E_INVALID, "unknown kernel '%s'", name);
    }
    return 0;
  };
  g_kernel_sig[0] = 0; g_kernel_sig_on = true;
  const int rc_first = once();
  g_kernel_sig_on = false;
  c->bench_sig = g_kernel_sig;   // "" when the launcher of this kernel records none
  TRY(rc_first);
  HIPCHK(c, hipStreamSynchronize(s));
  hipEventRecord(c->ev[7], s);
  for (int i = 0; i < iters; ++i) TRY(once());
  hipEventRecord(c->ev[0], s);
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  float ms = 0.f;
  hipEventElapsedTime(&ms, c->ev[7], c->ev[0]);
  *out_ms = ms / iters;
  if (out_bytes) *out_bytes = bytes;
  if (out_flops) *out_flops = flops;
  return TTASR_OK;
  });
}

int ttasr_bench_kernel_signature(ttasr_ctx* c, char* buf, int32_t len) {
  return guarded(c, [&]() -> int {   // not re-entrant like every other call: bench_sig belongs to the last ttasr_bench_kernel
  if (!c || !buf || len < 1) return TTASR_E_INVALID;
  snprintf(buf, (size_t)len, "%s", c->bench_sig.c_str());
  return TTASR_OK;
  });
}

}  // extern "C"
