/*
 * ttasr.h - C ABI of the MI355X-native Whisper inference hot path (libttasr.so).
 *
 * Drop-in boundary.  The reference has no native FFI: its hot path is reached through the Python object
 * faster_whisper.WhisperModel (constructor asr_core.py:141, api/file_asr.py:188,
 * api/stt_streaming/src/asr/faster_whisper_asr.py:107-109; transcribe() asr_core.py:159-167,
 * api/file_asr.py:457-465, faster_whisper_asr.py:170-172), which internally binds the CTranslate2 C++
 * engine (ctranslate2.models.Whisper.encode / .generate).  The entry points below are what a binding
 * for that pair of calls needs: everything from float32 PCM to token ids.  Plain pointers and sizes only;
 * no torch / Python types.  INTEGRATION.md shows the ctypes stub on the reference side.
 *
 * Conventions: every function returns 0 on success, a negative TTASR_E_* otherwise, and never aborts the
 * process; ttasr_last_error() gives the message.  A context is bound to one GPU and one HIP stream and is
 * NOT re-entrant: one call in flight per context (the reference never issues concurrent transcribe()
 * calls on one model: file_asr.py:175, streaming_asr.py:85-86); a call arriving while another runs on the same context
 * returns TTASR_E_INVALID at once and leaves the running call untouched.  Different contexts are independent.
 * No C++ exception crosses this boundary.  All device memory (weights, paged KV
 * pools, workspaces) is owned by the context from ttasr_create to ttasr_destroy.
 * Pointers named *_host are caller-owned host buffers; `pcm` may be a host or device pointer as stated
 * by `pcm_on_device`.
 */
#ifndef TTASR_H_
#define TTASR_H_

#include <stdint.h>

/* The ONLY symbols libttasr.so exports: the library is built with -fvisibility=hidden (csrc/Makefile), so none of its internal
 * C++ names (kernel launchers, device stubs, helpers) can collide with anything else loaded into the host process
 * (torch, RCCL, other extensions). */
#define TTASR_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define TTASR_OK 0
#define TTASR_E_INVALID (-1)   /* bad argument / wrong call order */
#define TTASR_E_HIP (-2)       /* HIP runtime error (message holds hipGetErrorString) */
#define TTASR_E_NOMEM (-3)
#define TTASR_E_WEIGHTS (-4)   /* unknown / missing / mis-shaped tensor */

#define TTASR_COMPUTE_F32 0    /* f32 weights+activations, exact-f32 MFMA: the 1e-3 parity mode */
#define TTASR_COMPUTE_BF16 1   /* bf16 weights+activations, f32 accumulate/LN/softmax: throughput mode */
#define TTASR_COMPUTE_F16 2    /* fp16 weights+activations, f32 accumulate/LN/softmax/residual stream: the reference's GPU regime
                                * (compute_type="float16": asr_core.py:141, api/config.py:12, faster_whisper_asr.py:95); same
                                * kernels and schedules as bf16 with the f16 MFMA forms, 3 more mantissa bits */

typedef struct ttasr_ctx ttasr_ctx;

/* Model geometry (what CTranslate2 reads from config.json / model.bin of the `models/` directory,
 * faster_whisper_asr.py:38) plus engine sizing. */
typedef struct ttasr_config {
  int32_t n_mels;       /* 80 (tiny..large-v2) or 128 (large-v3) */
  int32_t n_audio_ctx;  /* encoder positions, 1500; mel frames per window = 2 * n_audio_ctx */
  int32_t d_model;
  int32_t n_heads;      /* head_dim = d_model / n_heads must be 64 */
  int32_t ffn_dim;
  int32_t enc_layers;
  int32_t dec_layers;
  int32_t vocab;
  int32_t n_text_ctx;   /* 448 */
  int32_t compute_type; /* TTASR_COMPUTE_* */
  int32_t max_batch;    /* largest B any call will use; sizes the KV pools and workspaces */
  int32_t reserved;
} ttasr_config;

/* Decoding rules: the logits-processor stack of the reference path (CTranslate2 generate() options
 * suppress_blank / suppress_tokens / max_initial_timestamp_index; same rules as HF
 * generation/logits_process.py:1816,1869,1909-2047). */
typedef struct ttasr_gen_opts {
  int32_t max_new_tokens;
  int32_t eot;
  int32_t no_timestamps;
  int32_t timestamp_begin;
  int32_t no_speech;                   /* token id, or -1: do not compute no-speech probability */
  int32_t sot_index;                   /* prompt position of <|startoftranscript|> */
  int32_t timestamps;                  /* 1: apply the timestamp rules */
  int32_t max_initial_timestamp_index; /* -1: none */
  int32_t suppress_eot;                /* 1: fixed-length decode (benchmark mode) */
  int32_t n_suppress;
  int32_t n_begin_suppress;
  int32_t check_interval;              /* host polls "all rows finished" every this many steps (>=1) */
  const int32_t* suppress;             /* host, n_suppress ids masked at every step */
  const int32_t* begin_suppress;       /* host, n_begin_suppress ids masked at the first sampled position */
} ttasr_gen_opts;

/* ---- lifetime ---------------------------------------------------------------------------------- */
/* Geometry limits (TTASR_E_INVALID otherwise): d_model <= 1280, vocab <= 53248 (every Whisper checkpoint: <= 1280, <= 51866),
 * n_mels % 8 == 0, ffn_dim % 64 == 0. */
TTASR_API int ttasr_create(const ttasr_config* cfg, int device_id, ttasr_ctx** out_ctx);
/* A further context on the SAME GPU that shares `owner`'s device weights instead of loading its own copy (round 6): own stream,
 * own KV pools / workspaces / search state / captured graphs / kernel options, sized for max_batch rows (<= 0: the owner's) -
 * and zero bytes of weights (3.1 GB + the 1.9 GB packed decoder copies of large-v3 stay resident once).  What a host needs to keep
 * two batches in flight on one GPU (pass i + 1's log-mel / encoder under pass i's latency-bound decode chain: +27 % audio-s/s,
 * DESIGN.md 4.11) - the reference's folder loop is strictly serial (asr_core.py:151).  The owner's weights must be finalized; from
 * then on they are read-only for every context that shares them (ttasr_load_tensor* return TTASR_E_INVALID).  Contexts may be
 * destroyed in any order: an owner that is destroyed first stays alive, unusable, until its last sharer is gone. */
TTASR_API int ttasr_create_shared(ttasr_ctx* owner, int32_t max_batch, ttasr_ctx** out_ctx);
TTASR_API void ttasr_destroy(ttasr_ctx* ctx);
/* Message of the last failing call on this context (ctx == NULL: last ttasr_create failure). */
TTASR_API const char* ttasr_last_error(const ttasr_ctx* ctx);
/* Version / build string, e.g. "ttasr 0.1 gfx950". */
TTASR_API const char* ttasr_version(void);

/* ---- weights (replaces WhisperModel.__init__'s model.bin load) ---------------------------------- */
/* One tensor, float32 host data, HF state-dict name (model.encoder.conv1.weight, ...).  The engine
 * converts to its device layout (bf16 cast, QKV fusion, conv tap re-ordering, q pre-scaling by 1/8). */
TTASR_API int ttasr_load_tensor(ttasr_ctx* ctx, const char* name, const float* data_host, const int64_t* dims, int32_t ndim);
/* The same for a tensor that is already in DEVICE memory of this context's GPU (e.g. the bucket an RCCL broadcast just
 * filled: multi-GPU start-up moves every weight GPU-to-GPU over xGMI, in bf16 where the engine stores bf16, and never
 * stages it through a host): data_dev holds float32 (TTASR_DTYPE_F32), raw bfloat16 bits (TTASR_DTYPE_BF16) or IEEE half bits
 * (TTASR_DTYPE_F16) in the
 * HF layout.  Must be complete (the caller's stream synchronised) when the call is made.  The reference's
 * WhisperModel(..., device="cuda") does this copy inside CTranslate2 (asr_core.py:141). */
#define TTASR_DTYPE_F32 0
#define TTASR_DTYPE_BF16 1
#define TTASR_DTYPE_F16 2
TTASR_API int ttasr_load_tensor_device(ttasr_ctx* ctx, const char* name, const void* data_dev, int32_t dtype, const int64_t* dims,
                             int32_t ndim);
/* Checks every tensor arrived; must precede any compute call. */
TTASR_API int ttasr_finalize_weights(ttasr_ctx* ctx);

/* ---- a5: log-mel front end ----------------------------------------------------------------------- */
/* pcm: B clips, clip b at pcm + b*pcm_stride, n_samples[b] valid samples (zero-padded / trimmed to one
 * window = 2*n_audio_ctx*160 samples).  Result stays resident for ttasr_encode; if out_mel_host != NULL
 * it also receives float32 [B][n_mels][2*n_audio_ctx]. */
TTASR_API int ttasr_log_mel(ttasr_ctx* ctx, const float* pcm, int64_t pcm_stride, const int64_t* n_samples_host, int32_t B,
                  int32_t pcm_on_device, float* out_mel_host);
/* File-level form (faster-whisper computes the features of the WHOLE recording once and the 30-s window loop slices
 * them: generate_segments, called from WhisperModel.transcribe at asr_core.py:159-167): B windows, window b belonging to
 * the recording file_pcm_of_host[b][0..file_samples_of_host[b]) (the same pointer for windows of one file; different files
 * when several files advance in lock step) and starting at its frame seek_frames_host[b] (10-ms frames).  Each window's
 * frames are those of the whole-file STFT (true neighbour samples across window boundaries, reflection only at the ends
 * of the file), frames beyond the end of the recording are 0 in feature space (pad_or_trim of the feature slice), and the
 * dynamic-range floor is max - 8 of floor_max_host[b] when given (the caller passes the whole-file maximum, obtained from
 * a first pass with out_window_max_host) instead of the window's own maximum.  out_window_max_host (optional) receives
 * each window's log10-mel maximum over its valid frames.  Result resident for ttasr_encode like ttasr_log_mel. */
TTASR_API int ttasr_log_mel_windows(ttasr_ctx* ctx, const float* const* file_pcm_of_host, const int64_t* file_samples_of_host,
                          const int64_t* seek_frames_host, int32_t B, const float* floor_max_host, float* out_window_max_host,
                          float* out_mel_host);
/* Test hook: place a caller-computed mel [B][n_mels][2*n_audio_ctx] as the encoder input. */
TTASR_API int ttasr_set_mel(ttasr_ctx* ctx, const float* mel_host, int32_t B);

/* ---- a6-a8: encoder + cross-attention K/V (ctranslate2 Whisper.encode) ------------------------- */
/* Runs the conv stem, the encoder stack and the per-decoder-layer cross K/V projection for the B clips
 * whose mel is resident.  out_enc_host (optional) receives float32 [B][n_audio_ctx][d_model]. */
TTASR_API int ttasr_encode(ttasr_ctx* ctx, int32_t B, float* out_enc_host);
/* Test hooks. */
TTASR_API int ttasr_set_encoder_output(ttasr_ctx* ctx, const float* enc_host, int32_t B); /* then builds cross K/V */
TTASR_API int ttasr_get_cross_kv(ttasr_ctx* ctx, int32_t layer, int32_t which /*0 K, 1 V*/, int32_t B, float* out_host /*[B][H][T][64]*/);

/* Short-window option (opt-in, a behavioural change versus Whisper's fixed 30-s training window, the
 * same trade whisper.cpp's `audio_ctx` makes): subsequent log_mel / encode / generate calls use only the first n_ctx
 * encoder positions (= 2*n_ctx mel frames = n_ctx*320 samples), n_ctx even, 4 <= n_ctx <= cfg.n_audio_ctx; 0 restores
 * the model's window.  A 3-s utterance (n_ctx 150) then costs a tenth of the encoder flops and cross-KV bytes.
 * Drops the resident mel / encoder state. */
TTASR_API int ttasr_set_audio_ctx(ttasr_ctx* ctx, int32_t n_ctx);

/* ---- a9-a10: decoder (ctranslate2 Whisper.generate) --------------------------------------------- */
/* Greedy search.  prompt_host: [B][max_prompt] ids, prompt_len_host[b] of them valid (>=1).
 * out_tokens_host: [B][max_new_tokens] sampled ids (EOT included when emitted), out_len_host[b] count.
 * out_sum_logprob_host / out_no_speech_host: optional [B]. */
TTASR_API int ttasr_generate(ttasr_ctx* ctx, int32_t B, const int32_t* prompt_host, const int32_t* prompt_len_host,
                   int32_t max_prompt, const ttasr_gen_opts* opts, int32_t* out_tokens_host, int32_t* out_len_host,
                   float* out_sum_logprob_host, float* out_no_speech_host);
/* Greedy search with a token budget PER ROW (round 6): row b is finished after min(row_max_new_host[b], opts->max_new_tokens)
 * sampled tokens or at EOT, whichever comes first; row_max_new_host[b] in [1, opts->max_new_tokens].  Everything else as
 * ttasr_generate - in particular every row's tokens are identical, bit for bit, to the same row of a ttasr_generate call cut at its
 * budget: rows are computed independently, and a FINISHED row (here or in ttasr_generate after EOT, or a finished clip of a beam
 * search) leaves the attention kernels of the decode step - it no longer streams its 2 x n_audio_ctx x 128 B of cross-KV per
 * (layer, head), the bytes that dominate a step - while the launch grid and the captured graphs stay those of the full batch.
 * What the reference's consumers do with natural stopping (segments are consumed until the generator ends: asr_core.py:159-172)
 * on a batch whose rows stop at different lengths; synthetic weights never emit a meaningful EOT, so benchmarks and tests
 * state the lengths here.  Option "ragged_exit" = 0 restores the static batch (A/B). */
TTASR_API int ttasr_generate_capped(ttasr_ctx* ctx, int32_t B, const int32_t* prompt_host, const int32_t* prompt_len_host,
                                    int32_t max_prompt, const ttasr_gen_opts* opts, const int32_t* row_max_new_host,
                                    int32_t* out_tokens_host, int32_t* out_len_host, float* out_sum_logprob_host,
                                    float* out_no_speech_host);
/* Beam search (the reference call sites pass beam_size=5: asr_core.py:164, file_asr.py:462,
 * faster_whisper_asr.py:144).  n_audio clips x `beam` hypotheses = rows of the decode batch (<= max_batch; the bf16
 * weight-streaming GEMM carries up to 128 rows); the `beam` rows of a clip share its cross-attention K/V, and a re-index of the
 * hypotheses permutes the self-attention page tables (copy-on-write of the one partially filled page) instead
 * of copying caches.  prompt_host: [n_audio][prompt_len] (same length for every clip).  Candidate selection
 * follows Whisper's published beam search (top beam+1 per hypothesis, EOT hypotheses go to a finished pool of
 * round(beam * patience), winner = max sum_logprob / length).  Outputs as ttasr_generate (EOT stripped). */
TTASR_API int ttasr_generate_beam(ttasr_ctx* ctx, int32_t n_audio, int32_t beam, const int32_t* prompt_host, int32_t prompt_len,
                        const ttasr_gen_opts* opts, float patience, int32_t* out_tokens_host, int32_t* out_len_host,
                        float* out_sum_logprob_host, float* out_no_speech_host);
/* The same search with one prompt per clip (prompt_host [n_audio][max_prompt], prompt_len_host[a] tokens valid,
 * sot_index_host[a] = position of <|startoftranscript|> in clip a's prompt, or NULL for opts->sot_index everywhere): what
 * a caller needs to run several FILES through one engine pass when each carries its own previous-text prompt
 * (condition_on_previous_text).  The step loop is position-synchronous: clips with longer prompts are still being forced
 * while the others already search. */
TTASR_API int ttasr_generate_beam_ragged(ttasr_ctx* ctx, int32_t n_audio, int32_t beam, const int32_t* prompt_host,
                               const int32_t* prompt_len_host, const int32_t* sot_index_host, int32_t max_prompt,
                               const ttasr_gen_opts* opts, float patience, int32_t* out_tokens_host, int32_t* out_len_host,
                               float* out_sum_logprob_host, float* out_no_speech_host);
/* Temperature sampling (the fallback ladder of faster-whisper's generate_with_fallback: temperatures 0.2 ... 1.0
 * with best_of hypotheses).  n_audio clips x best_of independently sampled rows that share the clip's cross-KV;
 * tokens are drawn from softmax(processed logits / temperature) with a counter-based generator keyed by
 * (seed, row, position, token), so a run is reproducible; per clip the hypothesis with the highest
 * sum_logprob / length is returned (EOT kept when emitted, as ttasr_generate). */
TTASR_API int ttasr_generate_sample(ttasr_ctx* ctx, int32_t n_audio, int32_t best_of, const int32_t* prompt_host, int32_t prompt_len,
                          const ttasr_gen_opts* opts, float temperature, uint32_t seed, int32_t* out_tokens_host,
                          int32_t* out_len_host, float* out_sum_logprob_host, float* out_no_speech_host);
/* Step-level access for parity tests: reset the self-attention cache, then feed one token per row per
 * call; logits_host (optional) receives raw float32 [B][vocab] for the position just fed. */
TTASR_API int ttasr_decode_reset(ttasr_ctx* ctx, int32_t B);
TTASR_API int ttasr_decode_step(ttasr_ctx* ctx, const int32_t* tokens_host, int32_t B, float* logits_host);
/* Known-answer hook for the rule kernel alone: rows [n][vocab] raw logits, hist [n][hist_stride]
 * sampled-token histories padded with -1 -> processed rows (masked entries = -inf) and the selected id. */
TTASR_API int ttasr_apply_rules(ttasr_ctx* ctx, const float* rows_host, const int32_t* hist_host, int32_t hist_stride,
                      int32_t n, const ttasr_gen_opts* opts, float* out_rows_host, int32_t* out_choice_host);

/* ---- word timestamps (faster-whisper find_alignment -> ctranslate2 Whisper.align; WhisperModel.transcribe(word_timestamps=True),
 * requested at faster_whisper_asr.py:289-294) -------------------------------------------------------------------------- */
/* Teacher-forces tokens_host[0..n_tokens) (sot sequence + text tokens + eot) against the resident encoder state of
 * clip `clip` in one batched pass and returns
 *   out_weights_host  float32 [n_pairs][n_tokens][n_ctx]  softmax cross-attention rows of the (layer, head) pairs
 *                     pairs_host[2*i], pairs_host[2*i+1]   (n_ctx = the current audio window, 1500 by default)
 *   out_logprob_host  optional float32 [n_tokens - 1]       log p(tokens[i+1] | tokens[0..i]) from the raw logits.
 * Invalidates any step-level decode state (it reuses row 0's self-attention pages). */
TTASR_API int ttasr_align(ttasr_ctx* ctx, int32_t clip, const int32_t* tokens_host, int32_t n_tokens, const int32_t* pairs_host,
                int32_t n_pairs, float* out_weights_host, float* out_logprob_host);
/* Host-side dynamic time warping over a row-major cost matrix [n_rows][n_cols] (tokens x frames): the monotone path
 * of minimum total cost from (0,0) to (n_rows-1, n_cols-1); out_row / out_col need n_rows + n_cols entries.  Pure CPU
 * (no context): CTranslate2 does this step in C++ too. */
TTASR_API int ttasr_dtw(const float* cost, int32_t n_rows, int32_t n_cols, int32_t* out_row, int32_t* out_col, int32_t* out_len);

/* ---- kernel-selection overrides (tests, A/B measurements) ----------------------------------------- */
/* The release library reads NO environment variable; every deviation from the measured configuration is an explicit call.
 * Keys (value 0 / 1 unless stated; defaults in brackets): "flash" [1] MFMA flash attention in the encoder (0: the
 * one-query-per-wave f32 kernel); "prefill" [1] batched prompt prefill (0: prompts token by token); "vocab_persistent" [1] persistent vocabulary GEMM; "xsplit" [1] frame-split
 * cross-attention for small batches; "graph" [1] hipGraph replay of the decode step; "multi_step_graph" [1] runs of 8 / 4 greedy
 * steps between two host polls replay as one graph; "generic_kernels" [0] the 64x64 generic
 * GEMM / per-row kernels everywhere; "prefill_tiled" [0]; "prefill_ns_min" [2] (tokens); "enc_residual_epilogue" [0];
 * "enc_gemm" [0] = 1 | 2 | 3 | 4 forces one encoder GEMM kernel; "enc_gemm_persistent" [1] persistent 256x256 GEMM workgroups (bit-identical to the one-tile-per-workgroup form); "dec_narrow_blocks" [1] 20-row n-blocks in the packed decode matrix whose 20-row block count is a multiple of the 256 CUs (large-v3 family: fc1 = 256 workgroups instead of 160; a weight LAYOUT choice: set it before the first ttasr_load_tensor, later changes are refused); "enc_ln_defer" [1] one f32 read-modify-write of the encoder's residual stream per layer instead of two (bit-identical); "enc_gemm_tail" [1] the persistent GEMM's last partial round of workgroups re-tiled with 192- / 128-row tiles where the plan beats the plain tiling (bit-identical); "xkv_grouped" [1] the cross-KV projections of all decoder layers as ONE grouped launch of that kernel (bit-identical to one launch per layer); "ksplit_out" / "ksplit_q" / "ksplit_qkv" / "ksplit_fc2" [0 =
 * automatic] K slices of the decode GEMMs; "xattn_nontemporal" [1], "xattn_pipeline" [1] (software-pipelined cross-attention), "weights_nontemporal" [1], "dec_x_lds" [1] (the decode GEMMs stage their activation tile through LDS with coalesced loads; 0 = fragment loads straight from memory, bit-identical) (all per context
 * since round 4); "ragged_exit" [1] finished rows of a decode batch leave its attention kernels (0: static batch, every row
 * streams its cross-KV until the last one ends; live rows are bit-identical either way); "xattn_deep_items" [512] live (row, head)
 * items at or below which the decode step's cross-attention workgroups keep 8 instead of 3 rows per lane in flight (0: never;
 * bit-identical); "xattn_mq_slices" [0 = automatic] frame slices of the shared-clip cross-attention (beam rows, prompt positions;
 * 1 ... 8, A/B); "flash_qw" [2] query blocks of 32 per wave in the encoder's MFMA flash attention (1: the round-5 form with 32
 * queries per wave; bit-identical);
 * "enc_kernel_timing" [0] per-launch events in ttasr_encode (see ttasr_encoder_kernel_ms); "xkv_fp8" [0] (16-bit engines; opt-in serving
 * mode, NOT the measured configuration) keeps an OCP e4m3 copy of the cross-KV cache with one scale per (layer, K | V, clip, head),
 * built by the next ttasr_encode and read by the decode step's cross-attention (half the bytes of the dominant kernel).
 * Drops the captured decode graphs.  Unknown key or value out of range: TTASR_E_INVALID. */
TTASR_API int ttasr_set_option(ttasr_ctx* ctx, const char* key, int32_t value);

/* ---- measurement --------------------------------------------------------------------------------- */
/* hipEvent times (ms) of the last log_mel / encode (stem+layers, cross-KV) / generate calls:
 * out[0]=mel out[1]=encoder out[2]=cross_kv out[3]=decode. */
TTASR_API int ttasr_phase_ms(ttasr_ctx* ctx, float out_ms[4]);
/* Host-side split of the LAST beam search on this context (ms): out[0] enqueueing (copies, launches), out[1] waiting for the GPU
 * (the one stream synchronisation per searching position), out[2] candidate selection + page bookkeeping on the host, out[3] =
 * number of positions stepped (a count, not ms).  The search is position-synchronous with the candidate rule on the host
 * (C++ inside the library); this is what that costs. */
TTASR_API int ttasr_beam_profile(ttasr_ctx* ctx, float out_ms[4]);
/* Where the encoder phase went, IN SITU: with option "enc_kernel_timing" = 1 the next ttasr_encode records one hipEvent after
 * every launch of its schedule (not an isolated relaunch loop: each kernel runs between its real neighbours) and this call
 * returns the per-class sums of that pass in ms: out[0] conv stem, [1] LayerNorms, [2] qkv GEMMs, [3] attention, [4] out-proj
 * GEMMs, [5] fc1 GEMMs, [6] fc2 GEMMs, [7] cross-KV GEMMs.  The extra events cost a few microseconds per launch; keep the
 * option off in timed runs. */
TTASR_API int ttasr_encoder_kernel_ms(ttasr_ctx* ctx, float out_ms[8]);
/* Re-launches one named hot kernel `iters` times on the context's stream with the state left by the
 * last encode/generate (B clips) and returns its average duration measured with hipEvents, plus the
 * algorithmic bytes and flops one launch moves/does.  Names: "xattn" (decoder cross-attention),
 * "enc_gemm_qkv", "enc_gemm_out", "enc_gemm_fc1", "enc_gemm_fc2", "enc_attn", "dec_gemm_fc1", "logits_gemm". */
TTASR_API int ttasr_bench_kernel(ttasr_ctx* ctx, const char* name, int32_t B, int32_t iters, float* out_avg_ms,
                       double* out_bytes_per_launch, double* out_flops_per_launch);
/* Signature of the kernel the LAST ttasr_bench_kernel call on this context launched, in the spelling rocprofv3 prints it
 * ("kernel_name<template arguments> grid <threads>"; "" when that kernel's launcher records none: names "xattn" and "enc_gemm_*"
 * do).  bench.py compares it with the signature stored in the committed counter profiles (profiles/xattn_pmc.json, *_pmc.json)
 * and reports their numbers only when they describe the kernel this build launches. */
TTASR_API int ttasr_bench_kernel_signature(ttasr_ctx* ctx, char* buf, int32_t len);
/* Device-wide synchronisation of the context's stream. */
TTASR_API int ttasr_sync(ttasr_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* TTASR_H_ */
