// libttasr host side, shared by the engine translation units (round 4: the 1 900-line engine.hip split into allocation /
// weight intake, schedules, search and the C ABI - no behaviour change): the context, the error / dispatch macros and the entry
// points the units call in each other.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ttasr.h"
#include "common.hpp"

namespace ttasr_detail {


extern thread_local std::string g_create_error;   // error text of a failed ttasr_create (engine_alloc.hip)

struct Slot {              // where one named tensor lands on the device
  void* dst = nullptr;     // T* (matrix kinds) or float* (vector kinds)
  int64_t rows = 0, cols = 0;
  int kind = 0;            // 0 matrix->T, 1 vector->f32, 2 conv [out][in][3] -> T [out][3][in], 3 f32 matrix
  float scale = 1.0f;
  bool loaded = false;
  void* sh_base = nullptr;  // bf16 mode, decoder matrices: fragment-packed copy for the skinny GEMM
  int sh_row_off = 0;
  int sh_rows_total = 0;    // rows of the whole packed matrix (a fused q/k/v matrix is packed as three parts): decides the block height
};

struct EncLayerW { float *ln1g, *ln1b, *bqkv, *bo, *ln2g, *ln2b, *b1, *b2; void *wqkv, *wo, *w1, *w2; };
struct DecLayerW {
  float *ln1g, *ln1b, *bqkv, *bo, *ln2g, *ln2b, *bqx, *bkvx, *box, *ln3g, *ln3b, *b1, *b2;
  void *wqkv, *wo, *wqx, *wkvx, *wox, *w1, *w2;
  void *wqkv_sh = nullptr, *wo_sh = nullptr, *wqx_sh = nullptr, *wox_sh = nullptr, *w1_sh = nullptr, *w2_sh = nullptr;
};

}  // namespace ttasr_detail
using namespace ttasr_detail;

struct ttasr_ctx {
  ttasr_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t cur = nullptr;      // stream the schedule helpers enqueue on
  std::string err;
  bool lowp = false;   // 16-bit storage mode (bf16 or fp16 weights / activations, f32 accumulate / LayerNorm / softmax)
  bool f16 = false;    // ... and the 16-bit format is IEEE fp16 (TTASR_COMPUTE_F16) instead of bf16
  bool finalized = false;
  bool force_basic = false;
  bool use_graph = true;
  size_t esz = 4;  // sizeof(T)
  int T = 0, F = 0, d = 0, H = 0, ffn = 0, V = 0, ldv = 0, M = 0, maxB = 0, n_samples = 0;
  int pages_per_seq = 0;
  std::vector<void*> allocs;
  struct Pool { char* base = nullptr; size_t cap = 0, used = 0; } small_pool, big_pool;  // bump arenas (see dalloc)
  size_t arena_hint = 0;   // rough device footprint of this context (bytes): picks the big-arena chunk size
  std::unordered_map<std::string, Slot> slots;

  // weights
  void *conv1_w = nullptr, *conv2_w = nullptr, *emb = nullptr, *dpos = nullptr, *emb_sh = nullptr;
  float *conv1_b = nullptr, *conv2_b = nullptr, *epos = nullptr, *elnf_g = nullptr, *elnf_b = nullptr, *dlnf_g = nullptr,
        *dlnf_b = nullptr;
  std::vector<EncLayerW> enc;
  std::vector<DecLayerW> dec;
  float* stage_f32 = nullptr;  // upload staging (destination layout, f32)
  float* stage_raw = nullptr;  // host uploads land here first (source layout)
  size_t stage_elems = 0;

  // mel constants
  float *filters = nullptr, *dcos = nullptr, *dsin = nullptr, *window = nullptr;

  // workspaces
  float* pcm_dev = nullptr; int64_t* nsamp_dev = nullptr; unsigned* clip_max = nullptr; int64_t* mel_geom = nullptr;
  float* mel = nullptr; void* mel_t = nullptr; void* c1 = nullptr;
  float* x = nullptr; void *h = nullptr, *qkv = nullptr, *att = nullptr, *mid = nullptr, *enc_out = nullptr;
  void* xkv = nullptr; int64_t xkv_layer_elems = 0, xkv_which_elems = 0;
  // option xkv_fp8 (opt-in serving mode, kernels_fp8.hip): an e4m3 copy of the cross-KV cache (same element strides, one byte per
  // value) + one f32 scale per (layer, K | V, clip, head); read by the decode step's cross-attention only
  bool xkv_fp8 = false, xkv8_valid = false; uint8_t* xkv8 = nullptr; float* xkv8_scale = nullptr;
  void* pool = nullptr; int64_t pool_layer_elems = 0; int32_t* page_table = nullptr;
  float* xsplit_ws = nullptr;  // split-frame cross-attention (small batches)
  float* dx = nullptr; void *dh = nullptr, *dqkv = nullptr, *dq = nullptr, *datt = nullptr, *dmid = nullptr; float* logits = nullptr;
  float* rows_out = nullptr;
  int kv_div = 1;            // rows per clip sharing one cross-KV (beam width); 1 for greedy
  int identity_pages = 1;    // page_table is the identity map (greedy): the self-attention kernel computes page ids
  int32_t* pairs_dev = nullptr;  // beam search: copy-on-write page pairs
  float* topk_lp = nullptr; int32_t* topk_id = nullptr; int32_t* row_state = nullptr;  // beam search scratch
#ifdef TTASR_EXPERIMENTS
  int skip_mask = 0;  // TTASR_SKIP (experiment builds only): 1 LN, 2 decode GEMMs, 4 self-attn, 8 cross-attn, 16 select
#else
  static constexpr int skip_mask = 0;  // release builds cannot drop work from the decode step
#endif
  float* slab = nullptr;      // [16][maxB][3d] f32 partial tiles of the K-split decode GEMMs (bf16 mode)
  int ks_want[4] = {0, 0, 0, 0};  // option ksplit_out / _q / _qkv / _fc2: K slices of the out-proj / q / qkv / fc2 decode GEMMs (0 = automatic, 1 = unsplit)
  int gemm_force = 0;         // option enc_gemm = 1|2|3|4 (A/B testing of the encoder GEMM kernels)
  bool gemm_persistent = true;   // option enc_gemm_persistent [1]: the persistent 256x256 GEMM where a workgroup has >= 2 tiles (round 4: encoder + cross-KV 92.6 -> 89.1 ms, bit-identical)
  bool vocab_persistent = true;  // option vocab_persistent = 0: the one-workgroup-per-32-outputs kernel for the vocabulary projection (A/B)
  bool no_flash = false;      // option flash = 0
  int prefill_ns_min = 2;     // option prefill_ns_min: shortest prompt (positions before the last) whose <|startoftranscript|> position is taken
                              // from the prefill pass.  Round 3: 2 (was 16) - the small prefill pass now runs the decode-step launch plan
                              // (K-split GEMMs), so 3 prompt positions x 32 clips cost 6.4 ms against 8.8 ms as three steps
  bool enc_res_epilogue = false;  // option enc_residual_epilogue: keep the f32 residual add in the encoder GEMM epilogues (A/B testing)
  DecState st{}; int32_t* prompt_dev = nullptr; int32_t* plen_dev = nullptr; uint8_t* mask_dev = nullptr;
  // round 6: several contexts on one GPU may share ONE copy of the device weights (ttasr_create_shared: the pipelined folder
  // path runs pass i + 1's log-mel / encoder under pass i's decode from two contexts; a second private copy would double the
  // 5 GB of large-v3).  A sharer holds `weight_owner` and counts in the owner's `sharers`; an owner that is destroyed while
  // sharers live is only marked (`destroy_pending`) and freed by the last of them.  Weights are read-only once shared.
  ttasr_ctx* weight_owner = nullptr;
  std::atomic<int> sharers{0};
  bool destroy_pending = false;
  int32_t* row_cap_dev = nullptr;   // [maxB] per-row token budgets (st.row_cap; ttasr_generate_capped), "no budget" = 0x7f7f7f7f
  int flash_qw = 2;                 // option flash_qw: query blocks of 32 per wave in the encoder's flash attention (1 = the round-5 kernel)
  int xattn_mq_slices = 0;          // option xattn_mq_slices (A/B): 0 = automatic
  int xattn_deep_items = 512;       // option xattn_deep_items: see kernels_attn.hip cross_attn_pipe_kernel (0: never stream deep; 512 = 2 per CU: measured optimum, profiles/r6_xattn_deep_sweep.jsonl)
  bool ragged_exit = true;          // option ragged_exit [1]: finished rows (st.done) leave the attention kernels of the decode step
                                    // (0: the static batch of rounds 1-5 - every row streams its cross-KV until the last one ends; A/B)
  RuleDyn* rule_dyn_dev = nullptr; RuleDyn rule_dyn_host{};  // per-window rule scalars read by select_kernel (common.hpp RuleDyn)
  int32_t* pinned_i32 = nullptr;  // host pinned scratch
  // beam search (round 6): every per-position host <-> device exchange goes through ONE pinned block - page tables, fed tokens,
  // row histories and copy-on-write pairs out, top-k candidates back - so the copies are truly asynchronous (a pageable source
  // makes hipMemcpyAsync stage and block: measured ~0.5 ms of host time per position in round 5's loop) and a searching position
  // costs ONE stream synchronisation instead of two
  char* pinned_beam = nullptr; size_t pinned_beam_bytes = 0;
  float beam_prof_ms[4]{0, 0, 0, 0};   // last beam search: host time enqueueing, waiting for the GPU, selecting candidates; positions
  int max_new_alloc = 0, max_prompt_alloc = 0;

  int B_mel = 0, B_enc = 0, B_dec = 0;
  std::atomic_flag busy = ATOMIC_FLAG_INIT;  // one call in flight per context: a second concurrent call is refused
  int xattn_nt = 1, xattn_pipe = 1, weights_nt = 1;  // options xattn_nontemporal / xattn_pipeline / weights_nontemporal (per context; copied into the launchers' thread-locals by guarded())
  bool dec_narrow = true;   // option dec_narrow_blocks: 20-row n-blocks for the decode matrices whose 32-row block count does not fill the 256 CUs evenly (fixed once weights are packed)
  bool weights_packed = false;
  bool enc_ln_defer = true; // option enc_ln_defer = 0: every encoder LayerNorm folds its delta into the f32 residual stream (two read-modify-writes per layer; A/B testing, bit-identical)
  bool dec_x_lds = true;    // option dec_x_lds = 0: decode GEMM activation fragments loaded straight from memory (the round-5 form; A/B, bit-identical)
  bool gemm_tail = true;    // option enc_gemm_tail = 0: plain 256-row tiling in the persistent encoder GEMM (A/B testing; bit-identical)
  bool xkv_grouped = true;  // option xkv_grouped = 0: one cross-KV GEMM launch per decoder layer instead of one grouped launch (A/B testing; bit-identical)
  bool multi_step = true;   // option multi_step_graph = 0: one graph replay per decode step (A/B testing)
  bool no_xsplit = false;   // option xsplit = 0: never split the cross-attention frames over workgroups (A/B testing)
  bool no_prefill = false;  // option prefill = 0: feed prompts token by token (A/B testing)
  bool prefill_tiled = false;  // option prefill_tiled: tiled encoder GEMMs in the prefill pass whatever the row count (A/B testing)
  hipEvent_t ev[8]{};
  std::string bench_sig;     // signature of the kernel the last ttasr_bench_kernel call launched (ttasr_bench_kernel_signature)
  float phase_ms[4]{0, 0, 0, 0};
  // option enc_kernel_timing: one hipEvent after every launch of run_encoder / run_cross_kv, so the NEXT ttasr_encode also
  // reports where the phase went, in situ (class sums: ttasr_encoder_kernel_ms).  Off in the timed benchmark steps.
  bool enc_timing = false;
  std::vector<hipEvent_t> enc_ev;
  std::vector<int> enc_ev_class;
  float enc_class_ms[8]{0, 0, 0, 0, 0, 0, 0, 0};

  // decode-step graphs keyed by (B, with_logits)
  struct GraphKey { int B; int mode; int variant; hipGraphExec_t exec; };
  std::vector<GraphKey> graphs;   // least recently used first
  static constexpr size_t kMaxGraphs = 32;   // per batch size up to four step graphs exist (1- / 4- / 8-step greedy, logits-only): 8 batch sizes stay resident
  RuleParams rp{};
};

namespace ttasr_detail {

int fail(ttasr_ctx* c, int code, const char* fmt, ...);

#define HIPCHK(c, call)                                                                                       \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return fail((c), TTASR_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                                    \
  } while (0)


// Device memory comes from a few large arenas, not one hipMalloc per tensor: the decode step is ~350 dependent launches
// whose first access is to a small, rarely touched buffer (LayerNorm gamma / beta, a bias, the residual rows).  With ~2500
// separate allocations every one of those sat on its own page, and after the ~10 GB a step streams (weights + cross-KV)
// each launch opened with an address-translation miss.  The small pool (< 1 MiB requests: every vector, every decode
// activation) is one 64 MiB block that stays translation- and cache-resident; matrices and KV pools come from 1 GiB+
// blocks that the driver can map with its largest page fragments.
template <typename P>
int dalloc(ttasr_ctx* c, P** p, size_t bytes, bool zero = true) {
  if (bytes == 0) bytes = 16;
  bytes = (bytes + 255) & ~(size_t)255;
  ttasr_ctx::Pool& pool = bytes < (1u << 20) ? c->small_pool : c->big_pool;
  // big chunks: 1 GiB for real models; small geometries (tests, streaming-size engines) open 64 MiB chunks instead of pinning a
  // gigabyte each - the hint is the footprint ttasr_create estimated for this context
  const size_t big_chunk = c->arena_hint >= ((size_t)1 << 30) ? (size_t)1 << 30 : (size_t)64 << 20;
  const size_t chunk = &pool == &c->small_pool ? (size_t)64 << 20 : big_chunk;
  if (bytes > chunk / 2) {
    // an oversize request gets its own allocation and leaves the active chunk (and its unused tail) in service
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(c, TTASR_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    c->allocs.push_back(q);
    if (zero) HIPCHK(c, hipMemsetAsync(q, 0, bytes, c->stream));
    *p = (P*)q;
    return 0;
  }
  if (pool.used + bytes > pool.cap) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, chunk);
    if (e != hipSuccess) return fail(c, TTASR_E_NOMEM, "hipMalloc(%zu) failed: %s", chunk, hipGetErrorString(e));
    c->allocs.push_back(q);
    pool.base = (char*)q; pool.cap = chunk; pool.used = 0;
  }
  void* q = pool.base + pool.used;
  pool.used += bytes;
  if (zero) HIPCHK(c, hipMemsetAsync(q, 0, bytes, c->stream));
  *p = (P*)q;
  return 0;
}
#define TRY(expr) do { int rc_ = (expr); if (rc_ != 0) return rc_; } while (0)
// Run CALL with T = the context's storage type (float | bf16_t | f16_t)
#define TT_DISPATCH(c_, CALL)                                   \
  do {                                                          \
    if (!(c_)->lowp) { using T = float; CALL; }                 \
    else if ((c_)->f16) { using T = f16_t; CALL; }              \
    else { using T = bf16_t; CALL; }                            \
  } while (0)

template <typename T>
GemmArgs lin_args(const void* A, const void* W, int M, int N, int K) {
  GemmArgs g; g.A = A; g.W = W; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.epi.ldc = N;
  return g;
}

// in-situ kernel classes of the encoder phase (ttasr_encoder_kernel_ms)
enum EncClass { EC_CONV = 0, EC_LN = 1, EC_QKV = 2, EC_ATTN = 3, EC_OUT = 4, EC_FC1 = 5, EC_FC2 = 6, EC_XKV = 7 };
// Batched prompt prefill: positions 0..npos-1 of n_seq sequences in ONE pass (rows [sequence][position]) instead of
// npos token-by-token steps.  Only the self-attention K/V of those positions has to survive (no logits: every one of
// these positions is followed by another forced prompt token), so the pass borrows the encoder's activation
// workspaces, which are idle once the cross-KV is built.  Sequence s attends to the cross-KV of clip s / seq_per_clip.
// GEMMs go through the encoder dispatch (M = n_seq * npos rows; 256x256 MFMA tiles once M >= 256).
// Alignment variant (ttasr_align): one sequence of clip `al->clip`; the cross-attention rows of the selected
// (layer, head) pairs are written to al->probs, and the residual stream is left in c->x for the token log-probs.
struct AlignOut {
  int clip;
  const int* sel;   // device [dec_layers][H]: index into probs or -1
  float* probs;     // device [n_sel][npos][T]
};

// ---- engine_alloc.hip: arenas, weight slots, workspaces, weight intake ----
int build_weights(ttasr_ctx* c);
int build_workspaces(ttasr_ctx* c);
int ingest_tensor(ttasr_ctx* c, const char* name, const void* src, int src_type, const int64_t* dims, int32_t ndim);

// ---- engine_sched.hip: the typed launch schedules (T = the context's storage type is dispatched inside) ----
void enc_mark(ttasr_ctx* c, int cls);
void sched_encoder(ttasr_ctx* c, int B);                       // conv stem, encoder layers, final LayerNorm, cross-KV
void sched_cross_kv(ttasr_ctx* c, int B);
void sched_prefill(ttasr_ctx* c, int n_seq, int npos, int seq_per_clip, int max_prompt, const AlignOut* al = nullptr);
int sched_prefill_no_speech(ttasr_ctx* c, int n_seq, int npos, int sot, int no_speech_tok);
void sched_gemm(ttasr_ctx* c, const GemmArgs& g);              // encoder-side GEMM dispatch (ttasr_bench_kernel)
void sched_dec_gemm(ttasr_ctx* c, const GemmArgs& g, const void* Wsh);
int prefill_positions(const ttasr_ctx* c, int min_plen, const ttasr_gen_opts* o, bool ns_from_prefill = false);
int step_graph(ttasr_ctx* c, int B, int mode, int nsteps = 1);
void drop_graphs(ttasr_ctx* c);
void drop_rule_graphs(ttasr_ctx* c);

// ---- engine_search.hip: rules, options, greedy / sampled / beam search ----
int check_ready(ttasr_ctx* c, int B);
int upload_rules(ttasr_ctx* c, const ttasr_gen_opts* o, int max_prompt);
int commit_rules(ttasr_ctx* c, const RuleParams& old);
int set_option(ttasr_ctx* c, const std::string& key, int v);
int reset_search(ttasr_ctx* c, int B);
int generate_rows(ttasr_ctx* c, int R, int rows_per_clip, const int32_t* prompt, const int32_t* prompt_len, int max_prompt,
                  const ttasr_gen_opts* o, float temperature, uint32_t seed, int32_t* out_tokens, int32_t* out_len, float* out_lp,
                  float* out_ns, const int32_t* row_cap = nullptr /*host [R] per-row token budgets, each in [1, max_new_tokens]*/);
int beam_search_impl(ttasr_ctx* c, int32_t A, int32_t beam, const int32_t* prompt, int32_t max_prompt, const int32_t* plens,
                     const int32_t* sots, const ttasr_gen_opts* o, float patience, int32_t* out_tokens, int32_t* out_len,
                     float* out_lp, float* out_ns);

}  // namespace ttasr_detail
