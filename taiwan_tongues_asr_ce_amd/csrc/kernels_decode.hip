// Decoder glue kernels: prompt-prefill embedding (the per-step embedding is created by the first LayerNorm of the
// step, kernels_misc.hip layernorm_rows_kernel), and the fused logits-processor / greedy-selection
// kernel.  The selection kernel restates CTranslate2's Whisper logits processors + greedy search
// (un-vendored) == HF generation/logits_process.py:1816 (begin-suppress), :1869 (suppress),
// :2000-2047 (timestamp rules), generation_whisper.py:1774-1812 (stack order); tie-breaking = first
// maximum, as torch.argmax.  All search state lives in device memory so a whole decode step can be
// replayed as a hipGraph with no host round trip.
#include "common.hpp"

// prompt prefill: row b = position (b % npos) of sequence (b / npos); token from the uploaded prompt table
template <typename T>
__global__ __launch_bounds__(256) void embed_prefill_kernel(const int32_t* __restrict__ prompt, int max_prompt, int rows_per_prompt,
                                                            int npos, const T* __restrict__ emb, const T* __restrict__ pos,
                                                            float* __restrict__ x, int d) {
  const int b = blockIdx.x, seq = b / npos, p = b % npos;
  const int tok = prompt[(int64_t)(seq / rows_per_prompt) * max_prompt + p];
  for (int i = threadIdx.x; i < d; i += 256) x[(int64_t)b * d + i] = to_f<T>(emb[(int64_t)tok * d + i]) + to_f<T>(pos[(int64_t)p * d + i]);
}
template <typename T>
void launch_embed_prefill(const int32_t* prompt, int max_prompt, int rows_per_prompt, int n_seq, int npos, const T* emb, const T* pos,
                          float* x, int d, hipStream_t s) {
  hipLaunchKernelGGL(embed_prefill_kernel<T>, dim3(n_seq * npos), dim3(256), 0, s, prompt, max_prompt, rows_per_prompt, npos, emb,
                     pos, x, d);
}
template void launch_embed_prefill<float>(const int32_t*, int, int, int, int, const float*, const float*, float*, int, hipStream_t);
template void launch_embed_prefill<bf16_t>(const int32_t*, int, int, int, int, const bf16_t*, const bf16_t*, float*, int, hipStream_t);
template void launch_embed_prefill<f16_t>(const int32_t*, int, int, int, int, const f16_t*, const f16_t*, float*, int, hipStream_t);

__global__ void advance_kernel(int32_t* step) { *step += 1; }
void launch_advance(int32_t* step, hipStream_t s) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, s, step); }

// ------------------------------------------------------------------------------------------------
struct RowRule {  // per-row view of the rules at this step
  int n, last_is_ts, pen_is_ts, ts_floor;  // ts_floor: timestamps < ts_floor are masked (or 0)
};

// The rule stack of one row (HF logits_process.py SuppressTokens / SuppressTokensAtBegin / WhisperTimeStampLogitsProcessor) as
// INTERVALS, uniform over the row, so that the per-element test is a dozen branch-free vector instructions:
//   token i is masked  <=>  its mask byte says so (bit 0 always, bit 1 at the first sampled position)
//                          or i is one of two singletons (eot when suppressed, <|notimestamps|> in timestamp mode)
//                          or i lies outside  [txt_lo, tb)  U  [ts_lo, ts_hi]
// Timestamp mode (p.timestamps): text tokens (i < tb) are all masked at the first position (a timestamp must open the
// segment) and below eot after an odd timestamp (last is one, the one before is not); timestamps are all masked after a pair,
// below ts_floor (non-decreasing), and above tb + max_initial at the first position.  Without timestamps every i is "text".
struct RowRanges { int n0mask, x_eot, x_nots, tb, txt_lo, ts_lo, ts_hi; };
__device__ __forceinline__ RowRanges make_ranges(const RowRule& r, const RuleParams& p) {
  RowRanges q;
  q.n0mask = r.n == 0 ? 3 : 1;
  q.x_eot = p.suppress_eot ? p.eot : -1;
  q.x_nots = p.timestamps ? p.no_timestamps : -1;
  q.tb = p.timestamps ? p.timestamp_begin : 0x7fffffff;
  q.txt_lo = !p.timestamps ? 0 : (r.n == 0 ? 0x7fffffff : ((r.last_is_ts && !r.pen_is_ts) ? p.eot : 0));
  q.ts_lo = (r.last_is_ts && r.pen_is_ts) ? 0x7fffffff : max(p.timestamp_begin, r.ts_floor);
  q.ts_hi = (r.n == 0 && p.max_initial >= 0) ? p.timestamp_begin + p.max_initial : 0x7fffffff;
  return q;
}
__device__ __forceinline__ bool masked_mk(uint32_t mk, int i, const RowRanges& q) {
  const bool in_txt = (i < q.tb) & (i >= q.txt_lo), in_ts = (i >= q.tb) & (i >= q.ts_lo) & (i <= q.ts_hi);
  return ((mk & q.n0mask) != 0) | (i == q.x_eot) | (i == q.x_nots) | !(in_txt | in_ts);
}

// A logits row lives in REGISTERS for the whole kernel: with 1024 threads every Whisper vocabulary (51 864 .. 51 866 <= 13 x 4096)
// is 13 float4 + 13 mask words per thread, all requested back to back - ONE memory round trip instead of one per loop
// iteration and per pass.  Thread t holds elements 4t..4t+3, 4(t+1024).. (chunk k = elements [4096 k, 4096 k + 4095]); the < 4
// tail elements go to the first threads.  apply_rules_regs() overwrites every masked element with -inf ONCE; the passes after it
// carry no rule logic at all (max ignores -inf, exp2(-inf) = 0), and because the text / timestamp boundary is an index, whole
// chunks are classified by a uniform branch instead of a per-element test.  ttasr_create refuses vocabularies above
// LOGIT_NIT * 4096.  (The streaming-loop form cost ~45 vector instructions per element on ONE CU per row and a dependent round
// trip per iteration: 30 us of every decode step.)
constexpr int LOGIT_NIT = 13;
struct RowRegs {
  float v[LOGIT_NIT][4];
  float tv;
};
__device__ __forceinline__ void load_row_regs(RowRegs& R, uint32_t (&m)[LOGIT_NIT], uint32_t& tm, const float* __restrict__ row,
                                              const uint8_t* __restrict__ mask, int V, int tid) {
  const int V4 = V & ~3;
#pragma unroll
  for (int k = 0; k < LOGIT_NIT; ++k) {
    const int i = max(min(tid * 4 + k * 4096, V4 - 4), 0);  // clamped, unconditional (row and mask are 16-byte aligned, ldv % 4 == 0)
    const float4 t = *(const float4*)(row + i);
    R.v[k][0] = t.x; R.v[k][1] = t.y; R.v[k][2] = t.z; R.v[k][3] = t.w;
    m[k] = *(const uint32_t*)(mask + i);
  }
  const int it = min(V4 + tid, V - 1);
  R.tv = row[it];
  tm = mask[it];
}
// f(i, value&) for every element this thread holds
template <class F> __device__ __forceinline__ void for_each_logit(RowRegs& R, int V, int tid, F&& f) {
  const int V4 = V & ~3;
#pragma unroll
  for (int k = 0; k < LOGIT_NIT; ++k) {
    const int i = tid * 4 + k * 4096;
    if (i < V4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f(i + j, R.v[k][j]);
    }
  }
  if (V4 + tid < V) f(V4 + tid, R.tv);
}
// ftxt(i, value) on elements below `tb`, fts(i, value) on the others; a chunk entirely below tb takes a uniform branch (for the
// Whisper vocabularies that is every chunk but the last: the timestamp tokens are the top 1501 ids)
template <class FT, class FS> __device__ __forceinline__ void for_each_by_range(RowRegs& R, int V, int tid, int tb, FT&& ftxt, FS&& fts) {
  const int V4 = V & ~3;
#pragma unroll
  for (int k = 0; k < LOGIT_NIT; ++k) {
    const int i = tid * 4 + k * 4096;
    if (i < V4) {
      if (k * 4096 + 4096 <= tb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ftxt(i + j, R.v[k][j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (i + j < tb) ftxt(i + j, R.v[k][j]); else fts(i + j, R.v[k][j]); }
      }
    }
  }
  if (V4 + tid < V) { if (V4 + tid < tb) ftxt(V4 + tid, R.tv); else fts(V4 + tid, R.tv); }
}
__device__ __forceinline__ void apply_rules_regs(RowRegs& R, const uint32_t (&m)[LOGIT_NIT], uint32_t tm, int V, int tid,
                                                 const RowRanges& q) {
  const int V4 = V & ~3;
#pragma unroll
  for (int k = 0; k < LOGIT_NIT; ++k) {
    const int i = tid * 4 + k * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (masked_mk((m[k] >> (8 * j)) & 0xff, i + j, q)) R.v[k][j] = -INFINITY;
  }
  if (masked_mk(tm, V4 + tid, q)) R.tv = -INFINITY;
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax am_merge(ArgMax a, ArgMax b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
// wave-wide (max value, lowest index among equals): commutative and associative, so the DPP / permlane tree of common.hpp
// gives the same answer as any other order
template <int CTRL> __device__ __forceinline__ ArgMax am_dpp(ArgMax a) {
  ArgMax b;
  b.v = dpp_f<CTRL>(a.v);
  b.i = __builtin_amdgcn_update_dpp(0, a.i, CTRL, 0xf, 0xf, true);
  return am_merge(a, b);
}
__device__ __forceinline__ ArgMax am_wave(ArgMax a) {
  a = am_dpp<DPP_XOR1>(a); a = am_dpp<DPP_XOR2>(a); a = am_dpp<DPP_HALF_MIRROR>(a); a = am_dpp<DPP_MIRROR>(a);
  {
    const unsigned v = __builtin_bit_cast(unsigned, a.v), i = (unsigned)a.i;
    const auto rv = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const auto ri = __builtin_amdgcn_permlane16_swap(i, i, false, false);
    const unsigned v0 = rv[0], v1 = rv[1], i0 = ri[0], i1 = ri[1];
    a = am_merge(ArgMax{__builtin_bit_cast(float, v0), (int)i0}, ArgMax{__builtin_bit_cast(float, v1), (int)i1});
  }
  {
    const unsigned v = __builtin_bit_cast(unsigned, a.v), i = (unsigned)a.i;
    const auto rv = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    const auto ri = __builtin_amdgcn_permlane32_swap(i, i, false, false);
    const unsigned v0 = rv[0], v1 = rv[1], i0 = ri[0], i1 = ri[1];
    a = am_merge(ArgMax{__builtin_bit_cast(float, v0), (int)i0}, ArgMax{__builtin_bit_cast(float, v1), (int)i1});
  }
  return a;
}

// Counter-based uniform for sampling: u(seed, row, position, token) in (0, 1); the CPU oracle evaluates the same
// integer hash, so sampled decodes are reproducible and testable (oracle/whisper_ref.py sample_uniform).
__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
  x = x * 747796405u + 2891336453u;
  const uint32_t w = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
  return (w >> 22u) ^ w;
}
__device__ __forceinline__ float gumbel_noise(uint32_t key, int i) {
  const uint32_t h = pcg_hash(key + (uint32_t)i);
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f);
  return -logf(-logf(u));
}

// One workgroup (1024 threads) per row.  Everything the kernel reads is addressed by the row index alone and requested in ONE
// batch (the logits row, the mask words, the row's search state); masked elements become -inf in the registers; pass 1 finds the
// (max, first index) of the text and timestamp ranges, pass 2 the sums of exp (f32); then the "timestamp mass > best text token"
// rule, the choice, its log-probability (the chosen logit IS the range maximum: no load), and the state update.
// The position counter: every workgroup reads *st.step when it starts; the one that draws the last ticket (all
// `total_rows` workgroups of the step have then read it) stores step + 1.
__device__ __forceinline__ void step_ticket(int32_t* ticket, int total_rows, int32_t* step, int step_now) {
  if (ticket && atomicAdd(ticket, 1) == total_rows - 1) { *ticket = 0; *step = step_now + 1; }
}
// SAMPLE: temperature > 0 (Gumbel-max pass); HOOK: out_rows != nullptr (test API: the processed row is written out).  The decode
// step of the benchmark is <false, false>.
template <bool SAMPLE, bool HOOK>
__global__ __launch_bounds__(1024) void select_kernel(const float* logits, DecState st, RuleParams p, float* out_rows,
                                                      int32_t* ticket, int total_rows) {
  __shared__ ArgMax s_am[2][16];
  __shared__ float s_sum[3][16];
  __shared__ float s_b[8];
  __shared__ int s_i[4];
  // every kernel argument fetched in ONE batch at entry (common.hpp sgpr_pin)
  logits = sgpr_pin_ptr(logits); out_rows = sgpr_pin_ptr(out_rows); ticket = sgpr_pin_ptr(ticket); total_rows = sgpr_pin(total_rows);
  st.cur_tok = sgpr_pin_ptr(st.cur_tok); st.step = sgpr_pin_ptr(st.step); st.n_sampled = sgpr_pin_ptr(st.n_sampled);
  st.last_tok = sgpr_pin_ptr(st.last_tok); st.pen_tok = sgpr_pin_ptr(st.pen_tok); st.last_ts = sgpr_pin_ptr(st.last_ts);
  st.done = sgpr_pin_ptr(st.done); st.n_done = sgpr_pin_ptr(st.n_done); st.sum_logprob = sgpr_pin_ptr(st.sum_logprob);
  st.no_speech = sgpr_pin_ptr(st.no_speech); st.out_tokens = sgpr_pin_ptr(st.out_tokens); st.prompt = sgpr_pin_ptr(st.prompt);
  st.prompt_len = sgpr_pin_ptr(st.prompt_len); st.mask = sgpr_pin_ptr(st.mask); st.dyn = sgpr_pin_ptr(st.dyn);
  st.row_cap = sgpr_pin_ptr(st.row_cap);
  p.V = sgpr_pin(p.V); p.ldv = sgpr_pin(p.ldv);
  p.eot = sgpr_pin(p.eot); p.no_timestamps = sgpr_pin(p.no_timestamps); p.timestamp_begin = sgpr_pin(p.timestamp_begin);
  p.no_speech = sgpr_pin(p.no_speech); p.timestamps = sgpr_pin(p.timestamps);
  p.max_initial = sgpr_pin(p.max_initial); p.suppress_eot = sgpr_pin(p.suppress_eot);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * p.ldv;
  // ---- one batch of loads ----
  RowRegs R;
  uint32_t mw[LOGIT_NIT], tmw;
  load_row_regs(R, mw, tmw, row, st.mask, p.V, tid);
  const int step = *st.step;
  const int plen_raw = *(st.prompt_len ? st.prompt_len + b : st.n_sampled + b);  // unconditional: no branch in front of the loads below
  RowRule r;
  r.n = st.n_sampled[b];
  const int last = st.last_tok[b], pen = st.pen_tok[b], lts = st.last_ts[b], done_b = st.done[b];
  const float slp = st.sum_logprob[b];
  const int cap_b = st.row_cap[b];   // this row's token budget (ttasr_generate_capped; otherwise huge)
  const RuleDyn dyn = *st.dyn;  // the per-window rule scalars (the by-value copies in `p` are those of the capture)
  __builtin_amdgcn_sched_barrier(0);  // every load above is issued before the first use below
  p.max_prompt = dyn.max_prompt; p.max_new = dyn.max_new; p.sot_index = dyn.sot_index; p.seed = dyn.seed;
  const int plen = st.prompt_len ? plen_raw : 1;

  const bool forced = st.prompt && (step + 1 < plen);
  const bool want_ns = p.no_speech >= 0 && step == p.sot_index && st.no_speech;
  if (forced && !want_ns && !HOOK) {
    // every wave of this workgroup has read *st.step before its ticket is drawn: the workgroup that draws the last ticket
    // stores step + 1, and a slower wave must not see that value and leave this branch alone
    __syncthreads();
    if (tid == 0) { st.cur_tok[b] = st.prompt[b * p.max_prompt + step + 1]; step_ticket(ticket, total_rows, st.step, step); }
    return;
  }
  const int tb = p.timestamp_begin;
  r.last_is_ts = (r.n >= 1 && last >= tb);
  r.pen_is_ts = (r.n < 2 || pen >= tb);
  r.ts_floor = (lts >= 0) ? ((r.last_is_ts && !r.pen_is_ts) ? lts : lts + 1) : 0;
  const RowRanges rq = make_ranges(r, p);

  float raw_max = -INFINITY;  // of the UNPROCESSED row: only the no-speech probability needs it (one step per sequence)
  if (want_ns) for_each_logit(R, p.V, tid, [&](int, float& v) { raw_max = fmaxf(raw_max, v); });
  apply_rules_regs(R, mw, tmw, p.V, tid, rq);
  if constexpr (HOOK) for_each_logit(R, p.V, tid, [&](int i, float& v) { out_rows[(int64_t)b * p.V + i] = v; });

  // pass 1: (max, lowest index) of the allowed text and timestamp tokens.  A thread visits its elements in increasing index
  // order, so the strict comparison keeps the first maximum; across threads am_merge breaks ties towards the lower index.
  ArgMax a_txt{-INFINITY, 0x7fffffff}, a_ts{-INFINITY, 0x7fffffff};
  for_each_by_range(R, p.V, tid, rq.tb,
                    [&](int i, float& v) { if (v > a_txt.v) { a_txt.v = v; a_txt.i = i; } },
                    [&](int i, float& v) { if (v > a_ts.v) { a_ts.v = v; a_ts.i = i; } });
  a_txt = am_wave(a_txt);
  a_ts = am_wave(a_ts);
  raw_max = wave_max(raw_max);
  if (lane == 0) { s_am[0][wave] = a_txt; s_am[1][wave] = a_ts; s_sum[2][wave] = raw_max; }
  __syncthreads();
  if (wave == 0) {
    ArgMax x = lane < 16 ? s_am[0][lane] : ArgMax{-INFINITY, 0x7fffffff};
    ArgMax y = lane < 16 ? s_am[1][lane] : ArgMax{-INFINITY, 0x7fffffff};
    float z = lane < 16 ? s_sum[2][lane] : -INFINITY;
    x = am_wave(x); y = am_wave(y); z = wave_max(z);
    if (lane == 0) { s_b[0] = x.v; s_i[0] = x.i; s_b[1] = y.v; s_i[1] = y.i; s_b[2] = z; }
  }
  __syncthreads();
  const float mx_txt = s_b[0], mx_ts = s_b[1], mx_raw = s_b[2];
  const int i_txt = s_i[0], i_ts = s_i[1];
  const float mx_all = fmaxf(mx_txt, mx_ts);
  // pass 2: sums of exp relative to mx_all (masked elements are -inf: they add exp2(-inf) = 0; everything masked: mref = 0)
  constexpr float LOG2E = 1.4426950408889634f;
  const float mref = (mx_all == -INFINITY ? 0.f : mx_all) * LOG2E;
  float sum_txt = 0.f, sum_ts = 0.f, sum_raw = 0.f;
  for_each_by_range(R, p.V, tid, rq.tb,
                    [&](int, float& v) { sum_txt += __builtin_amdgcn_exp2f(fmaf(v, LOG2E, -mref)); },
                    [&](int, float& v) { sum_ts += __builtin_amdgcn_exp2f(fmaf(v, LOG2E, -mref)); });
  if (want_ns)  // the raw row once more, from memory (the registers hold the processed values)
    for (int i = tid; i < p.V; i += 1024) sum_raw += __expf(row[i] - mx_raw);
  sum_txt = wave_sum(sum_txt); sum_ts = wave_sum(sum_ts); sum_raw = wave_sum(sum_raw);
  __syncthreads();
  if (lane == 0) { s_sum[0][wave] = sum_txt; s_sum[1][wave] = sum_ts; s_sum[2][wave] = sum_raw; }
  __syncthreads();
  // decision by thread 0: forced-timestamp rule, log-normaliser, greedy choice
  __shared__ float s_lse, s_cv;
  __shared__ int s_choice, s_live;
  if (tid == 0) {
    s_i[2] = 0; s_live = 0;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { t0 += s_sum[0][w]; t1 += s_sum[1][w]; t2 += s_sum[2][w]; }
    if (want_ns) st.no_speech[b] = __expf(row[p.no_speech] - mx_raw) / t2;
    if (forced) {
      st.cur_tok[b] = st.prompt[b * p.max_prompt + step + 1];
    } else if (done_b) {
      st.cur_tok[b] = p.eot;
    } else {
      // logsumexp(timestamps) > max(text)  <=>  log(t1) + mx_all > mx_txt   (common -lse cancels)
      const bool force_ts = p.timestamps && t1 > 0.f && (__logf(t1) + mx_all > mx_txt);
      s_i[2] = force_ts; s_live = 1;
      if (force_ts) { s_choice = i_ts; s_cv = mx_ts; s_lse = __logf(t1) + mx_all; }
      else {
        const bool pick_ts = mx_ts > mx_txt;  // ties go to the lower index, i.e. text
        s_choice = pick_ts ? i_ts : i_txt; s_cv = pick_ts ? mx_ts : mx_txt; s_lse = __logf(t0 + t1) + mx_all;
      }
    }
  }
  __syncthreads();
  if (SAMPLE && s_live) {  // pass 3: Gumbel-max sample over the allowed set
    const bool force_ts = s_i[2] != 0;
    const uint32_t key = pcg_hash(p.seed ^ pcg_hash((uint32_t)b * 0x9E3779B9u + (uint32_t)step));
    ArgMax best{-INFINITY, 0x7fffffff};
    for_each_logit(R, p.V, tid, [&](int i, float& v) {
      if (force_ts && i < tb) return;
      if (v == -INFINITY) return;
      best = am_merge(best, ArgMax{v / p.temperature + gumbel_noise(key, i), i});
    });
    best = am_wave(best);
    if (lane == 0) s_am[0][wave] = best;
    __syncthreads();
    if (wave == 0) {
      ArgMax x = lane < 16 ? s_am[0][lane] : ArgMax{-INFINITY, 0x7fffffff};
      x = am_wave(x);
      if (lane == 0) { s_choice = x.i; s_cv = row[x.i]; }
    }
    __syncthreads();
  }
  if (tid == 0 && s_live) {
    const int choice = s_choice;
    st.cur_tok[b] = choice;
    st.sum_logprob[b] = slp + (s_cv - s_lse);
    if (r.n < p.max_new) st.out_tokens[b * p.max_new + r.n] = choice;
    st.n_sampled[b] = r.n + 1;
    st.pen_tok[b] = last;
    st.last_tok[b] = choice;
    if (choice >= tb && p.timestamps) st.last_ts[b] = choice;
    // finished: EOT, the call's token budget, or this row's own (round 6).  From the NEXT step on the attention kernels skip the
    // row (done[b]: kernels_attn.hip row_done_exit) and the `done_b` branch above feeds it EOT without touching its state.
    if (choice == p.eot || r.n + 1 >= min(p.max_new, cap_b)) { st.done[b] = 1; atomicAdd(st.n_done, 1); }
  }
  if (tid == 0) step_ticket(ticket, total_rows, st.step, step);
  if constexpr (HOOK) {  // known-answer hook: the forced-timestamp branch also masks the text range
    __syncthreads();
    if (s_i[2])
      for (int i = tid; i < tb; i += 1024) out_rows[(int64_t)b * p.V + i] = -INFINITY;
  }
}

void launch_select(const float* logits, DecState st, RuleParams rp, int B, float* out_rows, hipStream_t s, int32_t* ticket,
                   int total_rows) {
  if (rp.V > LOGIT_NIT * 4096) { launch_fault("select: vocabulary %d > %d", rp.V, LOGIT_NIT * 4096); return; }
  const bool sample = rp.temperature > 0.f;
#define TTASR_SELECT(S_, H_) hipLaunchKernelGGL((select_kernel<S_, H_>), dim3(B), dim3(1024), 0, s, logits, st, rp, out_rows, ticket, total_rows)
  if (out_rows) { if (sample) TTASR_SELECT(true, true); else TTASR_SELECT(false, true); }
  else { if (sample) TTASR_SELECT(true, false); else TTASR_SELECT(false, false); }
#undef TTASR_SELECT
}

// log p(target | row) from raw logits (alignment pass: probability of each text token), one workgroup per row
__global__ __launch_bounds__(256) void token_logprob_kernel(const float* __restrict__ logits, int ldv, int V,
                                                            const int32_t* __restrict__ target, float* __restrict__ out) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * ldv;
  float m = -INFINITY;
  for (int i = tid; i < V; i += 256) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < V; i += 256) sum += __expf(row[i] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) out[b] = row[target[b]] - m - __logf((red[0] + red[1]) + (red[2] + red[3]));
}
// softmax(row)[tok] of raw logits, one workgroup per row: the no-speech probability when the <|startoftranscript|> position
// was computed by the batched prompt prefill instead of a decode step (HF logits_process.py:2050-2113: probability of
// <|nospeech|> under the unprocessed distribution at the sot position)
__global__ __launch_bounds__(256) void token_prob_kernel(const float* __restrict__ logits, int ldv, int V, int tok,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * ldv;
  float m = -INFINITY;
  for (int i = tid; i < V; i += 256) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < V; i += 256) sum += __expf(row[i] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) out[b] = __expf(row[tok] - m) / ((red[0] + red[1]) + (red[2] + red[3]));
}
void launch_token_prob(const float* logits, int ldv, int V, int tok, float* out, int rows, hipStream_t s) {
  hipLaunchKernelGGL(token_prob_kernel, dim3(rows), dim3(256), 0, s, logits, ldv, V, tok, out);
}
void launch_token_logprob(const float* logits, int ldv, int V, const int32_t* target, float* out, int rows, hipStream_t s) {
  hipLaunchKernelGGL(token_logprob_kernel, dim3(rows), dim3(256), 0, s, logits, ldv, V, target, out);
}

// ------------------------------------------------------------------------------------------------
// Beam search candidate kernel: one workgroup per row.  Applies the same rule stack as select_kernel
// (row history comes from the host, which owns the beam bookkeeping), then returns the k largest
// log-softmax values and their ids (ties: lowest id first, as torch.topk on distinct values).
// The row is read once into registers (RowRegs); the k + 2 passes run from there, k <= 8.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void beam_topk_kernel(const float* __restrict__ logits, BeamRowState st, RuleParams p, int k,
                                                         float* __restrict__ out_lp, int32_t* __restrict__ out_id,
                                                         float* __restrict__ out_ns) {
  __shared__ ArgMax s_am[2][16];
  __shared__ float s_sum[3][16];
  __shared__ float s_f[4];
  __shared__ int s_chosen[8];
  __shared__ int s_flag;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * p.ldv;
  // one batch of loads: the row, the mask words, the row's history
  RowRegs R;
  uint32_t mw[LOGIT_NIT], tmw;
  load_row_regs(R, mw, tmw, row, st.mask, p.V, tid);
  RowRule r;
  r.n = st.n_sampled[b];
  const int last = st.last_tok[b], pen = st.pen_tok[b], lts = st.last_ts[b];
  __builtin_amdgcn_sched_barrier(0);
  const int tb = p.timestamp_begin;
  r.last_is_ts = (r.n >= 1 && last >= tb);
  r.pen_is_ts = (r.n < 2 || pen >= tb);
  r.ts_floor = (lts >= 0) ? ((r.last_is_ts && !r.pen_is_ts) ? lts : lts + 1) : 0;
  const RowRanges rq = make_ranges(r, p);
  float m_raw = -INFINITY;
  if (out_ns) for_each_logit(R, p.V, tid, [&](int, float& v) { m_raw = fmaxf(m_raw, v); });
  apply_rules_regs(R, mw, tmw, p.V, tid, rq);
  // pass 1: maxima of the text / timestamp ranges (masked elements are -inf)
  float m_txt = -INFINITY, m_ts = -INFINITY;
  for_each_by_range(R, p.V, tid, rq.tb, [&](int, float& v) { m_txt = fmaxf(m_txt, v); }, [&](int, float& v) { m_ts = fmaxf(m_ts, v); });
  m_txt = wave_max(m_txt); m_ts = wave_max(m_ts); m_raw = wave_max(m_raw);
  if (lane == 0) { s_sum[0][wave] = m_txt; s_sum[1][wave] = m_ts; s_sum[2][wave] = m_raw; }
  __syncthreads();
  if (tid == 0) {
    float a = -INFINITY, c = -INFINITY, d = -INFINITY;
#pragma unroll
    for (int w = 0; w < 16; ++w) { a = fmaxf(a, s_sum[0][w]); c = fmaxf(c, s_sum[1][w]); d = fmaxf(d, s_sum[2][w]); }
    s_f[0] = a; s_f[1] = c; s_f[2] = d;
  }
  __syncthreads();
  const float mx_txt = s_f[0], mx_ts = s_f[1], mx_raw = s_f[2], mx_all = fmaxf(mx_txt, mx_ts);
  // pass 2: exp sums
  constexpr float LOG2E = 1.4426950408889634f;
  const float mref = (mx_all == -INFINITY ? 0.f : mx_all) * LOG2E;
  float sum_txt = 0.f, sum_ts = 0.f, sum_raw = 0.f;
  for_each_by_range(R, p.V, tid, rq.tb,
                    [&](int, float& v) { sum_txt += __builtin_amdgcn_exp2f(fmaf(v, LOG2E, -mref)); },
                    [&](int, float& v) { sum_ts += __builtin_amdgcn_exp2f(fmaf(v, LOG2E, -mref)); });
  if (out_ns)  // the unprocessed row once more, from memory (first position of a sequence only)
    for (int i = tid; i < p.V; i += 1024) sum_raw += __expf(row[i] - mx_raw);
  sum_txt = wave_sum(sum_txt); sum_ts = wave_sum(sum_ts); sum_raw = wave_sum(sum_raw);
  __syncthreads();
  if (lane == 0) { s_sum[0][wave] = sum_txt; s_sum[1][wave] = sum_ts; s_sum[2][wave] = sum_raw; }
  __syncthreads();
  if (tid == 0) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { t0 += s_sum[0][w]; t1 += s_sum[1][w]; t2 += s_sum[2][w]; }
    if (out_ns) out_ns[b] = (p.no_speech >= 0) ? __expf(row[p.no_speech] - mx_raw) / t2 : 0.f;
    const bool force_ts = p.timestamps && t1 > 0.f && (__logf(t1) + mx_all > mx_txt);
    s_flag = force_ts;
    s_f[3] = force_ts ? __logf(t1) + mx_all : __logf(t0 + t1) + mx_all;  // logsumexp of the allowed set
  }
  __syncthreads();
  const bool force_ts = s_flag != 0;
  const float lse = s_f[3];
  // k rounds of argmax over what is left: the forced-timestamp branch masks the text range first, and every winner is
  // overwritten with -inf by the thread that holds it
  if (force_ts) for_each_by_range(R, p.V, tid, tb, [&](int, float& v) { v = -INFINITY; }, [&](int, float&) {});
  for (int round = 0; round < k; ++round) {
    ArgMax best{-INFINITY, 0x7fffffff};
    for_each_logit(R, p.V, tid, [&](int i, float& v) { if (v > best.v) { best.v = v; best.i = i; } });
    best = am_wave(best);
    if (lane == 0) s_am[0][wave] = best;
    __syncthreads();
    if (wave == 0) {
      ArgMax x = lane < 16 ? s_am[0][lane] : ArgMax{-INFINITY, 0x7fffffff};
      x = am_wave(x);
      if (lane == 0) {
        s_chosen[round] = x.i;
        out_lp[b * k + round] = (x.i == 0x7fffffff) ? -INFINITY : x.v - lse;
        out_id[b * k + round] = (x.i == 0x7fffffff) ? -1 : x.i;
      }
    }
    __syncthreads();
    const int won = s_chosen[round];
    for_each_logit(R, p.V, tid, [&](int i, float& v) { if (i == won) v = -INFINITY; });
  }
}

void launch_beam_topk(const float* logits, BeamRowState st, RuleParams rp, int R, int k, float* out_lp, int32_t* out_id,
                      float* out_no_speech, hipStream_t s) {
  hipLaunchKernelGGL(beam_topk_kernel, dim3(R), dim3(1024), 0, s, logits, st, rp, k, out_lp, out_id, out_no_speech);
}
