// Attention kernels (head_dim = 64 everywhere in Whisper).
//   enc_attn_simple<T>     encoder self-attention, f32 VALU flash loop (parity mode / fallback)
//   self_attn_decode<T>    one new token per row against the paged self-KV cache (append + attend)
//   cross_attn_decode<T>   one query per (row, head) against the 1500-frame cross-KV: the HBM-dominant
//                          kernel of the whole decode
// Replaces CTranslate2's MultiHeadAttention layer (un-vendored; arithmetic per HF modeling_whisper.py
// :215-238, 241-356): q arrives pre-scaled by 1/8 (folded into the weights), softmax in f32.
#include "common.hpp"
#include <cstdarg>
#include <cstdio>
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// encoder attention, simple form.  Block = 4 waves = 16 queries of one (b, h); K/V tiles of 64 keys
// staged in LDS as f32.  QK: lane = key.  PV: lane = output dim.  Online softmax per query.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void enc_attn_simple_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn,
                                                               int H) {
  constexpr int QB = 16, KB = 64, HD = 64;
  __shared__ float Ks[KB][HD + 1];
  __shared__ float Vs[KB][HD];
  __shared__ float Qs[QB][HD];
  __shared__ float Ps[4][KB];
  const int d = H * HD, ld = 3 * d;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* base = qkv + (int64_t)b * Tn * ld + h * HD;
  for (int i = tid; i < QB * HD; i += 256) {
    int qi = i >> 6, c = i & 63;
    Qs[qi][c] = (q0 + qi < Tn) ? to_f<T>(base[(int64_t)(q0 + qi) * ld + c]) : 0.f;
  }
  float m_run[4], l_run[4], o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { m_run[j] = -1e30f; l_run[j] = 0.f; o[j] = 0.f; }
  for (int k0 = 0; k0 < Tn; k0 += KB) {
    __syncthreads();
    for (int i = tid; i < KB * HD; i += 256) {
      int kr = i >> 6, c = i & 63;
      bool ok = k0 + kr < Tn;
      Ks[kr][c] = ok ? to_f<T>(base[(int64_t)(k0 + kr) * ld + d + c]) : 0.f;
      Vs[kr][c] = ok ? to_f<T>(base[(int64_t)(k0 + kr) * ld + 2 * d + c]) : 0.f;
    }
    __syncthreads();
    const bool kvalid = k0 + lane < Tn;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qi = wave * 4 + j;
      float s = 0.f;
#pragma unroll 16
      for (int c = 0; c < HD; ++c) s = fmaf(Qs[qi][c], Ks[lane][c], s);
      s = kvalid ? s : -1e30f;
      float mt = wave_max(s);
      float mn = fmaxf(m_run[j], mt);
      float p = kvalid ? __expf(s - mn) : 0.f;
      float alpha = __expf(m_run[j] - mn);
      float ps = wave_sum(p);
      l_run[j] = l_run[j] * alpha + ps;
      m_run[j] = mn;
      Ps[wave][lane] = p;
      __builtin_amdgcn_wave_barrier();
      float acc = 0.f;
#pragma unroll 16
      for (int kk = 0; kk < KB; ++kk) acc = fmaf(Ps[wave][kk], Vs[kk][lane], acc);
      o[j] = o[j] * alpha + acc;
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int qi = q0 + wave * 4 + j;
    if (qi < Tn) out[((int64_t)b * Tn + qi) * d + h * HD + lane] = from_f<T>(o[j] / l_run[j]);
  }
}
template <typename T>
void launch_enc_attn_simple(const T* qkv, T* out, int B, int Tn, int H, hipStream_t s) {
  dim3 grid((Tn + 15) / 16, H, B);
  hipLaunchKernelGGL(enc_attn_simple_kernel<T>, grid, dim3(256), 0, s, qkv, out, Tn, H);
}
template void launch_enc_attn_simple<float>(const float*, float*, int, int, int, hipStream_t);
template void launch_enc_attn_simple<bf16_t>(const bf16_t*, bf16_t*, int, int, int, hipStream_t);
template void launch_enc_attn_simple<f16_t>(const f16_t*, f16_t*, int, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// helpers: one 16-byte chunk of a K/V row per lane.  VEC elements, LPR lanes per 64-element row.
// ------------------------------------------------------------------------------------------------
template <typename T> struct RowVec;
template <> struct RowVec<float> {
  static constexpr int VEC = 4;
  __device__ static void load(const float* p, float (&v)[4]) {
    float4 t = *(const float4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
template <typename T16> struct RowVec16 {
  static constexpr int VEC = 8;
  __device__ static void load(const T16* p, float (&v)[8]) {
    const uint4 t = *(const uint4*)p;
    up8<T16>(t, v);
  }
};
template <> struct RowVec<bf16_t> : RowVec16<bf16_t> {};
template <> struct RowVec<f16_t> : RowVec16<f16_t> {};

// One 16-byte chunk of a query / key / value row from the K-split partial tiles of the decode GEMM that produced it:
// v = round_T(bias + slab[0] + ... + slab[n-1]) in slab order (bit-reproducible), i.e. exactly what the unsplit GEMM's
// epilogue would have stored.  All loads are issued first (slab index clamped), n <= 4.
template <typename T>
__device__ __forceinline__ void load_row_slabs(const SlabIn& si, int64_t off, int col /*= off % si.ld, known to the caller*/,
                                               float (&v)[RowVec<T>::VEC]) {
  constexpr int VEC = RowVec<T>::VEC, NF4 = VEC / 4, MAXS = 4;
  float4 t[MAXS][NF4], bs[NF4];
#pragma unroll
  for (int c = 0; c < NF4; ++c) bs[c] = *(const float4*)(si.bias + col + 4 * c);
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const float* p = si.slab + (int64_t)min(s, si.n - 1) * si.stride + off;
#pragma unroll
    for (int c = 0; c < NF4; ++c) t[s][c] = *(const float4*)(p + 4 * c);
  }
  __builtin_amdgcn_sched_barrier(0);  // all loads issued before the first use: one round trip
#pragma unroll
  for (int c = 0; c < NF4; ++c) {
    float4 a = bs[c];
#pragma unroll
    for (int s = 0; s < MAXS; ++s)   // slab 0 unconditionally (n >= 1): its load must not be sunk behind a branch
      if (s == 0 || s < si.n) { a.x += t[s][c].x; a.y += t[s][c].y; a.z += t[s][c].z; a.w += t[s][c].w; }
    v[4 * c] = to_f<T>(from_f<T>(a.x)); v[4 * c + 1] = to_f<T>(from_f<T>(a.y));
    v[4 * c + 2] = to_f<T>(from_f<T>(a.z)); v[4 * c + 3] = to_f<T>(from_f<T>(a.w));
  }
}
template <typename T> __device__ __forceinline__ void store_row(T* p, const float (&v)[RowVec<T>::VEC]);
template <> __device__ __forceinline__ void store_row<float>(float* p, const float (&v)[4]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_row<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 o;  // v holds exactly representable bf16 values
  o.x = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
  o.y = (__float_as_uint(v[2]) >> 16) | (__float_as_uint(v[3]) & 0xffff0000u);
  o.z = (__float_as_uint(v[4]) >> 16) | (__float_as_uint(v[5]) & 0xffff0000u);
  o.w = (__float_as_uint(v[6]) >> 16) | (__float_as_uint(v[7]) & 0xffff0000u);
  *(uint4*)p = o;
}
template <> __device__ __forceinline__ void store_row<f16_t>(f16_t* p, const float (&v)[8]) {
  uint4 o;  // v holds exactly representable fp16 values: the conversion is exact
  o.x = N16<f16_t>::pk(v[0], v[1]); o.y = N16<f16_t>::pk(v[2], v[3]);
  o.z = N16<f16_t>::pk(v[4], v[5]); o.w = N16<f16_t>::pk(v[6], v[7]);
  *(uint4*)p = o;
}

// ------------------------------------------------------------------------------------------------
// decoder self-attention with paged KV cache.
// Pool layout per layer: [page][2 (K,V)][H][PAGE=16 tokens][64]; page_table[b][i] = page of tokens
// 16i..16i+15 of row b (beam search re-indexes pages instead of copying the cache).
// One workgroup (4 waves) per (b, h): appends this step's k,v at position pos = *step, then attends
// over pos+1 keys (the new key/value are taken from registers, never re-read from memory).
// ------------------------------------------------------------------------------------------------
constexpr int PAGE = 16;
thread_local char g_launch_fault[160] = "";
void launch_fault(const char* fmt, ...) {
  if (g_launch_fault[0]) return;   // the first fault of a call is the one reported
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_launch_fault, sizeof(g_launch_fault), fmt, ap);
  va_end(ap);
}
thread_local bool g_kernel_sig_on = false;
thread_local char g_kernel_sig[192] = "";
thread_local int g_xattn_variant = 3;  // bit 0: option xattn_nontemporal (nontemporal K/V loads), bit 1: option xattn_pipeline (software-pipelined form); default both
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;

// Single pass, one memory round trip for pos <= 32*UNROLL cached keys: every lane keeps an online-softmax
// state (m, l, acc[VEC]) for its row slot, K and V rows of an iteration are requested together, and the
// 8 x 4 slots are merged once at the end (flash-decoding inside the workgroup).  With identity_pages the
// page index is computed (b * pages_per_seq + t / 16) instead of loaded, which removes a dependent load.
//
// MODE 0 is the decode step described above.  MODES 1 and 2 are the two launches of the batched prompt PREFILL:
// grid row b is position (b % npos) of sequence (b / npos); launch 1 only appends every position's k,v to the
// pool, launch 2 attends causally (keys 0..pos-1 from the pool, its own from registers) without appending - two
// launches because a position reads keys that other workgroups of the first launch write.
template <typename T, int MODE, bool SLAB, bool IDENT>
__global__ __launch_bounds__(256) void self_attn_decode_kernel(const T* qkv, T* pool, const int32_t* page_table,
                                                               int pages_per_seq, int identity_pages, int row0,
                                                               const int32_t* step, T* out, const int32_t* done, int H, int npos,
                                                               SlabIn sq) {
  constexpr int VEC = RowVec<T>::VEC, LPR = 64 / VEC, RPI = 64 / LPR;  // rows per wave-instruction
  constexpr int UNROLL = 4;
  __shared__ float part[4][64];
  __shared__ float red[4][2];
  // every kernel argument fetched in ONE batch at entry (common.hpp sgpr_pin)
  qkv = sgpr_pin_ptr(qkv); pool = sgpr_pin_ptr(pool); page_table = sgpr_pin_ptr(page_table); step = sgpr_pin_ptr(step);
  out = sgpr_pin_ptr(out); done = sgpr_pin_ptr(done);
  pages_per_seq = sgpr_pin(pages_per_seq); identity_pages = sgpr_pin(identity_pages); row0 = sgpr_pin(row0); H = sgpr_pin(H);
  npos = sgpr_pin(npos);
  sq.slab = sgpr_pin_ptr(sq.slab); sq.bias = sgpr_pin_ptr(sq.bias); sq.n = sgpr_pin(sq.n); sq.stride = sgpr_pin(sq.stride);
  sq.ld = sgpr_pin(sq.ld);
  const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64, pos = MODE == 0 ? *step : b % npos;
  const int done_raw = row_done_issue(done, b, pool);   // scalar load, in flight under the vector loads below (see row_done_exit)
  const T* qp = qkv + (int64_t)b * 3 * d + h * 64;
  const int sub = lane % LPR, rin = lane / LPR;
  // global row: qkv / out are already offset to the half-batch, the KV pages are not
  const int bg = MODE == 0 ? b + row0 : b / npos;
  const int32_t* pt = page_table + bg * pages_per_seq;
  // IDENT (greedy decoding never re-indexes the table): the page id is computed, not loaded - no dependent load
  auto page_of = [&](int t) { return IDENT ? bg * pages_per_seq + t / PAGE : pt[t / PAGE]; };
  // this lane's slot: rows t = (it*4 + wave)*RPI + rin of the cached keys 0..pos-1.  The first batch of cached rows is
  // requested BEFORE q, k, v of this step are fetched (their addresses depend on the position only): one round trip less
  float kv[UNROLL][VEC], vv[UNROLL][VEC];
  auto load_kv = [&](int it0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int tc = max(min(((it0 + u) * 4 + wave) * RPI + rin, pos - 1), 0);  // clamped, unconditional loads
      const int64_t base = ((int64_t)page_of(tc) * 2 * H + h) * PAGE + (tc % PAGE);
      RowVec<T>::load(pool + base * 64 + sub * VEC, kv[u]);
      RowVec<T>::load(pool + (base + (int64_t)H * PAGE) * 64 + sub * VEC, vv[u]);
    }
  };
  if (MODE != 1) load_kv(0);
  float q[VEC], kn[VEC], vn[VEC];
  if constexpr (SLAB) {  // qkv GEMM was K-split: complete q, k, v from its partial tiles
    // q is needed by every lane: the first LPR lanes of each wave sum one 16-byte chunk each and the wave shares them by
    // shuffles (lane `sub` holds chunk `sub`); k and v of this step are only needed by the lanes that append them and score
    // the new key (wave 0's first LPR lanes).  No LDS, no barrier in front of the cached-key loop.
    float t[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { t[j] = 0.f; kn[j] = 0.f; vn[j] = 0.f; }
    const int64_t off = (int64_t)b * 3 * d + h * 64 + sub * VEC;
    if (rin == 0) {
      const int col = h * 64 + sub * VEC;
      load_row_slabs<T>(sq, off, col, t);
      if (wave == 0) { load_row_slabs<T>(sq, off + d, col + d, kn); load_row_slabs<T>(sq, off + 2 * d, col + 2 * d, vn); }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) q[j] = __shfl(t[j], sub);
  } else {
    RowVec<T>::load(qp + sub * VEC, q);
    RowVec<T>::load(qp + d + sub * VEC, kn);
    RowVec<T>::load(qp + 2 * d + sub * VEC, vn);
  }
  // Round 6: a FINISHED row of the batch (select_kernel set done[b] at an earlier step: EOT sampled, or the row's token budget
  // reached) leaves the kernel here - nothing appended, nothing attended, `out` keeps the row's last live values.  Rows are
  // computed independently of their neighbours everywhere in the decode step, so the live rows' bits do not change.
  if (MODE == 0 && done && done_raw) { if (npos < 0) red[0][0] = kv[0][0] + vv[0][0] + q[0]; return; }   // (never-taken store: row_done_exit)
  if (MODE != 2 && wave == 0 && rin == 0) {  // append this step's k, v
    const int page = page_of(pos);
    T* kdst = pool + ((((int64_t)page * 2 + 0) * H + h) * PAGE + (pos % PAGE)) * 64;
    T* vdst = pool + ((((int64_t)page * 2 + 1) * H + h) * PAGE + (pos % PAGE)) * 64;
    store_row<T>(kdst + sub * VEC, kn);
    store_row<T>(vdst + sub * VEC, vn);
  }
  if (MODE == 1) return;
  float m_run = -1e30f, l_run = 0.f, acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  const int n_it = (pos + 4 * RPI - 1) / (4 * RPI);
  for (int it0 = 0; it0 < n_it; it0 += UNROLL) {
    if (it0 > 0) load_kv(it0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int t = ((it0 + u) * 4 + wave) * RPI + rin;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) s = fmaf(q[j], kv[u][j], s);
      s = group_reduce<LPR>(s, OpSum{});
      if (t < pos) {
        const float mn = fmaxf(m_run, s);
        const float sc = __expf(m_run - mn), p = __expf(s - mn);
        l_run = l_run * sc + p;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(acc[j], sc, p * vv[u][j]);
        m_run = mn;
      }
    }
  }
  if (wave == 0 && rin == 0) {  // the new token, from registers
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s = fmaf(q[j], kn[j], s);
    s = group_reduce<LPR>(s, OpSum{});
    const float mn = fmaxf(m_run, s);
    const float sc = __expf(m_run - mn), p = __expf(s - mn);
    l_run = l_run * sc + p;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(acc[j], sc, p * vn[j]);
    m_run = mn;
  }
  // merge the slots: workgroup max, then weights exp(m - M)
  float mw = wave_max(m_run);
  if (lane == 0) red[wave][0] = mw;
  __syncthreads();
  const float M = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
  const float wgt = __expf(m_run - M);
  float lw = (sub == 0) ? l_run * wgt : 0.f;  // l is replicated over the LPR lanes of a row slot: count it once
  lw = wave_sum(lw);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    acc[j] *= wgt;
    acc[j] = stride_reduce<LPR>(acc[j], OpSum{});
  }
  if (lane == 0) red[wave][1] = lw;
  if (rin == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[wave][sub * VEC + j] = acc[j];
  }
  __syncthreads();
  if (tid < 64) {
    const float denom = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const float v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    out[(int64_t)b * d + h * 64 + tid] = from_f<T>(v / denom);
  }
}
template <typename T>
void launch_self_attn_decode(const T* qkv, T* kv_pool, const int32_t* page_table, int pages_per_seq, int64_t pool_layer_off,
                             int identity_pages, int row0, const int32_t* step, T* out, int B, int H, hipStream_t s, SlabIn sq,
                             const int32_t* done) {
  // identity_pages: greedy decoding never re-indexes the table, so the page id is computed, not loaded
#define TTASR_SA(SLAB_, IDENT_)                                                                                              \
  hipLaunchKernelGGL((self_attn_decode_kernel<T, 0, SLAB_, IDENT_>), dim3(H, B), dim3(256), 0, s, qkv, kv_pool + pool_layer_off, \
                     page_table, pages_per_seq, identity_pages, row0, step, out, done, H, 1, sq)
  if (sq.n > 0) { if (identity_pages) TTASR_SA(true, true); else TTASR_SA(true, false); }
  else { if (identity_pages) TTASR_SA(false, true); else TTASR_SA(false, false); }
#undef TTASR_SA
}
// prompt prefill: rows = n_seq * npos, row-major [sequence][position]; positions 0..npos-1 of every sequence
template <typename T>
void launch_self_attn_prefill(const T* qkv, T* kv_pool, const int32_t* page_table, int pages_per_seq, int64_t pool_layer_off,
                              int identity_pages, T* out, int n_seq, int npos, int H, hipStream_t s) {
#define TTASR_SP(MODE_, IDENT_)                                                                                                 \
  hipLaunchKernelGGL((self_attn_decode_kernel<T, MODE_, false, IDENT_>), dim3(H, n_seq * npos), dim3(256), 0, s, qkv,          \
                     kv_pool + pool_layer_off, page_table, pages_per_seq, identity_pages, 0, (const int32_t*)nullptr, out,          \
                     (const int32_t*)nullptr, H, npos, SlabIn{})
  if (identity_pages) { TTASR_SP(1, true); TTASR_SP(2, true); } else { TTASR_SP(1, false); TTASR_SP(2, false); }
#undef TTASR_SP
}
template void launch_self_attn_prefill<float>(const float*, float*, const int32_t*, int, int64_t, int, float*, int, int, int, hipStream_t);
template void launch_self_attn_prefill<bf16_t>(const bf16_t*, bf16_t*, const int32_t*, int, int64_t, int, bf16_t*, int, int, int,
                                               hipStream_t);
template void launch_self_attn_prefill<f16_t>(const f16_t*, f16_t*, const int32_t*, int, int64_t, int, f16_t*, int, int, int,
                                               hipStream_t);

// copy-on-write of partially filled KV pages after a beam re-index: pairs (src, dst) x all layers
template <typename T>
__global__ __launch_bounds__(256) void copy_pages_kernel(T* __restrict__ pool, const int32_t* __restrict__ pairs, int H,
                                                         int64_t layer_elems) {
  const int src = pairs[2 * blockIdx.x], dst = pairs[2 * blockIdx.x + 1];
  const int64_t page_elems = (int64_t)2 * H * PAGE * 64;
  const uint4* s = (const uint4*)(pool + blockIdx.y * layer_elems + src * page_elems);
  uint4* d = (uint4*)(pool + blockIdx.y * layer_elems + dst * page_elems);
  const int n = (int)(page_elems * sizeof(T) / 16);
  for (int i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}
template <typename T>
void launch_copy_pages(T* pool, const int32_t* pairs_dev, int n_pairs, int n_layers, int H, int64_t layer_elems, hipStream_t s) {
  if (n_pairs > 0) hipLaunchKernelGGL(copy_pages_kernel<T>, dim3(n_pairs, n_layers), dim3(256), 0, s, pool, pairs_dev, H, layer_elems);
}
template void launch_copy_pages<float>(float*, const int32_t*, int, int, int, int64_t, hipStream_t);
template void launch_copy_pages<bf16_t>(bf16_t*, const int32_t*, int, int, int, int64_t, hipStream_t);
template void launch_copy_pages<f16_t>(f16_t*, const int32_t*, int, int, int, int64_t, hipStream_t);
template void launch_self_attn_decode<float>(const float*, float*, const int32_t*, int, int64_t, int, int, const int32_t*, float*,
                                             int, int, hipStream_t, SlabIn, const int32_t*);
template void launch_self_attn_decode<bf16_t>(const bf16_t*, bf16_t*, const int32_t*, int, int64_t, int, int, const int32_t*,
                                              bf16_t*, int, int, hipStream_t, SlabIn, const int32_t*);
template void launch_self_attn_decode<f16_t>(const f16_t*, f16_t*, const int32_t*, int, int64_t, int, int, const int32_t*,
                                              f16_t*, int, int, hipStream_t, SlabIn, const int32_t*);

// ------------------------------------------------------------------------------------------------
// decoder cross-attention.  K, V: [B][H][Tk][64] (head-major, written by the cross-KV GEMM epilogue), so
// the 2*Tk*128 B (bf16) a workgroup needs are two contiguous streams read with 16 B per lane, 1 KiB per
// wave-instruction.  One workgroup (4 waves) per (b, h):
//   phase A  stream K: lane owns one 16-B chunk of a frame, LPR lanes reduce by shuffles -> score to LDS
//   softmax  exact two-pass in f32 over the Tk scores held in LDS
//   phase B  stream V: acc[j] += p[t] * v[t][j]; reduce over the row-slots of the wave, then over waves
// Algorithmic bytes per launch: B*H*2*Tk*64*sizeof(T) (+ q, out): 245.8 MB per clip-step-layer... see DESIGN.md.
// ------------------------------------------------------------------------------------------------
// PROBS = true is the alignment (word-timestamp) pass: heads listed in `sel` (sel[h] >= 0) also write their softmax
// row to probs[sel[h]][b][0..Tk).  The PROBS = false instantiation is the decode-step kernel.
// NWV waves per workgroup, UNROLL rows in flight per lane, NT = nontemporal K/V loads (the 7.9 GB of cross-KV a step
// streams is read exactly once per step and can never stay in a cache: variants measured in DESIGN.md).
template <typename T, bool NT> __device__ __forceinline__ void load_row(const T* p, float (&v)[RowVec<T>::VEC]) {
  if constexpr (NT && sizeof(T) == 2) {
    const u32x4_t t = __builtin_nontemporal_load((const u32x4_t*)p);
    up8<T>(make_uint4(t.x, t.y, t.z, t.w), v);
  } else {
    RowVec<T>::load(p, v);
  }
}
// QMODE: where the query comes from - 0 the T rows `q`; 1 the K-split partial tiles of the q GEMM (`sq`); 2 (round-4 experiment,
// VERDICT r3 next #1c) computed HERE from the LayerNorm output rows and this head's 64 rows of Wq behind the first K batch
// (one launch less per layer; every workgroup re-reads its head's 164 KB of Wq through L2: measured in DESIGN.md 4.11).
template <typename T, bool PROBS, int NWV, int UNROLL, bool NT, int QMODE>
__global__ __launch_bounds__(NWV * 64) void cross_attn_decode_kernel(const T* q, const T* K, const T* V, T* out, const int32_t* done,
                                                                     int H, int Tk, int kv_div, const int* sel, float* probs, SlabIn sq,
                                                                     QProj qp) {
  constexpr int VEC = RowVec<T>::VEC, LPR = 64 / VEC, RPI = 64 / LPR;
  extern __shared__ float sc[];  // [Tk] scores, then [NWV][64] partial outputs, [2 * NWV] reductions
  // every kernel argument fetched in ONE batch at entry (common.hpp sgpr_pin): the K stream starts one round trip after launch
  q = sgpr_pin_ptr(q); K = sgpr_pin_ptr(K); V = sgpr_pin_ptr(V); out = sgpr_pin_ptr(out); done = sgpr_pin_ptr(done);
  H = sgpr_pin(H); Tk = sgpr_pin(Tk); kv_div = sgpr_pin(kv_div);
  if constexpr (PROBS) { sel = sgpr_pin_ptr(sel); probs = sgpr_pin_ptr(probs); }
  sq.slab = sgpr_pin_ptr(sq.slab); sq.bias = sgpr_pin_ptr(sq.bias); sq.n = sgpr_pin(sq.n); sq.stride = sgpr_pin(sq.stride);
  sq.ld = sgpr_pin(sq.ld);
  const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64;
  const int sub = lane % LPR, rin = lane / LPR;
  float* part = sc + Tk;          // [NWV][64]
  float* red = part + NWV * 64;   // [2 * NWV]
  const int bk = kv_div == 1 ? b : b / kv_div;  // beam search: the kv_div rows of one clip share its cross-KV (never replicated)
  const int done_raw = row_done_issue(done, b, K);   // finished row of the batch: see row_done_exit below
  const T* Kp = K + ((int64_t)bk * H + h) * Tk * 64;
  const T* Vp = V + ((int64_t)bk * H + h) * Tk * 64;
  float mloc = -1e30f;
  // rows handled by this wave: t = (it*NWV + wave)*RPI + rin
  const int n_it = (Tk + NWV * RPI - 1) / (NWV * RPI);
  float kv[UNROLL][VEC];
  auto load_k = [&](int it0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * NWV + wave) * RPI + rin;
      load_row<T, NT>(Kp + (int64_t)min(t, Tk - 1) * 64 + sub * VEC, kv[u]);  // clamped, unconditional
    }
  };
  // the first K batch goes out BEFORE the query is fetched (and, when the q GEMM was K-split, summed from its partial
  // tiles): the stream starts one round trip after launch instead of two
  load_k(0);
  float qv[VEC];
  if constexpr (QMODE == 1) load_row_slabs<T>(sq, (int64_t)b * d + h * 64 + sub * VEC, h * 64 + sub * VEC, qv);  // q GEMM was K-split
#ifdef TTASR_EXPERIMENTS
  else if constexpr (QMODE == 2 && sizeof(T) == 2 && NWV == 4) {
    // q[j] = round_T(bias[64 h + j] + sum_k x[b][k] * Wq[64 h + j][k]); wave w owns rows 16 w .. 16 w + 15, a lane owns the
    // 16-byte chunks lane, lane + 64, lane + 128 of a row (d <= 1536); x chunks stay in registers, Wq rows arrive 8 at a time
    qp.x = sgpr_pin_ptr(qp.x); qp.W = sgpr_pin_ptr(qp.W); qp.bias = sgpr_pin_ptr(qp.bias);
    const int nch = d >> 3;
    const u32x4_t* xp = (const u32x4_t*)((const T*)qp.x + (int64_t)b * d);
    u32x4_t xv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const u32x4_t t = xp[min(lane + 64 * i, nch - 1)];
      xv[i] = lane + 64 * i < nch ? t : u32x4_t{0u, 0u, 0u, 0u};   // chunks past the row multiply by zero
    }
    const u32x4_t* wp = (const u32x4_t*)((const T*)qp.W + (int64_t)(h * 64 + wave * 16) * d);
    float* qs = part;   // [64] staging (the region is rewritten only after the last barrier of the kernel)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {
      u32x4_t wv[8][3];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i) wv[r][i] = wp[(int64_t)(r0 + r) * nch + min(lane + 64 * i, nch - 1)];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          s = N16<T>::dot2(wv[r][i].x, xv[i].x, s); s = N16<T>::dot2(wv[r][i].y, xv[i].y, s);
          s = N16<T>::dot2(wv[r][i].z, xv[i].z, s); s = N16<T>::dot2(wv[r][i].w, xv[i].w, s);
        }
        s = wave_sum(s);
        if (lane == 0) qs[wave * 16 + r0 + r] = s + qp.bias[h * 64 + wave * 16 + r0 + r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) qv[j] = to_f<T>(from_f<T>(qs[sub * VEC + j]));
    __syncthreads();   // qs is inside `part`, which the epilogue rewrites - and sc[] must not be confused with it
  }
#endif
  else RowVec<T>::load(q + (int64_t)b * d + h * 64 + sub * VEC, qv);
  if (!PROBS && done && done_raw) { if (Tk < 0) sc[0] = kv[0][0] + kv[UNROLL - 1][0] + qv[0]; return; }   // row_done_exit (see cross_attn_pipe_kernel)
  for (int it0 = 0; it0 < n_it; it0 += UNROLL) {
    if (it0 > 0) load_k(it0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * NWV + wave) * RPI + rin;
      float s = 0.f;
      if (t < Tk) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) s = fmaf(qv[j], kv[u][j], s);
      }
      s = group_reduce<LPR>(s, OpSum{});
      if (t < Tk) {
        if (sub == 0) sc[t] = s;
        mloc = fmaxf(mloc, s);
      }
    }
  }
  mloc = wave_max(mloc);
  if (lane == 0) red[wave] = mloc;
  __syncthreads();
  float mx = red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, red[w]);
  float lsum = 0.f;
  for (int t = tid; t < Tk; t += NWV * 64) {
    float p = __expf(sc[t] - mx);
    sc[t] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  if (lane == 0) red[NWV + wave] = lsum;
  __syncthreads();
  float denom;
  if constexpr (NWV == 4) denom = (red[4] + red[5]) + (red[6] + red[7]);
  else { denom = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) denom += red[NWV + w]; }
  if constexpr (PROBS) {
    const int si = sel[h];
    if (si >= 0) {
      float* dst = probs + ((int64_t)si * gridDim.y + b) * Tk;
      const float inv = 1.f / denom;
      for (int t = tid; t < Tk; t += NWV * 64) dst[t] = sc[t] * inv;
    }
  }
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  for (int it0 = 0; it0 < n_it; it0 += UNROLL) {
    float vv[UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * NWV + wave) * RPI + rin;
      load_row<T, NT>(Vp + (int64_t)min(t, Tk - 1) * 64 + sub * VEC, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * NWV + wave) * RPI + rin;
      if (t < Tk) {
        float p = sc[t];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(p, vv[u][j], acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    acc[j] = stride_reduce<LPR>(acc[j], OpSum{});
  }
  if (rin == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[wave * 64 + sub * VEC + j] = acc[j];
  }
  __syncthreads();
  if (tid < 64) {
    float v;
    if constexpr (NWV == 4) v = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
    else { v = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) v += part[w * 64 + tid]; }
    out[(int64_t)b * d + h * 64 + tid] = from_f<T>(v / denom);
  }
}
// Software-pipelined form of the kernel above (round 4; the decode step's kernel since then): the same row -> lane mapping, the
// same per-lane accumulation order and the same reductions - BIT-IDENTICAL output - but two register sets per lane: batch i + 1
// of the K (then V) stream is requested BEFORE batch i is scored, so a wave always has U to 2U KiB in flight instead of falling
// to zero while it computes, and the first V batch is requested before the softmax.  Measured at B = 32 (640 workgroups on 256
// CUs: 2.5 per CU), us per launch: non-pipelined 8 rows per batch 40.8; pipelined U = 8 41.5, 6 40.6, 5 40.4, 4 39.8, 3 39.5,
// 2 43.8 (tools/microbench/xattn_bench.hip, profiles/r4_xattn_pipeline.txt).  At a balanced 2.97 workgroups per CU (B = 38) every
// form runs 6.42-6.48 TB/s: the gain at B = 32 is a shorter tail of the uneven 3-vs-2 workgroup split, not a faster stream.
// 16-bit storage, 4 waves, decode step only (no PROBS).
thread_local int g_xattn_mq_slices = 0;     // option xattn_mq_slices (A/B): frame slices of the shared-clip cross-attention, 0 = automatic
thread_local int g_xattn_deep_items = 512;   // option xattn_deep_items: live (row, head) items at or below which a workgroup streams
                                             // DEEP (default 2 per CU of 256 - sweep: 448 loses 0.08 ms per step at 24 live rows, 576 costs 0.15 ms at 26-28; 0 = never) - per context, like g_xattn_variant
constexpr int XATTN_DEEP_U = 8;              // rows per lane and batch of the deep form
template <typename T, bool NT, bool QSLAB, int U>
__global__ __launch_bounds__(256) void cross_attn_pipe_kernel(const T* q, const T* K, const T* V, T* out, const int32_t* done, int H,
                                                              int Tk, int kv_div, int deep_items, SlabIn sq) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int VEC = 8, LPR = 8, RPI = 8, NWV = 4;
  extern __shared__ float sc[];  // [Tk] scores, then [NWV][64] partial outputs, [2 * NWV] reductions
  q = sgpr_pin_ptr(q); K = sgpr_pin_ptr(K); V = sgpr_pin_ptr(V); out = sgpr_pin_ptr(out); done = sgpr_pin_ptr(done);
  H = sgpr_pin(H); Tk = sgpr_pin(Tk); kv_div = sgpr_pin(kv_div); deep_items = sgpr_pin(deep_items);
  sq.slab = sgpr_pin_ptr(sq.slab); sq.bias = sgpr_pin_ptr(sq.bias); sq.n = sgpr_pin(sq.n); sq.stride = sgpr_pin(sq.stride);
  sq.ld = sgpr_pin(sq.ld);
  const int slot = blockIdx.y, nrows = gridDim.y, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64;
  const int sub = lane % LPR, rin = lane / LPR;
  float* part = sc + Tk;
  float* red = part + NWV * 64;
  // Round 6: done[r] != 0 = row r of the decode batch is FINISHED (select_kernel set it at an earlier step:
  // EOT sampled, or the row's token budget reached) and must not stream its cross-KV any more.  Workgroup (h, slot) therefore
  // serves the slot-th LIVE row, not row `slot`: the live (row, head) items occupy the FIRST workgroups of the grid, which the
  // dispatcher spreads evenly over the CUs, and the workgroups behind them exit.  (Letting workgroup (h, b) just test done[b]
  // was measured first: a kernel is as slow as its most loaded CU, and with a random half of the rows gone most CUs still
  // held three live workgroups - the cross-KV bytes fell by 44 %, the decode time by 8 %.)  While no row has finished,
  // slot == row: the first K batch and the query are requested for row `slot` BEFORE the flags are looked at, so the all-live
  // step (the benchmark) waits for nothing new; only a workgroup whose row moved requests them again.
  // Lane i holds done[i] (batches of up to 64 rows; wider ones test their own row only).  The flags arrive with the query.
  const bool remap = nrows <= 64;
  const int32_t* fp = done ? done + (remap ? min(lane, nrows - 1) : slot) : (const int32_t*)K;   // unconditional load (K: any valid address)
  const int flag = *fp;
  int b = slot;
  const T* Kp; const T* Vp;
  auto place = [&](int row) {
    const int bk = kv_div == 1 ? row : row / kv_div;
    Kp = K + ((int64_t)bk * H + h) * Tk * 64 + sub * VEC;
    Vp = V + ((int64_t)bk * H + h) * Tk * 64 + sub * VEC;
  };
  place(b);
  const int n_it = (Tk + NWV * RPI - 1) / (NWV * RPI);
  // iteration `it` covers rows it * 32 .. + 31, wave w its rows 8 w .. 8 w + 7: the order of the kernel above (bit-identical).
  // (Each wave streaming its own contiguous quarter of the frames instead measured 1.0 us slower: DESIGN.md 4.11.)
  const int trow = wave * RPI + rin;
  constexpr int TSTEP = NWV * RPI;
  const int tend = Tk;
  auto issue = [&](const T* base, int it0, auto& r) {
    constexpr int NR = (int)std::extent<std::remove_reference_t<decltype(r)>>::value;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int t = min((it0 + u) * TSTEP + trow, Tk - 1);   // clamped, unconditional
      if constexpr (NT) r[u] = __builtin_nontemporal_load((const u32x4_t*)(base + (int64_t)t * 64));
      else r[u] = *(const u32x4_t*)(base + (int64_t)t * 64);
    }
  };
  u32x4_t ra0[U];
  issue(Kp, 0, ra0);
  float qv[VEC];
  auto fetch_q = [&](int row) {
    if constexpr (QSLAB) load_row_slabs<T>(sq, (int64_t)row * d + h * 64 + sub * VEC, h * 64 + sub * VEC, qv);
    else RowVec<T>::load(q + (int64_t)row * d + h * 64 + sub * VEC, qv);
  };
  fetch_q(b);
  int n_live = nrows;
  if (done) {
    // The exits below are workgroup-uniform and sit AFTER the first K batch and the query were requested; the never-taken store
    // (Tk < 0 is unknown to the compiler) keeps a use of that batch on this side of the branch - without it the optimiser sinks
    // the loads BELOW the branch, behind the wait for the flags.  A finished item's `out` row keeps its last live values
    // (finite; select_kernel ignores a finished row's logits); no live row's arithmetic depends on a neighbour, so the live
    // rows' outputs are bit-identical to the static batch's, whichever workgroup computes them.
#define TTASR_ROW_DONE_EXIT() do { if (Tk < 0) sc[0] = __uint_as_float(ra0[0].x ^ ra0[U - 1].x) + qv[0]; return; } while (0)
    if (remap) {
      const unsigned long long live = __ballot(lane < nrows && flag == 0);
      n_live = __popcll(live);
      if (slot >= n_live) TTASR_ROW_DONE_EXIT();
      // the slot-th live row: the lane whose bit is set with exactly `slot` set bits below it
      const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
      const unsigned long long hit = __ballot(((live >> lane) & 1ull) != 0 && below == slot);
      const int row = __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1);
      if (row != slot) {   // a row before this one has finished: this workgroup serves another row - request its stream and query
        b = row;
        place(b);
        issue(Kp, 0, ra0);
        fetch_q(b);
      }
    } else if (__builtin_amdgcn_readfirstlane(flag)) {
      TTASR_ROW_DONE_EXIT();
    }
#undef TTASR_ROW_DONE_EXIT
  }
  // The stream itself, with UU rows per lane and batch (two batches in flight).  UU = U = 3 is the measured optimum when the grid
  // over-subscribes the chip (640 workgroups: 2.5 per CU, bandwidth-bound, 39 us).  With few LIVE items a workgroup is alone
  // on its CU and its 2 x 3 KiB per wave in flight make it LATENCY-bound: measured (tools/ragged_curve.py) a step with 16 of 32
  // rows live cost 2.60 ms against 2.33 ms for a batch of 16, with ONE live row 2.44 against 1.47 - one workgroup needs ~25 us
  // for its 384 KB whatever else runs.  So below 1.75 live items per CU the same loop runs DEEP (8 rows per lane and batch:
  // 16 KiB per wave in flight).  Same row -> lane mapping, same per-lane order of the frames: bit-identical either way.
  auto stream = [&](auto utag) {
    constexpr int UU = decltype(utag)::value;
    u32x4_t ra[UU], rb[UU];
    if constexpr (UU == U) {
#pragma unroll
      for (int u = 0; u < U; ++u) ra[u] = ra0[u];      // the batch requested at kernel entry
    } else {
      issue(Kp, 0, ra);                                // (re-requests the 3 rows already in flight: 3 of 3 000)
    }
    float mloc = -1e30f;
    auto score = [&](int it0, const u32x4_t (&r)[UU]) {
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int t = (it0 + u) * TSTEP + trow;
        float kf[VEC];
        up8<T>(make_uint4(r[u].x, r[u].y, r[u].z, r[u].w), kf);
        float s = 0.f;
        if (t < tend) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) s = fmaf(qv[j], kf[j], s);
        }
        s = group_reduce<LPR>(s, OpSum{});
        if (t < tend) {
          if (sub == 0) sc[t] = s;
          mloc = fmaxf(mloc, s);
        }
      }
    };
    for (int it0 = 0; it0 < n_it; it0 += 2 * UU) {
      if (it0 + UU < n_it) issue(Kp, it0 + UU, rb);
      __builtin_amdgcn_sched_barrier(0);   // the next batch is requested before this one is scored
      score(it0, ra);
      __builtin_amdgcn_sched_barrier(0);
      if (it0 + 2 * UU < n_it) issue(Kp, it0 + 2 * UU, ra);
      __builtin_amdgcn_sched_barrier(0);
      if (it0 + UU < n_it) score(it0 + UU, rb);
      __builtin_amdgcn_sched_barrier(0);
    }
    issue(Vp, 0, ra);   // V rows do not depend on the softmax: in flight under it
    __builtin_amdgcn_sched_barrier(0);
    mloc = wave_max(mloc);
    if (lane == 0) red[wave] = mloc;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int t = tid; t < Tk; t += NWV * 64) {
      float p = __expf(sc[t] - mx);
      sc[t] = p;
      lsum += p;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[NWV + wave] = lsum;
    __syncthreads();
    const float denom = (red[4] + red[5]) + (red[6] + red[7]);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    auto accum = [&](int it0, const u32x4_t (&r)[UU]) {
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int t = (it0 + u) * TSTEP + trow;
        if (t < tend) {
          float vf[VEC];
          up8<T>(make_uint4(r[u].x, r[u].y, r[u].z, r[u].w), vf);
          const float p = sc[t];
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
        }
      }
    };
    for (int it0 = 0; it0 < n_it; it0 += 2 * UU) {
      if (it0 + UU < n_it) issue(Vp, it0 + UU, rb);
      __builtin_amdgcn_sched_barrier(0);
      accum(it0, ra);
      __builtin_amdgcn_sched_barrier(0);
      if (it0 + 2 * UU < n_it) issue(Vp, it0 + 2 * UU, ra);
      __builtin_amdgcn_sched_barrier(0);
      if (it0 + UU < n_it) accum(it0 + UU, rb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = stride_reduce<LPR>(acc[j], OpSum{});
    if (rin == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) part[wave * 64 + sub * VEC + j] = acc[j];
    }
    __syncthreads();
    if (tid < 64) {
      const float v = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
      out[(int64_t)b * d + h * 64 + tid] = from_f<T>(v / denom);
    }
  };
  if (n_live * H <= deep_items) stream(std::integral_constant<int, XATTN_DEEP_U>{});
  else stream(std::integral_constant<int, U>{});
}

// Small batches (B*H workgroups < CU count: single-file transcription, beam rows of one clip, streaming): the
// frames are split over gridDim.z workgroups per (b, h) (flash-decoding).  Each streams its slice of K then V
// exactly like the kernel above and leaves {max, sum, unnormalised out[64]}; a one-wave-per-(b, h) kernel merges
// the slices.  (Merging in the last-arriving workgroup behind a ticket was tried first: the device-scope fence every
// workgroup then needs costs more than the second launch - 3.8 vs 2.35 ms per step at B = 8.)  Not used at B = 32
// (the grid already over-subscribes the chip and the single-pass kernel is the measured roofline kernel).
template <typename T>
__global__ __launch_bounds__(256) void cross_attn_split_kernel(const T* __restrict__ q, const T* __restrict__ K,
                                                               const T* __restrict__ V, int H, int Tk, int kv_div, int chunk,
                                                               float* __restrict__ ws, SlabIn sq, const int32_t* __restrict__ done) {
  constexpr int VEC = RowVec<T>::VEC, LPR = 64 / VEC, RPI = 64 / LPR;
  constexpr int UNROLL = 8;
  extern __shared__ float sc[];  // [chunk] scores, then [4][64] partial outputs, [8] reductions
  const int b = blockIdx.y, h = blockIdx.x, z = blockIdx.z, S = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64;
  const int sub = lane % LPR, rin = lane / LPR;
  float* part = sc + chunk;
  float* red = part + 4 * 64;
  if (done && sload_i32(done + b)) return;   // finished row of the batch (round 6): its slices are not computed, the merge kernel skips it too
  float qv[VEC];
  if (sq.n > 0) load_row_slabs<T>(sq, (int64_t)b * d + h * 64 + sub * VEC, h * 64 + sub * VEC, qv);
  else RowVec<T>::load(q + (int64_t)b * d + h * 64 + sub * VEC, qv);
  const int bk = kv_div == 1 ? b : b / kv_div;
  const int t0 = z * chunk, n = min(chunk, Tk - t0);  // this slice: frames t0 .. t0+n-1 (n >= 1 by construction)
  const T* Kp = K + (((int64_t)bk * H + h) * Tk + t0) * 64;
  const T* Vp = V + (((int64_t)bk * H + h) * Tk + t0) * 64;
  float mloc = -1e30f;
  const int n_it = (n + 4 * RPI - 1) / (4 * RPI);
  for (int it0 = 0; it0 < n_it; it0 += UNROLL) {
    float kv[UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * 4 + wave) * RPI + rin;
      RowVec<T>::load(Kp + (int64_t)min(t, n - 1) * 64 + sub * VEC, kv[u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * 4 + wave) * RPI + rin;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) s = fmaf(qv[j], kv[u][j], s);
      s = group_reduce<LPR>(s, OpSum{});
      if (t < n) {
        if (sub == 0) sc[t] = s;
        mloc = fmaxf(mloc, s);
      }
    }
  }
  mloc = wave_max(mloc);
  if (lane == 0) red[wave] = mloc;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.f;
  for (int t = tid; t < n; t += 256) {
    float p = __expf(sc[t] - mx);
    sc[t] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  if (lane == 0) red[4 + wave] = lsum;
  __syncthreads();
  const float lsl = (red[4] + red[5]) + (red[6] + red[7]);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  for (int it0 = 0; it0 < n_it; it0 += UNROLL) {
    float vv[UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * 4 + wave) * RPI + rin;
      RowVec<T>::load(Vp + (int64_t)min(t, n - 1) * 64 + sub * VEC, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int t = ((it0 + u) * 4 + wave) * RPI + rin;
      if (t < n) {
        float p = sc[t];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(p, vv[u][j], acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    acc[j] = stride_reduce<LPR>(acc[j], OpSum{});
  }
  if (rin == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[wave * 64 + sub * VEC + j] = acc[j];
  }
  __syncthreads();
  const int bh = b * H + h;
  float* mine = ws + ((int64_t)bh * S + z) * 66;
  if (tid < 64) mine[2 + tid] = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
  if (tid == 0) { mine[0] = mx; mine[1] = lsl; }
}

// merge of the slices: one wave per (b, h)
template <typename T>
__global__ __launch_bounds__(64) void cross_attn_merge_kernel(const float* __restrict__ ws, T* __restrict__ out, int H, int S,
                                                              const int32_t* __restrict__ done) {
  const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x;
  if (done && sload_i32(done + b)) return;   // finished row: no slices were written for it this step; `out` keeps its last live values
  const float* all = ws + ((int64_t)b * H + h) * S * 66;
  float m[8], l[8], a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // S <= 8; clamped unconditional loads, one round trip
    const int ii = min(i, S - 1);
    m[i] = all[ii * 66]; l[i] = all[ii * 66 + 1]; a[i] = all[ii * 66 + 2 + tid];
  }
  float M = -1e30f;
#pragma unroll
  for (int i = 0; i < 8; ++i) if (i < S) M = fmaxf(M, m[i]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < S) {
      const float w = __expf(m[i] - M);
      den = fmaf(l[i], w, den);
      num = fmaf(a[i], w, num);
    }
  out[((int64_t)b * H + h) * 64 + tid] = from_f<T>(num / den);
}

// Rows that share a clip's cross-KV (the `kv_div` hypotheses of a beam, or the prompt positions of a prefill pass) in ONE
// workgroup per (clip, head, frame slice): K and V are streamed ONCE for all NQ queries instead of once per row - the per-row
// kernels above re-read the same 384 KB per (clip, head) NQ times (the Infinity Cache removes the HBM traffic, not the
// CU-side ingest that bounds the kernel).  Same slice protocol as cross_attn_split_kernel: every (row, slice) leaves
// {max, sum, unnormalised out[64]} for cross_attn_merge_kernel; rows of clip a are a*NQ .. a*NQ + NQ - 1.
// A clip's kv_div rows are cut into `groups` workgroup-rows of NQ consecutive rows (the last one holds nq_last <= NQ): prompts
// longer than 8 positions re-stream the clip's K/V once per 8 positions instead of once per position.  With a single frame
// slice (gridDim.z == 1: enough (group, head) items to fill the chip) the result is normalised and stored directly (`out`),
// without the workspace and the merge launch.
template <typename T, int NQ>
__global__ __launch_bounds__(256) void cross_attn_mq_kernel(const T* __restrict__ q, const T* __restrict__ K, const T* __restrict__ V,
                                                            int H, int Tk, int chunk, float* __restrict__ ws, SlabIn sq,
                                                            int kv_div, int groups, int nq_last, T* __restrict__ out,
                                                            const int32_t* __restrict__ done) {
  constexpr int VEC = RowVec<T>::VEC, LPR = 64 / VEC, RPI = 64 / LPR;
  constexpr int PIPE_U = NQ <= 5 ? 6 : 4;   // rows per lane and batch (two batches in flight); registers: NQ x 16 + PIPE_U x 8 + ...
  extern __shared__ float sc[];  // [NQ][chunk] scores, then [4][NQ][64] partial outputs, [2][NQ][4] reductions
  const int h = blockIdx.x, z = blockIdx.z, S = gridDim.z;
  const int clip = blockIdx.y / groups, grp = blockIdx.y - clip * groups;
  const int row0 = clip * kv_div + grp * NQ;                 // first row of this group
  const int nq = grp == groups - 1 ? nq_last : NQ;           // valid rows in it (uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64;
  const int sub = lane % LPR, rin = lane / LPR;
  float* part = sc + NQ * chunk;
  float* red = part + 4 * NQ * 64;
  if (done) {   // round 6: a group whose rows are ALL finished (a finished clip of a beam search, sampled rows that all ended)
    int all = 1;   // streams nothing; a partly finished group is computed whole (its finished rows' results are ignored)
    for (int qi = 0; qi < nq; ++qi) all &= sload_i32(done + row0 + qi) != 0;
    if (all) return;
  }
  const int t0 = z * chunk, n = min(chunk, Tk - t0);  // this slice: frames t0 .. t0+n-1 (n >= 1 by construction)
  const T* Kp = K + (((int64_t)clip * H + h) * Tk + t0) * 64;
  const T* Vp = V + (((int64_t)clip * H + h) * Tk + t0) * 64;
  const int n_it = (n + 4 * RPI - 1) / (4 * RPI);
  // Round 6: the K (then V) stream is software-pipelined like cross_attn_pipe_kernel - two register sets of U rows per lane, batch
  // i + 1 requested before batch i is scored, the first V batch before the softmax.  The one-batch-at-a-time form left a
  // workgroup with nothing in flight while it scored 5-8 queries against every row: 22.4 us per layer for the 61 MB of 8 clips
  // (2.7 TB/s) plus a 5 us merge launch - 30 % of a beam-5 step (profiles/r6_beam5_8clips_nograph_kernel_stats.csv).
  constexpr int U = PIPE_U;
  typedef typename std::conditional<sizeof(T) == 2, u32x4_t, float4>::type raw_t;   // one 16-byte chunk of a row, as loaded
  auto issue = [&](const T* base, int it0, raw_t (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = ((it0 + u) * 4 + wave) * RPI + rin;
      const T* p = base + (int64_t)min(t, n - 1) * 64 + sub * VEC;   // clamped, unconditional
      if constexpr (sizeof(T) == 2) r[u] = __builtin_nontemporal_load((const u32x4_t*)p);
      else r[u] = *(const float4*)p;
    }
  };
  auto unpack = [&](const raw_t& r, float (&v)[VEC]) {
    if constexpr (sizeof(T) == 2) up8<T>(make_uint4(r.x, r.y, r.z, r.w), v);
    else { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
  };
  raw_t ra[U], rb[U];
  issue(Kp, 0, ra);  // the stream starts before the queries are fetched
  float qv[NQ][VEC];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int64_t off = (int64_t)(row0 + min(qi, nq - 1)) * d + h * 64 + sub * VEC;  // rows past nq_last: a valid row, result dropped
    if (sq.n > 0) load_row_slabs<T>(sq, off, h * 64 + sub * VEC, qv[qi]);
    else RowVec<T>::load(q + off, qv[qi]);
  }
  float mloc[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) mloc[qi] = -1e30f;
  auto score = [&](int it0, const raw_t (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = ((it0 + u) * 4 + wave) * RPI + rin;
      float kf[VEC];
      unpack(r[u], kf);
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) s = fmaf(qv[qi][j], kf[j], s);
        s = group_reduce<LPR>(s, OpSum{});
        if (t < n) {
          if (sub == 0) sc[qi * chunk + t] = s;
          mloc[qi] = fmaxf(mloc[qi], s);
        }
      }
    }
  };
  for (int it0 = 0; it0 < n_it; it0 += 2 * U) {
    if (it0 + U < n_it) issue(Kp, it0 + U, rb);
    __builtin_amdgcn_sched_barrier(0);   // the next batch is requested before this one is scored
    score(it0, ra);
    __builtin_amdgcn_sched_barrier(0);
    if (it0 + 2 * U < n_it) issue(Kp, it0 + 2 * U, ra);
    __builtin_amdgcn_sched_barrier(0);
    if (it0 + U < n_it) score(it0 + U, rb);
    __builtin_amdgcn_sched_barrier(0);
  }
  issue(Vp, 0, ra);  // V rows do not depend on the softmax: requested before it
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float m = wave_max(mloc[qi]);
    if (lane == 0) red[qi * 4 + wave] = m;
  }
  __syncthreads();
  float mx[NQ], lsum[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    mx[qi] = fmaxf(fmaxf(red[qi * 4], red[qi * 4 + 1]), fmaxf(red[qi * 4 + 2], red[qi * 4 + 3]));
    lsum[qi] = 0.f;
  }
  for (int t = tid; t < n; t += 256) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float p = __expf(sc[qi * chunk + t] - mx[qi]);
      sc[qi * chunk + t] = p;
      lsum[qi] += p;
    }
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float l = wave_sum(lsum[qi]);
    if (lane == 0) red[NQ * 4 + qi * 4 + wave] = l;
  }
  __syncthreads();
  float acc[NQ][VEC];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[qi][j] = 0.f;
  auto accum = [&](int it0, const raw_t (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = ((it0 + u) * 4 + wave) * RPI + rin;
      if (t < n) {
        float vf[VEC];
        unpack(r[u], vf);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          const float p = sc[qi * chunk + t];
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[qi][j] = fmaf(p, vf[j], acc[qi][j]);
        }
      }
    }
  };
  for (int it0 = 0; it0 < n_it; it0 += 2 * U) {
    if (it0 + U < n_it) issue(Vp, it0 + U, rb);
    __builtin_amdgcn_sched_barrier(0);
    accum(it0, ra);
    __builtin_amdgcn_sched_barrier(0);
    if (it0 + 2 * U < n_it) issue(Vp, it0 + 2 * U, ra);
    __builtin_amdgcn_sched_barrier(0);
    if (it0 + U < n_it) accum(it0 + U, rb);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[qi][j] = stride_reduce<LPR>(acc[qi][j], OpSum{});
    if (rin == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) part[(wave * NQ + qi) * 64 + sub * VEC + j] = acc[qi][j];
    }
  }
  __syncthreads();
  for (int i = tid; i < NQ * 64; i += 256) {
    const int qi = i >> 6, c = i & 63;
    if (qi >= nq) continue;
    const float acc_c = (part[(0 * NQ + qi) * 64 + c] + part[(1 * NQ + qi) * 64 + c]) + (part[(2 * NQ + qi) * 64 + c] + part[(3 * NQ + qi) * 64 + c]);
    const float lsl = (red[NQ * 4 + qi * 4] + red[NQ * 4 + qi * 4 + 1]) + (red[NQ * 4 + qi * 4 + 2] + red[NQ * 4 + qi * 4 + 3]);
    if (S == 1) {  // the only slice: finished here
      out[(int64_t)(row0 + qi) * d + h * 64 + c] = from_f<T>(acc_c / lsl);
      continue;
    }
    float* mine = ws + ((int64_t)((row0 + qi) * H + h) * S + z) * 66;
    mine[2 + c] = acc_c;
    if (c == 0) { mine[0] = mx[qi]; mine[1] = lsl; }  // every thread holds every query's maximum
  }
}

int cross_attn_splits(int B, int H, int Tk) {
  const int bh = B * H;
  if (bh >= 256) return 1;
  int S = (480 + bh - 1) / bh;
  S = S > 8 ? 8 : S;
  const int max_s = (Tk + 63) / 64;  // keep at least 64 frames per slice
  return S > max_s ? (max_s < 1 ? 1 : max_s) : S;
}

template <typename T>
void launch_cross_attn_decode(const T* q, const T* K, const T* V, T* out, int B, int H, int Tk, int kv_div, hipStream_t s,
                              float* split_ws, SlabIn sq, int ws_rows, QProj qp, const int32_t* done) {
  if (ws_rows <= 0) ws_rows = B;
  if (qp.W) split_ws = nullptr;   // in-kernel q projection (lab builds): the single-pass per-row kernel only (B * H >= 256, kv_div == 1)  // the workspace holds ws_rows rows x 8 slices x H heads x 66 floats
  // many rows per clip (a long previous-text prompt in one prefill pass): the rows are the M dimension of an MFMA flash pass over
  // the clip's frames (kernels_flash.hip, CROSS) - K and V are streamed once per (clip, head, 128 rows)
  if constexpr (sizeof(T) == 2) {
    if (split_ws && kv_div >= 32 && B % kv_div == 0 && sq.n == 0) {
      launch_cross_attn_flash_bf16<T>(q, K, V, out, B / kv_div, kv_div, H, Tk, s);
      return;
    }
  }
  // rows sharing a clip (beam hypotheses, prefill positions): one K/V stream per (clip, group of <= 8 rows) for all of them.
  // Measured at beam 5 (kernel + merge): 30 rows 18.9 us against 27.9 us for one workgroup per row (39.3 us when the rows do
  // not share); below 256 (row, head) items the frame-split per-row kernels fill the chip better (5 rows: 9.5 vs 11.8 us) and
  // keep the job.
  if (split_ws && kv_div >= 2 && B % kv_div == 0 && B * H >= 256) {  // split_ws == nullptr: the caller asked for per-row kernels
    const int A = B / kv_div;
    const int NQ = kv_div < 8 ? kv_div : 8, groups = (kv_div + NQ - 1) / NQ, nq_last = kv_div - (groups - 1) * NQ;
    // frame slices per group (option xattn_mq_slices, 0 = the automatic rule of cross_attn_splits).  ONE slice per group - no
    // workspace, no merge launch - was measured with the pipelined stream and is SLOWER at 8 clips x beam 5 (160 workgroups,
    // one per CU: 2.98 vs 2.74 ms per step): a lone workgroup scores 5 queries against every row on one wave per SIMD.
    int Sq = cross_attn_splits(A * groups, H, Tk);
    if (g_xattn_mq_slices > 0) Sq = g_xattn_mq_slices > 8 ? 8 : g_xattn_mq_slices;
    const int chunk = ((Tk + Sq - 1) / Sq + 31) / 32 * 32;
    const int S2 = (Tk + chunk - 1) / chunk;  // every slice non-empty; S2 <= 8
    const size_t lds = sizeof(float) * ((size_t)NQ * chunk + 4 * NQ * 64 + 8 * NQ);
    const bool ws_ok = S2 == 1 || (int64_t)B * S2 <= (int64_t)ws_rows * 8;  // one slice: stored directly, no workspace
    if (lds <= 64 * 1024 && ws_ok) {
      const dim3 grid(H, A * groups, S2);
#define TTASR_MQ(NQ_) \
  hipLaunchKernelGGL((cross_attn_mq_kernel<T, NQ_>), grid, dim3(256), lds, s, q, K, V, H, Tk, chunk, split_ws, sq, kv_div, groups, nq_last, out, done)
      switch (NQ) {
        case 2: TTASR_MQ(2); break; case 3: TTASR_MQ(3); break; case 4: TTASR_MQ(4); break; case 5: TTASR_MQ(5); break;
        case 6: TTASR_MQ(6); break; case 7: TTASR_MQ(7); break; default: TTASR_MQ(8); break;
      }
#undef TTASR_MQ
      if (S2 > 1) hipLaunchKernelGGL(cross_attn_merge_kernel<T>, dim3(H, B), dim3(64), 0, s, split_ws, out, H, S2, done);
      return;
    }
  }
  const int S = (split_ws && B <= ws_rows) ? cross_attn_splits(B, H, Tk) : 1;
  if (S > 1) {
    int chunk = ((Tk + S - 1) / S + 31) / 32 * 32;
    const int S2 = (Tk + chunk - 1) / chunk;  // every slice non-empty
    size_t lds = sizeof(float) * (chunk + 4 * 64 + 8);
    hipLaunchKernelGGL(cross_attn_split_kernel<T>, dim3(H, B, S2), dim3(256), lds, s, q, K, V, H, Tk, kv_div, chunk, split_ws, sq, done);
    hipLaunchKernelGGL(cross_attn_merge_kernel<T>, dim3(H, B), dim3(64), 0, s, split_ws, out, H, S2, done);
    return;
  }
  // g_xattn_variant (option xattn_nontemporal, A/B testing): 1 = nontemporal K/V loads (default), 0 = plain.  (16 rows in flight per lane and
  // 8-wave workgroups were measured slower - DESIGN.md 4.11a - and are no longer instantiated.)
  size_t lds = sizeof(float) * (Tk + 4 * 64 + 2 * 4);
#ifdef TTASR_EXPERIMENTS   /* lab builds only: the in-kernel q projection (QMODE 2), measured slower - DESIGN.md 4.11 */
#define TTASR_XA_QPROJ(NT_)                                                                                                            \
  if constexpr (sizeof(T) == 2) {                                                                                                      \
    if (qp.W) {                                                                                                                        \
      hipLaunchKernelGGL((cross_attn_decode_kernel<T, false, 4, 8, NT_, 2>), dim3(H, B), dim3(256), lds, s, q, K, V, out, done, H, Tk, \
                         kv_div, (const int*)nullptr, (float*)nullptr, sq, qp);                                                        \
      return;                                                                                                                          \
    }                                                                                                                                  \
  }
#else
#define TTASR_XA_QPROJ(NT_)
#endif
#define TTASR_XA(NT_)                                                                                                                  \
  do {                                                                                                                                 \
    TTASR_XA_QPROJ(NT_)                                                                                                                \
    if (sq.n > 0) hipLaunchKernelGGL((cross_attn_decode_kernel<T, false, 4, 8, NT_, 1>), dim3(H, B), dim3(256), lds, s, q, K, V, out, done, \
                                     H, Tk, kv_div, (const int*)nullptr, (float*)nullptr, sq, qp);                                     \
    else hipLaunchKernelGGL((cross_attn_decode_kernel<T, false, 4, 8, NT_, 0>), dim3(H, B), dim3(256), lds, s, q, K, V, out, done, H, \
                            Tk, kv_div, (const int*)nullptr, (float*)nullptr, sq, qp);                                                 \
  } while (0)
  if constexpr (sizeof(T) == 2) {
    if ((g_xattn_variant & 2) && !qp.W) {   // software-pipelined form (default since round 4), 3 rows per lane and batch
#define TTASR_XP(NT_, QS_) hipLaunchKernelGGL((cross_attn_pipe_kernel<T, NT_, QS_, 3>), dim3(H, B), dim3(256), lds, s, q, K, V, out, done, H, Tk, kv_div, g_xattn_deep_items, sq)
      const bool nt = g_xattn_variant & 1, qs = sq.n > 0;
      if (g_kernel_sig_on) snprintf(g_kernel_sig, sizeof g_kernel_sig, "cross_attn_pipe_kernel<%s, %s, %s, 3> grid %d", sig_type<T>(),
                                    nt ? "true" : "false", qs ? "true" : "false", H * B * 256);
      if (nt) { if (qs) TTASR_XP(true, true); else TTASR_XP(true, false); } else { if (qs) TTASR_XP(false, true); else TTASR_XP(false, false); }
#undef TTASR_XP
      return;
    }
  }
  if (g_kernel_sig_on) snprintf(g_kernel_sig, sizeof g_kernel_sig, "cross_attn_decode_kernel<%s, false, 4, 8, %s, %d> grid %d", sig_type<T>(),
                                (g_xattn_variant & 1) ? "true" : "false", sq.n > 0 ? 1 : 0, H * B * 256);
  if (g_xattn_variant & 1) TTASR_XA(true); else TTASR_XA(false);
#undef TTASR_XA
}
// alignment pass: rows = token positions of one sequence; heads with sel[h] >= 0 dump their attention rows
template <typename T>
void launch_cross_attn_probs(const T* q, const T* K, const T* V, T* out, int rows, int H, int Tk, const int* sel, float* probs,
                             hipStream_t s) {
  size_t lds = sizeof(float) * (Tk + 4 * 64 + 8);
  hipLaunchKernelGGL((cross_attn_decode_kernel<T, true, 4, 8, false, 0>), dim3(H, rows), dim3(256), lds, s, q, K, V, out, (const int32_t*)nullptr, H, Tk, rows, sel, probs, SlabIn{}, QProj{});
}
template void launch_cross_attn_probs<float>(const float*, const float*, const float*, float*, int, int, int, const int*, float*,
                                             hipStream_t);
template void launch_cross_attn_probs<bf16_t>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, const int*, float*,
                                              hipStream_t);
template void launch_cross_attn_probs<f16_t>(const f16_t*, const f16_t*, const f16_t*, f16_t*, int, int, int, const int*, float*,
                                              hipStream_t);
template void launch_cross_attn_decode<float>(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t, float*,
                                              SlabIn, int, QProj, const int32_t*);
template void launch_cross_attn_decode<bf16_t>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, hipStream_t,
                                               float*, SlabIn, int, QProj, const int32_t*);
template void launch_cross_attn_decode<f16_t>(const f16_t*, const f16_t*, const f16_t*, f16_t*, int, int, int, int, hipStream_t,
                                               float*, SlabIn, int, QProj, const int32_t*);
