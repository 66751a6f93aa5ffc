// Shared device helpers and kernel-launcher declarations for libttasr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits; all conversions are explicit below

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using s16x4 = __attribute__((ext_vector_type(4))) short;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN (MI355X_MICROARCH "Correctness boundaries"): gfx950's v_cvt_pk_bf16_f32, one
// instruction per PAIR of values instead of the ~8-instruction integer sequence with a NaN branch
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
  typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
  typedef float f32x2_hw __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_hw){lo, hi}, bf16x2_hw));
}
// The second 16-bit storage format: IEEE fp16 (the reference's GPU arithmetic: compute_type="float16" at asr_core.py:141,
// api/config.py:12, faster_whisper_asr.py:95).  Raw bits in a struct of its own so that templates can tell the two 16-bit
// formats apart; every kernel that handles bf16_t also handles f16_t through N16<T> below - same tiles, same schedules, the
// f16 forms of the MFMAs (same cycles as the bf16 forms), v_cvt_pk_f16_f32 / v_cvt_f32_f16 for the conversions.
struct f16_t { uint16_t bits; };
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef _Float16 h16x2_hw __attribute__((ext_vector_type(2)));

template <typename T> struct N16;   // number format of a 16-bit storage type: conversions + MFMA forms
template <> struct N16<bf16_t> {
  static __device__ __forceinline__ float up(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
  static __device__ __forceinline__ void up2(uint32_t w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ uint32_t pk(float lo, float hi) { return f2bf_pk(lo, hi); }
  static __device__ __forceinline__ uint16_t down(float f) { return f2bf(f); }
  static __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  // c + a.lo * b.lo + a.hi * b.hi on one dword of two stored values each (v_dot2c_f32_bf16: f32 accumulate)
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, a), __builtin_bit_cast(bf16x2_hw, b), c, false);
  }
};
template <> struct N16<f16_t> {
  static __device__ __forceinline__ float up(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
  static __device__ __forceinline__ void up2(uint32_t w, float& lo, float& hi) {
    const h16x2_hw v = __builtin_bit_cast(h16x2_hw, w);
    lo = (float)v[0]; hi = (float)v[1];
  }
  // Saturating: |v| > 65504 would become inf and turn into NaN in the next LayerNorm / softmax; HF clamps its
  // fp16 hidden states for the same reason (modeling_whisper.py:403-407).  ONE v_med3_f32 per value (the compare / max / min /
  // select form that also kept a NaN a NaN cost 3.6 ms per encoder pass at large-v3: 128 values per lane and GEMM tile); med3
  // maps a NaN input to -65504, so a NaN can no longer be used to spot an upstream fault in the fp16 mode.
  // Lab builds (make EXTRA=-DTTASR_EXPERIMENTS) keep a NaN a NaN (one more v_cmp + v_cndmask per value), so that an upstream
  // numerical fault still surfaces as NaN logits when the fp16 mode is being debugged; the release build does not.
#ifdef TTASR_EXPERIMENTS
  static __device__ __forceinline__ float sat(float v) { return v != v ? v : __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
#else
  static __device__ __forceinline__ float sat(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
#endif
  static __device__ __forceinline__ uint32_t pk(float lo, float hi) {   // round-to-nearest-even, one v_cvt_pk_f16_f32
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_hw){sat(lo), sat(hi)}, h16x2_hw));
  }
  static __device__ __forceinline__ uint16_t down(float f) { return __builtin_bit_cast(uint16_t, (_Float16)sat(f)); }
  static __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {   // v_dot2c_f32_f16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2_hw, a), __builtin_bit_cast(h16x2_hw, b), c, false);
  }
};
// 8 stored values (one 16-byte chunk) -> f32
template <typename T> __device__ __forceinline__ void up8(const uint4& t, float (&v)[8]) {
  N16<T>::up2(t.x, v[0], v[1]); N16<T>::up2(t.y, v[2], v[3]); N16<T>::up2(t.z, v[4], v[5]); N16<T>::up2(t.w, v[6], v[7]);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <> __device__ __forceinline__ float to_f<f16_t>(f16_t v) { return N16<f16_t>::up(v.bits); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ f16_t from_f<f16_t>(float v) { return f16_t{N16<f16_t>::down(v)}; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// Cross-lane reductions without the LDS crossbar: `__shfl_xor` is ds_bpermute_b32 (an LDS round trip, ~100 cycles, and the
// butterfly is a chain of six of them); within a row of 16 lanes the DPP modifiers move data inside the VALU (quad_perm,
// row_half_mirror, row_mirror, row_ror), and gfx950's v_permlane16_swap / v_permlane32_swap exchange rows and wave halves.
// Every lane ends with the result, as with the xor butterfly (the association order differs: not the same bits).
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {  // every lane has a source lane for the controls used here
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1 /*quad_perm [1,0,3,2]*/, DPP_XOR2 = 0x4E /*quad_perm [2,3,0,1]*/, DPP_HALF_MIRROR = 0x141 /*i <-> 7-i*/,
              DPP_MIRROR = 0x140 /*i <-> 15-i*/, DPP_ROR8 = 0x128 /*i <- (i+8)%16: lane ^ 8*/;
struct OpSum { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
// combine with the lanes 16 / 32 away (lane ^ 16, lane ^ 32): swap(v, v) leaves {v.r0, v.r0, v.r2, v.r2} and {v.r1, v.r1, v.r3, v.r3}
// (rows of 16), resp. {v.lo, v.lo} and {v.hi, v.hi} (halves) - op() of the two is op(v, v[lane ^ 16 or 32]) in every lane
// (the results are copied to scalars first: __builtin_bit_cast applied to r[1] directly reads element 0 with this compiler)
template <typename Op> __device__ __forceinline__ float xor16_reduce(float v, Op op) {
  const unsigned w = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(w, w, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return op(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}
template <typename Op> __device__ __forceinline__ float xor32_reduce(float v, Op op) {
  const unsigned w = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return op(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}
// reduce over aligned groups of N consecutive lanes (N = 2 .. 64)
template <int N, typename Op> __device__ __forceinline__ float group_reduce(float v, Op op) {
  if constexpr (N >= 2) v = op(v, dpp_f<DPP_XOR1>(v));
  if constexpr (N >= 4) v = op(v, dpp_f<DPP_XOR2>(v));
  if constexpr (N >= 8) v = op(v, dpp_f<DPP_HALF_MIRROR>(v));
  if constexpr (N >= 16) v = op(v, dpp_f<DPP_MIRROR>(v));
  if constexpr (N >= 32) v = xor16_reduce(v, op);
  if constexpr (N >= 64) v = xor32_reduce(v, op);
  return v;
}
// reduce over the lanes that share (lane % N): the values N, 2N, ... lanes apart (N = 8, 16, 32)
template <int N, typename Op> __device__ __forceinline__ float stride_reduce(float v, Op op) {
  if constexpr (N <= 8) v = op(v, dpp_f<DPP_ROR8>(v));
  if constexpr (N <= 16) v = xor16_reduce(v, op);
  v = xor32_reduce(v, op);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_reduce<64>(v, OpSum{}); }
__device__ __forceinline__ float wave_max(float v) { return group_reduce<64>(v, OpMax{}); }

// Kernel arguments live in the kernarg segment and reach SGPRs through s_load; the compiler treats those loads as free to
// re-materialise and sinks them next to their first use, which in these kernels produced two or three SERIALISED
// round trips (pointer -> wait -> address -> second pointer -> wait ...) in front of the first weight load - about a
// microsecond each in a decode chain whose kernels only take 3-8.  Passing every argument through an empty asm at kernel
// entry makes the values opaque, so all s_loads are issued together, once.
template <class T> __device__ __forceinline__ T sgpr_pin(T v) {
  asm volatile("" : "+s"(v));
  return v;
}
template <class T> __device__ __forceinline__ T* sgpr_pin_ptr(T* p) {
  typedef __attribute__((address_space(1))) T* global_ptr;  // stay in the global address space: no flat_load fallback
  global_ptr g = (global_ptr)p;
  asm volatile("" : "+s"(g));
  return (T*)g;
}


// One uniform 32-bit word that an EARLIER kernel wrote (a row's `done` flag), fetched by the SCALAR unit (s_load_dword through the
// constant address space): it is requested at kernel entry without occupying the vector-memory queue and is waited for with
// lgkmcnt only where it is first used - a plain load of a possibly-aliased pointer becomes a global_load + s_waitcnt vmcnt(0) in
// front of everything else (measured in the ISA: one dependent round trip at the top of the kernel).  The scalar cache is
// invalidated at every kernel boundary, so a value stored by a previous kernel of the stream is what arrives.
__device__ __forceinline__ int32_t sload_i32(const int32_t* p) {
  typedef const __attribute__((address_space(4))) int32_t* const_ptr;
  return *(const_ptr)p;
}
// done[row] of the decode batch, REQUESTED here and tested later as `done && raw` (done == nullptr: the caller tracks no finished
// rows).  The load is unconditional - from `valid`, any 4-byte-aligned device address, when there are no flags - because a load
// inside an `if (done)` block is waited for at the end of that block, i.e. at once.
__device__ __forceinline__ int row_done_issue(const int32_t* done, int row, const void* valid) {
  return sload_i32(done ? done + row : (const int32_t*)valid);
}

// ---- GEMM epilogue description (shared by every GEMM flavour) ------------------------------------
// C[m][n] = act(alpha-free acc + bias[n]) (+ rowtab[(m % rowmod)][n]) (+ residual[m][n]); written as f32
// and/or T.  headsplit != 0 scatters T output into the cross-KV layout [which][b][h][t][64].
struct GemmEpi {
  const float* bias = nullptr;      // [N]
  int act = 0;                      // 0 none, 1 exact-erf GELU
  const float* rowtab = nullptr;    // [rowmod][N] f32 (encoder positions)
  int rowmod = 0;
  const float* residual = nullptr;  // f32 [M][ldc] (may alias out_f32)
  float* out_f32 = nullptr;         // f32 [M][ldc]
  void* out_t = nullptr;            // T   [M][ldc] (or head-split)
  int64_t ldc = 0;
  int64_t batch_stride_c = 0;       // elements, applied to residual/out_f32/out_t per blockIdx.z
  int headsplit = 0;                // 1: out_t index = which*hs_which + ((b*H+h)*T + t)*64 + j
  int hs_T = 0, hs_H = 0, hs_d = 0;
  int64_t hs_which = 0;
};

struct GemmArgs {
  const void* A = nullptr;  // T [M][lda] (rows may overlap: conv-as-GEMM)
  const void* W = nullptr;  // T [N][ldw]
  int M = 0, N = 0, K = 0;
  int64_t lda = 0, ldw = 0;
  int64_t batch_stride_a = 0;
  int batch = 1;
  // groups > 1 (persistent 256 x 256 kernel only): `groups` GEMMs that share A and differ in W / bias / output - group g uses
  // W + g * group_stride_w, bias + g * N and writes at out + g * group_stride_out (elements); the tiles of group 0 come first,
  // then group 1 ...: one launch walks all of them, so only the LAST group ends in a partial round of workgroups (the 32
  // cross-KV projections of the decoder layers: 60 160 tiles = 235.0 rounds of 256 instead of 32 x 8)
  int groups = 1;
  int64_t group_stride_w = 0, group_stride_out = 0;
  GemmEpi epi;
};

// ---- launchers (defined in the .hip files) ---------------------------------------------------------
template <typename T> void launch_gemm_basic(const GemmArgs& g, hipStream_t s);
// the tiled 16-bit GEMMs (names keep "bf16": the kernels were written for it; T16 = bf16_t or f16_t picks the MFMA form and
// the output conversion, nothing else differs)
template <typename T16> void launch_gemm_bf16_fast(const GemmArgs& g, hipStream_t s);
bool gemm_bf16_fast_ok(const GemmArgs& g);
template <typename T16> void launch_gemm_bf16_v2(const GemmArgs& g, hipStream_t s);
bool gemm_bf16_v2_ok(const GemmArgs& g);
template <typename T16> void launch_gemm_bf16_v3(const GemmArgs& g, hipStream_t s);
bool gemm_bf16_v3_ok(const GemmArgs& g);
template <typename T16> void launch_gemm_bf16_v4(const GemmArgs& g, hipStream_t s);   // persistent form of v3 (T-output epilogues)
bool gemm_bf16_v4_ok(const GemmArgs& g);
template <typename T16> bool launch_gemm_bf16_v5(const GemmArgs& g, hipStream_t s);   // v4 with the last partial round re-tiled (shorter tiles); false = no gain for this shape
bool gemm_bf16_v5_ok(const GemmArgs& g);

template <typename T>
void launch_layernorm(const float* x, const float* gamma, const float* beta, T* out, int rows, int d, hipStream_t s);
// x += delta (the T-typed output of the preceding out-proj / fc2 GEMM), then LayerNorm(x) -> out; delta may alias out
template <typename T>
void launch_layernorm_add(float* x, const T* delta, const float* gamma, const float* beta, T* out, int rows, int d, hipStream_t s);
// round 5: out = LayerNorm(x + delta) without updating x ... and the LayerNorm that later folds that delta and a second one in:
// x = (x + delta) + delta2 (delta2 may alias out).  Same f32 additions in the same order as two launch_layernorm_add calls.
template <typename T>
void launch_layernorm_peek(const float* x, const T* delta, const float* gamma, const float* beta, T* out, int rows, int d, hipStream_t s);
template <typename T>
void launch_layernorm_add2(float* x, const T* delta, const T* delta2, const float* gamma, const float* beta, T* out, int rows, int d,
                           hipStream_t s);
// decode-step LayerNorm, one workgroup per row (kernels_misc.hip), that first completes the residual row:
//   slab form:  x_out[row] = x[row] + bias + slab[0][row] + ... + slab[n_slab-1][row]   (fixed order, no atomics)
//   embed form (tok != nullptr): x_out[row] = emb[tok[row]] + pos[*step]
//   n_slab == 0 and tok == nullptr: plain LayerNorm of x
struct LnPre {
  const float* bias = nullptr;   // [d]
  const float* slab = nullptr;   // [n_slab][slab_stride] f32 partial tiles of the preceding K-split residual GEMM
  int n_slab = 0;                // <= 16 (the decode slabs of the context hold 16 K slices)
  int64_t slab_stride = 0;
  float* x_out = nullptr;        // [rows][d] updated residual rows (may alias x)
  const int32_t* tok = nullptr;  // [rows]
  const int32_t* step = nullptr; // [1]
  const void* emb = nullptr;     // T [V][d]
  const void* pos = nullptr;     // T [n_text_ctx][d]
};
template <typename T>
void launch_layernorm_rows(const float* x, const float* gamma, const float* beta, T* out, int rows, int d, const LnPre& pre,
                           hipStream_t s);

// K-split partial results of a decode GEMM handed to an attention kernel instead of T rows: value = round_T(bias +
// slab[0] + ... + slab[n-1]) (slab order, bit-reproducible); n == 0 means "read the T rows as before"
struct SlabIn {
  const float* slab = nullptr;  // [n][stride] f32, row-major [rows][ld] inside a slab
  const float* bias = nullptr;  // [ld]
  int n = 0;
  int64_t stride = 0;
  int ld = 0;
};

// cross-attention query computed inside the attention kernel (round-4 experiment): x = LayerNorm output rows T [B][d],
// W = the q projection T [d][d] row-major (pre-scaled by 1/8), bias f32 [d]; W == nullptr: not used
struct QProj {
  const void* x = nullptr;
  const void* W = nullptr;
  const float* bias = nullptr;
};

// decode-time weight-streaming GEMM over MFMA-fragment-packed weights (kernels_skinny.hip)
template <typename T16> void launch_shuffle_cast(const float* src, T16* dst_base, int rows, int K, int row_offset, hipStream_t s, int rows_total = 0);
int skinny_rows_per_block(int N, int K);   // 32 or 20 output rows per n-block of a packed decode matrix (kernels_skinny.hip)
extern thread_local int g_skinny_narrow;    // option dec_narrow_blocks (set per C-ABI call from the context, like g_skinny_nt)
// ksplit > 1 (from gemm_skinny_ksplit): workgroup (nb, ks) writes its partial tile to slab[ks]; bias is the consumer's
template <typename T16>
bool launch_gemm_skinny(const T16* Wsh, const T16* x, int B, int N, int K, const GemmEpi& e, hipStream_t s, int ksplit = 1,
                        float* slab = nullptr, int64_t slab_stride = 0);
int gemm_skinny_ksplit(int B, int N, int K, int want);
// vocabulary projection: persistent workgroups that keep the activation rows in registers (kernels_skinny.hip); false: shape unsupported
template <typename T16> bool launch_gemm_vocab(const T16* Wsh, const T16* x, int B, int N, int K, float* out, int64_t ldc, hipStream_t s, int device);
void gemm_vocab_init(int device);   // once per device (std::call_once), outside any stream capture (ttasr_create)
void gemm_tiles_init(int device);   // same for the tiled encoder GEMMs: dynamic-LDS opt-ins + CU count of the persistent grid
// mel
// geom_dev (optional): int64 [B][3] = {lead, reflect_end, valid_frames} per clip - window-of-a-file geometry, kernels_misc.hip
void launch_mel(const float* pcm, int64_t pcm_stride, const int64_t* n_samples_dev, int B, int n_mels, int n_frames,
                const float* filters /*[201][n_mels]*/, const float* dft_cos, const float* dft_sin /*[400]*/,
                const float* window /*[400]*/, float* logmel /*[B][n_mels][n_frames]*/, unsigned* clip_max /*[B]*/,
                hipStream_t s, const int64_t* geom_dev = nullptr);
unsigned mel_max_to_ordered(float v);   // clip_max holds the per-clip maximum as an order-preserving unsigned
float mel_max_from_ordered(unsigned u);
template <typename T>
void launch_mel_finish(float* logmel, const unsigned* clip_max, T* mel_t /*[B][n_frames+2][n_mels]*/, int B, int n_mels,
                       int n_frames, hipStream_t s, const int64_t* geom_dev = nullptr);
template <typename T>
void launch_mel_transpose(const float* mel, T* mel_t, int B, int n_mels, int n_frames, hipStream_t s);

// encoder attention over fused qkv [B*T][3d] -> out [B*T][d]
template <typename T> void launch_enc_attn_simple(const T* qkv, T* out, int B, int T_, int H, hipStream_t s);
template <typename T16> void launch_enc_attn_flash_bf16(const T16* qkv, T16* out, int B, int T_, int H, hipStream_t s);
// cross-attention of n_q consecutive rows per clip against that clip's cross-KV cache, as one MFMA flash pass (prefill)
template <typename T16>
void launch_cross_attn_flash_bf16(const T16* q, const T16* K, const T16* V, T16* out, int n_clips, int n_q, int H, int Tk, hipStream_t s);

// decoder
// The rule scalars that change from one 30-s window to the next (prompt geometry with condition_on_previous_text, the token
// budget, the seed of a fallback attempt).  select_kernel reads them from device memory, so the captured decode-step graphs
// survive the change (everything else in RuleParams is baked into the captured launch).
struct RuleDyn { int32_t max_prompt, max_new, sot_index; uint32_t seed; };
struct DecState {            // device-resident per-row search state
  int32_t* cur_tok;          // [B] token fed at this step
  int32_t* step;             // [1] position of cur_tok
  int32_t* n_sampled;        // [B]
  int32_t* last_tok;         // [B] last sampled (or -1)
  int32_t* pen_tok;          // [B] penultimate sampled (or -1)
  int32_t* last_ts;          // [B] most recent timestamp token sampled (or -1)
  int32_t* done;             // [B]
  int32_t* n_done;           // [1]
  float* sum_logprob;        // [B]
  float* no_speech;          // [B]
  int32_t* out_tokens;       // [B][max_new]
  const int32_t* prompt;     // [B][max_prompt]
  const int32_t* prompt_len; // [B]
  const uint8_t* mask;       // [V] bit0 suppress, bit1 begin-suppress
  const RuleDyn* dyn;        // [1] overrides RuleParams.{max_prompt, max_new, sot_index, seed} in select_kernel
  const int32_t* row_cap;    // [B] per-row token budget (round 6, ttasr_generate_capped): row b is finished after
                             // min(row_cap[b], max_new) sampled tokens; reset_search fills it with a huge value
};
struct RuleParams {
  int V, ldv, max_prompt, max_new;
  int eot, no_timestamps, timestamp_begin, no_speech, sot_index, timestamps, max_initial, suppress_eot;
  float temperature;  // 0: argmax; > 0: sample from softmax(logits / temperature) by Gumbel-max with a counter hash
  uint32_t seed;
};
template <typename T>
void launch_cross_attn_probs(const T* q, const T* K, const T* V, T* out, int rows, int H, int Tk, const int* sel /*[H] dev*/,
                             float* probs /*[n_sel][rows][Tk]*/, hipStream_t s);
void launch_token_logprob(const float* logits, int ldv, int V, const int32_t* target, float* out, int rows, hipStream_t s);
void launch_token_prob(const float* logits, int ldv, int V, int tok, float* out /*[rows]*/, int rows, hipStream_t s);
template <typename T>
void launch_embed_prefill(const int32_t* prompt, int max_prompt, int rows_per_prompt, int n_seq, int npos, const T* emb, const T* pos,
                          float* x, int d, hipStream_t s);
template <typename T>
void launch_self_attn_prefill(const T* qkv /*[n_seq*npos][3d]*/, T* kv_pool, const int32_t* page_table, int pages_per_seq,
                              int64_t pool_layer_off, int identity_pages, T* out, int n_seq, int npos, int H, hipStream_t s);
template <typename T>
void launch_self_attn_decode(const T* qkv /*[B][3d]*/, T* kv_pool, const int32_t* page_table, int pages_per_seq,
                             int64_t pool_layer_off, int identity_pages, int row0, const int32_t* step, T* out /*[B][d]*/, int B, int H,
                             hipStream_t s, SlabIn sq = SlabIn{} /*qkv from K-split partial tiles*/,
                             const int32_t* done = nullptr /*[B] (already offset by row0): finished rows leave the kernel*/);
template <typename T>
void launch_copy_pages(T* pool, const int32_t* pairs_dev, int n_pairs, int n_layers, int H, int64_t layer_elems, hipStream_t s);
template <typename T>
void launch_cross_attn_decode(const T* q /*[B][d]*/, const T* K, const T* V /*[B / kv_div][H][Tk][64]*/, T* out, int B, int H,
                              int Tk, int kv_div, hipStream_t s,
                              float* split_ws = nullptr /*[B*H*8][66]: enables the split-frame variant for small B*H*/,
                              SlabIn sq = SlabIn{} /*q from K-split partial tiles*/,
                              int ws_rows = 0 /*rows the workspace was sized for (0: B); rows that share a clip (kv_div 2..8)
                                                are served by one K/V stream per clip when they fit*/,
                              QProj qp = QProj{} /*W != nullptr: the query is projected inside the kernel (B * H >= 256, kv_div == 1)*/,
                              const int32_t* done = nullptr /*[B] device flags: rows with done[b] != 0 are FINISHED (EOT / token budget) and
                                                              leave the kernel without streaming their cross-KV; their `out` rows keep
                                                              the last live values.  nullptr: every row is live*/);
// Signature of the kernel a launcher picked, in the spelling rocprofv3 prints ("name<template arguments> grid <threads>"):
// recorded by the launchers of the measured kernels while g_kernel_sig_on is set (ttasr_bench_kernel), so that bench.py can tell
// whether a committed counter profile still describes what the run launches.
// opt-in fp8 (e4m3) copy of the cross-KV cache and the decode-step cross-attention that reads it (kernels_fp8.hip)
template <typename T> void launch_xkv_quant(const T* src, uint8_t* dst, float* scale, int64_t n_blocks, int rows, hipStream_t s);
template <typename T>
bool launch_cross_attn_fp8(const T* q, const uint8_t* K8, const uint8_t* V8, const float* kscale, const float* vscale, T* out, int B, int H,
                           int Tk, hipStream_t s, struct SlabIn sq, const int32_t* done = nullptr);
extern thread_local bool g_kernel_sig_on;
extern thread_local char g_kernel_sig[192];
template <typename T> inline const char* sig_type() { return sizeof(T) == 4 ? "float" : "unsigned short"; }
template <> inline const char* sig_type<struct f16_t>() { return "f16_t"; }

// Kernel-variant switches of the launchers (A/B experiments).  Thread-local: every C-ABI call copies its CONTEXT's setting in
// before it launches anything (engine.hip guarded()), so an option set on one context never changes what another context's
// thread launches or what its captured graphs hold.
// A launcher asked for a configuration it has no kernel for (unreachable behind ttasr_create's geometry limits and
// ttasr_set_option's ranges; kept as a guard for future callers): it records the reason here and launches NOTHING; the C-ABI
// call that was running returns TTASR_E_INVALID with this text (engine.hip guarded()).  A library never abort()s its host.
extern thread_local char g_launch_fault[160];
void launch_fault(const char* fmt, ...);
extern thread_local int g_skinny_x_lds;  // option dec_x_lds: decode GEMMs stage their activation tile through LDS (kernels_skinny.hip)
extern thread_local int g_skinny_nt;     // option weights_nontemporal: nontemporal weight loads in the decode GEMMs
extern thread_local int g_xattn_variant;  // option xattn_nontemporal: cross-attention kernel variant
extern thread_local int g_flash_qw;           // option flash_qw (kernels_flash.hip): query blocks of 32 per wave, 1 | 2
extern thread_local int g_xattn_mq_slices;    // option xattn_mq_slices (A/B)
extern thread_local int g_xattn_deep_items;   // option xattn_deep_items (kernels_attn.hip cross_attn_pipe_kernel)
// beam search: processed log-probabilities and ids of the k best tokens of every row (rules applied from the
// per-row history state uploaded by the host)
struct BeamRowState { const int32_t *n_sampled, *last_tok, *pen_tok, *last_ts; const uint8_t* mask; };
void launch_beam_topk(const float* logits, BeamRowState st, RuleParams rp, int R, int k, float* out_lp /*[R][k]*/,
                      int32_t* out_id /*[R][k]*/, float* out_no_speech /*[R] or null*/, hipStream_t s);
// ticket != nullptr: the workgroup that finishes last (of `total_rows` over all select launches of the step) advances
// *st.step, which replaces the separate advance launch
void launch_select(const float* logits, DecState st, RuleParams rp, int B, float* out_rows /*nullable*/, hipStream_t s,
                   int32_t* ticket, int total_rows);
void launch_advance(int32_t* step, hipStream_t s);
void launch_prep_weight(const void* src, int src_type /*0 f32, 1 bf16 bits, 2 fp16 bits*/, float* dst, int64_t n, int64_t conv_in, float scale,
                        hipStream_t s);
template <typename T> void launch_cast(const float* in, T* out, int64_t n, hipStream_t s);
template <typename T> void launch_uncast(const T* in, float* out, int64_t n, hipStream_t s);
