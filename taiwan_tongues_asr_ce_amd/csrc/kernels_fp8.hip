// Opt-in serving mode (option "xkv_fp8", never the headline configuration): the decoder's cross-attention K / V cache in
// OCP fp8 e4m3 with one f32 scale per (layer, K | V, clip, head) block.  SURVEY.md section 7 names it as the lever on the 7.87 GB
// a decode step streams: the cache is written once per clip and read once per (token, layer), so halving its bytes halves the
// HBM time of the dominant kernel.  The 16-bit cache stays the source of truth (every other path - beam search, prefill,
// alignment, the split-frame small-batch kernels - keeps reading it); this file adds
//   xkv_quant_kernel       one workgroup per (layer, K | V, clip, head) block of Tk x 64 stored values: pass 1 the block's |max|,
//                          pass 2 (the block is L2 / Infinity-Cache resident) value * 448 / |max| -> e4m3, scale = |max| / 448
//   cross_attn_fp8_kernel  the software-pipelined decode-step kernel of kernels_attn.hip on 64-byte rows: 16 values per lane, 4 lanes
//                          per frame, 16 frames per wave-instruction; the K scale is folded into the query, the V scale into the
//                          normalisation; scores, softmax and accumulation in f32 exactly as in the 16-bit kernel.
// Accuracy is reported, not assumed: tools/fp8_agreement.py + bench.py --xkv-fp8 (token agreement with the bf16 engine and the f32
// parity engine on the headline workload), tests/test_gpu_fp8.py (attention output vs the 16-bit kernel, token equality under margin).
#include "common.hpp"

namespace {
using u32x4q = __attribute__((ext_vector_type(4))) unsigned;
typedef float f32x2q __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void unpack16_fp8(const u32x4q& r, float (&v)[16]) {
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2q lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
    const f32x2q hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
    v[4 * i] = lo[0]; v[4 * i + 1] = lo[1]; v[4 * i + 2] = hi[0]; v[4 * i + 3] = hi[1];
  }
}
constexpr int DPP_ROR4 = 0x124;   // row_ror:4 - lane i <- lane (i + 4) % 16 of its row
// sum over the lanes that share (lane % 4): 4, 8, 12 lanes apart inside a row of 16, then the other rows and the other half
__device__ __forceinline__ float stride4_sum(float v) {
  v += dpp_f<DPP_ROR4>(v);
  v += dpp_f<DPP_ROR8>(v);
  v = xor16_reduce(v, OpSum{});
  v = xor32_reduce(v, OpSum{});
  return v;
}
}  // namespace

// src: T [n_blocks][rows * 64]; dst: fp8 [n_blocks][rows * 64]; scale: f32 [n_blocks] (dequantised value = fp8 * scale)
template <typename T>
__global__ __launch_bounds__(256) void xkv_quant_kernel(const T* __restrict__ src, uint8_t* __restrict__ dst, float* __restrict__ scale,
                                                        int rows) {
  static_assert(sizeof(T) == 2, "16-bit cache only");
  __shared__ float red[4];
  const int64_t blk = blockIdx.x;
  const int n16 = rows * 64 / 8;   // 16-byte chunks of 8 stored values
  const uint4* s = (const uint4*)(src + blk * rows * 64);
  float amax = 0.f;
  for (int i = threadIdx.x; i < n16; i += 256) {
    float v[8];
    up8<T>(s[i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;   // e4m3 finite maximum 448
  const float inv = 1.0f / sc;
  if (threadIdx.x == 0) scale[blk] = sc;
  uint2* d = (uint2*)(dst + blk * rows * 64);
  for (int i = threadIdx.x; i < n16; i += 256) {
    float v[8];
    up8<T>(s[i], v);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
    d[i] = make_uint2((unsigned)lo, (unsigned)hi);
  }
}
template <typename T>
void launch_xkv_quant(const T* src, uint8_t* dst, float* scale, int64_t n_blocks, int rows, hipStream_t s) {
  if constexpr (sizeof(T) == 2) hipLaunchKernelGGL(xkv_quant_kernel<T>, dim3((unsigned)n_blocks), dim3(256), 0, s, src, dst, scale, rows);
}
template void launch_xkv_quant<bf16_t>(const bf16_t*, uint8_t*, float*, int64_t, int, hipStream_t);
template void launch_xkv_quant<f16_t>(const f16_t*, uint8_t*, float*, int64_t, int, hipStream_t);
template void launch_xkv_quant<float>(const float*, uint8_t*, float*, int64_t, int, hipStream_t);

// One workgroup (4 waves) per (row b, head h); K8 / V8: fp8 [B][H][Tk][64]; kscale / vscale: f32 [B][H].
template <typename T, bool QSLAB, int U>
__global__ __launch_bounds__(256) void cross_attn_fp8_kernel(const T* q, const uint8_t* K8, const uint8_t* V8, const float* kscale,
                                                             const float* vscale, T* out, const int32_t* done, int H, int Tk,
                                                             SlabIn sq) {
  static_assert(sizeof(T) == 2, "16-bit activations only");
  constexpr int NWV = 4, LPR = 4, RPI = 16, TSTEP = NWV * RPI;   // 64 frames per iteration of the workgroup
  extern __shared__ float sc[];  // [Tk] scores, then [NWV][64] partial outputs, [2 * NWV] reductions
  q = sgpr_pin_ptr(q); K8 = sgpr_pin_ptr(K8); V8 = sgpr_pin_ptr(V8); kscale = sgpr_pin_ptr(kscale); vscale = sgpr_pin_ptr(vscale);
  out = sgpr_pin_ptr(out); done = sgpr_pin_ptr(done); H = sgpr_pin(H); Tk = sgpr_pin(Tk);
  sq.slab = sgpr_pin_ptr(sq.slab); sq.bias = sgpr_pin_ptr(sq.bias); sq.n = sgpr_pin(sq.n); sq.stride = sgpr_pin(sq.stride);
  const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int done_raw = row_done_issue(done, b, K8);   // finished row of the batch (round 6; kernels_attn.hip cross_attn_pipe_kernel row_done_exit)
  const int d = H * 64;
  const int sub = lane % LPR, rin = lane / LPR;
  float* part = sc + Tk;
  float* red = part + NWV * 64;
  const uint8_t* Kp = K8 + ((int64_t)b * H + h) * Tk * 64 + sub * 16;
  const uint8_t* Vp = V8 + ((int64_t)b * H + h) * Tk * 64 + sub * 16;
  const int n_it = (Tk + TSTEP - 1) / TSTEP;
  const int trow = wave * RPI + rin;
  auto issue = [&](const uint8_t* base, int it0, u32x4q (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min((it0 + u) * TSTEP + trow, Tk - 1);   // clamped, unconditional
      r[u] = __builtin_nontemporal_load((const u32x4q*)(base + (int64_t)t * 64));
    }
  };
  u32x4q ra[U], rb[U];
  issue(Kp, 0, ra);
  const float ks = kscale[b * H + h], vs = vscale[b * H + h];
  float qv[16];
  {
    float q0[8], q1[8];
    const int64_t off = (int64_t)b * d + h * 64 + sub * 16;
    if constexpr (QSLAB) {
      // the two 8-value chunks of this lane's 16 query values, each summed from the q GEMM's K-split partial tiles
      float4 t[4][4], bs[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[c] = *(const float4*)(sq.bias + h * 64 + sub * 16 + 4 * c);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* p = sq.slab + (int64_t)min(s, sq.n - 1) * sq.stride + off;
#pragma unroll
        for (int c = 0; c < 4; ++c) t[s][c] = *(const float4*)(p + 4 * c);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 a = bs[c];
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s == 0 || s < sq.n) { a.x += t[s][c].x; a.y += t[s][c].y; a.z += t[s][c].z; a.w += t[s][c].w; }
        float* dst = c < 2 ? q0 + 4 * c : q1 + 4 * (c - 2);
        dst[0] = to_f<T>(from_f<T>(a.x)); dst[1] = to_f<T>(from_f<T>(a.y)); dst[2] = to_f<T>(from_f<T>(a.z)); dst[3] = to_f<T>(from_f<T>(a.w));
      }
    } else {
      const uint4 a = *(const uint4*)(q + off), c = *(const uint4*)(q + off + 8);
      up8<T>(a, q0); up8<T>(c, q1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { qv[j] = q0[j] * ks; qv[8 + j] = q1[j] * ks; }   // K scale folded into the query
  }
  if (done && done_raw) { if (Tk < 0) sc[0] = __uint_as_float(ra[0].x ^ ra[U - 1].x) + qv[0] + vs; return; }   // row_done_exit
  float mloc = -1e30f;
  auto score = [&](int it0, const u32x4q (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (it0 + u) * TSTEP + trow;
      float kf[16];
      unpack16_fp8(r[u], kf);
      float s = 0.f;
      if (t < Tk) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s = fmaf(qv[j], kf[j], s);
      }
      s = group_reduce<LPR>(s, OpSum{});
      if (t < Tk) {
        if (sub == 0) sc[t] = s;
        mloc = fmaxf(mloc, s);
      }
    }
  };
  for (int it0 = 0; it0 < n_it; it0 += 2 * U) {
    if (it0 + U < n_it) issue(Kp, it0 + U, rb);
    __builtin_amdgcn_sched_barrier(0);
    score(it0, ra);
    __builtin_amdgcn_sched_barrier(0);
    if (it0 + 2 * U < n_it) issue(Kp, it0 + 2 * U, ra);
    __builtin_amdgcn_sched_barrier(0);
    if (it0 + U < n_it) score(it0 + U, rb);
    __builtin_amdgcn_sched_barrier(0);
  }
  issue(Vp, 0, ra);
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
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  auto accum = [&](int it0, const u32x4q (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (it0 + u) * TSTEP + trow;
      if (t < Tk) {
        float vf[16];
        unpack16_fp8(r[u], vf);
        const float p = sc[t];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
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
  for (int j = 0; j < 16; ++j) acc[j] = stride4_sum(acc[j]);
  if (rin == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) part[wave * 64 + sub * 16 + j] = acc[j];
  }
  __syncthreads();
  if (tid < 64) {
    const float v = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
    out[(int64_t)b * d + h * 64 + tid] = from_f<T>(v * vs / denom);   // V scale folded into the normalisation
  }
}

// false: shape unsupported (the caller takes the 16-bit kernel)
template <typename T>
bool launch_cross_attn_fp8(const T* q, const uint8_t* K8, const uint8_t* V8, const float* kscale, const float* vscale, T* out, int B, int H,
                           int Tk, hipStream_t s, SlabIn sq, const int32_t* done) {
  if constexpr (sizeof(T) != 2) return false;
  else {
    if (sq.n > 4 || Tk < 1) return false;
    const size_t lds = sizeof(float) * (Tk + 4 * 64 + 2 * 4);
    if (sq.n > 0) hipLaunchKernelGGL((cross_attn_fp8_kernel<T, true, 4>), dim3(H, B), dim3(256), lds, s, q, K8, V8, kscale, vscale, out, done, H, Tk, sq);
    else hipLaunchKernelGGL((cross_attn_fp8_kernel<T, false, 4>), dim3(H, B), dim3(256), lds, s, q, K8, V8, kscale, vscale, out, done, H, Tk, sq);
    return true;
  }
}
template bool launch_cross_attn_fp8<bf16_t>(const bf16_t*, const uint8_t*, const uint8_t*, const float*, const float*, bf16_t*, int, int, int,
                                            hipStream_t, SlabIn, const int32_t*);
template bool launch_cross_attn_fp8<f16_t>(const f16_t*, const uint8_t*, const uint8_t*, const float*, const float*, f16_t*, int, int, int,
                                           hipStream_t, SlabIn, const int32_t*);
template bool launch_cross_attn_fp8<float>(const float*, const uint8_t*, const uint8_t*, const float*, const float*, float*, int, int, int,
                                           hipStream_t, SlabIn, const int32_t*);
