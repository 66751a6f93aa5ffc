// Log-mel front end, LayerNorm and small utility kernels.
#include "common.hpp"
#include <cstring>

// ------------------------------------------------------------------------------------------------
// a5: log-mel.  Restates FeatureExtractor.__call__ of faster-whisper == HF feature_extraction_whisper.py
// :135-168: reflect-pad 200, Hann-400 frames at hop 160, |DFT|^2 (201 bins), slaney mel matmul,
// log10(max(., 1e-10)); the per-clip "max - 8" clamp and (x+4)/4 need the clip maximum, so this kernel
// writes raw log10 values and an ordered-uint atomic max per clip, and mel_finish applies the rest.
// One workgroup = 8 consecutive frames of one clip: the 1520 samples they span are read once, coalesced,
// into LDS; thread k owns DFT bin k for all 8 frames (twiddles from a 400-entry LDS table, index k*n mod
// 400); then thread m owns mel bin m.  HBM traffic = PCM once + output once.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

constexpr int MEL_F = 8;             // frames per workgroup
constexpr int MEL_SPAN = 160 * (MEL_F - 1) + 400;  // 1520

// Window geometry (geom != nullptr; file-level transcription): clip b is a 30-s WINDOW of a longer recording.  geom[b] =
// {lead, reflect_end, valid_frames}: x points `lead` samples (0 at the start of the file, else 200) before the centre of
// frame 0, so interior windows see their true neighbour samples instead of a reflection; the signal is reflected at
// sample index reflect_end (the end of the FILE when it falls inside this window's span, as a whole-file STFT does) and
// frames >= valid_frames lie beyond the recording: the finishing kernel sets them to 0 in feature space, exactly what
// faster-whisper's pad_or_trim(features[:, seek:seek + 3000]) and HF's long-form loop feed the encoder.  geom == nullptr
// is the single-clip form (pad / trim the clip to one window, reflect at the window end), unchanged.
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ pcm, int64_t pcm_stride,
                                                  const int64_t* __restrict__ n_samples, int n_mels, int n_frames,
                                                  const float* __restrict__ filters, const float* __restrict__ dcos,
                                                  const float* __restrict__ dsin, const float* __restrict__ window,
                                                  float* __restrict__ logmel, unsigned* __restrict__ clip_max,
                                                  const int64_t* __restrict__ geom) {
  __shared__ float xs[MEL_SPAN];
  __shared__ float tc[400], tsn[400], win[400];
  __shared__ float pw[MEL_F][208];
  __shared__ float red[4];
  const int b = blockIdx.y, f0 = blockIdx.x * MEL_F, tid = threadIdx.x;
  const int64_t lead = geom ? geom[3 * b] : 0;
  const int64_t total = geom ? geom[3 * b + 1] : (int64_t)n_frames * 160;  // where the signal is reflected
  const int valid_frames = geom ? (int)geom[3 * b + 2] : n_frames;
  const int64_t n_valid = geom ? n_samples[b] : min((int64_t)n_frames * 160, n_samples[b]);
  const float* x = pcm + (int64_t)b * pcm_stride;
  for (int i = tid; i < 400; i += 256) { tc[i] = dcos[i]; tsn[i] = dsin[i]; win[i] = window[i]; }
  // sample index of xs[i] in the padded clip: s = f0*160 - 200 + i, reflected at both ends
  for (int i = tid; i < MEL_SPAN; i += 256) {
    int64_t sidx = lead + (int64_t)f0 * 160 - 200 + i;
    if (sidx < 0) sidx = -sidx;
    if (sidx >= total) sidx = 2 * (total - 1) - sidx;
    xs[i] = (sidx >= 0 && sidx < n_valid) ? x[sidx] : 0.0f;
  }
  __syncthreads();
  if (tid < 201) {
    float re[MEL_F], im[MEL_F];
#pragma unroll
    for (int f = 0; f < MEL_F; ++f) { re[f] = 0.f; im[f] = 0.f; }
    int idx = 0;
    for (int n = 0; n < 400; ++n) {
      float c = tc[idx], s = tsn[idx], w = win[n];
#pragma unroll
      for (int f = 0; f < MEL_F; ++f) {
        float v = xs[f * 160 + n] * w;
        re[f] = fmaf(v, c, re[f]);
        im[f] = fmaf(v, s, im[f]);
      }
      idx += tid;
      if (idx >= 400) idx -= 400;
    }
#pragma unroll
    for (int f = 0; f < MEL_F; ++f) pw[f][tid] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();
  float lmax = -1e30f;
  for (int o = tid; o < n_mels * MEL_F; o += 256) {
    int m = o % n_mels, f = o / n_mels;
    if (f0 + f < n_frames) {
      float acc = 0.f;
      for (int k = 0; k < 201; ++k) acc = fmaf(pw[f][k], filters[k * n_mels + m], acc);
      float l = log10f(fmaxf(acc, 1e-10f));
      logmel[((int64_t)b * n_mels + m) * n_frames + f0 + f] = l;
      if (f0 + f < valid_frames) lmax = fmaxf(lmax, l);  // the maximum is taken over the recording, not over its padding
    }
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) atomicMax(&clip_max[b], f2ord(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

void launch_mel(const float* pcm, int64_t pcm_stride, const int64_t* n_samples_dev, int B, int n_mels, int n_frames,
                const float* filters, const float* dft_cos, const float* dft_sin, const float* window, float* logmel,
                unsigned* clip_max, hipStream_t s, const int64_t* geom_dev) {
  hipMemsetAsync(clip_max, 0, sizeof(unsigned) * B, s);
  dim3 grid((n_frames + MEL_F - 1) / MEL_F, B);
  hipLaunchKernelGGL(mel_kernel, grid, dim3(256), 0, s, pcm, pcm_stride, n_samples_dev, n_mels, n_frames, filters,
                     dft_cos, dft_sin, window, logmel, clip_max, geom_dev);
}
unsigned mel_max_to_ordered(float v) { unsigned u; memcpy(&u, &v, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
float mel_max_from_ordered(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float v; memcpy(&v, &u, 4); return v; }

// normalise in place ([B][M][F] f32, the API-visible layout) and write the encoder's input image:
// time-major [B][F+2][M] in T with a zero row before and after each clip (conv padding, see engine).
template <typename T>
__global__ __launch_bounds__(256) void mel_finish_kernel(float* __restrict__ logmel, const unsigned* __restrict__ clip_max,
                                                         T* __restrict__ mel_t, int n_mels, int n_frames, int normalise,
                                                         const int64_t* __restrict__ geom) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, f0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float floor_v = normalise ? ord2f(clip_max[b]) - 8.0f : 0.f;
  const int valid_frames = geom ? (int)geom[3 * b + 2] : n_frames;  // frames beyond the recording: 0 in feature space
  for (int r = ty; r < 32; r += 8) {
    int m = m0 + r, f = f0 + tx;
    if (m < n_mels && f < n_frames) {
      int64_t i = ((int64_t)b * n_mels + m) * n_frames + f;
      float v = logmel[i];
      if (normalise) {
        v = f < valid_frames ? (fmaxf(v, floor_v) + 4.0f) * 0.25f : 0.f;
        logmel[i] = v;
      }
      tile[r][tx] = v;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int f = f0 + r, m = m0 + tx;
    if (m < n_mels && f < n_frames) mel_t[((int64_t)b * (n_frames + 2) + f + 1) * n_mels + m] = from_f<T>(tile[tx][r]);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {  // zero the two padding rows of this clip
    for (int i = threadIdx.x; i < n_mels; i += 256) {
      mel_t[(int64_t)b * (n_frames + 2) * n_mels + i] = from_f<T>(0.f);
      mel_t[((int64_t)b * (n_frames + 2) + n_frames + 1) * n_mels + i] = from_f<T>(0.f);
    }
  }
}

template <typename T>
void launch_mel_finish(float* logmel, const unsigned* clip_max, T* mel_t, int B, int n_mels, int n_frames, hipStream_t s,
                       const int64_t* geom_dev) {
  dim3 grid((n_frames + 31) / 32, (n_mels + 31) / 32, B);
  hipLaunchKernelGGL(mel_finish_kernel<T>, grid, dim3(256), 0, s, logmel, clip_max, mel_t, n_mels, n_frames, 1, geom_dev);
}
template <typename T>
void launch_mel_transpose(const float* mel, T* mel_t, int B, int n_mels, int n_frames, hipStream_t s) {
  dim3 grid((n_frames + 31) / 32, (n_mels + 31) / 32, B);
  hipLaunchKernelGGL(mel_finish_kernel<T>, grid, dim3(256), 0, s, (float*)mel, (const unsigned*)nullptr, mel_t, n_mels,
                     n_frames, 0, (const int64_t*)nullptr);
}
template void launch_mel_finish<float>(float*, const unsigned*, float*, int, int, int, hipStream_t, const int64_t*);
template void launch_mel_finish<bf16_t>(float*, const unsigned*, bf16_t*, int, int, int, hipStream_t, const int64_t*);
template void launch_mel_finish<f16_t>(float*, const unsigned*, f16_t*, int, int, int, hipStream_t, const int64_t*);
template void launch_mel_transpose<float>(const float*, float*, int, int, int, hipStream_t);
template void launch_mel_transpose<bf16_t>(const float*, bf16_t*, int, int, int, hipStream_t);
template void launch_mel_transpose<f16_t>(const float*, f16_t*, int, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5), f32 residual stream in -> T out.  One wave per row; the row is read ONCE into
// registers (d <= 1280), then mean and the two-pass variance (as the oracle) come from registers.
// HBM-bound: rows*d*(4 + sizeof(T)) bytes.
// ------------------------------------------------------------------------------------------------
// ADD (encoder, bf16 mode): the row is first completed as x += delta, where delta is the T-typed output of the
// out-proj / fc2 GEMM that precedes the LayerNorm (bias already added), and written back to the f32 residual stream.
// Moving the residual add out of those GEMMs' epilogues (a 492 MB f32 read-modify-write per GEMM at B = 32, exposed at one
// workgroup per CU) into this HBM-streaming kernel lets them run the plain "bias -> T" epilogue.  delta may alias `out`
// (each lane reads its delta chunks before it writes the same positions of the output).
// Round 5 - two more forms that save a third of the f32 residual round trips of an encoder layer (bit-identical: the same two
// f32 additions in the same order):  ADD = 2: normalise x + delta WITHOUT writing the sum back (the LayerNorm after the attention
// out-projection: its delta stays in its buffer);  ADD = 3: x = (x + delta) + delta2, written back, then normalised (the next
// LayerNorm, after fc2: it folds both deltas of the layer in).  Per layer and element 8 + 14 bytes instead of 12 + 12.
template <typename T, int NV, int ADD>  // NV = float4 per lane kept in registers: d <= 256 * NV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* out, int rows, int d,
                                                        const T* delta, float* x_out /* ADD 1 / 3: the same rows as x */,
                                                        const T* delta2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + (int64_t)row * d);
  const int nv = d >> 2;
  // every load of the kernel is issued up front (clamped, unconditional): x row, gamma, beta - one memory
  // round trip in total; a load placed after the reductions costs a second one (~1 us in a decode chain)
  float4 v[NV], gm[NV], bt[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = min(lane + 64 * j, nv - 1);
    v[j] = xr[i];
    gm[j] = ((const float4*)gamma)[i];
    bt[j] = ((const float4*)beta)[i];
  }
  if constexpr (ADD != 0) {
    auto load_delta = [&](const T* dp, float4 (&dl)[NV]) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i = min(lane + 64 * j, nv - 1);
        if constexpr (sizeof(T) == 4) {
          dl[j] = ((const float4*)(dp + (int64_t)row * d))[i];
        } else {
          const uint2 t = ((const uint2*)(dp + (int64_t)row * d))[i];
          N16<T>::up2(t.x, dl[j].x, dl[j].y);
          N16<T>::up2(t.y, dl[j].z, dl[j].w);
        }
      }
    };
    float4 dl[NV];
    load_delta(delta, dl);
    if constexpr (ADD == 3) {
      float4 d2[NV];
      load_delta(delta2, d2);     // both deltas requested before either is used
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        v[j].x += dl[j].x; v[j].y += dl[j].y; v[j].z += dl[j].z; v[j].w += dl[j].w;
        v[j].x += d2[j].x; v[j].y += d2[j].y; v[j].z += d2[j].z; v[j].w += d2[j].w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) { v[j].x += dl[j].x; v[j].y += dl[j].y; v[j].z += dl[j].z; v[j].w += dl[j].w; }
    }
    if constexpr (ADD != 2) {
      float4* xo = (float4*)(x_out + (int64_t)row * d);
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (lane + 64 * j < nv) xo[lane + 64 * j] = v[j];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (lane + 64 * j < nv) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (lane + 64 * j < nv) {
      float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
      q += (a * a + b * b) + (c * c + e * e);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / d + 1e-5f);
  T* o = out + (int64_t)row * d;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      float r0 = (v[j].x - mean) * rstd * gm[j].x + bt[j].x, r1 = (v[j].y - mean) * rstd * gm[j].y + bt[j].y;
      float r2 = (v[j].z - mean) * rstd * gm[j].z + bt[j].z, r3 = (v[j].w - mean) * rstd * gm[j].w + bt[j].w;
      if constexpr (sizeof(T) == 4) {
        ((float4*)o)[i] = make_float4(r0, r1, r2, r3);
      } else {
        uint2 p;
        p.x = N16<T>::pk(r0, r1);
        p.y = N16<T>::pk(r2, r3);
        ((uint2*)o)[i] = p;
      }
    }
  }
}
template <typename T, int ADD>
static void launch_layernorm_impl(const float* x, const float* gamma, const float* beta, T* out, int rows, int d, const T* delta,
                                  float* x_out, hipStream_t s, const T* delta2 = nullptr) {
  dim3 grid((rows + 3) / 4), block(256);
  if (d <= 256) hipLaunchKernelGGL((layernorm_kernel<T, 1, ADD>), grid, block, 0, s, x, gamma, beta, out, rows, d, delta, x_out, delta2);
  else if (d <= 512) hipLaunchKernelGGL((layernorm_kernel<T, 2, ADD>), grid, block, 0, s, x, gamma, beta, out, rows, d, delta, x_out, delta2);
  else if (d <= 768) hipLaunchKernelGGL((layernorm_kernel<T, 3, ADD>), grid, block, 0, s, x, gamma, beta, out, rows, d, delta, x_out, delta2);
  else if (d <= 1024) hipLaunchKernelGGL((layernorm_kernel<T, 4, ADD>), grid, block, 0, s, x, gamma, beta, out, rows, d, delta, x_out, delta2);
  else hipLaunchKernelGGL((layernorm_kernel<T, 5, ADD>), grid, block, 0, s, x, gamma, beta, out, rows, d, delta, x_out, delta2);
}
template <typename T>
void launch_layernorm(const float* x, const float* gamma, const float* beta, T* out, int rows, int d, hipStream_t s) {
  launch_layernorm_impl<T, 0>(x, gamma, beta, out, rows, d, nullptr, nullptr, s);
}
template <typename T>
void launch_layernorm_add(float* x, const T* delta, const float* gamma, const float* beta, T* out, int rows, int d, hipStream_t s) {
  launch_layernorm_impl<T, 1>(x, gamma, beta, out, rows, d, delta, x, s);
}
// out = LayerNorm(x + delta); x itself is NOT updated (the caller folds delta in later: launch_layernorm_add2)
template <typename T>
void launch_layernorm_peek(const float* x, const T* delta, const float* gamma, const float* beta, T* out, int rows, int d, hipStream_t s) {
  launch_layernorm_impl<T, 2>(x, gamma, beta, out, rows, d, delta, nullptr, s);
}
// x = (x + delta) + delta2; out = LayerNorm(x).  delta2 may alias out.
template <typename T>
void launch_layernorm_add2(float* x, const T* delta, const T* delta2, const float* gamma, const float* beta, T* out, int rows, int d,
                           hipStream_t s) {
  launch_layernorm_impl<T, 3>(x, gamma, beta, out, rows, d, delta, x, s, delta2);
}
#define TTASR_LN_EXTRA(T_)                                                                                                         \
  template void launch_layernorm_peek<T_>(const float*, const T_*, const float*, const float*, T_*, int, int, hipStream_t);         \
  template void launch_layernorm_add2<T_>(float*, const T_*, const T_*, const float*, const float*, T_*, int, int, hipStream_t)
TTASR_LN_EXTRA(float); TTASR_LN_EXTRA(bf16_t); TTASR_LN_EXTRA(f16_t);
#undef TTASR_LN_EXTRA
template void launch_layernorm_add<float>(float*, const float*, const float*, const float*, float*, int, int, hipStream_t);
template void launch_layernorm_add<bf16_t>(float*, const bf16_t*, const float*, const float*, bf16_t*, int, int, hipStream_t);
template void launch_layernorm_add<f16_t>(float*, const f16_t*, const float*, const float*, f16_t*, int, int, hipStream_t);
// ------------------------------------------------------------------------------------------------
// Decode-step LayerNorm (rows <= 128): one WORKGROUP per row, one float4 per thread, so the few rows of a decode
// step spread over as many CUs as there are rows and every thread has a single round trip of (4 + n_slab)
// independent loads.  The row is first completed from the K-split partial tiles of the residual GEMM before it:
//     x_new = x + bias + slab[0] + slab[1] + ... + slab[n_slab-1]        (fixed order: bit-reproducible)
// written back to the f32 residual stream, then normalised to T.  `tok != nullptr` instead CREATES the row as the
// token + position embedding of this step (first LayerNorm of the decoder: replaces the embed launch).
// MAXS = compile-time bound on n_slab: the loads of all slabs are issued unconditionally (index clamped), nothing
// branches around a load.
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXS, bool EMBED>
__global__ __launch_bounds__(320) void layernorm_rows_kernel(const float* __restrict__ x_, const float* __restrict__ gamma_,
                                                             const float* __restrict__ beta_, T* __restrict__ out_, int d_,
                                                             LnPre pre) {
  constexpr int NWAVE = 5;  // always launched with 320 threads (d <= 1280): the cross-wave sums are five fixed reads, no loop
  __shared__ float red[2][8];
  // every kernel argument fetched in ONE batch at entry (common.hpp sgpr_pin)
  const float* x = sgpr_pin_ptr(x_); const float* gamma = sgpr_pin_ptr(gamma_); const float* beta = sgpr_pin_ptr(beta_);
  T* out = sgpr_pin_ptr(out_);
  const int d = sgpr_pin(d_);
  pre.bias = sgpr_pin_ptr(pre.bias); pre.slab = sgpr_pin_ptr(pre.slab); pre.n_slab = sgpr_pin(pre.n_slab);
  pre.slab_stride = sgpr_pin(pre.slab_stride); pre.x_out = sgpr_pin_ptr(pre.x_out);
  if constexpr (EMBED) { pre.tok = sgpr_pin_ptr(pre.tok); pre.step = sgpr_pin_ptr(pre.step); pre.emb = sgpr_pin_ptr(pre.emb); pre.pos = sgpr_pin_ptr(pre.pos); }
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = d >> 2, i = min(tid, nv - 1);
  const bool live = tid < nv;
  const float inv_d = __builtin_amdgcn_rcpf((float)d);  // computed while the loads fly (a division here is ten dependent instructions)
  float4 v;
  const float4 gm = ((const float4*)gamma)[i], bt = ((const float4*)beta)[i];
  if constexpr (EMBED) {
    const T* er = (const T*)pre.emb + (int64_t)pre.tok[row] * d;
    const T* pr = (const T*)pre.pos + (int64_t)(*pre.step) * d;
    if constexpr (sizeof(T) == 4) {
      const float4 a = ((const float4*)er)[i], b = ((const float4*)pr)[i];
      v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    } else {
      const uint2 a = ((const uint2*)er)[i], b = ((const uint2*)pr)[i];
      float4 fa, fb;
      N16<T>::up2(a.x, fa.x, fa.y); N16<T>::up2(a.y, fa.z, fa.w);
      N16<T>::up2(b.x, fb.x, fb.y); N16<T>::up2(b.y, fb.z, fb.w);
      v = make_float4(fa.x + fb.x, fa.y + fb.y, fa.z + fb.z, fa.w + fb.w);
    }
  } else {
    v = ((const float4*)(x + (int64_t)row * d))[i];
    if constexpr (MAXS > 0) {
      const float4 bs = ((const float4*)pre.bias)[i];
      float4 sl[MAXS];
#pragma unroll
      for (int s = 0; s < MAXS; ++s)
        sl[s] = ((const float4*)(pre.slab + (int64_t)min(s, pre.n_slab - 1) * pre.slab_stride + (int64_t)row * d))[i];
      __builtin_amdgcn_sched_barrier(0);  // every load above is issued before the first use below: ONE round trip
      v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
      // slab 0 always exists in this instantiation (n_slab >= 1): adding it unconditionally keeps its load from being sunk
      // behind a branch (= a second, serialised round trip)
#pragma unroll
      for (int s = 0; s < MAXS; ++s)
        if (s == 0 || s < pre.n_slab) { v.x += sl[s].x; v.y += sl[s].y; v.z += sl[s].z; v.w += sl[s].w; }
    }
  }
  if (live && (EMBED || MAXS > 0)) ((float4*)(pre.x_out + (int64_t)row * d))[i] = v;
  float s1 = live ? (v.x + v.y) + (v.z + v.w) : 0.f;
  s1 = wave_sum(s1);
  if (lane == 0) red[0][wave] = s1;
  __syncthreads();
  static_assert(NWAVE == 5, "fixed-order sum of the five wave partials");
  const float mean = (((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) + red[0][4]) * inv_d;
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
  float s2 = live ? (a * a + b * b) + (c * c + e * e) : 0.f;
  s2 = wave_sum(s2);
  if (lane == 0) red[1][wave] = s2;
  __syncthreads();
  const float rstd = rsqrtf((((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) + red[1][4]) * inv_d + 1e-5f);
  if (!live) return;
  const float r0 = a * rstd * gm.x + bt.x, r1 = b * rstd * gm.y + bt.y, r2 = c * rstd * gm.z + bt.z, r3 = e * rstd * gm.w + bt.w;
  T* o = out + (int64_t)row * d;
  if constexpr (sizeof(T) == 4) {
    ((float4*)o)[i] = make_float4(r0, r1, r2, r3);
  } else {
    uint2 p;
    p.x = N16<T>::pk(r0, r1);
    p.y = N16<T>::pk(r2, r3);
    ((uint2*)o)[i] = p;
  }
}
template <typename T>
void launch_layernorm_rows(const float* x, const float* gamma, const float* beta, T* out, int rows, int d, const LnPre& pre,
                           hipStream_t s) {
  if (d > 1280 || (d & 3)) { launch_fault("layernorm_rows needs d <= 1280, d %% 4 == 0 (got %d)", d); return; }
  if (pre.n_slab > 16) { launch_fault("layernorm_rows sums at most 16 slabs (got %d)", pre.n_slab); return; }
  dim3 grid(rows), block(320);  // five waves whatever d is (ttasr_create: d <= 1280): threads past d / 4 contribute zeros
  if (pre.tok) hipLaunchKernelGGL((layernorm_rows_kernel<T, 0, true>), grid, block, 0, s, x, gamma, beta, out, d, pre);
  else if (pre.n_slab == 0) hipLaunchKernelGGL((layernorm_rows_kernel<T, 0, false>), grid, block, 0, s, x, gamma, beta, out, d, pre);
  else if (pre.n_slab <= 2) hipLaunchKernelGGL((layernorm_rows_kernel<T, 2, false>), grid, block, 0, s, x, gamma, beta, out, d, pre);
  else if (pre.n_slab <= 4) hipLaunchKernelGGL((layernorm_rows_kernel<T, 4, false>), grid, block, 0, s, x, gamma, beta, out, d, pre);
  else if (pre.n_slab <= 8) hipLaunchKernelGGL((layernorm_rows_kernel<T, 8, false>), grid, block, 0, s, x, gamma, beta, out, d, pre);
  else hipLaunchKernelGGL((layernorm_rows_kernel<T, 16, false>), grid, block, 0, s, x, gamma, beta, out, d, pre);
}
template void launch_layernorm_rows<float>(const float*, const float*, const float*, float*, int, int, const LnPre&, hipStream_t);
template void launch_layernorm_rows<bf16_t>(const float*, const float*, const float*, bf16_t*, int, int, const LnPre&, hipStream_t);
template void launch_layernorm_rows<f16_t>(const float*, const float*, const float*, f16_t*, int, int, const LnPre&, hipStream_t);
template void launch_layernorm<float>(const float*, const float*, const float*, float*, int, int, hipStream_t);
template void launch_layernorm<bf16_t>(const float*, const float*, const float*, bf16_t*, int, int, hipStream_t);
template void launch_layernorm<f16_t>(const float*, const float*, const float*, f16_t*, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// Weight intake: source layout (f32 or bf16 bits, as the checkpoint / the RCCL broadcast delivers it) -> f32 in the
// engine's layout.  conv_in > 0: [out][in][3] -> [out][3][in]; `scale` folds the q pre-scaling (1/8: exact in bf16 too).
__global__ void prep_weight_kernel(const void* __restrict__ src, int src_type /*0 f32, 1 bf16 bits, 2 fp16 bits*/, float* __restrict__ dst,
                                   int64_t n, int64_t conv_in, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = i;
    if (conv_in > 0) {  // i = (o*3 + k)*I + c  <-  (o*I + c)*3 + k
      const int64_t c = i % conv_in, ok = i / conv_in, k = ok % 3, o = ok / 3;
      j = (o * conv_in + c) * 3 + k;
    }
    const float v = src_type == 1 ? bf2f(((const bf16_t*)src)[j]) : (src_type == 2 ? N16<f16_t>::up(((const uint16_t*)src)[j]) : ((const float*)src)[j]);
    dst[i] = v * scale;
  }
}
void launch_prep_weight(const void* src, int src_type, float* dst, int64_t n, int64_t conv_in, float scale, hipStream_t s) {
  int64_t nb = (n + 255) / 256; int blocks = (int)(nb < 8192 ? nb : 8192);
  hipLaunchKernelGGL(prep_weight_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, src, src_type, dst, n, conv_in, scale);
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = from_f<T>(in[i]);
}
template <typename T>
__global__ void uncast_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = to_f<T>(in[i]);
}
template <typename T> void launch_cast(const float* in, T* out, int64_t n, hipStream_t s) {
  int64_t nb = (n + 255) / 256; int blocks = (int)(nb < 4096 ? nb : 4096);
  hipLaunchKernelGGL(cast_kernel<T>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, in, out, n);
}
template <typename T> void launch_uncast(const T* in, float* out, int64_t n, hipStream_t s) {
  int64_t nb = (n + 255) / 256; int blocks = (int)(nb < 4096 ? nb : 4096);
  hipLaunchKernelGGL(uncast_kernel<T>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, in, out, n);
}
template void launch_cast<float>(const float*, float*, int64_t, hipStream_t);
template void launch_cast<bf16_t>(const float*, bf16_t*, int64_t, hipStream_t);
template void launch_cast<f16_t>(const float*, f16_t*, int64_t, hipStream_t);
template void launch_uncast<float>(const float*, float*, int64_t, hipStream_t);
template void launch_uncast<bf16_t>(const bf16_t*, float*, int64_t, hipStream_t);
template void launch_uncast<f16_t>(const f16_t*, float*, int64_t, hipStream_t);
