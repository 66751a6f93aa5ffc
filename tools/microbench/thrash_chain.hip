// Why do the small kernels of the real decode chain take ~5 us each when the same shapes replayed alone take ~2-3 us
// (chain_floor.hip)?  Hypothesis: between two uses of anything small (kernel code, kernel arguments, LayerNorm parameters,
// the activation rows) the step streams ~10 GB (cross-KV + weights) through L2 and the Infinity Cache, so every launch
// opens with several serialised misses to HBM.  This chain interleaves the small kernels with a 246 MB streaming kernel
// (the cross-attention's traffic) and is meant to be run under `rocprofv3 --kernel-trace --stats`: compare the average
// duration of k_rows_cold / k_gemm_like<..> between the variants (the variant number is a template argument so that the
// kernel names differ per variant).
//   VAR 0: no streaming kernel      1: plain streaming loads      2: nontemporal streaming loads
//   VAR 3: nontemporal streaming + nontemporal weight loads in the gemm-like kernel
// Build twice, without and with  -mllvm -amdgpu-kernarg-preload-count=8  (kernel arguments preloaded into SGPRs).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
using f32x4v = __attribute__((ext_vector_type(4))) float;

template <int VAR>
__global__ __launch_bounds__(320) void k_rows_cold(const float4* __restrict__ in, float4* __restrict__ out,
                                                   const float4* __restrict__ params) {
  const int i = blockIdx.x * 320 + threadIdx.x;
  float4 v = in[i];
  const float4 g = params[threadIdx.x], b = params[320 + threadIdx.x], c = params[640 + threadIdx.x];
  __shared__ float red[8];
  float s = (v.x + v.y) + (v.z + v.w) + g.x + b.y + c.z;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < 5; ++w) t += red[w];
  v.x += t * 1e-9f;
  out[i] = v;
}
template <int VAR>
__global__ __launch_bounds__(256) void k_gemm_like(const float4* __restrict__ in, float4* __restrict__ out,
                                                   const f32x4v* __restrict__ w) {
  f32x4v acc = {0, 0, 0, 0};
  const f32x4v* wp = w + (size_t)blockIdx.x * 1280 + threadIdx.x;
  f32x4v t[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) t[j] = VAR == 3 ? __builtin_nontemporal_load(wp + 256 * j) : wp[256 * j];
  float4 xs[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) xs[j] = in[(threadIdx.x + 256 * j + blockIdx.x * 64) % 10240];
#pragma unroll
  for (int j = 0; j < 5; ++j) { acc += t[j]; acc.x += xs[j].x; }
  out[(blockIdx.x * 256 + threadIdx.x) % 10240] = make_float4(acc.x, acc.y, acc.z, acc.w);
}
// 640 workgroups x 256 threads, 384 KB each (= one (row, head) of the cross-attention): 246 MB per launch
template <int VAR>
__global__ __launch_bounds__(256) void k_stream(const f32x4v* __restrict__ src, float4* __restrict__ out) {
  const f32x4v* p = src + (size_t)blockIdx.x * 24576 + threadIdx.x;
  f32x4v acc = {0, 0, 0, 0};
  for (int it = 0; it < 96; it += 8) {
    f32x4v t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = VAR >= 2 ? __builtin_nontemporal_load(p + (it + u) * 256) : p[(it + u) * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += t[u];
  }
  if (acc.x == 123.456f) out[threadIdx.x] = make_float4(acc.x, acc.y, acc.z, acc.w);
}

template <int VAR>
void run(hipStream_t s, float4* a, float4* b, f32x4v* big, size_t big_elems16) {
  const int LAYERS = 16;
  hipGraph_t g; hipGraphExec_t e;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  size_t off = 0;
  auto cold = [&](size_t bytes) { f32x4v* p = big + off; off += (bytes + (1 << 20)) / 16; if (off + (300u << 20) / 16 > big_elems16) off = 0; return p; };
  for (int l = 0; l < LAYERS; ++l) {
    // LN, qkv, (self-attn ~ gemm-like), out, LN, q, [xattn stream], out, LN, fc1, fc2  - as the real layer
    for (int k = 0; k < 3; ++k) {
      hipLaunchKernelGGL(k_rows_cold<VAR>, dim3(32), dim3(320), 0, s, a, b, (const float4*)cold(15 << 10));
      hipLaunchKernelGGL(k_gemm_like<VAR>, dim3(160), dim3(256), 0, s, b, a, cold(3300 << 10));
      if (k == 1 && VAR > 0) hipLaunchKernelGGL(k_stream<VAR>, dim3(640), dim3(256), 0, s, cold(246u << 20), b);
      hipLaunchKernelGGL(k_gemm_like<VAR>, dim3(160), dim3(256), 0, s, a, b, cold(3300 << 10));
    }
  }
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(e, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipGraphLaunch(e, s);
  hipStreamSynchronize(s);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
  printf("variant %d: %.1f us per replay of %d layers (9 small launches%s per layer) = %.2f us per layer\n", VAR, us, LAYERS,
         VAR > 0 ? " + one 246 MB stream" : "", us / LAYERS);
  hipGraphExecDestroy(e); hipGraphDestroy(g);
}

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float4 *a, *b; f32x4v* big;
  const size_t bytes = 10240 * 16, big_bytes = (size_t)12 << 30;
  hipMalloc((void**)&a, bytes); hipMalloc((void**)&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  if (hipMalloc((void**)&big, big_bytes) != hipSuccess) { printf("big alloc failed\n"); return 1; }
  hipMemset(big, 0, big_bytes);
  run<0>(s, a, b, big, big_bytes / 16);
  run<1>(s, a, b, big, big_bytes / 16);
  run<2>(s, a, b, big, big_bytes / 16);
  run<3>(s, a, b, big, big_bytes / 16);
  return 0;
}
