// What does a dependent kernel boundary cost in a hipGraph chain, and does the MEMORY TYPE of the handed-over
// activation buffers change it?  (round 2: the decode step is ~355 launches of ~5 us each, of which an empty kernel
// accounts for 1.6 us; the rest is the load round trip of data the previous kernel wrote plus cache maintenance.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench/chain_floor.hip -o tools/microbench/bin/chain_floor
// Chain of N kernels, each reading the 160 KB block its predecessor wrote (32 workgroups x 320 threads, float4 each:
// the shape of the decode LayerNorm) and writing the other buffer; buffers allocated as
//   coarse      hipMalloc (default: cached in L2, written back / invalidated at kernel boundaries)
//   finegrained hipExtMallocWithFlags(hipDeviceMallocFinegrained)
//   uncached    hipExtMallocWithFlags(hipDeviceMallocUncached)
// plus store / load flavours (plain, nontemporal, __hip_atomic relaxed agent = sc1) on coarse memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
using f32x4v = __attribute__((ext_vector_type(4))) float;

__global__ void k_empty() {}
template <int MODE>  // 0 plain, 1 nontemporal load+store, 2 sc1 (agent-scope relaxed atomics on dwords)
__global__ __launch_bounds__(320) void k_rows(const float4* __restrict__ in, float4* __restrict__ out) {
  const int i = blockIdx.x * 320 + threadIdx.x;
  float4 v;
  if (MODE == 1) { const f32x4v t = __builtin_nontemporal_load((const f32x4v*)(in + i)); v = make_float4(t.x, t.y, t.z, t.w); }
  else if (MODE == 2) {
    const float* p = (const float*)(in + i);
    v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else v = in[i];
  // a block reduction like LayerNorm's (keeps the kernel honest: the store depends on every load of the row)
  __shared__ float red[8];
  float s = (v.x + v.y) + (v.z + v.w);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < 5; ++w) t += red[w];
  v.x += t * 1e-9f;
  if (MODE == 1) { f32x4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (f32x4v*)(out + i)); }
  else if (MODE == 2) {
    float* p = (float*)(out + i);
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else out[i] = v;
}
// the same row kernel with three parameter vectors (gamma, beta, bias: 5 KB each) that are COLD: every launch reads them
// at a different offset of a 2 GiB buffer (the decode step streams ~10 GB between two uses of a LayerNorm's parameters)
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
// 160 workgroups x 256 threads each streaming 20 KB of COLD weights (3.3 MB per launch at a moving offset) + the hot 160 KB
// activation block, then storing 4 KB: the shape of a K-split decode GEMM without the MFMAs
template <bool NT>
__global__ __launch_bounds__(256) void k_gemm_like(const float4* __restrict__ in, float4* __restrict__ out,
                                                   const f32x4v* __restrict__ w) {
  f32x4v acc = {0, 0, 0, 0};
  const f32x4v* wp = w + (size_t)blockIdx.x * 1280 + threadIdx.x;
  f32x4v t[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) t[j] = NT ? __builtin_nontemporal_load(wp + 256 * j) : wp[256 * j];
  float4 xs[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) xs[j] = in[(threadIdx.x + 256 * j + blockIdx.x * 64) % 10240];
#pragma unroll
  for (int j = 0; j < 5; ++j) { acc += t[j]; acc.x += xs[j].x; }
  out[(blockIdx.x * 256 + threadIdx.x) % 10240] = make_float4(acc.x, acc.y, acc.z, acc.w);
}
// a 160-workgroup consumer that reads the whole 160 KB block (every workgroup, like a decode GEMM's activation operand)
__global__ __launch_bounds__(256) void k_fan(const float4* __restrict__ in, float4* __restrict__ out) {
  float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 10; ++j) { const float4 v = in[(threadIdx.x + 256 * j + blockIdx.x * 64) % 10240]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  if (blockIdx.x < 40) out[blockIdx.x * 256 + threadIdx.x] = a;
}

template <class F> double timeit(hipStream_t s, F f, int reps) {
  f(); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) f();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const size_t bytes = 10240 * 16;  // 32 rows x 1280 f32
  const int N = 200;
  auto graph_of = [&](auto body) { hipGraph_t g; hipGraphExec_t e; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) body(i); hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0); return e; };
  struct Mem { const char* name; unsigned flags; bool ext; } mems[] = {
    {"coarse (hipMalloc)", 0, false}, {"finegrained", hipDeviceMallocFinegrained, true}, {"uncached", hipDeviceMallocUncached, true}};
  {
    hipGraphExec_t e = graph_of([&](int) { hipLaunchKernelGGL(k_empty, dim3(32), dim3(320), 0, s); });
    printf("%-56s %.2f us/kernel\n", "empty kernel chain", timeit(s, [&] { hipGraphLaunch(e, s); }, 20) / N);
  }
  {
    float4 *a, *b; f32x4v* big;
    hipMalloc((void**)&a, bytes); hipMalloc((void**)&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    const size_t big_bytes = (size_t)2 << 30;
    hipMalloc((void**)&big, big_bytes); hipMemset(big, 0, big_bytes);
    auto run = [&](const char* what, auto body) {
      hipGraphExec_t e = graph_of(body);
      printf("%-72s %.2f us/kernel\n", what, timeit(s, [&] { hipGraphLaunch(e, s); }, 20) / N);
      hipGraphExecDestroy(e);
    };
    // launch i of replay r uses offset (i * 9 MB) within the 2 GiB buffer: 200 launches x 9 MB = 1.8 GB per replay > MALL
    auto cold = [&](int i) { return big + (size_t)i * (9u << 20) / 16; };
    run("rows + HOT params (same 15 KB every launch)", [&](int i) { hipLaunchKernelGGL(k_rows_cold, dim3(32), dim3(320), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, (const float4*)big); });
    run("rows + COLD params (15 KB at a new offset every launch)", [&](int i) { hipLaunchKernelGGL(k_rows_cold, dim3(32), dim3(320), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, (const float4*)cold(i)); });
    run("gemm-like 160 WG + HOT 3.3 MB weights", [&](int i) { hipLaunchKernelGGL(k_gemm_like<false>, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, big); });
    run("gemm-like 160 WG + COLD 3.3 MB weights", [&](int i) { hipLaunchKernelGGL(k_gemm_like<false>, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, cold(i)); });
    run("gemm-like 160 WG + COLD 3.3 MB weights, nontemporal", [&](int i) { hipLaunchKernelGGL(k_gemm_like<true>, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, cold(i)); });
    run("alternating: gemm-like COLD -> rows COLD params", [&](int i) {
      if (i & 1) hipLaunchKernelGGL(k_rows_cold, dim3(32), dim3(320), 0, s, b, a, (const float4*)cold(i));
      else hipLaunchKernelGGL(k_gemm_like<false>, dim3(160), dim3(256), 0, s, a, b, cold(i)); });
    hipFree(a); hipFree(b); hipFree(big);
  }
  for (auto& m : mems) {
    float4 *a = nullptr, *b = nullptr;
    hipError_t e1 = m.ext ? hipExtMallocWithFlags((void**)&a, bytes, m.flags) : hipMalloc((void**)&a, bytes);
    hipError_t e2 = m.ext ? hipExtMallocWithFlags((void**)&b, bytes, m.flags) : hipMalloc((void**)&b, bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) { printf("%s: allocation failed (%s)\n", m.name, hipGetErrorString(e1 != hipSuccess ? e1 : e2)); continue; }
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    char label[128];
    auto run = [&](const char* what, auto body) {
      hipGraphExec_t e = graph_of(body);
      snprintf(label, sizeof label, "%s: %s", m.name, what);
      printf("%-56s %.2f us/kernel\n", label, timeit(s, [&] { hipGraphLaunch(e, s); }, 20) / N);
      hipGraphExecDestroy(e);
    };
    run("rows plain (ping-pong)", [&](int i) { hipLaunchKernelGGL(k_rows<0>, dim3(32), dim3(320), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
    run("rows nontemporal", [&](int i) { hipLaunchKernelGGL(k_rows<1>, dim3(32), dim3(320), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
    run("rows sc1 (agent-scope relaxed atomics)", [&](int i) { hipLaunchKernelGGL(k_rows<2>, dim3(32), dim3(320), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
    run("rows -> 160-WG fan-in reader, alternating", [&](int i) {
      if (i & 1) hipLaunchKernelGGL(k_fan, dim3(160), dim3(256), 0, s, b, a);
      else hipLaunchKernelGGL(k_rows<0>, dim3(32), dim3(320), 0, s, a, b); });
    hipFree(a); hipFree(b);
  }
  return 0;
}
