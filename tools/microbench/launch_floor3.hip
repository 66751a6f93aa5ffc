#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_misc.hip"
#include <cstdio>
#include <chrono>
#pragma clang diagnostic ignored "-Wunused-value"
__global__ void k_atomic(float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) atomicAdd(&y[i], 1.f); }
__global__ void k_read_h(const bf16_t* h, float* out, int n) {  // 240 WGs each reading all of h (like the GEMM's B operand)
  float s = 0; for (int i = threadIdx.x; i < n / 8; i += blockDim.x) { uint4 v = ((const uint4*)h)[i]; s += __uint_as_float(v.x << 16); }
  if (s == 12345.f) out[0] = s;
}
template <class F> double timeit(hipStream_t s, F f, int reps) {
  f(); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) f();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  const int n = 32 * 1280;
  float *x, *g, *b, *o; bf16_t* h; hipMalloc(&x, n * 4); hipMalloc(&g, 1280 * 4); hipMalloc(&b, 1280 * 4); hipMalloc(&h, n * 2); hipMalloc(&o, 64);
  hipMemset(x, 0, n * 4); hipMemset(g, 0, 5120); hipMemset(b, 0, 5120);
  const int N = 200;
  auto graph_of = [&](auto body) { hipGraph_t gr; hipGraphExec_t e; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) body(i); hipStreamEndCapture(s, &gr); hipGraphInstantiate(&e, gr, nullptr, nullptr, 0); return e; };
  struct { const char* name; hipGraphExec_t e; int per; } tests[] = {
    {"LN only                 ", graph_of([&](int i) { launch_layernorm<bf16_t>(x, g, b, h, 32, 1280, s); }), 1},
    {"atomic only             ", graph_of([&](int i) { hipLaunchKernelGGL(k_atomic, dim3(160), dim3(256), 0, s, x, n); }), 1},
    {"read_h only 240 WG      ", graph_of([&](int i) { hipLaunchKernelGGL(k_read_h, dim3(240), dim3(256), 0, s, h, o, n); }), 1},
    {"atomic -> LN            ", graph_of([&](int i) { hipLaunchKernelGGL(k_atomic, dim3(160), dim3(256), 0, s, x, n); launch_layernorm<bf16_t>(x, g, b, h, 32, 1280, s); }), 2},
    {"LN -> read_h            ", graph_of([&](int i) { launch_layernorm<bf16_t>(x, g, b, h, 32, 1280, s); hipLaunchKernelGGL(k_read_h, dim3(240), dim3(256), 0, s, h, o, n); }), 2},
    {"atomic -> LN -> read_h  ", graph_of([&](int i) { hipLaunchKernelGGL(k_atomic, dim3(160), dim3(256), 0, s, x, n); launch_layernorm<bf16_t>(x, g, b, h, 32, 1280, s); hipLaunchKernelGGL(k_read_h, dim3(240), dim3(256), 0, s, h, o, n); }), 3},
  };
  for (auto& t : tests) printf("%s: %.2f us per iteration (%d kernels)\n", t.name, timeit(s, [&] { hipGraphLaunch(t.e, s); }, 20) / N, t.per);
  return 0;
}
