#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#pragma clang diagnostic ignored "-Wunused-value"
__global__ void k_copy(const float* x, float* y, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = x[i] + 1.f; }
__global__ void k_copy4(const float4* x, float4* y, int n4) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n4) { float4 v = x[i]; v.x += 1.f; y[i] = v; } }
__global__ void k_row(const float* x, float* y, int d) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0; for (int i = lane; i < d; i += 64) s += x[row * d + i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  for (int i = lane; i < d; i += 64) y[row * d + i] = x[row * d + i] - s;
}
__global__ void k_row4(const float4* x, float4* y, int d4) {  // wave per row, 5 float4 per lane in registers
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float4 v[5]; float s = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) { v[j] = x[row * d4 + lane + 64 * j]; s += v[j].x + v[j].y + v[j].z + v[j].w; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
  for (int j = 0; j < 5; ++j) { v[j].x -= s; y[row * d4 + lane + 64 * j] = v[j]; }
}
__global__ void k_atomic(float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) atomicAdd(&y[i], 1.f); }
template <class F> double timeit(hipStream_t s, F f, int reps) {
  f(); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) f();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  float *x, *y, *big; const int n = 32 * 1280; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
  hipMalloc(&big, 512 << 20);
  const int N = 200;
  auto graph_of = [&](auto body) { hipGraph_t g; hipGraphExec_t e; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) body(i); hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0); return e; };
  auto pp = [&](int i, auto f) { if (i & 1) f(y, x); else f(x, y); };
  struct { const char* name; hipGraphExec_t e; } tests[] = {
    {"copy scalar 8x256 (ping-pong)", graph_of([&](int i) { pp(i, [&](float* a, float* b) { hipLaunchKernelGGL(k_copy, dim3(8), dim3(256), 0, s, a, b, n); }); })},
    {"copy scalar 160x256          ", graph_of([&](int i) { pp(i, [&](float* a, float* b) { hipLaunchKernelGGL(k_copy, dim3(160), dim3(256), 0, s, a, b, n); }); })},
    {"copy float4 40x256           ", graph_of([&](int i) { pp(i, [&](float* a, float* b) { hipLaunchKernelGGL(k_copy4, dim3(40), dim3(256), 0, s, (const float4*)a, (float4*)b, n / 4); }); })},
    {"row scalar 8x256             ", graph_of([&](int i) { pp(i, [&](float* a, float* b) { hipLaunchKernelGGL(k_row, dim3(8), dim3(256), 0, s, a, b, 1280); }); })},
    {"row float4-in-regs 8x256     ", graph_of([&](int i) { pp(i, [&](float* a, float* b) { hipLaunchKernelGGL(k_row4, dim3(8), dim3(256), 0, s, (const float4*)a, (float4*)b, 320); }); })},
    {"atomicAdd 160x256            ", graph_of([&](int i) { hipLaunchKernelGGL(k_atomic, dim3(160), dim3(256), 0, s, y, n); })},
    {"row4 then atomic alternating ", graph_of([&](int i) { if (i & 1) hipLaunchKernelGGL(k_atomic, dim3(160), dim3(256), 0, s, y, n); else hipLaunchKernelGGL(k_row4, dim3(8), dim3(256), 0, s, (const float4*)y, (float4*)x, 320); })},
  };
  for (auto& t : tests) printf("%s: %.2f us/kernel\n", t.name, timeit(s, [&] { hipGraphLaunch(t.e, s); }, 20) / N);
  return 0;
}
