#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty(int* p) { if (threadIdx.x == 9999) *p = 1; }
__global__ void k_inc(int* p) { *p += 1; }
__global__ void k_row(const float* x, float* y, int d) {  // LN-like: one wave per row
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0; for (int i = lane; i < d; i += 64) s += x[row * d + i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  for (int i = lane; i < d; i += 64) y[row * d + i] = x[row * d + i] - s;
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
  int* p; hipMalloc(&p, 64); hipMemset(p, 0, 64);
  float *x, *y; hipMalloc(&x, 32 * 1280 * 4); hipMalloc(&y, 32 * 1280 * 4); hipMemset(x, 0, 32 * 1280 * 4);
  const int N = 300;
  auto graph_of = [&](auto body) { hipGraph_t g; hipGraphExec_t e; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) body(); hipStreamEndCapture(s, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0); return e; };
  auto e1 = graph_of([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); });
  auto e2 = graph_of([&] { hipLaunchKernelGGL(k_inc, dim3(1), dim3(1), 0, s, p); });
  auto e3 = graph_of([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); });
  auto e4 = graph_of([&] { hipLaunchKernelGGL(k_row, dim3(8), dim3(256), 0, s, x, y, 1280); hipLaunchKernelGGL(k_row, dim3(8), dim3(256), 0, s, y, x, 1280); });
  printf("graph empty 1x64      : %.2f us/kernel\n", timeit(s, [&] { hipGraphLaunch(e1, s); }, 20) / N);
  printf("graph inc (dep load)  : %.2f us/kernel\n", timeit(s, [&] { hipGraphLaunch(e2, s); }, 20) / N);
  printf("graph empty 256x256   : %.2f us/kernel\n", timeit(s, [&] { hipGraphLaunch(e3, s); }, 20) / N);
  printf("graph LN-like 8x256   : %.2f us/kernel\n", timeit(s, [&] { hipGraphLaunch(e4, s); }, 20) / (2 * N));
  printf("stream empty 1x64     : %.2f us/kernel\n", timeit(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }, 5) / N);
  printf("stream LN-like        : %.2f us/kernel\n", timeit(s, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_row, dim3(8), dim3(256), 0, s, x, y, 1280); }, 5) / N);
  return 0;
}
