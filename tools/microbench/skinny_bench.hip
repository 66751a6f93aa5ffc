#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_skinny.hip"
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
template <class F> double timeit(hipStream_t s, F f, int reps) {
  f(); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) f();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}
static void* dmal(size_t bytes, int fill_rand) {
  void* p; hipMalloc(&p, bytes);
  if (fill_rand) { std::vector<uint16_t> h(bytes / 2); for (auto& v : h) v = 0x3c00 + (rand() & 0x1ff); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); }
  else hipMemset(p, 0, bytes);
  return p;
}
int main() {
  const int B = 32, L = 16;
  hipStream_t s; hipStreamCreate(&s);
  struct Shape { const char* name; int N, K; int residual; int act; } shapes[] = {
    {"qkv  N3840 K1280", 3840, 1280, 0, 0}, {"out  N1280 K1280 res", 1280, 1280, 1, 0}, {"q    N1280 K1280", 1280, 1280, 0, 0},
    {"fc1  N5120 K1280 gelu", 5120, 1280, 0, 1}, {"fc2  N1280 K5120 res", 1280, 5120, 1, 0}};
  float* x = (float*)dmal(B * 5120 * 4, 0); float* bias = (float*)dmal(5120 * 4, 0);
  bf16_t* in = (bf16_t*)dmal(B * 5120 * 2, 1); bf16_t* out = (bf16_t*)dmal(B * 5120 * 2, 1);
  for (auto& sh : shapes) {
    std::vector<bf16_t*> w(L);
    for (auto& p : w) p = (bf16_t*)dmal((size_t)sh.N * sh.K * 2, 1);
    hipGraph_t gr; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 64; ++i) {
      GemmEpi e; e.bias = bias; e.ldc = sh.N; e.act = sh.act;
      if (sh.residual) { e.residual = x; e.out_f32 = x; } else e.out_t = out;
      launch_gemm_skinny(w[i % L], in, B, sh.N, sh.K, e, s);
    }
    hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    double us = timeit(s, [&] { hipGraphLaunch(ex, s); }, 10) / 64;
    double mb = (double)sh.N * sh.K * 2 / 1e6;
    printf("%-24s: %6.2f us  (%.1f MB -> %.2f TB/s)\n", sh.name, us, mb, mb / us / 1e6 * 1e6 / 1e6);
    for (auto& p : w) hipFree(p);
  }
  return 0;
}
