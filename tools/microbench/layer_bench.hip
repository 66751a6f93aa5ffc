// Decode-layer chain microbenchmark using the real kernels (timing only; data is random).
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_misc.hip"
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_skinny.hip"
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_attn.hip"
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
int main(int argc, char** argv) {
  const int Bfull = 32, d = 1280, F = 5120, H = 20, T = 1500, L = 8;
  int mask = argc > 1 ? atoi(argv[1]) : 0;
  const int dual = argc > 2 ? atoi(argv[2]) : 0;
  const int B = dual ? 16 : 32;
  hipStream_t s2; hipStreamCreate(&s2); hipEvent_t ef, ej; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);  // bit0: skip LN, 1: skip GEMMs, 2: skip self-attn, 3: skip xattn
  hipStream_t s; hipStreamCreate(&s);
  float* x = (float*)dmal(Bfull * d * 4, 0); float* g = (float*)dmal(d * 4, 0); float* bt = (float*)dmal(d * 4, 0);
  float* bias = (float*)dmal(F * 4, 0);
  bf16_t *h = (bf16_t*)dmal(B * d * 2, 1), *qkv = (bf16_t*)dmal(B * 3 * d * 2, 1), *att = (bf16_t*)dmal(B * d * 2, 1),
         *q = (bf16_t*)dmal(B * d * 2, 1), *mid = (bf16_t*)dmal(B * F * 2, 1);
  std::vector<bf16_t*> wqkv(L), wo(L), wq(L), wox(L), w1(L), w2(L), xk(L), xv(L);
  for (int l = 0; l < L; ++l) {
    wqkv[l] = (bf16_t*)dmal((size_t)3 * d * d * 2, 1); wo[l] = (bf16_t*)dmal((size_t)d * d * 2, 1); wq[l] = (bf16_t*)dmal((size_t)d * d * 2, 1);
    wox[l] = (bf16_t*)dmal((size_t)d * d * 2, 1); w1[l] = (bf16_t*)dmal((size_t)F * d * 2, 1); w2[l] = (bf16_t*)dmal((size_t)F * d * 2, 1);
    xk[l] = (bf16_t*)dmal((size_t)B * H * T * 64 * 2, 1); xv[l] = (bf16_t*)dmal((size_t)B * H * T * 64 * 2, 1);
  }
  const int pps = 28; bf16_t* pool = (bf16_t*)dmal((size_t)B * pps * 2 * H * 16 * 64 * 2, 1);
  int32_t* pt = (int32_t*)dmal(B * pps * 4, 0); int32_t* step = (int32_t*)dmal(16, 0);
  { int v = 64; hipMemcpy(step, &v, 4, hipMemcpyHostToDevice); }
  auto layer = [&](int l, hipStream_t s, int r0) {
    float* x_ = x + r0 * d; (void)x_;
    GemmEpi e;
    if (!(mask & 1)) launch_layernorm<bf16_t>(x, g, bt, h, B, d, s);
    if (!(mask & 2)) { e = GemmEpi(); e.bias = bias; e.out_t = qkv; e.ldc = 3 * d; launch_gemm_skinny(wqkv[l], h, B, 3 * d, d, e, s); }
    if (!(mask & 4)) launch_self_attn_decode<bf16_t>(qkv, pool, pt, pps, 0, 1, 0, step, att, B, H, s);
    if (!(mask & 2)) { e = GemmEpi(); e.bias = bias; e.residual = x; e.out_f32 = x; e.ldc = d; launch_gemm_skinny(wo[l], att, B, d, d, e, s); }
    if (!(mask & 1)) launch_layernorm<bf16_t>(x, g, bt, h, B, d, s);
    if (!(mask & 2)) { e = GemmEpi(); e.bias = bias; e.out_t = q; e.ldc = d; launch_gemm_skinny(wq[l], h, B, d, d, e, s); }
    if (!(mask & 8)) launch_cross_attn_decode<bf16_t>(q, xk[l], xv[l], att, B, H, T, 1, s);
    if (!(mask & 2)) { e = GemmEpi(); e.bias = bias; e.residual = x; e.out_f32 = x; e.ldc = d; launch_gemm_skinny(wox[l], att, B, d, d, e, s); }
    if (!(mask & 1)) launch_layernorm<bf16_t>(x, g, bt, h, B, d, s);
    if (!(mask & 2)) { e = GemmEpi(); e.bias = bias; e.act = 1; e.out_t = mid; e.ldc = F; launch_gemm_skinny(w1[l], h, B, F, d, e, s); }
    if (!(mask & 2)) { e = GemmEpi(); e.bias = bias; e.residual = x; e.out_f32 = x; e.ldc = d; launch_gemm_skinny(w2[l], mid, B, d, F, e, s); }
  };
  hipGraph_t gr; hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (!dual) { for (int rep = 0; rep < 4; ++rep) for (int l = 0; l < L; ++l) layer(l, s, 0); }
  else {
    hipEventRecord(ef, s); hipStreamWaitEvent(s2, ef, 0);
    for (int rep = 0; rep < 4; ++rep) for (int l = 0; l < L; ++l) layer(l, s, 0);
    for (int rep = 0; rep < 4; ++rep) for (int l = 0; l < L; ++l) layer(l, s2, 16);
    hipEventRecord(ej, s2); hipStreamWaitEvent(s, ej, 0);
  }
  hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
  double us = timeit(s, [&] { hipGraphLaunch(ex, s); }, 10) / (4 * L);
  printf("mask=%d dual=%d: %.2f us per layer (full batch)\n", mask, dual, us);
  return 0;
}
