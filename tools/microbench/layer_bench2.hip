// Decode-layer chain with the REAL kernels in the round-2 launch plan (K-split GEMMs -> f32 slabs, per-row LayerNorm that
// sums them, attention kernels that sum the q / qkv slabs), timed un-profiled by wall clock over graph replays
// (rocprofv3's per-kernel durations are not trustworthy for 2-5 us kernels: it reports ~5 us for kernels whose chain runs at
// 2.8 us per launch).  32 layers of distinct cold weights and cross-KV, B = 32, position 64.
//   layer_bench2 <mask>   bit0 skip LayerNorms, bit1 skip GEMMs, bit2 skip self-attention, bit3 skip cross-attention
// Class cost in the chain = time(mask 0) - time(mask with the class skipped).
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_misc.hip"
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_skinny.hip"
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_attn.hip"
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
  void* p;
  if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc of %zu failed\n", bytes); exit(1); }
  if (fill_rand) { std::vector<uint16_t> h(1 << 20); for (auto& v : h) v = 0x3c00 + (rand() & 0x1ff);
    for (size_t o = 0; o < bytes; o += h.size() * 2) hipMemcpy((char*)p + o, h.data(), std::min(bytes - o, h.size() * 2), hipMemcpyHostToDevice); }
  else hipMemset(p, 0, bytes);
  return p;
}
int main(int argc, char** argv) {
  const int B = 32, d = 1280, F = 5120, H = 20, T = 1500, L = 32;
  const int ks_d = argc > 2 ? atoi(argv[2]) : 4, ks_q = argc > 3 ? atoi(argv[3]) : 4, ks_qkv = argc > 4 ? atoi(argv[4]) : 4, ks_f = argc > 5 ? atoi(argv[5]) : 8;
  g_xattn_variant = 1;
  g_skinny_nt = getenv("TTASR_W_NT") != nullptr;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* x = (float*)dmal(B * d * 4, 0); float* g = (float*)dmal(d * 4, 0); float* bt = (float*)dmal(d * 4, 0);
  float* bias = (float*)dmal(F * 4, 0); float* slab = (float*)dmal((size_t)16 * B * 3 * d * 4, 0);
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
  for (int mask : {0, 1, 2, 4, 8, 7, 13, 14, 11}) {
    if (argc > 1 && atoi(argv[1]) >= 0 && mask != atoi(argv[1])) continue;
    auto split = [&](const bf16_t* W, const bf16_t* A, int N, int K, int ks, const float* b) {
      SlabIn si;
      if (mask & 2) return si;
      GemmEpi e; e.ldc = N;
      if (ks > 1) { launch_gemm_skinny(W, A, B, N, K, e, s, ks, slab, (int64_t)B * N); si.slab = slab; si.bias = b; si.n = ks; si.stride = (int64_t)B * N; si.ld = N; }
      return si;
    };
    int pend = 0;
    auto ln = [&]() {
      if (mask & 1) { pend = 0; return; }
      LnPre pre; pre.x_out = x;
      if (pend) { pre.bias = bias; pre.slab = slab; pre.n_slab = pend; pre.slab_stride = (int64_t)B * d; }
      launch_layernorm_rows<bf16_t>(x, g, bt, h, B, d, pre, s);
      pend = 0;
    };
    auto layer = [&](int l) {
      ln();
      SlabIn sqkv = split(wqkv[l], h, 3 * d, d, ks_qkv, bias);
      if (!(mask & 2) && !sqkv.n) { GemmEpi e; e.bias = bias; e.out_t = qkv; e.ldc = 3 * d; launch_gemm_skinny(wqkv[l], h, B, 3 * d, d, e, s); }
      if (!(mask & 4)) launch_self_attn_decode<bf16_t>(qkv, pool, pt, pps, 0, 1, 0, step, att, B, H, s, sqkv);
      pend = split(wo[l], att, d, d, ks_d, bias).n;
      ln();
      SlabIn sq = split(wq[l], h, d, d, ks_q, bias);
      if (!(mask & 2) && !sq.n) { GemmEpi e; e.bias = bias; e.out_t = q; e.ldc = d; launch_gemm_skinny(wq[l], h, B, d, d, e, s); }
      if (!(mask & 8)) launch_cross_attn_decode<bf16_t>(q, xk[l], xv[l], att, B, H, T, 1, s, nullptr, sq);
      pend = split(wox[l], att, d, d, ks_d, bias).n;
      ln();
      if (!(mask & 2)) { GemmEpi e; e.bias = bias; e.act = 1; e.out_t = mid; e.ldc = F; launch_gemm_skinny(w1[l], h, B, F, d, e, s); }
      pend = split(w2[l], mid, d, F, ks_f, bias).n;
    };
    hipGraph_t gr; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int l = 0; l < L; ++l) layer(l);
    hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    double us = timeit(s, [&] { hipGraphLaunch(ex, s); }, 10) / L;
    printf("mask=%2d (skip%s%s%s%s) ks=%d,%d,%d,%d: %.2f us per layer\n", mask, mask & 1 ? " LN" : "", mask & 2 ? " GEMM" : "",
           mask & 4 ? " self" : "", mask & 8 ? " xattn" : "", ks_d, ks_q, ks_qkv, ks_f, us);
    hipGraphExecDestroy(ex); hipGraphDestroy(gr);
  }
  return 0;
}
