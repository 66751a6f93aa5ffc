// fp8 cross-attention decode kernel (kernels_fp8.hip): rows per batch sweep, B = 32, large-v3, 32 layers of cold e4m3 cross-KV
// (123 MB per launch), graph replays, interleaved rounds.  Checks every U against U = 3 (bit-identical: same accumulation order).
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_fp8.hip"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
thread_local bool g_kernel_sig_on = false; thread_local char g_kernel_sig[192] = ""; thread_local int g_xattn_variant = 3; thread_local int g_skinny_nt = 1;
static void* dmal(size_t bytes, int fill) {
  void* p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  if (fill) { std::vector<uint8_t> h(1 << 22); for (auto& v : h) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v &= 0xf7; }   // no NaN encodings
    for (size_t o = 0; o < bytes; o += h.size()) hipMemcpy((char*)p + o, h.data(), std::min(bytes - o, h.size()), hipMemcpyHostToDevice); }
  else hipMemset(p, 0, bytes);
  return p;
}
template <int U> void launch(const bf16_t* q, const uint8_t* K, const uint8_t* V, const float* ks, const float* vs, bf16_t* out, int B, int H, int T, SlabIn sq, hipStream_t s) {
  const size_t lds = sizeof(float) * (T + 4 * 64 + 8);
  hipLaunchKernelGGL((cross_attn_fp8_kernel<bf16_t, true, U>), dim3(H, B), dim3(256), lds, s, q, K, V, ks, vs, out, H, T, sq);
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, d = 1280, H = 20, T = 1500, L = 32;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  std::vector<float> hb(d, 0.01f), hs((size_t)4 * B * d), hsc(B * H, 0.002f);
  for (auto& v : hs) v = ((rand() & 0xffff) / 32768.f - 1.f) * 0.3f;
  float* bias = (float*)dmal(d * 4, 0); hipMemcpy(bias, hb.data(), d * 4, hipMemcpyHostToDevice);
  float* slab = (float*)dmal(hs.size() * 4, 0); hipMemcpy(slab, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  float* sc = (float*)dmal(B * H * 4, 0); hipMemcpy(sc, hsc.data(), B * H * 4, hipMemcpyHostToDevice);
  bf16_t *q = (bf16_t*)dmal(B * d * 2, 0), *att = (bf16_t*)dmal(B * d * 2, 0);
  std::vector<uint8_t*> xk(L), xv(L);
  for (int l = 0; l < L; ++l) { xk[l] = (uint8_t*)dmal((size_t)B * H * T * 64, 1); xv[l] = (uint8_t*)dmal((size_t)B * H * T * 64, 1); }
  SlabIn sq; sq.slab = slab; sq.bias = bias; sq.n = 4; sq.stride = (int64_t)B * d; sq.ld = d;
  std::vector<uint16_t> ref((size_t)B * d), got((size_t)B * d);
  auto run = [&](int U, int l) {
    switch (U) { case 2: launch<2>(q, xk[l], xv[l], sc, sc, att, B, H, T, sq, s); break; case 3: launch<3>(q, xk[l], xv[l], sc, sc, att, B, H, T, sq, s); break;
      case 4: launch<4>(q, xk[l], xv[l], sc, sc, att, B, H, T, sq, s); break; case 6: launch<6>(q, xk[l], xv[l], sc, sc, att, B, H, T, sq, s); break;
      default: launch<8>(q, xk[l], xv[l], sc, sc, att, B, H, T, sq, s); }
  };
  const int us_[] = {3, 2, 4, 6, 8};
  for (int U : us_) {
    hipMemset(att, 0, B * d * 2); run(U, 0); hipStreamSynchronize(s);
    hipMemcpy(U == 3 ? ref.data() : got.data(), att, ref.size() * 2, hipMemcpyDeviceToHost);
    if (U != 3) { size_t nd = 0; for (size_t i = 0; i < ref.size(); ++i) nd += ref[i] != got[i]; printf("U=%d: %zu of %zu outputs differ from U=3 (%s)\n", U, nd, ref.size(), hipGetErrorString(hipGetLastError())); }
  }
  hipGraphExec_t ex[16];
  for (int U : us_) { hipGraph_t gr; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); for (int l = 0; l < L; ++l) run(U, l);
    hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex[U], gr, nullptr, nullptr, 0); hipGraphDestroy(gr); }
  const double bytes = (double)B * (2.0 * T * d) + B * 2.0 * d * 2;
  for (int round = 0; round < 3; ++round)
    for (int U : us_) {
      hipGraphLaunch(ex[U], s); hipStreamSynchronize(s);
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int i = 0; i < 10; ++i) hipGraphLaunch(ex[U], s);
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 10 / L;
      printf("round %d U=%d: %.2f us per launch = %.3f TB/s\n", round, U, us, bytes / us / 1e6);
    }
  return 0;
}
