// Cross-attention decode kernel variants over 32 layers of cold cross-KV (B = 32, large-v3: 246 MB per launch), graph replays.
//   g_xattn_variant: bit0 nontemporal loads, bit1 software-pipelined form (the U / row-mapping sweep of round 4 edited the
//   template argument in launch_cross_attn_decode: results in profiles/r4_xattn_pipeline.txt)
// Checks that every variant's output is bit-identical to variant 1 (the shipped kernel), then interleaved timing rounds.
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_attn.hip"
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <typename T16> void launch_cross_attn_flash_bf16(const T16*, const T16*, const T16*, T16*, int, int, int, int, hipStream_t) {}
template void launch_cross_attn_flash_bf16<bf16_t>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, hipStream_t);
template void launch_cross_attn_flash_bf16<f16_t>(const f16_t*, const f16_t*, const f16_t*, f16_t*, int, int, int, int, hipStream_t);
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static void* dmal(size_t bytes, int fill, float scale = 1.f) {
  void* p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  if (fill == 1) { std::vector<uint16_t> h(1 << 20); for (auto& v : h) v = f2b(scale * ((rand() & 0xffff) / 32768.f - 1.f));
    for (size_t o = 0; o < bytes; o += h.size() * 2) hipMemcpy((char*)p + o, h.data(), std::min(bytes - o, h.size() * 2), hipMemcpyHostToDevice); }
  else if (fill == 2) { std::vector<float> h(bytes / 4); for (auto& v : h) v = scale * ((rand() & 0xffff) / 32768.f - 1.f); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); }
  else hipMemset(p, 0, bytes);
  return p;
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, d = 1280, H = 20, T = argc > 2 ? atoi(argv[2]) : 1500, L = 32;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* bias = (float*)dmal(d * 4, 2, 0.1f);
  float* slab = (float*)dmal((size_t)4 * B * d * 4, 2, 0.3f);
  bf16_t *q = (bf16_t*)dmal(B * d * 2, 1), *att = (bf16_t*)dmal(B * d * 2, 0);
  std::vector<bf16_t*> xk(L), xv(L);
  for (int l = 0; l < L; ++l) { xk[l] = (bf16_t*)dmal((size_t)B * H * T * 64 * 2, 1); xv[l] = (bf16_t*)dmal((size_t)B * H * T * 64 * 2, 1); }
  SlabIn sq; sq.slab = slab; sq.bias = bias; sq.n = 4; sq.stride = (int64_t)B * d; sq.ld = d;
  const int variants[] = {1, 3, 0, 2};
  std::vector<uint16_t> ref((size_t)B * d), got((size_t)B * d);
  for (int v : variants) {
    g_xattn_variant = v;
    hipMemset(att, 0, B * d * 2);
    launch_cross_attn_decode<bf16_t>(q, xk[0], xv[0], att, B, H, T, 1, s, nullptr, sq);
    hipStreamSynchronize(s);
    hipMemcpy(v == 1 ? ref.data() : got.data(), att, ref.size() * 2, hipMemcpyDeviceToHost);
    if (v != 1) { size_t nd = 0; for (size_t i = 0; i < ref.size(); ++i) nd += ref[i] != got[i]; printf("variant %d: %zu of %zu outputs differ from variant 1 (%s)\n", v, nd, ref.size(), hipGetErrorString(hipGetLastError())); }
  }
  hipGraphExec_t ex[32];
  for (int v : variants) {
    g_xattn_variant = v;
    hipGraph_t gr;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int l = 0; l < L; ++l) launch_cross_attn_decode<bf16_t>(q, xk[l], xv[l], att, B, H, T, 1, s, nullptr, sq);
    hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex[v], gr, nullptr, nullptr, 0); hipGraphDestroy(gr);
  }
  const double bytes = (double)B * (2.0 * T * d + 2 * d) * 2;
  for (int round = 0; round < 4; ++round)
    for (int v : variants) {
      hipGraphLaunch(ex[v], s); hipStreamSynchronize(s);
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int i = 0; i < 10; ++i) hipGraphLaunch(ex[v], s);
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 10 / L;
      printf("round %d variant %d (%s%s): %.2f us per launch = %.3f TB/s\n", round, v, v & 1 ? "nt" : "plain", v & 2 ? ", pipelined 3 rows per batch" : ", 8 rows per batch", us, bytes / us / 1e6);
    }
  return 0;
}
