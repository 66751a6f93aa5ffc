#define TTASR_EXPERIMENTS 1   // QMODE 2 of the cross-attention kernel is compiled in lab builds only
// Focused check of self_attn_oproj_kernel against self_attn_decode + K-split out-proj GEMM on random data (one layer).
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_misc.hip"
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_skinny.hip"
#include "../../taiwan_tongues_asr_ce_amd/csrc/kernels_attn.hip"
#include "fused_attn_oproj.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
template <typename T16> void launch_cross_attn_flash_bf16(const T16*, const T16*, const T16*, T16*, int, int, int, int, hipStream_t) {}
template void launch_cross_attn_flash_bf16<bf16_t>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, hipStream_t);
template void launch_cross_attn_flash_bf16<f16_t>(const f16_t*, const f16_t*, const f16_t*, f16_t*, int, int, int, int, hipStream_t);
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static void* dmal(size_t bytes, int fill, float scale = 1.f) {
  void* p; hipMalloc(&p, bytes);
  if (fill == 1) { std::vector<uint16_t> h(bytes / 2); for (auto& v : h) v = f2b(scale * ((rand() & 0xffff) / 32768.f - 1.f)); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); }
  else if (fill == 2) { std::vector<float> h(bytes / 4); for (auto& v : h) v = scale * ((rand() & 0xffff) / 32768.f - 1.f); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); }
  else hipMemset(p, 0, bytes);
  return p;
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, d = 1280, H = 20, posv = argc > 2 ? atoi(argv[2]) : 64, G = argc > 3 ? atoi(argv[3]) : 4;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* bias = (float*)dmal(3 * d * 4, 2, 0.1f);
  float* slab = (float*)dmal((size_t)24 * 32 * 3 * d * 4, 0);
  float* sl_qkv = (float*)dmal((size_t)2 * B * 3 * d * 4, 2, 0.5f);
  bf16_t *qkv = (bf16_t*)dmal(B * 3 * d * 2, 1), *att = (bf16_t*)dmal(B * d * 2, 1), *wo = (bf16_t*)dmal((size_t)d * d * 2, 1, 0.03f);
  const int pps = 28; const size_t pool_bytes = (size_t)B * pps * 2 * H * 16 * 64 * 2;
  bf16_t* pool = (bf16_t*)dmal(pool_bytes, 1); bf16_t* pool2 = (bf16_t*)dmal(pool_bytes, 0);
  hipMemcpy(pool2, pool, pool_bytes, hipMemcpyDeviceToDevice);
  int32_t* pt = (int32_t*)dmal(B * pps * 4, 0); int32_t* step = (int32_t*)dmal(16, 0);
  hipMemcpy(step, &posv, 4, hipMemcpyHostToDevice);
  SlabIn sq; sq.slab = sl_qkv; sq.bias = bias; sq.n = 2; sq.stride = (int64_t)B * 3 * d; sq.ld = 3 * d;
  // reference
  launch_self_attn_decode<bf16_t>(qkv, pool, pt, pps, 0, 1, 0, step, att, B, H, s, sq);
  GemmEpi e; e.ldc = d;
  launch_gemm_skinny(wo, att, B, d, d, e, s, 4, slab, (int64_t)B * d);
  hipStreamSynchronize(s);
  std::vector<float> r((size_t)4 * B * d), ref((size_t)B * d, 0.f);
  hipMemcpy(r.data(), slab, r.size() * 4, hipMemcpyDeviceToHost);
  for (int k = 0; k < 4; ++k) for (size_t i = 0; i < ref.size(); ++i) ref[i] += r[(size_t)k * B * d + i];
  // fused
  float* oslab = slab + (size_t)4 * B * d;
  bool ok = launch_self_attn_oproj<bf16_t>(pool2, pt, pps, 0, 1, 0, step, B, H, sq, qkv, wo, oslab, (int64_t)B * d, G, s);
  hipStreamSynchronize(s);
  printf("launched %d err %s\n", (int)ok, hipGetErrorString(hipGetLastError()));
  std::vector<float> f((size_t)H * B * d), got((size_t)B * d, 0.f);
  hipMemcpy(f.data(), oslab, f.size() * 4, hipMemcpyDeviceToHost);
  for (int k = 0; k < H; ++k) for (size_t i = 0; i < got.size(); ++i) got[i] += f[(size_t)k * B * d + i];
  double md = 0, mx = 0; int bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) { double df = fabs(ref[i] - got[i]); md = std::max(md, df); mx = std::max(mx, (double)fabs(ref[i]));
    if (df > 0.02 && bad < 12) { printf("  row %zu col %zu ref %.4f got %.4f\n", i / d, i % d, ref[i], got[i]); ++bad; } }
  printf("B %d pos %d G %d: max|diff| %.4g  max|ref| %.4g\n", B, posv, G, md, mx);
  // per-head slab of head 0 vs the reference's head-0 contribution cannot be separated from the K-split slabs; compare the KV append
  std::vector<uint16_t> p1(pool_bytes / 2), p2(pool_bytes / 2);
  hipMemcpy(p1.data(), pool, pool_bytes, hipMemcpyDeviceToHost); hipMemcpy(p2.data(), pool2, pool_bytes, hipMemcpyDeviceToHost);
  size_t nd = 0; for (size_t i = 0; i < p1.size(); ++i) nd += p1[i] != p2[i];
  printf("KV pool words that differ after the append: %zu\n", nd);
  return 0;
}
