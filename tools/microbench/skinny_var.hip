// variants of the weight-streaming loop to find what limits it (timing only)
#include "../../taiwan_tongues_asr_ce_amd/csrc/common.hpp"
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
// MODE 0: nontemporal W + x loads + mfma; 1: plain W loads; 2: no x loads (x in regs const); 3: W loads only, xor-reduce (pure stream)
template <int MODE, int NW, int STEPS>
__global__ __launch_bounds__(NW * 64) void k(const bf16_t* __restrict__ Wsh, const bf16_t* __restrict__ x, float* __restrict__ out, int K) {
  __shared__ float red[NW][2][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nb = blockIdx.x;
  const int kb_per = K / 32, kb0 = wave * STEPS;
  const u32x4* wp = (const u32x4*)Wsh + ((int64_t)nb * kb_per + kb0) * 64 + lane;
  const bf16_t* xp[2];
  for (int bb = 0; bb < 2; ++bb) xp[bb] = x + (int64_t)(bb * 16 + (lane & 15)) * K + kb0 * 32 + 8 * (lane >> 4);
  u32x4 w[STEPS], xv[STEPS][2];
  if (MODE == 4 || MODE == 5) {
#pragma unroll
    for (int u = 0; u < STEPS; ++u) for (int bb = 0; bb < 2; ++bb) xv[u][bb] = (MODE == 5 && bb == 1) ? u32x4{1u,2u,3u,(unsigned)lane} : *(const u32x4*)(xp[bb] + u * 32);
  }
#pragma unroll
  for (int u = 0; u < STEPS; ++u) w[u] = (MODE == 1 || MODE >= 4) ? wp[u * 64] : __builtin_nontemporal_load(wp + u * 64);
  if (MODE >= 4) {
  } else if (MODE <= 1) {
#pragma unroll
    for (int u = 0; u < STEPS; ++u) for (int bb = 0; bb < 2; ++bb) xv[u][bb] = *(const u32x4*)(xp[bb] + u * 32);
  } else {
#pragma unroll
    for (int u = 0; u < STEPS; ++u) for (int bb = 0; bb < 2; ++bb) xv[u][bb] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};
  }
  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  if (MODE == 3) {
    unsigned a = 0;
#pragma unroll
    for (int u = 0; u < STEPS; ++u) a ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    if (a == 0x12345u) out[tid] = 1.f;
    return;
  }
#pragma unroll
  for (int u = 0; u < STEPS; ++u) for (int bb = 0; bb < 2; ++bb)
    acc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(s16x8*)&w[u], *(s16x8*)&xv[u][bb], acc[bb], 0, 0, 0);
  for (int bb = 0; bb < 2; ++bb) for (int r = 0; r < 4; ++r) red[wave][bb][(lane & 15) * 16 + (lane >> 4) * 4 + r] = acc[bb][r];
  __syncthreads();
  if (tid >= 256) return;
  for (int bb = 0; bb < 2; ++bb) { float v = 0; for (int w2 = 0; w2 < NW; ++w2) v += red[w2][bb][tid];
    ((bf16_t*)out)[(int64_t)(bb * 16 + (tid >> 4)) * 5120 + nb * 16 + (tid & 15)] = f2bf(v); }
}
template <class F> double timeit(hipStream_t s, F f, int reps) {
  f(); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; ++i) f();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}
static void* dmal(size_t bytes) { void* p; hipMalloc(&p, bytes); std::vector<uint16_t> h(bytes / 2); for (auto& v : h) v = 0x3c00 + (rand() & 0x1ff); hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice); return p; }
int main() {
  const int L = 16, N = 5120, K = 1280;
  hipStream_t s; hipStreamCreate(&s);
  bf16_t* in = (bf16_t*)dmal(32 * 5120 * 2); float* out = (float*)dmal(32 * 5120 * 4);
  std::vector<bf16_t*> w(L); for (auto& p : w) p = (bf16_t*)dmal((size_t)N * K * 2);
  auto run = [&](const char* name, auto launch) {
    hipGraph_t gr; hipGraphExec_t ex; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 64; ++i) launch(w[i % L]);
    hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    double us = timeit(s, [&] { hipGraphLaunch(ex, s); }, 10) / 64;
    printf("%-44s: %6.2f us  %.2f TB/s\n", name, us, (double)N * K * 2 / us / 1e6);
  };
  run("fc1 4 waves x10 steps, nt, x, mfma", [&](bf16_t* W) { hipLaunchKernelGGL((k<0, 4, 10>), dim3(N / 16), dim3(256), 0, s, W, in, out, K); });
  run("fc1 4 waves x10 steps, plain loads", [&](bf16_t* W) { hipLaunchKernelGGL((k<1, 4, 10>), dim3(N / 16), dim3(256), 0, s, W, in, out, K); });
  run("fc1 4 waves x10, no x loads", [&](bf16_t* W) { hipLaunchKernelGGL((k<2, 4, 10>), dim3(N / 16), dim3(256), 0, s, W, in, out, K); });
  run("fc1 4 waves x10, W stream only", [&](bf16_t* W) { hipLaunchKernelGGL((k<3, 4, 10>), dim3(N / 16), dim3(256), 0, s, W, in, out, K); });
  run("fc1 4 waves x10, x loads FIRST, plain", [&](bf16_t* W) { hipLaunchKernelGGL((k<4, 4, 10>), dim3(N / 16), dim3(256), 0, s, W, in, out, K); });
  run("fc1 4 waves x10, half x traffic, plain", [&](bf16_t* W) { hipLaunchKernelGGL((k<5, 4, 10>), dim3(N / 16), dim3(256), 0, s, W, in, out, K); });
  run("fc1 8 waves x5, x first, plain", [&](bf16_t* W) { hipLaunchKernelGGL((k<4, 8, 5>), dim3(N / 16), dim3(512), 0, s, W, in, out, K); });
  run("fc1 8 waves x5, nt, x, mfma", [&](bf16_t* W) { hipLaunchKernelGGL((k<0, 8, 5>), dim3(N / 16), dim3(512), 0, s, W, in, out, K); });
  run("fc1 8 waves x5, W stream only", [&](bf16_t* W) { hipLaunchKernelGGL((k<3, 8, 5>), dim3(N / 16), dim3(512), 0, s, W, in, out, K); });
  run("fc1 2 waves x20, W stream only", [&](bf16_t* W) { hipLaunchKernelGGL((k<3, 2, 20>), dim3(N / 16), dim3(128), 0, s, W, in, out, K); });
  return 0;
}
