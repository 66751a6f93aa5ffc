#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "common.hpp"
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x];
  out[0 * 64 + threadIdx.x] = wave_sum(v);
  out[1 * 64 + threadIdx.x] = wave_max(v);
  out[2 * 64 + threadIdx.x] = group_reduce<8>(v, OpSum{});
  out[3 * 64 + threadIdx.x] = group_reduce<16>(v, OpSum{});
  out[4 * 64 + threadIdx.x] = stride_reduce<8>(v, OpSum{});
  out[5 * 64 + threadIdx.x] = stride_reduce<16>(v, OpSum{});
  out[6 * 64 + threadIdx.x] = xor32_reduce(v, OpMax{});
  out[7 * 64 + threadIdx.x] = group_reduce<32>(v, OpMax{});
}
int main() {
  std::vector<float> h(64), o(8 * 64);
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 101) - 50.f + 0.25f * i;
  float *d, *r; hipMalloc(&d, 256); hipMalloc(&r, 8 * 256);
  hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, r);
  hipMemcpy(o.data(), r, 8 * 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    double s = 0, m = -1e30, g8 = 0, g16 = 0, s8 = 0, s16 = 0, m32 = -1e30;
    for (int j = 0; j < 64; ++j) { s += h[j]; m = fmax(m, h[j]); if (j / 8 == l / 8) g8 += h[j]; if (j / 16 == l / 16) g16 += h[j];
      if (j % 8 == l % 8) s8 += h[j]; if (j % 16 == l % 16) s16 += h[j]; if (j / 32 == l / 32) m32 = fmax(m32, h[j]); }
    double x32 = fmax(h[l], h[l ^ 32]);
    double want[8] = {s, m, g8, g16, s8, s16, x32, m32};
    for (int t = 0; t < 8; ++t) if (fabs(o[t * 64 + l] - want[t]) > 1e-3) { if (bad < 10) printf("lane %d test %d got %f want %f\n", l, t, o[t * 64 + l], want[t]); ++bad; }
  }
  printf(bad ? "FAILED %d\n" : "reductions ok\n", bad);
  return bad != 0;
}
