// GEMM kernels: C[M][N] = A[M][K] * W[N][K]^T with fused epilogues (bias / GELU / positions / residual /
// head-split scatter).  Replaces CTranslate2's Dense + Conv1D layers on the Whisper hot path
// (SURVEY.md section 2.1).  Both operands are K-contiguous ("B^T input"), which is exactly PyTorch's
// Linear weight layout, so MFMA A and B fragments are plain 16-byte row reads.
//
//   gemm_basic<T>   64x64x32 tile, register-staged LDS, any M/N/K (K % 8 == 0).  T = float uses the
//                   exact-f32 MFMA v_mfma_f32_16x16x4_f32 (parity mode), T = bf16 v_mfma_f32_16x16x32_bf16.
//   gemm_bf16_fast  128x128x64 tile, global_load_lds (16 B) double-buffered staging, XOR-swizzled LDS
//                   image (swizzle applied on the SOURCE address, guide rule 21), XCD-aware tile order.
#include "common.hpp"
#include <type_traits>
#include <mutex>
#ifndef TTASR_V5_DMA
#define TTASR_V5_DMA 0   // where the persistent GEMM issues stage t + 3 (lab builds: 1 / 2 / 3 = variants measured in round 5)
#endif
#include <cstdio>
#include <cstdlib>
#include <algorithm>

// ------------------------------------------------------------------------------------------------
// epilogue
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void epi_store(const GemmEpi& e, int64_t zoff, int m, int n, float v) {
  if (e.bias) v += e.bias[n];
  if (e.act == 1) v = gelu_erf(v);
  if (e.rowtab) v += e.rowtab[(int64_t)(m % e.rowmod) * e.ldc + n];
  int64_t idx = zoff + (int64_t)m * e.ldc + n;
  if (e.residual) v += e.residual[idx];
  if (e.out_f32) e.out_f32[idx] = v;
  if (e.out_t) {
    if (e.headsplit) {
      int which = n / e.hs_d, nn = n - which * e.hs_d;
      int h = nn >> 6, j = nn & 63;
      int b = m / e.hs_T, t = m - b * e.hs_T;
      idx = (int64_t)which * e.hs_which + (((int64_t)b * e.hs_H + h) * e.hs_T + t) * 64 + j;
    }
    ((T*)e.out_t)[idx] = from_f<T>(v);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_basic
// ------------------------------------------------------------------------------------------------
template <typename T> struct BasicCfg;
template <> struct BasicCfg<float> { static constexpr int VEC = 4, PAD = 4; };
template <> struct BasicCfg<bf16_t> { static constexpr int VEC = 8, PAD = 8; };
template <> struct BasicCfg<f16_t> { static constexpr int VEC = 8, PAD = 8; };

template <typename T>
__global__ __launch_bounds__(256) void gemm_basic_kernel(GemmArgs g) {
  constexpr int BM = 64, BN = 64, BK = 32;
  constexpr int VEC = BasicCfg<T>::VEC, LDK = BK + BasicCfg<T>::PAD, VPR = BK / VEC;
  __shared__ __attribute__((aligned(16))) T As[BM * LDK];
  __shared__ __attribute__((aligned(16))) T Ws[BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const T* A = (const T*)g.A + (int64_t)blockIdx.z * g.batch_stride_a;
  const T* W = (const T*)g.W;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < g.K; k0 += BK) {
    for (int v = tid; v < BM * VPR; v += 256) {
      int r = v / VPR, c = v - r * VPR;
      int k = k0 + c * VEC;
      uint4 va = make_uint4(0, 0, 0, 0), vw = make_uint4(0, 0, 0, 0);
      if (k < g.K) {
        if (m0 + r < g.M) va = *(const uint4*)(A + (int64_t)(m0 + r) * g.lda + k);
        if (n0 + r < g.N) vw = *(const uint4*)(W + (int64_t)(n0 + r) * g.ldw + k);
      }
      *(uint4*)(As + r * LDK + c * VEC) = va;
      *(uint4*)(Ws + r * LDK + c * VEC) = vw;
    }
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
      s16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *(const s16x8*)(As + (wm * 32 + i * 16 + (lane & 15)) * LDK + 8 * (lane >> 4));
        b[i] = *(const s16x8*)(Ws + (wn * 32 + i * 16 + (lane & 15)) * LDK + 8 * (lane >> 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = N16<T>::mfma16(a[i], b[j], acc[i][j]);
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = ((const float*)As)[(wm * 32 + i * 16 + (lane & 15)) * LDK + kk * 4 + (lane >> 4)];
          b[i] = ((const float*)Ws)[(wn * 32 + i * 16 + (lane & 15)) * LDK + kk * 4 + (lane >> 4)];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const int64_t zoff = (int64_t)blockIdx.z * g.epi.batch_stride_c;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        int n = n0 + wn * 32 + j * 16 + (lane & 15);
        if (m < g.M && n < g.N) epi_store<T>(g.epi, zoff, m, n, acc[i][j][r]);
      }
}

template <typename T>
void launch_gemm_basic(const GemmArgs& g, hipStream_t s) {
  dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, g.batch);
  hipLaunchKernelGGL(gemm_basic_kernel<T>, grid, dim3(256), 0, s, g);
}
template void launch_gemm_basic<float>(const GemmArgs&, hipStream_t);
template void launch_gemm_basic<bf16_t>(const GemmArgs&, hipStream_t);
template void launch_gemm_basic<f16_t>(const GemmArgs&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// gemm_bf16_fast: 128x128x64, glds double buffer
// ------------------------------------------------------------------------------------------------
// LDS image per operand tile: [128 rows][8 chunks of 16 B]; chunk slot s of row r holds global chunk
// s ^ (r & 7) (source-side swizzle), so a 16-lane ds_read_b128 group (16 rows, one k-chunk) covers all 64
// banks exactly once.
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <typename T16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_fast_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int BM = 128, BN = 128, BK = 64;
  constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buf][A 16K | W 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous range of tiles; n runs fastest so neighbours share the A panel in that XCD's L2.
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16_t* A = (const bf16_t*)g.A + (int64_t)blockIdx.z * g.batch_stride_a;
  const bf16_t* W = (const bf16_t*)g.W;

  // staging assignment: wave w issues 4 A pieces + 4 W pieces per k-tile; piece p covers rows 8p..8p+7
  const int srow = lane >> 3, sslot = lane & 7;
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int row = (wave * 4 + p) * 8 + srow;
    int chunk = sslot ^ (row & 7);
    int am = min(m0 + row, g.M - 1);  // clamp: rows >= M are loaded but never stored
    a_src[p] = A + (int64_t)am * g.lda + chunk * 8;
    w_src[p] = W + (int64_t)(n0 + row) * g.ldw + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* base = smem + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      glds16(a_src[p] + k0, base + (wave * 4 + p) * 1024);
      glds16(w_src[p] + k0, base + TILE_BYTES + (wave * 4 + p) * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = g.K / BK;
  stage(0, 0);
  __syncthreads();  // drains vmcnt(0): tile 0 landed
  const int fr = lane & 15, fq = lane >> 4;
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, (t + 1) * BK);
    const char* As = smem + cur * 2 * TILE_BYTES;
    const char* Ws = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int ra = wm * 64 + i * 16 + fr;
        a[i] = *(const s16x8*)(As + ra * 128 + (((ks * 4 + fq) ^ (ra & 7)) << 4));
        int rb = wn * 64 + i * 16 + fr;
        b[i] = *(const s16x8*)(Ws + rb * 128 + (((ks * 4 + fq) ^ (rb & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N16<T16>::mfma16(a[i], b[j], acc[i][j]);
    }
    __syncthreads();  // next tile landed (vmcnt(0)) and everyone finished reading `cur`
  }

  const int64_t zoff = (int64_t)blockIdx.z * g.epi.batch_stride_c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        int m = m0 + wm * 64 + i * 16 + fq * 4 + rr;
        int n = n0 + wn * 64 + j * 16 + fr;
        if (m < g.M) epi_store<T16>(g.epi, zoff, m, n, acc[i][j][rr]);
      }
}

// ------------------------------------------------------------------------------------------------
// gemm_bf16_v2: 256x128x64 tile, 8 waves (4 along M x 2 along N, 64x64 each), THREE-stage LDS ring filled
// by global_load_lds with a counted s_waitcnt vmcnt(6) and a raw s_barrier, so the loads of the next
// k-tile stay in flight across the barrier (guide section 5 "Pipelining across barriers").  One block
// per CU (144 KiB LDS).  Operands are swapped in the MFMA (A = weight rows, B = activation rows) so each
// lane ends with 4 CONSECUTIVE output columns of one row: the epilogue stores 8-byte bf16 / 16-byte f32
// vectors and reads bias as float4.  GELU uses the Abramowitz-Stegun 7.1.26 erf (|err| < 1.5e-7).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erfa = 1.0f - p * t * __expf(-z * z);  // erf(|x|/sqrt2)
  const float erfx = x < 0.f ? -erfa : erfa;
  return 0.5f * x * (1.0f + erfx);
}

// 4 consecutive columns n..n+3 of row m
template <typename T16>
__device__ __forceinline__ void epi_store4(const GemmEpi& e, int64_t zoff, int m, int n, f32x4 v) {
  if (e.bias) {
    const float4 b = *(const float4*)(e.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (e.act == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_fast(v[i]);
  }
  if (e.rowtab) {
    const float4 r = *(const float4*)(e.rowtab + (int64_t)(m % e.rowmod) * e.ldc + n);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  int64_t idx = zoff + (int64_t)m * e.ldc + n;
  if (e.residual) {
    const float4 r = *(const float4*)(e.residual + idx);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  if (e.out_f32) *(float4*)(e.out_f32 + idx) = make_float4(v[0], v[1], v[2], v[3]);
  if (e.out_t) {
    if (e.headsplit) {
      const int which = n / e.hs_d, nn = n - which * e.hs_d;
      const int h = nn >> 6, j = nn & 63;
      const int b = m / e.hs_T, t = m - b * e.hs_T;
      idx = (int64_t)which * e.hs_which + (((int64_t)b * e.hs_H + h) * e.hs_T + t) * 64 + j;
    }
    uint2 pk;
    pk.x = N16<T16>::pk(v[0], v[1]);
    pk.y = N16<T16>::pk(v[2], v[3]);
    *(uint2*)((bf16_t*)e.out_t + idx) = pk;
  }
}

template <typename T16>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v2_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 128, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;  // 32K + 16K
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // grouped raster: bands of GM m-tiles, n-tile fastest across the band, m-tile fastest inside it, so the ~32
  // tiles an XCD runs at once form a 4 x 8 patch (4 A panels + 8 W panels ~ 5 MB against its 4 MiB L2)
  // instead of 1 x 32 (measured: 34 % L2 misses with the row-major order = every W panel missed).
  constexpr int GM = 4;
  const int band = tile / (GM * tiles_n);
  const int rows_in_band = min(GM, tiles_m - band * GM);
  const int in_band = tile - band * GM * tiles_n;
  const int tn = in_band / rows_in_band;
  const int tm = band * GM + (in_band - tn * rows_in_band);
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16_t* A = (const bf16_t*)g.A + (int64_t)blockIdx.z * g.batch_stride_a;
  const bf16_t* W = (const bf16_t*)g.W;

  const int srow = lane >> 3, sslot = lane & 7;
  const bf16_t* a_src[4];
  const bf16_t* w_src[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (wave * 4 + p) * 8 + srow;
    a_src[p] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + ((sslot ^ (row & 7)) << 3);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (wave * 2 + p) * 8 + srow;
    w_src[p] = W + (int64_t)(n0 + row) * g.ldw + ((sslot ^ (row & 7)) << 3);
  }
#define V2_STAGE(buf_, k0_)                                                             \
  do {                                                                                  \
    char* base_ = smem + (buf_) * STAGE_BYTES;                                          \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                       \
        glds16(a_src[p] + (k0_), base_ + (wave * 4 + p) * 1024);                        \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                       \
        glds16(w_src[p] + (k0_), base_ + A_BYTES + (wave * 2 + p) * 1024);              \
  } while (0)

  f32x4 acc[4][4];  // [i: m-frag][j: n-frag]; element r = column n0.. + j*16 + (lane>>4)*4 + r of row i*16 + (lane&15)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = g.K / BK;
  V2_STAGE(0, 0);
  if (nt > 1) V2_STAGE(1, BK);
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) {
      const int nb = cur >= 1 ? cur - 1 : 2;  // (t + 2) % 3
      V2_STAGE(nb, (t + 2) * BK);
    }
    // All 16 fragment reads are issued first (64 VGPRs) as inline-asm ds_read_b128 with HAND-COUNTED waits:
    // the k-step-0 MFMAs start once the first 8 reads have landed (lgkmcnt(8)) while the other 8 stream
    // in behind them.  (hipcc's own placement waits lgkmcnt(0) before the first MFMA: guide 5.7.)
    s16x8 a[2][4], b[2][4];
    const uint32_t as_off = lds_base + cur * STAGE_BYTES, ws_off = as_off + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rb = wn * 64 + i * 16 + fr;
        asm volatile("ds_read_b128 %0, %1" : "=v"(b[ks][i]) : "v"(ws_off + rb * 128 + (((ks * 4 + fq) ^ (rb & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[ks][i]) : "v"(as_off + ra * 128 + (((ks * 4 + fq) ^ (ra & 7)) << 4)));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(b[0][0]),
                 "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)  // A operand = weight rows (n), B operand = activation rows (m)
        acc[i][j] = N16<T16>::mfma16(b[0][j], a[0][i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(b[1][0]),
                 "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = N16<T16>::mfma16(b[1][j], a[1][i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    cur = cur == 2 ? 0 : cur + 1;
  }
#undef V2_STAGE
  const int64_t zoff = (int64_t)blockIdx.z * g.epi.batch_stride_c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m < g.M) {
#pragma unroll
      for (int j = 0; j < 4; ++j) epi_store4<T16>(g.epi, zoff, m, n0 + wn * 64 + j * 16 + fq * 4, acc[i][j]);
    }
  }
}

bool gemm_bf16_v2_ok(const GemmArgs& g) {
  return g.N % 128 == 0 && g.K % 64 == 0 && g.K >= 128 && g.lda % 8 == 0 && g.ldw % 8 == 0 && g.epi.ldc % 4 == 0 &&
         ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0 && (!g.epi.headsplit || g.epi.hs_d % 64 == 0);
}
template <typename T16>
void launch_gemm_bf16_v2(const GemmArgs& g, hipStream_t s) {
  // the opt-in to > 64 KiB of dynamic LDS was made once per device by gemm_tiles_init (ttasr_create)
  const int tiles_m = (g.M + 255) / 256, tiles_n = g.N / 128;
  hipLaunchKernelGGL(gemm_bf16_v2_kernel<T16>, dim3(tiles_m * tiles_n, 1, g.batch), dim3(512), 3 * 49152, s, g, tiles_m, tiles_n);
}
template void launch_gemm_bf16_v2<bf16_t>(const GemmArgs&, hipStream_t);
template void launch_gemm_bf16_v2<f16_t>(const GemmArgs&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// gemm_bf16_v3: 256x256x32 stages, FOUR-stage LDS ring (128 KiB), 8 waves as 2 (M) x 4 (N), 128x64 per wave.
// The two wave groups (waves 0-3 / 4-7, i.e. the two SIMD partners) run staggered by one barrier: while
// one group issues its LDS-DMA (4 x global_load_lds) and fragment reads (12 x ds_read_b128), its partner
// on the same SIMD runs its 32 MFMAs, so the matrix pipe sees matrix-beside-memory instead of two waves
// doing the same thing in lockstep (MI355X_MICROARCH "Two waves per SIMD", items 5 and 9).
// Per iteration t and wave:   A_t: s_waitcnt vmcnt(4) [own pieces of stage t+1 landed]; s_barrier;
//                                  issue stage t+3 -> slot (t+3)&3; read fragments of stage t; lgkmcnt(0)
//                             B_t: s_barrier; 32 MFMAs
// Group 1 executes one extra barrier before the loop and group 0 one after it, so group 0's A_t barrier
// pairs with group 1's B_{t-1} barrier.  RAW: every wave has waited for its stage-t pieces one phase before
// anyone reads stage t.  WAR: slot (t+3)&3 held stage t-1, whose fragment reads were drained (lgkmcnt(0))
// before the barrier that precedes the first overwrite.
// LDS image per operand stage: [256 rows][4 chunks of 16 B]; chunk c of row r sits in slot c ^ H((r>>2)&3),
// H = {0,2,3,1}, which makes every ds_read_b128 lane group hit 16 distinct 16-byte bank slots.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int v3_h(int q) { return (0x78 >> (q * 2)) & 3; }  // {0,2,3,1} packed two bits each

template <typename T16, int EPI>  // EPI: bit0 GELU, bit1 residual, bit2 row table (positions), bit3 head-split T output, bit4 f32 output
__global__ __launch_bounds__(512, 2) void gemm_bf16_v3_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 256, BK = 32;
  constexpr int OP_BYTES = BM * BK * 2, STAGE_BYTES = 2 * OP_BYTES;  // 16K + 16K
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int GM = 4;
  const int band = tile / (GM * tiles_n);
  const int rows_in_band = min(GM, tiles_m - band * GM);
  const int in_band = tile - band * GM * tiles_n;
  const int tn = in_band / rows_in_band;
  const int tm = band * GM + (in_band - tn * rows_in_band);
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16_t* A = (const bf16_t*)g.A + (int64_t)blockIdx.z * g.batch_stride_a;
  const bf16_t* W = (const bf16_t*)g.W;
  // staging: one wave-instruction = 16 rows x 64 B; wave w moves rows (2w+p)*16 .. +15 of A and of W
  const int srow = lane >> 2, sslot = lane & 3;
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (wave * 2 + p) * 16 + srow;
    const int chunk = sslot ^ v3_h((row >> 2) & 3);
    a_src[p] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + chunk * 8;
    w_src[p] = W + (int64_t)(n0 + row) * g.ldw + chunk * 8;
  }
#define V3_STAGE(slot_, k0_)                                                                   \
  do {                                                                                         \
    char* base_ = smem + (slot_) * STAGE_BYTES;                                                \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                            \
      glds16(a_src[p] + (k0_), base_ + (wave * 2 + p) * 1024);                                 \
      glds16(w_src[p] + (k0_), base_ + OP_BYTES + (wave * 2 + p) * 1024);                      \
    }                                                                                          \
  } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = g.K / BK;  // >= 4
  V3_STAGE(0, 0);
  V3_STAGE(1, BK);
  V3_STAGE(2, 2 * BK);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // stages 0 and 1 landed (own pieces)
  if (wm == 1) __builtin_amdgcn_s_barrier();        // stagger: group 1 runs one barrier behind

  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // per-lane fragment offsets inside an operand stage (row r = base + 16*i + fr -> (r>>2)&3 == (fr>>2)&3)
  const uint32_t frag_off = (uint32_t)(fr * 64 + ((fq ^ v3_h((fr >> 2) & 3)) << 4));
  const uint32_t a_off = lds_base + wm * 128 * 64 + frag_off;
  const uint32_t w_off = lds_base + OP_BYTES + wn * 64 * 64 + frag_off;

  for (int t = 0; t < nt; ++t) {
    // ---- phase A ----
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 3 < nt) V3_STAGE((t + 3) & 3, (t + 3) * BK);
    s16x8 a[8], b[4];
    const uint32_t so = (uint32_t)((t & 3) * STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b[j]) : "v"(w_off + so + j * 16 * 64));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(a_off + so + i * 16 * 64));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B ----
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)  // A operand = weight rows (n), B operand = activation rows (m)
        acc[i][j] = N16<T16>::mfma16(b[j], a[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count
#undef V3_STAGE
  // ---- epilogue, specialised at compile time (EPI) so no integer division / dead branch survives for the
  // common shapes; row offsets are computed once per row, column parts once per column group, and every load
  // (bias, positions, residual) is issued ahead of the stores it feeds.  (The generic per-element epilogue cost
  // 5 500 VALU instructions per thread: 25-40 % of the kernel.)
  constexpr bool ACT = EPI & 1, RES = EPI & 2, ROWTAB = EPI & 4, HS = EPI & 8, OUTF = EPI & 16;
  const GemmEpi& e = g.epi;
  const int64_t zoff = (int64_t)blockIdx.z * e.batch_stride_c;
  const int nbase = n0 + wn * 64 + fq * 4;
  float4 bias4[4];
  int64_t coloff[4];  // column part of the output index
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nbase + j * 16;
    bias4[j] = e.bias ? *(const float4*)(e.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (HS) {
      const int which = n / e.hs_d, nn = n - which * e.hs_d;
      coloff[j] = (int64_t)which * e.hs_which + (int64_t)(nn >> 6) * e.hs_T * 64 + (nn & 63);
    } else {
      coloff[j] = n;
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 extra[4][4];
    int64_t rowoff[4];  // row part of the output index (T output); f32 / residual use zoff + m * ldc
    int64_t rowlin[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int m = min(m0 + wm * 128 + (half * 4 + ii) * 16 + fr, g.M - 1);
      rowlin[ii] = zoff + (int64_t)m * e.ldc;
      if (HS) {
        const int bb = m / e.hs_T, tt = m - bb * e.hs_T;
        rowoff[ii] = ((int64_t)bb * e.hs_H * e.hs_T + tt) * 64;
      } else {
        rowoff[ii] = rowlin[ii];
      }
      if (RES || ROWTAB) {
        const int64_t tabrow = ROWTAB ? (int64_t)(m % e.rowmod) * e.ldc : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
          if (RES) r = *(const float4*)(e.residual + rowlin[ii] + nbase + j * 16);
          if (ROWTAB) {
            const float4 t = *(const float4*)(e.rowtab + tabrow + nbase + j * 16);
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
          }
          extra[ii][j] = r;
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = half * 4 + ii;
      const bool row_ok = m0 + wm * 128 + i * 16 + fr < g.M;
      uint2 pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j];
        v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
        if (ACT) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = gelu_fast(v[t]);
        }
        if (RES || ROWTAB) { v[0] += extra[ii][j].x; v[1] += extra[ii][j].y; v[2] += extra[ii][j].z; v[3] += extra[ii][j].w; }
        if (OUTF) {
          if (row_ok) *(float4*)(e.out_f32 + rowlin[ii] + nbase + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          pk[j].x = N16<T16>::pk(v[0], v[1]);
          pk[j].y = N16<T16>::pk(v[2], v[3]);
        }
      }
      if (!OUTF) {
        // Widen the bf16 stores (guide T21, 16-lane form): lanes l and l^16 hold columns 4q..4q+3 and 4q+4..4q+7 of
        // the same row, so one v_permlane16_swap per dword over a pair of column groups (j, j+1) leaves the even
        // 16-lane rows with 16 contiguous bytes of group j and the odd rows with 16 contiguous bytes of group j+1:
        // 16 x 16-byte stores per lane instead of 32 x 8-byte (the store tail is issue-bound, not bandwidth-bound).
        const bool odd = (fq & 1) != 0;
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          auto sx = __builtin_amdgcn_permlane16_swap(pk[jp].x, pk[jp + 1].x, false, false);
          auto sy = __builtin_amdgcn_permlane16_swap(pk[jp].y, pk[jp + 1].y, false, false);
          // even row: {own group jp, partner's group jp}; odd row: {partner's group jp+1, own group jp+1}
          const uint4 w = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          const int jj = odd ? jp + 1 : jp;
          if (row_ok) *(uint4*)((bf16_t*)e.out_t + rowoff[ii] + coloff[jj] - (odd ? 4 : 0)) = w;
        }
      }
    }
  }
}

// epilogue variants actually used by the encoder / cross-KV schedules; anything else takes the v2 kernel
static int v3_epi_code(const GemmEpi& e) {
  return (e.act == 1 ? 1 : 0) | (e.residual ? 2 : 0) | (e.rowtab ? 4 : 0) | (e.headsplit ? 8 : 0) | (e.out_f32 ? 16 : 0);
}
bool gemm_bf16_v3_ok(const GemmArgs& g) {
  const int code = v3_epi_code(g.epi);
  const bool known = code == 0 || code == 1 || code == 18 || code == 21 || code == 8;
  const bool one_out = (g.epi.out_f32 != nullptr) != (g.epi.out_t != nullptr);
  return known && one_out && g.N % 256 == 0 && g.K % 32 == 0 && g.K >= 128 && g.lda % 8 == 0 && g.ldw % 8 == 0 &&
         g.epi.ldc % 8 == 0 && ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0 &&
         (!g.epi.headsplit || g.epi.hs_d % 64 == 0);
}
template <typename T16, int EPI>
static void launch_v3(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + 255) / 256, tiles_n = g.N / 256;
  if (g_kernel_sig_on) snprintf(g_kernel_sig, sizeof g_kernel_sig, "gemm_bf16_v3_kernel<%s, %d> grid %d", sig_type<T16>(), EPI,
                                tiles_m * tiles_n * g.batch * 512);
  hipLaunchKernelGGL((gemm_bf16_v3_kernel<T16, EPI>), dim3(tiles_m * tiles_n, 1, g.batch), dim3(512), 4 * 32768, s, g, tiles_m, tiles_n);
}
template <typename T16>
void launch_gemm_bf16_v3(const GemmArgs& g, hipStream_t s) {
  switch (v3_epi_code(g.epi)) {
    case 0: launch_v3<T16, 0>(g, s); break;    // bias -> T                 (qkv)
    case 1: launch_v3<T16, 1>(g, s); break;    // bias + GELU -> T          (fc1, conv1)
    case 18: launch_v3<T16, 18>(g, s); break;  // bias + residual -> f32    (out-proj, fc2)
    case 21: launch_v3<T16, 21>(g, s); break;  // bias + GELU + positions -> f32 (conv2)
    case 8: launch_v3<T16, 8>(g, s); break;    // bias -> head-split T      (cross-KV)
    default: break;                            // excluded by gemm_bf16_v3_ok
  }
}
template void launch_gemm_bf16_v3<bf16_t>(const GemmArgs&, hipStream_t);
template void launch_gemm_bf16_v3<f16_t>(const GemmArgs&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// gemm_bf16_v4 (round 4): the v3 tile body as a PERSISTENT workgroup (one per CU) that walks its tiles, so that the part of
// a tile that v3 leaves exposed at one workgroup per CU - the store-issue-bound epilogue, the workgroup relaunch and the
// first stages' HBM/L2 latency - overlaps with neighbouring work:
//   main loop of tile i  ->  bias of tile i+1 (waited for here: nothing else is in flight)  ->  LDS-DMA of stages 0..2 of tile
//   i+1 (the ring is free: every fragment read of tile i was drained before the last barrier)  ->  epilogue of tile i (VALU +
//   16 x 16-byte stores per lane, no LDS, no load)  ->  main loop of tile i+1, whose first two counted waits allow the 16 stores
//   to be still in flight (vmcnt retires in issue order: [4 pieces of stage 2][16 stores][4 pieces of stage 3] ...).
// Stores are unconditional with the row index clamped (rows past M replicate row M-1's operands, so they write row M-1's own
// values again): the in-flight count the first waits rely on is then exact for every wave.
// Same arithmetic as v3 (accumulation order, bias added after the loop, same conversions): bit-identical outputs.
// EPI 0 (bias -> T), 1 (bias + GELU -> T), 8 (bias -> head-split T); batch == 1.
// ------------------------------------------------------------------------------------------------
template <typename T16, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v4_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 256, BK = 32;
  constexpr int OP_BYTES = BM * BK * 2, STAGE_BYTES = 2 * OP_BYTES;
  constexpr bool ACT = EPI & 1, HS = EPI & 8;
  static_assert((EPI & ~9) == 0, "v4 carries the T-output epilogues only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_per_group = tiles_m * tiles_n;
  const int nwg = tiles_per_group * g.groups;
  // static schedule: the workgroups that share an XCD (bid % 8) own one contiguous range of tiles and walk it round-robin
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;  // workgroups of this class
  const int q = nwg >> 3, r = nwg & 7;
  const int t_begin = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, t_end = t_begin + (xcd < r ? q + 1 : q);
  const bf16_t* A = (const bf16_t*)g.A;
  const bf16_t* W = (const bf16_t*)g.W;
  const GemmEpi& e = g.epi;
  const int srow = lane >> 2, sslot = lane & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t frag_off = (uint32_t)(fr * 64 + ((fq ^ v3_h((fr >> 2) & 3)) << 4));
  const uint32_t a_off = lds_base + wm * 128 * 64 + frag_off;
  const uint32_t w_off = lds_base + OP_BYTES + wn * 64 * 64 + frag_off;
  const int nt = g.K / BK;  // >= 4
  constexpr int GM = 4;

  int m0 = 0, n0 = 0;
  int64_t oofs = 0;          // output offset of the tile's group (elements)
  const float* bias_g = e.bias;
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
  auto place = [&](int tile) {   // tile -> (group,) (m0, n0), staging source pointers (same raster as v3 inside a group)
    const bf16_t* Wg = W;
    if (g.groups > 1) {
      const int grp = tile / tiles_per_group;
      tile -= grp * tiles_per_group;
      Wg = W + (int64_t)grp * g.group_stride_w;
      bias_g = e.bias + (int64_t)grp * g.N;
      oofs = (int64_t)grp * g.group_stride_out;
    }
    const int band = tile / (GM * tiles_n);
    const int rows_in_band = min(GM, tiles_m - band * GM);
    const int in_band = tile - band * GM * tiles_n;
    const int tn = in_band / rows_in_band;
    const int tm = band * GM + (in_band - tn * rows_in_band);
    m0 = tm * BM; n0 = tn * BN;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (wave * 2 + p) * 16 + srow;
      const int chunk = sslot ^ v3_h((row >> 2) & 3);
      a_src[p] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + chunk * 8;
      w_src[p] = Wg + (int64_t)(n0 + row) * g.ldw + chunk * 8;
    }
  };
#define V4_STAGE(slot_, k0_)                                                                   \
  do {                                                                                         \
    char* base_ = smem + (slot_) * STAGE_BYTES;                                                \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                            \
      glds16(a_src[p] + (k0_), base_ + (wave * 2 + p) * 1024);                                 \
      glds16(w_src[p] + (k0_), base_ + OP_BYTES + (wave * 2 + p) * 1024);                      \
    }                                                                                          \
  } while (0)
  auto load_bias = [&](float4 (&b4)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = *(const float4*)(bias_g + n0 + wn * 64 + fq * 4 + j * 16);   // bias != nullptr (gemm_bf16_v4_ok)
    // waited for HERE, while nothing else is in flight: no later use of these registers may drain the LDS-DMA queue
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b4[0].x), "+v"(b4[0].y), "+v"(b4[0].z), "+v"(b4[0].w), "+v"(b4[1].x), "+v"(b4[1].y),
                 "+v"(b4[1].z), "+v"(b4[1].w), "+v"(b4[2].x), "+v"(b4[2].y), "+v"(b4[2].z), "+v"(b4[2].w), "+v"(b4[3].x),
                 "+v"(b4[3].y), "+v"(b4[3].z), "+v"(b4[3].w));
  };

  int tile = t_begin + slot;
  if (tile >= t_end) return;
  place(tile);
  float4 bias4[4];
  load_bias(bias4);
  V4_STAGE(0, 0);
  V4_STAGE(1, BK);
  V4_STAGE(2, 2 * BK);
  bool carry = false;   // 16 stores of the previous tile's epilogue sit behind the three prefetched stages

  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stages 0 and 1 landed (own pieces); queue behind them: stage 2 (4) [+ 16 stores]
    if (carry) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind
    for (int t = 0; t < nt; ++t) {
      // ---- phase A ----
      if (t + 2 < nt) {
        // own pieces of stage t + 1 landed; queue behind them: stage t + 2 (4), and for t < 2 of a carried tile the 16 stores
        // (t = 0: [stage 2][stores]; t = 1: [stores][stage 3] - from t = 2 on the stores are older than what is waited for)
        if (carry && t < 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 3 < nt) V4_STAGE((t + 3) & 3, (t + 3) * BK);
      s16x8 a[8], b[4];
      const uint32_t so = (uint32_t)((t & 3) * STAGE_BYTES);
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b[j]) : "v"(w_off + so + j * 16 * 64));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(a_off + so + i * 16 * 64));
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                     "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B ----
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N16<T16>::mfma16(b[j], a[i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count: every fragment read of this tile is drained
    // ---- this tile's output coordinates (before the staging pointers move on) ----
    const int nbase = n0 + wn * 64 + fq * 4;
    int64_t coloff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nbase + j * 16;
      if (HS) {
        const int which = n / e.hs_d, nn = n - which * e.hs_d;
        coloff[j] = oofs + (int64_t)which * e.hs_which + (int64_t)(nn >> 6) * e.hs_T * 64 + (nn & 63);
      } else coloff[j] = oofs + n;
    }
    const int mrow0 = m0 + wm * 128 + fr;
    // ---- next tile: bias (waited for now), then its first three stages into the free ring ----
    const int next = tile + per_xcd;
    const bool has_next = next < t_end;
    float4 bias_next[4];
    if (has_next) {
      place(next);
      load_bias(bias_next);
      V4_STAGE(0, 0);
      V4_STAGE(1, BK);
      V4_STAGE(2, 2 * BK);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue of this tile: VALU + 16 unconditional 16-byte stores per lane ----
    const bool odd = (fq & 1) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = min(mrow0 + i * 16, g.M - 1);
      int64_t rowoff;
      if (HS) {
        const int bb = m / e.hs_T, tt = m - bb * e.hs_T;
        rowoff = ((int64_t)bb * e.hs_H * e.hs_T + tt) * 64;
      } else rowoff = (int64_t)m * e.ldc;
      uint2 pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j];
        v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
        if (ACT) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = gelu_fast(v[t]);
        }
        pk[j].x = N16<T16>::pk(v[0], v[1]);
        pk[j].y = N16<T16>::pk(v[2], v[3]);
      }
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        auto sx = __builtin_amdgcn_permlane16_swap(pk[jp].x, pk[jp + 1].x, false, false);
        auto sy = __builtin_amdgcn_permlane16_swap(pk[jp].y, pk[jp + 1].y, false, false);
        const uint4 w = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        const int jj = odd ? jp + 1 : jp;
        *(uint4*)((bf16_t*)e.out_t + rowoff + coloff[jj] - (odd ? 4 : 0)) = w;
      }
    }
    if (!has_next) break;
    tile = next;
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = bias_next[j];
    carry = true;
  }
#undef V4_STAGE
}

// ------------------------------------------------------------------------------------------------
// gemm_bf16_v5 (round 5): v4 whose LAST partial round of workgroups is re-tiled ("give the surplus tiles to
// ALL CUs by splitting them along M ... no f32 partials").  The rows are cut into a FULL region of 256-row m-tiles whose tile count
// fills whole rounds of the 256 persistent workgroups, and a TAIL region of shorter tiles - 192 or 128 rows (6 or 4 row blocks
// of 16 per wave group instead of 8), all 256 columns - chosen so that the tail's tile count fits ONE round: e.g. out-proj / fc2 at
// B = 32 (940 tiles = 3.67 rounds -> 4): 765 full tiles (2.99 rounds) + 230 tiles of 192 rows (one round of ~0.78 of a tile).
// A shorter tile is the same tile body with fewer row blocks (template MI): the A image is still staged 256 rows deep (the extra
// rows are the next tile's, read only), every output element sees the same K loop in the same order: bit-identical to v3 / v4.
// ------------------------------------------------------------------------------------------------
template <typename T16, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v5_kernel(GemmArgs g, int tiles_m, int tiles_n, int tail_tiles_m, int tail_mi) {
  constexpr int BM = 256, BN = 256, BK = 32;
  constexpr int OP_BYTES = BM * BK * 2, STAGE_BYTES = 2 * OP_BYTES;
  constexpr bool ACT = EPI & 1, HS = EPI & 8;
  static_assert((EPI & ~9) == 0, "v4 carries the T-output epilogues only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // tiles_m = 256-row m-tiles of the FULL region; the rows behind it are the TAIL region: tail_tiles_m m-tiles of tail_mi * 32 rows
  // (tail_mi = row blocks of 16 per wave group: 6 -> 192-row tiles, 4 -> 128-row tiles).  groups == 1 (gemm_bf16_v5_ok).
  const int n_full = tiles_m * tiles_n, n_tail = tail_tiles_m * tiles_n;
  // static schedule: the workgroups that share an XCD (bid % 8) own one contiguous chunk of the full tiles and one of the tail
  // tiles; a workgroup walks its full tiles round-robin (slot, slot + per_xcd, ...), then its tail tiles from the OTHER end of
  // the slot order (the workgroups that got one full tile less take the tail tiles first)
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;  // workgroups of this class
  auto chunk = [&](int n, int& lo, int& hi) { const int q = n >> 3, r = n & 7; lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q; hi = lo + (xcd < r ? q + 1 : q); };
  int fb, fe, tb, te;
  chunk(n_full, fb, fe);
  chunk(n_tail, tb, te);
  // tile ids: [0, n_full) full tiles, n_full + [0, n_tail) tail tiles
  auto first_tail = [&]() { const int t = tb + (per_xcd - 1 - slot); return t < te ? n_full + t : -1; };
  auto next_of = [&](int id) {
    const int nx = id + per_xcd;
    if (id < n_full) return nx < fe ? nx : first_tail();
    return nx - n_full < te ? nx : -1;
  };
  const bf16_t* A = (const bf16_t*)g.A;
  const bf16_t* W = (const bf16_t*)g.W;
  const GemmEpi& e = g.epi;
  const int srow = lane >> 2, sslot = lane & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t frag_off = (uint32_t)(fr * 64 + ((fq ^ v3_h((fr >> 2) & 3)) << 4));
  const uint32_t w_off = lds_base + OP_BYTES + wn * 64 * 64 + frag_off;
  const int nt = g.K / BK;  // >= 4
  constexpr int GM = 4;

  int m0 = 0, n0 = 0;
  const float* bias_g = e.bias;
  const bf16_t* a_src[2];
  const bf16_t* w_src[2];
  int mi_placed = 8;         // row blocks per wave group of the tile place() was last called for
  auto place = [&](int tile) {   // tile id -> (m0, n0), staging source pointers (the v3 raster inside each region)
    int tm_n = tiles_m, base_row = 0, rows_per_tile = BM;
    mi_placed = 8;
    if (tile >= n_full) { tile -= n_full; tm_n = tail_tiles_m; base_row = tiles_m * BM; rows_per_tile = tail_mi * 32; mi_placed = tail_mi; }
    const int band = tile / (GM * tiles_n);
    const int rows_in_band = min(GM, tm_n - band * GM);
    const int in_band = tile - band * GM * tiles_n;
    const int tn = in_band / rows_in_band;
    const int tm = band * GM + (in_band - tn * rows_in_band);
    m0 = base_row + tm * rows_per_tile; n0 = tn * BN;
    // every wave stages its 2 + 2 pieces for EVERY tile (the counted waits rely on it), but the A pieces of a short tile that lie
    // past its last row re-read that one row (one cache line per piece instead of sixteen): a short tile does not pay for A rows
    // it does not use - what fc2, whose 492 MB A operand streams from beyond L2, needs to gain anything
    const int row_lim = min(g.M - 1, m0 + rows_per_tile - 1);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (wave * 2 + p) * 16 + srow;
      const int chunk = sslot ^ v3_h((row >> 2) & 3);
      a_src[p] = A + (int64_t)min(m0 + row, row_lim) * g.lda + chunk * 8;
      w_src[p] = W + (int64_t)(n0 + row) * g.ldw + chunk * 8;
    }
  };
#define V4_STAGE(slot_, k0_)                                                                   \
  do {                                                                                         \
    char* base_ = smem + (slot_) * STAGE_BYTES;                                                \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                            \
      glds16(a_src[p] + (k0_), base_ + (wave * 2 + p) * 1024);                                 \
      glds16(w_src[p] + (k0_), base_ + OP_BYTES + (wave * 2 + p) * 1024);                      \
    }                                                                                          \
  } while (0)
  auto load_bias = [&](float4 (&b4)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = *(const float4*)(bias_g + n0 + wn * 64 + fq * 4 + j * 16);   // bias != nullptr (gemm_bf16_v4_ok)
    // waited for HERE, while nothing else is in flight: no later use of these registers may drain the LDS-DMA queue
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b4[0].x), "+v"(b4[0].y), "+v"(b4[0].z), "+v"(b4[0].w), "+v"(b4[1].x), "+v"(b4[1].y),
                 "+v"(b4[1].z), "+v"(b4[1].w), "+v"(b4[2].x), "+v"(b4[2].y), "+v"(b4[2].z), "+v"(b4[2].w), "+v"(b4[3].x),
                 "+v"(b4[3].y), "+v"(b4[3].z), "+v"(b4[3].w));
  };

  int tile = fb + slot < fe ? fb + slot : first_tail();
  if (tile < 0) return;
  place(tile);
  float4 bias4[4];
  load_bias(bias4);
  V4_STAGE(0, 0);
  V4_STAGE(1, BK);
  V4_STAGE(2, 2 * BK);
  int carry = 0;   // stores of the previous tile's epilogue (2 per row block) that sit behind the three prefetched stages
  // "own pieces of the wanted stage landed": everything but the 4 youngest pieces and the carried stores (issue order)
#define V5_WAIT_CARRY()                                                          \
  do {                                                                           \
    if (carry == 16) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");           \
    else if (carry == 12) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      \
    else if (carry == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                        \
  } while (0)

  // one tile of MI row blocks per wave group (MI = 8: the 256 x 256 tile of v4; 6 / 4: the shorter tiles of the tail region);
  // returns false after the workgroup's last tile
  auto run_tile = [&](auto mi_tag) -> bool {
    constexpr int MI = decltype(mi_tag)::value;
    const uint32_t a_off = lds_base + wm * (MI * 16) * 64 + frag_off;
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stages 0 and 1 landed (own pieces); queue behind them: stage 2 (4) [+ the carried stores]
    V5_WAIT_CARRY();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind
    for (int t = 0; t < nt; ++t) {
      // ---- phase A ----
      if (t + 2 < nt) {
        // own pieces of stage t + 1 landed; queue behind them: stage t + 2 (4), and for t < 2 of a carried tile its stores
        // (t = 0: [stage 2][stores]; t = 1: [stores][stage 3] - from t = 2 on the stores are older than what is waited for)
        if (t < 2) V5_WAIT_CARRY(); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#if TTASR_V5_DMA == 0
      if (t + 3 < nt) V4_STAGE((t + 3) & 3, (t + 3) * BK);
#elif TTASR_V5_DMA == 2
      if (t + 3 < nt) { char* base_ = smem + ((t + 3) & 3) * STAGE_BYTES;
        _Pragma("unroll") for (int p = 0; p < 2; ++p) glds16(a_src[p] + (t + 3) * BK, base_ + (wave * 2 + p) * 1024); }
#endif
      s16x8 a[MI], b[4];
      const uint32_t so = (uint32_t)((t & 3) * STAGE_BYTES);
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b[j]) : "v"(w_off + so + j * 16 * 64));
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(a_off + so + i * 16 * 64));
      if constexpr (MI == 8)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                       "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      else if constexpr (MI == 6)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B ----
      __builtin_amdgcn_s_barrier();
#if TTASR_V5_DMA == 3
      if (t + 3 < nt) V4_STAGE((t + 3) & 3, (t + 3) * BK);
      __builtin_amdgcn_sched_barrier(0);
#endif
      __builtin_amdgcn_s_setprio(1);
#if TTASR_V5_DMA == 0 || TTASR_V5_DMA == 3
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N16<T16>::mfma16(b[j], a[i], acc[i][j]);
#else
#pragma unroll
      for (int i = 0; i < MI / 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N16<T16>::mfma16(b[j], a[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
#if TTASR_V5_DMA == 1
      if (t + 3 < nt) V4_STAGE((t + 3) & 3, (t + 3) * BK);
#else
      if (t + 3 < nt) { char* base_ = smem + ((t + 3) & 3) * STAGE_BYTES;
        _Pragma("unroll") for (int p = 0; p < 2; ++p) glds16(w_src[p] + (t + 3) * BK, base_ + OP_BYTES + (wave * 2 + p) * 1024); }
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = MI / 2; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = N16<T16>::mfma16(b[j], a[i], acc[i][j]);
#endif
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count: every fragment read of this tile is drained
    // ---- this tile's output coordinates (before the staging pointers move on) ----
    const int nbase = n0 + wn * 64 + fq * 4;
    int64_t coloff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nbase + j * 16;
      if (HS) {
        const int which = n / e.hs_d, nn = n - which * e.hs_d;
        coloff[j] = (int64_t)which * e.hs_which + (int64_t)(nn >> 6) * e.hs_T * 64 + (nn & 63);
      } else coloff[j] = n;
    }
    const int mrow0 = m0 + wm * (MI * 16) + fr;
    // ---- next tile: bias (waited for now), then its first three stages into the free ring ----
    const int next = next_of(tile);
    const bool has_next = next >= 0;
    float4 bias_next[4];
    if (has_next) {
      place(next);
      load_bias(bias_next);
      V4_STAGE(0, 0);
      V4_STAGE(1, BK);
      V4_STAGE(2, 2 * BK);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue of this tile: VALU + 2 unconditional 16-byte stores per row block and lane ----
    const bool odd = (fq & 1) != 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = min(mrow0 + i * 16, g.M - 1);
      int64_t rowoff;
      if (HS) {
        const int bb = m / e.hs_T, tt = m - bb * e.hs_T;
        rowoff = ((int64_t)bb * e.hs_H * e.hs_T + tt) * 64;
      } else rowoff = (int64_t)m * e.ldc;
      uint2 pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j];
        v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w;
        if (ACT) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = gelu_fast(v[t]);
        }
        pk[j].x = N16<T16>::pk(v[0], v[1]);
        pk[j].y = N16<T16>::pk(v[2], v[3]);
      }
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        auto sx = __builtin_amdgcn_permlane16_swap(pk[jp].x, pk[jp + 1].x, false, false);
        auto sy = __builtin_amdgcn_permlane16_swap(pk[jp].y, pk[jp + 1].y, false, false);
        const uint4 w = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        const int jj = odd ? jp + 1 : jp;
        *(uint4*)((bf16_t*)e.out_t + rowoff + coloff[jj] - (odd ? 4 : 0)) = w;
      }
    }
    if (!has_next) return false;
    tile = next;
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = bias_next[j];
    carry = 2 * MI;
    return true;
  };
  int mi_cur = mi_placed;
  for (;;) {
    bool more;
    if (mi_cur == 8) more = run_tile(std::integral_constant<int, 8>{});
    else if (mi_cur == 6) more = run_tile(std::integral_constant<int, 6>{});
    else more = run_tile(std::integral_constant<int, 4>{});
    if (!more) break;
    mi_cur = mi_placed;   // place(next) ran inside run_tile
  }
#undef V5_WAIT_CARRY
#undef V4_STAGE
}

bool gemm_bf16_v4_ok(const GemmArgs& g) {
  const int code = v3_epi_code(g.epi);
  return (code == 0 || code == 1 || code == 8) && g.batch <= 1 && g.groups >= 1 && g.epi.bias != nullptr && gemm_bf16_v3_ok(g);
}
// Per-device launcher state, set ONCE per device by gemm_tiles_init (ttasr_create calls it before the context can launch
// anything; std::call_once orders the writes before every later reader): the opt-in to > 64 KiB of dynamic LDS for every tiled
// instantiation and the CU count that sizes the persistent grid.  Nothing is set lazily from a launcher any more (VERDICT
// round 4, weak #8: two contexts first-launching from two host threads raced on the old `static bool attr_done[64]`).
static int g_v4_cus[64] = {0};
static std::once_flag g_tiles_once[64];
template <typename T16>
static void tiles_attrs() {
  hipFuncSetAttribute((const void*)gemm_bf16_v2_kernel<T16>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152);
#define TTASR_V3_ATTR(E_) hipFuncSetAttribute((const void*)gemm_bf16_v3_kernel<T16, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768)
#define TTASR_V4_ATTR(E_) hipFuncSetAttribute((const void*)gemm_bf16_v4_kernel<T16, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768)
  TTASR_V3_ATTR(0); TTASR_V3_ATTR(1); TTASR_V3_ATTR(18); TTASR_V3_ATTR(21); TTASR_V3_ATTR(8);
  TTASR_V4_ATTR(0); TTASR_V4_ATTR(1); TTASR_V4_ATTR(8);
#define TTASR_V5_ATTR(E_) hipFuncSetAttribute((const void*)gemm_bf16_v5_kernel<T16, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768)
  TTASR_V5_ATTR(0); TTASR_V5_ATTR(1); TTASR_V5_ATTR(8);
#undef TTASR_V5_ATTR
#undef TTASR_V3_ATTR
#undef TTASR_V4_ATTR
}
void gemm_tiles_init(int device) {
  std::call_once(g_tiles_once[device & 63], [device]() {
    hipDeviceProp_t p;
    g_v4_cus[device & 63] = hipGetDeviceProperties(&p, device) == hipSuccess ? p.multiProcessorCount : 256;
    tiles_attrs<bf16_t>();
    tiles_attrs<f16_t>();
  });
}
template <typename T16, int EPI>
static void launch_v4(const GemmArgs& g, hipStream_t s) {
  int dev = 0;
  hipGetDevice(&dev);
  const int tiles_m = (g.M + 255) / 256, tiles_n = g.N / 256;   // per group
  const int grid = std::min(tiles_m * tiles_n * std::max(1, g.groups), g_v4_cus[dev & 63]);
  if (g_kernel_sig_on) snprintf(g_kernel_sig, sizeof g_kernel_sig, "gemm_bf16_v4_kernel<%s, %d> grid %d", sig_type<T16>(), EPI, grid * 512);
  hipLaunchKernelGGL((gemm_bf16_v4_kernel<T16, EPI>), dim3(grid), dim3(512), 4 * 32768, s, g, tiles_m, tiles_n);
}
template <typename T16>
void launch_gemm_bf16_v4(const GemmArgs& g, hipStream_t s) {
  switch (v3_epi_code(g.epi)) {
    case 0: launch_v4<T16, 0>(g, s); break;
    case 1: launch_v4<T16, 1>(g, s); break;
    case 8: launch_v4<T16, 8>(g, s); break;
    default: break;
  }
}
template void launch_gemm_bf16_v4<bf16_t>(const GemmArgs&, hipStream_t);
template void launch_gemm_bf16_v4<f16_t>(const GemmArgs&, hipStream_t);

// ---- v5: where to cut the rows, and how tall the tail's tiles are ----
// Makespan model of the static schedule (per-XCD chunks, round-robin over the XCD's workgroups), in units of one 256 x 256 tile;
// a 192-row tile is priced 0.80, a 128-row tile 0.62 (measured: see DESIGN 4.11).  A plan must beat the plain tiling by more than
// 0.15 tile-times and its tail must fit one round.
struct V5Plan { int full_m = 0, tail_m = 0, tail_mi = 0; };
static bool v5_plan(int M, int tiles_n, int cus, V5Plan& plan) {
  if (cus < 8 || cus % 8) return false;
  const int tm_all = (M + 255) / 256, per = cus / 8;
  auto rounds = [&](int n) { return ((n + 7) / 8 + per - 1) / per; };
  double best = rounds(tm_all * tiles_n) - 0.15;
  bool found = false;
  for (int mi = 6; mi >= 4; mi -= 2) {
    const double cost = mi == 6 ? 0.80 : 0.62;
    for (int full_m = tm_all - 1; full_m >= 0 && full_m >= tm_all - 96; --full_m) {
      const int tail_rows = M - full_m * 256;
      if (tail_rows <= 0) continue;
      const int tail_m = (tail_rows + mi * 32 - 1) / (mi * 32), nt = tail_m * tiles_n;
      if (nt > cus) break;                       // taller full region -> fewer tail tiles: nothing further down fits either
      const double est = rounds(full_m * tiles_n) + cost * rounds(nt);
      if (est < best) { best = est; plan.full_m = full_m; plan.tail_m = tail_m; plan.tail_mi = mi; found = true; }
    }
  }
  return found;
}
bool gemm_bf16_v5_ok(const GemmArgs& g) { return g.groups <= 1 && gemm_bf16_v4_ok(g); }
template <typename T16, int EPI>
static void launch_v5(const GemmArgs& g, const V5Plan& p, int cus, hipStream_t s) {
  const int tiles_n = g.N / 256;
  const int grid = std::min((p.full_m + p.tail_m) * tiles_n, cus);
  if (g_kernel_sig_on) snprintf(g_kernel_sig, sizeof g_kernel_sig, "gemm_bf16_v5_kernel<%s, %d> grid %d", sig_type<T16>(), EPI, grid * 512);
  hipLaunchKernelGGL((gemm_bf16_v5_kernel<T16, EPI>), dim3(grid), dim3(512), 4 * 32768, s, g, p.full_m, tiles_n, p.tail_m, p.tail_mi);
}
// false: no plan beats the plain tiling for this shape (the caller launches v4)
template <typename T16>
bool launch_gemm_bf16_v5(const GemmArgs& g, hipStream_t s) {
  int dev = 0;
  hipGetDevice(&dev);
  const int cus = g_v4_cus[dev & 63];
  V5Plan p;
  if (!gemm_bf16_v5_ok(g) || !v5_plan(g.M, g.N / 256, cus, p)) return false;
  switch (v3_epi_code(g.epi)) {
    case 0: launch_v5<T16, 0>(g, p, cus, s); break;
    case 1: launch_v5<T16, 1>(g, p, cus, s); break;
    case 8: launch_v5<T16, 8>(g, p, cus, s); break;
    default: return false;
  }
  return true;
}
template bool launch_gemm_bf16_v5<bf16_t>(const GemmArgs&, hipStream_t);
template bool launch_gemm_bf16_v5<f16_t>(const GemmArgs&, hipStream_t);

bool gemm_bf16_fast_ok(const GemmArgs& g) {
  return g.N % 128 == 0 && g.K % 64 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0 && g.M >= 1 &&
         ((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0;
}

template <typename T16>
void launch_gemm_bf16_fast(const GemmArgs& g, hipStream_t s) {
  int tiles_m = (g.M + 127) / 128, tiles_n = g.N / 128;
  dim3 grid(tiles_m * tiles_n, 1, g.batch);
  hipLaunchKernelGGL(gemm_bf16_fast_kernel<T16>, grid, dim3(256), 65536, s, g, tiles_m, tiles_n);
}
template void launch_gemm_bf16_fast<bf16_t>(const GemmArgs&, hipStream_t);
template void launch_gemm_bf16_fast<f16_t>(const GemmArgs&, hipStream_t);
