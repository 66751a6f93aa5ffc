// Decode-time weight-streaming GEMM (bf16): out[b][n] = sum_k x[b][k] * W[n][k] for B <= 32 rows (one row per
// sequence in the batch).  HBM-bound on the weights: every weight byte is read once per step for the whole
// batch (SURVEY.md section 8d "algorithmic bytes per decode step").
//
// MI355X-native layout: the decoder weights are re-packed ONCE at load time into MFMA-fragment order so the
// stream is perfectly coalesced with no LDS staging ("GEMV / M <= 16 decode weights: load straight to VGPRs",
// cdna_hip_programming.md section 5): for n-block nb (32 output rows) and k-step ks (16 inputs) the 1 KiB at
// ((nb * K/16 + ks) * 64 + lane) * 16 B holds W[nb*32 + (lane & 31)][ks*16 + 8*(lane >> 5) ..+8], i.e. exactly
// lane `lane`'s A operand of v_mfma_f32_32x32x16_bf16.  The batch rows are the B operand
// (x[b = lane & 31][k...], L2-resident), so D = W_tile * x^T with rows = n, cols = b.
//
// What the measurements said (tools/microbench/skinny_var.hip, fc1 shape, graph-replayed chain, boundary = 1.6 us):
//   weight stream alone 3.5 us; + MFMA/LDS-reduce/store 4.6-4.8 us; + batch-row loads at a 1:1 byte ratio
//   5.0 us, at the 2:1 ratio of the 16x16x32 form 7.1 us; nontemporal weight loads +1.8 us.
// Hence: 32x32x16 (1 KiB of x per 1 KiB of W), plain loads, and every load of the kernel (fragments, bias,
// residual) issued before the first use: one memory round trip per kernel (each extra one costs ~1.5 us in
// the decode chain).  NW waves split K inside the workgroup (LDS reduce, fixed order).
//
// What bounds these kernels (round 2, rocprofv3 timeline of the real chain): a CU takes in only ~25 GB/s of HBM-cold
// bytes (its outstanding-miss queue over ~1 us of latency), and after every kernel boundary the activation rows come from
// the Infinity Cache, not L2.  So a GEMM wants its weights spread over ALL 256 CUs in pieces of <= ~40 KB: every decode GEMM
// whose consumer can sum partial results splits K across `ksplit` workgroups, each writing its partial tile to f32 slab
// `ks` ([ksplit][rows][ldc]) - no float atomics anywhere, so every result is bit-reproducible.  The consumers add the
// slabs in slab order: the LayerNorm after a residual GEMM (layernorm_rows_kernel: x + bias + slab[0] + ...), the
// self-attention kernel (q, k, v) and the cross-attention kernel (q).  fc1 (GELU needs the full sum) and the vocabulary
// GEMM stay unsplit.
#include "common.hpp"
#include <mutex>
#include <cstdlib>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// Rows per n-block of a fragment-packed decode matrix (round 5).  The MFMA tile has 32 output rows, but a workgroup need not
// use all of them: with 20-row blocks fc1 of the large-v3 family (N 5120, never K-split: GELU needs the full sum) runs as 256
// workgroups of 51 KB of weights - one per CU - instead of 160 of 80 KB.  The packed stream holds ONLY the 20 rows (640 B per
// k-step, contiguous); lanes 20..31 of each half re-read row 19 (same cache lines, no traffic) and their output rows are never
// stored.  Measured per launch in a graph-replayed chain of cold matrices (tools/microbench/skinny_bench3.hip, round 5):
//   fc1 7.83 -> 7.53 us;  fc2 (8 K slices) 7.27 -> 7.26;  out-proj / q 3.45 (32 rows, 5 slices) vs 3.77 (20 rows, 4 slices).
// So only fc1 takes the narrow blocks (rule: the 20-row block count is a multiple of the 256 CUs); what these kernels wait for is
// not the per-CU share of the stream alone - they sit 1.3-1.7 us above a kernel that only moves the same bytes on the same grid.
thread_local int g_skinny_narrow = 1;   // option dec_narrow_blocks (per context; fixed once the weights are packed)
int skinny_rows_per_block(int N, int K) {
  (void)K;
  if (!g_skinny_narrow) return 32;
  return (N % 20 == 0 && (N / 20) % 256 == 0) ? 20 : 32;
}

template <typename T16>
__global__ void shuffle_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows, int K, int row_offset, int rpb) {
  // one thread per 16-byte output chunk; rows past `rows` (padding of the last block) are left zero.  Chunk order inside
  // (n-block, k-step): [half = which 8 of the 16 inputs][row in block] - for rpb = 32 exactly lane order of the A operand
  const int ks_per = K / 16, per = 2 * rpb;
  const int n_blocks = (rows + rpb - 1) / rpb;
  const int64_t n_chunks = (int64_t)n_blocks * ks_per * per;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
    const int within = (int)(c % per);
    const int64_t blk = c / per;
    const int ks = (int)(blk % ks_per), nb = (int)(blk / ks_per);
    const int half = within / rpb, lr = within - half * rpb;
    const int r = nb * rpb + lr;
    if (r >= rows) continue;
    const float* s = src + (int64_t)r * K + ks * 16 + 8 * half;
    uint4 o;
    o.x = N16<T16>::pk(s[0], s[1]);
    o.y = N16<T16>::pk(s[2], s[3]);
    o.z = N16<T16>::pk(s[4], s[5]);
    o.w = N16<T16>::pk(s[6], s[7]);
    ((uint4*)dst)[((int64_t)(nb + row_offset / rpb) * ks_per + ks) * per + within] = o;
  }
}
// rows_total = rows of the whole packed matrix (a fused q/k/v matrix arrives as three parts): it decides the block height
template <typename T16>
void launch_shuffle_cast(const float* src, T16* dst_base, int rows, int K, int row_offset, hipStream_t s, int rows_total) {
  const int rpb = skinny_rows_per_block(rows_total > 0 ? rows_total : rows, K);
  const int64_t n_chunks = (int64_t)((rows + rpb - 1) / rpb) * (K / 16) * 2 * rpb;
  const int64_t nb = (n_chunks + 255) / 256;
  hipLaunchKernelGGL(shuffle_cast_kernel<T16>, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, s, src, (bf16_t*)dst_base, rows, K,
                     row_offset, rpb);
}
template void launch_shuffle_cast<bf16_t>(const float*, bf16_t*, int, int, int, hipStream_t, int);
template void launch_shuffle_cast<f16_t>(const float*, f16_t*, int, int, int, hipStream_t, int);

// NW waves per workgroup, each owning steps_per_wave k-steps of 16; RB groups of 32 batch rows share every weight
// fragment (RB = 1 is the B <= 32 kernel of the benchmark; RB = 2..4 carry 64..128 rows through one weight stream,
// which is what amortises the per-step latency when more clips are in flight).
thread_local int g_skinny_x_lds = 1;   // option dec_x_lds: the decode GEMMs stage their activation tile through LDS (0 = fragment loads straight from memory; bit-identical)
// dynamic LDS of gemm_skinny_kernel: the reduction buffer, or the (larger) activation tile it is aliased with
static size_t skinny_lds_bytes(int nw, int rb, int steps, bool x_lds) {
  const size_t red = (size_t)nw * rb * 4096;
  const size_t tile = x_lds ? (size_t)rb * 32 * ((size_t)nw * steps * 32 + 16) : 0;
  return std::max(red, tile);
}
// more than 64 KiB of dynamic LDS needs the opt-in, once per (instantiation, device); std::call_once: two contexts may first-launch
// the same instantiation from two host threads
template <auto Kern>
static void skinny_allow_big_lds() {
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 63], []() { (void)hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
}
thread_local int g_skinny_nt = 1;  // nontemporal weight loads in the decode GEMMs (option weights_nontemporal = 0 switches them off: A/B experiments; the 1.8 GB of decoder weights a step streams can never stay cached: -1 % per step)

// ONE = the wave's k-steps fit one batch of loads (steps <= U): straight-line code.  (As a loop, the register reuse of
// the next iteration forces an early s_waitcnt that, in the first iteration, waits for the bias / residual prefetch
// before the bulk of the weight loads is even issued: one more serialised round trip.)
template <typename T16, int NW, int RB, int U, bool NT, bool ONE>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const bf16_t* __restrict__ Wsh_, const bf16_t* __restrict__ x_,
                                                              int B_, int N_, int K_, int ksplit_, int steps_, GemmEpi e,
                                                              float* __restrict__ slab_, int64_t slab_stride_, int rpb_) {
  // U = k-steps in flight per wave (register budget: U * (1 + RB) * 4); the launcher picks the smallest instantiated
  // U >= steps so that no load is issued twice
  // dynamic LDS: the reduction buffer red[wave][row group][b*32 + n] (NW * RB * 4 KiB) and - aliased with it, separated by a
  // barrier - the activation tile of the LDS-staged form (skinny_lds_bytes)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float (*red)[RB][32 * 32] = (float (*)[RB][32 * 32])smem;
  const bf16_t* Wsh = sgpr_pin_ptr(Wsh_);
  const bf16_t* x = sgpr_pin_ptr(x_);
  const int B = sgpr_pin(B_), N = sgpr_pin(N_), K = sgpr_pin(K_), ksplit = sgpr_pin(ksplit_);
  const int rpb = sgpr_pin(rpb_ & 255);      // rows per n-block of the packed matrix (32 or 20: skinny_rows_per_block)
  const int xl = sgpr_pin(rpb_ >> 8);        // 1: activation tile through LDS (launcher: skinny_x_lds)
  const int steps = sgpr_pin(steps_);  // k-steps per wave = K / 16 / (NW * ksplit), divided on the host: an integer division here is
                                       // ~40 dependent instructions in front of the first load
  float* slab = sgpr_pin_ptr(slab_);
  const int64_t slab_stride = sgpr_pin(slab_stride_);
  e.bias = sgpr_pin_ptr(e.bias); e.residual = sgpr_pin_ptr(e.residual); e.out_f32 = sgpr_pin_ptr(e.out_f32);
  e.out_t = sgpr_pin_ptr(e.out_t); e.ldc = sgpr_pin(e.ldc); e.act = sgpr_pin(e.act);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = blockIdx.x, ks = blockIdx.y;
  const int ks_per = K / 16;
  const int k0 = (ks * NW + wave) * steps;
  // 16-byte chunk of lane `lane` in (n-block, k-step): [half][row]; lanes past the block's rows re-read its last row
  const int wstep = 2 * rpb;
  const u32x4* wp = (const u32x4*)Wsh + ((int64_t)nb * ks_per + k0) * wstep + (lane >> 5) * rpb + min(lane & 31, rpb - 1);
  const bf16_t* xp[RB];
#pragma unroll
  for (int g = 0; g < RB; ++g) xp[g] = x + (int64_t)min(g * 32 + (lane & 31), B - 1) * K + k0 * 16 + 8 * (lane >> 5);
  const int en = nb * rpb + 4 * (tid & 7);
  const bool ecell = tid < 256 && 4 * (tid & 7) < rpb;   // this thread owns 4 output columns of the block
  float4 ebias = make_float4(0.f, 0.f, 0.f, 0.f), eres[RB];
#pragma unroll
  for (int g = 0; g < RB; ++g) eres[g] = ebias;
  // epilogue operands of this thread's 4 cells per row group (b = 32g + tid>>3, n = nb*32 + 4*(tid&7) ..+3): requested
  // together with the fragments (after them: the weights are the long pole)
  auto prefetch_epilogue = [&]() {
    if (ecell && en + 3 < N) {
      if (e.bias && ksplit == 1) ebias = *(const float4*)(e.bias + en);
      if (e.residual && ksplit == 1) {
#pragma unroll
        for (int g = 0; g < RB; ++g) eres[g] = *(const float4*)(e.residual + (int64_t)min(g * 32 + (tid >> 3), B - 1) * e.ldc + en);
      }
    }
  };
  f32x16 acc[RB];
#pragma unroll
  for (int g = 0; g < RB; ++g)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[g][j] = 0.f;
  // Activation tile through LDS (round 6).  A fragment load straight from the row-major activations touches 32 lines for 32 B
  // each (row stride = K) - four times the line work of a weight load, every workgroup of the GEMM again for its k-range.  Here
  // the workgroup requests its tile (RB * 32 rows x NW * steps * 16 k) COALESCED - 2 NW threads cover consecutive 16-byte chunks
  // of a row, so 8 threads take one whole 128-byte line - BEFORE the weights (vmcnt retires in order: the tile can then be waited
  // for while the U weight loads fly on), parks it in LDS (rows padded by 16 B: ds_read_b128 of 8 consecutive rows covers all
  // banks) and reads the MFMA fragments back.  Same fragments, same MFMA order: bit-identical to the register form.  Measured
  // (NW = 4, one row group, large-v3 decode, 32 rows): 368.9 -> 357.3 ms per 128-token decode.
  constexpr bool XLC = ONE && (NW == 4 || (NW == 8 && RB == 1));
  if (XLC && xl) {
    if constexpr (XLC) {
    constexpr int TPR = 2 * NW;                       // threads per tile row (each pass of the workgroup covers 32 rows)
    const int xrow = tid / TPR, xcq = tid % TPR;
    const int stride = NW * steps * 32 + 16;          // bytes per LDS row
    u32x4 w[U], xr[RB][U], xv[RB][U];
#pragma unroll
    for (int g = 0; g < RB; ++g) {
      const bf16_t* xg = x + (int64_t)min(g * 32 + xrow, B - 1) * K + (int64_t)(ks * NW * steps) * 16 + xcq * 8;
#pragma unroll
      for (int u = 0; u < U; ++u) xr[g][u] = *(const u32x4*)(xg + min(u, steps - 1) * TPR * 8);
    }
    prefetch_epilogue();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(u, steps - 1);
      if constexpr (NT) w[u] = __builtin_nontemporal_load(wp + (int64_t)i * wstep); else w[u] = wp[(int64_t)i * wstep];
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U) : "memory");   // tile pieces (and epilogue operands) landed; the U weight loads fly on
#pragma unroll
    for (int g = 0; g < RB; ++g)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < steps) *(u32x4*)(smem + (g * 32 + xrow) * stride + (u * TPR + xcq) * 16) = xr[g][u];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int g = 0; g < RB; ++g)
#pragma unroll
      for (int u = 0; u < U; ++u)
        xv[g][u] = *(const u32x4*)(smem + (g * 32 + (lane & 31)) * stride + ((wave * steps + min(u, steps - 1)) * 2 + (lane >> 5)) * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // every fragment is in registers: the tile's LDS becomes the reduction buffer
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < steps) {
#pragma unroll
        for (int g = 0; g < RB; ++g)
          acc[g] = N16<T16>::mfma32(*(s16x8*)&w[u], *(s16x8*)&xv[g][u], acc[g]);
      }
    }
  } else if constexpr (ONE) {
    u32x4 w[U], xv[RB][U];
    // Issue order (round 6): ALL weight fragments first - they come from HBM and are the long pole -, then the activation
    // fragments row group by row group.  A lane's activation pieces of consecutive k-steps are 32 B apart in ONE 128-byte line
    // (row stride = K) and every wave-load touches 32 lines for 32 B each: issued k-step by k-step (W, X0 .. X3, W, ...) the
    // four touches of a line were (1 + RB) loads x 4 waves apart and the data the workgroup has in flight evicted it from the
    // CU's L1 in between; issued back to back they hit it.  Same arithmetic, bit-identical; measured per beam-5 position
    // 2.69 -> 2.63 ms at 40 rows, 4.15 -> 3.72 at 80, 5.02 -> 4.65 at 120, and the 32-row greedy decode 376.5 -> 370.1 ms.
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(u, steps - 1);  // clamped and unconditional: nothing branches around a load
      if constexpr (NT) w[u] = __builtin_nontemporal_load(wp + (int64_t)i * wstep); else w[u] = wp[(int64_t)i * wstep];
    }
#pragma unroll
    for (int g = 0; g < RB; ++g)
#pragma unroll
      for (int u = 0; u < U; ++u) xv[g][u] = *(const u32x4*)(xp[g] + min(u, steps - 1) * 16);
    prefetch_epilogue();
    __builtin_amdgcn_sched_barrier(0);  // every load is issued before the first MFMA waits: one round trip
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < steps) {
#pragma unroll
        for (int g = 0; g < RB; ++g)
          acc[g] = N16<T16>::mfma32(*(s16x8*)&w[u], *(s16x8*)&xv[g][u], acc[g]);
      }
  } else {
    prefetch_epilogue();
    for (int i0 = 0; i0 < steps; i0 += U) {
      u32x4 w[U], xv[RB][U];
#pragma unroll
      for (int u = 0; u < U; ++u) {   // weights first, then the activations row group by row group (see the straight-line form)
        const int i = min(i0 + u, steps - 1);
        if constexpr (NT) w[u] = __builtin_nontemporal_load(wp + (int64_t)i * wstep); else w[u] = wp[(int64_t)i * wstep];
      }
#pragma unroll
      for (int g = 0; g < RB; ++g)
#pragma unroll
        for (int u = 0; u < U; ++u) xv[g][u] = *(const u32x4*)(xp[g] + min(i0 + u, steps - 1) * 16);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u < steps) {
#pragma unroll
          for (int g = 0; g < RB; ++g)
            acc[g] = N16<T16>::mfma32(*(s16x8*)&w[u], *(s16x8*)&xv[g][u], acc[g]);
        }
    }
  }
  // D: col = lane & 31 = batch row b, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) = output n
#pragma unroll
  for (int g = 0; g < RB; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4*)&red[wave][g][(lane & 31) * 32 + 8 * q + 4 * (lane >> 5)] =
          make_float4(acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
  __syncthreads();
  if (!ecell) return;
#pragma unroll
  for (int g = 0; g < RB; ++g) {
    float4 v = *(const float4*)&red[0][g][(tid >> 3) * 32 + 4 * (tid & 7)];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float4 t = *(const float4*)&red[w][g][(tid >> 3) * 32 + 4 * (tid & 7)];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    v.x += ebias.x; v.y += ebias.y; v.z += ebias.z; v.w += ebias.w;
    const int b = g * 32 + (tid >> 3);
    if (b >= B) continue;
    float vv[4] = {v.x, v.y, v.z, v.w};
    const float rr[4] = {eres[g].x, eres[g].y, eres[g].z, eres[g].w};
    const int64_t idx0 = (int64_t)b * e.ldc + en;
    if (en + 3 < N && ksplit > 1) {  // K-split partial: one 16-byte store into this slice's slab
      *(float4*)(slab + (int64_t)ks * slab_stride + idx0) = v;
      continue;
    }
    if (en + 3 < N && ksplit == 1) {  // full 4-column cell: vector stores
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (e.act == 1) vv[j] = gelu_erf(vv[j]);
        if (e.residual) vv[j] += rr[j];
      }
      if (e.out_f32) *(float4*)(e.out_f32 + idx0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (e.out_t) {
        uint2 pk;
        pk.x = N16<T16>::pk(vv[0], vv[1]);
        pk.y = N16<T16>::pk(vv[2], vv[3]);
        *(uint2*)((bf16_t*)e.out_t + idx0) = pk;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = en + j;
      if (n < N) {
        const int64_t idx = idx0 + j;
        if (ksplit > 1) {
          slab[(int64_t)ks * slab_stride + idx] = vv[j];  // partial tile of K-slice ks (bias / residual: the following LayerNorm)
        } else {
          float o = vv[j];
          if (e.bias) o += e.bias[n];  // ragged tail block (vocabulary): operands were not prefetched
          if (e.act == 1) o = gelu_erf(o);
          if (e.residual) o += e.residual[idx];
          if (e.out_f32) e.out_f32[idx] = o;
          if (e.out_t) ((bf16_t*)e.out_t)[idx] = N16<T16>::down(o);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Vocabulary projection (round 3): logits[b][n] = sum_k x[b][k] * E[n][k], N = 51 866, K = d <= 1280, B <= 64 rows.
// The generic kernel above launches one workgroup per 32 outputs - 1 621 workgroups that EACH re-read the 82 KB of
// activation rows for their 82 KB of weights (40 us per step against 21 us for the 133 MB of weights alone).  Here a workgroup
// is PERSISTENT: 8 waves split K exactly as above (wave w owns k-steps [w * steps, (w + 1) * steps)), each wave loads its slice
// of the activation rows ONCE into registers (steps * RB * 4 VGPRs) and then walks n-blocks blockIdx.x, + gridDim.x, ... with
// the NEXT block's weight fragments in flight while the current block's MFMAs and the 8-wave LDS reduction run (two LDS
// buffers: one barrier per n-block).  Same per-wave accumulation order and the same fixed-order 8-wave sum as the generic
// kernel with NW = 8: the logits are bit-identical to it.
// ------------------------------------------------------------------------------------------------
template <typename T16, int RB, bool NT>
__global__ __launch_bounds__(512) void gemm_vocab_kernel(const bf16_t* __restrict__ Wsh_, const bf16_t* __restrict__ x_, int B_, int N_,
                                                         int K_, int steps_, float* __restrict__ out_, int64_t ldc_) {
  constexpr int NW = 8, U = 10;
  extern __shared__ __attribute__((aligned(16))) float vred[];   // [2][NW][RB][32 * 32]
  const bf16_t* Wsh = sgpr_pin_ptr(Wsh_);
  const bf16_t* x = sgpr_pin_ptr(x_);
  const int B = sgpr_pin(B_), N = sgpr_pin(N_), K = sgpr_pin(K_), steps = sgpr_pin(steps_);
  float* out = sgpr_pin_ptr(out_);
  const int64_t ldc = sgpr_pin(ldc_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks_per = K / 16, n_blocks = (N + 31) / 32;
  const int k0 = wave * steps;
  const u32x4* wbase = (const u32x4*)Wsh + (int64_t)k0 * 64 + lane;
  auto load_w = [&](int nb, u32x4 (&w)[U]) {
    const u32x4* wp = wbase + (int64_t)min(nb, n_blocks - 1) * ks_per * 64;   // clamped: the last prefetch re-reads a valid block
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(u, steps - 1);
      if constexpr (NT) w[u] = __builtin_nontemporal_load(wp + (int64_t)i * 64); else w[u] = wp[(int64_t)i * 64];
    }
  };
  int nb = blockIdx.x;
  u32x4 wa[U], wb[U];
  load_w(nb, wa);                                 // the weight stream starts before the activation rows are fetched
  u32x4 xv[RB][U];
#pragma unroll
  for (int g = 0; g < RB; ++g) {
    const bf16_t* xp = x + (int64_t)min(g * 32 + (lane & 31), B - 1) * K + k0 * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int u = 0; u < U; ++u) xv[g][u] = *(const u32x4*)(xp + min(u, steps - 1) * 16);
  }
  int par = 0;
  auto block = [&](int nb_cur, u32x4 (&wc)[U], u32x4 (&wn)[U]) {
    load_w(nb_cur + gridDim.x, wn);               // next block's fragments: in flight under this block's MFMAs + reduction
    f32x16 acc[RB];
#pragma unroll
    for (int g = 0; g < RB; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[g][j] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < steps) {
#pragma unroll
        for (int g = 0; g < RB; ++g) acc[g] = N16<T16>::mfma32(*(s16x8*)&wc[u], *(s16x8*)&xv[g][u], acc[g]);
      }
    float* red = vred + (size_t)par * NW * RB * 1024;
#pragma unroll
    for (int g = 0; g < RB; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(float4*)&red[(wave * RB + g) * 1024 + (lane & 31) * 32 + 8 * q + 4 * (lane >> 5)] =
            make_float4(acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
    __syncthreads();   // one barrier per block: the other LDS buffer is only rewritten after the NEXT barrier
    if (tid < 256) {
      const int en = nb_cur * 32 + 4 * (tid & 7);
#pragma unroll
      for (int g = 0; g < RB; ++g) {
        float4 v = *(const float4*)&red[(0 * RB + g) * 1024 + (tid >> 3) * 32 + 4 * (tid & 7)];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float4 t = *(const float4*)&red[(w * RB + g) * 1024 + (tid >> 3) * 32 + 4 * (tid & 7)];
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int b = g * 32 + (tid >> 3);
        if (b < B) {
          float* o = out + (int64_t)b * ldc + en;
          if (en + 3 < N) *(float4*)o = v;
          else { const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (en + j < N) o[j] = vv[j]; }
        }
      }
    }
    par ^= 1;
  };
  // two blocks per trip so that the register sets alternate with static names (a runtime-indexed array would go to scratch)
  while (nb < n_blocks) {
    block(nb, wa, wb);
    nb += gridDim.x;
    if (nb >= n_blocks) break;
    block(nb, wb, wa);
    nb += gridDim.x;
  }
}

// Per-device set-up, called from ttasr_create (NOT lazily from the launcher: the first launch of a shape happens inside a
// hipGraph stream capture, where attribute / property calls do not belong - under rocprofv3 they crashed the capture): the
// opt-in to > 64 KiB of dynamic LDS for every instantiation and the CU count that sizes the persistent grid.
static int g_vocab_cus[64] = {0};
static std::once_flag g_vocab_once[64];
static void vocab_init_once(int device);
void gemm_vocab_init(int device) { std::call_once(g_vocab_once[device & 63], vocab_init_once, device); }
static void vocab_init_once(int device) {
  hipDeviceProp_t p;
  g_vocab_cus[device & 63] = hipGetDeviceProperties(&p, device) == hipSuccess ? p.multiProcessorCount : 256;
#define TTASR_VOCAB_ATTR(T_, RB_)                                                                                                          \
  hipFuncSetAttribute((const void*)gemm_vocab_kernel<T_, RB_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * RB_ * 4096);     \
  hipFuncSetAttribute((const void*)gemm_vocab_kernel<T_, RB_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * RB_ * 4096)
  TTASR_VOCAB_ATTR(bf16_t, 1); TTASR_VOCAB_ATTR(bf16_t, 2); TTASR_VOCAB_ATTR(f16_t, 1); TTASR_VOCAB_ATTR(f16_t, 2);
#undef TTASR_VOCAB_ATTR
}

// Returns false when the shape does not fit (caller falls back to launch_gemm_skinny).
template <typename T16>
bool launch_gemm_vocab(const T16* Wsh, const T16* x, int B, int N, int K, float* out, int64_t ldc, hipStream_t s, int device) {
  const int n_cu = g_vocab_cus[device & 63];
  if (n_cu <= 0 || B < 1 || B > 64 || K % 128 != 0 || K / 128 > 10 || N < 8192 || ldc % 4 != 0) return false;
  // this kernel walks 32-row n-blocks (64 chunks per k-step); a vocabulary whose size makes skinny_rows_per_block choose the
  // 20-row layout (V % 5120 == 0: 10 240, 51 200 ...) was packed that way by build_weights and belongs to the generic kernel,
  // which takes the block height as an argument (read as 32-row blocks the logits were silently wrong)
  if (skinny_rows_per_block(N, K) != 32) return false;
  const int rb = (B + 31) / 32, steps = K / 128, n_blocks = (N + 31) / 32;
  const int grid = n_blocks < n_cu ? n_blocks : n_cu;
  const size_t lds = (size_t)2 * 8 * rb * 1024 * sizeof(float);   // 64 KiB per row group
#define TTASR_VOCAB(RB_)                                                                                                           \
  do {                                                                                                                             \
    if (g_skinny_nt) hipLaunchKernelGGL((gemm_vocab_kernel<T16, RB_, true>), dim3(grid), dim3(512), lds, s, (const bf16_t*)Wsh, (const bf16_t*)x, B, N, K, steps, out, ldc); \
    else hipLaunchKernelGGL((gemm_vocab_kernel<T16, RB_, false>), dim3(grid), dim3(512), lds, s, (const bf16_t*)Wsh, (const bf16_t*)x, B, N, K, steps, out, ldc); \
  } while (0)
  if (rb == 1) TTASR_VOCAB(1); else TTASR_VOCAB(2);
#undef TTASR_VOCAB
  return true;
}
template bool launch_gemm_vocab<bf16_t>(const bf16_t*, const bf16_t*, int, int, int, float*, int64_t, hipStream_t, int);
template bool launch_gemm_vocab<f16_t>(const f16_t*, const f16_t*, int, int, int, float*, int64_t, hipStream_t, int);

// K slices a split decode GEMM is cut into (1 = unsplit).  `want` = requested slice count (0 = automatic: weights in
// pieces of <= ~20 KB per workgroup, a few hundred workgroups); the result divides the k-steps evenly over 4 waves.
int gemm_skinny_ksplit(int B, int N, int K, int want) {
  if (B < 1 || B > 128 || K % 64 != 0) return 1;
  const int rpb = skinny_rows_per_block(N, K);
  const int rb = (B + 31) / 32, n_blocks = (N + rpb - 1) / rpb, ks_per = K / 16;
  const int per4 = ks_per / 4;  // k-steps per wave of a 4-wave workgroup when unsplit
  if (ks_per % 4 != 0) return 1;
  int best = 1;
  if (want > 0) {  // largest divisor of per4 that is <= want
    for (int s = 1; s <= want && s <= 16; ++s) if (per4 % s == 0) best = s;
    return best;
  }
  const int u_max = rb == 1 ? 10 : (rb == 2 ? 8 : 5);
  for (int s = 1; s <= 16; ++s) {
    if (per4 % s != 0) continue;
    best = s;
    const int steps = per4 / s;
    // one round trip per wave, and either every CU busy or <= 5 KiB of weights per wave on >= 160 workgroups
    if (steps <= u_max && ((steps <= 10 && n_blocks * s >= 256) || (steps <= 5 && n_blocks * s >= 160))) break;
  }
  return best;
}

// Chooses the waves per workgroup so that each wave owns <= 10 k-steps (one round trip) where the shape allows.
// ksplit > 1 (gemm_skinny_ksplit): partial tiles go to `slab`, bias / activation / residual are the consumer's job.
// Returns false when the shape does not fit (caller falls back to gemm_basic).
template <typename T16>
bool launch_gemm_skinny(const T16* Wsh_, const T16* x_, int B, int N, int K, const GemmEpi& e, hipStream_t s, int ksplit,
                        float* slab, int64_t slab_stride) {
  const bf16_t* Wsh = (const bf16_t*)Wsh_;   // raw 16-bit words: the kernel only moves them; T16 picks the MFMA form
  const bf16_t* x = (const bf16_t*)x_;
  if (B < 1 || B > 128 || K % 64 != 0 || e.rowtab || e.headsplit) return false;
  const int rb = (B + 31) / 32;  // 32-row groups sharing one weight stream
  const int rpb = skinny_rows_per_block(N, K);
  const int n_blocks = (N + rpb - 1) / rpb;
  const int ks_per = K / 16;
  const int u_max = rb == 1 ? 10 : (rb == 2 ? 8 : 5);  // k-steps a wave can keep in flight (registers)
  int nw = 4;
  if (ksplit > 1) {
    if (!slab || e.act != 0) return false;
  } else {
    ksplit = 1;
    const int nw_max = rb == 1 ? 16 : 8;  // 16 waves leave 128 VGPRs per lane: only the single row group fits
    while (nw < nw_max && ks_per % (nw * 2) == 0 && ks_per / nw > u_max) nw *= 2;
    if (nw < 8 && ks_per % 8 == 0 && ks_per / 8 >= 5 && n_blocks < 256) nw = 8;
  }
  if (ks_per % (nw * ksplit) != 0) return false;
  const int steps = ks_per / (nw * ksplit);
  dim3 grid(n_blocks, ksplit);
#define TTASR_SKINNY(NW_, RB_, U_, ONE_)                                                                                              \
  do {                                                                                                                                \
    const bool xl_ = g_skinny_x_lds && ONE_ && (NW_ == 4 || (NW_ == 8 && RB_ == 1));                                                  \
    const size_t lds_ = skinny_lds_bytes(NW_, RB_, steps, xl_);                                                                       \
    const int rpbx_ = rpb | (xl_ ? 256 : 0);                                                                                          \
    if (g_skinny_nt) {                                                                                                                \
      if (lds_ > 65536) skinny_allow_big_lds<gemm_skinny_kernel<T16, NW_, RB_, U_, true, ONE_>>();                                     \
      hipLaunchKernelGGL((gemm_skinny_kernel<T16, NW_, RB_, U_, true, ONE_>), grid, dim3(NW_ * 64), lds_, s, Wsh, x, B, N, K, ksplit, steps, e, slab, slab_stride, rpbx_); \
    } else {                                                                                                                          \
      if (lds_ > 65536) skinny_allow_big_lds<gemm_skinny_kernel<T16, NW_, RB_, U_, false, ONE_>>();                                    \
      hipLaunchKernelGGL((gemm_skinny_kernel<T16, NW_, RB_, U_, false, ONE_>), grid, dim3(NW_ * 64), lds_, s, Wsh, x, B, N, K, ksplit, steps, e, slab, slab_stride, rpbx_); \
    }                                                                                                                                 \
  } while (0)
#define TTASR_SKINNY_U(NW_, RB_, UMAX_)                                            \
  do {                                                                             \
    if (steps <= 2 && UMAX_ >= 2) TTASR_SKINNY(NW_, RB_, 2, true);                 \
    else if (steps <= 4 && UMAX_ >= 4) TTASR_SKINNY(NW_, RB_, 4, true);            \
    else if (steps <= 5) TTASR_SKINNY(NW_, RB_, 5, true);                          \
    else if (steps <= 8 && UMAX_ >= 8) TTASR_SKINNY(NW_, RB_, 8, true);            \
    else if (steps <= UMAX_) TTASR_SKINNY(NW_, RB_, UMAX_, true);                  \
    else TTASR_SKINNY(NW_, RB_, UMAX_, false);                                     \
  } while (0)
  if (rb == 1) {
    if (nw == 16) TTASR_SKINNY_U(16, 1, 10); else if (nw == 8) TTASR_SKINNY_U(8, 1, 10); else TTASR_SKINNY_U(4, 1, 10);
  } else if (rb == 2) {
    // 8-wave workgroups run 2 waves per SIMD (256 VGPRs each): the unsplit K = 1280 GEMMs (fc1, qkv of a prefill pass: 10 k-steps
    // per wave) keep all 10 in flight in the straight-line form - the looped form cost 13.2 us against 5.4 us at one row group
    if (nw == 8) TTASR_SKINNY_U(8, 2, 10); else TTASR_SKINNY_U(4, 2, 8);
  } else if (rb == 3) {
    if (nw == 8) TTASR_SKINNY_U(8, 3, 10); else TTASR_SKINNY_U(4, 3, 5);
  } else {
    if (nw == 8) TTASR_SKINNY_U(8, 4, 5); else TTASR_SKINNY_U(4, 4, 5);
  }
#undef TTASR_SKINNY_U
#undef TTASR_SKINNY
  return true;
}
template bool launch_gemm_skinny<bf16_t>(const bf16_t*, const bf16_t*, int, int, int, const GemmEpi&, hipStream_t, int, float*, int64_t);
template bool launch_gemm_skinny<f16_t>(const f16_t*, const f16_t*, int, int, int, const GemmEpi&, hipStream_t, int, float*, int64_t);
