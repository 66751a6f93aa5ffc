// Encoder self-attention, bf16 MFMA flash kernel for gfx950 (head_dim 64, no mask, T = 1500).
// Replaces CTranslate2's MultiHeadAttention (QK^T / softmax / PV as three cuBLAS+CUDA launches) with one
// fused pass; arithmetic per HF modeling_whisper.py:215-238 (q pre-scaled, softmax in f32).
//
// Structure (wave64, v_mfma_f32_32x32x16_bf16):
//   workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
//   K/V tiles of 64 keys are register-staged into a double-buffered LDS image (issue the global loads
//   for tile t+1, compute tile t, then write the registers to the other buffer: guide T14).
//   S^T = K Q^T  (keys on the MFMA rows, the query on the lane) so that each lane owns ONE query column:
//   the row max / row sum are in-lane reductions over registers plus one cross-half shuffle, and the
//   f32 accumulator converts in place into the B operand of O^T += V^T P^T (guide section 3, "An
//   accumulator tile as the next MFMA's operand").  V stays row-major [key][d] in LDS and its A operand
//   (4 consecutive keys of one d) comes from ds_read_b64_tr_b16 (guide T10).
//   LDS swizzles: K chunk c of row k at c ^ ((k>>1)&7); V 64-byte half hf of row k at hf ^ ((k>>1)&1).
#include "common.hpp"
#include <type_traits>

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
}

constexpr int FA_QB = 128, FA_KB = 64;

// CROSS = false: encoder self-attention, q / k / v rows interleaved in one [B][Tn][3d] tensor (n_k == Tn).
// CROSS = true: the decoder's cross-attention for MANY query rows per clip (a prompt prefill pass): q = [clip][Tn][d] rows, K / V
// from the cross-KV cache [clip][head][n_k][64] - the clip's frames are streamed once for all of its prompt positions.
template <typename T16, bool CROSS>
__global__ __launch_bounds__(256) void enc_attn_flash_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int Tn,
                                                             int H, const bf16_t* __restrict__ kx, const bf16_t* __restrict__ vx,
                                                             int n_k_cross) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (K 8 KiB | V 8 KiB); reused for the O transpose
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD whole (batch, head)
  // pairs: its q-blocks then re-read the same K/V from that XCD's L2 instead of each XCD fetching every K/V
  // from HBM (measured 2.1 GB fetched per launch against 0.37 GB of q/k/v with the naive order).
  const int nq = gridDim.x, n_bh = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + nq * (blockIdx.y + gridDim.y * blockIdx.z);
  int qb = blockIdx.x, bh = blockIdx.y + gridDim.y * blockIdx.z;
  if ((n_bh & 7) == 0) {
    const int xcd = lin & 7, slot = lin >> 3;
    qb = slot % nq;
    bh = (slot / nq) * 8 + xcd;
  }
  const int b = bh / H, h = bh - b * H, q0 = qb * FA_QB + wave * 32;
  const int d = H * 64;
  const int64_t ld = CROSS ? (int64_t)d : 3 * (int64_t)d;                      // query row stride
  const bf16_t* base = qkv + (int64_t)b * Tn * ld + h * 64;                     // query rows of (b, h)
  const int n_k = CROSS ? n_k_cross : Tn;                                       // keys
  const int64_t kld = CROSS ? 64 : ld;                                          // key / value row stride
  const bf16_t* kbase = CROSS ? kx + ((int64_t)b * H + h) * n_k * 64 : base + d;
  const bf16_t* vbase = CROSS ? vx + ((int64_t)b * H + h) * n_k * 64 : base + 2 * d;

  // Q fragments (B operand of S^T): lane holds Q[q0 + r][16*ks + 8*hf .. +8]
  s16x8 qf[4];
  {
    const bf16_t* qp = base + (int64_t)min(q0 + r, Tn - 1) * ld + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const s16x8*)(qp + 16 * ks);
  }

  // staging map: thread -> (row, chunk) x 2 for K and V
  const int srow = tid >> 2, sc0 = (tid & 3) * 2;  // rows 0..63, chunks {sc0, sc0+1}
  uint4 kreg0, kreg1, vreg0, vreg1;
#define FA_G_LOAD(kt_)                                                      \
  do {                                                                      \
    const int key_ = min((kt_) * FA_KB + srow, n_k - 1);                    \
    const bf16_t* kp_ = kbase + (int64_t)key_ * kld + sc0 * 8;              \
    const bf16_t* vp_ = vbase + (int64_t)key_ * kld + sc0 * 8;              \
    kreg0 = *(const uint4*)kp_; kreg1 = *(const uint4*)(kp_ + 8);           \
    vreg0 = *(const uint4*)vp_; vreg1 = *(const uint4*)(vp_ + 8);           \
  } while (0)
  // K: chunk c of row k at c ^ ((k>>1)&7); V: 64-byte half (c >> 2) swapped by bit 1 of the row
#define FA_S_STORE(buf_)                                                                                   \
  do {                                                                                                     \
    char* Kb_ = smem + (buf_) * 16384;                                                                     \
    char* Vb_ = Kb_ + 8192;                                                                                \
    const int sw_ = (srow >> 1) & 7;                                                                       \
    *(uint4*)(Kb_ + srow * 128 + ((sc0 ^ sw_) << 4)) = kreg0;                                              \
    *(uint4*)(Kb_ + srow * 128 + (((sc0 + 1) ^ sw_) << 4)) = kreg1;                                        \
    *(uint4*)(Vb_ + srow * 128 + ((((sc0 >> 2) ^ (sw_ & 1)) << 6) | ((sc0 & 3) << 4))) = vreg0;            \
    *(uint4*)(Vb_ + srow * 128 + (((((sc0 + 1) >> 2) ^ (sw_ & 1)) << 6) | (((sc0 + 1) & 3) << 4))) = vreg1; \
  } while (0)

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  constexpr float LOG2E = 1.4426950408889634f;

  const int n_tiles = (n_k + FA_KB - 1) / FA_KB;
  FA_G_LOAD(0);
  FA_S_STORE(0);
  __syncthreads();

  // per-lane constant parts of the LDS read addresses
  //   K fragment (kb2, ks): row = 32*kb2 + r, chunk = 2*ks + hf
  //   V tr-read (s, db, part): row = 16*s + 8*part + 4*hf + ((lane & 15) >> 2); byte = 64*db' + 32*((lane>>4)&1) + 8*(lane&3)
  const int vq = (lane & 15) >> 2, vp4 = lane & 3, vg = (lane >> 4) & 1;

  auto tile = [&](const int kt, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const int cur = kt & 1;
    if (!LAST) FA_G_LOAD(kt + 1);
    const char* Kb = smem + cur * 16384;
    const char* Vb = Kb + 8192;

    // ---- S^T = K Q^T : two 32-key blocks ----
    f32x16 s[2];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s[kb2][j] = 0.f;
      const int krow = 32 * kb2 + r;
      const int ksw = (krow >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s16x8 kf = *(const s16x8*)(Kb + krow * 128 + (((2 * ks + hf) ^ ksw) << 4));
        s[kb2] = N16<T16>::mfma32(kf, qf[ks], s[kb2]);
      }
    }
    if (LAST) {  // mask keys past the end of the sequence (only the peeled last tile carries this code)
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          int key = kt * FA_KB + 32 * kb2 + (j & 3) + 8 * (j >> 2) + 4 * hf;
          if (key >= n_k) s[kb2][j] = -1e30f;
        }
    }
    // ---- online softmax: this lane's query column ----
    float tmax = s[0][0];
#pragma unroll
    for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, s[0][j]);
#pragma unroll
    for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, s[1][j]);
    tmax = xor32_reduce(tmax, OpMax{});  // v_permlane32_swap: no LDS round trip in the tile loop
    const float m_new = fmaxf(m_run, tmax);
    const float mb = -m_new * LOG2E;
    float psum = 0.f;
    uint32_t pf[2][8];  // bf16-packed P^T: [kb2][2*s' + pair]
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        float p0 = __builtin_amdgcn_exp2f(fmaf(s[kb2][j], LOG2E, mb));
        float p1 = __builtin_amdgcn_exp2f(fmaf(s[kb2][j + 1], LOG2E, mb));
        psum += p0 + p1;
        pf[kb2][j >> 1] = N16<T16>::pk(p0, p1);
      }
    // rescale the running sums only when some query of this wave saw a new maximum (exact: alpha == 1 otherwise);
    // after the first few tiles this is rare, and it keeps 32 multiplies + an exp out of the steady-state loop
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] *= alpha;
      m_run = m_new;
    }
    l_run += psum;

    // ---- O^T += V^T P^T : 4 k-steps of 16 keys, 2 d-blocks ----
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int kb2 = ss >> 1, sp = ss & 1;
      s16x8 pb;
      {
        uint4 t = make_uint4(pf[kb2][4 * sp + 0], pf[kb2][4 * sp + 1], pf[kb2][4 * sp + 2], pf[kb2][4 * sp + 3]);
        pb = __builtin_bit_cast(s16x8, t);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        s16x4_t lo, hi;
        {
          const int row0 = 16 * ss + 4 * hf + vq;
          const int row1 = row0 + 8;
          lo = lds_tr16(Vb + row0 * 128 + (((db ^ ((row0 >> 1) & 1)) << 6) | (vg << 5) | (vp4 << 3)));
          hi = lds_tr16(Vb + row1 * 128 + (((db ^ ((row1 >> 1) & 1)) << 6) | (vg << 5) | (vp4 << 3)));
        }
        s16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[db] = N16<T16>::mfma32(vf, pb, o[db]);
      }
    }
    if (!LAST) FA_S_STORE(cur ^ 1);
    __syncthreads();
  };
  for (int kt = 0; kt + 1 < n_tiles; ++kt) tile(kt, std::false_type{});
  tile(n_tiles - 1, std::true_type{});

  // ---- epilogue: O^T[d][q] / l  ->  out[q][h*64 + d], transposed through LDS so rows leave as 128 B ----
  const float l_tot = xor32_reduce(l_run, OpSum{});
  const float inv = 1.0f / l_tot;
  char* ob = smem + wave * (32 * 144);  // [32 q][64 d] bf16, row stride 144 B (128 + 16 pad)
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      uint2 pk;
      pk.x = N16<T16>::pk(o[db][4 * rg + 0] * inv, o[db][4 * rg + 1] * inv);
      pk.y = N16<T16>::pk(o[db][4 * rg + 2] * inv, o[db][4 * rg + 3] * inv);
      const int dcol = 32 * db + 8 * rg + 4 * hf;
      *(uint2*)(ob + r * 144 + dcol * 2) = pk;
    }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int id = it * 64 + lane, row = id >> 3, c = id & 7;
    if (q0 + row < Tn) {
      uint4 v = *(const uint4*)(ob + row * 144 + c * 16);
      *(uint4*)(out + ((int64_t)b * Tn + q0 + row) * d + h * 64 + c * 8) = v;
    }
  }
}

template <typename T16>
void launch_enc_attn_flash_bf16(const T16* qkv, T16* out, int B, int T_, int H, hipStream_t s) {
  dim3 grid((T_ + FA_QB - 1) / FA_QB, H, B);
  hipLaunchKernelGGL((enc_attn_flash_kernel<T16, false>), grid, dim3(256), 32768, s, (const bf16_t*)qkv, (bf16_t*)out, T_, H,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0);
}
// q = [n_clips][n_q][d] (the n_q rows of a clip are consecutive), K / V = cross-KV cache of those clips, out like q
template <typename T16>
void launch_cross_attn_flash_bf16(const T16* q, const T16* K, const T16* V, T16* out, int n_clips, int n_q, int H, int Tk, hipStream_t s) {
  dim3 grid((n_q + FA_QB - 1) / FA_QB, H, n_clips);
  hipLaunchKernelGGL((enc_attn_flash_kernel<T16, true>), grid, dim3(256), 32768, s, (const bf16_t*)q, (bf16_t*)out, n_q, H,
                     (const bf16_t*)K, (const bf16_t*)V, Tk);
}
template void launch_enc_attn_flash_bf16<bf16_t>(const bf16_t*, bf16_t*, int, int, int, hipStream_t);
template void launch_enc_attn_flash_bf16<f16_t>(const f16_t*, f16_t*, int, int, int, hipStream_t);
template void launch_cross_attn_flash_bf16<bf16_t>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, hipStream_t);
template void launch_cross_attn_flash_bf16<f16_t>(const f16_t*, const f16_t*, const f16_t*, f16_t*, int, int, int, int, hipStream_t);
