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

constexpr int FA_KB = 64;   // keys per tile; a workgroup = 4 waves x QW blocks of 32 queries (QW = 1: 128 queries, QW = 2: 256)

// CROSS = false: encoder self-attention, q / k / v rows interleaved in one [B][Tn][3d] tensor (n_k == Tn).
// CROSS = true: the decoder's cross-attention for MANY query rows per clip (a prompt prefill pass): q = [clip][Tn][d] rows, K / V
// from the cross-KV cache [clip][head][n_k][64] - the clip's frames are streamed once for all of its prompt positions.
// QW = query blocks of 32 per WAVE (round 6).  With one block (rounds 1-5) every wave re-reads the whole 16 KB K / V tile from LDS
// for its 32 queries: 16 KB per wave and tile = 64 cycles of the LDS array at ds_read_b128's 256 B per cycle, so a round of four
// wave-tiles (one per SIMD) keeps the one LDS port of the CU busy for 256 cycles against the 512 cycles its 16 MFMAs of 32 cycles
// take on each SIMD - half the matrix time, before bank conflicts and the LDS-DMA landing of the next tile.  With TWO blocks a wave
// feeds every K / V fragment it reads to two MFMAs (LDS bytes per flop halve), and it owns two independent softmax chains.
// Measured: bit-identical, 13.87 -> 13.55 ms in situ (profiles/r6_flash_qw.jsonl) - a few per cent, because the kernel's first
// limit is vector issue (two v_exp_f32 per MFMA at head_dim 64: DESIGN.md 4.11), not the LDS port.
template <typename T16, bool CROSS, int QW>
__global__ __launch_bounds__(256, QW == 2 ? 2 : 3) void enc_attn_flash_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int Tn,
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
  const int b = bh / H, h = bh - b * H, q0 = qb * (128 * QW) + wave * (32 * QW);   // this wave's first query; block w: q0 + 32 w
  const int d = H * 64;
  const int64_t ld = CROSS ? (int64_t)d : 3 * (int64_t)d;                      // query row stride
  const bf16_t* base = qkv + (int64_t)b * Tn * ld + h * 64;                     // query rows of (b, h)
  const int n_k = CROSS ? n_k_cross : Tn;                                       // keys
  const int64_t kld = CROSS ? 64 : ld;                                          // key / value row stride
  const bf16_t* kbase = CROSS ? kx + ((int64_t)b * H + h) * n_k * 64 : base + d;
  const bf16_t* vbase = CROSS ? vx + ((int64_t)b * H + h) * n_k * 64 : base + 2 * d;

  // Q fragments (B operand of S^T): lane holds Q[q0 + r][16*ks + 8*hf .. +8], multiplied by log2(e) ONCE per workgroup
  // (round 5): the scores then leave the MFMA in exp2 units and the per-element multiply before v_exp_f32 disappears.  One more
  // rounding of q to the storage type (relative 2^-9 / 2^-11, random per element like the rounding q already carries from
  // the qkv GEMM's epilogue); the exact 1/8 pre-scaling stays in the weights.
  constexpr float LOG2E = 1.4426950408889634f;
  s16x8 qf[QW][4];
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    const bf16_t* qp = base + (int64_t)min(q0 + 32 * w + r, Tn - 1) * ld + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 t = *(const uint4*)(qp + 16 * ks);
      float v[8];
      up8<T16>(t, v);
      t.x = N16<T16>::pk(v[0] * LOG2E, v[1] * LOG2E); t.y = N16<T16>::pk(v[2] * LOG2E, v[3] * LOG2E);
      t.z = N16<T16>::pk(v[4] * LOG2E, v[5] * LOG2E); t.w = N16<T16>::pk(v[6] * LOG2E, v[7] * LOG2E);
      qf[w][ks] = __builtin_bit_cast(s16x8, t);
    }
  }
  // a wave whose 32 queries all lie past the end of the sequence (the last q-block of T = 1500 carries 92 queries: wave 3 has
  // none) stages its share of K / V and keeps the barriers, but runs no MFMA / softmax
  const bool live = q0 < Tn;

  // K / V staging by LDS-DMA (round 5): a wave-instruction of global_load_lds moves 1 KiB = 8 rows x 128 B straight into the LDS
  // image (no VGPR staging, no ds_write).  A tile is 8 K pieces + 8 V pieces; wave w issues K pieces 2w, 2w+1 and V pieces 2w,
  // 2w+1.  Lane l of piece p lands in row 8p + l/8, 16-byte slot l%8, so it FETCHES the chunk that belongs in that slot under
  // the image's swizzle (K: chunk c of row k sits in slot c ^ ((k>>1)&7); V: slot c ^ (((k>>1)&1) << 2)): source-side swizzle.
  // The per-lane source offsets are loop constants (32 bit); the tile base advances on the scalar unit.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int prow = lane >> 3, pslot = lane & 7;
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (2 * wave_u + p) * 8 + prow;
    koff[p] = (uint32_t)(row * (int)kld + ((pslot ^ ((row >> 1) & 7)) << 3)) * 2u;
    voff[p] = (uint32_t)(row * (int)kld + ((pslot ^ (((row >> 1) & 1) << 2)) << 3)) * 2u;
  }
  const int64_t tstep = (int64_t)FA_KB * kld * 2;              // bytes per tile
  const int full_tiles = n_k / FA_KB;                          // tiles 0 .. full_tiles-1 need no clamp
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
#define FA_DMA(kt_, buf_)                                                                                          \
  do {                                                                                                             \
    char* Kb_ = smem + (buf_) * 16384 + (2 * wave_u) * 1024;                                                       \
    char* Vb_ = Kb_ + 8192;                                                                                        \
    if ((kt_) < full_tiles) {                                                                                      \
      const char* kt_base_ = (const char*)kbase + (int64_t)(kt_) * tstep;                                          \
      const char* vt_base_ = (const char*)vbase + (int64_t)(kt_) * tstep;                                          \
      _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                              \
        __builtin_amdgcn_global_load_lds((gptr_t)(kt_base_ + koff[p]), (lptr_t)(Kb_ + p * 1024), 16, 0, 0);        \
        __builtin_amdgcn_global_load_lds((gptr_t)(vt_base_ + voff[p]), (lptr_t)(Vb_ + p * 1024), 16, 0, 0);        \
      }                                                                                                            \
    } else {                                                                                                       \
      _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                              \
        const int row_ = (2 * wave_u + p) * 8 + prow;                                                              \
        const int key_ = min((kt_) * FA_KB + row_, n_k - 1);                                                       \
        const bf16_t* kp_ = kbase + (int64_t)key_ * kld + ((pslot ^ ((row_ >> 1) & 7)) << 3);                      \
        const bf16_t* vp_ = vbase + (int64_t)key_ * kld + ((pslot ^ (((row_ >> 1) & 1) << 2)) << 3);               \
        __builtin_amdgcn_global_load_lds((gptr_t)kp_, (lptr_t)(Kb_ + p * 1024), 16, 0, 0);                         \
        __builtin_amdgcn_global_load_lds((gptr_t)vp_, (lptr_t)(Vb_ + p * 1024), 16, 0, 0);                         \
      }                                                                                                            \
    }                                                                                                              \
  } while (0)

  f32x16 o[QW][2];
#pragma unroll
  for (int w = 0; w < QW; ++w)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) o[w][i][j] = 0.f;
  // Online softmax with the reference maximum INSIDE the score MFMA (round 5): the accumulator of S^T = K Q^T starts at -m_ref
  // (sinit: 16 registers holding this lane's query's -m_ref), so the MFMA delivers s - m_ref and p = exp2(s - m_ref) costs one
  // v_exp_f32 per element and nothing else.  m_ref follows the running maximum lazily (guide T13): it moves only when a tile's
  // maximum exceeds it by more than FA_THR (in exp2 units: p <= 2^FA_THR = 64, well inside bf16 / fp16), and the first tile
  // always sets it.  The decision is taken BEFORE the tile's P is exponentiated and scales o, l and this tile's scores exactly
  // once (T13's hazard), so the result is the exact softmax whatever m_ref is; what changes against the running-maximum form is
  // rounding only (the largest p of a row is no longer exactly 1).
  constexpr float FA_THR = 6.0f;
  float m_ref[QW], l_run[QW];   // (the MFMA accumulator's start value -m_ref is splat from m_ref at every tile: 16 moves
#pragma unroll                 //  per block and tile instead of 16 registers per block held for the whole kernel)
  for (int w = 0; w < QW; ++w) { m_ref[w] = 0.f; l_run[w] = 0.f; }

  const int n_tiles = (n_k + FA_KB - 1) / FA_KB;
  FA_DMA(0, 0);
  // explicit, not left to the compiler's lowering of __syncthreads: a wave with no live query reads no LDS
  // itself, so nothing else forces ITS LDS-DMA pieces to have landed before the others pass the barrier and read them
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // every wave's own pieces landed + barrier

  // per-lane constant parts of the LDS read addresses
  //   K fragment (kb2, ks): row = 32*kb2 + r, chunk = 2*ks + hf
  //   V tr-read (s, db, part): row = 16*s + 8*part + 4*hf + ((lane & 15) >> 2); byte = 64*db' + 32*((lane>>4)&1) + 8*(lane&3)
  const int vq = (lane & 15) >> 2, vp4 = lane & 3, vg = (lane >> 4) & 1;

  // cur_tag: which LDS image this tile reads (compile-time: every LDS address is base + immediate); last_tag: the peeled last tile
  auto tile = [&](const int kt, auto cur_tag, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    constexpr int cur = decltype(cur_tag)::value;
    // tile kt + 1 into the other image: its last readers (tile kt - 1) are behind the barrier every wave has passed
    if (!LAST) FA_DMA(kt + 1, cur ^ 1);
    const char* Kb = smem + cur * 16384;
    const char* Vb = Kb + 8192;
    if (live) {
    // ---- S^T - m_ref = K Q^T - m_ref : two 32-key blocks, every K fragment read ONCE for the wave's QW query blocks ----
    f32x16 s[QW][2];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      f32x16 si;
#pragma unroll
      for (int j = 0; j < 16; ++j) si[j] = -m_ref[w];
      s[w][0] = si; s[w][1] = si;
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const int krow = 32 * kb2 + r;
      const int ksw = (krow >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s16x8 kf = *(const s16x8*)(Kb + krow * 128 + (((2 * ks + hf) ^ ksw) << 4));
#pragma unroll
        for (int w = 0; w < QW; ++w) s[w][kb2] = N16<T16>::mfma32(kf, qf[w][ks], s[w][kb2]);
      }
    }
    uint32_t pf[QW][2][8];  // 16-bit-packed P^T: [block][kb2][2*s' + pair]
#pragma unroll
    for (int w = 0; w < QW; ++w) {   // block w's softmax VALU work runs while block w + 1's score MFMAs are still in flight
      if (LAST) {  // mask keys past the end of the sequence (only the peeled last tile carries this code)
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            int key = kt * FA_KB + 32 * kb2 + (j & 3) + 8 * (j >> 2) + 4 * hf;
            if (key >= n_k) s[w][kb2][j] = -1e30f;
          }
      }
      // ---- this lane's query column: how far above the reference is the tile's maximum? ----
      float tmax = s[w][0][0];
#pragma unroll
      for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, s[w][0][j]);
#pragma unroll
      for (int j = 0; j < 16; ++j) tmax = fmaxf(tmax, s[w][1][j]);
      tmax = xor32_reduce(tmax, OpMax{});  // v_permlane32_swap: no LDS round trip in the tile loop
      const bool move = (kt == 0) | (tmax > FA_THR);
      if (__builtin_amdgcn_ballot_w64(move) != 0) {   // rare after the first tile: move the reference of the queries that need it
        const float delta = move ? tmax : 0.f;       // exp2(0) = 1 exactly for the others
        // First tile: o and l are still 0, only the reference moves - and the rescale factor must not be evaluated: a first-tile
        // maximum below about -128 exp2 units (a large negative per-query score offset, e.g. q bias x mean key of a trained
        // checkpoint) would make exp2(-tmax) = +inf and 0 * inf = NaN for the whole query row.
        const float alpha = kt == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        l_run[w] *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 16; ++j) { o[w][i][j] *= alpha; s[w][i][j] -= delta; }
        m_ref[w] += delta;
      }
      float psum = 0.f;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const float p0 = __builtin_amdgcn_exp2f(s[w][kb2][j]);
          const float p1 = __builtin_amdgcn_exp2f(s[w][kb2][j + 1]);
          psum += p0 + p1;
          pf[w][kb2][j >> 1] = N16<T16>::pk(p0, p1);
        }
      l_run[w] += psum;
    }

    // ---- O^T += V^T P^T : 4 k-steps of 16 keys, 2 d-blocks; every V fragment read ONCE for the QW blocks ----
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      const int kb2 = ss >> 1, sp = ss & 1;
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
#pragma unroll
        for (int w = 0; w < QW; ++w) {
          const uint4 t = make_uint4(pf[w][kb2][4 * sp + 0], pf[w][kb2][4 * sp + 1], pf[w][kb2][4 * sp + 2], pf[w][kb2][4 * sp + 3]);
          o[w][db] = N16<T16>::mfma32(vf, __builtin_bit_cast(s16x8, t), o[w][db]);
        }
      }
    }
    }  // live
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt + 1 landed (explicit: see above)
    __syncthreads();   // barrier: everybody's did
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 2 < n_tiles; kt += 2) {
    tile(kt, B0{}, std::false_type{});
    tile(kt + 1, B1{}, std::false_type{});
  }
  if (kt + 2 == n_tiles) {
    tile(kt, B0{}, std::false_type{});
    tile(kt + 1, B1{}, std::true_type{});
  } else {
    tile(kt, B0{}, std::true_type{});
  }

  // ---- epilogue: O^T[d][q] / l  ->  out[q][h*64 + d], transposed through LDS so rows leave as 128 B (one block at a time) ----
  char* ob = smem + wave * (32 * 144);  // [32 q][64 d] bf16, row stride 144 B (128 + 16 pad); wave-private
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    const float l_tot = xor32_reduce(l_run[w], OpSum{});
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        uint2 pk;
        pk.x = N16<T16>::pk(o[w][db][4 * rg + 0] * inv, o[w][db][4 * rg + 1] * inv);
        pk.y = N16<T16>::pk(o[w][db][4 * rg + 2] * inv, o[w][db][4 * rg + 3] * inv);
        const int dcol = 32 * db + 8 * rg + 4 * hf;
        *(uint2*)(ob + r * 144 + dcol * 2) = pk;
      }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = it * 64 + lane, row = id >> 3, c = id & 7;
      if (q0 + 32 * w + row < Tn) {
        uint4 v = *(const uint4*)(ob + row * 144 + c * 16);
        *(uint4*)(out + ((int64_t)b * Tn + q0 + 32 * w + row) * d + h * 64 + c * 8) = v;
      }
    }
    if (w + 1 < QW) {   // the staging rows are rewritten by the next block: its reads above must have landed first
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

thread_local int g_flash_qw = 2;   // option flash_qw: query blocks of 32 per wave in the encoder's flash attention (1 = the round-5 form)
template <typename T16>
void launch_enc_attn_flash_bf16(const T16* qkv, T16* out, int B, int T_, int H, hipStream_t s) {
  if (g_flash_qw >= 2) {
    dim3 grid((T_ + 255) / 256, H, B);
    hipLaunchKernelGGL((enc_attn_flash_kernel<T16, false, 2>), grid, dim3(256), 32768, s, (const bf16_t*)qkv, (bf16_t*)out, T_, H,
                       (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0);
    return;
  }
  dim3 grid((T_ + 127) / 128, H, B);
  hipLaunchKernelGGL((enc_attn_flash_kernel<T16, false, 1>), grid, dim3(256), 32768, s, (const bf16_t*)qkv, (bf16_t*)out, T_, H,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0);
}
// q = [n_clips][n_q][d] (the n_q rows of a clip are consecutive), K / V = cross-KV cache of those clips, out like q
template <typename T16>
void launch_cross_attn_flash_bf16(const T16* q, const T16* K, const T16* V, T16* out, int n_clips, int n_q, int H, int Tk, hipStream_t s) {
  dim3 grid((n_q + 127) / 128, H, n_clips);   // prompt positions: at most 447 rows per clip - the 128-query form fills more workgroups
  hipLaunchKernelGGL((enc_attn_flash_kernel<T16, true, 1>), grid, dim3(256), 32768, s, (const bf16_t*)q, (bf16_t*)out, n_q, H,
                     (const bf16_t*)K, (const bf16_t*)V, Tk);
}
template void launch_enc_attn_flash_bf16<bf16_t>(const bf16_t*, bf16_t*, int, int, int, hipStream_t);
template void launch_enc_attn_flash_bf16<f16_t>(const f16_t*, f16_t*, int, int, int, hipStream_t);
template void launch_cross_attn_flash_bf16<bf16_t>(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, hipStream_t);
template void launch_cross_attn_flash_bf16<f16_t>(const f16_t*, const f16_t*, const f16_t*, f16_t*, int, int, int, int, hipStream_t);
