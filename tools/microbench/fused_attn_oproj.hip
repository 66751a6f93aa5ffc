// EXPERIMENT, not part of libttasr (measured slower than the shipped plan: profiles/r4_attn_projection_fusion.txt).
// Decode-chain fusions (round 4): attention kernels that also apply the OUTPUT PROJECTION of their
// head, so the separate out-proj launch disappears from the decoder layer.  Included by layer_bench4.hip / fused_check.hip.
//
//   self_attn_oproj<T, G>   one workgroup = G rows x one head, ONE WAVE per (row, head): append k, v to the paged cache, attend
//                           over pos + 1 keys, then multiply the G x 64 head outputs by the head's 64 columns of Wo on the MFMA
//                           pipe and store the G x d partial result into f32 slab `h` ([H][rows][d]).  The LayerNorm that
//                           follows sums x + bias + slab[0] + ... + slab[H - 1] in slab order - no float atomics, bit-reproducible,
//                           exactly the protocol of the K-split GEMMs (kernels_skinny.hip), with K slices = heads.
//
// Why G rows per workgroup and not one: the head's slice of Wo is 64 x d x 2 B = 164 KB at large-v3; with one workgroup per
// (row, head) every one of the 32 rows would pull it through L2 again (105 MB per layer against the 3.3 MB the matrix has).
// G = 4: 26 MB, G = 8: 13 MB.  Fragment-packed weights (kernels_skinny.hip layout): for n-block nb (32 outputs) and k-step ks
// (16 inputs) the 1 KiB at ((nb * K/16 + ks) * 64 + lane) * 16 B is lane `lane`'s operand W[nb*32 + (lane & 31)][ks*16 + 8*(lane>>5) ..+8];
// head h owns k-steps 4h .. 4h + 3.  Used as the B operand (k x n) with the head outputs as the A operand (rows = batch rows):
// D[m = row][n], D column = lane & 31 = n, so each accumulator register is 32 consecutive outputs of one row - 128-byte stores.
#include "../../taiwan_tongues_asr_ce_amd/csrc/common.hpp"

namespace {
using fu32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int FPAGE = 16;   // tokens per KV page (kernels_attn.hip PAGE)

// value = round_T(bias + slab[0] + ... + slab[n-1]) for one 16-byte chunk of 8 stored values (n <= 4), all loads first
template <typename T>
__device__ __forceinline__ void fused_row_slabs(const SlabIn& si, int64_t off, int col, float (&v)[8]) {
  float4 t[4][2], bs[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) bs[c] = *(const float4*)(si.bias + col + 4 * c);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float* p = si.slab + (int64_t)min(s, si.n - 1) * si.stride + off;
#pragma unroll
    for (int c = 0; c < 2; ++c) t[s][c] = *(const float4*)(p + 4 * c);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float4 a = bs[c];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s == 0 || s < si.n) { a.x += t[s][c].x; a.y += t[s][c].y; a.z += t[s][c].z; a.w += t[s][c].w; }
    v[4 * c] = to_f<T>(from_f<T>(a.x)); v[4 * c + 1] = to_f<T>(from_f<T>(a.y));
    v[4 * c + 2] = to_f<T>(from_f<T>(a.z)); v[4 * c + 3] = to_f<T>(from_f<T>(a.w));
  }
}
template <typename T> __device__ __forceinline__ fu32x4 pack8(const float (&v)[8]) {
  fu32x4 o;
  o.x = N16<T>::pk(v[0], v[1]); o.y = N16<T>::pk(v[2], v[3]); o.z = N16<T>::pk(v[4], v[5]); o.w = N16<T>::pk(v[6], v[7]);
  return o;
}
}  // namespace

// NB10 = n-blocks per wave (d / 32 / G rounded up, <= 10 at d = 1280, G = 4)
// OPROJ = false (round-4 side experiment): only the one-wave-per-(row, head) attention - no workgroup barrier at all - writing the
// head outputs as T rows to att_out like self_attn_decode_kernel; the out-projection stays the K-split GEMM launch.
template <typename T, int G, bool IDENT, int NBW, bool OPROJ = true>
__global__ __launch_bounds__(G * 64) void self_attn_oproj_kernel(T* pool_, const int32_t* page_table_, int pages_per_seq_, int row0_,
                                                                 const int32_t* step_, int H_, int B_, SlabIn sq, const T* qkv_,
                                                                 const bf16_t* Wo_sh_, float* oslab_, int64_t oslab_stride_,
                                                                 T* att_out_ = nullptr) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int UNROLL = G <= 4 ? 8 : 4;   // 8 keys per wave-instruction x UNROLL cached keys per round trip (registers: 2 waves per SIMD at G = 8)
  __shared__ __attribute__((aligned(16))) uint16_t o16[G][64];
  T* pool = sgpr_pin_ptr(pool_); const int32_t* page_table = sgpr_pin_ptr(page_table_); const int32_t* step = sgpr_pin_ptr(step_);
  const T* qkv = sgpr_pin_ptr(qkv_); const bf16_t* Wo_sh = sgpr_pin_ptr(Wo_sh_); float* oslab = sgpr_pin_ptr(oslab_);
  const int pages_per_seq = sgpr_pin(pages_per_seq_), row0 = sgpr_pin(row0_), H = sgpr_pin(H_), B = sgpr_pin(B_);
  const int64_t oslab_stride = sgpr_pin(oslab_stride_);
  sq.slab = sgpr_pin_ptr(sq.slab); sq.bias = sgpr_pin_ptr(sq.bias); sq.n = sgpr_pin(sq.n); sq.stride = sgpr_pin(sq.stride);
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64, ks_per = d >> 4, n_blocks = d >> 5;
  const int b = min((int)blockIdx.y * G + wave, B - 1);   // rows past B repeat the last row; their stores are dropped
  const bool row_ok = (int)blockIdx.y * G + wave < B;
  const int sub = lane & 7, rin = lane >> 3;
  // ---- this wave's share of the head's Wo slice: n-blocks wave, wave + G, ...; 4 k-steps each.  Requested FIRST: nothing
  // they depend on is produced by the preceding launch
  fu32x4 wv[NBW][4];
  if constexpr (OPROJ) {
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int nb = min(wave + G * i, n_blocks - 1);
      const fu32x4* wp = (const fu32x4*)Wo_sh + ((int64_t)nb * ks_per + 4 * h) * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wv[i][kk] = wp[kk * 64];
    }
  }
  const int pos = *step;
  const int bg = b + row0;
  const int32_t* pt = page_table + bg * pages_per_seq;
  auto page_of = [&](int t) { return IDENT ? bg * pages_per_seq + t / FPAGE : pt[t / FPAGE]; };
  fu32x4 kraw[UNROLL], vraw[UNROLL];
  auto load_kv = [&](int it0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int tc = max(min((it0 + u) * 8 + rin, pos - 1), 0);
      const int64_t base = ((int64_t)page_of(tc) * 2 * H + h) * FPAGE + (tc % FPAGE);
      kraw[u] = *(const fu32x4*)(pool + base * 64 + sub * 8);
      vraw[u] = *(const fu32x4*)(pool + (base + (int64_t)H * FPAGE) * 64 + sub * 8);
    }
  };
  load_kv(0);
  // ---- q (every lane: its 16-byte chunk), k and v of this step (row slot 0 appends them and scores the new key)
  float q[8], kn[8], vn[8];
  {
    const int64_t off = (int64_t)b * 3 * d + h * 64 + sub * 8;
    const int col = h * 64 + sub * 8;
    if (sq.n > 0) {
      fused_row_slabs<T>(sq, off, col, q);
      fused_row_slabs<T>(sq, off + d, col + d, kn);
      fused_row_slabs<T>(sq, off + 2 * d, col + 2 * d, vn);
    } else {
      const fu32x4 a = *(const fu32x4*)(qkv + off), bb = *(const fu32x4*)(qkv + off + d), cc = *(const fu32x4*)(qkv + off + 2 * d);
      up8<T>(make_uint4(a.x, a.y, a.z, a.w), q); up8<T>(make_uint4(bb.x, bb.y, bb.z, bb.w), kn);
      up8<T>(make_uint4(cc.x, cc.y, cc.z, cc.w), vn);
    }
  }
  if (rin == 0 && row_ok) {
    const int page = page_of(pos);
    T* kdst = pool + ((((int64_t)page * 2 + 0) * H + h) * FPAGE + (pos % FPAGE)) * 64;
    T* vdst = pool + ((((int64_t)page * 2 + 1) * H + h) * FPAGE + (pos % FPAGE)) * 64;
    *(fu32x4*)(kdst + sub * 8) = pack8<T>(kn);   // kn / vn hold exactly representable values: the conversion is exact
    *(fu32x4*)(vdst + sub * 8) = pack8<T>(vn);
  }
  // ---- online softmax, one state per row slot
  float m_run = -1e30f, l_run = 0.f, acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int n_it = (pos + 7) >> 3;
  for (int it0 = 0; it0 < n_it; it0 += UNROLL) {
    if (it0 > 0) load_kv(it0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int t = (it0 + u) * 8 + rin;
      float kf[8], vf[8];
      up8<T>(make_uint4(kraw[u].x, kraw[u].y, kraw[u].z, kraw[u].w), kf);
      up8<T>(make_uint4(vraw[u].x, vraw[u].y, vraw[u].z, vraw[u].w), vf);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(q[j], kf[j], s);
      s = group_reduce<8>(s, OpSum{});
      if (t < pos) {
        const float mn = fmaxf(m_run, s);
        const float sc = __expf(m_run - mn), p = __expf(s - mn);
        l_run = l_run * sc + p;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(acc[j], sc, p * vf[j]);
        m_run = mn;
      }
    }
  }
  if (rin == 0) {   // the new token, from registers
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(q[j], kn[j], s);
    s = group_reduce<8>(s, OpSum{});
    const float mn = fmaxf(m_run, s);
    const float sc = __expf(m_run - mn), p = __expf(s - mn);
    l_run = l_run * sc + p;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(acc[j], sc, p * vn[j]);
    m_run = mn;
  }
  // merge the 8 row slots of the wave (lanes sub, sub + 8, ... hold the same columns)
  const float M = stride_reduce<8>(m_run, OpMax{});
  const float wgt = __expf(m_run - M);
  const float denom = stride_reduce<8>(l_run * wgt, OpSum{});
  const float inv = 1.f / denom;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = stride_reduce<8>(acc[j] * wgt, OpSum{}) * inv;
  if constexpr (!OPROJ) {
    T* att_out = sgpr_pin_ptr(att_out_);
    if (rin == 0 && row_ok) *(fu32x4*)(att_out + (int64_t)b * d + h * 64 + sub * 8) = pack8<T>(o);
    return;
  }
  if (rin == 0) *(fu32x4*)&o16[wave][sub * 8] = pack8<T>(o);
  __syncthreads();
  // ---- out-projection of this head: A = head outputs (m = row of the group), B = Wo fragments (k x n)
  s16x8 af[4];
  {
    const int m = min(lane & 31, G - 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) af[kk] = *(const s16x8*)&o16[m][kk * 16 + 8 * (lane >> 5)];
  }
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int nb = wave + G * i;
    f32x16 c;
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) c = N16<T>::mfma32(af[kk], *(s16x8*)&wv[i][kk], c);
    // D: column = lane & 31 = n, row m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): rows 0..3 are registers 0..3 of lanes 0..31,
    // rows 4..7 registers 0..3 of lanes 32..63 (G <= 8)
    if (nb < n_blocks) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = r + 4 * (lane >> 5);
        const int row = (int)blockIdx.y * G + m;
        if (m < G && row < B) oslab[(int64_t)h * oslab_stride + (int64_t)row * d + nb * 32 + (lane & 31)] = c[r];
      }
    }
  }
}

template <typename T>
bool launch_self_attn_oproj(T* kv_pool, const int32_t* page_table, int pages_per_seq, int64_t pool_layer_off, int identity_pages, int row0,
                            const int32_t* step, int B, int H, SlabIn sqkv, const T* qkv, const T* Wo_sh, float* oslab,
                            int64_t oslab_stride, int G, hipStream_t s) {
  if constexpr (sizeof(T) != 2) return false;
  else {
    const int n_blocks = H * 2;
    if (H * 64 > 1280 || (G != 4 && G != 8) || sqkv.n > 4) return false;
    T* pool = kv_pool + pool_layer_off;
#define TTASR_SAO(G_, IDENT_, NBW_)                                                                                                   \
  hipLaunchKernelGGL((self_attn_oproj_kernel<T, G_, IDENT_, NBW_>), dim3(H, (B + G_ - 1) / G_), dim3(G_ * 64), 0, s, pool, page_table, \
                     pages_per_seq, row0, step, H, B, sqkv, qkv, (const bf16_t*)Wo_sh, oslab, oslab_stride)
    if (G == 4) {
      if (n_blocks > 40) return false;
      if (identity_pages) TTASR_SAO(4, true, 10); else TTASR_SAO(4, false, 10);
    } else {
      if (n_blocks > 40) return false;
      if (identity_pages) TTASR_SAO(8, true, 5); else TTASR_SAO(8, false, 5);
    }
#undef TTASR_SAO
    return true;
  }
}
// attention only, one wave per (row, head), 4 rows per workgroup (no workgroup barrier); T rows out
template <typename T>
void launch_self_attn_wave(T* kv_pool, const int32_t* page_table, int pages_per_seq, int64_t pool_layer_off, int identity_pages, int row0,
                           const int32_t* step, int B, int H, SlabIn sqkv, const T* qkv, T* att, hipStream_t s) {
  T* pool = kv_pool + pool_layer_off;
  if (identity_pages)
    hipLaunchKernelGGL((self_attn_oproj_kernel<T, 4, true, 1, false>), dim3(H, (B + 3) / 4), dim3(256), 0, s, pool, page_table, pages_per_seq, row0,
                       step, H, B, sqkv, qkv, (const bf16_t*)nullptr, (float*)nullptr, (int64_t)0, att);
  else
    hipLaunchKernelGGL((self_attn_oproj_kernel<T, 4, false, 1, false>), dim3(H, (B + 3) / 4), dim3(256), 0, s, pool, page_table, pages_per_seq, row0,
                       step, H, B, sqkv, qkv, (const bf16_t*)nullptr, (float*)nullptr, (int64_t)0, att);
}
template void launch_self_attn_wave<bf16_t>(bf16_t*, const int32_t*, int, int64_t, int, int, const int32_t*, int, int, SlabIn, const bf16_t*, bf16_t*, hipStream_t);

template bool launch_self_attn_oproj<bf16_t>(bf16_t*, const int32_t*, int, int64_t, int, int, const int32_t*, int, int, SlabIn, const bf16_t*,
                                             const bf16_t*, float*, int64_t, int, hipStream_t);
template bool launch_self_attn_oproj<f16_t>(f16_t*, const int32_t*, int, int64_t, int, int, const int32_t*, int, int, SlabIn, const f16_t*,
                                            const f16_t*, float*, int64_t, int, hipStream_t);
template bool launch_self_attn_oproj<float>(float*, const int32_t*, int, int64_t, int, int, const int32_t*, int, int, SlabIn, const float*,
                                            const float*, float*, int64_t, int, hipStream_t);
