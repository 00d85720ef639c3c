// Flash-attention backward on MFMA for gfx950 (D = 64 heads of the SD-Turbo UNet / ControlNet; SURVEY.md section 8 row a12,
// xformers' memory-efficient attention backward in the reference's step, diffusion/train_controlnet_genima.py:1125-1126, :1402).
//
// P is never materialised in HBM: it is recomputed per tile from q, k and the forward's log-sum-exp (gn_attn_desc.lse),
//     P = exp2(s*c - lse2),   dS = scale * P * (dP - delta),   delta[q] = sum_d dO[q, d] * O[q, d],
// in two deterministic kernels (no atomics):
//   * attn_bwd_dq_kernel   -- a block owns 128 queries (registers) and streams 64-key tiles:  dQ^T += K^T . dS^T
//   * attn_bwd_dkv_kernel  -- a block owns 128 keys (registers) and streams 64-query tiles:   dV^T += dO^T . P,  dK^T += Q^T . dS
// Both follow the forward kernel's transposed formulation (attention.hip): the owner side sits in the MFMA B operand / the lane
// (= accumulator column), the streamed side is the A operand read from XOR-swizzled LDS tiles whose rows are stored with index bits
// 2 and 3 swapped, so the 8 accumulators a lane holds per 16-row step are 8 consecutive streamed rows and convert straight into the
// next MFMA's B operand.  The second product of each kernel needs the streamed operand transposed ([d][row]): it is read out of
// the SAME row-major LDS tile with ds_read_b64_tr_b16 (tr_frag below) -- round 1 streamed separate Q^T / K^T / dO^T copies made by
// gn_transpose2d (96 transposes per step, twice the streamed bytes and LDS).
#include <type_traits>

#include "common.h"

// waves per SIMD passed to __launch_bounds__: with the hint the dK / dV kernel comes out at 166 VGPRs (three waves per SIMD) instead of 206
// (two): 818 -> 799 us at 8 x 5 x 4096^2, 132 -> 124 us at 8 x 10 x 1024^2; three for the dQ kernel spills
#ifndef GN_ATTNB_DQ_WAVES
#define GN_ATTNB_DQ_WAVES 2
#endif
#ifndef GN_ATTNB_DKV_WAVES
#define GN_ATTNB_DKV_WAVES 2
#endif

namespace {

constexpr int D = 64;          // head dim
constexpr int TS = 64;         // streamed rows per tile
constexpr int TILE = 64 * 128; // bytes of one [64][64] f16 tile

struct AttnBwdParams {
  const f16 *q, *k, *v, *o, *d_o, *qt, *kt, *dot;
  const float* lse;
  float* delta;
  f16 *dq, *dk, *dv;
  long q_bs, k_bs, v_bs, o_bs, do_bs, qt_bs, kt_bs, dot_bs, dq_bs, dk_bs, dv_bs;
  int q_rs, k_rs, v_rs, o_rs, do_rs, qt_rs, kt_rs, dot_rs, dq_rs, dk_rs, dv_rs;
  int heads, Nq, Nk, Nk_rows;
  float scale, scale_log2;
};

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ int perm_row(int r) { return (r & 32) | swap23(r & 31); }

// Chunk swizzle of the streamed [rows][128 B] tiles: the three bits of (row >> 1) & 7 as the forward kernels' lds_swz<128> uses them, ROTATED so
// that row bit 1 lands on chunk bit 2.  The tiles are read two ways: ds_read_b128 MFMA fragments (rows = lane & 31, one chunk column: any
// bijection of the three bits is conflict-free) and ds_read_b64_tr_b16 transposed fragments, whose half-wave is FOUR consecutive rows x 64
// contiguous bytes -- rows r and r + 2 sit on the same half of the banks and must differ in the 64-byte quarter (chunk bit 2).  With the
// forward swizzle they differed in chunk bit 0 only: SQ_LDS_BANK_CONFLICT was 25 % of both kernels' LDS cycles (round 5,
// tools/probes/lds_conflict_pmc.sh), with the LDS array as busy as the matrix pipe in the dK / dV kernel.
__device__ __forceinline__ int bwd_swz(int row) {
  const int s = (row >> 1) & 7;
  return ((s & 1) << 2) | (s >> 1);
}
__device__ __forceinline__ int bwd_lds(int row, int chunk) { return row * 128 + ((chunk ^ bwd_swz(row)) << 4); }

// ---- LDS-DMA staging of the streamed [64 rows][128 B] tiles (same scheme as attention.hip / gemm_dma_kernel): a DMA instruction fills
// 8 consecutive LDS rows lane-linearly (16 B per lane), so the lane that owns LDS (row, physical chunk) fetches source row
// perm(row) (row-major operands) or row (transposed operands), logical chunk = chunk ^ bwd_swz(row).  Anything past a
// descriptor's extent reads as zero (rows >= the valid count of Q / dO / K / V).  4 waves: wave w issues groups w and w + 4.
typedef __attribute__((address_space(3))) void* bwd_lds_ptr_t;
struct TileStream {
  unsigned off[2];  // this lane's byte offsets for its two row groups; advance by `step` bytes per tile
  unsigned step;
};
__device__ __forceinline__ TileStream make_stream(int wave, int lane, long row_stride_elems, bool permute, long tile_step_bytes) {
  TileStream s;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + 4 * i) + (lane >> 3);
    const int chunk = (lane & 7) ^ bwd_swz(row);
    const int src = permute ? perm_row(row) : row;
    s.off[i] = (unsigned)(((long)src * row_stride_elems + chunk * 8) * 2);
  }
  s.step = (unsigned)tile_step_bytes;
  return s;
}
__device__ __forceinline__ void dma_stream(TileStream& s, const __amdgpu_buffer_rsrc_t rs, unsigned char* tile, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bwd_lds_ptr_t)(tile + (wave + 4 * i) * 1024), 16, s.off[i], 0, 0, 0);
    s.off[i] += s.step;
  }
}

// delta[b][h][q] = sum_d dO * O
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnBwdParams p, int B) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * p.Nq * p.heads) return;
  const int h = (int)(idx % p.heads);
  const long bq = idx / p.heads;
  const int q = (int)(bq % p.Nq), b = (int)(bq / p.Nq);
  const f16* o = p.o + (long)b * p.o_bs + (long)q * p.o_rs + h * D;
  const f16* g = p.d_o + (long)b * p.do_bs + (long)q * p.do_rs + h * D;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < D / 8; ++c) {
    const uint4 a = *reinterpret_cast<const uint4*>(o + c * 8), d = *reinterpret_cast<const uint4*>(g + c * 8);
    const f16x8 ah = *reinterpret_cast<const f16x8*>(&a), dh = *reinterpret_cast<const f16x8*>(&d);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)ah[e] * (float)dh[e];
  }
  p.delta[((long)b * p.heads + h) * p.Nq + q] = s;
}

// ---- dQ --------------------------------------------------------------------------------------------------------------------
// XCD-aware block order (as in the forward kernel): consecutive block ids go round-robin over the 8 XCDs, so the blocks that stream
// the same (batch, head)'s tiles get ids that land on ONE XCD's L2.  Returns (row block, head, batch) of this workgroup; the
// divisions run in the VALU, so the results are pinned back to SGPRs (descriptors built from them must stay uniform).
__device__ __forceinline__ void attn_bwd_block(int nblk, int heads, int& blk, int& h, int& b) {
  const int total = gridDim.x;
  const int slot = (total % 8 == 0) ? (int)(blockIdx.x % 8) * (total / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int bh = __builtin_amdgcn_readfirstlane(slot / nblk);
  b = __builtin_amdgcn_readfirstlane(bh / heads);
  h = bh - b * heads;
  blk = slot - bh * nblk;
}

// ---- the streamed operand TRANSPOSED, straight out of its row-major LDS tile (no Q^T / K^T / dO^T copies in HBM, no second tile
// per operand in LDS): gfx950's ds_read_b64_tr_b16 hands lane c of a 16-lane group column c of a 4-row x 16-column block whose rows
// the group's lanes address themselves.  Wanted: the MFMA A fragment of X^T -- d = 32 dt + (lane & 31), streamed rows
// 32 u + 16 g + 8 hi + (0..7).  Logical row 16 g + 8 hi + 4 half + j sits at physical row 16 g + 8 half + 4 hi + j (rows are stored
// with index bits 2 and 3 swapped), and the tile's chunk swizzle bwd_swz(row) (row bits 1 .. 3) does not depend on u or g: one
// byte offset per (dt, half) and lane, immediates for u and g.
typedef __fp16 bwd_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) bwd_h4* bwd_lds_h4_ptr;
struct BwdH8 { bwd_h4 lo, hi; };
struct TrOffsets { int off[2][2]; };  // [dt][half]
__device__ __forceinline__ TrOffsets make_tr_offsets(int lane) {
  TrOffsets t;
  const int ti = lane & 15, g2 = (lane >> 4) & 1, hb = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int prow = 8 * half + 4 * hb + (ti >> 2);
      const int col = dt * 32 + 16 * g2 + 4 * (ti & 3);
      t.off[dt][half] = prow * 128 + ((((col >> 3) ^ bwd_swz(prow))) << 4) + ((col & 7) << 1);
    }
  return t;
}
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* tile, const TrOffsets& t, int dt, int u, int g) {
  const unsigned char* q = tile + (32 * u + 16 * g) * 128;
  const bwd_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((bwd_lds_h4_ptr)(q + t.off[dt][0]));
  const bwd_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((bwd_lds_h4_ptr)(q + t.off[dt][1]));
  return __builtin_bit_cast(f16x8, BwdH8{lo, hi});
}

__global__ __launch_bounds__(256, GN_ATTNB_DQ_WAVES) void attn_bwd_dq_kernel(const AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int blk, h, b;
  attn_bwd_block((p.Nq + 127) / 128, p.heads, blk, h, b);
  const int qrow = blk * 128 + wave * 32 + l31;
  const bool qlive = qrow < p.Nq;

  const f16* kp = p.k + (long)b * p.k_bs + h * D;
  const f16* vp = p.v + (long)b * p.v_bs + h * D;
  const TrOffsets tro = make_tr_offsets(lane);

  f16x8 qf[4], gf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 a = make_uint4(0, 0, 0, 0), g = make_uint4(0, 0, 0, 0);
    if (qlive) {
      a = *reinterpret_cast<const uint4*>(p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + h * D + ks * 16 + hi * 8);
      g = *reinterpret_cast<const uint4*>(p.d_o + (long)b * p.do_bs + (long)qrow * p.do_rs + h * D + ks * 16 + hi * 8);
    }
    qf[ks] = *reinterpret_cast<f16x8*>(&a);
    gf[ks] = *reinterpret_cast<f16x8*>(&g);
  }
  const long sidx = ((long)b * p.heads + h) * p.Nq + qrow;
  const float L = qlive ? p.lse[sidx] : 0.0f;
  const float dl = qlive ? p.delta[sidx] : 0.0f;
  const float c = p.scale_log2, sc = p.scale;

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 acc[2] = {zero16, zero16};

  const int ntiles = (p.Nk + TS - 1) / TS;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // K / V rows past Nk_rows lie beyond the descriptors and read as zeros
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)((((long)p.Nk_rows - 1) * p.k_rs + D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)((((long)p.Nk_rows - 1) * p.v_rs + D) * 2), 0x00020000);
  TileStream sk = make_stream(wv, lane, p.k_rs, true, (long)TS * p.k_rs * 2);
  TileStream sv = make_stream(wv, lane, p.v_rs, true, (long)TS * p.v_rs * 2);
  auto dma_tile = [&](int buf) {  // the next tile in sequence
    unsigned char* Ks = smem + buf * 2 * TILE;
    dma_stream(sk, rs_k, Ks, wv);
    dma_stream(sv, rs_v, Ks + TILE, wv);
  };
  if (ntiles > 0) dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  // One branch per key tile, the back edge (conditional branches inside the loop cost issue slots even when they fall through --
  // tools/probes/attn_phase_model.hip): the last tile, the only one that can be ragged, is peeled; the loop body prefetches unconditionally.
  auto tile = [&](auto last_c, int t) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    if constexpr (!LAST) dma_tile(cur ^ 1);  // buffer cur^1 was last read before the barrier that ended the previous iteration
    const unsigned char* Ks = smem + cur * 2 * TILE;
    const unsigned char* Vs = Ks + TILE;
    const int j0 = t * TS;

    f32x16 s[2], dp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + bwd_lds(u * 32 + l31, ks * 2 + hi));
        const f16x8 vf = *reinterpret_cast<const f16x8*>(Vs + bwd_lds(u * 32 + l31, ks * 2 + hi));
        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero16 : s[u], 0, 0, 0);
        dp[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, gf[ks], ks == 0 ? zero16 : dp[u], 0, 0, 0);
      }
    // accumulator r of sub-tile u holds key j0 + 32u + 16(r>>3) + 8hi + (r&7)
    f16x8 dsf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(fmaf(s[u][r], c, -L));
        if constexpr (LAST) pv = (j0 + 32 * u + 16 * (r >> 3) + 8 * hi + (r & 7) >= p.Nk) ? 0.0f : pv;  // keys past Nk (only the last tile has any)
        dsf[u][r >> 3][r & 7] = (f16)(pv * (dp[u][r] - dl) * sc);
      }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f16x8 tf = tr_frag(Ks, tro, dt, u, g);  // K^T[d][key] out of the row-major K tile
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tf, dsf[u][g], acc[dt], 0, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next tile have landed
    __syncthreads();
    cur ^= 1;
  };
  for (int t = 0; t + 1 < ntiles; ++t) tile(std::false_type{}, t);
  if (ntiles > 0) tile(std::true_type{}, ntiles - 1);

  if (qlive) {
    f16* op = p.dq + (long)b * p.dq_bs + (long)qrow * p.dq_rs + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (f16)acc[dt][4 * g + i];
        *reinterpret_cast<f16x4*>(op + dt * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

// ---- dK, dV ----------------------------------------------------------------------------------------------------------------
constexpr int KV_BUF = 2 * TILE + 2 * 64 * 4;  // Q, dO tiles + lse + delta of the query tile

__global__ __launch_bounds__(256, GN_ATTNB_DKV_WAVES) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KV_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int blk, h, b;
  attn_bwd_block((p.Nk + 127) / 128, p.heads, blk, h, b);
  const int key = blk * 128 + wave * 32 + l31;
  const bool klive = key < p.Nk;

  const f16* qp = p.q + (long)b * p.q_bs + h * D;
  const f16* gp = p.d_o + (long)b * p.do_bs + h * D;
  const TrOffsets tro = make_tr_offsets(lane);
  const float* lsep = p.lse + ((long)b * p.heads + h) * p.Nq;
  const float* delp = p.delta + ((long)b * p.heads + h) * p.Nq;

  f16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 a = make_uint4(0, 0, 0, 0), g = make_uint4(0, 0, 0, 0);
    if (klive) {
      a = *reinterpret_cast<const uint4*>(p.k + (long)b * p.k_bs + (long)key * p.k_rs + h * D + ks * 16 + hi * 8);
      g = *reinterpret_cast<const uint4*>(p.v + (long)b * p.v_bs + (long)key * p.v_rs + h * D + ks * 16 + hi * 8);
    }
    kf[ks] = *reinterpret_cast<f16x8*>(&a);
    vf[ks] = *reinterpret_cast<f16x8*>(&g);
  }
  const float c = p.scale_log2, sc = p.scale;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 accK[2] = {zero16, zero16}, accV[2] = {zero16, zero16};

  const int ntiles = (p.Nq + TS - 1) / TS;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // Q / dO rows past Nq read as zeros (and their lse is +inf: P = dS = 0 there)
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)qp, 0, (int)((((long)p.Nq - 1) * p.q_rs + D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, (int)((((long)p.Nq - 1) * p.do_rs + D) * 2), 0x00020000);
  TileStream sq = make_stream(wv, lane, p.q_rs, true, (long)TS * p.q_rs * 2);
  TileStream sg = make_stream(wv, lane, p.do_rs, true, (long)TS * p.do_rs * 2);
  int jn = 0;  // first query row of the next tile to stage
  float rl = 0.f, rd = 0.f;
  auto dma_tile = [&](int buf) {  // the next tile in sequence; the 2 x 64 row statistics ride along through one register each
    unsigned char* Qs = smem + buf * KV_BUF;
    dma_stream(sq, rs_q, Qs, wv);
    dma_stream(sg, rs_g, Qs + TILE, wv);
    {  // every wave fetches the same 64 values (one wave would do, but "if (tid < 64)" is a branch per tile): clamped index, then a select
      const int row = jn + lane, rc = min(row, p.Nq - 1);
      const float l = lsep[rc], dd = delp[rc];
      rl = row < p.Nq ? l : INFINITY;  // dead query rows: P = exp2(.. - inf) = 0
      rd = row < p.Nq ? dd : 0.0f;
    }
    jn += TS;
  };
  auto store_stats = [&](int buf) {  // (all four waves write the same values)
    unsigned char* Qs = smem + buf * KV_BUF;
    reinterpret_cast<float*>(Qs + 2 * TILE)[lane] = rl;
    reinterpret_cast<float*>(Qs + 2 * TILE + 256)[lane] = rd;
  };
  if (ntiles > 0) {
    dma_tile(0);
    store_stats(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  auto tile = [&](auto last_c) __attribute__((always_inline)) {  // (one branch per query tile: see the dQ kernel)
    constexpr bool LAST = decltype(last_c)::value;
    if constexpr (!LAST) dma_tile(cur ^ 1);  // buffer cur^1 was last read before the barrier that ended the previous iteration
    const unsigned char* Qs = smem + cur * KV_BUF;
    const unsigned char* Gs = Qs + TILE;
    const float* Ls = reinterpret_cast<const float*>(Qs + 2 * TILE);
    const float* Ds = Ls + 64;

    f32x16 s[2], dp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f16x8 qfr = *reinterpret_cast<const f16x8*>(Qs + bwd_lds(u * 32 + l31, ks * 2 + hi));
        const f16x8 gfr = *reinterpret_cast<const f16x8*>(Gs + bwd_lds(u * 32 + l31, ks * 2 + hi));
        s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr, kf[ks], ks == 0 ? zero16 : s[u], 0, 0, 0);
        dp[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfr, vf[ks], ks == 0 ? zero16 : dp[u], 0, 0, 0);
      }
    // accumulator r of sub-tile u holds query j0 + 32u + 16(r>>3) + 8hi + (r&7)  (row statistics come from LDS)
    f16x8 pf[2][2], dsf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int base = 32 * u + 16 * g + 8 * hi;
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(Ls + base), l1 = *reinterpret_cast<const f32x4*>(Ls + base + 4);
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(Ds + base), d1 = *reinterpret_cast<const f32x4*>(Ds + base + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float Lq = j < 4 ? l0[j & 3] : l1[j & 3];
          const float dq_ = j < 4 ? d0[j & 3] : d1[j & 3];
          const int r = 8 * g + j;
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[u][r], c, -Lq));
          pf[u][g][j] = (f16)pv;
          dsf[u][g][j] = (f16)(pv * (dp[u][r] - dq_) * sc);
        }
      }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const f16x8 gt = tr_frag(Gs, tro, dt, u, g);  // dO^T[d][query] and Q^T[d][query] out of the row-major tiles
          const f16x8 qt = tr_frag(Qs, tro, dt, u, g);
          accV[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gt, pf[u][g], accV[dt], 0, 0, 0);
          accK[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qt, dsf[u][g], accK[dt], 0, 0, 0);
        }
    if constexpr (!LAST) store_stats(cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next tile have landed
    __syncthreads();
    cur ^= 1;
  };
  for (int t = 0; t + 1 < ntiles; ++t) tile(std::false_type{});
  if (ntiles > 0) tile(std::true_type{});

  if (key < p.Nk_rows) {  // the padding rows [Nk, Nk_rows) of this 128-key block come out as zeros (no fill launch in front of the kernel)
    f16* okp = p.dk + (long)b * p.dk_bs + (long)key * p.dk_rs + h * D;
    f16* ovp = p.dv + (long)b * p.dv_bs + (long)key * p.dv_rs + h * D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 a, v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = klive ? (f16)accK[dt][4 * g + i] : (f16)0.0f;
          v[i] = klive ? (f16)accV[dt][4 * g + i] : (f16)0.0f;
        }
        *reinterpret_cast<f16x4*>(okp + dt * 32 + 8 * g + 4 * hi) = a;
        *reinterpret_cast<f16x4*>(ovp + dt * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

}  // namespace

extern "C" int32_t gn_attention_bwd(gn_ctx* ctx, const gn_attn_bwd_desc* d) {
  GN_REQUIRE(ctx && d, "gn_attention_bwd: null ctx/desc");
  GN_REQUIRE(d->q && d->k && d->v && d->o && d->d_o && d->lse && d->delta && d->dq && d->dk && d->dv, "gn_attention_bwd: null pointer");
  GN_REQUIRE(d->D == 64, "gn_attention_bwd: head dim %d unsupported (64)", d->D);
  GN_REQUIRE(d->B > 0 && d->heads > 0 && d->Nq > 0 && d->Nk > 0 && d->Nq % 8 == 0 && d->Nk_rows % 8 == 0 && d->Nk_rows >= d->Nk,
             "gn_attention_bwd: Nq (%d) and Nk_rows (%d) must be multiples of 8, Nk_rows >= Nk (%d)", d->Nq, d->Nk_rows, d->Nk);
  const int64_t bs[] = {d->q_bs, d->k_bs, d->v_bs, d->o_bs, d->do_bs};
  const int32_t rs[] = {d->q_rs, d->k_rs, d->v_rs, d->o_rs, d->do_rs};
  for (int i = 0; i < 5; ++i) GN_REQUIRE(bs[i] % 8 == 0 && rs[i] % 8 == 0, "gn_attention_bwd: input strides must be multiples of 8");
  GN_REQUIRE(d->dq_rs % 4 == 0 && d->dk_rs % 4 == 0 && d->dv_rs % 4 == 0 && d->dq_bs % 4 == 0 && d->dk_bs % 4 == 0 && d->dv_bs % 4 == 0,
             "gn_attention_bwd: output strides must be multiples of 4");
  const void* in[] = {d->q, d->k, d->v, d->o, d->d_o};
  for (int i = 0; i < 5; ++i) GN_REQUIRE(((uintptr_t)in[i] & 15) == 0, "gn_attention_bwd: inputs must be 16-byte aligned");
  GN_REQUIRE(((uintptr_t)d->dq & 7) == 0 && ((uintptr_t)d->dk & 7) == 0 && ((uintptr_t)d->dv & 7) == 0, "gn_attention_bwd: outputs must be 8-byte aligned");
  GN_REQUIRE(d->scale > 0.0f, "gn_attention_bwd: scale must be positive");
  AttnBwdParams p;
  p.q = (const f16*)d->q; p.k = (const f16*)d->k; p.v = (const f16*)d->v; p.o = (const f16*)d->o; p.d_o = (const f16*)d->d_o;
  p.qt = (const f16*)d->qt; p.kt = (const f16*)d->kt; p.dot = (const f16*)d->dot;
  p.lse = d->lse; p.delta = d->delta;
  p.dq = (f16*)d->dq; p.dk = (f16*)d->dk; p.dv = (f16*)d->dv;
  p.q_bs = d->q_bs; p.k_bs = d->k_bs; p.v_bs = d->v_bs; p.o_bs = d->o_bs; p.do_bs = d->do_bs; p.qt_bs = d->qt_bs; p.kt_bs = d->kt_bs;
  p.dot_bs = d->dot_bs; p.dq_bs = d->dq_bs; p.dk_bs = d->dk_bs; p.dv_bs = d->dv_bs;
  p.q_rs = d->q_rs; p.k_rs = d->k_rs; p.v_rs = d->v_rs; p.o_rs = d->o_rs; p.do_rs = d->do_rs; p.qt_rs = d->qt_rs; p.kt_rs = d->kt_rs;
  p.dot_rs = d->dot_rs; p.dq_rs = d->dq_rs; p.dk_rs = d->dk_rs; p.dv_rs = d->dv_rs;
  p.heads = d->heads; p.Nq = d->Nq; p.Nk = d->Nk; p.Nk_rows = d->Nk_rows;
  p.scale = d->scale; p.scale_log2 = d->scale * 1.4426950408889634f;
  const long nd = (long)d->B * d->Nq * d->heads;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, ctx->stream, p, d->B);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(((d->Nq + 127) / 128) * d->heads * d->B), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(((d->Nk + 127) / 128) * d->heads * d->B), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
