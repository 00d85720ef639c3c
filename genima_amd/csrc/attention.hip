// Flash-style attention forward on MFMA for gfx950 (self-, cross- and causal CLIP attention).
//
// Everything is computed transposed so the softmax row statistics are lane-local:
//   S^T[key, q] = K_tile . Q^T        (A operand = K rows from LDS, B operand = Q held in registers)
//   O^T[d,  q]  = V^T_tile . P^T      (A operand = V^T rows from LDS, B operand = P^T straight from the S^T accumulators)
// With v_mfma_f32_32x32x16_f16 the D layout puts column (= query) lane&31 in each lane, so the running max / sum of a query
// row live in the two lanes {l, l+32}; the only cross-lane traffic per key tile is one exchange with lane^32.
// The K tile is written to LDS with key bits 2 and 3 swapped so that the 8 S^T accumulators a lane holds per 16-key step are
// 8 *consecutive* keys: converting them to f16 gives the PV B-operand directly and the V^T fragment is one ds_read_b128.
// V arrives already transposed ([b][h*D+d][key]) from the V projection's epilogue, so both tiles use the same 16-byte-chunk
// XOR-swizzled LDS image as the GEMM kernel (conflict-free ds_read_b128).
//
// At D = 64 the kernel is VALU-issue-bound, not MFMA-bound (PMC, MI355X: VALU issue active 59 % of the SIMD cycles with the
// matrix pipe busy 40 % when the softmax costs ~4.5 VALU per score), so the common tile runs an "optimistic" softmax of
// 2 VALU per score -- v_exp_f32, half a v_cvt_pk_f16_f32, half a v_dot2c_f32_f16 -- and nothing else:
//   * the scale c = scale*log2(e) is multiplied into the Q fragments once, and the softmax reference m enters as the accumulator
//     init of the first QK^T MFMA (16 registers holding -m), so the MFMA accumulators ARE the exponents s*c - m;
//   * no row max is taken: m is whatever the last careful tile left.  P = exp2(s*c - m) may exceed 1; f16's relative precision
//     is scale-free and O and l carry the same factor, so only range matters.  A lane's 32-key sum of P above 2^13 (or inf / NaN)
//     flags the tile (wave-uniform) before anything is accumulated: the scores are recomputed and the tile takes the careful path;
//   * the careful path (first tile, masked tiles, flagged tiles) masks, takes the row max with v_max3_f32 chains, moves the
//     reference (rescaling O^T and l) and re-bases the exponents -- on typical data once or twice per row;
//   * P is packed with v_cvt_pk_f16_f32 and its row sum taken with v_dot2c_f32_f16 on the packed pairs (the sum then matches
//     the f16 P that enters the PV MFMA exactly).
// K/V tiles are double-buffered in LDS (one barrier per 64-key tile); at D = 64 tile t+1 arrives by LDS-DMA under the MFMAs of tile
// t, at D = 32 through registers.  Block = NW waves x TQ x 32 query rows (template); 4 waves x 32 rows is the default, the other
// shapes stay selectable (GN_ATTN_VARIANT) for measurements.  The non-causal D = 64 problems with a key count that is a multiple of 64 and V^T
// given -- the self-attention of the U-Net / ControlNet -- go to the software-pipelined, branch-free kernel of attention_stream.hip.
// Blocks are ordered XCD-aware: the query blocks of one (batch, head) share an L2.
#include <stdlib.h>

#include <type_traits>

#include "attention_common.h"

// waves per SIMD the 4-wave x 32-row D = 64 kernels are held to (register budget 512 / 3 = 170): the V^T kernel fits anyway (167 with
// -amdgpu-mfma-vgpr-form), the row-major-V one needs the hint (184 -> 168: 222 -> 206-216 us at 8 x 5 x 4096^2)
#ifndef GN_ATTN_WAVES
#define GN_ATTN_WAVES 3
#endif

namespace {

// VROW (D = 64 only): `vt` points at V in row-major form [B][Nk][vt_rs] (head h at column h * D) -- the layout a plain q | k | v
// projection leaves it in.  Its LDS tile is then filled like the K tile and the P.V A-operand (8 consecutive keys of one d) comes out
// through ds_read_b64_tr_b16, as in attention_bwd.hip.
typedef __fp16 attn_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) attn_h4* attn_lds_h4_ptr;
struct AttnH8 { attn_h4 lo, hi; };

template <int D, int NW, int TQ, bool VROW = false>
__global__ __launch_bounds__(NW * 64, (D == 64 && NW == 4 && TQ == 1 && GN_ATTN_WAVES > 0) ? GN_ATTN_WAVES : 1) void attn_fwd_kernel(const AttnParams p) {
  static_assert(!VROW || D == 64, "row-major V: the LDS-DMA (D = 64) path only");
  constexpr int NT = NW * 64;
  constexpr int QB = NW * TQ * 32;   // query rows per block
  constexpr int ROWB = D * 2;        // K tile row bytes
  constexpr int KCH = D / 8;         // 16-byte chunks per K row
  constexpr int KS = D / 16;         // k16 steps of QK^T
  constexpr int DT = D / 32;         // 32-row d tiles of O^T
  constexpr int K_BYTES = KT * ROWB; // K tile: [64 keys][D]
  constexpr int V_BYTES = D * 128;   // V^T tile: [D rows][64 keys]
  constexpr int KLD = (KT * KCH) / NT;  // chunks per thread (K)
  constexpr int VLD = (D * 8) / NT;     // chunks per thread (V^T)
  static_assert(KLD >= 1 && VLD >= 1, "tile too small for the block");

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (K_BYTES + V_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // XCD-aware block order: consecutive block ids go round-robin over the 8 XCDs (each with its own L2), so the query blocks that
  // share one (batch, head)'s K / V^T get ids that land on ONE XCD, next to each other in dispatch order (+9 % at 8 x 5 x 4096^2).
  // (the integer divisions run in the VALU: pin the results back to SGPRs, or the buffer descriptors below turn "divergent")
  const int nqb = (p.Nq + QB - 1) / QB, total = gridDim.x;
  const int slot = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;
  const int bh = __builtin_amdgcn_readfirstlane(slot / nqb);
  const int b = __builtin_amdgcn_readfirstlane(bh / p.heads), h = bh - b * p.heads;
  const int q0 = (slot - bh * nqb) * QB;
  const int qw = q0 + wave * (TQ * 32) + l31;  // this lane's query row in q-tile 0 (+32 per further tile)

  const f16* qp = p.q + (long)b * p.q_bs + (long)h * D;
  const f16* kp = p.k + (long)b * p.k_bs + (long)h * D;
  const f16* vp = VROW ? p.vt + (long)b * p.vt_bs + (long)h * D : p.vt + (long)b * p.vt_bs + (long)h * D * p.vt_rs;

  // Q fragments (B operand): lane holds Q[qrow][16*ks + 8*hi .. +8]
  f16x8 qf[TQ][KS];
#pragma unroll
  for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      const int qrow = qw + 32 * tq;
      if (qrow < p.Nq) v = *reinterpret_cast<const uint4*>(qp + (long)qrow * p.q_rs + ks * 16 + hi * 8);
      f16x8 q8 = *reinterpret_cast<f16x8*>(&v);
#pragma unroll
      for (int x = 0; x < 8; ++x) q8[x] = (f16)((float)q8[x] * p.scale_log2);  // exponent units straight out of the MFMA
      qf[tq][ks] = q8;
    }

  f32x16 oacc[TQ][DT], negm[TQ];
  float m_run[TQ], l_run[TQ];
#pragma unroll
  for (int tq = 0; tq < TQ; ++tq) {
    m_run[tq] = 0.0f;  // softmax reference in exponent units (score * scale * log2 e); the first key tile sets it
    l_run[tq] = 0.0f;  // this lane's half of the row sum
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[tq][r] = 0.0f;  // -m_run as an MFMA accumulator init (all 16 entries equal)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[tq][t][r] = 0.0f;
  }

  // key range: causal rows never look past their own index
  int nk_eff = p.Nk;
  if (p.causal) nk_eff = min(p.Nk, q0 + QB);
  const int ntiles = (nk_eff + KT - 1) / KT;

  // ---- D = 64: K / V^T tiles go global -> LDS directly (buffer_load ... lds, 16 B per lane, as in gemm_dma_kernel): no staging
  // registers, no ds_write, one integer add per tile and DMA instruction.  A DMA instruction fills 8 consecutive 128-byte LDS rows
  // lane-linearly, so the row permutation of K (key bits 2 <-> 3) and the XOR chunk swizzle are applied on the SOURCE side: the
  // lane that owns LDS (row, physical chunk) fetches key perm(row), logical chunk = chunk ^ ((row >> 1) & 7).  Keys >= Nk lie past
  // the K descriptor's extent and read as zeros; V^T's padded key columns are finite by contract (P is exactly 0 there).
  constexpr bool DMA = (D == 64);
  constexpr int NG = 8 / NW > 0 ? 8 / NW : 1;  // 8-row groups per wave and tile (K and V^T each have 8)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  static_assert(NG <= 2, "blocks have at least 4 waves");
  unsigned koff[2], voff[2];  // (fixed extent: a template-dependent extent here makes hipcc's host pass drop the kernel stub)
  if constexpr (DMA) {
    static_assert(!DMA || 8 % NW == 0, "8 row groups must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int row = 8 * (wv + NW * i) + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const int key = (row & 32) | swap23(row & 31);
      koff[i] = (unsigned)(((long)key * p.k_rs + chunk * 8) * 2);
      voff[i] = VROW ? (unsigned)(((long)key * p.vt_rs + chunk * 8) * 2) : (unsigned)(((long)row * p.vt_rs + chunk * 8) * 2);
    }
  }
  // transpose-read offsets of the row-major V tile: [dt][half] (attention_bwd.hip make_tr_offsets)
  int troff[2][2];
  if constexpr (VROW) {
    const int ti = lane & 15, g2 = (lane >> 4) & 1, hb = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int prow = 8 * half + 4 * hb + (ti >> 2);
        const int col = dt * 32 + 16 * g2 + 4 * (ti & 3);
        troff[dt][half] = prow * 128 + ((((col >> 3) ^ ((prow >> 1) & 7))) << 4) + ((col & 7) << 1);
      }
  }
  const long kbytes = ((long)(p.Nk - 1) * p.k_rs + D) * 2;
  const long vbytes = VROW ? ((long)(p.Nk - 1) * p.vt_rs + D) * 2 : ((long)(D - 1) * p.vt_rs + (long)((p.Nk + KT - 1) / KT) * KT) * 2;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)vbytes, 0x00020000);
  auto dma_tile = [&](int buf) {  // issues the NEXT tile in sequence (offsets advance by one tile per call)
    if constexpr (DMA) {
      unsigned char* Ks = smem + buf * (K_BYTES + V_BYTES);
      unsigned char* Vs = Ks + K_BYTES;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(Ks + (wv + NW * i) * 1024), 16, koff[i], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (attn_lds_ptr_t)(Vs + (wv + NW * i) * 1024), 16, voff[i], 0, 0, 0);
        koff[i] += (unsigned)(KT * p.k_rs * 2);
        voff[i] += VROW ? (unsigned)(KT * p.vt_rs * 2) : (unsigned)(KT * 2);
      }
    }
  };

  // ---- D = 32 (CLIP-B / ACT heads): register-staged tiles (64-byte K rows do not fit the 8-rows-per-DMA image)
  uint4 rk[KLD], rv[VLD];
  auto load_tile = [&](int t) {
    if constexpr (DMA) { (void)t; return; }
    const int j0 = t * KT;
#pragma unroll
    for (int i = 0; i < KLD; ++i) {
      const int id = tid + NT * i;
      const int key = id / KCH, ch = id % KCH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (j0 + key < p.Nk) v = *reinterpret_cast<const uint4*>(kp + (long)(j0 + key) * p.k_rs + ch * 8);
      rk[i] = v;
    }
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
      const int id = tid + NT * i;
      const int drow = id >> 3, ch = id & 7;
      uint4 v = *reinterpret_cast<const uint4*>(vp + (long)drow * p.vt_rs + j0 + ch * 8);
      const int kb = j0 + ch * 8;
      if (kb + 8 > p.Nk) {  // chunk straddles / lies beyond Nk: zero the dead keys (P is 0 there; 0 * garbage must stay 0)
        f16x8 e = *reinterpret_cast<f16x8*>(&v);
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (kb + x >= p.Nk) e[x] = (f16)0.0f;
        v = *reinterpret_cast<uint4*>(&e);
      }
      rv[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (DMA) { (void)buf; return; }
    unsigned char* Ks = smem + buf * (K_BYTES + V_BYTES);
    unsigned char* Vs = Ks + K_BYTES;
#pragma unroll
    for (int i = 0; i < KLD; ++i) {
      const int id = tid + NT * i;
      const int key = id / KCH, ch = id % KCH;
      const int row = (key & 32) | swap23(key & 31);
      *reinterpret_cast<uint4*>(Ks + lds_swz<ROWB>(row, ch)) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
      const int id = tid + NT * i;
      const int drow = id >> 3, ch = id & 7;
      *reinterpret_cast<uint4*>(Vs + lds_swz<128>(drow, ch)) = rv[i];
    }
  };

  if (ntiles > 0) {
    if constexpr (DMA) {
      dma_tile(0);
    } else {
      load_tile(0);
      store_tile(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMA pieces of this wave have landed (no-op for the register path)
  __syncthreads();

  const f16x2 ones = {(f16)1.0f, (f16)1.0f};

  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) {
      if constexpr (DMA) dma_tile(cur ^ 1);  // buffer cur^1 was last read before the barrier that ended the previous iteration
      else load_tile(t + 1);
    }
    const unsigned char* Ks = smem + cur * (K_BYTES + V_BYTES);
    const unsigned char* Vs = Ks + K_BYTES;
    const int j0 = t * KT;
    const bool need_mask = (j0 + KT > p.Nk) || (p.causal && j0 + KT - 1 > q0);  // block-uniform
    if (DMA && !VROW && j0 + KT > p.Nk) {  // (row-major V: rows past Nk lie beyond the descriptor and arrive as zeros)
      // last tile of a ragged key count (block-uniform, rare): the DMA brought V^T's pad columns in as they are, and they are not
      // trusted (P is exactly 0 there, but 0 * NaN is NaN): clear the dead keys of the tile in LDS before anyone reads it
      unsigned char* Vw = smem + cur * (K_BYTES + V_BYTES) + K_BYTES;
      for (int idx = tid; idx < D * 8; idx += NT) {
        const int row = idx >> 3, ch = idx & 7, kb = j0 + ch * 8;
        if (kb + 8 > p.Nk) {
          f16x8* ptr = reinterpret_cast<f16x8*>(Vw + lds_swz<128>(row, ch));
          f16x8 e = *ptr;
#pragma unroll
          for (int x = 0; x < 8; ++x)
            if (kb + x >= p.Nk) e[x] = (f16)0.0f;
          *ptr = e;
        }
      }
      __syncthreads();
    }

    // ---- S'^T = (K . (cQ)^T) - m for the two 32-key sub-tiles: the scale c rides in the Q fragments and the running reference m
    // enters as the accumulator init of the first MFMA, so the accumulators come out as exponents; each K fragment feeds TQ MFMAs
    f32x16 s[TQ][2];
    auto scores = [&]() {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + lds_swz<ROWB>(u * 32 + l31, ks * 2 + hi));
#pragma unroll
          for (int tq = 0; tq < TQ; ++tq)
            s[tq][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[tq][ks], ks == 0 ? negm[tq] : s[tq][u], 0, 0, 0);
        }
    };
    // P = exp2(S') packed to f16 (the PV B operand), returns this lane's part of the row sum (of the f16 values that enter PV);
    // accumulator r of sub-tile u holds key j0 + 32u + 16(r>>3) + 8hi + (r&7)
    f16x8 pf[TQ][2][2];
    float psum[TQ];
    // (row sum on the f32 exponentials with v_pk_add_f32 instead of v_dot2c on the packed pairs: measured 1-2 % slower -- the pairs have to
    // stay live in f32 until the add, 76 bytes of scratch at three waves per SIMD)
    auto exps = [&](int tq) {
      float acc = 0.0f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f16x2 pp;
          pp[0] = (f16)__builtin_amdgcn_exp2f(s[tq][u][r]);
          pp[1] = (f16)__builtin_amdgcn_exp2f(s[tq][u][r + 1]);
          acc = __builtin_amdgcn_fdot2(pp, ones, acc, false);
          pf[tq][u][r >> 3][r & 7] = pp[0];
          pf[tq][u][r >> 3][(r & 7) + 1] = pp[1];
        }
      psum[tq] = acc;
    };

    scores();
    const bool careful = need_mask || t == 0;  // block-uniform: these tiles always take the max-tracking path
    bool redo = careful;
    if (!careful) {
      // optimistic path: no row max at all.  P <= 2^13 keeps f16 finite and exact enough (relative precision is scale-free; O and
      // l carry the same factor); a lane sum beyond PLIM (or inf / NaN) means some score outgrew the reference by > 2^8
      bool trig = false;
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq) {
        exps(tq);
        trig |= !(psum[tq] <= PLIM);
      }
      redo = __any(trig);
      if (redo) scores();  // the exponentials overwrote the scores: recompute them (rare)
    }
    if (redo) {
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq) {
        if (need_mask) {
          const int qrow = qw + 32 * tq;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = j0 + 32 * u + 16 * (r >> 3) + 8 * hi + (r & 7);
              const bool dead = (key >= p.Nk) || (p.causal && key > qrow);
              s[tq][u][r] = dead ? -INFINITY : s[tq][u][r];
            }
        }
        float mx = fmaxf(fmaxf(s[tq][0][0], s[tq][0][1]), s[tq][1][0]);
        mx = fmaxf(mx, s[tq][1][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) {
          mx = fmaxf(fmaxf(mx, s[tq][0][r]), s[tq][0][r + 1]);
          mx = fmaxf(fmaxf(mx, s[tq][1][r]), s[tq][1][r + 1]);
        }
        mx = pair_max(mx);  // relative to the current reference; -inf for a row with no live key yet
        // the reference only grows, except on the first tile where it is set (O and l are still zero there)
        const float delta = mx == -INFINITY ? 0.0f : (t == 0 ? mx : fmaxf(mx, 0.0f));
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run[tq] += delta;
        l_run[tq] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[tq][dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[tq][r] = -m_run[tq];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[tq][u][r] -= delta;
        exps(tq);
      }
    }
#pragma unroll
    for (int tq = 0; tq < TQ; ++tq) l_run[tq] += psum[tq];
    // the next tile goes to the other LDS buffer here, not at the end of the iteration: its global loads (issued at the top) have
    // had the S^T MFMAs + softmax to land, and the ds_writes then retire under the P.V MFMAs instead of right in front of the barrier
    if (more) store_tile(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T . P^T; each V^T fragment feeds TQ MFMAs ----------------------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
          f16x8 vf;
          if constexpr (VROW) {
            const unsigned char* q = Vs + (32 * u + 16 * sstep) * 128;
            const attn_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((attn_lds_h4_ptr)(q + troff[dt][0]));
            const attn_h4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((attn_lds_h4_ptr)(q + troff[dt][1]));
            vf = __builtin_bit_cast(f16x8, AttnH8{lo, hi4});
          } else {
            vf = *reinterpret_cast<const f16x8*>(Vs + lds_swz<128>(dt * 32 + l31, u * 4 + sstep * 2 + hi));
          }
#pragma unroll
          for (int tq = 0; tq < TQ; ++tq)
            oacc[tq][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[tq][u][sstep], oacc[tq][dt], 0, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l ----------------------------------------------------------------------------------
#pragma unroll
  for (int tq = 0; tq < TQ; ++tq) {
    const float l_tot = pair_sum(l_run[tq]);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    const int qrow = qw + 32 * tq;
    if (p.lse && hi == 0 && qrow < p.Nq)
      p.lse[((long)b * p.heads + h) * p.Nq + qrow] = l_tot > 0.0f ? m_run[tq] + __builtin_amdgcn_logf(l_tot) : INFINITY;
    if (qrow < p.Nq) {
      f16* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_rs + (long)h * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f16x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (f16)(oacc[tq][dt][4 * g + i] * inv);
          *reinterpret_cast<f16x4*>(op + dt * 32 + 8 * g + 4 * hi) = v;
        }
    }
  }
}

template <int D, int NW, int TQ, bool VROW = false>
void launch_attn(const AttnParams& p, int B, hipStream_t st) {
  constexpr int QB = NW * TQ * 32;
  dim3 grid(((p.Nq + QB - 1) / QB) * p.heads * B);
  hipLaunchKernelGGL((attn_fwd_kernel<D, NW, TQ, VROW>), grid, dim3(NW * 64), 0, st, p);
}

bool stream_default() {
  static const bool on = getenv("GN_ATTN_STREAM") ? atoi(getenv("GN_ATTN_STREAM")) != 0 : true;  // attention_stream.hip for the eligible shapes (GN_ATTN_STREAM=0: this file's kernel everywhere)
  return on;
}

int g_attn_variant = -2;  // -2: not read yet
int attn_variant_override() {
  if (g_attn_variant == -2) {
    // tuning aid: 0 = this file's 4 waves x 32 rows everywhere, 1 = 4 waves x 64 rows, 2 = 8 waves x 32 rows, 4 = attention_stream.hip wherever
    // eligible, 5 = attention_pwg.hip wherever eligible (gn_attention_set_variant changes it at run time)
    const char* e = getenv("GN_ATTN_VARIANT");
    g_attn_variant = e ? atoi(e) : -1;
  }
  return g_attn_variant;
}

// attention_pwg.hip (one wave per SIMD, 64 rows per wave, split blocks for the last round) by default from GN_ATTN_PWG_MIN_KEYS keys on:
// measured against attention_stream.hip on one box (tools/probes/attn_pwg_rounds.py, profiles/r05_v12_attn_pwg_rounds.txt), 4 096 keys:
// 8 x 5 heads 172 vs 183 us, 8 x 10 336 vs 360, 4 x 5 96 vs 101, 2 x 5 53 vs 60, 1 x 5 31.5 vs 37; at 1 024 keys it loses (35.5 vs 31.7)
// Inside the B = 1 call (two streams of the recorded program run side by side) its all-split grid of 160 blocks -- 31.5 vs 37 us alone -- is
// no gain (tiled B = 1 28.76 vs 28.58 ms, same box, alternating: a block takes a CU's whole register file and 96 KB of its LDS, the
// other stream's launches wait), so grids below GN_ATTN_PWG_MIN_BLOCKS 256-row blocks stay with attention_stream.hip; the B = 8 call
// gets 93.87 vs 94.64 ms (profiles/r05_v12_ab_attn_pwg.txt).
int pwg_min_keys() {
  static const int v = getenv("GN_ATTN_PWG_MIN_KEYS") ? atoi(getenv("GN_ATTN_PWG_MIN_KEYS")) : 2048;
  return v;
}
int pwg_min_blocks() {
  static const int v = getenv("GN_ATTN_PWG_MIN_BLOCKS") ? atoi(getenv("GN_ATTN_PWG_MIN_BLOCKS")) : 256;
  return v;
}

}  // namespace

extern "C" int32_t gn_attention_set_variant(int32_t variant) {
  const int prev = attn_variant_override();
  g_attn_variant = variant < -1 ? -1 : variant;
  return prev;
}

int32_t gn_launch_attention(gn_ctx* ctx, const gn_attn_desc* d) {
  GN_REQUIRE(d && d->q && d->k && d->vt && d->o, "gn_attention_fwd: null pointer");
  GN_REQUIRE(d->D == 64 || d->D == 32, "gn_attention_fwd: head dim %d unsupported (32 or 64)", d->D);
  GN_REQUIRE(d->B > 0 && d->heads > 0 && d->Nq > 0 && d->Nk > 0, "gn_attention_fwd: empty problem");
  GN_REQUIRE(d->q_rs % 8 == 0 && d->k_rs % 8 == 0 && d->vt_rs % 8 == 0 && d->o_rs % 4 == 0, "gn_attention_fwd: row strides must be multiples of 8 (o: 4)");
  if (d->v_rowmajor) GN_REQUIRE(d->D == 64 && d->vt_rs >= d->heads * d->D, "gn_attention_fwd: row-major V needs D = 64 and a row stride >= heads * D");
  else GN_REQUIRE(d->vt_rs >= ((d->Nk + 63) / 64) * 64, "gn_attention_fwd: vt row stride %d must cover round_up(Nk=%d, 64)", d->vt_rs, d->Nk);
  GN_REQUIRE(((uintptr_t)d->q & 15) == 0 && ((uintptr_t)d->k & 15) == 0 && ((uintptr_t)d->vt & 15) == 0 && ((uintptr_t)d->o & 7) == 0, "gn_attention_fwd: pointer alignment");
  GN_REQUIRE(d->q_bs % 8 == 0 && d->k_bs % 8 == 0 && d->vt_bs % 8 == 0 && d->o_bs % 4 == 0, "gn_attention_fwd: batch strides alignment");
  GN_REQUIRE(d->scale > 0.0f, "gn_attention_fwd: scale must be positive");
  AttnParams p;
  p.q = (const f16*)d->q; p.k = (const f16*)d->k; p.vt = (const f16*)d->vt; p.o = (f16*)d->o;
  p.q_bs = d->q_bs; p.k_bs = d->k_bs; p.vt_bs = d->vt_bs; p.o_bs = d->o_bs;
  p.q_rs = d->q_rs; p.k_rs = d->k_rs; p.vt_rs = d->vt_rs; p.o_rs = d->o_rs;
  p.heads = d->heads; p.Nq = d->Nq; p.Nk = d->Nk; p.causal = d->causal;
  p.scale_log2 = d->scale * 1.4426950408889634f;
  p.lse = d->lse;
  if (d->D == 32) {
    launch_attn<32, 4, 1>(p, d->B, ctx->stream);
  } else {
    // measured on MI355X (tools/bench_attn.py, 8 x 5 x 4096^2 / 8 x 10 x 1024^2): 4 waves x 32 rows 220 / 39 us, 4 waves x 64 rows
    // 268 / 52 (occupancy 1), 8 waves x 32 rows 252 / 46; attention_stream.hip 194 / 32 against 218 / 35 on one box (round 3).
    const int ov = attn_variant_override();
    // (the block-shape overrides exist for V^T only: a row-major V -- the cross-attention reading the combined K | V projection -- keeps its kernel;
    //  taking it as V^T read far outside the tensor: tools/probes/cross_attn_variants.py faulted on it)
    if (ov == 1 && !d->v_rowmajor) launch_attn<64, 4, 2>(p, d->B, ctx->stream);
    else if (ov == 2 && !d->v_rowmajor) launch_attn<64, 8, 1>(p, d->B, ctx->stream);
    else if (!d->causal && !d->v_rowmajor && d->Nk % 64 == 0 && d->Nk >= 128 &&
             (ov == 5 || (ov < 0 && pwg_min_keys() > 0 && d->Nk >= pwg_min_keys() && (long)((d->Nq + 255) / 256) * d->heads * d->B >= pwg_min_blocks())))
      gn_launch_attention_pwg(p, d->B, ctx->stream);
    else if ((ov == 4 || (ov < 0 && stream_default())) && !d->causal && !d->v_rowmajor && d->Nk % 64 == 0 && d->Nk >= 128) gn_launch_attention_stream(p, d->B, ctx->stream);
    else if (d->v_rowmajor) launch_attn<64, 4, 1, true>(p, d->B, ctx->stream);
    else launch_attn<64, 4, 1>(p, d->B, ctx->stream);
  }
  GN_LAUNCH_CHECK();
  return GN_OK;
}
