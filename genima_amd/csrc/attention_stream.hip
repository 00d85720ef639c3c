// Flash-style attention forward for head dim 64 on gfx950, the self-attention shapes of the U-Net / ControlNet (non-causal, key count a
// multiple of 64): ONE instruction stream per wave in which the matrix pipe and the VALU overlap, and NO branch inside the
// key loop.  Same transposed formulation, LDS image and optimistic softmax as attention.hip:
//   S'^T[key, q] = K_tile . (cQ)^T - m    (A = K rows from LDS, B = Q fragments pre-multiplied by c = scale * log2 e, C init = -m)
//   P = exp2(S'),   O^T[d, q] += V^T_tile . P^T
//
// What the measurements behind this kernel say (tools/probes/attn_phase_model.hip, profiles/r03_v5_attn_phase_model*.txt; register-only
// model of the 64-key iteration, three waves per SIMD, ns per wave tile): phases in program order 389, software-pipelined inside the wave
// 314 (16 MFMAs alone 296), + one ds_read_b128 per MFMA just in time 373 -- 432 with random operand bits, which is what real K / V tiles
// look like to a chip that clocks to its power budget.  A conditional branch per 32-key stage on top of that (the optimistic softmax's "is
// the lane sum in range?") costs +45 % with the smooth operands (544) but +3 ... 6 % with the random ones (450 - 465 against 437), whatever it
// tests; a sticky flag instead costs nothing in either.  On the real kernel: this loop WITH the per-stage branch 206 - 212 us, without 194,
// attention.hip (five branches per 64 keys, phases in program order) 218, at 8 x 5 x 4096^2 on one box.
// So:
//   * the kernel is software-pipelined over 32-key sub-tiles; stage j is one stream of 8 x { MFMA, <= 5 other instructions }:
//       QK^T(j + 1) 4 MFMA, PV(j - 1) 4 MFMA (alternating: neighbours never share an accumulator), softmax(j) = 16 v_exp_f32 +
//       8 v_cvt_pk_f16_f32 + 8 v_dot2c_f32_f16, the 8 fragment reads of its own MFMAs two groups ahead of their use (three fragments
//       live: 164 registers, three waves per SIMD), and in every other stage the four LDS-DMA pieces of the tile two iterations ahead;
//   * the optimistic softmax keeps NO row max and takes NO decision inside the loop: a lane sum out of range (> 2^13, inf, NaN) only sets
//     a sticky scalar flag.  The reference m is taken, before the loop, from the row maxima over the 64 keys of the block's own diagonal
//     tile (the keys at the queries' own positions: for self-attention that is where the maximum lives; any tile is a valid reference);
//   * if any wave of the block ends with the flag set (some score outgrew the reference by more than 2^8 somewhere), the block redoes
//     its rows with the careful max-tracking loop at the end of the kernel -- correctness never depends on the guess, only speed does;
//   * ragged / causal problems have masked tiles, i.e. decisions: they stay with attention.hip.
// LDS: tile image X(t) = {K(t + 1), V^T(t)} (16 KB) in a three-slot ring; iteration t runs stages 2t+1 and 2t+2 on X(t); the barrier
// that closes it publishes X(t+1) and frees X(t)'s slot... for X(t+3), whose DMA rides in stage 2(t+1)+1: every piece has two whole
// iterations to land.  One barrier per 64 keys, one loop branch per 192.
#include <type_traits>

#include "attention_common.h"

namespace {

#define GN_FENCE __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256, 3) void attn_fwd_stream_kernel(const AttnParams p) {
  constexpr int NW = 4, QB = 128;
  constexpr int K_BYTES = KT * 128, V_BYTES = 64 * 128, BUF = K_BYTES + V_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // XCD-aware block order: consecutive block ids go round-robin over the 8 XCDs (each with its own L2), so the query blocks that
  // share one (batch, head)'s K / V^T are given ids that land on ONE XCD, next to each other in dispatch order
  const int nqb = (p.Nq + QB - 1) / QB, total = gridDim.x;
  const int slot = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;
  // (the integer divisions run in the VALU: pin the results back to SGPRs, or the buffer descriptors below turn "divergent")
  const int bh = __builtin_amdgcn_readfirstlane(slot / nqb);
  const int b = __builtin_amdgcn_readfirstlane(bh / p.heads), h = bh - b * p.heads;
  const int q0 = (slot - bh * nqb) * QB;
  const int qrow = q0 + wave * 32 + l31;

  const f16* qp = p.q + (long)b * p.q_bs + (long)h * 64;
  const f16* kp = p.k + (long)b * p.k_bs + (long)h * 64;
  const f16* vp = p.vt + (long)b * p.vt_bs + (long)h * 64 * p.vt_rs;

  f16x8 qf[4];  // (c Q)^T fragments: lane holds Q[qrow][16 ks + 8 hi .. +8] * scale * log2(e)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qrow < p.Nq) v = *reinterpret_cast<const uint4*>(qp + (long)qrow * p.q_rs + ks * 16 + hi * 8);
    f16x8 q8 = *reinterpret_cast<f16x8*>(&v);
#pragma unroll
    for (int x = 0; x < 8; ++x) q8[x] = (f16)((float)q8[x] * p.scale_log2);
    qf[ks] = q8;
  }

  f32x16 oacc[2], negm;  // O^T accumulators (d tiles), and -m as an MFMA accumulator init (all 16 entries equal)
  float m_run = 0.0f, l_run = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = negm[r] = 0.0f;
  const int ntiles = p.Nk / KT;  // the launcher guarantees Nk % 64 == 0, Nk >= 128, not causal

  // LDS-DMA pieces of this wave: rows 8 (wave + 4 i) .. + 8 of a K tile / a V^T tile.  A DMA instruction fills 8 consecutive
  // 128-byte LDS rows lane-linearly, so K's row permutation (key bits 2 <-> 3) and the XOR chunk swizzle are applied on the source side.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wv + NW * i) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int key = (row & 32) | swap23(row & 31);
    koff[i] = (unsigned)(((long)key * p.k_rs + chunk * 8) * 2);
    voff[i] = (unsigned)(((long)row * p.vt_rs + chunk * 8) * 2);
  }
  const long kbytes = ((long)(p.Nk - 1) * p.k_rs + 64) * 2;
  const long vbytes = ((long)63 * p.vt_rs + (long)p.Nk) * 2;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)vbytes, 0x00020000);
  auto dma_k = [&](int tile, int buf) {  // K(tile) into the K part of ring slot buf
    const unsigned adv = (unsigned)tile * (unsigned)(KT * p.k_rs * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(smem + buf * BUF + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
  };
  // piece i of the image X(tile) = {K(tile + 1), V^T(tile)}: i < 2 K rows, else V^T rows.  Unconditional: tiles past the end read
  // zeros (beyond the descriptors) into a ring slot nobody consumes.
  auto dma_piece = [&](int tile, int buf, int i) {
    if (i < 2) {
      const unsigned adv = (unsigned)(tile + 1) * (unsigned)(KT * p.k_rs * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(smem + buf * BUF + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
    } else {
      const unsigned adv = (unsigned)tile * (unsigned)(KT * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (attn_lds_ptr_t)(smem + buf * BUF + K_BYTES + (wv + NW * (i - 2)) * 1024), 16, voff[i - 2] + adv, 0, 0, 0);
    }
  };

  // fragment i of the 32-key sub-tile u of the tile image at X: i < 4 K rows (k16 step i), i >= 4 V^T (d tile (i - 4) & 1, k16 step (i - 4) >> 1)
  int offk[4], offv[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) offk[i] = lds_swz<128>(l31, i * 2 + hi);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int s = 0; s < 2; ++s) offv[u][s] = K_BYTES + lds_swz<128>(l31, u * 4 + s * 2 + hi);
  auto frag = [&](const unsigned char* X, int u, int i) -> f16x8 {
    if (i < 4) return *reinterpret_cast<const f16x8*>(X + offk[i] + u * 4096);
    const int n = i - 4;
    return *reinterpret_cast<const f16x8*>(X + offv[u][n >> 1] + (n & 1) * 4096);
  };

  const f16x2 ones = {(f16)1.0f, (f16)1.0f};
  // P = exp2(S') of one sub-tile, packed to f16 (the PV B operand: accumulator r holds key 32 j + 16 (r >> 3) + 8 hi + (r & 7), i.e.
  // 8 consecutive keys per k16 step); returns this lane's part of the row sum (of the f16 values that enter PV)
  auto exps = [&](const f32x16& s, f16x8 (&pf)[2]) -> float {
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f16x2 pp;
      pp[0] = (f16)__builtin_amdgcn_exp2f(s[r]);
      pp[1] = (f16)__builtin_amdgcn_exp2f(s[r + 1]);
      acc = __builtin_amdgcn_fdot2(pp, ones, acc, false);
      pf[r >> 3][r & 7] = pp[0];
      pf[r >> 3][(r & 7) + 1] = pp[1];
    }
    return acc;
  };
  auto rowmax16 = [&](const f32x16& s) -> float {
    float mx = fmaxf(s[0], s[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
    return mx;
  };
  unsigned long long sticky = 0;  // some lane sum left the range the optimistic softmax is exact in (wave-uniform, never branched on in the loop)

  // One steady-state stage on the tile image X, sub-tile U: softmax of sub-tile j (exponents sc -> pc), QK^T of sub-tile j + 1 (K part of
  // X -> sn), PV of sub-tile j - 1 (pp, V^T part of X).  f0 / f1: the fragments of the first two MFMAs, read by the caller; with MORE,
  // groups 6 and 7 read the first two fragments of the next stage (same image, sub-tile 1) into g0 / g1.  Units of the softmax per pair
  // k of scores: E(k) two v_exp_f32, C(k) one v_cvt_pk_f16_f32, S(k) one v_dot2c_f32_f16 -- each a group or two behind its producer.
  auto stage = [&](const f32x16& sc, f32x16& sn, f16x8 (&pc)[2], const f16x8 (&pp)[2], const unsigned char* X, auto u_c, f16x8 f0, f16x8 f1,
                   auto more_c, f16x8& g0, f16x8& g1, auto dma_buf_c, int dma_tile) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value;
    constexpr bool MORE = decltype(more_c)::value;
    constexpr int DMA_BUF = decltype(dma_buf_c)::value;  // >= 0: this stage also issues the 4 LDS-DMA pieces of X(dma_tile)
    auto D = [&](int i) {
      if constexpr (DMA_BUF >= 0) dma_piece(dma_tile, DMA_BUF, i);
    };
    float ex[16], psum = 0.0f;
    f16x2 pk[8];
    f16x8 f[8];
    f[0] = f0; f[1] = f1;
    auto FI = [](int i) { return (i & 1) ? 4 + (i >> 1) : (i >> 1); };  // fragment of MFMA i: even QK^T k16 step, odd V^T (d tile, k16 step)
    auto R = [&](int i) {
      if (i < 8) f[i] = frag(X, U, FI(i));
      else if constexpr (MORE) { if (i == 8) g0 = frag(X, 1, FI(0)); else g1 = frag(X, 1, FI(1)); }
    };
    auto E = [&](int k) {
      ex[2 * k] = __builtin_amdgcn_exp2f(sc[2 * k]);
      ex[2 * k + 1] = __builtin_amdgcn_exp2f(sc[2 * k + 1]);
    };
    auto C = [&](int k) {
      pk[k][0] = (f16)ex[2 * k];
      pk[k][1] = (f16)ex[2 * k + 1];
      pc[k >> 2][(2 * k) & 7] = pk[k][0];
      pc[k >> 2][((2 * k) & 7) + 1] = pk[k][1];
    };
#ifdef GN_ATTN_SUM_F32
    auto S = [&](int k) { psum += ex[2 * k] + ex[2 * k + 1]; };  // (experiment: the row sum from the f32 exponentials, two v_add_f32 instead of one v_dot2c_f32_f16)
#else
    auto S = [&](int k) { psum = __builtin_amdgcn_fdot2(pk[k], ones, psum, false); };
#endif
    auto M = [&](int i) {
      const int n = i >> 1;
      if ((i & 1) == 0) sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], qf[n], n == 0 ? negm : sn, 0, 0, 0);
      else oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], pp[n >> 1], oacc[n & 1], 0, 0, 0);
    };
    GN_FENCE; M(0); GN_FENCE; R(2); D(0); E(0);
    GN_FENCE; M(1); GN_FENCE; R(3); D(1); E(1); C(0);
    GN_FENCE; M(2); GN_FENCE; R(4); D(2); E(2); C(1); S(0);
    GN_FENCE; M(3); GN_FENCE; R(5); D(3); E(3); C(2); S(1);
    GN_FENCE; M(4); GN_FENCE; R(6); E(4); C(3); S(2);
    GN_FENCE; M(5); GN_FENCE; R(7); E(5); C(4); S(3);
    GN_FENCE; M(6); GN_FENCE; R(8); E(6); C(5); S(4);
    GN_FENCE; M(7); GN_FENCE; R(9); E(7); C(6); S(5);
    GN_FENCE; C(7); S(6); S(7);
    GN_FENCE;
    sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));  // v_cmp + s_or: no branch
    l_run += psum;
  };
  const std::true_type yes{};
  const std::false_type no{};

  // ---- the reference: row maxima over the block's own diagonal keys (waves 0, 1: tile q0 / 64; waves 2, 3: the next one) ------------------
  const int tref0 = min(q0 / KT, ntiles - 1), tref1 = min(q0 / KT + 1, ntiles - 1);
  dma_k(tref0, 0);
  dma_k(tref1, 1);
  dma_k(0, 2);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // the two reference tiles of this wave's pieces (K(0) may still be on its way)
  __syncthreads();
  f32x16 sa, sb;
  {
    const unsigned char* Kr = smem + (wv >> 1) * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(Kr, 0, ks), qf[ks], ks == 0 ? negm : sa, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(Kr, 1, ks), qf[ks], ks == 0 ? negm : sb, 0, 0, 0);
    m_run = pair_max(fmaxf(rowmax16(sa), rowmax16(sb)));
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -m_run;
  }
  __syncthreads();  // slots 0 and 1 are free

  // ---- prologue: X(0) -> slot 0, X(1) -> slot 1 (stays in flight); exponents of sub-tiles 0 and 1 from K(0) in slot 2; P of sub-tile 0 -------
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(1, 1, i);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  f16x8 pa[2], pb[2];
  {
    const unsigned char* K0 = smem + 2 * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(K0, 0, ks), qf[ks], ks == 0 ? negm : sa, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(K0, 1, ks), qf[ks], ks == 0 ? negm : sb, 0, 0, 0);
    const float psum = exps(sa, pa);
    sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));
    l_run += psum;
  }
  __syncthreads();  // slot 2 is free

  // ---- steady state: iteration t runs stages 2t+1 and 2t+2 on X(t) (ring slot t % 3) and fetches X(t+2); no branch inside -------------------
  // X(t+2) goes to slot (t + 2) % 3, which held X(t-1): every wave left it before the barrier that closed iteration t - 1.  The barrier
  // that closes iteration t publishes X(t+1): every wave has waited for its own pieces of it (all but the 4 of X(t+2) just issued).
  // (the slot index is a compile-time constant: LDS addresses fold into immediates; must inline, or the captures go through scratch)
  auto iteration = [&](auto cur_c, int t) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, fill = (cur + 2) % 3;
    const unsigned char* X = smem + cur * BUF;
    f16x8 a0 = frag(X, 0, 0), a1 = frag(X, 0, 4), b0, b1, dummy0, dummy1;
    stage(sb, sa, pb, pa, X, std::integral_constant<int, 0>{}, a0, a1, yes, b0, b1, std::integral_constant<int, fill>{}, t + 2);
    stage(sa, sb, pa, pb, X, std::integral_constant<int, 1>{}, b0, b1, no, dummy0, dummy1, std::integral_constant<int, -1>{}, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
  };
  const int nit = ntiles - 1;
  int t = 0;
  for (; t + 3 <= nit; t += 3) {
    iteration(std::integral_constant<int, 0>{}, t);
    iteration(std::integral_constant<int, 1>{}, t + 1);
    iteration(std::integral_constant<int, 2>{}, t + 2);
  }
  if (t < nit) iteration(std::integral_constant<int, 0>{}, t);
  if (t + 1 < nit) iteration(std::integral_constant<int, 1>{}, t + 1);

  // ---- drain: softmax of the last sub-tile, PV of the last two ---------------------------------------------------------------------------
  {
    const unsigned char* X = smem + (nit % 3) * BUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the wave
    const float psum = exps(sb, pb);
    sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));
    l_run += psum;
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(X, 0, 4 + n), pa[n >> 1], oacc[n & 1], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(X, 1, 4 + n), pb[n >> 1], oacc[n & 1], 0, 0, 0);
  }

  // ---- the guess failed somewhere in this block (rare): redo its rows with the max-tracking loop ---------------------------------------------
  // block-wide OR of the flags through the ring itself (an extra __shared__ word -- what __syncthreads_or allocates -- pushes the block
  // past a third of the CU's LDS: two blocks per CU instead of three, measured +17 %)
  __syncthreads();  // every wave is done with the ring
  if (lane == 0) reinterpret_cast<int*>(smem)[wv] = sticky != 0;
  __syncthreads();
  const int4 flags = *reinterpret_cast<const int4*>(smem);
  if (__builtin_amdgcn_readfirstlane(flags.x | flags.y | flags.z | flags.w)) {
    __syncthreads();  // the flags have been read
    m_run = 0.0f; l_run = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = negm[r] = 0.0f;
    for (int tt = 0; tt < ntiles; ++tt) {
      dma_k(tt, 0);
      dma_piece(tt, 0, 2);
      dma_piece(tt, 0, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x16 s;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(smem, u, ks), qf[ks], ks == 0 ? negm : s, 0, 0, 0);
        float mx = pair_max(rowmax16(s));  // relative to the current reference
        const float delta = (tt == 0 && u == 0) ? mx : fmaxf(mx, 0.0f);  // the reference only grows, except on the first sub-tile, which sets it
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          oacc[0][r] *= alpha;
          oacc[1][r] *= alpha;
          negm[r] -= delta;
          s[r] -= delta;
        }
        f16x8 pc[2];
        l_run += exps(s, pc);
#pragma unroll
        for (int n = 0; n < 4; ++n) oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(smem, u, 4 + n), pc[n >> 1], oacc[n & 1], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l -------------------------------------------------------------------------------------------
  const float l_tot = pair_sum(l_run);
  const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (p.lse && hi == 0 && qrow < p.Nq)
    p.lse[((long)b * p.heads + h) * p.Nq + qrow] = l_tot > 0.0f ? m_run + __builtin_amdgcn_logf(l_tot) : INFINITY;
  if (qrow < p.Nq) {
    f16* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_rs + (long)h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (f16)(oacc[dt][4 * g + i] * inv);
        *reinterpret_cast<f16x4*>(op + dt * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

#undef GN_FENCE

}  // namespace

void gn_launch_attention_stream(const AttnParams& p, int B, hipStream_t stream) {
  dim3 grid(((p.Nq + 127) / 128) * p.heads * B);
  hipLaunchKernelGGL(attn_fwd_stream_kernel, grid, dim3(256), 0, stream, p);
}
