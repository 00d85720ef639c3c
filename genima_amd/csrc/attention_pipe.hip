// Flash-style attention forward for head dim 64 on gfx950: the main kernel of the U-Net / ControlNet / CLIP-H self- and
// cross-attention.  Same transposed formulation, LDS image and optimistic softmax as attention.hip:
//   S'^T[key, q] = K_tile . (cQ)^T - m    (A = K rows from LDS, B = Q fragments pre-multiplied by c = scale * log2 e, C init = -m)
//   P = exp2(S'),   O^T[d, q] += V^T_tile . P^T
//
// Why a second kernel.  PMC on the generic kernel (MI355X, 8 x 5 x 4096^2): matrix pipe busy 40 % + VALU issue active 59 % = 100 % --
// MFMA and VALU instructions share a SIMD's issue port and the oldest wave owns it, so the QK^T / softmax / PV phases of
// co-resident waves run one after the other, not beside each other.  Here the two pipes overlap INSIDE each wave: the kernel is
// software-pipelined over 32-key sub-tiles, and stage j is ONE instruction stream made of
//     PV(j - 1)     4 MFMA   P of the previous sub-tile . V^T
//     QK^T(j + 1)   4 MFMA   exponents of the next sub-tile
//     softmax(j)    16 v_exp_f32 + 8 v_cvt_pk_f16_f32 + 8 v_dot2c_f32_f16
//     8 ds_read_b128         the K / V^T fragments of stage j + 1
// issued as 8 x { MFMA, <= 5 other instructions } (an MFMA holds the matrix pipe for 32 cycles = 8 issue slots, about 5 of which take
// other instructions for free), pinned with scheduling fences.  The optimistic softmax has no row max, so the three pieces do not
// depend on each other; the MFMAs never wait for LDS because their fragments were fetched a stage earlier.
//
// LDS: the tile image that serves stages 2t+1 and 2t+2 is X(t) = {K(t+1), V^T(t)} (16 KB); two images are double-buffered by
// LDS-DMA with ONE barrier per 64 keys, placed between the two stages: by then stage 2t+2's fragments are already in registers,
// so X(t)'s buffer can be refilled with X(t+2) at once, a full iteration ahead of its first use.
//
// The exponents of sub-tile j stay live until its row sums have been checked.  A flagged sub-tile (lane sum > 2^13, inf or NaN), the
// first one and masked ones take the careful path: mask, row max, move the reference (rescale O^T and l) and shift the already
// computed exponents of sub-tile j + 1 by the same amount.
#include <type_traits>

#include "attention_common.h"

namespace {

struct Frags {
  f16x8 k[4];  // K rows of one 32-key sub-tile, k16 step ks
  f16x8 v[4];  // V^T: d tile (i & 1), k16 step (i >> 1) of one 32-key sub-tile
};

#define GN_FENCE __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256) void attn_fwd_pipe_kernel(const AttnParams p) {
  constexpr int NW = 4, NT = 256, QB = 128;
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

  int nk_eff = p.Nk;
  if (p.causal) nk_eff = min(p.Nk, q0 + QB);  // causal rows never look past their own index
  const int ntiles = (nk_eff + KT - 1) / KT;

  // LDS-DMA pieces of this wave: rows 8 (wave + 4 i) .. + 8 of a K tile / a V^T tile.  A DMA instruction fills 8 consecutive
  // 128-byte LDS rows lane-linearly, so K's row permutation (key bits 2 <-> 3) and the XOR chunk swizzle are applied on the source
  // side.  Keys >= Nk lie past the K descriptor's extent and read as zeros.
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
  const long vbytes = ((long)63 * p.vt_rs + (long)((p.Nk + KT - 1) / KT) * KT) * 2;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)vbytes, 0x00020000);
  auto dma_k = [&](int tile, int buf) {
    const unsigned adv = (unsigned)tile * (unsigned)(KT * p.k_rs * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(smem + buf * BUF + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
  };
  // piece i of the image X(tile) = {K(tile + 1), V^T(tile)}: i < 2 K rows, else V^T rows.  Unconditional: tiles past the end read
  // zeros (beyond the descriptors) or rows nobody consumes, into a ring slot that is free anyway.
  auto dma_piece = [&](int tile, int buf, int i) {
    if (i < 2) {
      const unsigned adv = (unsigned)(tile + 1) * (unsigned)(KT * p.k_rs * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(smem + buf * BUF + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
    } else {
      const unsigned adv = (unsigned)tile * (unsigned)(KT * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (attn_lds_ptr_t)(smem + buf * BUF + K_BYTES + (wv + NW * (i - 2)) * 1024), 16, voff[i - 2] + adv, 0, 0, 0);
    }
  };
  // V^T pad columns of a ragged last tile arrive as they are and are not trusted (P is exactly 0 there, but 0 * NaN is NaN)
  auto sanitize_v = [&](int tile, int buf) {
    unsigned char* Vw = smem + buf * BUF + K_BYTES;
    for (int idx = tid; idx < 64 * 8; idx += NT) {
      const int row = idx >> 3, ch = idx & 7, kb = tile * KT + ch * 8;
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
  };

  // fragment i of the 32-key sub-tile u of the tile image at X: i < 4 K rows (k16 step i), i >= 4 V^T (d tile, k16 step)
  // The 8 per-lane byte offsets are loop invariants (the swizzle only involves row bits 1..3, so the sub-tile / d-tile row offsets are
  // plain +4096 immediates); with the buffer index a compile-time constant in the main loop a fragment read is one ds_read_b128.
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
  // block-uniform: does sub-tile j need masking (ragged key count / causal diagonal)?
  auto masked = [&](int j) -> bool {
    const int tile = j >> 1;
    return (tile * KT + KT > p.Nk) | ((p.causal != 0) & (tile * KT + KT - 1 > q0));
  };
  // careful softmax of sub-tile j from its exponents sc: mask, row max, move the reference, P -> pc; sn = exponents of sub-tile j + 1
  // that were computed against the old reference (shifted here).  Returns the lane's row-sum part.
  auto careful = [&](auto has_next, f32x16& sc, f32x16& sn, f16x8 (&pc)[2], int j) __attribute__((always_inline)) -> float {
    if (masked(j)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * j + 16 * (r >> 3) + 8 * hi + (r & 7);
        const bool dead = (key >= p.Nk) || (p.causal && key > qrow);
        sc[r] = dead ? -INFINITY : sc[r];
      }
    }
    float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
    mx = pair_max(mx);  // relative to the current reference; -inf for a row with no live key yet
    // the reference only grows, except on the first sub-tile, which sets it (O and l are still zero there)
    const float delta = mx == -INFINITY ? 0.0f : (j == 0 ? mx : fmaxf(mx, 0.0f));
    const float alpha = __builtin_amdgcn_exp2f(-delta);
    m_run += delta;
    l_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      oacc[0][r] *= alpha;
      oacc[1][r] *= alpha;
      negm[r] -= delta;
      sc[r] -= delta;
      if constexpr (decltype(has_next)::value) sn[r] -= delta;
    }
    return exps(sc, pc);
  };
  const std::true_type yes{};
  const std::false_type no{};

  // One steady-state stage: softmax of sub-tile j (exponents sc -> pc), QK^T of sub-tile j + 1 (fc.k -> sn), PV of sub-tile j - 1
  // (pp, fc.v), and the fragment reads of stage j + 1 (sub-tile un of the tile image at Xn -> fn).  Units of the softmax per pair k
  // of scores: E(k) two v_exp_f32, C(k) one v_cvt_pk_f16_f32, S(k) one v_dot2c_f32_f16 -- each a gap or two behind its producer.
  // MFMA order PV(d 0), QK^T, PV(d 1), QK^T, ...: neighbours never share an accumulator.
  auto stage = [&](f32x16& sc, f32x16& sn, f16x8 (&pc)[2], const f16x8 (&pp)[2], const Frags& fc, Frags& fn, int j,
                   const unsigned char* Xn, int un, auto dma_buf_c, int dma_tile) __attribute__((always_inline)) {
    constexpr int DMA_BUF = decltype(dma_buf_c)::value;  // >= 0: this stage also issues the 4 LDS-DMA pieces of X(dma_tile)
    auto D = [&](int i) {
      if constexpr (DMA_BUF >= 0) dma_piece(dma_tile, DMA_BUF, i);
    };
    const bool slow = masked(j);
    float ex[16], psum = 0.0f;
    f16x2 pk[8];
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
    auto S = [&](int k) { psum = __builtin_amdgcn_fdot2(pk[k], ones, psum, false); };
    auto M = [&](int i) {  // QK^T first: the last exponent MFMA then retires a slot before the next stage's first v_exp reads it
      const int n = i >> 1;
      if ((i & 1) == 0) sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(fc.k[n], qf[n], n == 0 ? negm : sn, 0, 0, 0);
      else oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fc.v[n], pp[n >> 1], oacc[n & 1], 0, 0, 0);
    };
    auto R = [&](int i) {
      if (i < 4) fn.k[i] = frag(Xn, un, i);
      else fn.v[i - 4] = frag(Xn, un, i);
    };
    GN_FENCE; M(0); GN_FENCE; R(0); R(1); E(0);
    GN_FENCE; M(1); GN_FENCE; R(2); R(3); E(1); C(0);
    GN_FENCE; M(2); GN_FENCE; R(4); R(5); E(2); C(1);
    GN_FENCE; M(3); GN_FENCE; R(6); R(7); E(3); C(2);
    GN_FENCE; M(4); GN_FENCE; D(0); E(4); C(3); S(0); S(1);
    GN_FENCE; M(5); GN_FENCE; D(1); E(5); C(4); S(2); S(3);
    GN_FENCE; M(6); GN_FENCE; D(2); E(6); C(5); S(4); S(5);
    GN_FENCE; M(7); GN_FENCE; D(3); E(7); C(6); S(6);
    GN_FENCE; C(7); S(7);
    GN_FENCE;
    // a masked tile, or some lane sum out of range: one compare, one branch on vcc (the fast path falls through)
    const float chk = slow ? INFINITY : psum;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(chk <= PLIM)) != 0, 0)) psum = careful(yes, sc, sn, pc, j);
    l_run += psum;
  };

  // ---- prologue: K(0) -> ring slot 2 (read here only), X(0) -> slot 0, X(1) -> slot 1 (stays in flight); sub-tiles 0 and 1 ---------------
  dma_k(0, 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(1, 1, i);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  f32x16 sa, sb;
  f16x8 pa[2], pb[2];
  Frags fa, fb;
  {
    const unsigned char* K0 = smem + 2 * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(K0, 0, ks), qf[ks], ks == 0 ? negm : sa, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(K0, 1, ks), qf[ks], ks == 0 ? negm : sb, 0, 0, 0);
    l_run += careful(yes, sa, sb, pa, 0);  // stage 0: sets the reference
  }
  __syncthreads();  // slot 2 is free
  if (ntiles == 1 && KT > p.Nk) sanitize_v(0, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // fragments of stage 1: K sub-tile 2, V^T sub-tile 0
    if (i < 4) fa.k[i] = frag(smem, 0, i);
    else fa.v[i - 4] = frag(smem, 0, i);
  }

  // ---- steady state: iteration t runs stages 2t+1 and 2t+2 on X(t) (ring slot t % 3) and fetches X(t+2) ---------------------------------
  // Slot (t+2) % 3 held X(t-1), whose last reader was stage 2t-1, in front of the previous barrier: its refill needs no wait and its
  // four DMA instructions ride in stage 2t+1's MFMA gaps (issued back to back after the barrier they cost ~330 cycles per wave).
  // The one barrier per 64 keys publishes X(t+1): every wave has waited for its own pieces (all but the 4 just issued).
  // (the slot index is a compile-time constant: LDS addresses fold into immediates; must inline, or the captures go through scratch)
  auto iteration = [&](auto cur_c, int t) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, nxt = (cur + 1) % 3, fill = (cur + 2) % 3;
    stage(sb, sa, pb, pa, fa, fb, 2 * t + 1, smem + cur * BUF, 1, std::integral_constant<int, fill>{}, t + 2);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if ((t + 1) * KT + KT > p.Nk) sanitize_v(t + 1, nxt);  // block-uniform: the ragged last tile
    stage(sa, sb, pa, pb, fb, fa, 2 * t + 2, smem + nxt * BUF, 0, std::integral_constant<int, -1>{}, 0);
  };
  for (int t = 0; t + 1 < ntiles; t += 3) {
    iteration(std::integral_constant<int, 0>{}, t);
    if (t + 2 < ntiles) iteration(std::integral_constant<int, 1>{}, t + 1);
    if (t + 3 < ntiles) iteration(std::integral_constant<int, 2>{}, t + 2);
  }

  // ---- drain: softmax of the last sub-tile, PV of the last two ---------------------------------------------------------------------------
  {
    const int t = ntiles - 1, j = 2 * t + 1;
    const unsigned char* X = smem + (t % 3) * BUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the wave
    float psum = exps(sb, pb);
    if ((__builtin_amdgcn_ballot_w64(!(psum <= PLIM)) != 0) | masked(j)) psum = careful(no, sb, sa, pb, j);
    l_run += psum;
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.v[n], pa[n >> 1], oacc[n & 1], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(X, 1, 4 + n), pb[n >> 1], oacc[n & 1], 0, 0, 0);
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

void gn_launch_attention_pipe(const AttnParams& p, int B, hipStream_t stream) {
  dim3 grid(((p.Nq + 127) / 128) * p.heads * B);
  hipLaunchKernelGGL(attn_fwd_pipe_kernel, grid, dim3(256), 0, stream, p);
}
