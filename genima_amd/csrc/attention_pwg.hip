// Flash-style attention forward for head dim 64 on gfx950, the LARGE self-attention problems of the U-Net / ControlNet (the 64 x 64 latent
// level at B >= 4: 4 096 keys, >= 256 query blocks): ONE WAVE PER SIMD, 64 query rows per wave, the whole register file per wave.
//
// Why another formulation (round 5).  attention_stream.hip runs 32 query rows per wave at three waves per SIMD: every
// 32-key step of a wave is 8 MFMAs beside 8 ds_read_b128 fragment reads + the softmax, and three co-resident waves arbitrate for one
// matrix pipe and one VALU issue port (MI355X_MICROARCH.md "Two waves per SIMD": moving work between the waves of a SIMD is zero-sum).
// Here a wave owns TWO 32-row query blocks (a, b) that share every K / V^T fragment it reads: the LDS fragment traffic, the LDS-DMA
// pieces and the barriers per MFMA all halve, and the in-order stream of the one wave on a SIMD is the schedule -- nothing arbitrates.
//
// Same arithmetic as attention.hip / attention_stream.hip (transposed formulation, optimistic softmax, K rows with key bits 2 <-> 3
// swapped, V^T operand):
//   S'^T[key, q] = K_tile . (cQ)^T - m    (A = K rows from LDS, B = Q fragments pre-multiplied by c = scale * log2 e, C init = -m)
//   P = exp2(S'),   O^T[d, q] += V^T_tile . P^T
// The software pipeline runs over UNITS (s, x) = (32-key sub-tile s, query block x) in the order (0,a) (0,b) (1,a) (1,b) ...:
// the half-stage of unit u holds softmax(u) in the VALU beside QK^T(u + 1) and P.V(u - 1) on the matrix pipe -- 8 x { MFMA, a handful of
// other instructions }, neighbouring MFMAs never share an accumulator.  Two S accumulators and two P fragments are live in total (as in
// the 32-row kernel).  The fragment set F(s) = { K sub-tile s + 1, V^T sub-tile s } feeds the PAIR STAGE [(s,b), (s+1,a)]: its 8
// ds_read_b128 are issued in the second half of the previous pair stage, each right behind the MFMA that consumed the register's
// previous contents (8 MFMA slots of flight).
// LDS: tile image X(t) = { K keys 64 t + 32 .. 64 t + 95, V^T keys 64 t .. 64 t + 63 } (16 KB) = the fragment sets F(2t), F(2t + 1), in a
// three-slot ring.  Iteration t runs the pair stages 2t and 2t + 1; ONE barrier per 64 keys sits BETWEEN them: it publishes X(t + 1)
// (whose pieces were issued a whole iteration earlier) and retires X(t - 1), whose slot takes the LDS-DMA of X(t + 2) in the gaps of
// the following half-stage.  No fragment read is ever exposed behind a barrier.
// The reference m per query row comes from the row maxima over the block's own diagonal keys (one 64-key tile per wave), no decision is
// taken inside the loop (a sticky flag), and a block whose guess failed redoes its rows with the max-tracking loop (as attention_stream.hip).
#include <type_traits>

#include "attention_common.h"

namespace {

#define GN_FENCE __builtin_amdgcn_sched_barrier(0)

// The softmax's conversions and row-sum adds as single pinned instructions: left to the compiler, the SLP vectoriser packs the adds into
// v_pk_add_f32 (an anti-lever beside MFMAs: MI355X_MICROARCH.md price list) and the conversions collect at the end of the half-stage.
// (Inline asm is invisible to the hazard recogniser: a v_exp_f32 result must not be consumed by the very next instruction -- the
// half-stage's order keeps at least one instruction between.)
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float dot2_ones(unsigned pk, float acc) {  // acc + lo(pk) + hi(pk), the halves as f16
  asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(pk), "v"(0x3c003c00u));
  return acc;
}
__device__ __forceinline__ float add_f32(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

// ABL: timing-only ablations (wrong results on purpose; tools/probes/attn_pwg_abl.py): 1 no v_exp, 2 no conversions / row sums, 4 no LDS-DMA
// in the loop, 8 no fragment reads in the loop, 16 no barrier in the loop, 32 no P.V MFMAs, 64 no QK^T MFMAs
template <int ABL>
__global__ __launch_bounds__(256, 1) void attn_fwd_pwg_kernel(const AttnParams p) {
  constexpr int NW = 4, QB = 256, NS = 3;
  constexpr int K_BYTES = KT * 128, V_BYTES = 64 * 128, BUF = K_BYTES + V_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  unsigned long long tk0 = 0, tr0 = 0;
  if constexpr ((ABL & 4096) != 0) { tk0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
  // XCD-aware block order (as attention_stream.hip): the query blocks of one (batch, head) land on ONE XCD, next to each other in dispatch order
  const int nqb = (p.Nq + QB - 1) / QB, total = gridDim.x;
  const int slot = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;
  const int bh = __builtin_amdgcn_readfirstlane(slot / nqb);
  const int b = __builtin_amdgcn_readfirstlane(bh / p.heads), h = bh - b * p.heads;
  const int q0 = (slot - bh * nqb) * QB;
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  const f16* qp = p.q + (long)b * p.q_bs + (long)h * 64;
  const f16* kp = p.k + (long)b * p.k_bs + (long)h * 64;
  const f16* vp = p.vt + (long)b * p.vt_bs + (long)h * 64 * p.vt_rs;

  f16x8 qf[2][4];  // (c Q)^T fragments of the two query blocks: lane holds Q[qrow][16 ks + 8 hi .. +8] * scale * log2(e)
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int qrow = q0 + wv * 64 + x * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow < p.Nq) v = *reinterpret_cast<const uint4*>(qp + (long)qrow * p.q_rs + ks * 16 + hi * 8);
      f16x8 q8 = *reinterpret_cast<f16x8*>(&v);
#pragma unroll
      for (int e = 0; e < 8; ++e) q8[e] = (f16)((float)q8[e] * p.scale_log2);
      qf[x][ks] = q8;
    }
  }

  f32x16 oacc[2][2], negm[2];  // O^T accumulators [query block][d tile], and -m as an MFMA accumulator init (all 16 entries equal)
  float m_run[2] = {0.0f, 0.0f}, l_run[2] = {0.0f, 0.0f};
  constexpr bool LMFMA = (ABL & 2048) != 0;  // the row sums on the matrix pipe: l^T[*, q] += ones . P^T (2 MFMAs per unit, no VALU adds)
  f32x16 lacc[2];
  f16x8 ones8;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones8[e] = (f16)1.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[0][r] = lacc[1][r] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[0][0][r] = oacc[0][1][r] = oacc[1][0][r] = oacc[1][1][r] = negm[0][r] = negm[1][r] = 0.0f;
  const int ntiles = p.Nk / KT;  // the launcher guarantees Nk % 64 == 0, Nk >= 128, not causal

  // LDS-DMA pieces of this wave: rows 8 (wave + 4 i) .. + 8 of a K tile / a V^T tile.  A DMA instruction fills 8 consecutive
  // 128-byte LDS rows lane-linearly, so K's row permutation (key bits 2 <-> 3) and the XOR chunk swizzle are applied on the source side.
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
  // this wave's two pieces of the 64 K rows that start at key key0, into the 8 KB at lds
  auto dma_krows = [&](int key0, unsigned char* lds) {
    const unsigned adv = (unsigned)key0 * (unsigned)(p.k_rs * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(lds + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
  };
  // piece i of the image X(tile) = {K keys 64 tile + 32 .. + 64, V^T keys 64 tile .. + 64}: i < 2 K rows, else V^T rows.  Unconditional:
  // rows past the end read zeros (beyond the descriptor) or a neighbouring row's bytes (V^T) into LDS bytes nobody consumes.
  auto dma_piece = [&](int tile, unsigned char* X, int i) {
    if (i < 2) {
      const unsigned adv = (unsigned)(tile * KT + 32) * (unsigned)(p.k_rs * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(X + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
    } else {
      const unsigned adv = (unsigned)tile * (unsigned)(KT * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (attn_lds_ptr_t)(X + K_BYTES + (wv + NW * (i - 2)) * 1024), 16, voff[i - 2] + adv, 0, 0, 0);
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
  auto FI = [](int i) { return (i & 1) ? 4 + (i >> 1) : (i >> 1); };  // fragment of MFMA i: even QK^T k16 step, odd V^T (d tile, k16 step)

  // P = exp2(S') of one sub-tile, packed to f16 (the PV B operand: accumulator r holds key 32 j + 16 (r >> 3) + 8 hi + (r & 7), i.e.
  // 8 consecutive keys per k16 step); returns this lane's part of the row sum
  auto exps = [&](const f32x16& s, u32x4 (&pf)[2]) -> float {
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float e0 = __builtin_amdgcn_exp2f(s[r]), e1 = __builtin_amdgcn_exp2f(s[r + 1]);
      acc += e0 + e1;
      const f16x2 pk = {(f16)e0, (f16)e1};
      pf[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(unsigned, pk);
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

  // One half-stage = the unit whose softmax runs: exponents sc -> P fragments pc (query block XS); QK^T of the next unit into sn (query
  // block 1 - XS... the block the NEXT unit belongs to: XN); P.V of the previous unit from pp (query block XP).  f[]: the 8 fragments of
  // this pair stage.  READ: this is the pair stage's second half -- behind MFMA i, f[i] is re-read for the NEXT pair stage from image Xn,
  // sub-tile UN (KONLY... VONLY: the drain needs the V^T fragments only).  DMA: the 4 LDS-DMA pieces of X(dma_tile) into Xd ride in the gaps.
  // Units of the softmax per pair k of scores: E(k) two v_exp_f32, C(k) one v_cvt_pk_f16_f32, S(k) the row-sum adds -- each a group or
  // two behind its producer.
  auto half = [&](const f32x16& sc, f32x16& sn, u32x4 (&pc)[2], const u32x4 (&pp)[2], auto xs_c, const f16x8 (&f)[8], f16x8 (&g)[8], auto read_c, const unsigned char* Xn,
                  auto un_c, auto dma_c, unsigned char* Xd, int dma_tile) __attribute__((always_inline)) {
    constexpr int XS = decltype(xs_c)::value;  // the softmax's query block; QK^T goes to the other one's NEXT unit, P.V to the other one's previous
    constexpr int XN = 1 - XS, XP = 1 - XS;
    constexpr int READ = decltype(read_c)::value;  // 0: none, 1: all eight, 2: the V^T fragments only
    constexpr int UN = decltype(un_c)::value;
    constexpr bool DMA = decltype(dma_c)::value;
    float ex[16], ps0, ps1;
    auto D = [&](int i) {
      if constexpr (DMA && !(ABL & 4)) dma_piece(dma_tile, Xd, i);
    };
    auto R = [&](int i) {
      if constexpr (ABL & 8) return;
      if constexpr (READ == 1) g[i] = frag(Xn, UN, FI(i));
      else if constexpr (READ == 2) { if (i & 1) g[i] = frag(Xn, UN, FI(i)); }
    };
    auto E = [&](int k) {
      if constexpr (ABL & 1) { ex[2 * k] = sc[2 * k]; ex[2 * k + 1] = sc[2 * k + 1]; return; }
      ex[2 * k] = __builtin_amdgcn_exp2f(sc[2 * k]);
      ex[2 * k + 1] = __builtin_amdgcn_exp2f(sc[2 * k + 1]);
    };
    auto C = [&](int k) {
      if constexpr (ABL & 2) { if ((k & 3) == 0) pc[k >> 2] = __builtin_bit_cast(u32x4, f32x4{ex[2 * k], ex[2 * k + 1], ex[2 * k + 2], ex[2 * k + 3]}); return; }
      pc[k >> 2][k & 3] = cvt_pk(ex[2 * k], ex[2 * k + 1]);
    };
    auto S = [&](int k) {
      if constexpr ((ABL & 1024) != 0) {  // row sums from the packed f16 pairs (one v_dot2c_f32_f16 per pair, two chains)
        if (k == 0) { ps0 = 0.0f; ps1 = 0.0f; }
        if (k & 1) ps1 = dot2_ones(pc[k >> 2][k & 3], ps1); else ps0 = dot2_ones(pc[k >> 2][k & 3], ps0);
        return;
      }
      if (k == 0 || (ABL & 2)) { ps0 = ex[0]; ps1 = ex[1]; }
      else { ps0 = add_f32(ps0, ex[2 * k]); ps1 = add_f32(ps1, ex[2 * k + 1]); }
    };
    auto M = [&](int i) {
      const int n = i >> 1;
      if ((i & 1) == 0) { if constexpr (!(ABL & 64)) sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], qf[XN][n], n == 0 ? negm[XN] : sn, 0, 0, 0); }
      else { if constexpr (!(ABL & 32)) oacc[XP][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], as_f16x8(pp[n >> 1]), oacc[XP][n & 1], 0, 0, 0); }
    };
    if constexpr ((ABL & 512) != 0 && READ != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // EARLY: the set f was read a pair stage ago
    if constexpr (LMFMA) {
      auto ML = [&](int j) { lacc[XP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, as_f16x8(pp[j]), lacc[XP], 0, 0, 0); };
      GN_FENCE; M(0); GN_FENCE; R(0); D(0); E(0);
      GN_FENCE; M(1); GN_FENCE; R(1); D(1); E(1); C(0);
      GN_FENCE; M(2); GN_FENCE; R(2); D(2); E(2); C(1);
      GN_FENCE; M(3); GN_FENCE; R(3); D(3); C(2);
      GN_FENCE; ML(0); GN_FENCE; R(4); E(3);
      GN_FENCE; M(4); GN_FENCE; R(5); E(4); C(3);
      GN_FENCE; M(5); GN_FENCE; R(6); E(5); C(4);
      GN_FENCE; M(6); GN_FENCE; R(7); E(6); C(5);
      GN_FENCE; M(7); GN_FENCE; E(7); C(6);
      GN_FENCE; ML(1); GN_FENCE; C(7);
      GN_FENCE;
    } else {
      GN_FENCE; M(0); GN_FENCE; R(0); D(0); E(0);
      GN_FENCE; M(1); GN_FENCE; R(1); D(1); E(1); C(0);
      GN_FENCE; M(2); GN_FENCE; R(2); D(2); E(2); C(1); S(0);
      GN_FENCE; M(3); GN_FENCE; R(3); D(3); E(3); C(2); S(1);
      GN_FENCE; M(4); GN_FENCE; R(4); E(4); C(3); S(2);
      GN_FENCE; M(5); GN_FENCE; R(5); E(5); C(4); S(3);
      GN_FENCE; M(6); GN_FENCE; R(6); E(6); C(5); S(4);
      GN_FENCE; M(7); GN_FENCE; R(7); E(7); C(6); S(5);
      GN_FENCE; S(6); C(7); S(7);
      GN_FENCE;
      const float psum = ps0 + ps1;
      sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));  // v_cmp + s_or: no branch
      l_run[XS] += psum;
    }
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  const std::true_type yes{};
  const std::false_type no{};

  // ---- the reference: row maxima over the wave's own diagonal keys (K tile q0 / 64 + wave, clamped) --------------------------------------
  // every wave stages its two pieces of all four reference tiles (4 x 8 KB at the bottom of the ring)
#pragma unroll
  for (int w = 0; w < NW; ++w) dma_krows(min(q0 / KT + w, ntiles - 1) * KT, smem + w * K_BYTES);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sa, sb;
  {
    const unsigned char* Kr = smem + wv * K_BYTES;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(Kr, 0, ks), qf[x][ks], ks == 0 ? negm[x] : sa, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(Kr, 1, ks), qf[x][ks], ks == 0 ? negm[x] : sb, 0, 0, 0);
      m_run[x] = pair_max(fmaxf(rowmax16(sa), rowmax16(sb)));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[0][r] = -m_run[0]; negm[1][r] = -m_run[1]; }
  }
  __syncthreads();  // the ring is free

  // ---- prologue: X(0) -> slot 0, X(1) -> slot 1 (stays in flight); K keys 0 .. 63 -> slot 2: exponents of the units (0,a), (0,b); P of (0,a) ---
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(0, smem, i);
  dma_krows(0, smem + 2 * BUF);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece(1, smem + BUF, i);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  u32x4 pa[2], pb[2];
  f16x8 fa[8], fb[8];
  constexpr bool EARLY = (ABL & 256) != 0;  // fragment reads in the FIRST half of a pair stage, into the other register set
  {
    const unsigned char* K0 = smem + 2 * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(K0, 0, ks), qf[0][ks], ks == 0 ? negm[0] : sa, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(K0, 0, ks), qf[1][ks], ks == 0 ? negm[1] : sb, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = frag(smem, 0, FI(i));  // F(0)
    const float psum = exps(sa, pa);
    if constexpr (!LMFMA) {
      sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));
      l_run[0] += psum;
    }
  }
  __syncthreads();  // slot 2 is free

  // ---- steady state: iteration t = pair stages 2t, 2t + 1 on X(t) (ring slot t % 3); no branch inside ------------------------------------
  // (the slot index is a compile-time constant: LDS addresses fold into immediates; must inline, or the captures go through scratch)
  auto iteration = [&](auto cur_c, int t) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, nxt = (cur + 1) % NS, fill = (cur + 2) % NS;
    unsigned char* X = smem + cur * BUF;
    if constexpr (EARLY) {
      half(sb, sa, pb, pa, c1, fa, fb, c1, X, c1, no, X, 0);                         // (2t, b):     QK^T (2t+1, a), P.V (2t, a); reads F(2t+1) -> fb
      half(sa, sb, pa, pb, c0, fa, fb, c0, X, c0, no, X, 0);                         // (2t+1, a):   QK^T (2t+1, b), P.V (2t, b)
    } else {
      half(sb, sa, pb, pa, c1, fa, fa, c0, X, c0, no, X, 0);                         // (2t, b):     QK^T (2t+1, a), P.V (2t, a)
      half(sa, sb, pa, pb, c0, fa, fa, c1, X, c1, no, X, 0);                         // (2t+1, a):   QK^T (2t+1, b), P.V (2t, b); reads F(2t+1)
    }
    if constexpr (!(ABL & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's pieces of X(t+1), issued an iteration ago
      __syncthreads();                                                        // X(t+1) is published, X(t-1) retired
    }
    if constexpr (EARLY) {
      half(sb, sa, pb, pa, c1, fb, fa, c1, smem + nxt * BUF, c0, no, X, 0);          // (2t+1, b):   QK^T (2t+2, a), P.V (2t+1, a); reads F(2t+2) -> fa
      half(sa, sb, pa, pb, c0, fb, fa, c0, X, c0, yes, smem + fill * BUF, t + 2);    // (2t+2, a):   QK^T (2t+2, b), P.V (2t+1, b); fetches X(t+2)
    } else {
      half(sb, sa, pb, pa, c1, fa, fa, c0, X, c0, yes, smem + fill * BUF, t + 2);    // (2t+1, b):   QK^T (2t+2, a), P.V (2t+1, a); fetches X(t+2)
      half(sa, sb, pa, pb, c0, fa, fa, c1, smem + nxt * BUF, c0, no, X, 0);          // (2t+2, a):   QK^T (2t+2, b), P.V (2t+1, b); reads F(2t+2)
    }
  };
  const int nit = ntiles - 1;
  int t = 0;
  for (; t + 3 <= nit; t += 3) {
    iteration(c0, t);
    iteration(c1, t + 1);
    iteration(c2, t + 2);
  }
  if (t < nit) { iteration(c0, t); ++t; }
  if (t < nit) { iteration(c1, t); ++t; }

  // ---- last tile: pair stage 2 nt - 2, then the drain (softmax of the last unit, P.V of the last two) ---------------------------------------
  {
    unsigned char* X = smem + (nit % NS) * BUF;
    if constexpr (EARLY) {
      half(sb, sa, pb, pa, c1, fa, fb, c2, X, c1, no, X, 0);   // (L-1, b): QK^T (L, a), P.V (L-1, a); reads the V^T fragments of sub-tile L -> fb
      half(sa, sb, pa, pb, c0, fa, fb, c0, X, c0, no, X, 0);   // (L, a):   QK^T (L, b), P.V (L-1, b)
    } else {
      half(sb, sa, pb, pa, c1, fa, fa, c0, X, c0, no, X, 0);   // (L-1, b): QK^T (L, a), P.V (L-1, a)
      half(sa, sb, pa, pb, c0, fa, fa, c2, X, c1, no, X, 0);   // (L, a):   QK^T (L, b), P.V (L-1, b); reads the V^T fragments of sub-tile L
    }
    const f16x8 (&f)[8] = EARLY ? fb : fa;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA may outlive the wave
    const float psum = exps(sb, pb);
    if constexpr (!LMFMA) {
      sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));
      l_run[1] += psum;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        lacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, as_f16x8(pa[j]), lacc[0], 0, 0, 0);
        lacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, as_f16x8(pb[j]), lacc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[0][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * n + 1], as_f16x8(pa[n >> 1]), oacc[0][n & 1], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[1][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * n + 1], as_f16x8(pb[n >> 1]), oacc[1][n & 1], 0, 0, 0);
  }

  // ---- the guess failed somewhere in this block (rare): redo its rows with the max-tracking loop ---------------------------------------------
  if constexpr (LMFMA) {  // every row of l^T holds the column's sum: an exponential that left f16's range made it inf (or NaN)
    sticky |= __builtin_amdgcn_ballot_w64(!(lacc[0][0] < 3.0e38f) || !(lacc[1][0] < 3.0e38f));
  }
  bool redone = false;
  __syncthreads();  // every wave is done with the ring
  if (lane == 0) reinterpret_cast<int*>(smem)[wv] = sticky != 0;
  __syncthreads();
  const int4 flags = *reinterpret_cast<const int4*>(smem);
  if (__builtin_amdgcn_readfirstlane(flags.x | flags.y | flags.z | flags.w)) {
    __syncthreads();  // the flags have been read
    redone = true;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      m_run[x] = 0.0f; l_run[x] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[x][0][r] = oacc[x][1][r] = negm[x][r] = 0.0f;
    }
    for (int tt = 0; tt < ntiles; ++tt) {
      dma_krows(tt * KT, smem);
      dma_piece(tt, smem, 2);
      dma_piece(tt, smem, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x16 s;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(smem, u, ks), qf[x][ks], ks == 0 ? negm[x] : s, 0, 0, 0);
          float mx = pair_max(rowmax16(s));  // relative to the current reference
          const float delta = (tt == 0 && u == 0) ? mx : fmaxf(mx, 0.0f);  // the reference only grows, except on the first sub-tile, which sets it
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[x] += delta;
          l_run[x] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            oacc[x][0][r] *= alpha;
            oacc[x][1][r] *= alpha;
            negm[x][r] -= delta;
            s[r] -= delta;
          }
          u32x4 pc[2];
          l_run[x] += exps(s, pc);
#pragma unroll
          for (int n = 0; n < 4; ++n) oacc[x][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(smem, u, 4 + n), as_f16x8(pc[n >> 1]), oacc[x][n & 1], 0, 0, 0);
        }
      __syncthreads();
    }
  }

  if constexpr ((ABL & 4096) != 0) {  // (probe) shader cycles and 100 MHz ticks this block took, in the first floats of its lse rows
    const unsigned long long tk1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    if (p.lse && tid == 0) {
      float* dbg = p.lse + ((long)b * p.heads + h) * p.Nq + q0;
      dbg[128] = (float)(tk1 - tk0);
      dbg[129] = (float)(tr1 - tr0);
    }
  }
  // ---- finalize: O[q][d] = O^T[d][q] / l -------------------------------------------------------------------------------------------
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int qrow = q0 + wv * 64 + x * 32 + l31;
    const float l_tot = (LMFMA && !redone) ? lacc[x][0] : pair_sum(l_run[x]);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    if (p.lse && hi == 0 && qrow < p.Nq && !(ABL & 4096))
      p.lse[((long)b * p.heads + h) * p.Nq + qrow] = l_tot > 0.0f ? m_run[x] + __builtin_amdgcn_logf(l_tot) : INFINITY;
    if (qrow < p.Nq) {
      f16* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_rs + (long)h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f16x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (f16)(oacc[x][dt][4 * g + i] * inv);
          *reinterpret_cast<f16x4*>(op + dt * 32 + 8 * g + 4 * hi) = v;
        }
    }
  }
}

#undef GN_FENCE

}  // namespace

void gn_launch_attention_pwg(const AttnParams& p, int B, hipStream_t stream) {
  dim3 grid(((p.Nq + 255) / 256) * p.heads * B);
#ifdef GN_PWG_ABLATIONS
  const char* e = getenv("GN_PWG_ABL");
  switch (e ? atoi(e) : 0) {
#define ABL_CASE(n) case n: hipLaunchKernelGGL(attn_fwd_pwg_kernel<n>, grid, dim3(256), 0, stream, p); return;
    ABL_CASE(4096) ABL_CASE(4096 + 8) ABL_CASE(4096 + 3) ABL_CASE(4096 + 31) ABL_CASE(2048) ABL_CASE(2304) ABL_CASE(768) ABL_CASE(1024) ABL_CASE(1792) ABL_CASE(1280) ABL_CASE(256) ABL_CASE(259) ABL_CASE(260) ABL_CASE(272) ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(28) ABL_CASE(31) ABL_CASE(96) ABL_CASE(99) ABL_CASE(124) ABL_CASE(32) ABL_CASE(64)
#undef ABL_CASE
    default: break;
  }
#endif
  hipLaunchKernelGGL(attn_fwd_pwg_kernel<0>, grid, dim3(256), 0, stream, p);
}
