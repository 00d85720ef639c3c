// Flash-style attention forward for head dim 64 on gfx950, the LARGE self-attention problems of the U-Net / ControlNet (the 64 x 64 latent
// level: 4 096 keys): ONE WAVE PER SIMD, 64 query rows per wave, the whole register file per wave.
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
//
// SPLIT blocks (load balance): a 256-row block holds a CU for ~62 us at 4 096 keys, and B x heads x Nq / 256 blocks rarely fill whole rounds of
// 256 CUs (8 x 5 x 4096: 640 = 2.5 rounds).  When the remainder r of the block count is <= 128, the last heads' r blocks run as 2 r SPLIT blocks:
// 128 query rows, the two waves of a row block take one HALF of the keys each (their own ring, same reference m, so the partial
// (O, l) of the halves simply add -- through LDS, at the end).  A split block streams both key halves (twice the LDS-DMA pieces per
// MFMA) for half as long: the last round takes about half a round.  Small grids (r = the whole problem, e.g. B = 1) run all-split.
//
// What the measurements say (profiles/r05_v12_*; tools/probes/attn_pwg_abl.py / attn_pwg_clock.py / mfma_read_price.hip / lds_read_pattern.hip):
// 50 shader cycles per MFMA at 1.72 - 1.75 GHz in the loop (8 x 10 x 4096^2: 1.04 - 1.10 PF/s against 0.96 - 1.04 for the 32-row kernel on the
// same boxes); without the fragment reads 35.6 cycles -- a ds_read_b128 beside the MFMAs costs ~29 cycles of issue however it is placed (behind
// the consuming MFMA or a pair stage ahead into a second register set; bank-conflict free by the LDS pattern probe) -- without the softmax
// VALU 42.8.  Tried and dropped: fragment reads a pair stage EARLY (neutral, 48 more registers), one lgkmcnt(0) per pair stage instead
// of counted waits (+1.5 %), v_dot2c_f32_f16 row sums (+4 %), the row sums as two extra MFMAs per unit against a ones fragment (25 % more
// matrix work for 17 fewer VALU per half-stage: +9 %).  A random-operand MFMA stream alone sustains 1.70 PF/s on this part (19.75 ns per
// MFMA and SIMD: the power-limited clock), which is the ceiling the fractions of 2.5 PF are quoted against.
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
__device__ __forceinline__ float add_f32(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

constexpr int PWG_NS = 3;                                   // ring slots
constexpr int PWG_K_BYTES = KT * 128, PWG_BUF = 2 * PWG_K_BYTES, PWG_RING = PWG_NS * PWG_BUF;
constexpr int PWG_LDS = 2 * PWG_RING;                       // a split block runs two rings (one per key half)

// ABL: timing-only ablations (wrong results on purpose; tools/probes/attn_pwg_abl.py): 1 no v_exp, 2 no conversions / row sums, 4 no LDS-DMA
// in the loop, 8 no fragment reads in the loop, 16 no barrier in the loop, 32 no P.V MFMAs, 64 no QK^T MFMAs; 4096: the block's shader
// cycles and 100 MHz ticks go to its lse rows (tools/probes/attn_pwg_clock.py)
template <int ABL, bool SPLIT>
__device__ __forceinline__ void pwg_block(const AttnParams& p, unsigned char* const smem, const int b, const int h, const int q0) {
  constexpr int NW = 4, NS = PWG_NS, K_BYTES = PWG_K_BYTES, BUF = PWG_BUF, RING = PWG_RING;
  constexpr int NH = SPLIT ? 2 : 1;  // key halves the block streams

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  unsigned long long tk0 = 0, tr0 = 0;
  if constexpr ((ABL & 4096) != 0) { tk0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int rb = SPLIT ? (wv >> 1) : wv;   // the wave's 64-row block
  const int kh = SPLIT ? (wv & 1) : 0;     // the wave's key half
  const int rowbase = q0 + rb * 64;

  const f16* qp = p.q + (long)b * p.q_bs + (long)h * 64;
  const f16* kp = p.k + (long)b * p.k_bs + (long)h * 64;
  const f16* vp = p.vt + (long)b * p.vt_bs + (long)h * 64 * p.vt_rs;

  f16x8 qf[2][4];  // (c Q)^T fragments of the two query blocks: lane holds Q[qrow][16 ks + 8 hi .. +8] * scale * log2(e)
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int qrow = rowbase + x * 32 + l31;
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
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[0][0][r] = oacc[0][1][r] = oacc[1][0][r] = oacc[1][1][r] = negm[0][r] = negm[1][r] = 0.0f;
  const int ntiles = p.Nk / KT;                 // the launcher guarantees Nk % 64 == 0, Nk >= 128, not causal (split: ntiles even)
  const int ntl = SPLIT ? ntiles / 2 : ntiles;  // tiles of a key half

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
  // piece i of the image X(tile) = {K keys 64 tile + 32 .. + 64, V^T keys 64 tile .. + 64} (tile: global index): i < 2 K rows, else V^T
  // rows.  Unconditional: rows past the end read zeros (beyond the descriptor) or a neighbouring row's bytes (V^T) into LDS bytes nobody consumes.
  auto dma_piece = [&](int tile, unsigned char* X, int i) {
    if (i < 2) {
      const unsigned adv = (unsigned)(tile * KT + 32) * (unsigned)(p.k_rs * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(X + (wv + NW * i) * 1024), 16, koff[i] + adv, 0, 0, 0);
    } else {
      const unsigned adv = (unsigned)tile * (unsigned)(KT * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (attn_lds_ptr_t)(X + K_BYTES + (wv + NW * (i - 2)) * 1024), 16, voff[i - 2] + adv, 0, 0, 0);
    }
  };

  // fragment i of the 32-key sub-tile u of the tile image at X: i < 4 K rows (k16 step i), i >= 4 V^T (d tile (i - 4) & 1, k16 step (i - 4) >> 1).
  // `ring`: the per-lane offsets carry this wave's ring (kh * RING), so the image addresses stay compile-time immediates.
  struct Offs { int k[4], v[2][2]; };
  auto make_offs = [&](int base) {
    Offs o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.k[i] = base + lds_swz<128>(l31, i * 2 + hi);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) o.v[u][s] = base + K_BYTES + lds_swz<128>(l31, u * 4 + s * 2 + hi);
    return o;
  };
  const Offs ring = make_offs(kh * RING);
  auto frag = [&](const Offs& o, const unsigned char* X, int u, int i) -> f16x8 {
    if (i < 4) return *reinterpret_cast<const f16x8*>(X + o.k[i] + u * 4096);
    const int n = i - 4;
    return *reinterpret_cast<const f16x8*>(X + o.v[u][n >> 1] + (n & 1) * 4096);
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

  // One half-stage = the unit whose softmax runs (query block XS): exponents sc -> P fragments pc; QK^T of the NEXT unit (the other query
  // block) into sn; P.V of the PREVIOUS unit (the other query block) from pp.  f[]: the 8 fragments of this pair stage.  READ: this is
  // the pair stage's second half -- behind MFMA i, f[i] is re-read for the NEXT pair stage from image Xn, sub-tile UN (2: the V^T
  // fragments only, for the drain).  DMA: the LDS-DMA pieces of X(dma_tile) (of every key half) into Xd ride in the gaps.
  // Units of the softmax per pair k of scores: E(k) two v_exp_f32, C(k) one v_cvt_pk_f16_f32, S(k) the row-sum adds -- each a group or
  // two behind its producer.
  auto half = [&](const f32x16& sc, f32x16& sn, u32x4 (&pc)[2], const u32x4 (&pp)[2], auto xs_c, f16x8 (&f)[8], auto read_c, const unsigned char* Xn,
                  auto un_c, auto dma_c, unsigned char* Xd, int dma_tile) __attribute__((always_inline)) {
    constexpr int XS = decltype(xs_c)::value;
    constexpr int XN = 1 - XS, XP = 1 - XS;
    constexpr int READ = decltype(read_c)::value;  // 0: none, 1: all eight, 2: the V^T fragments only
    constexpr int UN = decltype(un_c)::value;
    constexpr bool DMA = decltype(dma_c)::value;
    float ex[16], ps0, ps1;
    auto D = [&](int i) {  // pieces 0 .. 3: this block's first (only) key half, 4 .. 7: a split block's second
      if constexpr (DMA && !(ABL & 4)) {
        if (i < 4) dma_piece(dma_tile, Xd, i);
        else if constexpr (SPLIT) dma_piece(ntl + dma_tile, Xd + RING, i - 4);
      }
    };
    f16x8 dummy[8];
    auto R = [&](int i) {
      if constexpr ((ABL & 8) != 0) return;
      if constexpr ((ABL & 128) != 0) {  // (probe) the reads are issued but no MFMA consumes them
        if constexpr (READ != 0) dummy[i] = frag(ring, Xn, UN, FI(i));
        return;
      }
      if constexpr (READ == 1) f[i] = frag(ring, Xn, UN, FI(i));
      else if constexpr (READ == 2) { if (i & 1) f[i] = frag(ring, Xn, UN, FI(i)); }
    };
    auto E = [&](int k) {
      if constexpr ((ABL & 1) != 0) { ex[2 * k] = sc[2 * k]; ex[2 * k + 1] = sc[2 * k + 1]; return; }
      ex[2 * k] = __builtin_amdgcn_exp2f(sc[2 * k]);
      ex[2 * k + 1] = __builtin_amdgcn_exp2f(sc[2 * k + 1]);
    };
    auto C = [&](int k) {
      if constexpr ((ABL & 2) != 0) { if ((k & 3) == 0) pc[k >> 2] = __builtin_bit_cast(u32x4, f32x4{ex[2 * k], ex[2 * k + 1], ex[2 * k + 2], ex[2 * k + 3]}); return; }
      pc[k >> 2][k & 3] = cvt_pk(ex[2 * k], ex[2 * k + 1]);
    };
    auto S = [&](int k) {
      if (k == 0 || (ABL & 2)) { ps0 = ex[0]; ps1 = ex[1]; }
      else { ps0 = add_f32(ps0, ex[2 * k]); ps1 = add_f32(ps1, ex[2 * k + 1]); }
    };
    auto M = [&](int i) {
      const int n = i >> 1;
      if ((i & 1) == 0) { if constexpr (!(ABL & 64)) sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], qf[XN][n], n == 0 ? negm[XN] : sn, 0, 0, 0); }
      else { if constexpr (!(ABL & 32)) oacc[XP][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], as_f16x8(pp[n >> 1]), oacc[XP][n & 1], 0, 0, 0); }
    };
    GN_FENCE; M(0); GN_FENCE; R(0); D(0); E(0);
    GN_FENCE; M(1); GN_FENCE; R(1); D(1); E(1); C(0);
    GN_FENCE; M(2); GN_FENCE; R(2); D(2); E(2); C(1); S(0);
    GN_FENCE; M(3); GN_FENCE; R(3); D(3); E(3); C(2); S(1);
    GN_FENCE; M(4); GN_FENCE; R(4); D(4); E(4); C(3); S(2);
    GN_FENCE; M(5); GN_FENCE; R(5); D(5); E(5); C(4); S(3);
    GN_FENCE; M(6); GN_FENCE; R(6); D(6); E(6); C(5); S(4);
    GN_FENCE; M(7); GN_FENCE; R(7); D(7); E(7); C(6); S(5);
    GN_FENCE; S(6); C(7); S(7);
    GN_FENCE;
    if constexpr ((ABL & 128) != 0 && READ != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(dummy[i]));
    }
    const float psum = ps0 + ps1;
    sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));  // v_cmp + s_or: no branch
    l_run[XS] += psum;
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  const std::true_type yes{};
  const std::false_type no{};

  // ---- the reference: row maxima over the wave's own diagonal keys (the K tile at its rows, clamped) ---------------------------------------
  // every wave stages its two pieces of all four reference tiles (4 x 8 KB at the bottom of the LDS; the waves of a split row block share theirs)
  const Offs plain = make_offs(0);
#pragma unroll
  for (int w = 0; w < NW; ++w) dma_krows(min((q0 + (SPLIT ? (w >> 1) : w) * 64) / KT, ntiles - 1) * KT, smem + w * K_BYTES);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 sa, sb;
  {
    const unsigned char* Kr = smem + wv * K_BYTES;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(plain, Kr, 0, ks), qf[x][ks], ks == 0 ? negm[x] : sa, 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(plain, Kr, 1, ks), qf[x][ks], ks == 0 ? negm[x] : sb, 0, 0, 0);
      m_run[x] = pair_max(fmaxf(rowmax16(sa), rowmax16(sb)));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[0][r] = -m_run[0]; negm[1][r] = -m_run[1]; }
  }
  __syncthreads();  // the LDS is free

  // ---- prologue (per key half): X(0) -> slot 0, X(1) -> slot 1 (stays in flight); the half's first 64 K rows -> slot 2: exponents of the
  // units (0,a), (0,b); P of (0,a) ------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(hf * ntl, smem + hf * RING, i);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) dma_krows(hf * ntl * KT, smem + hf * RING + 2 * BUF);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(hf * ntl + 1, smem + hf * RING + BUF, i);
  if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  u32x4 pa[2], pb[2];
  f16x8 f[8];
  {
    const unsigned char* K0 = smem + 2 * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(ring, K0, 0, ks), qf[0][ks], ks == 0 ? negm[0] : sa, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(ring, K0, 0, ks), qf[1][ks], ks == 0 ? negm[1] : sb, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = frag(ring, smem, 0, FI(i));  // F(0)
    const float psum = exps(sa, pa);
    sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));
    l_run[0] += psum;
  }
  __syncthreads();  // slot 2 is free

  // ---- steady state: iteration t = pair stages 2t, 2t + 1 on X(t) (ring slot t % 3); no branch inside ------------------------------------
  // (the slot index is a compile-time constant: LDS addresses fold into immediates; must inline, or the captures go through scratch)
  auto iteration = [&](auto cur_c, int t) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value, nxt = (cur + 1) % NS, fill = (cur + 2) % NS;
    unsigned char* X = smem + cur * BUF;
    half(sb, sa, pb, pa, c1, f, c0, X, c0, no, X, 0);                         // (2t, b):     QK^T (2t+1, a), P.V (2t, a)
    half(sa, sb, pa, pb, c0, f, c1, X, c1, no, X, 0);                         // (2t+1, a):   QK^T (2t+1, b), P.V (2t, b); reads F(2t+1)
    if constexpr (!(ABL & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's pieces of X(t+1), issued an iteration ago
      __syncthreads();                                                        // X(t+1) is published, X(t-1) retired
    }
    half(sb, sa, pb, pa, c1, f, c0, X, c0, yes, smem + fill * BUF, t + 2);    // (2t+1, b):   QK^T (2t+2, a), P.V (2t+1, a); fetches X(t+2)
    half(sa, sb, pa, pb, c0, f, c1, smem + nxt * BUF, c0, no, X, 0);          // (2t+2, a):   QK^T (2t+2, b), P.V (2t+1, b); reads F(2t+2)
  };
  const int nit = ntl - 1;
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
    half(sb, sa, pb, pa, c1, f, c0, X, c0, no, X, 0);   // (L-1, b): QK^T (L, a), P.V (L-1, a)
    half(sa, sb, pa, pb, c0, f, c2, X, c1, no, X, 0);   // (L, a):   QK^T (L, b), P.V (L-1, b); reads the V^T fragments of sub-tile L
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA may outlive the wave
    const float psum = exps(sb, pb);
    sticky |= __builtin_amdgcn_ballot_w64(!(psum <= PLIM));
    l_run[1] += psum;
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[0][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * n + 1], as_f16x8(pa[n >> 1]), oacc[0][n & 1], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < 4; ++n) oacc[1][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 * n + 1], as_f16x8(pb[n >> 1]), oacc[1][n & 1], 0, 0, 0);
  }

  // ---- flags; a split block's second-half waves hand their partial (O, l) over ------------------------------------------------------------------
  constexpr int PART = 1024, PART_WAVE = 66 * 256;  // 64 accumulator registers + 2 row sums, one 256-byte line per register
  __syncthreads();  // every wave is done with the rings
  if (lane == 0) reinterpret_cast<int*>(smem)[wv] = sticky != 0;
  if constexpr (SPLIT) {
    if (kh == 1) {
      float* part = reinterpret_cast<float*>(smem + PART + rb * PART_WAVE) + lane;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[((x * 2 + dt) * 16 + r) * 64] = oacc[x][dt][r];
        part[(64 + x) * 64] = l_run[x];
      }
    }
  }
  __syncthreads();
  const int4 flags = *reinterpret_cast<const int4*>(smem);
  if (__builtin_amdgcn_readfirstlane(flags.x | flags.y | flags.z | flags.w)) {
    // ---- the guess failed somewhere in this block (rare): redo its rows with the max-tracking loop over ALL keys (in a split block both
    // waves of a row block compute the same rows; the first one stores) ------------------------------------------------------------------------
    __syncthreads();  // the flags (and partials) have been read
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
          for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(plain, smem, u, ks), qf[x][ks], ks == 0 ? negm[x] : s, 0, 0, 0);
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
          for (int n = 0; n < 4; ++n) oacc[x][n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(plain, smem, u, 4 + n), as_f16x8(pc[n >> 1]), oacc[x][n & 1], 0, 0, 0);
        }
      __syncthreads();
    }
  } else if constexpr (SPLIT) {
    if (kh == 0) {  // same reference m in both halves: the partial sums add
      const float* part = reinterpret_cast<const float*>(smem + PART + rb * PART_WAVE) + lane;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[x][dt][r] += part[((x * 2 + dt) * 16 + r) * 64];
        l_run[x] += part[(64 + x) * 64];
      }
    }
  }

  if constexpr ((ABL & 4096) != 0) {  // (probe) shader cycles and 100 MHz ticks this block took, in two floats of its lse rows
    const unsigned long long tk1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    if (p.lse && tid == 0) {
      float* dbg = p.lse + ((long)b * p.heads + h) * p.Nq + q0;
      dbg[0] = (float)(tk1 - tk0);
      dbg[1] = (float)(tr1 - tr0);
    }
  }
  // ---- finalize: O[q][d] = O^T[d][q] / l -------------------------------------------------------------------------------------------
  if (SPLIT && kh != 0) return;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int qrow = rowbase + x * 32 + l31;
    const float l_tot = pair_sum(l_run[x]);
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

// blocks 0 .. nfull - 1: 256 query rows each, over heads 0 .. hfull - 1 of every batch element; blocks nfull .. nfull + nsplit - 1: split blocks
// of 128 rows over the remaining heads.  (Which rows run split depends on the HEAD only, never on the position in the batch: the two block
// kinds round differently, and permuting the episodes of a call must permute its output bit for bit.)
// XCD-aware order inside each range (as attention_stream.hip): consecutive block ids go round-robin over the 8 XCDs, so the blocks that
// share one (batch, head)'s K / V^T get ids that land on ONE XCD, next to each other in dispatch order.
template <int ABL>
__global__ __launch_bounds__(256, 1) void attn_fwd_pwg_kernel(const AttnParams p, const int nfull, const int nsplit, const int hfull) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[PWG_LDS];  // ONE LDS object (a second one makes hipcc drain vmcnt around the DMAs)
  const int nqb = (p.Nq + 255) / 256;
  const int bid = blockIdx.x;
  if (bid < nfull) {
    const int slot = (nfull % 8 == 0) ? (bid % 8) * (nfull / 8) + bid / 8 : bid;
    const int bh = __builtin_amdgcn_readfirstlane(slot / nqb);
    const int b = __builtin_amdgcn_readfirstlane(bh / hfull);
    pwg_block<ABL, false>(p, smem, b, bh - b * hfull, (slot - bh * nqb) * 256);
  } else {
    const int j = bid - nfull, hsplit = p.heads - hfull;
    const int unit = (nsplit % 8 == 0) ? (j % 8) * (nsplit / 8) + j / 8 : j;
    const int bh = __builtin_amdgcn_readfirstlane(unit / (2 * nqb));
    const int b = __builtin_amdgcn_readfirstlane(bh / hsplit);
    pwg_block<ABL, true>(p, smem, b, hfull + bh - b * hsplit, (unit - bh * 2 * nqb) * 128);
  }
}

#undef GN_FENCE

// GN_ATTN_PWG_SPLIT=0: no split blocks (every block 256 rows)
bool split_enabled() {
  static const bool on = getenv("GN_ATTN_PWG_SPLIT") ? atoi(getenv("GN_ATTN_PWG_SPLIT")) != 0 : true;
  return on;
}

}  // namespace

void gn_launch_attention_pwg(const AttnParams& p, int B, hipStream_t stream) {
  const long nqb = (p.Nq + 255) / 256, total = nqb * p.heads * B;
  const int ntiles = p.Nk / KT;
  const int r = (int)(total % 256);
  // the remainder of the last round as split blocks, in whole heads: needs whole 128-row units, an even tile count and key halves long
  // enough to pipeline
  const bool split = split_enabled() && r > 0 && r <= 128 && p.Nq % 256 == 0 && ntiles % 2 == 0 && ntiles >= 8;
  const int hsplit = split ? (int)(r / (B * nqb)) : 0, hfull = p.heads - hsplit;
  const int nsplit = (int)(2 * hsplit * B * nqb), nfull = (int)(hfull * B * nqb);
  dim3 grid(nfull + nsplit);
#ifdef GN_PWG_ABLATIONS
  const char* e = getenv("GN_PWG_ABL");
  switch (e ? atoi(e) : 0) {
#define ABL_CASE(n) case n: hipLaunchKernelGGL(attn_fwd_pwg_kernel<n>, grid, dim3(256), 0, stream, p, nfull, nsplit, hfull); return;
    ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(28) ABL_CASE(31) ABL_CASE(32) ABL_CASE(64) ABL_CASE(96) ABL_CASE(99) ABL_CASE(124)
    ABL_CASE(4096) ABL_CASE(4096 + 8) ABL_CASE(4096 + 3) ABL_CASE(4096 + 128) ABL_CASE(128)
#undef ABL_CASE
    default: break;
  }
#endif
  hipLaunchKernelGGL(attn_fwd_pwg_kernel<0>, grid, dim3(256), 0, stream, p, nfull, nsplit, hfull);
}
