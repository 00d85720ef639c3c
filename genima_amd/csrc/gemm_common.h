// Shared pieces of the MFMA GEMM kernels (gemm.hip, gemm_pp.hip): kernel parameter block, activations and the fused
// epilogues (bias / time shift / residual / activation / GEGLU / f32 + accumulate / transposed / split-K slab).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gn_bridge.h"

namespace {

constexpr int BK = 64;  // K tile (f16 elements) = 128-byte LDS rows

struct GemmParams {
  const f16* a;
  const f16* a2;
  const f16* w;
  const f16* bias;
  const f16* shift;
  const f16* res;
  f16* out;
  float* ws;
  int M, N, K;
  long lda, ldw, ldr, ldo, ldshift;
  int H, W, C1, C2, KH, KW, stride, pad_t, pad_l, Ho, Wo, ups;
  int act, out_mode, rpb, res_first;
  float out_scale;
  int splitk, kper;
  int tiles_m, tiles_n;
  unsigned a_bytes, a2_bytes, w_bytes;  // LDS-DMA variant: buffer-descriptor extents (everything else reads as zero)
  int accumulate;                        // GN_OUT_F32: out += result
  int nbatch, binner;                    // batched GEMM: blockIdx.z in [0, nbatch) = outer * binner + inner
  long a_bs, a_bs2, w_bs, w_bs2, o_bs, o_bs2, r_bs, r_bs2;  // batch strides (elements; o_* in output elements)
  const float* sa;                       // fp8: per-row scales of A [M]
  const float* sw;                       // fp8: per-row scales of W [N]
  f16* out2;                             // two-destination output: columns >= split_n go here, batch-transposed (row stride ldo2)
  long ldo2;
  int split_n;
  const float* ln_c1;                    // LayerNorm folded into the GEMM (ln_fold_apply below): per-column sums of the gamma-scaled weight
  float ln_eps;
  int cm_tiles;                          // 1: column-major tile order (the weight is the big operand)
  int up_ph;                             // 1: blockIdx.z = output phase 2 dy + dx of an upsampling conv (batch_offset below)
  int orw;                               // row-major f16 output with a two-level row pitch: row m lives at (m / orw) * ldo_hi + (m % orw) * ldo
  long ldo_hi;                           // (one phase of a nearest-2x upsampling conv writes every other pixel of every other image row); 0 = m * ldo
  int kapp, kapp_k0;                     // conv with a 1x1 conv APPENDED along K (gn_gemm_desc.k_append): K tiles from kapp_k0 = KH * KW * C1 on read
                                         // the output pixel of the appended sources a2 (C2 channels), then a3 (C3) -- the LDS-DMA kernels' loaders
  const f16* a3; int C3; unsigned a3_bytes;
  long lda2;                             // dense k_append: out = [A | A2] . W^T -- K tiles from kapp_k0 on read A2 (row stride lda2) through a2
  struct NormOutP { f16* y; const f16* gamma; const f16* beta; float eps; int groups, act, rps; } nout;  // GroupNorm inside the split-K reduce (gemm.hip)
  GnSinkP sink;                          // GroupNorm bridge, producer side (gn_bridge.h): stats == nullptr = off
  GnInP gin;                             // GroupNorm bridge, consumer side (gemm_s3.hip's normalising A path): stats == nullptr = off
  // persistent skewed ping-pong kernel (gemm_ppp.hip, tile 25): the grid is ppG workgroups (one per CU) that walk the tile list; `ws` holds the f32
  // hand-off slabs of the tiles whose K range two (or more) workgroups share, ppflags their ready words (zero between launches: the consumer resets them)
  struct FastDiv { unsigned mul, shift; };  // n / d for 32-bit n as (umulhi(n, mul) + n) >> shift (Granlund - Montgomery; gn_ppp_plan fills them)
  FastDiv dG, dS, dTm, dTn, dTmn, dHw, dWo, dCin, dKW, dOrw;
  unsigned* ppflags;
  unsigned* pptmo;                       // counts bounded hand-off waits that gave up (gn_ppp_timeouts)
  int ppG, ppR, ppTail, ppS, ppNz, ppSkew;  // workgroups, full rounds, tiles of the last partial round, workgroups per such tile, blockIdx.z extent folded in, skew on
};

// GroupNorm bridge, producer side (gn_bridge.h).  Where the workgroup's LDS can hold its f16 output tile, the epilogue's 16-byte row stores
// are mirrored into that tile (SinkLds) and the statistics tail reads LDS -- no store drain, no memory round trip at the end of the small-M
// launches the tail would otherwise dominate; the largest tiles (256 x 256 / 256 x 320) re-read their output from L2 instead.
struct SinkLds { unsigned char* base = nullptr; int pitch = 0, m0 = 0, n0 = 0; };  // pitch in BYTES; base == nullptr: off
constexpr int kSinkScratch = 20 * 1024;  // the tail's reduction scratch behind the tile
constexpr int sink_pitch(int bn) { return (bn + 8) * 2; }
#ifdef GN_SINK_NO_LDS
constexpr bool sink_in_lds(int, int, int) { return false; }
#else
constexpr bool sink_in_lds(int bm, int bn, int smem_bytes) { return bm * sink_pitch(bn) + kSinkScratch <= smem_bytes; }
#endif

template <int BM, int BN, int SMEM>
__device__ __forceinline__ SinkLds gemm_sink_lds(const GemmParams& p, int m0, int n0, unsigned char* smem) {
  SinkLds sl;
  if constexpr (sink_in_lds(BM, BN, SMEM)) {
    if (p.sink.stats && p.splitk <= 1) { sl.base = smem; sl.pitch = sink_pitch(BN); sl.m0 = m0; sl.n0 = n0; }
  }
  return sl;
}

// the tail of a GEMM workgroup whose tile has gone through gemm_epilogue (row-major f16, no K split)
template <int NT, int BM, int BN, int SMEM>
__device__ __forceinline__ void gemm_sink_tail(const GemmParams& p, int m0, int n0, unsigned char* smem) {
  if (p.sink.stats && p.splitk <= 1) {  // workgroup-uniform
    const int mend = min(m0 + BM, p.M), nend = min(n0 + BN, p.N);
    if constexpr (sink_in_lds(BM, BN, SMEM)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();  // every wave's mirrored rows are in the LDS tile
      gn_sink_tile<NT, true>(p.sink, reinterpret_cast<const f16*>(smem), sink_pitch(BN) / 2, 0, 0, m0, mend, n0, nend, m0, n0,
                             reinterpret_cast<float*>(smem + BM * sink_pitch(BN)), (SMEM - BM * sink_pitch(BN)) / 4, (m0 / BM) % p.sink.reps);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's output stores are in L2
      __syncthreads();                                  // ... every wave's; and nobody reads the K loop's LDS tiles any more
      gn_sink_tile<NT, false>(p.sink, p.out, p.ldo, p.orw, p.ldo_hi, m0, mend, n0, nend, 0, 0, reinterpret_cast<float*>(smem), SMEM / 4,
                              (m0 / BM) % p.sink.reps);
    }
  }
}

// the appended 1x1 segment of a k_append conv: source offset of staged row i for the K tile whose lane offset inside the segment is `co`
// (iy0 / ix0 / pbase as the conv loaders keep them: the row's top-left tap coordinate and image base; rows >= M carry -(1 << 28))
// cs / co: channel count of the source this K tile lies in and the lane's channel inside it (kapp_src below)
__device__ __forceinline__ unsigned kapp_voff(const GemmParams& p, int iy0, int ix0, int pbase, int cs, int co) {
  const int cy = iy0 + p.pad_t, cx = ix0 + p.pad_l;  // stride 1: the output pixel itself
  return iy0 > -(1 << 27) ? (unsigned)(((long)(pbase + cy * p.W + cx) * cs + co) * 2) : 0xFFFFFFF0u;
}
// which appended source the K tile at (wave-uniform) origin kt0 reads: a2 for the first C2 channels of the segment, a3 behind it (C2 % 64 == 0
// when there are two).  -> true = a2
__device__ __forceinline__ bool kapp_src(const GemmParams& p, int kt0, int& cs, int& cbase) {
  const bool s2 = kt0 - p.kapp_k0 < p.C2;
  cs = s2 ? p.C2 : p.C3;
  cbase = s2 ? p.kapp_k0 : p.kapp_k0 + p.C2;
  return s2;
}

// element offset of output row m (row-major f16 outputs)
__device__ __forceinline__ long out_row_off(const GemmParams& p, int m) {
  if (p.orw) {
    const int hi = m / p.orw;
    return (long)hi * p.ldo_hi + (long)(m - hi * p.orw) * p.ldo;
  }
  return (long)m * p.ldo;
}

// batched GEMM: offset every operand of this workgroup's problem by its (outer, inner) batch strides
__device__ __forceinline__ GemmParams batch_offset(const GemmParams& pin) {
  GemmParams p = pin;
  if (pin.up_ph) {
    // the four phase convs of an Upsample2D as ONE launch: phase z = 2 dy + dx has its own 2x2 weights (w + z * w_bs), top / left padding
    // 1 - dy / 1 - dx and writes pixels (2y + dy, 2x + dx): half an ldo_hi down, half an ldo to the right of phase (0, 0)
    const int bz = blockIdx.z, dy = bz >> 1, dx = bz & 1;
    p.w += (long)bz * pin.w_bs;
    p.pad_t = 1 - dy;
    p.pad_l = 1 - dx;
    p.out = pin.out + (long)dy * (pin.ldo_hi >> 1) + (long)dx * (pin.ldo >> 1);
    return p;
  }
  if (pin.binner > 0) {
    const int bz = blockIdx.z;
    const long bo = bz / pin.binner, bi = bz - bo * pin.binner;
    p.a += bo * pin.a_bs + bi * pin.a_bs2;
    p.w += bo * pin.w_bs + bi * pin.w_bs2;
    if (pin.res) p.res += bo * pin.r_bs + bi * pin.r_bs2;
    const long oo = bo * pin.o_bs + bi * pin.o_bs2;
    p.out = pin.out_mode == GN_OUT_F32 ? reinterpret_cast<f16*>(reinterpret_cast<float*>(pin.out) + oo) : pin.out + oo;
  }
  return p;
}

// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below f16 resolution): ~12 VALU instead of ocml erff's ~50, which
// matters because the GEGLU / GELU epilogues run on K = 320..1280 GEMMs where the epilogue is a visible share of the tile time.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gemm_act(float x, int act) {
  switch (act) {
    case GN_ACT_SILU: return act_silu(x);
    case GN_ACT_GELU: return gelu_fast(x);
    case GN_ACT_QUICK_GELU: return act_quick_gelu(x);
    case GN_ACT_RELU: return fmaxf(x, 0.0f);
    default: return x;
  }
}

// ---- epilogue for 4 consecutive output channels [nb, nb+4) of row m -------------------------------------------------
// bias / shift / residual / activation / scale of 4 consecutive output channels; bidx = batch index of row m (when needed)
__device__ __forceinline__ void epilogue_vals4(const GemmParams& p, int m, int nb, float (&v)[4], int& bidx) {
  if (p.bias) {
    f16x4 b = *reinterpret_cast<const f16x4*>(p.bias + nb);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)b[i];
  }
  bidx = 0;
  if (p.shift || p.out_mode == GN_OUT_BATCH_TRANSPOSED) bidx = m / p.rpb;
  if (p.shift) {
    f16x4 s = *reinterpret_cast<const f16x4*>(p.shift + (long)bidx * p.ldshift + nb);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)s[i];
  }
  if (p.res && p.res_first) {  // ResNet basic block: act(conv + identity)
    f16x4 r = *reinterpret_cast<const f16x4*>(p.res + (long)m * p.ldr + nb);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)r[i];
  }
  if (p.act != GN_ACT_NONE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gemm_act(v[i], p.act);
  }
  if (p.out_scale != 1.0f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= p.out_scale;
  }
  if (p.res && !p.res_first) {
    f16x4 r = *reinterpret_cast<const f16x4*>(p.res + (long)m * p.ldr + nb);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)r[i];
  }
}

__device__ __forceinline__ void epilogue_store4(const GemmParams& p, int m, int nb, float v0, float v1, float v2, float v3) {
  float v[4] = {v0, v1, v2, v3};
  int bidx;
  epilogue_vals4(p, m, nb, v, bidx);
  if (p.out_mode == GN_OUT_BATCH_TRANSPOSED) {
    const int ml = m - bidx * p.rpb;
    f16* o = p.out + ((long)bidx * p.N + nb) * p.ldo + ml;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[(long)i * p.ldo] = (f16)v[i];
  } else if (p.out_mode == GN_OUT_F32) {  // f32 result (weight gradients: fp32 like the reference's master grads)
    f32x4 o = {v[0], v[1], v[2], v[3]};
    float* op = reinterpret_cast<float*>(p.out) + (long)m * p.ldo + nb;
    if (p.accumulate) o += *reinterpret_cast<const f32x4*>(op);
    *reinterpret_cast<f32x4*>(op) = o;
  } else {
    f16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (f16)v[i];
    *reinterpret_cast<f16x4*>(p.out + out_row_off(p, m) + nb) = o;
  }
}

// ---- wave-level epilogue: the lane holds D[n = 8g + 4hi + (r&3)][m = lane&31] of each 32x32 tile ----------------------------
// The epilogue of a K = 320..1280 launch is a visible share of its time and it is LATENCY-bound when written store by store:
// a load that follows a store cannot be hoisted over it (may alias), and gfx9's single vmcnt retires loads and stores in order,
// so "load bias/residual -> wait -> store" per 8 channels costs one memory round trip each (tools/probes/gemm_ksweep.py: the
// K -> 0 intercept of a 128x64 tile launch was 8.8 us against a 4.3 us store-only kernel; 27 us for the 128x320 tile).  The f16
// row-major path therefore batches: every bias vector up front, then per 32-row band all shift / residual vectors of the NEXT band
// are requested before the current band's stores are issued.

// compile-time loop: indices are template constants whatever the unroller's size thresholds decide (a `#pragma unroll` loop whose
// unrolled body passes -pragma-unroll-threshold stays a loop, and ONE dynamic index sends the whole accumulator array to scratch)
template <int N>
struct static_for {
  template <class F>
  __device__ __forceinline__ static void run(F&& f) {
    static_for<N - 1>::run(f);
    f(std::integral_constant<int, N - 1>{});
  }
};
template <>
struct static_for<0> {
  template <class F>
  __device__ __forceinline__ static void run(F&&) {}
};

// One 32x32 tile at a time: its 16 values go through whole-tile passes (bias / shift / residual / activation / scale: one
// wave-uniform branch per pass instead of one per 4 channels -- the activation switch inside the store loop compiled to a thicket
// of ~10 branches per 4 channels), then its stores are issued back to back.
__device__ __forceinline__ void epilogue_tile_math(const GemmParams& p, float (&v)[16], const f16x4 (&b)[4], const f16x4 (&a)[4],
                                                   const f16x4 (&r)[4], bool add_a_pre, bool add_r_pre, bool add_a_post, bool add_r_post) {
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (float)b[e >> 2][e & 3];
  }
  if (add_a_pre) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (float)a[e >> 2][e & 3];
  }
  if (add_r_pre) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (float)r[e >> 2][e & 3];
  }
  switch (p.act) {
    case GN_ACT_SILU:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = act_silu(v[e]);
      break;
    case GN_ACT_GELU:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = gelu_fast(v[e]);
      break;
    case GN_ACT_QUICK_GELU:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = act_quick_gelu(v[e]);
      break;
    case GN_ACT_RELU:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.0f);
      break;
    default: break;
  }
  if (p.out_scale != 1.0f) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] *= p.out_scale;
  }
  if (add_a_post) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (float)a[e >> 2][e & 3];
  }
  if (add_r_post) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += (float)r[e >> 2][e & 3];
  }
}

// The shift / residual vectors of one 32-column tile for this lane: channel groups g = 0..3 at columns 8 g + 4 hi + (0..3).  Row-major
// rows whose 8-channel groups are 16-byte aligned are fetched as TWO 16-byte loads per lane instead of four 8-byte ones (the load path is
// issue-bound like the store path): lane l takes columns 8 g .. 8 g + 7, lane l + 32 columns 8 (g + 1) .. + 7, and one
// v_permlane32_swap per dword hands each lane its halves of both groups.
__device__ __forceinline__ void load_groups4(const f16* row, int col0, int hi, int N, bool wide, f16x4 (&a)[4]) {
  const f16x4 zero = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
  if (wide) {
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      const int col = col0 + 8 * g + 8 * hi;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row && col < N) v = *reinterpret_cast<const uint4*>(row + col);
      const auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
      const uint2 lo = make_uint2(r0[0], r1[0]), hi2 = make_uint2(r0[1], r1[1]);
      a[g] = *reinterpret_cast<const f16x4*>(&lo);
      a[g + 1] = *reinterpret_cast<const f16x4*>(&hi2);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = col0 + 8 * g + 4 * hi;
      a[g] = (row && nb < N) ? *reinterpret_cast<const f16x4*>(row + nb) : zero;
    }
  }
}

// Residual values of a whole wave tile, requested BEFORE the K loop: the epilogue's residual read is then a register move instead of a
// memory round trip at the end of the workgroup (with every residual / bias load removed the tiled b8 call is 5 ms shorter: most of
// that is latency the K loop can hide).  Costs TM x TN x 8 registers for the length of the loop; kernels opt in where they fit.
template <int TM, int TN>
struct EpiPre { f16x4 r[TM][TN][4]; };

template <int TM, int TN>
__device__ __forceinline__ bool epilogue_prefetch(const GemmParams& p, EpiPre<TM, TN>& pre, int mbase, int nbase, int l31, int hi) {
  if (!(p.res && !p.shift && p.splitk <= 1 && p.out_mode == GN_OUT_ROWMAJOR && p.act != GN_ACT_GEGLU)) return false;  // wave-uniform
  const bool wide = (p.ldr & 7) == 0 && (p.N & 7) == 0 && ((uintptr_t)p.res & 15) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + i * 32 + l31;
    const f16* row = m < p.M ? p.res + (long)m * p.ldr : nullptr;  // (lanes l and l + 32 share the row: the swap pairs agree)
#pragma unroll
    for (int j = 0; j < TN; ++j) load_groups4(row, nbase + j * 32, hi, p.N, wide, pre.r[i][j]);
  }
  return true;
}
// registers: accumulators + prefetched residuals + loop / epilogue working set against the occupancy's budget
constexpr bool epi_prefetch_fits(int tm, int tn, int budget) { return tm * tn <= 8 && tm * tn * 16 + tn * 8 + tm * tn * 8 + 80 <= budget; }
// the batched (RICH) epilogue: accumulators + every bias vector + one or two bands of shift / residual vectors + working set
constexpr bool epi_rich_fits(int tm, int tn, int budget) { return tm * tn * 16 + tn * 8 * (1 + (tm < 2 ? tm : 2)) + 96 <= budget; }

// RICH (registers to spare: small wave tiles, or one wave per SIMD): every bias vector up front and the shift / residual vectors of
// the NEXT 32-row band requested before the current band's stores.  Otherwise the vectors are fetched tile by tile.
template <int TM, int TN, bool RICH>
__device__ __forceinline__ void gemm_epilogue_direct(const GemmParams& p, const f32x16 (&acc)[TN][TM], int mbase, int nbase, int l31,
                                                     int hi, const EpiPre<TM, TN>& pre, bool use_pre, const SinkLds& sl) {
  constexpr bool PIPE = RICH && TM * TN <= 8;
  const bool pre_ok = RICH && use_pre;  // the residual of every band is already in registers (epilogue_prefetch)
  const bool has_shift = p.shift != nullptr, has_res = p.res != nullptr;
  const bool both = has_shift && has_res;  // rare: the residual is then read tile by tile
  const bool aux_is_res = has_res && !has_shift;
  const bool a_pre = has_shift || (aux_is_res && p.res_first), a_post = aux_is_res && !p.res_first;
  const bool r_pre = both && p.res_first, r_post = both && !p.res_first;
  const f16x4 zero = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
  f16x4 bv[RICH ? TN : 1][4];
  auto load_bias = [&](int j, f16x4 (&b)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = nbase + j * 32 + 8 * g + 4 * hi;
      b[g] = (p.bias && nb < p.N) ? *reinterpret_cast<const f16x4*>(p.bias + nb) : zero;
    }
  };
  if (RICH) {
#pragma unroll
    for (int j = 0; j < TN; ++j) load_bias(j, bv[RICH ? j : 0]);
  }
  f16x4 ax[PIPE ? 2 : 1][RICH ? TN : 1][4];
  const bool aux_wide = (p.N & 7) == 0 && (aux_is_res ? ((p.ldr & 7) == 0 && ((uintptr_t)p.res & 15) == 0)
                                                      : (has_shift && (p.ldshift & 7) == 0 && ((uintptr_t)p.shift & 15) == 0));
  auto load_aux1 = [&](int i, int j, f16x4 (&a)[4]) {
    const int m = mbase + i * 32 + l31;
    const f16* row = nullptr;
    if (m < p.M) row = aux_is_res ? p.res + (long)m * p.ldr : p.shift + (long)(m / p.rpb) * p.ldshift;
    load_groups4(row, nbase + j * 32, hi, p.N, aux_wide, a);
  };
  auto load_aux = [&](int i, f16x4 (&a)[RICH ? TN : 1][4]) {
    const int m = mbase + i * 32 + l31;
    const f16* row = nullptr;
    if (m < p.M) row = aux_is_res ? p.res + (long)m * p.ldr : p.shift + (long)(m / p.rpb) * p.ldshift;
#pragma unroll
    for (int j = 0; j < (RICH ? TN : 1); ++j) load_groups4(row, nbase + j * 32, hi, p.N, aux_wide, a[j]);
  };
  const bool wide = p.out_mode == GN_OUT_ROWMAJOR && (p.ldo & 7) == 0 && (p.N & 7) == 0 && ((uintptr_t)p.out & 15) == 0;
  if (PIPE && (has_shift || has_res) && !pre_ok) load_aux(0, ax[0]);
  static_for<TM>::run([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int kOne = PIPE ? 1 : 0;
    if (pre_ok) {
      if constexpr (RICH) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) ax[i & kOne][j][g] = pre.r[i][j][g];
      }
    } else if (has_shift || has_res) {
      if (PIPE) { if (i + 1 < TM) load_aux(i + 1, ax[(i + 1) & kOne]); }
      else if (RICH) load_aux(i, ax[0]);
    }
    const int m = mbase + i * 32 + l31;
    const bool mok = m < p.M;
    const int bidx = (p.out_mode == GN_OUT_BATCH_TRANSPOSED || p.out2) ? m / p.rpb : 0;
    static_for<TN>::run([&](auto J) {
      constexpr int j = decltype(J)::value;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[j][i][e];
      f16x4 rr[4] = {zero, zero, zero, zero};
      if (both && mok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          if (nb < p.N) rr[g] = *reinterpret_cast<const f16x4*>(p.res + (long)m * p.ldr + nb);
        }
      }
      if (!RICH) {
        load_bias(j, bv[0]);
        if (has_shift || has_res) load_aux1(i, j, ax[0][0]);
      }
      epilogue_tile_math(p, v, bv[RICH ? j : 0], ax[i & kOne][RICH ? j : 0], rr, a_pre, r_pre, a_post, r_post);
      if (!mok) return;
      if (p.out2 && nbase + j * 32 >= p.split_n) {  // wave-uniform: a 32-column tile never straddles split_n (split_n % 32 == 0)
        const int ml = m - bidx * p.rpb;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          if (nb < p.N) {
            f16* o = p.out2 + ((long)bidx * (p.N - p.split_n) + (nb - p.split_n)) * p.ldo2 + ml;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[(long)e * p.ldo2] = (f16)v[4 * g + e];
          }
        }
      } else if (wide) {
        // lanes l and l + 32 hold the same row: one v_permlane32_swap per dword trades group g of the upper half against group
        // g + 1 of the lower half; afterwards the lower lane owns channels 8g .. 8g+7 and the upper lane 8(g+1) .. 8(g+1)+7,
        // a 16-byte store each (the write path is issue-bound on 8-byte pieces)
        f16* orow = p.out + out_row_off(p, m);
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          f16x4 ha, hb;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ha[e] = (f16)v[4 * g + e]; hb[e] = (f16)v[4 * g + 4 + e]; }
          const uint2 ua = *reinterpret_cast<const uint2*>(&ha), ub = *reinterpret_cast<const uint2*>(&hb);
          const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
          const int col = nbase + j * 32 + 8 * g + 8 * hi;
          if (col < p.N) {
            const uint4 o4 = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            *reinterpret_cast<uint4*>(orow + col) = o4;
            if (sl.base) *reinterpret_cast<uint4*>(sl.base + (m - sl.m0) * sl.pitch + (col - sl.n0) * 2) = o4;  // (wave-uniform) GroupNorm bridge
          }
        }
      } else if (p.out_mode == GN_OUT_ROWMAJOR) {
        f16* orow = p.out + out_row_off(p, m);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          f16x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = (f16)v[4 * g + e];
          if (nb < p.N) *reinterpret_cast<f16x4*>(orow + nb) = h;
        }
      } else if (p.out_mode == GN_OUT_F32) {  // f32 result (weight gradients: fp32 like the reference's master grads)
        float* orow = reinterpret_cast<float*>(p.out) + (long)m * p.ldo;
        if (p.accumulate) {
          f32x4 old[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nb = nbase + j * 32 + 8 * g + 4 * hi;
            const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            old[g] = nb < p.N ? *reinterpret_cast<const f32x4*>(orow + nb) : z4;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += old[e >> 2][e & 3];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          if (nb < p.N) *reinterpret_cast<f32x4*>(orow + nb) = o;
        }
      } else {  // GN_OUT_BATCH_TRANSPOSED: out[b][n][m - b * rpb]
        const int ml = m - bidx * p.rpb;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          if (nb < p.N) {
            f16* o = p.out + ((long)bidx * p.N + nb) * p.ldo + ml;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[(long)e * p.ldo] = (f16)v[4 * g + e];
          }
        }
      }
    });
  });
}

template <int TM, int TN, bool RICH = (TM * TN <= 4)>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TN][TM], int mbase, int nbase, int l31, int hi,
                                              int z, const EpiPre<TM, TN>& pre, bool use_pre, const SinkLds& sl = SinkLds{}) {
  if (p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = mbase + i * 32 + l31;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          if (nb < p.N) {
            f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
            *reinterpret_cast<f32x4*>(p.ws + ((long)z * p.M + m) * p.N + nb) = v;
          }
        }
    }
  } else if (p.act == GN_ACT_GEGLU) {
    if constexpr (TN % 2 == 0) {
      const f16x4 zero = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
      f16x4 bh[TN / 2][4], bg[TN / 2][4];
#pragma unroll
      for (int j = 0; j < TN; j += 2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nh = nbase + j * 32 + 8 * g + 4 * hi;
          const bool ok = p.bias && nh + 32 < p.N;
          bh[j / 2][g] = ok ? *reinterpret_cast<const f16x4*>(p.bias + nh) : zero;
          bg[j / 2][g] = ok ? *reinterpret_cast<const f16x4*>(p.bias + nh + 32) : zero;
        }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = mbase + i * 32 + l31;
        if (m >= p.M) continue;
        const bool wide = (p.ldo & 7) == 0 && ((uintptr_t)p.out & 15) == 0 && (p.N & 127) == 0;  // whole hidden | gate blocks, 16-byte rows
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          f16x4 o[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              o[g][e] = (f16)((acc[j][i][4 * g + e] + (float)bh[j / 2][g][e]) * gelu_fast(acc[j + 1][i][4 * g + e] + (float)bg[j / 2][g][e]));
          const int ocb = (nbase + j * 32) >> 1;  // first output column of this hidden | gate pair of tiles
          if (wide) {
            // as in the row-major path: lanes l / l + 32 trade channel groups so that each owns 8 consecutive outputs (one 16-byte store
            // instead of two 8-byte ones; the K = 320 GEGLU launches are bound by the store path)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
              const uint2 ua = *reinterpret_cast<const uint2*>(&o[g]), ub = *reinterpret_cast<const uint2*>(&o[g + 1]);
              const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
              if (nbase + j * 32 + 32 < p.N)
                *reinterpret_cast<uint4*>(p.out + (long)m * p.ldo + ocb + 8 * g + 8 * hi) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nh = nbase + j * 32 + 8 * g + 4 * hi;
              if (nh + 32 < p.N) *reinterpret_cast<f16x4*>(p.out + (long)m * p.ldo + ocb + 8 * g + 4 * hi) = o[g];
            }
          }
        }
      }
    }
  } else {
    gemm_epilogue_direct<TM, TN, RICH>(p, acc, mbase, nbase, l31, hi, pre, use_pre, sl);
  }
}

template <int TM, int TN, bool RICH = (TM * TN <= 4)>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TN][TM], int mbase, int nbase, int l31, int hi, int z,
                                              const SinkLds& sl = SinkLds{}) {
  EpiPre<TM, TN> none;  // never read
  gemm_epilogue<TM, TN, RICH>(p, acc, mbase, nbase, l31, hi, z, none, false, sl);
}

// ---- LayerNorm folded into the consuming Linear (gn_gemm_desc::ln_c1; SURVEY.md K7: "LN -> QKV without a round trip") ---------------------
// With W' = W * gamma (column-wise, folded into the packed weight), c1[n] = sum_k W'[n, k] and c2[n] = sum_k W[n, k] * beta[k] + b[n]:
//   Linear(LayerNorm(x))[m, n] = rstd[m] * (sum_k x[m, k] W'[n, k] - mean[m] * c1[n]) + c2[n]
// so the GEMM runs on the RAW rows and needs only each row's mean / rstd -- which it can take from the A fragments it reads anyway (every
// workgroup walks the whole K range of its rows; no split-K in this mode).  The K steps are dealt round-robin to the WN waves that share a
// row band (each A fragment is summed by exactly one of them: 8 v_dot2_f32_f16 per fragment), the partial sums meet in LDS after the K loop.
// c1 is summed from the f16-rounded W' the MFMA multiplies, so the subtraction is exact algebra on the same numbers: the result is the GEMM
// of the exactly centred rows (no f16 rounding of the normalised activations at all -- closer to the fp32 reference than LN -> f16 -> GEMM).
template <int TM>
struct LnStats {
  float s1[TM], s2[TM];
};
template <int TM>
__device__ __forceinline__ void ln_stats_init(LnStats<TM>& st) {
#pragma unroll
  for (int i = 0; i < TM; ++i) st.s1[i] = st.s2[i] = 0.0f;
}
template <int TM>
__device__ __forceinline__ void ln_stats_step(LnStats<TM>& st, const f16x8 (&fa)[TM]) {
  const f16x2 ones = {(f16)1.0f, (f16)1.0f};
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const f16x2 h0 = __builtin_shufflevector(fa[i], fa[i], 0, 1), h1 = __builtin_shufflevector(fa[i], fa[i], 2, 3);
    const f16x2 h2 = __builtin_shufflevector(fa[i], fa[i], 4, 5), h3 = __builtin_shufflevector(fa[i], fa[i], 6, 7);
    st.s1[i] = __builtin_amdgcn_fdot2(h0, ones, st.s1[i], false);
    st.s2[i] = __builtin_amdgcn_fdot2(h0, h0, st.s2[i], false);
    st.s1[i] = __builtin_amdgcn_fdot2(h1, ones, st.s1[i], false);
    st.s2[i] = __builtin_amdgcn_fdot2(h1, h1, st.s2[i], false);
    st.s1[i] = __builtin_amdgcn_fdot2(h2, ones, st.s1[i], false);
    st.s2[i] = __builtin_amdgcn_fdot2(h2, h2, st.s2[i], false);
    st.s1[i] = __builtin_amdgcn_fdot2(h3, ones, st.s1[i], false);
    st.s2[i] = __builtin_amdgcn_fdot2(h3, h3, st.s2[i], false);
  }
}
// before the first K tile: the workgroup's BN values of c1 (zeros past N) -> `dst` in LDS (visible after the barrier that ends the first tile's wait)
template <int BN>
__device__ __forceinline__ void ln_c1_to_lds(const GemmParams& p, float* dst, int n0) {
  const int t = threadIdx.x;
  if (t < BN / 4) {
    const int n = n0 + 4 * t;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (n + 3 < p.N) v = *reinterpret_cast<const float4*>(p.ln_c1 + n);
    *reinterpret_cast<float4*>(dst + 4 * t) = v;
  }
}

// after the K loop (every wave of the workgroup calls it; `lds` = WN * BM * 2 floats no wave is reading any more):
// acc <- rstd * (acc - mean * c1[n]); the usual epilogue follows with c2 as the bias
template <int TM, int TN, int WN, int BM>
__device__ __forceinline__ void ln_fold_apply(const GemmParams& p, f32x16 (&acc)[TN][TM], LnStats<TM>& st, float* lds, int row0, int wn,
                                              const float* c1_lds, int l31, int hi) {
  // the c1 values of this wave's columns: the workgroup copied its BN of them to LDS before the first K tile (ln_c1_to_lds), so no global
  // round trip stands between the K loop and the epilogue (one per workgroup otherwise: measured +3 ... 6 us on the 1 500-workgroup launches)
  float4 c1[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) c1[j][g] = *reinterpret_cast<const float4*>(c1_lds + j * 32 + 8 * g + 4 * hi);
  float mean[TM], rstd[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {  // the two lane halves hold the two 8-element halves of every 16-element K step
    st.s1[i] += __shfl_xor(st.s1[i], 32);
    st.s2[i] += __shfl_xor(st.s2[i], 32);
  }
  if constexpr (WN > 1) {
    if (hi == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        *reinterpret_cast<float2*>(lds + ((wn * BM) + row0 + i * 32 + l31) * 2) = make_float2(st.s1[i], st.s2[i]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float a = 0.0f, b = 0.0f;
#pragma unroll
      for (int w = 0; w < WN; ++w) {  // fixed order: every wave of a row band computes the same bits
        const float2 v = *reinterpret_cast<const float2*>(lds + ((w * BM) + row0 + i * 32 + l31) * 2);
        a += v.x;
        b += v.y;
      }
      st.s1[i] = a;
      st.s2[i] = b;
    }
  }
  const float invk = 1.0f / (float)p.K;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    mean[i] = st.s1[i] * invk;
    const float var = fmaxf(st.s2[i] * invk - mean[i] * mean[i], 0.0f);
    rstd[i] = __frsqrt_rn(var + p.ln_eps);
    mean[i] *= -rstd[i];  // now -rstd * mean
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 c = c1[j][g];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        acc[j][i][4 * g + 0] = rstd[i] * acc[j][i][4 * g + 0] + mean[i] * c.x;
        acc[j][i][4 * g + 1] = rstd[i] * acc[j][i][4 * g + 1] + mean[i] * c.y;
        acc[j][i][4 * g + 2] = rstd[i] * acc[j][i][4 * g + 2] + mean[i] * c.z;
        acc[j][i][4 * g + 3] = rstd[i] * acc[j][i][4 * g + 3] + mean[i] * c.w;
      }
    }
}

// LDS-DMA (`buffer_load_dwordx4 ... lds`) plumbing shared by the DMA-staged kernels
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// waves per SIMD a tile kernel reaches when LDS is the only limit (160 KB per CU): passed to __launch_bounds__ so that the register
// allocator keeps the (load-batching, hence register-hungry) epilogue inside the budget of that occupancy
constexpr int gemm_waves_per_simd(int lds_bytes, int waves_per_block) {
  const int blocks = 160 * 1024 / lds_bytes;
  const int w = blocks * waves_per_block / 4;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}
// GN_PIN(x): x is ONE value in ONE register here.  Without it hipcc turns `offset = ok ? computed : kOOB` in front of an LDS-DMA builtin
// into two exec-masked arms with a DMA instruction each (s_and_saveexec / s_cbranch_execz around every piece: sixteen branches per K
// tile in the 128x128 kernel) -- removing them: 98.8 vs 100.3 ms per tiled call on one box.
#define GN_PIN(x) asm volatile("" : "+v"(x))
constexpr unsigned kOOB = 0xFFFFFFF0u;  // out-of-range buffer offset: the hardware writes zeros to LDS for such lanes

}  // namespace

// the ping-pong 256x256 kernel lives in its own translation unit (gemm_pp.hip); `params` is a GemmParams
void gn_launch_gemm_pp(const void* params, bool conv, int grid_x, int grid_y, int grid_z, hipStream_t st);
// the persistent skewed ping-pong kernel (gemm_ppp.hip); the plan fields pp* of `params` are filled by gn_ppp_plan
void gn_launch_gemm_ppp(const void* params, bool conv, hipStream_t st);
int gn_ppp_plan(void* params, int tiles, int G);  // fills the pp* plan fields -> hand-off slabs (of 256 KB) the launch may use
// the 3-stage ring variants (gemm_s3.hip); cfg 0..3 = {128x128, 128x64, 64x64, 256x64}
void gn_launch_gemm_s3(const void* params, int cfg, bool conv, int grid_x, int grid_y, int grid_z, hipStream_t st);
