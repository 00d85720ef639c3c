// Shared pieces of the MFMA GEMM kernels (gemm.hip, gemm_pp.hip): kernel parameter block, activations and the fused
// epilogues (bias / time shift / residual / activation / GEGLU / f32 + accumulate / transposed / split-K slab).
#pragma once
#include <stdlib.h>

#include "common.h"

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
};

// batched GEMM: offset every operand of this workgroup's problem by its (outer, inner) batch strides
__device__ __forceinline__ GemmParams batch_offset(const GemmParams& pin) {
  GemmParams p = pin;
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

// Row-major f16 output, 16 bytes per lane.  The MFMA layout leaves a lane with channels 8g + 4hi + (0..3) of its row: 8-byte
// pieces, and the write path is issue-bound on those (removing the stores took 8 % (K = 1280) to 26 % (K = 320) off the Linear
// launches).  Lanes l and l + 32 hold the same row, so one v_permlane32_swap per dword trades group g of the upper half against
// group g + 1 of the lower half: afterwards the lower lane owns channels 8g .. 8g + 7 and the upper lane 8(g+1) .. 8(g+1) + 7.
__device__ __forceinline__ void epilogue_store8_pair(const GemmParams& p, int m, int nb8, int hi, const float* a, const float* b) {
  // a: this lane's 4 values of group g (channels nb8 + 4 hi ..), b: of group g + 1 (channels nb8 + 8 + 4 hi ..)
  float va[4] = {a[0], a[1], a[2], a[3]}, vb[4] = {b[0], b[1], b[2], b[3]};
  int bidx;
  if (nb8 + 4 * hi < p.N) epilogue_vals4(p, m, nb8 + 4 * hi, va, bidx);
  if (nb8 + 8 + 4 * hi < p.N) epilogue_vals4(p, m, nb8 + 8 + 4 * hi, vb, bidx);
  f16x4 ha, hb;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ha[i] = (f16)va[i]; hb[i] = (f16)vb[i]; }
  uint2 ua = *reinterpret_cast<uint2*>(&ha), ub = *reinterpret_cast<uint2*>(&hb);
  const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
  const uint4 o = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  const int col = nb8 + 8 * hi;
  if (col < p.N) *reinterpret_cast<uint4*>(p.out + (long)m * p.ldo + col) = o;
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
    *reinterpret_cast<f16x4*>(p.out + (long)m * p.ldo + nb) = o;
  }
}

// GEGLU: 4 consecutive packed hidden rows at nh and the matching gate rows at ng -> 4 output columns at oc.
__device__ __forceinline__ void epilogue_geglu4(const GemmParams& p, int m, int nh, int ng, int oc, const float* h,
                                                const float* g) {
  float hv[4], gv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { hv[i] = h[i]; gv[i] = g[i]; }
  if (p.bias) {
    f16x4 bh = *reinterpret_cast<const f16x4*>(p.bias + nh);
    f16x4 bg = *reinterpret_cast<const f16x4*>(p.bias + ng);
#pragma unroll
    for (int i = 0; i < 4; ++i) { hv[i] += (float)bh[i]; gv[i] += (float)bg[i]; }
  }
  f16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (f16)(hv[i] * gelu_fast(gv[i]));
  *reinterpret_cast<f16x4*>(p.out + (long)m * p.ldo + oc) = o;
}

// ---- wave-level epilogue: the lane holds D[n = 8g + 4hi + (r&3)][m = lane&31] of each 32x32 tile ----------------------------
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TN][TM], int mbase, int nbase, int l31, int hi,
                                              int z) {
  // plain f16 rows whose 8-channel groups are 16-byte aligned take the paired 16-byte stores
  const bool wide = p.out_mode == GN_OUT_ROWMAJOR && (p.ldo & 7) == 0 && (p.N & 7) == 0 && ((uintptr_t)p.out & 15) == 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + i * 32 + l31;
    if (m >= p.M) continue;
    if (p.splitk > 1) {
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
    } else if (p.act == GN_ACT_GEGLU) {
      if constexpr (TN % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TN; j += 2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nh = nbase + j * 32 + 8 * g + 4 * hi;
            if (nh + 32 < p.N) {
              const int oc = ((nbase + j * 32) >> 1) + 8 * g + 4 * hi;
              float h[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
              float gg[4] = {acc[j + 1][i][4 * g], acc[j + 1][i][4 * g + 1], acc[j + 1][i][4 * g + 2],
                             acc[j + 1][i][4 * g + 3]};
              epilogue_geglu4(p, m, nh, nh + 32, oc, h, gg);
            }
          }
      }
    } else if (wide) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          const int nb8 = nbase + j * 32 + 8 * g;
          if (nb8 < p.N) {  // wave-uniform
            const float a[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
            const float b[4] = {acc[j][i][4 * g + 4], acc[j][i][4 * g + 5], acc[j][i][4 * g + 6], acc[j][i][4 * g + 7]};
            epilogue_store8_pair(p, m, nb8, hi, a, b);
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nb = nbase + j * 32 + 8 * g + 4 * hi;
          if (nb < p.N)
            epilogue_store4(p, m, nb, acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
        }
    }
  }
}

// LDS-DMA (`buffer_load_dwordx4 ... lds`) plumbing shared by the DMA-staged kernels
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kOOB = 0xFFFFFFF0u;  // out-of-range buffer offset: the hardware writes zeros to LDS for such lanes

}  // namespace

// the ping-pong 256x256 kernel lives in its own translation unit (gemm_pp.hip); `params` is a GemmParams
void gn_launch_gemm_pp(const void* params, bool conv, int grid_x, int grid_y, int grid_z, hipStream_t st);
// the 3-stage ring variants (gemm_s3.hip); cfg 0..3 = {128x128, 128x64, 64x64, 256x64}
void gn_launch_gemm_s3(const void* params, int cfg, bool conv, int grid_x, int grid_y, int grid_z, hipStream_t st);
