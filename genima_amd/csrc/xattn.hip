// Cross-attention block front, fused for gfx950: [LayerNorm -> to_q] -> softmax(q K^T / sqrt(d)) V for the SHORT key sequences of the
// prompt (Nk <= 96: CLIP's 77 tokens), head dim 64 -- diffusers BasicTransformerBlock.norm2 -> attn2.to_q -> attn2 (SDPA) of every
// Transformer2DModel in the UNet / ControlNet (the reference's calls: a LayerNorm kernel, a cuBLAS GEMM and an SDPA launch per block;
// SURVEY.md K4/K5/K7).  K and V^T of the prompt are hoisted out of the denoise loop by the caller (graphs.emit_cross_kv).
//
// One workgroup = 4 waves = 128 query rows of ONE head:
//   phase 1 (wq != NULL)  q^T[d, row] = Wq'_h[d, :] . x[row, :]  over K = C in tiles of 64: the weight tile goes through LDS (every wave
//            needs all of it), each lane reads ITS row's 16-byte slices of x straight from global memory as the MFMA B operand (no LDS,
//            no barrier traffic for the activation), and takes the row's sum / sum of squares from the same registers (v_dot2) -- the
//            LayerNorm is folded as in gn_gemm_desc.ln_c1:  q = rstd * (x . W'^T - mean * c1) + c2;
//   phase 2  S^T = K . q^T for all (<= 96) keys at once -- the whole key range is ONE tile, so the softmax is the exact two-pass one (row
//            max, exp2, sum; no online rescaling, no key loop, no second barrier) -- then O^T = V^T . P^T.
// The accumulators of one MFMA feed the next one's B operand without a shuffle: a 32 x 32 accumulator block leaves lane (l31, hi) with
// rows 8 g + 4 hi + e of column l31, so k-slot t of half hi is bound to row 16 s + 8 (t >> 2) + 4 hi + (t & 3) and the A operand of the
// consuming MFMA (K rows / V^T rows in LDS) is read as two 8-byte pieces in that order (a consistent bijection of the contraction index).
// LDS: 2 x 8 KB weight tiles (XOR-swizzled 128-byte rows) + K [96][64] at a 136-byte row pitch + V^T [64][96] at a 200-byte pitch (both
// conflict-free for the 8-byte fragment reads) = 42 KB.
#include "attention_common.h"

namespace {

struct XAttnParams {
  const f16* x; const f16* wq; const float* c1; const f16* c2; const f16* k; const f16* vt; f16* o;
  long x_rs, w_rs, o_rs, k_bs, k_rs, vt_bs, vt_rs;
  int M, Nq, C, heads, Nk;
  float scale_log2, ln_eps;
};

constexpr int XK_PITCH = 136;   // bytes per K row in LDS (64 d + pad)
constexpr int XV_PITCH = 200;   // bytes per V^T row in LDS (96 keys + pad)
constexpr int XW_TILE = 64 * 128;
constexpr int XKEYS = 96;

typedef _Float16 xh4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 join8(const xh4 a, const xh4 b) {
  f16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
  return r;
}

template <bool PROJ>
__global__ __launch_bounds__(256, 2) void xattn_kernel(const XAttnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * XW_TILE + XKEYS * XK_PITCH + 64 * XV_PITCH];
  unsigned char* Ws = smem;
  unsigned char* Ks = smem + 2 * XW_TILE;
  unsigned char* Vs = Ks + XKEYS * XK_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nbands = p.M / 128;
  // heads of one row band sit next to each other in dispatch order: they read the same x rows (L2) -- and bands of one sample share K / V^T
  const int band = blockIdx.x / p.heads, h = blockIdx.x - band * p.heads;
  (void)nbands;
  const int row0 = band * 128;
  const int b = row0 / p.Nq;
  const int row = row0 + wave * 32 + l31;  // this lane's query row (global row index b * Nq + n)
  const int hc = h * 64;

  // ---- K [96][64] and V^T [64][96] of this (sample, head) into LDS: 8-byte pieces, rows / columns >= Nk zeroed -------------------
  {
    const f16* kp = p.k + (long)b * p.k_bs + hc;
    for (int idx = tid; idx < XKEYS * 16; idx += 256) {
      const int key = idx >> 4, piece = idx & 15;  // 16 pieces of 4 d per key row
      xh4 v = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
      if (key < p.Nk) v = *reinterpret_cast<const xh4*>(kp + (long)key * p.k_rs + piece * 4);
      *reinterpret_cast<xh4*>(Ks + key * XK_PITCH + piece * 8) = v;
    }
    const f16* vp = p.vt + (long)b * p.vt_bs + (long)hc * p.vt_rs;
    for (int idx = tid; idx < 64 * 24; idx += 256) {
      const int d = idx / 24, piece = idx - d * 24;  // 24 pieces of 4 keys per d row
      xh4 v = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
      const int k0 = piece * 4;
      if (k0 < p.Nk) {
        v = *reinterpret_cast<const xh4*>(vp + (long)d * p.vt_rs + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0 + e >= p.Nk) v[e] = (f16)0.0f;  // pad columns of V^T are not trusted (0 * NaN)
      }
      *reinterpret_cast<xh4*>(Vs + d * XV_PITCH + piece * 8) = v;
    }
  }

  // ---- q fragments: B operand of S^T = K . q^T; slot t of half hi of k16-step (jb, s) holds d = 32 jb + 16 s + 8 (t >> 2) + 4 hi + (t & 3)
  f16x8 qf[2][2];
  if constexpr (PROJ) {
    const f16* xrow = p.x + (long)row * p.x_rs;
    const f16* wbase = p.wq + (long)hc * p.w_rs;
    const int ntile = p.C / 64;
    f32x16 acc[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jb][r] = 0.0f;
    float sx = 0.0f, sxx = 0.0f;
    const f16x2 ones = {(f16)1.0f, (f16)1.0f};
    uint4 w0, w1;  // (named, not an array: hipcc parks a small private array that lambdas touch in LDS -- 8 KB per workgroup here)
    f16x8 xr[4], xn[4];
    const int wrow = tid >> 3, wch = tid & 7;
    const f16* wsrc = wbase + (long)wrow * p.w_rs + wch * 8;
    const long w32 = 32 * p.w_rs;
    auto load_w = [&](int t) {
      w0 = *reinterpret_cast<const uint4*>(wsrc + t * 64);
      w1 = *reinterpret_cast<const uint4*>(wsrc + w32 + t * 64);
    };
    auto store_w = [&](int buf) {
      *reinterpret_cast<uint4*>(Ws + buf * XW_TILE + lds_swz<128>(wrow, wch)) = w0;
      *reinterpret_cast<uint4*>(Ws + buf * XW_TILE + lds_swz<128>(wrow + 32, wch)) = w1;
    };
    auto load_x = [&](int t, f16x8 (&dst)[4]) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = *reinterpret_cast<const uint4*>(xrow + t * 64 + ks * 16 + hi * 8);
        dst[ks] = *reinterpret_cast<const f16x8*>(&v);
      }
    };
    load_w(0);
    load_x(0, xr);
    store_w(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < ntile; ++t) {
      const bool more = t + 1 < ntile;
      if (more) {
        load_w(t + 1);
        load_x(t + 1, xn);
      }
      const unsigned char* Wt = Ws + cur * XW_TILE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // row statistics from the fragment this lane holds anyway (its half of the row's 16 values of this k16 step)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f16x2 pr = {xr[ks][e], xr[ks][e + 1]};
          sx = __builtin_amdgcn_fdot2(pr, ones, sx, false);
          sxx = __builtin_amdgcn_fdot2(pr, pr, sxx, false);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const f16x8 wf = *reinterpret_cast<const f16x8*>(Wt + lds_swz<128>(jb * 32 + l31, ks * 2 + hi));
          acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xr[ks], acc[jb], 0, 0, 0);
        }
      }
      if (more) store_w(cur ^ 1);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xr[ks] = xn[ks];
      cur ^= 1;
    }
    // LayerNorm fold (biased variance) + softmax scale, straight into the B-operand fragments
    const float invC = 1.0f / (float)p.C;
    const float mean = pair_sum(sx) * invC;
    float var = pair_sum(sxx) * invC - mean * mean;
    var = var < 0.0f ? 0.0f : var;
    const float rstd = rsqrtf(var + p.ln_eps);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = hc + 32 * jb + 8 * g + 4 * hi;
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(p.c1 + d0);
        const xh4 c2 = *reinterpret_cast<const xh4*>(p.c2 + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float qv = rstd * (acc[jb][4 * g + e] - mean * c1[e]) + (float)c2[e];
          // the f16 rounding the stand-alone to_q launch applies to its output, then the exponent scale (as the attention kernel does)
          qf[jb][g >> 1][4 * (g & 1) + e] = (f16)((float)(f16)qv * p.scale_log2);
        }
      }
  } else {
    const f16* qrow = p.x + (long)row * p.x_rs + hc;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int d0 = 32 * jb + 16 * s + 4 * hi;
        const f16x8 q8 = join8(*reinterpret_cast<const xh4*>(qrow + d0), *reinterpret_cast<const xh4*>(qrow + d0 + 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[jb][s][e] = (f16)((float)q8[e] * p.scale_log2);
      }
    __syncthreads();  // K / V^T tiles are in place
  }

  // ---- S^T[key, row] for the three 32-key blocks: 12 MFMAs -----------------------------------------------------------------------
  f32x16 sc[3];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[kb][r] = 0.0f;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned char* kr = Ks + (kb * 32 + l31) * XK_PITCH + (32 * jb + 16 * s + 4 * hi) * 2;
        const f16x8 kf = join8(*reinterpret_cast<const xh4*>(kr), *reinterpret_cast<const xh4*>(kr + 16));
        sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[jb][s], sc[kb], 0, 0, 0);
      }
  }
  // ---- exact softmax over the (<= 96) keys of this row: accumulator r of block kb is key 32 kb + 8 (r >> 2) + 4 hi + (r & 3) ---------
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3);
      const float v = key < p.Nk ? sc[kb][r] : -INFINITY;
      sc[kb][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = pair_max(mx);
  float lsum = 0.0f;
  f16x8 pf[3][2];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f16 ph = (f16)__builtin_amdgcn_exp2f(sc[kb][r] - mx);
      lsum += (float)ph;  // the sum of the f16 values that enter P . V
      pf[kb][r >> 3][r & 7] = ph;
    }
  lsum = pair_sum(lsum);
  // ---- O^T[d, row] = V^T . P^T: 12 MFMAs --------------------------------------------------------------------------------------------
  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned char* vr = Vs + (dt * 32 + l31) * XV_PITCH + (32 * kb + 16 * s + 4 * hi) * 2;
        const f16x8 vf = join8(*reinterpret_cast<const xh4*>(vr), *reinterpret_cast<const xh4*>(vr + 16));
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s], oacc[dt], 0, 0, 0);
      }
  }
  const float inv = 1.0f / lsum;
  f16* op = p.o + (long)row * p.o_rs + hc;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      xh4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (f16)(oacc[dt][4 * g + e] * inv);
      *reinterpret_cast<xh4*>(op + dt * 32 + 8 * g + 4 * hi) = v;
    }
}

}  // namespace

int32_t gn_launch_xattn(gn_ctx* ctx, const gn_xattn_desc* d) {
  GN_REQUIRE(d && d->x && d->k && d->vt && d->o, "gn_cross_attention: null pointer");
  GN_REQUIRE(d->B > 0 && d->Nq > 0 && d->heads > 0 && d->C == d->heads * 64, "gn_cross_attention: head dim must be 64 (C = %d, heads = %d)", d->C, d->heads);
  GN_REQUIRE(d->Nq % 128 == 0, "gn_cross_attention: Nq (%d) must be a multiple of 128 (use gn_attention_fwd otherwise)", d->Nq);
  GN_REQUIRE(d->Nk >= 1 && d->Nk <= XKEYS, "gn_cross_attention: Nk (%d) must be in [1, %d]", d->Nk, XKEYS);
  GN_REQUIRE(d->x_rs % 8 == 0 && d->o_rs % 4 == 0 && d->k_rs % 4 == 0 && d->vt_rs % 4 == 0 && d->k_bs % 4 == 0 && d->vt_bs % 4 == 0,
             "gn_cross_attention: stride alignment");
  GN_REQUIRE(d->vt_rs >= ((d->Nk + 3) / 4) * 4, "gn_cross_attention: vt row stride %ld must cover Nk = %d rounded up to 4", (long)d->vt_rs, d->Nk);
  GN_REQUIRE(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->k & 7) == 0 && ((uintptr_t)d->vt & 7) == 0 && ((uintptr_t)d->o & 7) == 0,
             "gn_cross_attention: pointer alignment");
  GN_REQUIRE(d->scale > 0.0f, "gn_cross_attention: scale must be positive");
  XAttnParams p;
  p.x = (const f16*)d->x; p.wq = (const f16*)d->wq; p.c1 = d->ln_c1; p.c2 = (const f16*)d->ln_c2;
  p.k = (const f16*)d->k; p.vt = (const f16*)d->vt; p.o = (f16*)d->o;
  p.x_rs = d->x_rs; p.w_rs = d->w_rs; p.o_rs = d->o_rs; p.k_bs = d->k_bs; p.k_rs = d->k_rs; p.vt_bs = d->vt_bs; p.vt_rs = d->vt_rs;
  p.M = d->B * d->Nq; p.Nq = d->Nq; p.C = d->C; p.heads = d->heads; p.Nk = d->Nk;
  p.scale_log2 = d->scale * 1.4426950408889634f; p.ln_eps = d->ln_eps;
  const dim3 grid((unsigned)((p.M / 128) * d->heads));
  if (d->wq) {
    GN_REQUIRE(d->ln_c1 && d->ln_c2 && d->ln_eps > 0.0f && d->C % 64 == 0 && d->w_rs % 8 == 0 && ((uintptr_t)d->wq & 15) == 0 &&
               ((uintptr_t)d->ln_c1 & 15) == 0 && ((uintptr_t)d->ln_c2 & 7) == 0,
               "gn_cross_attention: the fused to_q needs the folded LayerNorm vectors (ln_c1 f32, ln_c2 f16), C %% 64 == 0 and aligned operands");
    hipLaunchKernelGGL((xattn_kernel<true>), grid, dim3(256), 0, ctx->stream, p);
  } else {
    hipLaunchKernelGGL((xattn_kernel<false>), grid, dim3(256), 0, ctx->stream, p);
  }
  GN_LAUNCH_CHECK();
  return GN_OK;
}

extern "C" int32_t gn_cross_attention(gn_ctx* ctx, const gn_xattn_desc* d) {
  GN_REQUIRE(ctx, "gn_cross_attention: null ctx");
  return gn_launch_xattn(ctx, d);
}
