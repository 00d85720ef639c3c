// GroupNorm(+SiLU) and LayerNorm for NHWC / token-major f16 activations on gfx950.  HBM-bound kernels:
// every global access is a coalesced 16-byte chunk, statistics are f32 (combined in f64), math is f32, storage f16.
//
// GroupNorm over NHWC cannot give one workgroup a contiguous (n, group) slab (a group's channels are 2..80 B of every
// C*2-byte pixel), so it is three launches:
//   1. gn_stats_kernel:    each block reduces a slab of pixels x ALL channels (fully coalesced) to per-channel sums in LDS,
//                          then to per-group (sum, sumsq) partials            -> ws.partials[B][chunks][G][2]
//   2. gn_finalize_kernel: combines the partials in a fixed order (f64) and folds mean/rstd/gamma/beta into per-(b, c)
//                          scale/shift                                          -> ws.scsh[B][C][2]
//   3. gn_apply_kernel:    y = act(x * scale + shift), one 16-byte chunk per thread; reads two virtually concatenated
//                          sources (UNet up-block skip concat) and writes the concatenated activated tensor.
// Algorithmic traffic is read + write once; the stats pass re-reads x (an L2 / Infinity-Cache hit for UNet-sized tensors).
#include "common.h"

namespace {

struct GNParams {
  const f16* x; const f16* x2; const f16* gamma; const f16* beta; f16* y;
  float* partials;  // [B][chunks][G][2]
  float* scsh;      // [B][C][2]
  int B, HW, C1, C2, C, G, cpg, chunks, rows, act;
  float eps;
};

__global__ __launch_bounds__(256) void gn_stats_kernel(const GNParams p) {
  // LDS: per-(pixel-lane, channel) partial sums [PY][C] and sums of squares [PY][C]; every slot has exactly one writer and
  // the group reduction below walks them in a fixed order, so the statistics are bit-reproducible run to run.
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int CC = p.C >> 3;
  const int TX = CC < 256 ? CC : 256;
  const int PY = 256 / TX;
  const int cxt = tid % TX, py = tid / TX;
  const int r0 = chunk * p.rows;
  const int r1 = min(p.HW, r0 + p.rows);
  float* lsum = lds;
  float* lsq = lds + PY * p.C;
  if (py < PY) {
    for (int cx = cxt; cx < CC; cx += TX) {
      const int c0 = cx * 8;
      const f16* src;
      int cs, co;
      if (c0 < p.C1) { src = p.x; cs = p.C1; co = c0; } else { src = p.x2; cs = p.C2; co = c0 - p.C1; }
      float s[8], ss[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
      for (int r = r0 + py; r < r1; r += PY) {
        const uint4 raw = *reinterpret_cast<const uint4*>(src + ((long)b * p.HW + r) * cs + co);
        const f16x8 v = *reinterpret_cast<const f16x8*>(&raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; ss[e] += f * f; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lsum[py * p.C + c0 + e] = s[e];
        lsq[py * p.C + c0 + e] = ss[e];
      }
    }
  }
  __syncthreads();
  if (tid < p.G) {
    float s = 0.f, ss = 0.f;
    for (int c = tid * p.cpg; c < (tid + 1) * p.cpg; ++c)
      for (int y = 0; y < PY; ++y) { s += lsum[y * p.C + c]; ss += lsq[y * p.C + c]; }
    float* out = p.partials + (((long)b * p.chunks + chunk) * p.G + tid) * 2;
    out[0] = s;
    out[1] = ss;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const GNParams p) {
  __shared__ float mean_s[256], rstd_s[256];
  const int tid = threadIdx.x, b = blockIdx.x;
  if (tid < p.G) {
    double s = 0.0, ss = 0.0;
    for (int ch = 0; ch < p.chunks; ++ch) {
      const float* in = p.partials + (((long)b * p.chunks + ch) * p.G + tid) * 2;
      s += (double)in[0];
      ss += (double)in[1];
    }
    const double n = (double)p.HW * (double)p.cpg;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)mean;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  for (int c = tid; c < p.C; c += 256) {
    const int g = c / p.cpg;
    const float a = rstd_s[g] * (float)p.gamma[c];
    float* o = p.scsh + ((long)b * p.C + c) * 2;
    o[0] = a;
    o[1] = (float)p.beta[c] - mean_s[g] * a;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const GNParams p) {
  const int CC = p.C >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= (long)p.HW * CC) return;
  const int r = (int)(idx / CC);
  const int c0 = (int)(idx - (long)r * CC) * 8;
  const f16* src;
  int cs, co;
  if (c0 < p.C1) { src = p.x; cs = p.C1; co = c0; } else { src = p.x2; cs = p.C2; co = c0 - p.C1; }
  const long pix = (long)b * p.HW + r;
  const uint4 raw = *reinterpret_cast<const uint4*>(src + pix * cs + co);
  const f16x8 v = *reinterpret_cast<const f16x8*>(&raw);
  const f32x4* sc = reinterpret_cast<const f32x4*>(p.scsh + ((long)b * p.C + c0) * 2);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 q = sc[e];  // (a0, s0, a1, s1)
    float y0 = (float)v[2 * e] * q[0] + q[1];
    float y1 = (float)v[2 * e + 1] * q[2] + q[3];
    if (p.act == GN_ACT_SILU) { y0 = act_silu(y0); y1 = act_silu(y1); }
    o[2 * e] = (f16)y0;
    o[2 * e + 1] = (f16)y1;
  }
  *reinterpret_cast<uint4*>(p.y + pix * p.C + c0) = *reinterpret_cast<uint4*>(&o);
}

int gn_pick_chunks(int B, int HW) {
  long c = cdiv64(2048, B);
  const long maxc = cdiv64(HW, 8);
  if (c > maxc) c = maxc;
  if (c < 1) c = 1;
  return (int)c;
}

// ---- LayerNorm: one wave per row, the row lives in registers ---------------------------------------------------------
constexpr int LN_MAXCH = 8;  // 16-byte chunks per lane -> C <= 4096

__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, const f16* __restrict__ gamma,
                                                        const f16* __restrict__ beta, f16* __restrict__ y, long M, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int CC = C >> 3;
  const f16* xr = x + row * C;
  f16x8 v[LN_MAXCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
      const uint4 raw = *reinterpret_cast<const uint4*>(xr + cx * 8);
      v[i] = *reinterpret_cast<const f16x8*>(&raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
  f16* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
      const uint4 graw = *reinterpret_cast<const uint4*>(gamma + cx * 8);
      const uint4 braw = *reinterpret_cast<const uint4*>(beta + cx * 8);
      const f16x8 g = *reinterpret_cast<const f16x8*>(&graw);
      const f16x8 bb = *reinterpret_cast<const f16x8*>(&braw);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)v[i][e] - mean) * rstd * (float)g[e] + (float)bb[e]);
      *reinterpret_cast<uint4*>(yr + cx * 8) = *reinterpret_cast<uint4*>(&o);
    }
  }
}

}  // namespace

extern "C" int64_t gn_groupnorm_workspace_bytes(const gn_groupnorm_desc* d) {
  if (!d) return 0;
  const int C = d->C1 + d->C2;
  const int chunks = gn_pick_chunks(d->B, d->HW);
  return ((int64_t)d->B * chunks * d->groups * 2 + (int64_t)d->B * C * 2) * (int64_t)sizeof(float);
}

int32_t gn_launch_groupnorm(gn_ctx* ctx, const gn_groupnorm_desc* d) {
  GN_REQUIRE(d && d->x && d->gamma && d->beta && d->y && d->workspace, "gn_groupnorm_fwd: null pointer");
  const int C = d->C1 + d->C2;
  GN_REQUIRE(d->B > 0 && d->HW > 0 && d->C1 > 0, "gn_groupnorm_fwd: empty problem");
  GN_REQUIRE(d->C1 % 8 == 0 && d->C2 % 8 == 0, "gn_groupnorm_fwd: C1/C2 (%d/%d) must be multiples of 8", d->C1, d->C2);
  GN_REQUIRE((d->C2 == 0) == (d->x2 == nullptr), "gn_groupnorm_fwd: x2 and C2 must be given together");
  GN_REQUIRE(d->groups > 0 && d->groups <= 256 && C % d->groups == 0, "gn_groupnorm_fwd: C=%d not divisible by groups=%d", C, d->groups);
  GN_REQUIRE(d->act == GN_ACT_NONE || d->act == GN_ACT_SILU, "gn_groupnorm_fwd: act must be NONE or SILU");
  GN_REQUIRE(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->y & 15) == 0 && ((uintptr_t)d->workspace & 15) == 0, "gn_groupnorm_fwd: 16-byte alignment");
  GNParams p;
  p.x = (const f16*)d->x; p.x2 = (const f16*)d->x2; p.gamma = (const f16*)d->gamma; p.beta = (const f16*)d->beta;
  p.y = (f16*)d->y;
  p.B = d->B; p.HW = d->HW; p.C1 = d->C1; p.C2 = d->C2; p.C = C; p.G = d->groups; p.cpg = C / d->groups;
  p.chunks = gn_pick_chunks(d->B, d->HW);
  p.rows = (int)cdiv64(d->HW, p.chunks);
  p.chunks = (int)cdiv64(d->HW, p.rows);
  p.act = d->act; p.eps = d->eps;
  p.partials = (float*)d->workspace;
  p.scsh = p.partials + (long)d->B * gn_pick_chunks(d->B, d->HW) * d->groups * 2;
  const int cc = C >> 3, tx = cc < 256 ? cc : 256, pyn = 256 / tx;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(p.chunks, p.B), dim3(256), (size_t)2 * pyn * C * sizeof(float), ctx->stream, p);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.B), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  const long per_b = (long)p.HW * (C >> 3);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)cdiv64(per_b, 256), p.B), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

extern "C" int32_t gn_layernorm_fwd(gn_ctx* ctx, const void* x, const void* gamma, const void* beta, void* y, int64_t M,
                                    int32_t C, float eps) {
  GN_REQUIRE(ctx && x && gamma && beta && y, "gn_layernorm_fwd: null pointer");
  GN_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAXCH, "gn_layernorm_fwd: C=%d must be a multiple of 8 and <= %d", C, 64 * 8 * LN_MAXCH);
  GN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0, "gn_layernorm_fwd: 16-byte alignment");
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)cdiv64(M, 4)), dim3(256), 0, ctx->stream, (const f16*)x,
                     (const f16*)gamma, (const f16*)beta, (f16*)y, (long)M, C, eps);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
