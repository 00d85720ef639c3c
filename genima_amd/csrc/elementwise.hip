// Small HBM-bound kernels of the Genima hot path (SURVEY.md K9, K10, K14 + gathers): timestep embedding, scheduler
// elementwise ops, image pre/post-processing, residual adds, CLIP embedding gather, row softmax (VAE attention), max-pool.
// All f16 storage / f32 math, 16-byte accesses where the layout allows.
#include "common.h"
#include "gn_bridge.h"

namespace {

__global__ void timestep_embedding_kernel(const float* __restrict__ t, f16* __restrict__ out, int B, int dim, int flip,
                                          float freq_shift) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * dim) return;
  const int b = idx / dim, j = idx - b * dim;
  const int half = dim >> 1;
  const int k = j < half ? j : j - half;
  const float freq = expf(-9.210340371976184f * (float)k / ((float)half - freq_shift));  // ln(10000)
  const float arg = t[b] * freq;
  // diffusers: cat[sin, cos], then flip_sin_to_cos swaps the halves -> [cos, sin]
  const bool is_sin = flip ? (j >= half) : (j < half);
  out[idx] = (f16)(is_sin ? sinf(arg) : cosf(arg));
}

__global__ void scale_pad_kernel(const f16* __restrict__ x, f16* __restrict__ out, long pixels, int C, int Cpad, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * Cpad) return;
  const long p = idx / Cpad;
  const int c = (int)(idx - p * Cpad);
  out[idx] = c < C ? (f16)((float)x[p * C + c] * scale) : (f16)0.0f;
}

// out[p, 0:C] = x[p * ld1 + 0:C] * scale | out[p, C:C+C2] = x2[p * ld2 + 0:C2] * scale2 | zeros up to Cpad: channel concatenation of two
// narrow latent tensors into one 8-channel pixel (InstructPix2Pix: torch.cat([scaled noisy latents, image latents], dim=1))
__global__ void scale_cat_pad_kernel(const f16* __restrict__ x, const f16* __restrict__ x2, f16* __restrict__ out, long pixels, int C,
                                     int ld1, int C2, int ld2, int Cpad, float scale, float scale2) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * Cpad) return;
  const long p = idx / Cpad;
  const int c = (int)(idx - p * Cpad);
  float v = 0.0f;
  if (c < C) v = (float)x[p * ld1 + c] * scale;
  else if (c < C + C2) v = (float)x2[p * ld2 + (c - C)] * scale2;
  out[idx] = (f16)v;
}

__global__ void euler_step_kernel(f16* __restrict__ x, const f16* __restrict__ eps, long pixels, int C, int ld, float sigma,
                                  float sigma_next) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * C) return;
  const long p = idx / C;
  const int c = (int)(idx - p * C);
  // diffusers EulerDiscreteScheduler.step, epsilon prediction, gamma = 0, in f32 then cast back
  const float xf = (float)x[idx];
  const float e = (float)eps[p * ld + c];
  // explicit round-to-nearest intrinsics: no FMA contraction, so the f32 op order matches the reference bit for bit
  const float x0 = __fsub_rn(xf, __fmul_rn(sigma, e));
  const float d = __fdiv_rn(__fsub_rn(xf, x0), sigma);
  x[idx] = (f16)__fadd_rn(xf, __fmul_rn(d, __fsub_rn(sigma_next, sigma)));
}

__global__ void add_noise_kernel(const f16* __restrict__ x0, const f16* __restrict__ noise, const float* __restrict__ a,
                                 const float* __restrict__ c, f16* __restrict__ out, long per_sample) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= per_sample) return;
  const long idx = (long)b * per_sample + i;
  out[idx] = (f16)(a[b] * (float)x0[idx] + c[b] * (float)noise[idx]);
}

__global__ void image_u8_to_f16_kernel(const uint8_t* __restrict__ in, f16* __restrict__ out, long pixels, int Cpad, float mul,
                                       float add) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  f16 v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = (f16)0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = (f16)((float)in[p * 3 + c] / 255.0f * mul + add);
  for (int c = 0; c < Cpad; ++c) out[p * Cpad + c] = v[c < 8 ? c : 7];
}

struct Norm3 { float m[3], a[3]; };
__global__ void image_normalize_u8_kernel(const uint8_t* __restrict__ in, f16* __restrict__ out, long pixels, int Cpad, Norm3 n) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  for (int c = 0; c < Cpad; ++c)
    out[p * Cpad + c] = c < 3 ? (f16)((float)in[p * 3 + c] * n.m[c] + n.a[c]) : (f16)0.0f;
}

__global__ void gather_rows_kernel(const f16* __restrict__ x, const int32_t* __restrict__ idx, f16* __restrict__ out, int B, int L,
                                   int D) {
  const int DC = D >> 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * DC) return;
  const int b = i / DC, c0 = (i - b * DC) * 8;
  *reinterpret_cast<uint4*>(out + (long)b * D + c0) = *reinterpret_cast<const uint4*>(x + ((long)b * L + idx[b]) * D + c0);
}

// generic 4-D strided copy of contiguous L-element (L % 8 == 0) runs: the layout shuffles between kernels that the host would
// otherwise do with torch permute/cat (e.g. ACT: per-view feature maps -> views-along-width token rows inside the encoder
// sequence buffer)
struct Copy4 { long n[4], is[4], os[4]; };
__global__ void copy4d_kernel(const f16* __restrict__ in, f16* __restrict__ out, Copy4 c, int L8) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = c.n[0] * c.n[1] * c.n[2] * c.n[3] * L8;
  if (idx >= total) return;
  const int ch = (int)(idx % L8);
  long r = idx / L8;
  const long i3 = r % c.n[3]; r /= c.n[3];
  const long i2 = r % c.n[2]; r /= c.n[2];
  const long i1 = r % c.n[1];
  const long i0 = r / c.n[1];
  const long io = i0 * c.is[0] + i1 * c.is[1] + i2 * c.is[2] + i3 * c.is[3] + ch * 8;
  const long oo = i0 * c.os[0] + i1 * c.os[1] + i2 * c.os[2] + i3 * c.os[3] + ch * 8;
  *reinterpret_cast<uint4*>(out + oo) = *reinterpret_cast<const uint4*>(in + io);
}

// first index of the row maximum of an int32 matrix (CLIP EOT token = highest id, genima_act.py:339-342)
__global__ void argmax_rows_i32_kernel(const int32_t* __restrict__ x, int32_t* __restrict__ out, int rows, int cols) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int best = 0, bv = x[(long)r * cols];
  for (int j = 1; j < cols; ++j) {
    const int v = x[(long)r * cols + j];
    if (v > bv) { bv = v; best = j; }
  }
  out[r] = best;
}

__global__ void image_f16_to_u8_kernel(const f16* __restrict__ in, uint8_t* __restrict__ out, long pixels, int ld) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // the reference does (x / 2 + 0.5).clamp(0, 1) on the f16 tensor, then float -> *255 -> round (half to even)
    f16 h = (f16)((float)in[p * ld + c] * 0.5f + 0.5f);
    float f = fminf(fmaxf((float)h, 0.0f), 1.0f);
    out[p * 3 + c] = (uint8_t)rintf(f * 255.0f);
  }
}

__global__ void add_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 ra = a[i], rb = b[i];
  const f16x8 va = *reinterpret_cast<const f16x8*>(&ra), vb = *reinterpret_cast<const f16x8*>(&rb);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)((float)va[e] + (float)vb[e]);
  out[i] = *reinterpret_cast<uint4*>(&o);
}

// up to GN_ADD_MULTI_MAX independent  out = a + b  in ONE launch (blockIdx.y = which): the UNet's twelve skip + ControlNet-residual adds and the
// mid-block one sit behind the stream join on the call's critical path, 13 launch boundaries per denoise step for a few microseconds of work
struct AddMultiArgs { const uint4* a[GN_ADD_MULTI_MAX]; const uint4* b[GN_ADD_MULTI_MAX]; uint4* out[GN_ADD_MULTI_MAX]; long n8[GN_ADD_MULTI_MAX]; };
__global__ void add_multi_kernel(const AddMultiArgs p) {
  const int t = blockIdx.y;
  const uint4* __restrict__ a = p.a[t];
  const uint4* __restrict__ b = p.b[t];
  uint4* __restrict__ out = p.out[t];
  const long n8 = p.n8[t];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const uint4 ra = a[i], rb = b[i];
    const f16x8 va = *reinterpret_cast<const f16x8*>(&ra), vb = *reinterpret_cast<const f16x8*>(&rb);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)va[e] + (float)vb[e]);
    out[i] = *reinterpret_cast<uint4*>(&o);
  }
}

// the same adds as producers of the GroupNorm bridge (gn_bridge.h): tensor t is [rows][C] and a block owns a slab of rows x all channels
// (coalesced 16-byte chunks, the (chunk, pixel-lane) thread map of norm.hip's statistics kernel), so the sums of what it stores are a fixed-
// order LDS reduction + at most 2 x groups integer atomics per block
struct AddMultiStatsArgs { AddMultiArgs base; int C[GN_ADD_MULTI_MAX]; GnSinkP sink[GN_ADD_MULTI_MAX]; };
__global__ __launch_bounds__(256) void add_multi_stats_kernel(const AddMultiStatsArgs p) {
  extern __shared__ float lds[];  // [PY][C] sums, [PY][C] squares
  const int t = blockIdx.y, tid = threadIdx.x;
  const int C = p.C[t], CC = C >> 3;
  const long rows = p.base.n8[t] / CC;
  const long rows_blk = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * rows_blk, r1 = r0 + rows_blk < rows ? r0 + rows_blk : rows;
  if (r0 >= rows) return;  // (block-uniform)
  const uint4* __restrict__ a = p.base.a[t];
  const uint4* __restrict__ b = p.base.b[t];
  uint4* __restrict__ out = p.base.out[t];
  const GnSinkP sk = p.sink[t];
  const int TX = CC < 256 ? CC : 256, PY = 256 / TX;
  const int cxt = tid % TX, py = tid / TX;
  float* lsum = lds;
  float* lsq = lds + PY * C;
  const long rps = sk.stats ? sk.rps : rows;
  for (long sb = r0 / rps; sb <= (r1 - 1) / rps; ++sb) {
    const long s0 = r0 > sb * rps ? r0 : sb * rps, s1 = r1 < (sb + 1) * rps ? r1 : (sb + 1) * rps;
    if (py < PY) {
      for (int cx = cxt; cx < CC; cx += TX) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        for (long r = s0 + py; r < s1; r += PY) {
          const uint4 ra = a[r * CC + cx], rb = b[r * CC + cx];
          const f16x8 va = *reinterpret_cast<const f16x8*>(&ra), vb = *reinterpret_cast<const f16x8*>(&rb);
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o[e] = (f16)((float)va[e] + (float)vb[e]);
            const float f = (float)o[e];
            s[e] += f;
            q[e] += f * f;
          }
          out[r * CC + cx] = *reinterpret_cast<uint4*>(&o);
        }
        if (sk.stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { lsum[py * C + cx * 8 + e] = s[e]; lsq[py * C + cx * 8 + e] = q[e]; }
        }
      }
    }
    if (!sk.stats) continue;  // (block-uniform)
    __syncthreads();
    const int g0 = sk.coff / sk.cpg, g1 = (sk.coff + C - 1) / sk.cpg;
    for (int g = g0 + tid; g <= g1; g += 256) {
      const int c0 = max(g * sk.cpg - sk.coff, 0), c1 = min((g + 1) * sk.cpg - sk.coff, C);
      float sa = 0.f, sq = 0.f;
      for (int c = c0; c < c1; ++c)
        for (int y = 0; y < PY; ++y) { sa += lsum[y * C + c]; sq += lsq[y * C + c]; }
      gn_stats_add(sk.stats + gn_stats_line((int)(blockIdx.x % sk.reps), (int)sb, g, sk.nb, sk.groups), sa, sq);
    }
    __syncthreads();
  }
}

__global__ void act_kernel(const uint4* __restrict__ a, uint4* __restrict__ out, long n8, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 ra = a[i];
  const f16x8 va = *reinterpret_cast<const f16x8*>(&ra);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)apply_act((float)va[e], act);
  out[i] = *reinterpret_cast<uint4*>(&o);
}

// FiLM: out[r, c] = act((1 + gamma[b, c]) * x[r, c] + beta[b, c]),  b = r / rows_per_film  (language conditioning inside the ACT image
// encoder's ResNet blocks: after bn1, before the ReLU).  gamma / beta are rows of one [B, ld_film] feature buffer.
__global__ void film_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                            long ld_film, long rows_per_film, long rows, int C8, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C8) return;
  const long r = i / C8;
  const int c = (int)(i - r * C8) * 8;
  const long b = r / rows_per_film;
  const uint4 rx = x[i];
  const uint4 rg = *reinterpret_cast<const uint4*>(gamma + b * ld_film + c), rb = *reinterpret_cast<const uint4*>(beta + b * ld_film + c);
  const f16x8 vx = *reinterpret_cast<const f16x8*>(&rx), vg = *reinterpret_cast<const f16x8*>(&rg), vb = *reinterpret_cast<const f16x8*>(&rb);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)apply_act((1.0f + (float)vg[e]) * (float)vx[e] + (float)vb[e], act);
  out[i] = *reinterpret_cast<uint4*>(&o);
}

__global__ void embedding_kernel(const int32_t* __restrict__ ids, const f16* __restrict__ tok, const f16* __restrict__ pos,
                                 f16* __restrict__ out, int B, int L, int D) {
  const int DC = D >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * L * DC) return;
  const long row = idx / DC;
  const int c0 = (int)(idx - row * DC) * 8;
  const int i = (int)(row % L);
  const long id = ids[row];
  const uint4 rt = *reinterpret_cast<const uint4*>(tok + id * D + c0);
  const uint4 rp = *reinterpret_cast<const uint4*>(pos + (long)i * D + c0);
  const f16x8 vt = *reinterpret_cast<const f16x8*>(&rt), vp = *reinterpret_cast<const f16x8*>(&rp);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)((float)vt[e] + (float)vp[e]);
  *reinterpret_cast<uint4*>(out + row * D + c0) = *reinterpret_cast<uint4*>(&o);
}

constexpr int SM_MAXCH = 8;  // cols <= 4096
// columns >= valid (key padding up to the 8-column granule) take no probability mass and are written as zeros
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16* __restrict__ x, long rows, int cols, int ld, float scale, int valid) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f16* xr = x + row * ld;
  const int CC = cols >> 3;
  float v[SM_MAXCH][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SM_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
      const uint4 raw = *reinterpret_cast<const uint4*>(xr + cx * 8);
      const f16x8 h = *reinterpret_cast<const f16x8*>(&raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (cx * 8 + e < valid) ? (float)h[e] * scale : -INFINITY;
        mx = fmaxf(mx, v[i][e]);
      }
    }
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = __expf(v[i][e] - mx); s += v[i][e]; }
    }
  }
  const float inv = 1.0f / wave_sum(s);
#pragma unroll
  for (int i = 0; i < SM_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(v[i][e] * inv);
      *reinterpret_cast<uint4*>(xr + cx * 8) = *reinterpret_cast<uint4*>(&o);
    }
  }
}

__global__ void maxpool3x3s2_kernel(const f16* __restrict__ x, f16* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
  const int CC = C >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Ho * Wo * CC) return;
  const long pix = idx / CC;
  const int c0 = (int)(idx - pix * CC) * 8;
  const int ox = (int)(pix % Wo);
  const int oy = (int)((pix / Wo) % Ho);
  const int b = (int)(pix / ((long)Wo * Ho));
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const int iy = oy * 2 - 1 + dy, ix = ox * 2 - 1 + dx;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        const uint4 raw = *reinterpret_cast<const uint4*>(x + (((long)b * H + iy) * W + ix) * C + c0);
        const f16x8 v = *reinterpret_cast<const f16x8*>(&raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
      }
    }
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)m[e];
  *reinterpret_cast<uint4*>(y + pix * C + c0) = *reinterpret_cast<uint4*>(&o);
}

inline unsigned nblk(long n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" {

int32_t gn_timestep_embedding(gn_ctx* ctx, const float* t, void* out, int32_t B, int32_t dim, int32_t flip, float freq_shift) {
  GN_REQUIRE(ctx && t && out && B > 0 && dim > 0 && dim % 2 == 0, "gn_timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblk((long)B * dim)), dim3(256), 0, ctx->stream, t, (f16*)out, B, dim, flip, freq_shift);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_scale_pad(gn_ctx* ctx, const void* x, void* out, int64_t pixels, int32_t C, int32_t Cpad, float scale) {
  GN_REQUIRE(ctx && x && out && pixels > 0 && C > 0 && Cpad >= C, "gn_scale_pad: bad arguments");
  hipLaunchKernelGGL(scale_pad_kernel, dim3(nblk(pixels * Cpad)), dim3(256), 0, ctx->stream, (const f16*)x, (f16*)out, (long)pixels, C, Cpad, scale);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_scale_cat_pad(gn_ctx* ctx, const void* x, const void* x2, void* out, int64_t pixels, int32_t C, int32_t ld1, int32_t C2,
                         int32_t ld2, int32_t Cpad, float scale, float scale2) {
  GN_REQUIRE(ctx && x && x2 && out && pixels > 0 && C > 0 && ld1 >= C && C2 > 0 && ld2 >= C2 && Cpad >= C + C2, "gn_scale_cat_pad: bad arguments");
  hipLaunchKernelGGL(scale_cat_pad_kernel, dim3(nblk(pixels * Cpad)), dim3(256), 0, ctx->stream, (const f16*)x, (const f16*)x2, (f16*)out,
                     (long)pixels, C, ld1, C2, ld2, Cpad, scale, scale2);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_euler_step(gn_ctx* ctx, void* x, const void* eps, int64_t pixels, int32_t C, int32_t ld_eps, float sigma, float sigma_next) {
  GN_REQUIRE(ctx && x && eps && pixels > 0 && C > 0 && ld_eps >= C && sigma > 0.0f, "gn_euler_step: bad arguments");
  hipLaunchKernelGGL(euler_step_kernel, dim3(nblk(pixels * C)), dim3(256), 0, ctx->stream, (f16*)x, (const f16*)eps, (long)pixels, C, ld_eps, sigma, sigma_next);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_add_noise(gn_ctx* ctx, const void* x0, const void* noise, const float* sqrt_ac, const float* sqrt_1mac, void* out, int32_t B, int64_t per_sample) {
  GN_REQUIRE(ctx && x0 && noise && sqrt_ac && sqrt_1mac && out && B > 0 && per_sample > 0, "gn_add_noise: bad arguments");
  hipLaunchKernelGGL(add_noise_kernel, dim3(nblk(per_sample), B), dim3(256), 0, ctx->stream, (const f16*)x0, (const f16*)noise, sqrt_ac, sqrt_1mac, (f16*)out, (long)per_sample);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_image_u8_to_f16(gn_ctx* ctx, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float mul, float add) {
  GN_REQUIRE(ctx && in && out && pixels > 0 && Cpad >= 3 && Cpad <= 8, "gn_image_u8_to_f16: bad arguments");
  hipLaunchKernelGGL(image_u8_to_f16_kernel, dim3(nblk(pixels)), dim3(256), 0, ctx->stream, in, (f16*)out, (long)pixels, Cpad, mul, add);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_image_normalize_u8(gn_ctx* ctx, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float m0, float m1, float m2,
                              float a0, float a1, float a2) {
  GN_REQUIRE(ctx && in && out && pixels > 0 && Cpad >= 3, "gn_image_normalize_u8: bad arguments");
  Norm3 n{{m0, m1, m2}, {a0, a1, a2}};
  hipLaunchKernelGGL(image_normalize_u8_kernel, dim3(nblk(pixels)), dim3(256), 0, ctx->stream, in, (f16*)out, (long)pixels, Cpad, n);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_gather_rows(gn_ctx* ctx, const void* x, const int32_t* idx, void* out, int32_t B, int32_t L, int32_t D) {
  GN_REQUIRE(ctx && x && idx && out && B > 0 && L > 0 && D > 0 && D % 8 == 0, "gn_gather_rows: bad arguments");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk((long)B * (D / 8))), dim3(256), 0, ctx->stream, (const f16*)x, idx, (f16*)out, B, L, D);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_copy4d(gn_ctx* ctx, const void* in, void* out, const int64_t* sizes, const int64_t* in_strides,
                  const int64_t* out_strides, int32_t L) {
  GN_REQUIRE(ctx && in && out && sizes && in_strides && out_strides && L > 0 && L % 8 == 0, "gn_copy4d: bad arguments");
  Copy4 c;
  long total = L / 8;
  for (int i = 0; i < 4; ++i) {
    GN_REQUIRE(sizes[i] > 0 && in_strides[i] % 8 == 0 && out_strides[i] % 8 == 0, "gn_copy4d: sizes > 0 and strides multiples of 8");
    c.n[i] = sizes[i]; c.is[i] = in_strides[i]; c.os[i] = out_strides[i];
    total *= sizes[i];
  }
  hipLaunchKernelGGL(copy4d_kernel, dim3(nblk(total)), dim3(256), 0, ctx->stream, (const f16*)in, (f16*)out, c, L / 8);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_argmax_rows_i32(gn_ctx* ctx, const int32_t* x, int32_t* out, int32_t rows, int32_t cols) {
  GN_REQUIRE(ctx && x && out && rows > 0 && cols > 0, "gn_argmax_rows_i32: bad arguments");
  hipLaunchKernelGGL(argmax_rows_i32_kernel, dim3(nblk(rows, 64)), dim3(64), 0, ctx->stream, x, out, rows, cols);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_image_f16_to_u8(gn_ctx* ctx, const void* in, uint8_t* out, int64_t pixels, int32_t ld) {
  GN_REQUIRE(ctx && in && out && pixels > 0 && ld >= 3, "gn_image_f16_to_u8: bad arguments");
  hipLaunchKernelGGL(image_f16_to_u8_kernel, dim3(nblk(pixels)), dim3(256), 0, ctx->stream, (const f16*)in, out, (long)pixels, ld);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_add(gn_ctx* ctx, const void* a, const void* b, void* out, int64_t n) {
  GN_REQUIRE(ctx && a && b && out && n > 0 && n % 8 == 0, "gn_add: n must be a positive multiple of 8");
  hipLaunchKernelGGL(add_kernel, dim3(nblk(n / 8)), dim3(256), 0, ctx->stream, (const uint4*)a, (const uint4*)b, (uint4*)out, (long)(n / 8));
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_add_multi(gn_ctx* ctx, const void* const* a, const void* const* b, void* const* out, const int64_t* n, int32_t count) {
  GN_REQUIRE(ctx && a && b && out && n && count >= 1 && count <= GN_ADD_MULTI_MAX, "gn_add_multi: 1 .. %d tensors", GN_ADD_MULTI_MAX);
  AddMultiArgs p;
  long nmax = 0;
  for (int i = 0; i < count; ++i) {
    GN_REQUIRE(a[i] && b[i] && out[i] && n[i] > 0 && n[i] % 8 == 0, "gn_add_multi: tensor %d: n must be a positive multiple of 8", i);
    p.a[i] = (const uint4*)a[i]; p.b[i] = (const uint4*)b[i]; p.out[i] = (uint4*)out[i]; p.n8[i] = (long)(n[i] / 8);
    nmax = p.n8[i] > nmax ? p.n8[i] : nmax;
  }
  long blocks = (nmax + 255) / 256;
  if (blocks > 2048) blocks = 2048;  // (grid-stride: the largest tensor sets the x extent, smaller ones leave their tail blocks idle)
  hipLaunchKernelGGL(add_multi_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_add_multi_stats(gn_ctx* ctx, const void* const* a, const void* const* b, void* const* out, const int64_t* n, const int32_t* C,
                           const gn_stats_sink* sinks, int32_t count) {
  GN_REQUIRE(ctx && a && b && out && n && C && count >= 1 && count <= GN_ADD_MULTI_MAX, "gn_add_multi_stats: 1 .. %d tensors", GN_ADD_MULTI_MAX);
  AddMultiStatsArgs p;
  long rmax = 0;
  size_t lds = 0;
  for (int i = 0; i < count; ++i) {
    GN_REQUIRE(a[i] && b[i] && out[i] && n[i] > 0 && C[i] > 0 && C[i] % 8 == 0 && n[i] % C[i] == 0, "gn_add_multi_stats: tensor %d: [rows x C] with C %% 8 == 0", i);
    p.base.a[i] = (const uint4*)a[i]; p.base.b[i] = (const uint4*)b[i]; p.base.out[i] = (uint4*)out[i]; p.base.n8[i] = (long)(n[i] / 8);
    p.C[i] = C[i];
    gn_stats_sink none = {nullptr, 1, 0, 1, 1, 1, 1};
    const gn_stats_sink& s = sinks ? sinks[i] : none;
    p.sink[i] = gn_sink_params(s);
    const long rows = (long)(n[i] / C[i]);
    if (s.stats) {
      GN_REQUIRE(s.cpg > 0 && s.groups > 0 && s.coff >= 0 && s.rows_per_sample > 0 && rows % s.rows_per_sample == 0 && ((uintptr_t)s.stats & 7) == 0 &&
                     (int64_t)s.coff + C[i] <= (int64_t)s.cpg * s.groups && s.samples == rows / s.rows_per_sample && s.replicas >= 1,
                 "gn_add_multi_stats: tensor %d: bad sink", i);
      const int cc = C[i] / 8, tx = cc < 256 ? cc : 256;
      const size_t need = (size_t)2 * (256 / tx) * C[i] * sizeof(float);
      lds = need > lds ? need : lds;
    }
    rmax = rows > rmax ? rows : rmax;
  }
  GN_REQUIRE(lds <= 64 * 1024, "gn_add_multi_stats: C too large for the statistics slab");
  long blocks = (rmax + 31) / 32;  // >= 32 rows per block: at most 2 x groups atomics per 32 rows
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(add_multi_stats_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), lds, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_memset(gn_ctx* ctx, void* ptr, int64_t bytes) {
  GN_REQUIRE(ctx && ptr && bytes > 0, "gn_memset: null / empty");
  GN_HIP(hipMemsetAsync(ptr, 0, (size_t)bytes, ctx->stream));
  return GN_OK;
}

int32_t gn_film(gn_ctx* ctx, const void* x, void* out, const void* gamma, const void* beta, int64_t ld_film, int64_t rows_per_film,
                int64_t rows, int32_t C, int32_t act) {
  GN_REQUIRE(ctx && x && out && gamma && beta && rows > 0 && rows_per_film > 0 && C > 0 && C % 8 == 0 && ld_film % 8 == 0,
             "gn_film: C (%d) and ld_film must be multiples of 8", C);
  GN_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "gn_film: 16-byte alignment");
  const long n8 = rows * (C / 8);
  hipLaunchKernelGGL(film_kernel, dim3(nblk(n8)), dim3(256), 0, ctx->stream, (const uint4*)x, (uint4*)out, (const f16*)gamma, (const f16*)beta,
                     (long)ld_film, (long)rows_per_film, (long)rows, C / 8, act);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_act(gn_ctx* ctx, const void* x, void* out, int64_t n, int32_t act) {
  GN_REQUIRE(ctx && x && out && n > 0 && n % 8 == 0, "gn_act: n must be a positive multiple of 8");
  hipLaunchKernelGGL(act_kernel, dim3(nblk(n / 8)), dim3(256), 0, ctx->stream, (const uint4*)x, (uint4*)out, (long)(n / 8), act);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_embedding(gn_ctx* ctx, const int32_t* ids, const void* tok, const void* pos, void* out, int32_t B, int32_t L, int32_t D) {
  GN_REQUIRE(ctx && ids && tok && pos && out && B > 0 && L > 0 && D > 0 && D % 8 == 0, "gn_embedding: bad arguments");
  hipLaunchKernelGGL(embedding_kernel, dim3(nblk((long)B * L * (D / 8))), dim3(256), 0, ctx->stream, ids, (const f16*)tok, (const f16*)pos, (f16*)out, B, L, D);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_softmax_rows(gn_ctx* ctx, void* x, int64_t rows, int32_t cols, int32_t ld, float scale) {
  GN_REQUIRE(ctx && x && rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 64 * 8 * SM_MAXCH && ld % 8 == 0 && ld >= cols, "gn_softmax_rows: cols must be a multiple of 8, <= %d", 64 * 8 * SM_MAXCH);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(nblk(rows, 4)), dim3(256), 0, ctx->stream, (f16*)x, (long)rows, cols, ld, scale, cols);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_softmax_rows_masked(gn_ctx* ctx, void* x, int64_t rows, int32_t cols, int32_t ld, float scale, int32_t valid) {
  GN_REQUIRE(ctx && x && rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 64 * 8 * SM_MAXCH && ld % 8 == 0 && ld >= cols && valid > 0 && valid <= cols,
             "gn_softmax_rows_masked: cols must be a multiple of 8, <= %d, 0 < valid <= cols", 64 * 8 * SM_MAXCH);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(nblk(rows, 4)), dim3(256), 0, ctx->stream, (f16*)x, (long)rows, cols, ld, scale, valid);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_maxpool3x3s2(gn_ctx* ctx, const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C) {
  GN_REQUIRE(ctx && x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gn_maxpool3x3s2: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(nblk((long)B * Ho * Wo * (C / 8))), dim3(256), 0, ctx->stream, (const f16*)x, (f16*)y, B, H, W, C, Ho, Wo);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

}  // extern "C"
