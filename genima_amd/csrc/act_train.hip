// Training-side kernels of the ACT controller update (reference controller/method/genima_act.py:94-139
// ``calculate_loss``, :27-92 the CVAE branch of ``GenimaMVTransformer.forward``, :348-422 ``GenimaACT.update``).  All tensors here are
// small (actions [B, 20, 8], latents [B, 32], FiLM features [B, C]) or plain elementwise: HBM-bound, 16-byte accesses where it matters.
#include "common.h"

namespace {

__device__ __forceinline__ long gtid() { return (long)blockIdx.x * blockDim.x + threadIdx.x; }
inline unsigned nblk(long n, int t = 256) { return (unsigned)((n + t - 1) / t); }

// FiLM backward: z = (1 + gamma[b]) * x + beta[b], y = act(z) (NONE / RELU).  dz = dy * act'(z);  dx = dz * (1 + gamma[b]);
// optional dz / dz * x copies feed the per-(b, c) column sums that give dbeta / dgamma.
__global__ void film_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ x, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                long ld_film, long rows_per_film, long rows, int C8, int act, uint4* __restrict__ dx, uint4* __restrict__ dz_out,
                                uint4* __restrict__ dzx_out) {
  const long i = gtid();
  if (i >= rows * C8) return;
  const long r = i / C8;
  const int c = (int)(i - r * C8) * 8;
  const long b = r / rows_per_film;
  const uint4 rd = dy[i], rx = x[i];
  const uint4 rg = *reinterpret_cast<const uint4*>(gamma + b * ld_film + c), rb = *reinterpret_cast<const uint4*>(beta + b * ld_film + c);
  const f16x8 vd = *reinterpret_cast<const f16x8*>(&rd), vx = *reinterpret_cast<const f16x8*>(&rx);
  const f16x8 vg = *reinterpret_cast<const f16x8*>(&rg), vb = *reinterpret_cast<const f16x8*>(&rb);
  f16x8 odx, odz, odzx;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g1 = 1.0f + (float)vg[e];
    const float z = g1 * (float)vx[e] + (float)vb[e];
    const float dz = (act == GN_ACT_RELU && z <= 0.0f) ? 0.0f : (float)vd[e];
    odx[e] = (f16)(dz * g1);
    odz[e] = (f16)dz;
    odzx[e] = (f16)(dz * (float)vx[e]);
  }
  dx[i] = *reinterpret_cast<uint4*>(&odx);
  if (dz_out) dz_out[i] = *reinterpret_cast<uint4*>(&odz);
  if (dzx_out) dzx_out[i] = *reinterpret_cast<uint4*>(&odzx);
}

// inverted dropout with a caller-drawn keep mask (1 byte per element): out = x * mask * scale; the backward is the same op on dy
__global__ void dropout_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ mask, f16* __restrict__ out, long n, float scale) {
  const long i = gtid();
  if (i < n) out[i] = mask[i] ? (f16)((float)x[i] * scale) : (f16)0.0f;
}

// z = mu + exp(logvar / 2) * eps  (reparametrize, genima_act.py:64-68);  info = [mu | logvar] rows of stride ld
__global__ void cvae_sample_kernel(const f16* __restrict__ info, long ld, const float* __restrict__ eps, f16* __restrict__ z, long ldz, int B, int L) {
  const long i = gtid();
  if (i >= (long)B * L) return;
  const int b = (int)(i / L), j = (int)(i - (long)b * L);
  const float mu = (float)info[b * ld + j], lv = (float)info[b * ld + L + j];
  z[b * ldz + j] = (f16)(mu + __expf(0.5f * lv) * eps[i]);
}
// d_info = d(KL term) + d(reparametrize):  dmu = dz + kl_scale * mu;   dlogvar = dz * eps * exp(lv / 2) / 2 + kl_scale * (exp(lv) - 1) / 2
// (KL = mean_b sum_j -(1 + lv - mu^2 - exp(lv)) / 2; kl_scale = loss_scale * kl_weight / B)
__global__ void cvae_bwd_kernel(const f16* __restrict__ info, long ld, const float* __restrict__ eps, const f16* __restrict__ dz, long ldz,
                                f16* __restrict__ dinfo, int B, int L, float kl_scale) {
  const long i = gtid();
  if (i >= (long)B * L) return;
  const int b = (int)(i / L), j = (int)(i - (long)b * L);
  const float mu = (float)info[b * ld + j], lv = (float)info[b * ld + L + j];
  const float g = (float)dz[b * ldz + j], s = __expf(0.5f * lv);
  dinfo[b * ld + j] = (f16)(g + kl_scale * mu);
  dinfo[b * ld + L + j] = (f16)(0.5f * g * eps[i] * s + 0.5f * kl_scale * (s * s - 1.0f));
}

// calculate_loss (genima_act.py:115-139) on one block (B * T * A <= a few thousand elements; a fixed summation order):
//   l1 = mean over [B, T, A-1] of |a - a_hat| * !pad;   grip = mean over [B, T] of 0.05 * BCEWithLogits(a_hat[..., A-1], a[..., A-1]) * !pad;
//   kl = mean_b sum_j -(1 + lv - mu^2 - exp(lv)) / 2;   loss = l1 + grip + kl * kl_weight.
// out[0..3] = (loss, l1, grip, kl); d_a_hat = grad_scale * d(l1 + grip) / d(a_hat) (rows >= T and columns >= A: 0).
__global__ __launch_bounds__(256) void act_loss_kernel(const f16* __restrict__ a_hat, long ld_hat, long bs_hat, const float* __restrict__ actions,
                                                       const uint8_t* __restrict__ is_pad, const f16* __restrict__ info, long ld_info, int B, int T,
                                                       int Tp, int A, int L, float kl_weight, float grad_scale, float* __restrict__ out,
                                                       f16* __restrict__ d_a_hat) {
  __shared__ float red[3][256];
  float l1 = 0.f, gr = 0.f, kl = 0.f;
  const float inv_l1 = 1.0f / (float)((long)B * T * (A - 1)), inv_g = 1.0f / (float)((long)B * T);
  for (long i = threadIdx.x; i < (long)B * Tp * ld_hat; i += 256) {
    const int b = (int)(i / (Tp * ld_hat));
    const long r = i - (long)b * Tp * ld_hat;
    const int t = (int)(r / ld_hat), c = (int)(r - (long)t * ld_hat);
    float g = 0.0f;
    if (t < T && c < A) {
      const float keep = (is_pad && is_pad[b * T + t]) ? 0.0f : 1.0f;
      const float x = (float)a_hat[b * bs_hat + t * ld_hat + c], y = actions[((long)b * T + t) * A + c];
      if (c < A - 1) {
        const float d = x - y;
        l1 += fabsf(d) * keep;
        g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * keep * inv_l1;
      } else {  // BCEWithLogits(x, y) = max(x, 0) - x y + log(1 + exp(-|x|))
        gr += 0.05f * (fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)))) * keep;
        g = 0.05f * (1.0f / (1.0f + __expf(-x)) - y) * keep * inv_g;
      }
    }
    d_a_hat[b * bs_hat + t * ld_hat + c] = (f16)(g * grad_scale);
  }
  if (info)
    for (long i = threadIdx.x; i < (long)B * L; i += 256) {
      const int b = (int)(i / L), j = (int)(i - (long)b * L);
      const float mu = (float)info[b * ld_info + j], lv = (float)info[b * ld_info + L + j];
      kl += -0.5f * (1.0f + lv - mu * mu - __expf(lv));
    }
  red[0][threadIdx.x] = l1; red[1][threadIdx.x] = gr; red[2][threadIdx.x] = kl;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float a = red[0][0] * inv_l1, g = red[1][0] * inv_g, k = red[2][0] / (float)B;
    out[0] = a + g + k * kl_weight; out[1] = a; out[2] = g; out[3] = k;
  }
}

// ElasticTransform's warp (torchvision v2.ElasticTransform -> grid_sample bilinear, zero fill): out[b, y, x, :] = bilinear(in[b], x + dx, y + dy)
// with ONE displacement field (pixels) shared by every image of the batch, as when the transform is called on a batched tensor.
__global__ void warp_bilinear_kernel(const f16* __restrict__ in, f16* __restrict__ out, const float* __restrict__ disp, int B, int H, int W, int C8) {
  const long i = gtid();
  if (i >= (long)B * H * W * C8) return;
  const int c = (int)(i % C8) * 8;
  long r = i / C8;
  const int x = (int)(r % W);
  r /= W;
  const int y = (int)(r % H), b = (int)(r / H);
  const float sx = (float)x + disp[((long)y * W + x) * 2], sy = (float)y + disp[((long)y * W + x) * 2 + 1];
  const float fx = floorf(sx), fy = floorf(sy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float ax = sx - fx, ay = sy - fy;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
    const float w = ((t & 1) ? ax : 1.0f - ax) * ((t >> 1) ? ay : 1.0f - ay);
    if (xx >= 0 && xx < W && yy >= 0 && yy < H && w != 0.0f) {
      const uint4 rv = *reinterpret_cast<const uint4*>(in + (((long)b * H + yy) * W + xx) * (C8 * 8) + c);
      const f16x8 v = *reinterpret_cast<const f16x8*>(&rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += w * (float)v[e];
    }
  }
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)acc[e];
  *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<uint4*>(&o);
}

// f32 -> f16 add of per-(b, c) sums into a feature gradient  y[i] += a * x[i] is not needed; what IS needed: f32 -> f16 add of per-(b, c) sums into a feature gradient
__global__ void add_f32_to_f16_kernel(const float* __restrict__ src, long ld_src, f16* __restrict__ dst, long ld_dst, int B, int C) {
  const long i = gtid();
  if (i >= (long)B * C) return;
  const int b = (int)(i / C), c = (int)(i - (long)b * C);
  dst[b * ld_dst + c] = (f16)((float)dst[b * ld_dst + c] + src[b * ld_src + c]);
}

}  // namespace

extern "C" {

int32_t gn_film_bwd(gn_ctx* ctx, const void* dy, const void* x, const void* gamma, const void* beta, int64_t ld_film, int64_t rows_per_film,
                    int64_t rows, int32_t C, int32_t act, void* dx, void* dz, void* dzx) {
  GN_REQUIRE(ctx && dy && x && gamma && beta && dx && rows > 0 && rows_per_film > 0 && C > 0 && C % 8 == 0 && ld_film % 8 == 0,
             "gn_film_bwd: bad arguments (C %% 8, ld_film %% 8)");
  GN_REQUIRE(act == GN_ACT_NONE || act == GN_ACT_RELU, "gn_film_bwd: act must be NONE or RELU");
  const long n8 = rows * (C / 8);
  hipLaunchKernelGGL(film_bwd_kernel, dim3(nblk(n8)), dim3(256), 0, ctx->stream, (const uint4*)dy, (const uint4*)x, (const f16*)gamma, (const f16*)beta,
                     (long)ld_film, (long)rows_per_film, (long)rows, C / 8, act, (uint4*)dx, (uint4*)dz, (uint4*)dzx);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_dropout(gn_ctx* ctx, const void* x, const uint8_t* keep_mask, void* out, int64_t n, float scale) {
  GN_REQUIRE(ctx && x && keep_mask && out && n > 0, "gn_dropout: bad arguments");
  hipLaunchKernelGGL(dropout_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, (const f16*)x, keep_mask, (f16*)out, (long)n, scale);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_cvae_sample(gn_ctx* ctx, const void* info, int64_t ld_info, const float* eps, void* z, int64_t ld_z, int32_t B, int32_t L) {
  GN_REQUIRE(ctx && info && eps && z && B > 0 && L > 0 && ld_info >= 2 * L && ld_z >= L, "gn_cvae_sample: bad arguments");
  hipLaunchKernelGGL(cvae_sample_kernel, dim3(nblk((long)B * L)), dim3(256), 0, ctx->stream, (const f16*)info, (long)ld_info, eps, (f16*)z, (long)ld_z, B, L);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_cvae_bwd(gn_ctx* ctx, const void* info, int64_t ld_info, const float* eps, const void* dz, int64_t ld_z, void* dinfo, int32_t B, int32_t L,
                    float kl_scale) {
  GN_REQUIRE(ctx && info && eps && dz && dinfo && B > 0 && L > 0, "gn_cvae_bwd: bad arguments");
  hipLaunchKernelGGL(cvae_bwd_kernel, dim3(nblk((long)B * L)), dim3(256), 0, ctx->stream, (const f16*)info, (long)ld_info, eps, (const f16*)dz, (long)ld_z,
                     (f16*)dinfo, B, L, kl_scale);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_act_loss(gn_ctx* ctx, const void* a_hat, int64_t ld_hat, int64_t bs_hat, const float* actions, const uint8_t* is_pad, const void* info,
                    int64_t ld_info, int32_t B, int32_t T, int32_t T_rows, int32_t A, int32_t L, float kl_weight, float grad_scale, float* out4,
                    void* d_a_hat) {
  GN_REQUIRE(ctx && a_hat && actions && out4 && d_a_hat && B > 0 && T > 0 && T_rows >= T && A >= 2 && ld_hat >= A && bs_hat >= (int64_t)T_rows * ld_hat,
             "gn_act_loss: bad arguments");
  hipLaunchKernelGGL(act_loss_kernel, dim3(1), dim3(256), 0, ctx->stream, (const f16*)a_hat, (long)ld_hat, (long)bs_hat, actions, is_pad, (const f16*)info,
                     (long)ld_info, B, T, T_rows, A, L, kl_weight, grad_scale, out4, (f16*)d_a_hat);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_warp_bilinear(gn_ctx* ctx, const void* in, void* out, const float* disp, int32_t B, int32_t H, int32_t W, int32_t C) {
  GN_REQUIRE(ctx && in && out && disp && in != out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gn_warp_bilinear: bad arguments (C %% 8, no aliasing)");
  const long n = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(warp_bilinear_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, (const f16*)in, (f16*)out, disp, B, H, W, C / 8);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_add_f32_to_f16(gn_ctx* ctx, const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t B, int32_t C) {
  GN_REQUIRE(ctx && src && dst && B > 0 && C > 0, "gn_add_f32_to_f16: bad arguments");
  hipLaunchKernelGGL(add_f32_to_f16_kernel, dim3(nblk((long)B * C)), dim3(256), 0, ctx->stream, src, (long)ld_src, (f16*)dst, (long)ld_dst, B, C);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

}  // extern "C"
