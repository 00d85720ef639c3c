// Device-side train-time augmentation of the ControlNet trainer (SURVEY.md section 8 rows a12 "augment (:1321)" / a13;
// diffusion/train_controlnet_genima.py:775-830 with the README recipe `--augmentations=crop,colorjitter`):
//   * torchvision ColorJitter(brightness 0.2, contrast 0.2, saturation 0.1, hue 0.05) on the conditioning images -- the four
//     adjust_* ops in the drawn order with the drawn factors (one draw per batch tensor, as torchvision does for a batched call),
//   * reflect-pad by 2 + one random crop back to the resolution, shared by target and conditioning images.
// Images are NHWC f16 with 8-channel pixels (3 valid), values in [0, 1] (conditioning) or [-1, 1] (targets).  All colour math is
// f32 in registers; the only cross-pixel dependency is adjust_contrast's per-image grey mean, so the jitter is two passes: pass 1
// replays the ops that precede the contrast step and reduces the grey level per image (deterministic two-stage sum), pass 2
// replays the whole chain and writes the result.
#include "common.h"

namespace {

inline unsigned nblk(long n, int t = 256) { return (unsigned)((n + t - 1) / t); }

struct JitterP {
  int order[4];     // op ids in application order: 0 brightness, 1 contrast, 2 saturation, 3 hue (torchvision fn_idx)
  float factor[4];  // indexed by op id
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float grey(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

// torchvision.transforms._functional_tensor._rgb2hsv / _hsv2rgb / adjust_hue for float images
__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float hf) {
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.0f : maxc);
  const float div = eqc ? 1.0f : cr;
  const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
  float h = 0.0f;
  if (maxc == r) h = bc - gc;
  else if (maxc == g) h = 2.0f + rc - bc;
  else h = 4.0f + gc - rc;
  h = fmodf(h / 6.0f + 1.0f, 1.0f);
  h = h + hf;
  h = h - floorf(h);  // python-style (h + hue_factor) % 1.0
  const float v = maxc;
  const float h6 = h * 6.0f;
  const float fi = floorf(h6), f = h6 - fi;
  int i = (int)fi % 6;
  if (i < 0) i += 6;
  const float p = clamp01(v * (1.0f - s)), q = clamp01(v * (1.0f - s * f)), t = clamp01(v * (1.0f - s * (1.0f - f)));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// applies ops order[0 .. upto-1]; `mean` is used by the contrast op
__device__ __forceinline__ void apply_ops(float& r, float& g, float& b, const JitterP& p, int upto, float mean) {
  for (int k = 0; k < upto; ++k) {
    const int op = p.order[k];
    const float f = p.factor[op];
    if (op == 0) {
      r = clamp01(r * f); g = clamp01(g * f); b = clamp01(b * f);
    } else if (op == 1) {
      const float m = (1.0f - f) * mean;
      r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
    } else if (op == 2) {
      const float m = (1.0f - f) * grey(r, g, b);
      r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
    } else {
      hue_shift(r, g, b, f);
    }
  }
}

__global__ __launch_bounds__(256) void jitter_mean_kernel(const f16* __restrict__ x, float* __restrict__ part, long HW, int ld, JitterP p, int pre) {
  // grid (chunks, B): per-image partial sums of the grey level after the first `pre` ops
  __shared__ float red[4];
  const int b = blockIdx.y;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    const f16* px = x + ((long)b * HW + i) * ld;
    float r = (float)px[0], g = (float)px[1], bl = (float)px[2];
    apply_ops(r, g, bl, p, pre, 0.0f);
    s += grey(r, g, bl);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void jitter_mean_final_kernel(const float* __restrict__ part, float* __restrict__ mean, int chunks, long HW) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < chunks; ++i) s += (double)part[(long)b * chunks + i];
    mean[b] = (float)(s / (double)HW);
  }
}
__global__ __launch_bounds__(256) void jitter_apply_kernel(const f16* __restrict__ x, f16* __restrict__ out, const float* __restrict__ mean, long HW,
                                                           int ld, JitterP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= HW) return;
  const f16* px = x + ((long)b * HW + i) * ld;
  f16* po = out + ((long)b * HW + i) * ld;
  float r = (float)px[0], g = (float)px[1], bl = (float)px[2];
  apply_ops(r, g, bl, p, 4, mean ? mean[b] : 0.0f);
  po[0] = (f16)r; po[1] = (f16)g; po[2] = (f16)bl;
  for (int c = 3; c < ld; ++c) po[c] = (f16)0.0f;
}

// out[b, y, x, :] = in[b, refl(y + i - pad), refl(x + j - pad), :]   (F.pad(mode="reflect") + crop at (i, j))
__device__ __forceinline__ int reflect(int v, int n) {
  if (v < 0) v = -v;
  if (v >= n) v = 2 * n - 2 - v;
  return v;
}
__global__ void reflect_pad_crop_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int B, int H, int W, int C8, int pad, int ci, int cj) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W * C8) return;
  const int c = (int)(idx % C8);
  long r = idx / C8;
  const int ox = (int)(r % W); r /= W;
  const int oy = (int)(r % H);
  const int b = (int)(r / H);
  const int sy = reflect(oy + ci - pad, H), sx = reflect(ox + cj - pad, W);
  out[idx] = x[(((long)b * H + sy) * W + sx) * C8 + c];
}

}  // namespace

extern "C" {

int64_t gn_color_jitter_workspace_bytes(int32_t B) { return ((int64_t)B * 256 + B) * 4; }

/* order[k] = id of the k-th op (0 brightness, 1 contrast, 2 saturation, 3 hue; torchvision ColorJitter.get_params fn_idx);
 * factors[id] = that op's drawn factor.  x / out: [B, HW, ld] f16, channels 0..2 = RGB in [0, 1]; out may alias x. */
int32_t gn_color_jitter(gn_ctx* ctx, const void* x, void* out, int32_t B, int64_t HW, int32_t ld, const int32_t* order, const float* factors,
                        void* workspace) {
  GN_REQUIRE(ctx && x && out && order && factors && workspace && B > 0 && HW > 0 && ld >= 3, "gn_color_jitter: bad arguments");
  JitterP p;
  int seen = 0, pre = -1;
  for (int k = 0; k < 4; ++k) {
    GN_REQUIRE(order[k] >= 0 && order[k] < 4, "gn_color_jitter: op ids must be 0..3");
    seen |= 1 << order[k];
    p.order[k] = order[k];
    p.factor[k] = factors[k];
    if (order[k] == 1) pre = k;
  }
  GN_REQUIRE(seen == 15, "gn_color_jitter: order must be a permutation of 0..3");
  float* part = (float*)workspace;
  float* mean = part + (long)B * 256;
  const int chunks = 256;
  hipLaunchKernelGGL(jitter_mean_kernel, dim3(chunks, B), dim3(256), 0, ctx->stream, (const f16*)x, part, (long)HW, ld, p, pre);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(jitter_mean_final_kernel, dim3(B), dim3(64), 0, ctx->stream, (const float*)part, mean, chunks, (long)HW);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(jitter_apply_kernel, dim3(nblk(HW), B), dim3(256), 0, ctx->stream, (const f16*)x, (f16*)out, (const float*)mean, (long)HW, ld, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

/* reflect padding by `pad` on every side followed by an H x W crop at (crop_i, crop_j) of the padded image; C % 8 == 0 */
int32_t gn_reflect_pad_crop(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t crop_i,
                            int32_t crop_j) {
  GN_REQUIRE(ctx && x && out && x != out && B > 0 && H > pad && W > pad && C > 0 && C % 8 == 0 && pad >= 0, "gn_reflect_pad_crop: bad arguments");
  GN_REQUIRE(crop_i >= 0 && crop_i <= 2 * pad && crop_j >= 0 && crop_j <= 2 * pad, "gn_reflect_pad_crop: crop offset outside the padded image");
  hipLaunchKernelGGL(reflect_pad_crop_kernel, dim3(nblk((long)B * H * W * (C / 8))), dim3(256), 0, ctx->stream, (const uint4*)x, (uint4*)out, B, H,
                     W, C / 8, pad, crop_i, crop_j);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

}  // extern "C"
