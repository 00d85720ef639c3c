// Weight repacking under the C ABI ("an explicit gn_pack_* call producing a caller-owned buffer"): what
// genima_amd/packing.py does with torch ops for the Python host, for hosts that hand over a diffusers state dict as raw device arrays.
//   * gn_pack_conv_weight   OIHW (f32 or f16) -> [Cout_pad8][KH * KW * Cin_pad8] f16, the implicit-GEMM operand of gn_gemm(conv)
//   * gn_pack_geglu_rows    [2H, K] -> alternating 32-row blocks hidden | gate, the row order GN_ACT_GEGLU reads (weight and bias)
//   * gn_pack_fold_layernorm  W, gamma, beta, b -> W * gamma (f16), its f32 row sums c1, c2 = W beta + b: the operands of gn_gemm_desc::ln_c1
// Run once per checkpoint load: plain coalesced elementwise kernels, nothing to tune.
#include "common.h"

namespace {

inline unsigned pk_nblk(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

template <typename T>
__global__ __launch_bounds__(256) void pack_conv_kernel(const T* __restrict__ src, f16* __restrict__ dst, int O, int I, int KH, int KW, int Op, int Ip) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)Op * KH * KW * Ip;
  if (idx >= total) return;
  const int c = (int)(idx % Ip);
  long r = idx / Ip;
  const int kw = (int)(r % KW); r /= KW;
  const int kh = (int)(r % KH);
  const int o = (int)(r / KH);
  float v = 0.0f;
  if (o < O && c < I) v = (float)src[(((long)o * I + c) * KH + kh) * KW + kw];
  dst[idx] = (f16)v;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_geglu_kernel(const T* __restrict__ src, f16* __restrict__ dst, int H, long K) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2l * H * K) return;
  const long r = idx / K, k = idx - r * K;
  const long blk = r >> 6, within = r & 63;
  const long srow = within < 32 ? blk * 32 + within : (long)H + blk * 32 + (within - 32);
  dst[idx] = (f16)(float)src[srow * K + k];
}

// one wave per output row
__global__ __launch_bounds__(256) void fold_ln_kernel(const f16* __restrict__ w, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                      const f16* __restrict__ bias, f16* __restrict__ wg, float* __restrict__ c1, f16* __restrict__ c2,
                                                      int N, int K, long ldw) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  float s1 = 0.0f, s2 = 0.0f;
  for (int k = lane; k < K; k += 64) {
    const float wv = (float)w[(long)row * ldw + k];
    const f16 g = (f16)(wv * (float)gamma[k]);
    wg[(long)row * ldw + k] = g;
    s1 += (float)g;                 // summed from the ROUNDED value: the numbers the MFMA will multiply
    s2 += wv * (float)beta[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (lane == 0) {
    c1[row] = s1;
    c2[row] = (f16)(s2 + (bias ? (float)bias[row] : 0.0f));
  }
}

}  // namespace

extern "C" {

int32_t gn_pack_conv_weight(gn_ctx* ctx, const void* src_oihw, int32_t src_f16, void* dst, int32_t O, int32_t I, int32_t KH, int32_t KW) {
  GN_REQUIRE(ctx && src_oihw && dst && O > 0 && I > 0 && KH > 0 && KW > 0, "gn_pack_conv_weight: bad arguments");
  const int Op = (O + 7) / 8 * 8, Ip = (I + 7) / 8 * 8;
  const long total = (long)Op * KH * KW * Ip;
  if (src_f16) hipLaunchKernelGGL(pack_conv_kernel<f16>, dim3(pk_nblk(total)), dim3(256), 0, ctx->stream, (const f16*)src_oihw, (f16*)dst, O, I, KH, KW, Op, Ip);
  else hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(pk_nblk(total)), dim3(256), 0, ctx->stream, (const float*)src_oihw, (f16*)dst, O, I, KH, KW, Op, Ip);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_pack_geglu_rows(gn_ctx* ctx, const void* src, int32_t src_f16, void* dst, int32_t H, int64_t K) {
  GN_REQUIRE(ctx && src && dst && H > 0 && H % 32 == 0 && K > 0, "gn_pack_geglu_rows: H (%d) must be a positive multiple of 32", H);
  const long total = 2l * H * K;
  if (src_f16) hipLaunchKernelGGL(pack_geglu_kernel<f16>, dim3(pk_nblk(total)), dim3(256), 0, ctx->stream, (const f16*)src, (f16*)dst, H, (long)K);
  else hipLaunchKernelGGL(pack_geglu_kernel<float>, dim3(pk_nblk(total)), dim3(256), 0, ctx->stream, (const float*)src, (f16*)dst, H, (long)K);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_pack_fold_layernorm(gn_ctx* ctx, const void* w, const void* gamma, const void* beta, const void* bias, void* ln_weight, float* ln_c1,
                               void* ln_c2, int32_t N, int32_t K, int64_t ldw) {
  GN_REQUIRE(ctx && w && gamma && beta && ln_weight && ln_c1 && ln_c2 && N > 0 && K > 0 && ldw >= K, "gn_pack_fold_layernorm: bad arguments");
  hipLaunchKernelGGL(fold_ln_kernel, dim3((N + 3) / 4), dim3(256), 0, ctx->stream, (const f16*)w, (const f16*)gamma, (const f16*)beta, (const f16*)bias,
                     (f16*)ln_weight, ln_c1, (f16*)ln_c2, N, K, (long)ldw);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

}  // extern "C"
