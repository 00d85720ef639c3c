// fp8 (OCP e4m3) quantisation for the fp8 Linear of gemm.hip (SURVEY.md section 8 a15 / BASELINE configs[4] "fp8 MFMA").
// Row-wise dynamic scaling: scale[r] = amax(|x[r, :]|) / 448, q[r, k] = e4m3_rne(x[r, k] * (448 / amax)).  Used for activations
// (rows = tokens: per-token scales) and, once per weight, for W [N, K] (rows = output channels: per-channel scales).  HBM-bound:
// the row is read twice (the second read hits L2) and written once at half the bytes.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const f16* x, long ldx, long rows, int K, int Kp, unsigned char* q,
                                                                long ldq, float* scales) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const f16* xr = x + row * ldx;
  float amax = 0.0f;
  for (int c = lane; c < K / 8; c += 64) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + c * 8);
    const f16x8 h = *reinterpret_cast<const f16x8*>(&v);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf((float)h[i]));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  // correctly rounded divisions and an uncontracted product: a 1-ulp slip in 448 / amax moves values across e4m3 rounding ties
  // (about once per 10^4 elements for f16 inputs), and the bytes are compared bit for bit with the oracle
  const float scale = amax > 0.0f ? __fdiv_rn(amax, 448.0f) : 1.0f;
  const float inv = amax > 0.0f ? __fdiv_rn(448.0f, amax) : 0.0f;
  if (lane == 0) scales[row] = scale;
  unsigned char* qr = q + row * ldq;
  for (int c = lane; c < Kp / 8; c += 64) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c * 8 < K) v = *reinterpret_cast<const uint4*>(xr + c * 8);
    const f16x8 h = *reinterpret_cast<const f16x8*>(&v);
    int lo = 0, up = 0;
    float e[8];  // v_cvt_pk_fp8_f32 rounds to nearest even like torch's float8_e4m3fn cast (tests/test_fp8_gpu.py: bit-exact bytes)
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __fmul_rn((float)h[i], inv);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], lo, true);
    up = __builtin_amdgcn_cvt_pk_fp8_f32(e[4], e[5], up, false);
    up = __builtin_amdgcn_cvt_pk_fp8_f32(e[6], e[7], up, true);
    *reinterpret_cast<uint2*>(qr + c * 8) = make_uint2((unsigned)lo, (unsigned)up);
  }
}

}  // namespace

extern "C" int32_t gn_quantize_fp8_rows(gn_ctx* ctx, const void* x, int64_t ldx, int64_t rows, int32_t K, void* q, int64_t ldq,
                                        void* scales) {
  GN_REQUIRE(ctx && x && q && scales && rows > 0 && K > 0, "gn_quantize_fp8_rows: bad arguments");
  GN_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldx >= K && ((uintptr_t)x & 15) == 0, "gn_quantize_fp8_rows: K and ldx must be multiples of 8 (16-byte f16 chunks)");
  const int Kp = (K + 15) / 16 * 16;
  GN_REQUIRE(ldq % 16 == 0 && ldq >= Kp && ((uintptr_t)q & 15) == 0, "gn_quantize_fp8_rows: ldq must be a multiple of 16 and >= round_up(K, 16)");
  hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, (const f16*)x, (long)ldx,
                     (long)rows, K, Kp, (unsigned char*)q, (long)ldq, (float*)scales);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
