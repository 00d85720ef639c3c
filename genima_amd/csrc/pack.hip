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


// ---- weight tapes of the fused transformer-block chains (csrc/tblock.hip): one thread per 16-byte chunk of the tape ------------------------
struct TapeSrc {
  const f16* w_a; const f16* b_a;
  const f16* w_ln; const float* c1; const f16* c2;
  const f16* w2; const f16* b2;
  const f16* w_p; const f16* b_p;
  int kind, C, nslots;
  long nchunks;
};

// 16 bytes of the LDS image of an [rows x 32] sub-tile (64-byte rows, chunk ^ ((row >> 2) & 3): csrc/common.h lds_swz<64>): physical chunk
// `pc` of row `row` holds the logical chunk pc ^ ((row >> 2) & 3) of m[row][k0 .. k0 + 32)
__device__ __forceinline__ uint4 image_chunk(const f16* m, long ld, int row, int pc, int k0) {
  const int lc = pc ^ ((row >> 2) & 3);
  return *reinterpret_cast<const uint4*>(m + (long)row * ld + k0 + lc * 8);
}

__global__ __launch_bounds__(256) void pack_tblock_tape_kernel(const TapeSrc s, uint4* __restrict__ tape) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= s.nchunks) return;
  const int C = s.C;
  const long slot = q / 1280;
  const int w = (int)(q - slot * 1280);
  uint4 v = make_uint4(0, 0, 0, 0);
  auto nc = [&](const f16* m, long ld, int k0) { return image_chunk(m, ld, w >> 2, w & 3, k0); };
  if (slot < s.nslots) {
    const int sl = (int)slot;
    if (sl < 10) {
      v = nc(s.w_a, C, 32 * sl);
    } else if (s.kind == GN_TBLOCK_FRONT) {
      const int g = (sl - 10) / 10, j = (sl - 10) % 10;
      v = nc(s.w_ln + (long)g * C * C, C, 32 * j);
    } else if (s.kind == GN_TBLOCK_MID) {
      v = nc(s.w_ln, C, 32 * (sl - 10));
    } else if (sl >= 150) {
      v = nc(s.w_p, C, 32 * (sl - 150));
    } else {
      const int t = sl - 10, ch = t / 7, j = t % 7;
      if (j >= 5) {
        v = nc(s.w2, 4l * C, 64 * ch + 32 * (j - 5));
      } else if (w < 1024) {  // two [128 x 32] pieces of the chunk's packed GEGLU rows
        const int sub = w >> 9, r = (w & 511) >> 2;
        v = image_chunk(s.w_ln + (long)(128 * ch) * C, C, r, w & 3, 64 * j + 32 * sub);
      } else if (j == 4 && w < 1024 + 32) {  // the chunk's c1 (f32 [128]) ...
        v = *reinterpret_cast<const uint4*>(s.c1 + 128 * ch + (w - 1024) * 4);
      } else if (j == 4 && w < 1024 + 48) {  // ... and c2 (f16 [128])
        v = *reinterpret_cast<const uint4*>(s.c2 + 128 * ch + (w - 1056) * 8);
      }
    }
  } else {  // the vector block behind the last slot
    const int b = (int)((q - (long)s.nslots * 1280) * 16);
    const int nb = C * 2;  // bytes of an f16 [C] vector
    auto from = [&](const void* p, int off) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p) + off); };
    if (b < nb) {
      v = from(s.b_a, b);
    } else if (s.kind == GN_TBLOCK_TAIL) {
      if (b < 2 * nb) v = from(s.b2, b - nb);
      else if (b < 3 * nb) v = from(s.b_p, b - 2 * nb);
    } else {
      const int ncol = s.kind == GN_TBLOCK_FRONT ? 3 * C : C;  // columns of the folded Linear
      if (b < nb + ncol * 4) v = from(s.c1, b - nb);
      else if (b < nb + ncol * 6) v = from(s.c2, b - nb - ncol * 4);
    }
  }
  tape[q] = v;
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

extern "C" int32_t gn_pack_tblock_tape(gn_ctx* ctx, const gn_tblock_tape_src* d, void* tape, int64_t tape_bytes) {
  GN_REQUIRE(ctx && d && tape, "gn_pack_tblock_tape: null argument");
  GN_REQUIRE(tape_bytes > 0 && tape_bytes == gn_tblock_tape_bytes(d->kind, d->C), "gn_pack_tblock_tape: tape_bytes %ld != gn_tblock_tape_bytes(%d, %d) = %ld",
             (long)tape_bytes, d->kind, d->C, (long)gn_tblock_tape_bytes(d->kind, d->C));
  GN_REQUIRE(d->w_a && d->b_a && d->w_ln && d->c1 && d->c2, "gn_pack_tblock_tape: w_a / b_a / w_ln / c1 / c2 are required");
  if (d->kind == GN_TBLOCK_TAIL) GN_REQUIRE(d->w2 && d->b2 && d->w_p && d->b_p, "gn_pack_tblock_tape(tail): w2 / b2 / w_p / b_p are required");
  const void* ptrs[] = {d->w_a, d->b_a, d->w_ln, d->c1, d->c2, d->w2, d->b2, d->w_p, d->b_p, tape};
  for (const void* q : ptrs) GN_REQUIRE(((uintptr_t)q & 15) == 0, "gn_pack_tblock_tape: every operand must be 16-byte aligned");
  TapeSrc s;
  s.w_a = (const f16*)d->w_a; s.b_a = (const f16*)d->b_a; s.w_ln = (const f16*)d->w_ln; s.c1 = d->c1; s.c2 = (const f16*)d->c2;
  s.w2 = (const f16*)d->w2; s.b2 = (const f16*)d->b2; s.w_p = (const f16*)d->w_p; s.b_p = (const f16*)d->b_p;
  s.kind = d->kind; s.C = d->C;
  s.nslots = d->kind == GN_TBLOCK_TAIL ? 160 : (d->kind == GN_TBLOCK_MID ? 20 : 40);
  s.nchunks = tape_bytes / 16;
  hipLaunchKernelGGL(pack_tblock_tape_kernel, dim3(pk_nblk(s.nchunks)), dim3(256), 0, ctx->stream, s, (uint4*)tape);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
