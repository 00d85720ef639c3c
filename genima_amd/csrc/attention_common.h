// Declarations shared by the attention forward kernels (attention.hip: generic kernel, attention_stream.hip: the branch-free D = 64 self-attention kernel).
#pragma once
#include "common.h"

constexpr int KT = 64;       // keys per tile
constexpr float PLIM = 8192.0f;  // a lane's 32-key sum of P beyond this means some P > 2^8: re-reference the row

struct AttnParams {
  const f16* q; const f16* k; const f16* vt; f16* o;
  long q_bs, k_bs, vt_bs, o_bs;
  int q_rs, k_rs, vt_rs, o_rs;
  int heads, Nq, Nk, causal;
  float scale_log2;  // scale * log2(e)
  float* lse;        // optional [B][heads][Nq]: m + log2(l), so that P = exp2(s * scale_log2 - lse) (training)
};

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;

// max / sum over the lane pair {l, l ^ 32} that shares a query row: gfx950's v_permlane32_swap exchanges the wave's halves in the
// VALU (the generic __shfl_xor lowers to ds_bpermute: an LDS round trip on the critical path of every key tile)
__device__ __forceinline__ float pair_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// D = 64, non-causal, Nk a multiple of 64 (>= 128), V^T layout: the software-pipelined, branch-free kernel (attention_stream.hip)
void gn_launch_attention_stream(const AttnParams& p, int B, hipStream_t stream);
// D = 64, non-causal, Nk a multiple of 64 (>= 128), V^T layout, large grids: one wave per SIMD, 64 query rows per wave (attention_pwg.hip)
void gn_launch_attention_pwg(const AttnParams& p, int B, hipStream_t stream);
