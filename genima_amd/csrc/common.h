// Internal helpers shared by the libgenima_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/genima_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct gn_ctx {
  int device;
  hipStream_t stream;
};

void gn_set_error(const char* fmt, ...);

#define GN_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      gn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return GN_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

#define GN_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      gn_set_error(__VA_ARGS__);       \
      return GN_ERR_INVALID;           \
    }                                  \
  } while (0)

#define GN_LAUNCH_CHECK()                                                                  \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess) {                                                                \
      gn_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return GN_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// "done once per DEVICE" flags for per-function attributes (hipFuncSetAttribute applies to the current device's copy of the code object: a
// process-wide bool would leave a second device without the attribute -- ADVICE r5).  One slot array per call site.
struct GnOncePerDevice {
  bool done[64] = {};
  bool first() {  // true the first time the current device asks
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};

// ---- device math ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case GN_ACT_SILU: return act_silu(x);
    case GN_ACT_GELU: return act_gelu(x);
    case GN_ACT_QUICK_GELU: return act_quick_gelu(x);
    case GN_ACT_RELU: return fmaxf(x, 0.0f);
    case GN_ACT_TANH3: return 3.0f * tanhf(x * (1.0f / 3.0f));
    default: return x;
  }
}

// Whole-wave reductions in the VALU: four DPP steps fold each 16-lane row (quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror), four v_readlane pick up the row totals.  (The generic __shfl_xor butterfly is six ds_bpermute
// round trips through the LDS unit, ~400 cycles of latency per reduction: LayerNorm 32768 x 320 went 24.2 -> 11.4 us.)
// Every lane gets the result; ALL 64 LANES MUST BE ACTIVE at the call (every call site reduces under wave-uniform control flow).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  const int b = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const int b = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}

// LDS tile layout shared by the GEMM and attention kernels: rows of ROWB bytes split into 16-byte chunks, chunk index
// XOR-swizzled by the row so that the four 16-lane groups of a ds_read_b128 MFMA-fragment read (rows = lane&31, one chunk
// column) hit 16 distinct 16-byte slots of the 256-byte bank row (conflict-free; MI355X_MICROARCH.md LDS table).
template <int ROWB>
__device__ __forceinline__ int lds_swz(int row, int chunk) {
  if constexpr (ROWB == 128) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  } else if constexpr (ROWB == 64) {
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
  } else {
    return row * ROWB + (chunk << 4);
  }
}

// internal launchers (defined in the .hip files; take validated public descriptors)
int32_t gn_launch_gemm(gn_ctx* ctx, const gn_gemm_desc* d);
int32_t gn_ppp_pool_init(int device);  // gemm_ppp.hip: the zero-initialised hand-off flag pool of a device (allocated with the first context)
int32_t gn_launch_attention(gn_ctx* ctx, const gn_attn_desc* d);
int32_t gn_launch_groupnorm(gn_ctx* ctx, const gn_groupnorm_desc* d);
int32_t gn_launch_tblock(gn_ctx* ctx, const gn_tblock_desc* d);
