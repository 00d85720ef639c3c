// The GroupNorm bridge (include/genima_hip.h: gn_stats_sink / gn_norm_in): device helpers shared by the producers (GEMM epilogue tails,
// split-K reduce, add_multi) and the consumers (the ring GEMM's normalising A path, the apply-from-statistics kernel).
//
// diffusers runs  conv -> GroupNorm -> SiLU -> conv  as separate passes (ResnetBlock2D inside `self.pipe(...)`,
// controller/agent/sd_controlnet_agent.py:67-76).  GroupNorm needs the statistics of the WHOLE (sample, group) slab before the first
// element can be normalised, so it cannot live in one producer tile -- but the sums can: every producer workgroup adds the sum / sum of
// squares of the f16 values it just stored, and the consumer (which starts after the producer's launch has ended) turns the totals into
// scale / shift.  Totals are FIXED POINT (value * 2^24 in an int64, device-scope integer atomic adds): integer addition commutes, so the
// result does not depend on the arrival order of the workgroups and the whole call stays bit-reproducible (float atomics would not be).
// Range: |sum| * 2^24 < 2^63 holds for sums below 5.5e11 -- a (sample, group) slab of at most 327 680 f16 values with an rms below 1 300.
#pragma once
#include "common.h"

namespace {

typedef unsigned gn_u32x4 __attribute__((ext_vector_type(4)));

struct GnSinkP { unsigned long long* stats; int cpg, coff, groups, rps, nb, reps; };
struct GnInP { const long long* stats; const f16* gamma; const f16* beta; float eps; int groups, cpg, act, rps, nb, reps; };

__host__ __device__ __forceinline__ GnSinkP gn_sink_params(const gn_stats_sink& s) {
  GnSinkP o;
  o.stats = (unsigned long long*)s.stats; o.cpg = s.cpg; o.coff = s.coff; o.groups = s.groups; o.rps = s.rows_per_sample;
  o.nb = s.samples; o.reps = s.replicas > 0 ? s.replicas : 1;
  return o;
}
// the 128-byte line of (replica, sample, group)
__device__ __forceinline__ long gn_stats_line(int rep, int b, int g, int nb, int groups) { return (((long)rep * nb + b) * groups + g) * GN_STATS_LINE; }

// Fixed point: the SUM word carries 2^GN_STATS_SHIFT (24), the SUM-OF-SQUARES word 2^GN_STATS_SHIFT_SQ (12).  The coarser scale of the second word is its
// range: a (sample, group) slab's sum of squares may reach 2.2e15 before the int64 wraps (2^24 wrapped silently at 5.5e11 -- a few hundred f16 values near
// 65504); its quantum of 2.4e-4 per contribution is far below a variance's resolution.  Each contribution is clamped as well.
__device__ __forceinline__ unsigned long long gn_fixed(float s, double scale) {
  double d = (double)s * scale;
  d = fmin(fmax(d, -4.6e18), 4.6e18);
  return (unsigned long long)__double2ll_rn(d);  // two's complement: the unsigned atomic add IS the signed add
}
__device__ __forceinline__ void gn_stats_add(unsigned long long* slot, float sum, float sumsq) {
#ifdef GN_SINK_NO_ATOMICS
  return;
#endif
  (void)__hip_atomic_fetch_add(slot, gn_fixed(sum, (double)(1ll << GN_STATS_SHIFT)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  (void)__hip_atomic_fetch_add(slot + 1, gn_fixed(sumsq, (double)(1ll << GN_STATS_SHIFT_SQ)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (mean, rstd) of group g of sample b; inv_count = 1 / (elements of a (sample, group) slab).  f64 like gn_finalize_kernel (norm.hip).
__device__ __forceinline__ void gn_group_mean_rstd(const long long* stats, int b, int g, int nb, int groups, int reps, double inv_count, float eps,
                                                   float& mean, float& rstd) {
  long long s0 = 0, s1 = 0;
  for (int r0 = 0; r0 < reps; r0 += 8) {  // (replicas <= 8 in practice: every load of a trip in flight at once)
    long long v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v0[u] = 0; v1[u] = 0;
      if (r0 + u < reps) {
        const long long* s = stats + gn_stats_line(r0 + u, b, g, nb, groups);
        v0[u] = s[0];
        v1[u] = s[1];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += v0[u]; s1 += v1[u]; }  // integer adds: the replicas' order does not matter
  }
  const double m = (double)s0 * (1.0 / (double)(1ll << GN_STATS_SHIFT)) * inv_count;
  double var = (double)s1 * (1.0 / (double)(1ll << GN_STATS_SHIFT_SQ)) * inv_count - m * m;
  if (var < 0.0) var = 0.0;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// GroupNorm-apply (+ SiLU) of 8 channels; SiLU on v_exp_f32 / v_rcp_f32 (the consumers run this beside their MFMAs: an IEEE division is ~10 VALU)
__device__ __forceinline__ f16x8 gn_apply8(const f16x8 v, const float (&sc)[8], const float (&sh)[8], bool silu) {
  f16x8 o;
  if (silu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float y = fmaf((float)v[e], sc[e], sh[e]);
      o[e] = (f16)(y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y)));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)fmaf((float)v[e], sc[e], sh[e]);
  }
  return o;
}

// ---- producer side: statistics of one output tile, re-read from memory ---------------------------------------------------------------------
// Called by EVERY thread of the workgroup after the tile's rows [m0, mend) x columns [n0, nend) of `out` have been stored and every wave has
// passed `s_waitcnt vmcnt(0)` + a workgroup barrier (the stores are then in this XCD's L2; the loads below bypass the CU's L1).  Reading the
// tile back costs ONE L2 round trip at the very end of the workgroup (16-byte pieces, every load of a thread in flight at once) and keeps
// every GEMM epilogue untouched -- an in-epilogue reduction was 160 DPP-VALU per 32x32 block on the critical path of the short-K launches
// and 2-4 VGPRs in every kernel.
// Row m lives at (orw ? (m / orw) * ldo_hi + (m % orw) * ldo : m * ldo); n0, nend, ldo multiples of 8, out 16-byte aligned.
// lds: 2 * R * bn + 2 * bn floats with R = min(16, NT / (bn / 8)) (<= 33 KB).
// LDS_SRC: `out` is the workgroup's own f16 copy of the tile in LDS (row (m - sm0), column (n - sn0), row pitch ldo elements) that the
// epilogue mirrored its stores into -- no store drain, no memory round trip.  lds_floats: the scratch available at `lds`.
template <int NT, bool LDS_SRC>
__device__ __forceinline__ void gn_sink_tile(const GnSinkP& s, const f16* out, long ldo, int orw, long ldo_hi, int m0, int mend, int n0, int nend,
                                             int sm0, int sn0, float* lds, int lds_floats, int rep) {
  const int tid = threadIdx.x;
  const int bn = nend - n0;
  const int NC = bn >> 3;  // 16-byte chunks per row
  int R = NT / NC;
  const int rfit = (lds_floats - 2 * bn) / (2 * bn);
  R = R > 16 ? 16 : R;
  R = R > rfit ? rfit : R;
  R = R < 1 ? 1 : R;
  float* ls = lds;
  float* lq = lds + R * bn;
  float* cs = lds + 2 * R * bn;
  auto row_off = [&](int m) -> long {
    if constexpr (LDS_SRC) return (long)(m - sm0) * ldo;
    if (orw) { const int hi = m / orw; return (long)hi * ldo_hi + (long)(m - hi * orw) * ldo; }
    return (long)m * ldo;
  };
  for (int b = m0 / s.rps; b <= (mend - 1) / s.rps; ++b) {
    const int r0 = max(m0, b * s.rps), r1 = min(mend, (b + 1) * s.rps);
    for (int idx = tid; idx < NC * R; idx += NT) {
      const int rs = idx / NC, cx = idx - rs * NC;
      const f16* col = out + (LDS_SRC ? n0 - sn0 : n0) + 8 * cx;
      float sm[8], sq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
      for (int m = r0 + rs; m < r1; m += 8 * R) {  // eight 16-byte loads in flight (a 128-row tile on 12 row slices: two trips)
        gn_u32x4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          raw[u] = gn_u32x4{0u, 0u, 0u, 0u};
          if (m + u * R < r1) {
            if constexpr (LDS_SRC) raw[u] = *reinterpret_cast<const gn_u32x4*>(col + row_off(m + u * R));
            else raw[u] = __builtin_nontemporal_load(reinterpret_cast<const gn_u32x4*>(col + row_off(m + u * R)));  // nt: past the L1
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const f16x8 v = *reinterpret_cast<const f16x8*>(&raw[u]);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; sm[e] += f; sq[e] += f * f; }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { ls[rs * bn + 8 * cx + e] = sm[e]; lq[rs * bn + 8 * cx + e] = sq[e]; }
    }
    __syncthreads();
    for (int c = tid; c < bn; c += NT) {  // fixed order over the row slices
      float a = 0.f, q = 0.f;
      for (int r = 0; r < R; ++r) { a += ls[r * bn + c]; q += lq[r * bn + c]; }
      cs[c] = a;
      cs[bn + c] = q;
    }
    __syncthreads();
    const int g0 = (s.coff + n0) / s.cpg, g1 = (s.coff + nend - 1) / s.cpg;
    for (int g = g0 + tid; g <= g1; g += NT) {
      const int c0 = max(g * s.cpg - s.coff, n0) - n0, c1 = min((g + 1) * s.cpg - s.coff, nend) - n0;
      float a = 0.f, q = 0.f;
      for (int c = c0; c < c1; ++c) { a += cs[c]; q += cs[bn + c]; }
      gn_stats_add(s.stats + gn_stats_line(rep, b, g, s.nb, s.groups), a, q);
    }
    __syncthreads();  // lds is reused by the next sample segment
  }
}

}  // namespace
