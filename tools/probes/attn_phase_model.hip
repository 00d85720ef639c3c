// Probe: a register-only model of the attention wave's 64-key iteration -- 8 QK^T MFMAs (two 4-deep chains), the optimistic softmax on their
// results (32 v_exp_f32, 16 v_cvt_pkrtz, 16 v_dot2c), 8 PV MFMAs fed by the converted P -- with NO LDS, NO memory, NO barrier: what do W
// waves per SIMD make of the two pipes when every wave runs this three-phase cycle?
//   MODE 0: phases in program order (the generic kernel's order).
//   MODE 1: software-pipelined inside the wave: QK^T(t+1) and PV(t-1) MFMAs alternate with softmax(t) pieces (1 MFMA : 2 exp + cvt + dot).
//   PRIO 1: s_setprio(1) while a wave is in its MFMA phases (MODE 0 only).
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/attn_phase_model.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FENCE __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int MODE, int PRIO>
__global__ __launch_bounds__(768) void k(float* out, int iters) {  // (768 threads = three waves per SIMD: 170 registers, no spills)
  __shared__ __attribute__((aligned(16))) unsigned char smem[49152];
  const int lane = threadIdx.x & 63;
  // RANDOM_DATA: f16 pairs with pseudo-random mantissas and exponents around 1 (what real K / V tiles look like to the data paths: the chip
  // clocks to its power budget, and operand toggling is power) instead of a smooth ramp
#ifdef RANDOM_DATA
  for (int i = threadIdx.x; i < 49152 / 4; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    reinterpret_cast<unsigned*>(smem)[i] = (h & 0x83FF83FFu) | 0x38003800u;  // sign + mantissa random, exponent 14 (0.5 .. 1)
  }
#else
  for (int i = threadIdx.x; i < 49152 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 1023);
#endif
  __syncthreads();
  const int l31 = lane & 31, hi = lane >> 5;
  int offk[4], offv[2][2];
  for (int i = 0; i < 4; ++i) offk[i] = swz(l31, i * 2 + hi);
  for (int u = 0; u < 2; ++u)
    for (int q = 0; q < 2; ++q) offv[u][q] = 8192 + swz(l31, u * 4 + q * 2 + hi);
  auto frag = [&](const unsigned char* X, int u, int i) -> f16x8 {
    if (i < 4) return *reinterpret_cast<const f16x8*>(X + offk[i] + u * 4096);
    const int n = i - 4;
    return *reinterpret_cast<const f16x8*>(X + offv[u][n >> 1] + (n & 1) * 4096);
  };
  f16x8 kf[4], vf[4], qf[4];
  for (int i = 0; i < 4; ++i)
    for (int x = 0; x < 8; ++x) { kf[i][x] = (f16)(0.01f * (x + i)); vf[i][x] = (f16)(0.02f * (x - i)); qf[i][x] = (f16)(0.001f * (lane + x)); }
  f32x16 o0, o1, negm, s0, s1, t0, t1;
  for (int r = 0; r < 16; ++r) { o0[r] = o1[r] = 0.f; negm[r] = -1.f; s0[r] = s1[r] = t0[r] = t1[r] = 0.f; }
  const f16x2 ones = {(f16)1.f, (f16)1.f};
  float l = 0.f;
  f16x8 p0[2], p1[2];
  for (int i = 0; i < 2; ++i)
    for (int x = 0; x < 8; ++x) { p0[i][x] = (f16)0.5f; p1[i][x] = (f16)0.25f; }

  auto soft = [&](const f32x16& s, f16x8 (&pf)[2]) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f16x2 pp;
      pp[0] = (f16)__builtin_amdgcn_exp2f(s[r]);
      pp[1] = (f16)__builtin_amdgcn_exp2f(s[r + 1]);
      acc = __builtin_amdgcn_fdot2(pp, ones, acc, false);
      pf[r >> 3][r & 7] = pp[0];
      pf[r >> 3][(r & 7) + 1] = pp[1];
    }
    l += acc;
  };
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? negm : s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks], qf[ks], ks == 0 ? negm : s1, 0, 0, 0);
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      soft(s0, p0);
      soft(s1, p1);
      FENCE;
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[n], p0[n], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[n + 2], p0[n], o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[n], p1[n], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[n + 2], p1[n], o1, 0, 0, 0);
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      FENCE;
    }
  } else if (MODE == 2) {
    // inside ONE 64-key tile, no cross-tile pipelining and no extra registers: QK0 | QK1 beside softmax(0) | PV0 beside softmax(1) | PV1
    auto half = [&](const f32x16& sc, f16x8 (&pc)[2], auto mf) __attribute__((always_inline)) {
      float ex[16], acc = 0.f;
      f16x2 pk[8];
      auto E = [&](int q) { ex[2 * q] = __builtin_amdgcn_exp2f(sc[2 * q]); ex[2 * q + 1] = __builtin_amdgcn_exp2f(sc[2 * q + 1]); };
      auto C = [&](int q) { pk[q][0] = (f16)ex[2 * q]; pk[q][1] = (f16)ex[2 * q + 1]; pc[q >> 2][(2 * q) & 7] = pk[q][0]; pc[q >> 2][((2 * q) & 7) + 1] = pk[q][1]; };
      auto S = [&](int q) { acc = __builtin_amdgcn_fdot2(pk[q], ones, acc, false); };
      FENCE; mf(0); FENCE; E(0); E(1); C(0);
      FENCE; mf(1); FENCE; E(2); E(3); C(1); C(2); S(0); S(1);
      FENCE; mf(2); FENCE; E(4); E(5); C(3); C(4); S(2); S(3);
      FENCE; mf(3); FENCE; E(6); E(7); C(5); C(6); C(7); S(4); S(5); S(6); S(7);
      FENCE;
      l += acc;
    };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? negm : s0, 0, 0, 0);
      half(s0, p0, [&](int ks) { s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ks], qf[ks], ks == 0 ? negm : s1, 0, 0, 0); });
      half(s1, p1, [&](int n) { if (n & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[n], p0[n >> 1], o1, 0, 0, 0); else o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[n], p0[n >> 1], o0, 0, 0, 0); });
      FENCE;
#pragma unroll
      for (int n = 0; n < 4; ++n) { if (n & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[n], p1[n >> 1], o1, 0, 0, 0); else o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[n], p1[n >> 1], o0, 0, 0, 0); }
      FENCE;
    }
  } else if (MODE >= 3 && MODE <= 8) {
    unsigned long long sticky = 0;
    float prev_acc = 0.f;
    // MODE 1 + every MFMA's A fragment read from LDS just in time (MODE 3: two groups ahead, MODE 4: three)
    constexpr int AH = MODE == 4 ? 3 : 2;
    auto stage = [&](const f32x16& sc, f32x16& sn, f16x8 (&pc)[2], const f16x8 (&pp)[2], const unsigned char* X, int u) __attribute__((always_inline)) {
      float ex[16], acc = 0.f;
      f16x2 pk[8];
      f16x8 f[8 + 3];
      auto FI = [](int i) { return (i & 1) ? 4 + (i >> 1) : (i >> 1); };
      auto R = [&](int i) { if (i < 8) f[i] = frag(X, u, FI(i)); };
      auto E = [&](int q) { ex[2 * q] = __builtin_amdgcn_exp2f(sc[2 * q]); ex[2 * q + 1] = __builtin_amdgcn_exp2f(sc[2 * q + 1]); };
      auto C = [&](int q) { pk[q][0] = (f16)ex[2 * q]; pk[q][1] = (f16)ex[2 * q + 1]; pc[q >> 2][(2 * q) & 7] = pk[q][0]; pc[q >> 2][((2 * q) & 7) + 1] = pk[q][1]; };
      auto S = [&](int q) { acc = __builtin_amdgcn_fdot2(pk[q], ones, acc, false); };
      auto M = [&](int i) {
        const int n = i >> 1;
        if ((i & 1) == 0) sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], qf[n], n == 0 ? negm : sn, 0, 0, 0);
        else { if (n & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], pp[n >> 1], o1, 0, 0, 0); else o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], pp[n >> 1], o0, 0, 0, 0); }
      };
#pragma unroll
      for (int i = 0; i < AH; ++i) R(i);
      FENCE; M(0); FENCE; R(0 + AH); E(0);
      FENCE; M(1); FENCE; R(1 + AH); E(1); C(0);
      if (MODE == 8) {  // branch on the STICKY scalar flag the previous stage left (no VALU result involved at the branch)
        FENCE;
        if (__builtin_expect(sticky != 0, 0)) {
          const float a = __builtin_amdgcn_exp2f(-prev_acc);
          for (int r = 0; r < 16; ++r) { o0[r] *= a; o1[r] *= a; negm[r] -= prev_acc; }
          l *= a; sticky = 0;
        }
      }
      if (MODE == 7) {  // the PREVIOUS stage's check, two MFMAs into this stage (timing only: the slow path is a stand-in)
        FENCE;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(prev_acc <= 8192.0f)) != 0, 0)) {
          const float a = __builtin_amdgcn_exp2f(-prev_acc);
          for (int r = 0; r < 16; ++r) { o0[r] *= a; o1[r] *= a; negm[r] -= prev_acc; }
          l *= a;
        }
      }
      FENCE; M(2); FENCE; R(2 + AH); E(2); C(1); S(0);
      FENCE; M(3); FENCE; R(3 + AH); E(3); C(2); S(1);
      FENCE; M(4); FENCE; R(4 + AH); E(4); C(3); S(2);
      FENCE; M(5); FENCE; R(5 + AH); E(5); C(4); S(3);
      FENCE; M(6); FENCE; E(6); C(5); S(4);
      FENCE; M(7); FENCE; E(7); C(6); S(5);
      FENCE; C(7); S(6); S(7);
      FENCE;
      if (MODE == 6 || MODE == 8) sticky |= __builtin_amdgcn_ballot_w64(!(acc <= 8192.0f));
      if (MODE == 8) prev_acc = acc;  // no branch: a sticky flag, acted on after the loop
      if (MODE == 7) prev_acc = acc;
      if (MODE == 5) {  // the optimistic check: one compare, one branch on vcc; the (never taken) slow path rescales everything
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(acc <= 8192.0f)) != 0, 0)) {
          const float a = __builtin_amdgcn_exp2f(-acc);
          for (int r = 0; r < 16; ++r) { o0[r] *= a; o1[r] *= a; negm[r] -= acc; sn[r] -= acc; }
          l *= a;
        }
      }
      l += acc;
    };
    for (int it = 0; it < iters; ++it) {
      const unsigned char* X = smem + (it % 3) * 16384;
      stage(s0, s1, p0, p1, X, 0);
      stage(s1, s0, p1, p0, X, 1);
    }
    if (sticky) l = -1.f;
  } else {
    // stage j (a 32-key sub-tile): PV(j-1) 4 MFMA + QK^T(j+1) 4 MFMA interleaved with softmax(j): 16 exp, 8 cvt, 8 dot
    auto stage = [&](const f32x16& sc, f32x16& sn, f16x8 (&pc)[2], const f16x8 (&pp)[2]) __attribute__((always_inline)) {
      float ex[16], acc = 0.f;
      f16x2 pk[8];
      auto E = [&](int q) { ex[2 * q] = __builtin_amdgcn_exp2f(sc[2 * q]); ex[2 * q + 1] = __builtin_amdgcn_exp2f(sc[2 * q + 1]); };
      auto C = [&](int q) { pk[q][0] = (f16)ex[2 * q]; pk[q][1] = (f16)ex[2 * q + 1]; pc[q >> 2][(2 * q) & 7] = pk[q][0]; pc[q >> 2][((2 * q) & 7) + 1] = pk[q][1]; };
      auto S = [&](int q) { acc = __builtin_amdgcn_fdot2(pk[q], ones, acc, false); };
      auto M = [&](int i) {
        const int n = i >> 1;
        if ((i & 1) == 0) sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[n], qf[n], n == 0 ? negm : sn, 0, 0, 0);
        else { if (n & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[n], pp[n >> 1], o1, 0, 0, 0); else o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[n], pp[n >> 1], o0, 0, 0, 0); }
      };
      FENCE; M(0); FENCE; E(0);
      FENCE; M(1); FENCE; E(1); C(0);
      FENCE; M(2); FENCE; E(2); C(1); S(0);
      FENCE; M(3); FENCE; E(3); C(2); S(1);
      FENCE; M(4); FENCE; E(4); C(3); S(2);
      FENCE; M(5); FENCE; E(5); C(4); S(3);
      FENCE; M(6); FENCE; E(6); C(5); S(4);
      FENCE; M(7); FENCE; E(7); C(6); S(5);
      FENCE; C(7); S(6); S(7);
      FENCE;
      l += acc;
    };
    for (int it = 0; it < iters; ++it) {
      stage(s0, s1, p0, p1);
      stage(s1, s0, p1, p0);
    }
  }
  float s = l;
  for (int r = 0; r < 16; ++r) s += o0[r] + o1[r] + s0[r] + s1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int PRIO>
void run(const char* what, int threads, float* out) {
  const int iters = 3000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, PRIO><<<blocks, threads>>>(out, iters);
  (void)hipEventRecord(e0);
  k<MODE, PRIO><<<blocks, threads>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double tiles_per_simd = (double)iters * (threads / 256);
  printf("%-64s %d wave(s)/SIMD: %7.1f ns per 64-key wave tile per SIMD (%6.0f cycles at 1.9 GHz; 16 MFMAs alone = 296 ns)\n", what, threads / 256,
         ms * 1e6 / tiles_per_simd, ms * 1e6 / tiles_per_simd * 1.9);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
  for (int th : {256, 512, 768}) {
    if (th == 256) { run<0, 0>("phases in program order", 256, out); run<0, 1>("phases in program order, s_setprio(1) in MFMA phases", 256, out); run<1, 0>("software-pipelined in the wave", 256, out); run<2, 0>("inside one tile: QK0 | QK1+soft0 | PV0+soft1 | PV1", 256, out); run<3, 0>("software-pipelined + LDS fragment reads two groups ahead", 256, out); run<4, 0>("software-pipelined + LDS fragment reads three groups ahead", 256, out); run<5, 0>("software-pipelined + LDS reads (2 ahead) + optimistic check per stage", 256, out); run<6, 0>("  ... + sticky flag instead of the branch", 256, out); run<7, 0>("  ... + the check two MFMAs into the NEXT stage", 256, out); run<8, 0>("  ... + sticky flag, branched on (scalar) two MFMAs into the next stage", 256, out); }
    if (th == 512) { run<0, 0>("phases in program order", 512, out); run<0, 1>("phases in program order, s_setprio(1) in MFMA phases", 512, out); run<1, 0>("software-pipelined in the wave", 512, out); run<2, 0>("inside one tile: QK0 | QK1+soft0 | PV0+soft1 | PV1", 512, out); run<3, 0>("software-pipelined + LDS fragment reads two groups ahead", 512, out); run<4, 0>("software-pipelined + LDS fragment reads three groups ahead", 512, out); run<5, 0>("software-pipelined + LDS reads (2 ahead) + optimistic check per stage", 512, out); run<6, 0>("  ... + sticky flag instead of the branch", 512, out); run<7, 0>("  ... + the check two MFMAs into the NEXT stage", 512, out); run<8, 0>("  ... + sticky flag, branched on (scalar) two MFMAs into the next stage", 512, out); }
    if (th == 768) { run<0, 0>("phases in program order", 768, out); run<0, 1>("phases in program order, s_setprio(1) in MFMA phases", 768, out); run<1, 0>("software-pipelined in the wave", 768, out); run<2, 0>("inside one tile: QK0 | QK1+soft0 | PV0+soft1 | PV1", 768, out); run<3, 0>("software-pipelined + LDS fragment reads two groups ahead", 768, out); run<4, 0>("software-pipelined + LDS fragment reads three groups ahead", 768, out); run<5, 0>("software-pipelined + LDS reads (2 ahead) + optimistic check per stage", 768, out); run<6, 0>("  ... + sticky flag instead of the branch", 768, out); run<7, 0>("  ... + the check two MFMAs into the NEXT stage", 768, out); run<8, 0>("  ... + sticky flag, branched on (scalar) two MFMAs into the next stage", 768, out); }
  }
  return 0;
}
