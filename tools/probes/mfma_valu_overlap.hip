// Probe: do an MFMA stream in ONE wave and a VALU stream in ANOTHER wave of the same SIMD overlap (gfx950)?
//   (i) 4 waves, one per SIMD, MFMA only      (ii) 4 waves VALU only      (iii) 8 waves: 0-3 MFMA, 4-7 VALU (one of each per SIMD)
// (iii) ~ max((i), (ii)) means the pipes run side by side; ~ (i) + (ii) means they exclude each other.  Also the same with BOTH streams
// inside one wave (MFMA followed by independent VALU ops), accumulators in VGPRs ("v") or AGPRs ("a").
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define EXP4 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3))
#define FMA8 asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c))

// role: 0 = MFMA (VGPR acc), 1 = exp, 2 = fma, 3 = MFMA (AGPR acc); mode picks the role per wave
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  // mode 0: every wave MFMA; 1: every wave VALU; 2: waves 0-3 MFMA, 4-7 VALU
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * (threadIdx.x + i)); b[i] = (f16)(0.002f * i); }
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const float c = 0.999f;
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
      if (KIND == 3) { MFMA_A(acc0); MFMA_A(acc1); MFMA_A(acc2); MFMA_A(acc3); MFMA_A(acc0); MFMA_A(acc1); MFMA_A(acc2); MFMA_A(acc3); }
      else { MFMA_V(acc0); MFMA_V(acc1); MFMA_V(acc2); MFMA_V(acc3); MFMA_V(acc0); MFMA_V(acc1); MFMA_V(acc2); MFMA_V(acc3); }  // 256 cycles
    }
  } else {
    for (int i = 0; i < iters; ++i) {
      if (KIND == 1) { EXP4; EXP4; EXP4; EXP4; EXP4; EXP4; EXP4; EXP4; }   // 32 exps = 256 cycles
      else { FMA8; FMA8; FMA8; FMA8; FMA8; FMA8; FMA8; FMA8; }              // 64 fmas = 256 cycles
    }
  }
  float s = x0 + x1 + x2 + x3;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// both streams inside ONE wave: per MFMA, NF independent VALU ops
template <int KIND, int ACC>
__global__ __launch_bounds__(256) void k1(float* out, int iters) {
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * (threadIdx.x + i)); b[i] = (f16)(0.002f * i); }
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const float c = 0.999f;
  for (int i = 0; i < iters; ++i) {
#define GROUP(acc) if (ACC) MFMA_A(acc); else MFMA_V(acc); if (KIND == 1) { EXP4; } else if (KIND == 2) { FMA8; }
    GROUP(acc0) GROUP(acc1) GROUP(acc2) GROUP(acc3) GROUP(acc0) GROUP(acc1) GROUP(acc2) GROUP(acc3)
  }
  float s = x0 + x1 + x2 + x3;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* out; (void)hipMalloc(&out, 4096 * 512 * 4);
  const int iters = 4000, blocks = 256;  // one block per CU
  const double cyc = 2.4e6;  // cycles per ms at 2.4 GHz (nominal; ratios are what matter)
  auto rep = [&](const char* what, float ms) { printf("%-88s %8.3f ms  (%6.1f cycles per 8-MFMA-equivalent iteration at 2.4 GHz)\n", what, ms, ms * cyc / iters); };
  rep("(i)   256 threads: 4 waves MFMA only, one per SIMD (VGPR acc)", timeit([&] { k<1><<<blocks, 256>>>(out, iters, 0); }));
  rep("(i')  256 threads: 4 waves MFMA only (AGPR acc)", timeit([&] { k<3><<<blocks, 256>>>(out, iters, 0); }));
  rep("(ii)  256 threads: 4 waves v_exp_f32 only (32 per iteration)", timeit([&] { k<1><<<blocks, 256>>>(out, iters, 1); }));
  rep("(ii') 256 threads: 4 waves v_fma_f32 only (64 per iteration)", timeit([&] { k<2><<<blocks, 256>>>(out, iters, 1); }));
  rep("(iii) 512 threads: waves 0-3 MFMA (VGPR acc) + waves 4-7 v_exp_f32", timeit([&] { k<1><<<blocks, 512>>>(out, iters, 2); }));
  rep("(iii') 512 threads: waves 0-3 MFMA (VGPR acc) + waves 4-7 v_fma_f32", timeit([&] { k<2><<<blocks, 512>>>(out, iters, 2); }));
  rep("(iii'') 512 threads: waves 0-3 MFMA (AGPR acc) + waves 4-7 v_exp_f32", timeit([&] { k<3><<<blocks, 512>>>(out, iters, 2); }));
  rep("(iv)  512 threads: 8 waves MFMA only (two per SIMD)", timeit([&] { k<1><<<blocks, 512>>>(out, iters, 0); }));
  rep("(v)   512 threads: 8 waves v_exp_f32 only", timeit([&] { k<1><<<blocks, 512>>>(out, iters, 1); }));
  rep("(vi)  one wave per SIMD, per MFMA 4 v_exp_f32 in the same wave (VGPR acc)", timeit([&] { k1<1, 0><<<blocks, 256>>>(out, iters); }));
  rep("(vi') one wave per SIMD, per MFMA 4 v_exp_f32 in the same wave (AGPR acc)", timeit([&] { k1<1, 1><<<blocks, 256>>>(out, iters); }));
  rep("(vii) one wave per SIMD, per MFMA 8 v_fma_f32 in the same wave (VGPR acc)", timeit([&] { k1<2, 0><<<blocks, 256>>>(out, iters); }));
  rep("(vii') one wave per SIMD, per MFMA 8 v_fma_f32 in the same wave (AGPR acc)", timeit([&] { k1<2, 1><<<blocks, 256>>>(out, iters); }));
  return 0;
}
