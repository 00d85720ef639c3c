// Probe: how many VALU instructions issue for free beside v_mfma_f32_32x32x16_f16 on gfx950, with the accumulators in VGPRs
// (-mllvm -amdgpu-mfma-vgpr-form=1) or in AGPRs (default)?  One MFMA followed by NV independent VALU ops, pinned with scheduling
// fences; reports time per MFMA for NV = 0..10, for plain (v_fma_f32), transcendental (v_exp_f32) and v_dot2c fillers.
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form=1] tools/probes/mfma_coissue.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(64) void probe(float* out, int iters) {
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u)
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (f16)(0.001f * (threadIdx.x + i)); b[i] = (f16)(0.002f * (threadIdx.x + 2 * i)); }
  float v[10];
  for (int i = 0; i < 10; ++i) v[i] = 0.1f * (threadIdx.x + i);
  f16x2 h = {(f16)0.5f, (f16)0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        if (KIND == 0) v[n] = __builtin_fmaf(v[n], 1.0001f, 0.5f);
        else if (KIND == 1) v[n] = __builtin_amdgcn_exp2f(v[n]);
        else v[n] = __builtin_amdgcn_fdot2(h, h, v[n], false);
      }
    }
  }
  float s = 0.0f;
  for (int u = 0; u < 4; ++u)
    for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int i = 0; i < 10; ++i) s += v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NV, int KIND>
void run(float* out, int blocks, const char* tag) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NV, KIND><<<blocks, 64>>>(out, 100);
  hipEventRecord(e0);
  probe<NV, KIND><<<blocks, 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = blocks / 1024.0;
  printf("%s blocks=%d NV=%2d: %7.2f ns per MFMA per SIMD\n", tag, blocks, NV, ms * 1e6 / (iters * 4.0 * (waves_per_simd < 1 ? 1 : waves_per_simd)));
}

template <int KIND>
void sweep(float* out, int blocks, const char* tag) {
  run<0, KIND>(out, blocks, tag); run<1, KIND>(out, blocks, tag); run<2, KIND>(out, blocks, tag); run<3, KIND>(out, blocks, tag);
  run<4, KIND>(out, blocks, tag); run<5, KIND>(out, blocks, tag); run<6, KIND>(out, blocks, tag); run<8, KIND>(out, blocks, tag);
  run<10, KIND>(out, blocks, tag);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 64 * sizeof(float));
  for (int blocks : {1024, 2048}) {
    sweep<0>(out, blocks, "fma  ");
    sweep<1>(out, blocks, "exp  ");
    sweep<2>(out, blocks, "dot2c");
  }
  return 0;
}
