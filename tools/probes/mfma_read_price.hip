// Probe: what one ds_read_b128 costs beside v_mfma_f32_32x32x16_f16 in a stream of ONE WAVE PER SIMD (attention_pwg.hip's regime).
// Per loop iteration 8 MFMAs on 4 accumulators; NR fragment reads per iteration, read i issued right behind MFMA i * (8 / NR) into the
// OTHER register set and consumed as the A operand of the next iteration (8 MFMA slots of flight).  Variants: BURST (all reads of the
// iteration behind MFMA 0), VALU (4 independent v_fma beside every MFMA), SINK (reads never consumed by an MFMA: a v_or chain at the end).
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form=1] tools/probes/mfma_read_price.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FENCE __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int NR, int BURST, int NV, int LINEAR>
__global__ __launch_bounds__(256, 1) void k(const unsigned* seed, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[49152];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 49152 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = (seed[i % 4096] & 0x3fff3fffu) | 0x20002000u;  // finite f16 pairs
  __syncthreads();
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u)
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  f16x8 b;
  for (int i = 0; i < 8; ++i) b[i] = (f16)(0.002f * (i + lane));
  f16x8 fa[8], fb[8];
  int off[8];
  for (int i = 0; i < 8; ++i) off[i] = LINEAR ? ((i * 64 + lane) * 16) : swz((i >> 2) * 32 + l31, (i & 3) * 2 + hi) + (i >> 2) * 4096;
  for (int i = 0; i < 8; ++i) fa[i] = fb[i] = *reinterpret_cast<const f16x8*>(smem + off[i]);
  float v0 = lane, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  auto body = [&](const f16x8 (&f)[8], f16x8 (&g)[8], const unsigned char* T) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      FENCE;
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], b, acc[i & 3], 0, 0, 0);
      FENCE;
      if (NR > 0) {
        if (BURST) {
          if (i == 0)
#pragma unroll
            for (int j = 0; j < NR; ++j) g[j] = *reinterpret_cast<const f16x8*>(T + off[j]);
        } else if (i % (8 / (NR ? NR : 1)) == 0) {
          const int j = i / (8 / (NR ? NR : 1));
          g[j] = *reinterpret_cast<const f16x8*>(T + off[j]);
        }
      }
      if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
      if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v3));
      if (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v3), "v"(v0));
      if (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v0), "v"(v1));
    }
  };
  for (int it = 0; it < iters; it += 2) {
    body(fa, fb, smem + 16384);
    body(fb, fa, smem + 32768);
  }
  float s = v0 + v1 + v2 + v3;
  for (int u = 0; u < 4; ++u)
    for (int r = 0; r < 16; ++r) s += acc[u][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NR, int BURST, int NV, int LINEAR>
void run(const char* name, const unsigned* seed, float* out) {
  const int iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<NR, BURST, NV, LINEAR><<<256, 256>>>(seed, out, 64);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k<NR, BURST, NV, LINEAR><<<256, 256>>>(seed, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double ns = best * 1e6 / (iters * 8.0);
  printf("%-52s %7.2f ns per MFMA  (%5.1f cycles at 2.2 GHz)   %6.0f TF/s chip\n", name, ns, ns * 2.2, 256.0 * 4 * 32768 / ns / 1e3);
}

int main() {
  unsigned* seed; float* out;
  hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 4096);
  unsigned h[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand() * 2654435761u;
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 0, 0, 0>("8 MFMA, no reads", seed, out);
  run<2, 0, 0, 0>("8 MFMA + 2 ds_read_b128 (spread)", seed, out);
  run<4, 0, 0, 0>("8 MFMA + 4 ds_read_b128 (spread)", seed, out);
  run<8, 0, 0, 0>("8 MFMA + 8 ds_read_b128 (one per MFMA)", seed, out);
  run<4, 1, 0, 0>("8 MFMA + 4 ds_read_b128 (burst behind MFMA 0)", seed, out);
  run<8, 1, 0, 0>("8 MFMA + 8 ds_read_b128 (burst behind MFMA 0)", seed, out);
  run<4, 0, 0, 1>("8 MFMA + 4 ds_read_b128 (spread, lane-linear addr)", seed, out);
  run<0, 0, 4, 0>("8 MFMA + 4 v_fma each, no reads", seed, out);
  run<4, 0, 4, 0>("8 MFMA + 4 v_fma each + 4 ds_read_b128 (spread)", seed, out);
  run<8, 0, 4, 0>("8 MFMA + 4 v_fma each + 8 ds_read_b128", seed, out);
  return 0;
}
