// Probe: v_mfma_f32_32x32x16_f16 whose A operand comes fresh from LDS for every MFMA (attention's QK^T / PV phases at one q-tile per wave:
// one ds_read_b128 per MFMA) -- cycles per MFMA against the register-fed loop, at 1, 2, 3 waves per SIMD.
//   READS per MFMA: 0 (registers only), 1 (attention TQ = 1), variants: reads issued one iteration AHEAD (software prefetch) or just in time.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_lds_feed.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// MODE 0: no LDS reads.  1: per iteration 8 reads then 8 MFMAs (compiler places waits).  2: reads for iteration i+1 issued before the MFMAs of i.
// NACC accumulators in rotation (2 = attention's two chains, 4 = independent enough)
template <int MODE, int NACC>
__global__ void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u)
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  f16x8 b;
  for (int i = 0; i < 8; ++i) b[i] = (f16)(0.002f * (i + lane));
  f16x8 fr[8], fn[8];
  int off[8];
  for (int i = 0; i < 8; ++i) off[i] = swz((i >> 2) * 32 + l31, (i & 3) * 2 + hi);
  for (int i = 0; i < 8; ++i) fr[i] = *reinterpret_cast<const f16x8*>(smem + off[i]);
  for (int it = 0; it < iters; ++it) {
    const unsigned char* T = smem + (it & 1) * 16384;
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) fr[i] = *reinterpret_cast<const f16x8*>(T + off[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], b, acc[i % NACC], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) fn[i] = *reinterpret_cast<const f16x8*>(T + off[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], b, acc[i % NACC], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) fr[i] = fn[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], b, acc[i % NACC], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int u = 0; u < 4; ++u)
    for (int r = 0; r < 16; ++r) s += acc[u][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* what, int threads, float* out) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, NACC><<<blocks, threads>>>(out, iters);
  (void)hipEventRecord(e0);
  k<MODE, NACC><<<blocks, threads>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 8 * (threads / 256);
  printf("%-70s %d wave(s)/SIMD: %7.2f ns per MFMA per SIMD (%5.1f cycles at 1.9 GHz)\n", what, threads / 256, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 1.9);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
  for (int th : {256, 512, 768}) {
    if (th == 256) { run<0, 2>("registers only, 2 accumulator chains", 256, out); run<0, 4>("registers only, 4 accumulators", 256, out);
                     run<1, 2>("1 ds_read_b128 per MFMA, just in time, 2 chains", 256, out); run<1, 4>("1 ds_read_b128 per MFMA, just in time, 4 acc", 256, out);
                     run<2, 2>("1 ds_read_b128 per MFMA, one iteration ahead, 2 chains", 256, out); run<2, 4>("1 ds_read_b128 per MFMA, one iteration ahead, 4 acc", 256, out); }
    if (th == 512) { run<0, 2>("registers only, 2 accumulator chains", 512, out);
                     run<1, 2>("1 ds_read_b128 per MFMA, just in time, 2 chains", 512, out); run<2, 2>("1 ds_read_b128 per MFMA, one iteration ahead, 2 chains", 512, out); }
    if (th == 768) { run<0, 2>("registers only, 2 accumulator chains", 768, out);
                     run<1, 2>("1 ds_read_b128 per MFMA, just in time, 2 chains", 768, out); run<2, 2>("1 ds_read_b128 per MFMA, one iteration ahead, 2 chains", 768, out); }
  }
  return 0;
}
