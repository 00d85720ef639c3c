// Probe: what a GEMM epilogue's store pattern costs.  M x N f16 output written by (M/128) x (N/64) blocks of 256 threads.
//   MODE 0: empty kernel (launch + drain floor)
//   MODE 1: MFMA-layout stores: a wave instruction = 32 rows x 32 B (lane l31 = row, hi = 16-byte half)          [today's epilogue]
//   MODE 2: row-contiguous stores: a wave instruction = 8 rows x 128 B (full lines), same bytes per lane
//   MODE 3: mode 2 + a 8-byte residual read per lane pair in the same layout (16 B loads, full lines)
//   MODE 4: mode 1 + residual read in the MFMA layout (8-byte loads, as today)
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;

template <int MODE>
__global__ __launch_bounds__(256) void k(f16* out, const f16* res, int M, int N, float seed) {
  if (MODE == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 64;
  // wave tile 64 x 32 (2 x 1 MFMA tiles) for MODE 1/4;  each lane: 2 tiles x 2 stores of 16 B
  uint4 v = make_uint4(__float_as_uint(seed) + tid, tid, lane, wave);
  if (MODE == 1 || MODE == 4) {
    const int l31 = lane & 31, hi = lane >> 5;
    const int mb = m0 + (wave >> 1) * 64, nb = n0 + (wave & 1) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const long o = (long)(mb + i * 32 + l31) * N + nb + 16 * g + 8 * hi;
        uint4 w = v;
        if (MODE == 4) {
          const uint2 r0 = *reinterpret_cast<const uint2*>(res + o), r1 = *reinterpret_cast<const uint2*>(res + o + 4);
          w.x ^= r0.x; w.y ^= r0.y; w.z ^= r1.x; w.w ^= r1.y;
        }
        *reinterpret_cast<uint4*>(out + o) = w;
      }
  } else {
    // block tile 128 x 64 halfs = 128 rows x 128 B; a wave instruction covers 8 rows; wave w takes rows w*32 .. w*32+31
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = (long)(m0 + wave * 32 + i * 8 + (lane >> 3)) * N + n0 + (lane & 7) * 8;
      uint4 w = v;
      if (MODE == 3) { const uint4 r = *reinterpret_cast<const uint4*>(res + o); w.x ^= r.x; w.y ^= r.y; w.z ^= r.z; w.w ^= r.w; }
      *reinterpret_cast<uint4*>(out + o) = w;
    }
  }
}

template <int MODE>
float run(f16* out, f16* res, int M, int N, int iters) {
  dim3 g(M / 128, N / 64);
  for (int i = 0; i < 10; ++i) k<MODE><<<g, 256>>>(out, res, M, N, 1.0f);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipDeviceSynchronize(); hipEventRecord(a);
  for (int i = 0; i < iters; ++i) k<MODE><<<g, 256>>>(out, res, M, N, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / iters;
}

int main() {
  const int shapes[][2] = {{8192, 640}, {32768, 320}, {2048, 1280}, {32768, 1280}, {131072, 128}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1];
    f16 *out, *res; hipMalloc(&out, (size_t)M * N * 2); hipMalloc(&res, (size_t)M * N * 2); hipMemset(res, 1, (size_t)M * N * 2);
    printf("M=%6d N=%5d (%.1f MB)  empty %.2f  mfma-layout %.2f  rows %.2f  rows+res %.2f  mfma-layout+res %.2f us\n", M, N,
           M * (double)N * 2 / 1e6, run<0>(out, res, M, N, 300), run<1>(out, res, M, N, 300), run<2>(out, res, M, N, 300),
           run<3>(out, res, M, N, 300), run<4>(out, res, M, N, 300));
    hipFree(out); hipFree(res);
  }
  return 0;
}
