// Probe: LDS cycles per ds_read_b128 wave-instruction for the fragment address patterns of the attention / GEMM kernels (4 waves per CU,
// one per SIMD, nothing else running): s_memtime around 256 dependent-free reads per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_read_pattern.hip -o tools/probes/bin/lds_read_pattern && tools/probes/bin/lds_read_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int addr_of(int pat, int lane, int i) {
  const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, q = lane >> 4;
  switch (pat) {
    case 0: return lane * 16 + i * 1024;                                              // lane-linear
    case 1: return l31 * 128 + (((i * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);              // lds_swz<128>: rows l31, chunk pair (2i, 2i+1) by half-wave
    case 2: return l31 * 128 + ((i * 2 + hi) << 4);                                   // the same without the XOR
    case 3: return l31 * 128 + (((i * 2 + hi) ^ (l31 & 7)) << 4);                     // XOR by row & 7
    case 4: return l31 * 128 + (((i + 4 * hi) ^ ((l31 >> 1) & 7)) << 4);              // half-waves 64 bytes apart
    case 5: return (l31 + 32 * hi) * 128 + ((i ^ (((l31 + 32 * hi) >> 1) & 7)) << 4); // 64 rows, one chunk column
    case 6: return l15 * 128 + (((i * 4 + q) ^ ((l15 >> 1) & 7)) << 4) ;              // 16 rows x 4 chunks (16x16 MFMA operand)
    case 7: return l31 * 144 + ((i * 2 + hi) << 4);                                   // padded rows (128 + 16 bytes), no XOR
    case 8: return l31 * 128 + (((i * 2 + hi) ^ (((l31 >> 1) & 3) * 2 + ((l31 >> 3) & 1))) << 4);  // another mix
    default: return 0;
  }
}

template <int PAT>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  int off[4];
  for (int i = 0; i < 4; ++i) off[i] = addr_of(PAT, lane, i);
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
    u32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const u32x4*>(smem + off[j & 3] + (j >> 2) * 8192 + (it & 1) * 32768);
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= v[j];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345678u && acc[1] == 1 && acc[2] == 2 && acc[3] == 77) sink[threadIdx.x] = 1;
}

template <int PAT>
void run(const char* name, unsigned long long* out, unsigned* sink) {
  k<PAT><<<256, 256>>>(out, sink);
  k<PAT><<<256, 256>>>(out, sink);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 256; ++i) s += (double)h[i];
  s /= 256;
  printf("%-58s %8.0f cycles for 256 reads per wave, 4 waves  -> %5.2f cycles per wave-instruction (CU level)\n", name, s, s / (256.0 * 4));
}

int main() {
  unsigned long long* out; unsigned* sink;
  hipMalloc(&out, 256 * 8); hipMalloc(&sink, 4096);
  run<0>("0 lane-linear", out, sink);
  run<1>("1 lds_swz<128> (rows l31, chunks 2i | 2i+1 by half-wave)", out, sink);
  run<2>("2 same rows, no XOR", out, sink);
  run<3>("3 XOR by row & 7", out, sink);
  run<4>("4 half-waves 64 bytes apart, XOR (row >> 1) & 7", out, sink);
  run<5>("5 64 rows x one chunk column, XOR (row >> 1) & 7", out, sink);
  run<6>("6 16 rows x 4 chunks, XOR (row >> 1) & 7", out, sink);
  run<7>("7 rows padded to 144 bytes, no XOR", out, sink);
  run<8>("8 another XOR mix", out, sink);
  return 0;
}
