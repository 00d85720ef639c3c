// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) on gfx950.
// Hypothesis checked against a CPU product: lane l supplies A[row = l & 31][k = 32 (l >> 5) .. + 32] as 32 consecutive bytes
// (byte j of dword i = k offset 4 i + j), B[k][col = l & 31] with the same k mapping; D as the other 32x32 MFMAs
// (acc[r]: row 8 (r >> 2) + 4 (l >> 5) + (r & 3), col l & 31).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_fp8_layout.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int int8v __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A, const unsigned char* B, float* D) {  // A [32][64], B^T [32 cols][64 k], D [32][32]
  const int l = threadIdx.x, l31 = l & 31, hi = l >> 5;
  int8v a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(A + l31 * 64 + hi * 32 + 4 * i);
    b[i] = *reinterpret_cast<const int*>(B + l31 * 64 + hi * 32 + 4 * i);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * hi + (r & 3)) * 32 + l31] = acc[r];
}

int main() {
  const unsigned char codes[8] = {0x00, 0x38, 0x40, 0x30, 0xB8, 0x3C, 0x44, 0xC0};  // e4m3: 0, 1, 2, .5, -1, 1.5, 3, -2
  const float vals[8] = {0.f, 1.f, 2.f, .5f, -1.f, 1.5f, 3.f, -2.f};
  unsigned char hA[32 * 64], hB[32 * 64];
  float fA[32 * 64], fB[32 * 64], ref[32 * 32], out[32 * 32];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) {
    int x = rand() & 7, y = rand() & 7;
    hA[i] = codes[x]; fA[i] = vals[x];
    hB[i] = codes[y]; fB[i] = vals[y];
  }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float s = 0.f;
      for (int kk = 0; kk < 64; ++kk) s += fA[i * 64 + kk] * fB[j * 64 + kk];
      ref[i * 32 + j] = s;
    }
  unsigned char *dA, *dB;
  float* dD;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof out);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  (void)hipMemcpy(out, dD, sizeof out, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i)
    if (out[i] != ref[i]) ++bad;
  printf("fp8 32x32x64 layout probe: %d / 1024 mismatches (D[0][0] = %g, ref %g; D[5][7] = %g, ref %g)\n", bad, out[0], ref[0], out[5 * 32 + 7], ref[5 * 32 + 7]);
  return bad != 0;
}
