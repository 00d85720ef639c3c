#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(f16* out) {
  __shared__ f16 lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (f16)i;   // value = row*64 + col  (row stride 64 elements)
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, grp = l >> 4;
  // group grp: rows 4*grp .. 4*grp+3, cols 0..15 ; lane i supplies row (i/4), cols 4*(i%4)
  const f16* p = lds + (4 * grp + (i >> 2)) * 64 + 4 * (i & 3);
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16)))); h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  f16* d; hipMalloc(&d, 64 * 4 * 2); k<<<1, 64>>>(d); f16 h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5.0f", (float)h[l * 4 + j]); printf("\n"); }
  return 0;
}
