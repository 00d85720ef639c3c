// Probe: does `buffer_load_dwordx4 ... lds` (LDS-DMA) write ZEROS to LDS for out-of-range lanes (voffset >= num_records)?
// Build: hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int int4v;

__global__ void probe(const uint32_t* src, uint32_t* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // even lanes read their chunk, odd lanes point far out of range
  unsigned voff = (threadIdx.x & 1) ? 0x7FFFFFF0u : threadIdx.x * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<uint32_t> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  uint32_t *d, *o;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 1024);
  std::vector<uint32_t> r(256);
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  int ok_even = 1, zero_odd = 1, stale_odd = 1;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      uint32_t v = r[l * 4 + j];
      if (!(l & 1)) ok_even &= (v == 1000u + l * 4 + j);
      else { zero_odd &= (v == 0); stale_odd &= (v == 0xDEADBEEFu); }
    }
  printf("even lanes loaded correctly: %d; odd (OOB) lanes wrote zeros: %d; odd lanes left LDS untouched: %d\n", ok_even, zero_odd, stale_odd);
  printf("lane1 words: %08x %08x %08x %08x\n", r[4], r[5], r[6], r[7]);
  return 0;
}
