// Probe: how many bytes per second one CU can pull through the TA path into LDS (buffer_load ... lds) or VGPRs
// (buffer_load_dwordx4), L2-resident data, with and without a concurrent MFMA stream -- the question behind the GEMM K loop
// (gemm_pp.hip): is a 256x256x64 tile's 64 KB per K tile DMA-bound?    hipcc --offload-arch=gfx950 -O3 dma_bw.hip -o dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: LDS-DMA, 8 rows x 128 B per wave instruction (row stride `rs` bytes)    MODE 1: LDS-DMA, 1 KB contiguous per instruction
// MODE 2: buffer_load_dwordx4 to VGPRs, 8 rows x 128 B                          MODE 3: mode 0 on waves 0-3, MFMA stream on waves 4-7
// MODE 4: MFMA stream only on waves 4-7 (waves 0-3 idle)                        MODE 5: mode 0 with all 8 waves + MFMAs interleaved per wave
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const unsigned char* src, size_t bytes_per_block, int rs, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (size_t)blockIdx.x * bytes_per_block;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes_per_block, 0x00020000);
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  asm volatile("" : "+v"(fa), "+v"(fb));
  uint4 keep = make_uint4(0, 0, 0, 0);
  const bool loader = (MODE == 3) ? wave < 4 : (MODE == 4 ? false : true);
  const bool mfma = (MODE == 3 || MODE == 4) ? wave >= 4 : (MODE == 5);
  const int nload_waves = (MODE == 3) ? 4 : 8;
  // every loader wave moves 16 instructions (16 KB) per iteration: 8 waves -> 128 KB ... cycled over the block's region
  unsigned off = 0;
  for (int it = 0; it < iters; ++it) {
    if (loader) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        unsigned voff;
        if (MODE == 1) {
          voff = off + (unsigned)((wave * 16 + j) * 1024 + lane * 16);
        } else {
          const int row = (wave * 16 + j) * 8 + (lane >> 3);
          voff = off + (unsigned)(row * rs + (lane & 7) * 16);
        }
        voff %= (unsigned)bytes_per_block;
        voff &= ~15u;
        if (MODE == 2) {
          uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff, 0, 0));
          keep.x ^= v.x; keep.y ^= v.y; keep.z ^= v.z; keep.w ^= v.w;
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + ((wave * 16 + j) & 127) * 1024), 16, voff, 0, 0, 0);
          if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          if (DEPTH == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        }
        if (MODE == 5) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
          }
        }
      }
      off += (unsigned)(nload_waves * 16 * 8 * rs);
    }
    if (mfma && MODE != 5) {
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = acc0[0] + acc1[5] + (float)(keep.x ^ keep.y ^ keep.z ^ keep.w);
  if (s == 12345.678f) sink[tid] = s + smem[tid];
}

// GEMM-like mix per wave: per K tile 8 DMA + 24 ds_read_b128 + 32 MFMA (the 256x256x64 tile's budget), no data dependence.
// SYNC 0: no barrier   1: s_barrier per K tile (counted vmcnt(8))   2: vmcnt(0) + s_barrier per K tile (the drain)
// 3: s_barrier every 8 MFMAs (8-phase cadence, both groups in lockstep)     GROUPED 1: the 8 DMAs issued together at the tile top
template <int SYNC, int GROUPED>
__global__ __launch_bounds__(512) void mix(const unsigned char* src, size_t bytes_per_block, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (size_t)blockIdx.x * bytes_per_block;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes_per_block, 0x00020000);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x16{0};
  const int rd0 = ((lane & 31) * 128 + (((lane >> 5)) ^ ((lane >> 1) & 7)) * 16);
  unsigned off = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* S = smem + (it & 1) * 65536;
    if (GROUPED) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned voff = (off + (unsigned)(((wave * 8 + j) * 8 + (lane >> 3)) * 1024 + (lane & 7) * 16)) % (unsigned)bytes_per_block;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + ((it + 1) & 1) * 65536 + (wave * 8 + j) * 1024), 16, voff & ~15u, 0, 0, 0);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (!GROUPED) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned voff = (off + (unsigned)(((wave * 8 + kk * 2 + j) * 8 + (lane >> 3)) * 1024 + (lane & 7) * 16)) % (unsigned)bytes_per_block;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + ((it + 1) & 1) * 65536 + (wave * 8 + kk * 2 + j) * 1024), 16, voff & ~15u, 0, 0, 0);
        }
      }
      f16x8 fa[4], fw[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const f16x8*>(S + rd0 + ((wave >> 2) * 128 + i * 32) * 128 + kk * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = *reinterpret_cast<const f16x8*>(S + 32768 + rd0 + ((wave & 3) * 64 + j * 32) * 128 + kk * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[j * 4 + i], 0, 0, 0);
      if (SYNC == 3) __builtin_amdgcn_s_barrier();
    }
    if (SYNC == 1) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    if (SYNC == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    off += 65536;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += acc[i][3];
  if (t == 12345.678f) sink[tid] = t;
}

template <int SYNC, int GROUPED>
void run_mix(const char* name, const unsigned char* src, size_t bpb, float* sink) {
  const int iters = 400, blocks = 256;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((mix<SYNC, GROUPED>), dim3(blocks), dim3(512), 0, 0, src, bpb, 20, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((mix<SYNC, GROUPED>), dim3(blocks), dim3(512), 0, 0, src, bpb, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = (double)iters * 8 * 32 * 32 * 32 * 16 * 2 * blocks;
  printf("mix %-52s: %7.1f us  MFMA %7.1f TF   (%5.0f ns per K tile)\n", name, ms * 1e3, flops / ms / 1e9, ms * 1e6 / iters);
}

template <int MODE, int DEPTH>
void run(const char* name, const unsigned char* src, size_t bpb, int rs, float* sink) {
  const int iters = 200, blocks = 256;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(blocks), dim3(512), 0, 0, src, bpb, rs, 20, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(blocks), dim3(512), 0, 0, src, bpb, rs, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const int lw = (MODE == 3) ? 4 : (MODE == 4 ? 0 : 8);
  const double bytes = (double)iters * lw * 16 * 1024 * blocks;
  const int mw = (MODE == 3 || MODE == 4) ? 4 : (MODE == 5 ? 8 : 0);
  const double flops = (double)iters * mw * 64 * 32 * 32 * 16 * 2 * blocks;
  printf("%-46s rs=%5d depth=%2d: %7.1f us  %7.2f TB/s chip = %6.1f GB/s per CU   MFMA %7.1f TF\n", name, rs, DEPTH, ms * 1e3, bytes / ms / 1e9,
         bytes / ms / 1e6 / 256, flops / ms / 1e9);
}

int main() {
  const size_t bpb = 1 << 20;  // 1 MB per block: 256 MB total = MALL-resident, 32 MB per XCD > L2 (4 MB)
  unsigned char* src;
  float* sink;
  hipMalloc(&src, bpb * 256);
  hipMemset(src, 1, bpb * 256);
  hipMalloc(&sink, 4096);
  for (int pass = 0; pass < 2; ++pass) {
    const size_t use = pass == 0 ? (96 << 10) : bpb;  // 96 KB per block: 3 MB per XCD = L2-resident
    printf("---- footprint per block %zu KB\n", use >> 10);
    run<0, 8>("DMA 8 rows x 128 B", src, use, 1024, sink);
    run<0, 16>("DMA 8 rows x 128 B", src, use, 1024, sink);
    run<0, 32>("DMA 8 rows x 128 B", src, use, 1024, sink);
    run<0, 16>("DMA 8 rows x 128 B", src, use, 128, sink);
    run<1, 16>("DMA 1 KB contiguous", src, use, 128, sink);
    run<2, 16>("VGPR loads 8 rows x 128 B", src, use, 1024, sink);
    run<2, 16>("VGPR loads 8 rows x 128 B", src, use, 128, sink);
    run<3, 16>("DMA waves 0-3 || MFMA waves 4-7", src, use, 1024, sink);
    run<4, 16>("MFMA waves 4-7 only", src, use, 1024, sink);
    run<5, 16>("DMA + 8 MFMA per DMA, all 8 waves", src, use, 1024, sink);
  }
  for (int pass = 0; pass < 2; ++pass) {
    const size_t use = pass == 0 ? (128 << 10) : bpb;
    printf("---- mix, footprint per block %zu KB (zero-ish data: all bytes 1)\n", use >> 10);
    run_mix<0, 0>("no sync, DMA spread (2 per 8 MFMA)", src, use, sink);
    run_mix<0, 1>("no sync, 8 DMA grouped at tile top", src, use, sink);
    run_mix<1, 0>("vmcnt(8)+barrier per tile, DMA spread", src, use, sink);
    run_mix<1, 1>("vmcnt(8)+barrier per tile, DMA grouped", src, use, sink);
    run_mix<2, 0>("vmcnt(0)+barrier per tile, DMA spread", src, use, sink);
    run_mix<2, 1>("vmcnt(0)+barrier per tile, DMA grouped (= d256 loop)", src, use, sink);
    run_mix<3, 0>("barrier per 8 MFMA, DMA spread", src, use, sink);
  }
  return 0;
}
