// Probe: issue cost (SIMD cycles per wave64 instruction) of the VALU ops the attention softmax is made of, and whether the
// transcendental unit overlaps with ordinary VALU work of the SAME wave or of ANOTHER wave on the SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// One block of 64 (one wave on one SIMD) or 128 x 4... threads; s_memtime around an unrolled loop; cycles = ticks * (core clock / 100 MHz)
// is avoided by reporting RATIOS against v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int MODE>
__global__ void k(float* out, long long* ticks, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
  const float c = 0.999f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent v_exp_f32
      REP4(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 1) {  // 8 independent v_fma_f32
      REP4(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 2) {  // each v_exp followed by 3 independent v_fma on other registers
      REP4(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %16, %16\n v_fma_f32 %5, %5, %16, %16\n v_fma_f32 %6, %6, %16, %16\n"
                        "v_exp_f32 %1, %1\n v_fma_f32 %7, %7, %16, %16\n v_fma_f32 %8, %8, %16, %16\n v_fma_f32 %9, %9, %16, %16\n"
                        "v_exp_f32 %2, %2\n v_fma_f32 %10, %10, %16, %16\n v_fma_f32 %11, %11, %16, %16\n v_fma_f32 %12, %12, %16, %16\n"
                        "v_exp_f32 %3, %3\n v_fma_f32 %13, %13, %16, %16\n v_fma_f32 %14, %14, %16, %16\n v_fma_f32 %15, %15, %16, %16"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));)
    } else if (MODE == 3) {  // 8 v_pk_fma_f32 (2 values each)
      REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                        "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                        : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));)
    } else if (MODE == 4) {  // 8 v_cvt_pk_fp8_f32
      REP4(asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2\n v_cvt_pk_fp8_f32 %3, %4, %5\n v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]\n v_cvt_pk_fp8_f32 %3, %4, %5 op_sel:[0,0,1]\n"
                        "v_cvt_pk_fp8_f32 %6, %1, %2\n v_cvt_pk_fp8_f32 %7, %4, %5\n v_cvt_pk_fp8_f32 %6, %1, %2 op_sel:[0,0,1]\n v_cvt_pk_fp8_f32 %7, %4, %5 op_sel:[0,0,1]"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 5) {  // 8 v_exp_f16
      REP4(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 6) {  // 8 v_pk_fma_f16
      REP4(asm volatile("v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 7) {  // v_cvt_pkrtz_f16_f32
      REP4(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %4, %5\n v_cvt_pkrtz_f16_f32 %6, %1, %2\n v_cvt_pkrtz_f16_f32 %7, %4, %5\n"
                        "v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %4, %5\n v_cvt_pkrtz_f16_f32 %6, %1, %2\n v_cvt_pkrtz_f16_f32 %7, %4, %5"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 8) {  // v_dot2c_f32_f16
      REP4(asm volatile("v_dot2c_f32_f16 %0, %8, %8\n v_dot2c_f32_f16 %1, %8, %8\n v_dot2c_f32_f16 %2, %8, %8\n v_dot2c_f32_f16 %3, %8, %8\n v_dot2c_f32_f16 %4, %8, %8\n v_dot2c_f32_f16 %5, %8, %8\n v_dot2c_f32_f16 %6, %8, %8\n v_dot2c_f32_f16 %7, %8, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 9) {  // v_pk_add_f32
      REP4(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                        : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));)
    } else if (MODE == 10) {  // v_cvt_scalef32_pk_fp8_f16
      REP4(asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %8\n v_cvt_scalef32_pk_fp8_f16 %2, %3, %8\n v_cvt_scalef32_pk_fp8_f16 %4, %1, %8\n v_cvt_scalef32_pk_fp8_f16 %5, %3, %8\n"
                        "v_cvt_scalef32_pk_fp8_f16 %6, %1, %8\n v_cvt_scalef32_pk_fp8_f16 %7, %3, %8\n v_cvt_scalef32_pk_fp8_f16 %0, %1, %8 op_sel:[0,0,0,1]\n v_cvt_scalef32_pk_fp8_f16 %2, %3, %8 op_sel:[0,0,0,1]"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 11) {  // v_max3_f32
      REP4(asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 12) {  // v_cvt_scalef32_pk_fp8_f32
      REP4(asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %8\n v_cvt_scalef32_pk_fp8_f32 %3, %4, %5, %8\n v_cvt_scalef32_pk_fp8_f32 %6, %1, %2, %8\n v_cvt_scalef32_pk_fp8_f32 %7, %4, %5, %8\n"
                        "v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %8 op_sel:[0,0,0,1]\n v_cvt_scalef32_pk_fp8_f32 %3, %4, %5, %8 op_sel:[0,0,0,1]\n v_cvt_scalef32_pk_fp8_f32 %6, %1, %2, %8 op_sel:[0,0,0,1]\n v_cvt_scalef32_pk_fp8_f32 %7, %4, %5, %8 op_sel:[0,0,0,1]"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 13) {  // v_lshl_add_u32
      REP4(asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (threadIdx.x % 64 == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
double run(int threads, const char* what, int per_iter) {
  // whole-chip wall clock: 2048 blocks of `threads` (256 CUs x 8), every SIMD holds threads / 256 waves at a time (up to the CU's limits)
  float* out; long long* ticks;
  const int blocks = 2048, iters = 4000;
  (void)hipMalloc(&out, (size_t)blocks * threads * 4); (void)hipMalloc(&ticks, (size_t)blocks * 16 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(out, ticks, iters);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, ticks, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[4];
  (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
  const double wave_instr = (double)blocks * (threads / 64) * iters * per_iter;
  const double ns_per_simd_instr = ms * 1e6 / (wave_instr / 1024.0);  // 1024 SIMDs
  printf("%-50s %4d threads: %7.3f ns per wave-instruction per SIMD (%5.2f cycles at 2.4 GHz) | s_memtime %7.3f ticks per instruction in wave 0\n", what,
         threads, ns_per_simd_instr, ns_per_simd_instr * 2.4, (double)h[0] / ((double)iters * per_iter));
  (void)hipFree(out); (void)hipFree(ticks);
  return ns_per_simd_instr;
}

int main() {
  // 256 threads = one wave per SIMD; 512 = two, 1024 = four waves per SIMD (the per-wave cost at 4 waves / 4 = the SIMD's issue cost)
  for (int th : {256, 1024}) {
    run<1>(th, "v_fma_f32 x8", 32);
    run<0>(th, "v_exp_f32 x8", 32);
    run<2>(th, "(v_exp_f32 + 3 v_fma_f32) x4  [per instruction]", 64);
    run<3>(th, "v_pk_fma_f32 x8", 32);
    run<4>(th, "v_cvt_pk_fp8_f32 x8", 32);
    run<5>(th, "v_exp_f16 x8", 32);
    run<6>(th, "v_pk_fma_f16 x8", 32);
    run<7>(th, "v_cvt_pkrtz_f16_f32 x8", 32);
    run<8>(th, "v_dot2c_f32_f16 x8", 32);
    run<9>(th, "v_pk_add_f32 x8", 32);
    run<10>(th, "v_cvt_scalef32_pk_fp8_f16 x8", 32);
    run<11>(th, "v_max3_f32 x8", 32);
    run<12>(th, "v_cvt_scalef32_pk_fp8_f32 x8", 32);
    run<13>(th, "v_lshl_add_u32 x8", 32);
  }
  return 0;
}
