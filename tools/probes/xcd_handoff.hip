// Probe: what a workgroup -> workgroup hand-off of 16 KB costs inside one launch, as a function of WHERE the two workgroups sit
// (same XCD = same L2, or different XCDs) and of the cache policy bits on the producer's stores / the consumer's loads.
//   producer: 256 threads x 4 x 16 B stores (policy PW) of an iteration tag -> s_waitcnt vmcnt(0) -> agent-scope flag store
//   consumer: polls the flag (agent-scope atomic load), then times 256 x 4 x 16 B loads (policy PR), counts stale values
// Policy = buffer-instruction aux bits on gfx942/950: 1 = sc0, 16 = sc1, 17 = sc0 sc1, 0 = none.
// Pairs: blocks b and b + 8 (same XCD under the round-robin workgroup -> XCD dispatch) or b and b + 9 (different XCDs); every block
// records its XCC_ID so the placement is verified, not assumed.  The split-K fix-up question: is a same-L2 hand-off cheap?
// hipcc --offload-arch=gfx950 -O3 xcd_handoff.hip -o xcd_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 64;
constexpr int NPAIR = 8;

struct Out { unsigned long long cycles; unsigned stale; unsigned xcc_p, xcc_c; };

template <int PW, int PR>
__global__ __launch_bounds__(256) void k(unsigned* data, unsigned* flags, Out* out, int cross) {
  const int b = blockIdx.x;          // 0..15: 0..7 producers, 8..15 consumers
  const bool producer = b < NPAIR;
  const int pair = producer ? b : (cross ? (b - NPAIR + 1) % NPAIR : b - NPAIR);  // consumer b pairs with producer `pair`
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF;
  unsigned* buf = data + pair * 4096;  // 16 KB
  volatile unsigned* ready = flags + pair * 64;       // producer -> consumer (separate cache lines)
  volatile unsigned* done = flags + pair * 64 + 32;   // consumer -> producer
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 16384, 0x00020000);
  const int tid = threadIdx.x;
  unsigned long long cyc = 0;
  unsigned stale = 0;
  for (int it = 1; it <= ITERS; ++it) {
    if (producer) {
      if (tid == 0) while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(it - 1)) {}
      __syncthreads();
      const u32x4 v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (tid + 256 * i) * 16, 0, PW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(ready, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tid == 0) while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it) {}
      __syncthreads();
      const unsigned long long t0 = wall_clock64();
      u32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 256 * i) * 16, 0, PR);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1 = wall_clock64();
#pragma unroll
      for (int i = 0; i < 4; ++i) stale += (v[i].x != (unsigned)it) + (v[i].w != (unsigned)it);
      cyc += t1 - t0;
      __syncthreads();
      if (tid == 0) __hip_atomic_store(done, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    if (producer) out[pair].xcc_p = xcc;
    else { out[b - NPAIR + NPAIR].cycles = cyc; out[b - NPAIR + NPAIR].xcc_c = xcc; }
  }
  if (!producer) atomicAdd(&out[b].stale, stale);
}

template <int PW, int PR>
void run(unsigned* data, unsigned* flags, Out* out, int cross) {
  hipMemset(flags, 0, NPAIR * 64 * 4);
  hipMemset(out, 0, sizeof(Out) * 2 * NPAIR);
  hipMemset(data, 0, NPAIR * 16384);
  k<PW, PR><<<2 * NPAIR, 256>>>(data, flags, out, cross);
  hipDeviceSynchronize();
  Out h[2 * NPAIR];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double ns = 0; unsigned stale = 0; int same = 0;
  for (int c = 0; c < NPAIR; ++c) {
    const int pair = cross ? (c + 1) % NPAIR : c;
    ns += (double)h[NPAIR + c].cycles * 10.0 / ITERS;  // wall_clock64: 100 MHz
    stale += h[NPAIR + c].stale;
    same += h[pair].xcc_p == h[NPAIR + c].xcc_c;
  }
  printf("store aux %2d  load aux %2d  %s pairs: %d of %d pairs on one XCC   16 KB read %7.0f ns   stale values %u\n", PW, PR, cross ? "b/b+9" : "b/b+8", same, NPAIR,
         ns / NPAIR, stale);
}

int main() {
  unsigned *data, *flags; Out* out;
  hipMalloc(&data, NPAIR * 16384); hipMalloc(&flags, NPAIR * 64 * 4); hipMalloc(&out, sizeof(Out) * 2 * NPAIR);
  for (int cross = 0; cross < 2; ++cross) {
    run<0, 0>(data, flags, out, cross);
    run<0, 1>(data, flags, out, cross);
    run<0, 16>(data, flags, out, cross);
    run<0, 17>(data, flags, out, cross);
    run<16, 0>(data, flags, out, cross);
    run<16, 1>(data, flags, out, cross);
    run<16, 16>(data, flags, out, cross);
    run<17, 17>(data, flags, out, cross);
  }
  return 0;
}
