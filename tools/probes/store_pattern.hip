// Probe: what a GEMM epilogue's store pattern costs.  256 workgroups x 512 threads each write a 256 x 256 f16 tile (128 KB) of a row-major
// [M][N] matrix (N = 512: row stride 1 KB), all at the same moment -- (A) the MFMA-fragment pattern: one store instruction = 32 rows x 32
// contiguous bytes (lanes l, l + 32 adjacent), a wave walks a 64 x 32 sub-tile; (B) whole rows: one instruction = 2 rows x 512 bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern.hip -o tools/probes/bin/store_pattern && tools/probes/bin/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int PAT>
__global__ __launch_bounds__(512) void k(uint4* out, int N, int rounds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint4 v = make_uint4(tid, lane, wave, blockIdx.x);
  for (int r = 0; r < rounds; ++r) {
    const int tile = blockIdx.x + r * gridDim.x;
    const int tiles_n = N / 256;
    const long m0 = (long)(tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
    char* base = reinterpret_cast<char*>(out) + (m0 * N + n0) * 2;
    if (PAT == 0) {
      // wave (wr = wave >> 2, wc = wave & 3): quadrants at (64 wr + 128 qm, 32 wc + 128 qn), each 64 rows x 32 cols = 2 bands x 2 stores
      const int wr = wave >> 2, wc = wave & 3, l31 = lane & 31, hi = lane >> 5;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const long row = 64 * wr + 128 * (q >> 1) + 32 * i + l31;
            const int col = 32 * wc + 128 * ((q & 1) ^ (q >> 1)) + 16 * g + 8 * hi;
            *reinterpret_cast<uint4*>(base + row * N * 2 + col * 2) = v;
          }
    } else if (PAT == 2 || PAT == 3) {
      // round 6: SEG-byte row segments per instruction (PAT 2: 64 B = a wave's own 32 columns after a transpose through LDS, 16 rows per
      // instruction; PAT 3: 128 B = a whole cache line, 8 rows per instruction): wave (wr, wc) still owns the columns 32 wc + 128 qn of rows 64 wr + 128 qm ..
      constexpr int LPR = PAT == 2 ? 4 : 8, RPI = 64 / LPR;   // lanes per row, rows per instruction
      const int wr = wave >> 2, wc = wave & 3;
      // the wave's 4 quadrants x 64 rows x 64 bytes = 16 KB, as 16 instructions of 1 KB
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int idx = s * RPI + lane / LPR;            // 0 .. 16 * RPI - 1: (quadrant-row index) over 256 row-segments of 64 B (PAT 2) / 128 of 128 B (PAT 3)
        const int piece = lane % LPR;
        long row; int colb;
        if (PAT == 2) { const int q = idx >> 6, r = idx & 63; row = 64 * wr + 128 * (q >> 1) + r; colb = (32 * wc + 128 * ((q & 1) ^ (q >> 1))) * 2 + piece * 16; }
        else { const int q = idx >> 5, r = idx & 31; row = 64 * wr + 128 * (q >> 1) + 2 * r + (piece >> 2) * 0 + 0; row += (idx & 0) ; colb = (64 * (wc >> 1) + 128 * ((q & 1) ^ (q >> 1))) * 2 + piece * 16; row = 64 * wr + 128 * (q >> 1) + 32 * (wc & 1) + r; }
        *reinterpret_cast<uint4*>(base + row * N * 2 + colb) = v;
      }
    } else {
      // 256 rows x 512 bytes: instruction s of wave w writes rows 2 (16 w + s) .. + 1, lane = (row & 1) * 32 + 16-byte piece
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const long row = 2 * (16 * wave + s) + (lane >> 5);
        *reinterpret_cast<uint4*>(base + row * N * 2 + (lane & 31) * 16) = v;
      }
    }
  }
}

template <int PAT>
void run(const char* name, uint4* out, int N, int rounds) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<PAT><<<256, 512>>>(out, N, rounds);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    k<PAT><<<256, 512>>>(out, N, rounds);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double bytes = 256.0 * rounds * 131072.0;
  printf("%-44s %2d round(s): %8.1f us  %6.2f TB/s  (%.1f us per round)\n", name, rounds, best * 1e3, bytes / best / 1e9, best * 1e3 / rounds);
}

int main() {
  const int N = 512; const long M = 131072 * 4;
  uint4* out; hipMalloc(&out, M * N * 2);
  for (int rounds : {1, 4, 16}) {
    run<0>("A: MFMA-fragment stores (32 rows x 32 B)", out, N, rounds);
    run<1>("B: whole rows (2 rows x 512 B)", out, N, rounds);
    run<2>("C: 64-byte row segments (16 rows x 64 B)", out, N, rounds);
    run<3>("D: 128-byte row segments (8 rows x 128 B)", out, N, rounds);
  }
  return 0;
}
