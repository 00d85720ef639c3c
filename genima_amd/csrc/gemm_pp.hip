// Ping-pong ("8-phase") MFMA implicit-GEMM for gfx950: the 256 x 256 x 64 block tile of gemm.hip with the K loop rebuilt
// around what bounds it at one workgroup per CU -- the per-K-tile `vmcnt(0)` + barrier drain of the LDS-DMA queue.
//
//   * 8 waves = two groups of four (one wave of each group per SIMD).  Group 1 runs ONE s_barrier behind group 0, so on every
//     SIMD one wave is in its MFMA segment (8 x v_mfma_f32_32x32x16_f16 = one 64 x 32 output quadrant x K 64) while its partner is
//     in its load segment (ds_read_b128 fragment reads + 2 LDS-DMA issues): the matrix pipe never waits for LDS or DMA issue.
//   * A wave owns 64 rows of EACH half of the A tile and 32 rows of each half of the W tile, so a phase reads one A half and / or
//     one W half: quadrant order (A0,W0) (A0,W1) (A1,W1) (A1,W0) needs 12 / 4 / 8 / 0 fragment reads (W0's fragments stay in
//     registers for the fourth phase), and a half-tile of the LDS image is dead one phase after its single read.
//   * Every phase stages one half-tile (128 rows x 128 B = 2 DMA instructions per wave) 1.75 K tiles ahead of its use:
//       phase 0: W1(t+1)   phase 1: A1(t+1)   phase 2: A0(t+2)   phase 3: W0(t+2)        (two 64 KB stages of LDS)
//     and waits with a COUNTED `s_waitcnt vmcnt(8)` (never 0): four half-tiles stay in flight across the barriers.
//     RAW: the wait sits in the load segment of phase P-1, the reads in phase P (every wave has then passed a barrier behind
//     every wave's wait).  WAR: a half-tile is re-staged >= 2 phases after its last read (the lagging group's reads are retired
//     by then).  cdna_hip_programming.md section 5 (T3 + T4 + T5) describes the structure; this is an independent build of it on
//     the 32 x 32 x 16 f16 MFMA with the implicit-GEMM conv gather, swapped operands and the fused epilogues of gemm_common.h.
//   * Restrictions (the planner falls back to the other tiles): K % 64 == 0; conv sources need C1 % 64 == 0 and C2 % 64 == 0 (a K tile
//     lies inside one filter tap and one concat source, so the tap walk is wave-uniform); no GEGLU (a quadrant is one 32-column
//     tile wide); no batched problems.
#include "gemm_common.h"

namespace {

constexpr int PP_STAGE = 65536;  // bytes per LDS stage: A tile 256 x 128 B, then W tile 256 x 128 B
constexpr int PP_HALF = 16384;   // one half-tile (128 rows)

// ABL: 0 = the kernel (the only instantiation of a normal build).  Probe builds (-DGN_PP_ABLATIONS, env GN_PP_ABL, tools/probes/gemm_pp_abl.py)
// add ablations that say what bounds the K loop (results are wrong):
//   1 no DMA, 2 no fragment reads, 3 no MFMA, 4 no s_setprio, 5 no barriers, 6 no epilogue
template <bool CONV, int ABL>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmParams pin) {
  const GemmParams p = batch_offset(pin);  // blockIdx.z: the four phase convs of an upsampling conv (round 5) -- weights, padding, output offset
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PP_STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // wr = ping-pong group = which 64 rows of each A half; wc = which 32 rows of each W half
  const int l31 = lane & 31, hi = lane >> 5;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // tile order inside an XCD's contiguous run: row-major (the tiles of one A row band side by side: they share the band in L2) -- or, when the
  // WEIGHT is the big operand (few rows under a long K: the 8x8 / 16x16 latent levels), column-major, so that the row tiles of one weight
  // column tile run on ONE XCD and the tile is fetched from HBM once instead of once per L2
  const int tile_n = p.cm_tiles ? bid / p.tiles_m : bid % p.tiles_n, tile_m = p.cm_tiles ? bid % p.tiles_m : bid / p.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int z = blockIdx.y;
  const int kbeg = z * p.kper;
  const int kend = min(p.K, kbeg + p.kper);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- loader state: lane -> (row inside an 8-row DMA group, swizzled logical chunk) ------------------------------------------
  const int lr = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  const int kl = chunk * 8;  // this lane's K offset inside a tile

  const int Cin = p.kapp ? p.C1 : p.C1 + p.C2;  // channels under each filter tap (k_append: the second source is not under the taps)
  const int Hin = p.ups ? 2 * p.H : p.H, Win = p.ups ? 2 * p.W : p.W;
  // Out-of-range lanes OR kOOB into their 16-byte aligned offset (= exactly kOOB): pure ALU, because a select on a per-lane
  // condition invites hipcc to branch around the DMA -- and a counted vmcnt needs the SAME number of VMEM instructions on every path.
  int iy0[2][2], ix0[2][2], pbase[2][2], pix[2][2];  // conv, [half][piece]
  unsigned aoff[2][2], amask[2][2];                  // dense: row byte offset, kOOB mask of rows >= M
  unsigned woff[2][2], wmask[2][2];
  int kA[2] = {kbeg, kbeg}, kW[2] = {kbeg, kbeg};    // K origin of the next tile each half-tile stream stages (wave-uniform)
  int ccA[2] = {0, 0}, dyA[2] = {0, 0}, dxA[2] = {0, 0};

  auto set_tap = [&](int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = iy0[h][i] + dyA[h], ix = ix0[h][i] + dxA[h];
      const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
      const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
      pix[h][i] = ok ? pbase[h][i] + sy * p.W + sx : -1;
    }
  };

#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 128 * h + 8 * (wave + 8 * i) + lr;
      const int m = m0 + row;
      if constexpr (CONV) {
        const int hw = p.Ho * p.Wo;
        if (m < p.M) {
          const int b = m / hw, rem = m - b * hw;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          iy0[h][i] = oy * p.stride - p.pad_t;
          ix0[h][i] = ox * p.stride - p.pad_l;
          pbase[h][i] = b * p.H * p.W;
        } else {
          iy0[h][i] = -(1 << 28);
          ix0[h][i] = -(1 << 28);
          pbase[h][i] = 0;
        }
      } else {
        aoff[h][i] = (m < p.M) ? (unsigned)((long)m * p.lda * 2) : 0u;
        amask[h][i] = (m < p.M) ? 0u : kOOB;
      }
      const int n = n0 + row;
      woff[h][i] = (n < p.N) ? (unsigned)((long)n * p.ldw * 2) : 0u;
      wmask[h][i] = (n < p.N) ? 0u : kOOB;
    }
    if constexpr (CONV) {
      const int tap = kbeg / Cin;
      ccA[h] = kbeg - tap * Cin;
      dyA[h] = tap / p.KW;
      dxA[h] = tap - dyA[h] * p.KW;
      set_tap(h);
    }
  }

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, (int)p.a2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a3 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a3 ? p.a3 : p.a), 0, (int)p.a3_bytes, 0x00020000);  // k_append: second appended source

  // stage the next tile of A half `h` into LDS stage `buf` (2 DMA instructions) and advance that stream by one K tile
  auto stage_a = [&](int h, int buf) {
    if constexpr (ABL == 1) return;
    unsigned char* dst = smem + buf * PP_STAGE + h * PP_HALF + wave * 1024;
    if constexpr (CONV) {
      const unsigned kmask = kA[h] < kend ? 0u : kOOB;  // wave-uniform (K % 64 == 0)
      if (p.kapp && kA[h] >= p.kapp_k0) {  // the appended 1x1 segment of a k_append conv: the centre pixel of the second source
        int cs, cbase;
        const bool s2 = kapp_src(p, kA[h], cs, cbase);
        const int co = kA[h] - cbase + kl;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned voff = kapp_voff(p, iy0[h][i], ix0[h][i], pbase[h][i], cs, co) | kmask;
          if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a3, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
        }
        kA[h] += BK;
        return;
      }
      const bool first = p.kapp || ccA[h] < p.C1;
      const int cs = first ? p.C1 : p.C2;
      const int co = (first ? ccA[h] : ccA[h] - p.C1) + kl;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned voff = ((unsigned)(pix[h][i] * cs + co) * 2u) | ((unsigned)(pix[h][i] >> 31) & kOOB) | kmask;
        if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
      }
      kA[h] += BK;
      ccA[h] += BK;
      if (ccA[h] >= Cin) {
        ccA[h] = 0;
        if (++dxA[h] == p.KW) { dxA[h] = 0; ++dyA[h]; }
        set_tap(h);
      }
    } else {
      const unsigned kmask = kA[h] < kend ? 0u : kOOB;
      if (p.kapp && kA[h] >= p.kapp_k0) {  // dense k_append: the second operand's columns
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = min(m0 + 128 * h + 8 * (wave + 8 * i) + lr, p.M - 1);
          const unsigned voff = ((unsigned)((long)m * p.lda2 * 2) + (unsigned)(kA[h] - p.kapp_k0 + kl) * 2u) | amask[h][i] | kmask;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned voff = (aoff[h][i] + (unsigned)(kA[h] + kl) * 2u) | amask[h][i] | kmask;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
        }
      }
      kA[h] += BK;
    }
  };
  auto stage_w = [&](int h, int buf) {
    if constexpr (ABL == 1) return;
    unsigned char* dst = smem + buf * PP_STAGE + 2 * PP_HALF + h * PP_HALF + wave * 1024;
    const unsigned kmask = kW[h] < kend ? 0u : kOOB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned voff = (woff[h][i] + (unsigned)(kW[h] + kl) * 2u) | wmask[h][i] | kmask;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
    }
    kW[h] += BK;
  };

  // fragment read offsets inside a half-tile: rows wr*64 + mt*32 + l31 (A) / wc*32 + l31 (W); the swizzle depends on l31 only
  int a_rd[4], w_rd[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int c = ((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
    a_rd[kk] = (wr * 64 + l31) * 128 + c;
    w_rd[kk] = (wc * 32 + l31) * 128 + c;
  }

  auto rd = [&](const unsigned char* ptr) -> f16x8 {
    if constexpr (ABL == 2) {
      f16x8 v = {(f16)1.0f, (f16)0.5f, (f16)0.25f, (f16)-1.0f, (f16)1.0f, (f16)0.5f, (f16)0.25f, (f16)-1.0f};
      asm volatile("" : "+v"(v) : "v"(ptr));
      return v;
    } else {
      return *reinterpret_cast<const f16x8*>(ptr);
    }
  };
  auto mma = [&](const f16x8& w, const f16x8& a, f32x16& c) {
    if constexpr (ABL == 3) asm volatile("" : "+v"(c) : "v"(w), "v"(a));
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0);
  };
  auto prio = [&](int x) {
    if constexpr (ABL != 4) { if (x) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
  };
  auto bar = [&]() {
    if constexpr (ABL != 5) __builtin_amdgcn_s_barrier();
  };
  auto wait_dma8 = [&]() {
    if constexpr (ABL != 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };

  f32x16 acc[4][1][2];  // [quadrant][TN = 1][TM = 2]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][0][i][r] = 0.0f;

  // ---- prologue: all of tile 0, then A0 / W0 of tile 1 (the order the steady-state vmcnt counts assume) -------------------------
  stage_a(0, 0);
  stage_w(0, 0);
  stage_w(1, 0);
  stage_a(1, 0);
  stage_a(0, 1);
  stage_w(0, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) bar();  // group 1 runs one barrier behind from here on
  __builtin_amdgcn_sched_barrier(0);

  f16x8 fa[2][4], fw0[4], fw1[4];

  for (int t = 0; t < nk; ++t) {
    const int b = t & 1;
    const unsigned char* Ab = smem + b * PP_STAGE;
    const unsigned char* Wb = Ab + 2 * PP_HALF;

    // ---------------- phase 0: quadrant (A0, W0) ----------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fw0[kk] = rd(Wb + w_rd[kk]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt][kk] = rd(Ab + a_rd[kk] + mt * 4096);
    stage_w(1, b ^ 1);
    wait_dma8();
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
    prio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        mma(fw0[kk], fa[mt][kk], acc[0][0][mt]);
    prio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase 1: quadrant (A0, W1) ----------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fw1[kk] = rd(Wb + PP_HALF + w_rd[kk]);
    stage_a(1, b ^ 1);
    wait_dma8();
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
    prio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        mma(fw1[kk], fa[mt][kk], acc[1][0][mt]);
    prio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase 2: quadrant (A1, W1) ----------------
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt][kk] = rd(Ab + PP_HALF + a_rd[kk] + mt * 4096);
    stage_a(0, b);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
    prio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        mma(fw1[kk], fa[mt][kk], acc[2][0][mt]);
    prio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- phase 3: quadrant (A1, W0): no fragment reads ----------------
    stage_w(0, b);
    wait_dma8();
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
    prio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        mma(fw0[kk], fa[mt][kk], acc[3][0][mt]);
    prio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wr == 0) bar();  // pairs with group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead (out-of-range, zero-fill) stages

  const int mb = m0 + 64 * wr, nb = n0 + 32 * wc;
  if constexpr (ABL == 6) {  // (timing only) no epilogue: one store per lane keeps the accumulators alive
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][0][i][r];
    if (s == 12345.678f) p.out[(long)mb * p.ldo + nb + l31] = (f16)s;
    return;
  }
  gemm_epilogue<2, 1>(p, acc[0], mb, nb, l31, hi, z);
  gemm_epilogue<2, 1>(p, acc[1], mb, nb + 128, l31, hi, z);
  gemm_epilogue<2, 1>(p, acc[2], mb + 128, nb + 128, l31, hi, z);
  gemm_epilogue<2, 1>(p, acc[3], mb + 128, nb, l31, hi, z);
  gemm_sink_tail<512, 256, 256, 2 * PP_STAGE>(p, m0, n0, smem);  // (the 256 x 256 tile does not fit beside its scratch: re-read from L2)
}

}  // namespace

void gn_launch_gemm_pp(const void* params, bool conv, int grid_x, int grid_y, int grid_z, hipStream_t st) {
  const GemmParams& p = *static_cast<const GemmParams*>(params);
  const dim3 grid(grid_x, grid_y, grid_z);
#ifdef GN_PP_ABLATIONS  // probe builds only (GN_HIPCC_EXTRA=-DGN_PP_ABLATIONS, tools/probes/gemm_pp_abl.py): the ablated variants give wrong results
  static const int abl = [] { const char* e = getenv("GN_PP_ABL"); return e ? atoi(e) : 0; }();
  switch (abl) {
#define GN_PP_CASE(A)                                                                                  \
  case A:                                                                                              \
    if (conv) hipLaunchKernelGGL((gemm_pp_kernel<true, A>), grid, dim3(512), 0, st, p);               \
    else hipLaunchKernelGGL((gemm_pp_kernel<false, A>), grid, dim3(512), 0, st, p);                    \
    return;
    GN_PP_CASE(1) GN_PP_CASE(2) GN_PP_CASE(3) GN_PP_CASE(4) GN_PP_CASE(5) GN_PP_CASE(6)
    default: break;
#undef GN_PP_CASE
  }
#endif
  if (conv) hipLaunchKernelGGL((gemm_pp_kernel<true, 0>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((gemm_pp_kernel<false, 0>), grid, dim3(512), 0, st, p);
}
