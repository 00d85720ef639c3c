// MFMA implicit-GEMM convolution + Linear for gfx950 (CDNA4).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
//
// One kernel family serves every Conv2d (3x3/1x1/7x7, stride 1/2, fused nearest-2x upsample, virtual channel concat)
// and every Linear on the Genima hot path:
//   * activations are NHWC f16, so a conv's A operand is an on-the-fly gather of 16-byte channel chunks
//     (tap-major K = KH*KW*Cin) and a Linear's A operand is the same loader with a dense row stride;
//   * weights are [N][K] f16 (K contiguous) for both;
//   * BMxBNx64 block tile, 4 or 8 waves, v_mfma_f32_32x32x16_f16 with f32 accumulation;
//   * global -> registers -> LDS staging, double-buffered, next tile's loads in flight under the current tile's MFMAs
//     (one barrier per K tile); LDS rows are 128 B with the XOR swizzle of common.h (conflict-free ds_read_b128);
//   * the conv gather keeps, per staged row, the source pixel of the CURRENT filter tap (or -1 when the tap falls in the
//     padding); it is recomputed only when the K walk crosses into the next tap (every Cin/64 tiles), so the per-tile cost
//     of a gathered load is one multiply-add -- the same as the dense loader;
//   * operands are swapped (D = W_tile * A_tile^T) so each lane ends up with 4 consecutive output channels of one output
//     row: 8-byte bias / residual loads and 8-byte stores in the epilogue;
//   * epilogue fuses bias, per-batch time-embedding shift, SiLU/GELU/QuickGELU/ReLU/GEGLU, scale and residual add;
//   * optional split-K over gridDim.y with an f32 workspace and a fused reduce+epilogue kernel (small-M layers);
//   * blockIdx -> tile mapping is XCD-aware (each XCD walks a contiguous run of tiles, N fastest, so the A rows an XCD's L2
//     holds are reused across the N tiles).
#include <type_traits>

#include "gemm_common.h"

namespace {

template <int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__(WM* WN * 64, gemm_waves_per_simd(2 * (BM + BN) * 128, WM* WN)) void gemm_kernel(const GemmParams pin) {
  const GemmParams p = batch_offset(pin);
  constexpr int NT = WM * WN * 64;
  constexpr int RPP = NT / 8;  // tile rows staged per pass (8 lanes x 16 B cover one 128-byte row)
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile >= 32x32");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  constexpr int RA = BM / RPP, RB = BN / RPP;  // 16-byte chunks per thread per tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware, bijective block -> tile remap (block b runs on XCD b % 8; give each XCD a contiguous run of tiles)
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
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int kbeg = z * p.kper;
  const int kend = min(p.K, kbeg + p.kper);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- loader state -------------------------------------------------------------------------------------------------
  const int chunk = tid & 7;
  const int row0 = tid >> 3;
  int kcur = kbeg + chunk * 8;

  const int Cin = p.C1 + p.C2;
  const int Hin = p.ups ? 2 * p.H : p.H, Win = p.ups ? 2 * p.W : p.W;
  int iy0[RA], ix0[RA], pbase[RA];  // conv: output-pixel origin and batch pixel base of each staged row
  int pix[RA];                      // conv: source pixel of the current tap (-1 = padding / row out of range)
  const f16* arow[RA];              // dense: row pointers
  int cc = 0, dy = 0, dx = 0;

  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
      const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
      pix[i] = ok ? pbase[i] + sy * p.W + sx : -1;
    }
  };

  if constexpr (CONV) {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + row0 + RPP * i;
      if (m < p.M) {
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        iy0[i] = oy * p.stride - p.pad_t;
        ix0[i] = ox * p.stride - p.pad_l;
        pbase[i] = b * p.H * p.W;
      } else {
        iy0[i] = -(1 << 28);
        ix0[i] = -(1 << 28);
        pbase[i] = 0;
      }
    }
    const int tap = kcur / Cin;
    cc = kcur - tap * Cin;
    dy = tap / p.KW;
    dx = tap - dy * p.KW;
    set_tap();
  } else {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + row0 + RPP * i;
      arow[i] = (m < p.M) ? p.a + (long)m * p.lda : nullptr;
    }
  }
  const f16* wrow[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + row0 + RPP * i;
    wrow[i] = (n < p.N) ? p.w + (long)n * p.ldw : nullptr;
  }

  uint4 ra[RA], rb[RB];

  auto load_tile = [&]() {
    const bool kok = kcur < kend;
    if constexpr (CONV) {
      const bool first = cc < p.C1;
      const f16* src = first ? p.a : p.a2;
      const int cs = first ? p.C1 : p.C2;
      const int co = first ? cc : cc - p.C1;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (kok && pix[i] >= 0) v = *reinterpret_cast<const uint4*>(src + (long)pix[i] * cs + co);
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (kok && arow[i]) v = *reinterpret_cast<const uint4*>(arow[i] + kcur);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kok && wrow[i]) v = *reinterpret_cast<const uint4*>(wrow[i] + kcur);
      rb[i] = v;
    }
    // advance to the next K tile
    kcur += BK;
    if constexpr (CONV) {
      cc += BK;
      if (cc >= Cin) {
        do {
          cc -= Cin;
          if (++dx == p.KW) { dx = 0; ++dy; }
        } while (cc >= Cin);
        set_tap();
      }
    }
  };

  auto store_tile = [&](int buf) {
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<uint4*>(As + lds_swz<128>(row0 + RPP * i, chunk)) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<uint4*>(Bs + lds_swz<128>(row0 + RPP * i, chunk)) = rb[i];
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  load_tile();
  store_tile(0);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) load_tile();
    const unsigned char* As = smem + cur * (A_BYTES + B_BYTES);
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 fa[TM], fw[TN];
      const int c = kk * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const f16x8*>(As + lds_swz<128>(wm * WTM + i * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fw[j] = *reinterpret_cast<const f16x8*>(Bs + lds_swz<128>(wn * WTN + j * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[j][i], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  constexpr int SMEM = 2 * (A_BYTES + B_BYTES);
  gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi, z, gemm_sink_lds<BM, BN, SMEM>(p, m0, n0, smem));
  gemm_sink_tail<NT, BM, BN, SMEM>(p, m0, n0, smem);
}

// ======================================================================================================================
// LDS-DMA variant: the same tile / MFMA / epilogue structure, but the K tiles go global -> LDS directly with
// `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass).  One wave-instruction moves 64 lanes x 16 B = eight
// 128-byte tile rows; LDS-DMA destinations are lane-linear, so the XOR swizzle is applied on the SOURCE side: lane q fetches
// the logical chunk (q & 7) ^ ((row >> 1) & 7) of row 8*group + (q >> 3), which lands at physical chunk q & 7 -- exactly the
// image lds_swz<128> reads back.  Out-of-range lanes (conv padding, M/N/K tails) use an out-of-bounds buffer offset: the
// hardware writes ZEROS to LDS for them (tools/probes/lds_dma_probe.hip verifies this on gfx950), so the conv zero padding
// costs nothing.  Freed registers allow 128x64 wave tiles (256x256 block, 8 waves).

template <int BM, int BN, int WM, int WN, bool CONV, bool LNF = false>
__global__ __launch_bounds__(WM* WN * 64, gemm_waves_per_simd(2 * (BM + BN) * 128, WM* WN)) void gemm_dma_kernel(const GemmParams pin) {
  static_assert(!LNF || (!CONV && 4 % WN == 0), "LayerNorm fold: dense problems, K steps dealt over 1 / 2 / 4 column waves");
  const GemmParams p = batch_offset(pin);
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile >= 32x32");
  static_assert(NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row DMA groups per wave");
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;  // DMA instructions per wave per tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES) + (LNF ? BN * 4 : 0)];  // + the workgroup's c1 values (LNF)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (LDS-DMA bases live in M0)
  const int wm = wave / WN, wn = wave % WN;
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
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int kbeg = z * p.kper;
  const int kend = min(p.K, kbeg + p.kper);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- loader state: this lane's row inside each 8-row group and its (swizzled) logical chunk -----------------------------
  const int lr = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  int kcur = kbeg + chunk * 8;
  int kt0 = kbeg;  // wave-uniform K origin of the next tile to stage (k_append: which segment it lies in)

  const int Cin = p.kapp ? p.C1 : p.C1 + p.C2;  // channels under each filter tap (k_append: the second source is not under the taps)
  const int Hin = p.ups ? 2 * p.H : p.H, Win = p.ups ? 2 * p.W : p.W;
  int iy0[GA], ix0[GA], pbase[GA], pix[GA];
  unsigned aoff[GA];  // dense: byte offset of the row start (kOOB if the row is out of range)
  int cc = 0, dy = 0, dx = 0;
  int cu = 0;         // wave-uniform channel offset of the tile inside its tap (selects the concat source)

  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
      const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
      pix[i] = ok ? pbase[i] + sy * p.W + sx : -1;
    }
  };

  if constexpr (CONV) {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int m = m0 + 8 * (wave + NW * i) + lr;
      if (m < p.M) {
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        iy0[i] = oy * p.stride - p.pad_t;
        ix0[i] = ox * p.stride - p.pad_l;
        pbase[i] = b * p.H * p.W;
      } else {
        iy0[i] = -(1 << 28);
        ix0[i] = -(1 << 28);
        pbase[i] = 0;
      }
    }
    const int tap = kcur / Cin;
    cc = kcur - tap * Cin;
    dy = tap / p.KW;
    dx = tap - dy * p.KW;
    cu = kbeg % Cin;
    set_tap();
  } else {
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int m = m0 + 8 * (wave + NW * i) + lr;
      aoff[i] = (m < p.M) ? (unsigned)((long)m * p.lda * 2) : kOOB;
    }
  }
  unsigned woff[GB];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int n = n0 + 8 * (wave + NW * i) + lr;
    woff[i] = (n < p.N) ? (unsigned)((long)n * p.ldw * 2) : kOOB;
  }

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, (int)p.a2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a3 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a3 ? p.a3 : p.a), 0, (int)p.a3_bytes, 0x00020000);  // k_append: second appended source

  auto dma_tile = [&](int buf) {
    const bool kok = kcur < kend;
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
    if constexpr (CONV) {
      if (p.kapp && kt0 >= p.kapp_k0) {  // wave-uniform: the appended 1x1 segment (C1 % 64 == 0: a K tile lies on one side)
        int cs, cbase;
        const bool s2 = kapp_src(p, kt0, cs, cbase);
        const int co = kcur - cbase;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          unsigned voff = kok ? kapp_voff(p, iy0[i], ix0[i], pbase[i], cs, co) : kOOB;
          GN_PIN(voff);
          lds_ptr_t dst = (lds_ptr_t)(As + (wave + NW * i) * 1024);
          if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, dst, 16, voff, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a3, dst, 16, voff, 0, 0, 0);
        }
      } else {
        const bool first = p.kapp || cu < p.C1;  // wave-uniform: with two sources C1 % 64 == 0, so a K tile never straddles them
        const int cs = first ? p.C1 : p.C2;
        const int co = first ? cc : cc - p.C1;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          unsigned voff = (kok && pix[i] >= 0) ? (unsigned)(((long)pix[i] * cs + co) * 2) : kOOB;
          GN_PIN(voff);
          lds_ptr_t dst = (lds_ptr_t)(As + (wave + NW * i) * 1024);
          if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, dst, 16, voff, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, dst, 16, voff, 0, 0, 0);
        }
      }
    } else if (!LNF && p.kapp && kt0 >= p.kapp_k0) {  // dense k_append: the second operand's columns (wave-uniform: kapp_k0 % 64 == 0)
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const int m = m0 + 8 * (wave + NW * i) + lr;
        unsigned voff = (kok && m < p.M) ? (unsigned)((long)m * p.lda2 * 2) + (unsigned)(kcur - p.kapp_k0) * 2u : kOOB;
        GN_PIN(voff);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(As + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        unsigned voff = (kok && aoff[i] != kOOB) ? aoff[i] + (unsigned)kcur * 2u : kOOB;
        GN_PIN(voff);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(As + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      unsigned voff = (kok && woff[i] != kOOB) ? woff[i] + (unsigned)kcur * 2u : kOOB;
      GN_PIN(voff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
    }
    kcur += BK;
    kt0 += BK;
    if constexpr (CONV) {
      cu += BK;
      while (cu >= Cin) cu -= Cin;
      cc += BK;
      if (cc >= Cin) {
        do {
          cc -= Cin;
          if (++dx == p.KW) { dx = 0; ++dy; }
        } while (cc >= Cin);
        set_tap();
      }
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  // registers: accumulators + every bias vector + two bands of shift / residual vectors + working set, against this occupancy's budget
  constexpr int kBudget = 512 / gemm_waves_per_simd(2 * (A_BYTES + B_BYTES), NW);
  constexpr bool RICH = epi_rich_fits(TM, TN, kBudget);
  constexpr bool PRE = RICH && epi_prefetch_fits(TM, TN, kBudget);
  EpiPre<TM, TN> pre;
  bool use_pre = false;
  if constexpr (PRE) use_pre = epilogue_prefetch<TM, TN>(p, pre, m0 + wm * WTM, n0 + wn * WTN, l31, hi);  // lands under the K loop

  if constexpr (LNF) ln_c1_to_lds<BN>(p, reinterpret_cast<float*>(smem + 2 * (A_BYTES + B_BYTES)), n0);
  dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  LnStats<TM> lnst;  // LNF: row sums / sums of squares of the raw A rows (gemm_common.h ln_fold_apply)
  if constexpr (LNF) ln_stats_init(lnst);

  int cur = 0;
  // wsel: which of the WN column waves' share of the K steps this copy of the loop takes the LayerNorm statistics on (LNF).  The choice is
  // made ONCE, outside the K loop (one specialised copy of the loop per column wave): conditional branches inside it cost issue slots even when
  // they fall through (measured on the whole call: DESIGN.md, round 3); the loop body itself has the back edge and nothing else.
  auto k_tile = [&](auto wsel_c) __attribute__((always_inline)) {
    constexpr int WSEL = decltype(wsel_c)::value;
    const unsigned char* As = smem + cur * (A_BYTES + B_BYTES);
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 fa[TM], fw[TN];
      const int c = kk * 2 + hi;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const f16x8*>(As + lds_swz<128>(wm * WTM + i * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fw[j] = *reinterpret_cast<const f16x8*>(Bs + lds_swz<128>(wn * WTN + j * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[j][i], 0, 0, 0);
      if constexpr (LNF) {
        if (WN == 1 || (kk % WN) == WSEL) ln_stats_step(lnst, fa);  // compile-time: this copy's share of the K steps
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of the next tile have landed
    __syncthreads();
    cur ^= 1;
  };
  // the last tile is peeled so that the loop body issues its DMA unconditionally
  auto k_loop = [&](auto wsel_c) __attribute__((always_inline)) {
    for (int kt = 0; kt + 1 < nk; ++kt) {
      dma_tile(cur ^ 1);  // lands under this tile's MFMAs; buf[cur^1] was last read before the previous barrier
      k_tile(wsel_c);
    }
    k_tile(wsel_c);
  };
  if constexpr (LNF && WN > 1) {
    if (wn == 0) k_loop(std::integral_constant<int, 0>{});
    else if (WN > 2 && wn == 2) k_loop(std::integral_constant<int, 2 % WN>{});
    else if (WN > 2 && wn == 3) k_loop(std::integral_constant<int, 3 % WN>{});
    else k_loop(std::integral_constant<int, 1>{});
  } else {
    k_loop(std::integral_constant<int, 0>{});
  }

  if constexpr (LNF)  // (the barrier that ended the K loop freed the LDS tiles)
    ln_fold_apply<TM, TN, WN, BM>(p, acc, lnst, reinterpret_cast<float*>(smem), wm * WTM, wn,
                                  reinterpret_cast<const float*>(smem + 2 * (A_BYTES + B_BYTES)) + wn * WTN, l31, hi);
  constexpr int SMEM = 2 * (A_BYTES + B_BYTES);
  gemm_epilogue<TM, TN, RICH>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi, z, pre, PRE && use_pre, gemm_sink_lds<BM, BN, SMEM>(p, m0, n0, smem));
  gemm_sink_tail<NW * 64, BM, BN, SMEM>(p, m0, n0, smem);
}

// ======================================================================================================================
// fp8 (OCP e4m3) Linear on v_mfma_scale_f32_32x32x64_f8f6f4 (the SDXL transformer GEMMs).
//   out[m, n] = epilogue( sa[m] * sw[n] * sum_k Aq[m, k] * Wq[n, k] ),   Aq / Wq one byte per element, K contiguous,
// with per-row (per-token) activation scales and per-output-channel weight scales (gn_quantize_fp8_rows makes both).
// Same LDS-DMA tile machinery as gemm_dma_kernel: a 128-byte LDS row now holds 128 K elements (BK = 128), and a lane's MFMA
// operand is 32 consecutive bytes of its row -- K bytes 64 kk + 32 hi .. + 32 -- i.e. two swizzled 16-byte chunks
// (tools/probes/mfma_fp8_layout.hip pins the operand layout on the GPU).  One MFMA does 32 x 32 x 64 MACs, twice an f16 one.
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64, gemm_waves_per_simd(2 * (BM + BN) * 128, WM* WN)) void gemm_fp8_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int BKB = 128;  // K elements (= bytes) per tile

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
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
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = (p.K + BKB - 1) / BKB;

  const int lr = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  int kcur = chunk * 16;
  unsigned aoff[GA], woff[GB];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int m = m0 + 8 * (wave + NW * i) + lr;
    aoff[i] = (m < p.M) ? (unsigned)((long)m * p.lda) : kOOB;
  }
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int n = n0 + 8 * (wave + NW * i) + lr;
    woff[i] = (n < p.N) ? (unsigned)((long)n * p.ldw) : kOOB;
  }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  auto dma_tile = [&](int buf) {
    const bool kok = kcur < p.K;  // K % 16 == 0: a chunk is all in or all out; out-of-range chunks land as zeros
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      unsigned voff = (kok && aoff[i] != kOOB) ? aoff[i] + (unsigned)kcur : kOOB;
      GN_PIN(voff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(As + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      unsigned voff = (kok && woff[i] != kOOB) ? woff[i] + (unsigned)kcur : kOOB;
      GN_PIN(voff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
    }
    kcur += BKB;
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto frag = [&](const unsigned char* T, int row, int kk) -> i32x8 {
    const uint4 lo = *reinterpret_cast<const uint4*>(T + lds_swz<128>(row, kk * 4 + hi * 2));
    const uint4 up = *reinterpret_cast<const uint4*>(T + lds_swz<128>(row, kk * 4 + hi * 2 + 1));
    i32x8 f = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)up.x, (int)up.y, (int)up.z, (int)up.w};
    return f;
  };

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) dma_tile(cur ^ 1);
    const unsigned char* As = smem + cur * (A_BYTES + B_BYTES);
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      i32x8 fa[TM], fw[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = frag(As, wm * WTM + i * 32 + l31, kk);
#pragma unroll
      for (int j = 0; j < TN; ++j) fw[j] = frag(Bs, wn * WTN + j * 32 + l31, kk);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[j], fa[i], acc[j][i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // dequantise: the lane holds D[n = 8g + 4hi + (r&3)][m = lane&31] of each 32x32 tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + l31;
    const float sa = m < p.M ? p.sa[m] : 0.0f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = n0 + wn * WTN + j * 32 + 8 * g + 4 * hi;
        f32x4 sw = {0.f, 0.f, 0.f, 0.f};
        if (nb < p.N) sw = *reinterpret_cast<const f32x4*>(p.sw + nb);
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[j][i][4 * g + x] *= sa * sw[x];
      }
  }
  gemm_epilogue<TM, TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi, 0);
}

template <int BM, int BN, int WM, int WN>
void launch_fp8(const GemmParams& p, hipStream_t st) {
  hipLaunchKernelGGL((gemm_fp8_kernel<BM, BN, WM, WN>), dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), 0, st, p);
}

// split-K: sum the f32 partial slabs in a fixed order and apply the fused epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
  const long n4 = p.N >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.M * n4) return;
  const int m = (int)(idx / n4);
  const int nb = (int)(idx - (long)m * n4) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.splitk; ++z) s += *reinterpret_cast<const f32x4*>(p.ws + ((long)z * p.M + m) * p.N + nb);
  epilogue_store4(p, m, nb, s[0], s[1], s[2], s[3]);
}

// ---- split-K reduce + GroupNorm (+ SiLU) of the result in ONE launch (gn_gemm_desc.norm_out) -----------------------------------------------
// The small-M convs of the 8x8 .. 32x32 latent levels split K and finish in a reduce kernel; diffusers runs a GroupNorm on most of their outputs
// next (ResnetBlock2D conv1 -> norm2 -> SiLU, conv2 -> the next block's norm1 / Transformer2DModel.norm; `self.pipe(...)`,
// controller/agent/sd_controlnet_agent.py:67-76).  Here one 512-thread workgroup owns a (sample, group) slab, as norm.hip's single-launch
// GroupNorm does: it sums the partial slabs in the fixed z order, applies the fused epilogue, stores the raw f16 rows (the residual stream reads
// them) and keeps the rounded values in LDS with their f32 statistics, then writes act(GroupNorm) of them -- the arithmetic of
// splitk_reduce_kernel followed by gn_fused_kernel without the second launch and its read of the tensor.  Deterministic (no atomics).
constexpr int RGN_THREADS = 512;
__device__ __forceinline__ void epilogue_pair(const GemmParams& p, int m, int n, float& v0, float& v1) {
  if (p.bias) { const f16x2 b = *reinterpret_cast<const f16x2*>(p.bias + n); v0 += (float)b[0]; v1 += (float)b[1]; }
  if (p.shift) { const f16x2 t = *reinterpret_cast<const f16x2*>(p.shift + (long)(m / p.rpb) * p.ldshift + n); v0 += (float)t[0]; v1 += (float)t[1]; }
  f16x2 r = {(f16)0.0f, (f16)0.0f};
  if (p.res) r = *reinterpret_cast<const f16x2*>(p.res + (long)m * p.ldr + n);
  if (p.res && p.res_first) { v0 += (float)r[0]; v1 += (float)r[1]; }
  if (p.act != GN_ACT_NONE) { v0 = gemm_act(v0, p.act); v1 = gemm_act(v1, p.act); }
  if (p.out_scale != 1.0f) { v0 *= p.out_scale; v1 *= p.out_scale; }
  if (p.res && !p.res_first) { v0 += (float)r[0]; v1 += (float)r[1]; }
}
__global__ __launch_bounds__(RGN_THREADS) void splitk_reduce_gn_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) uint2 rgn_slab[];  // [rps][cpg / 4] packed f16 quads
  __shared__ float red[2 * (RGN_THREADS / 64)];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int G = p.nout.groups, cpg = p.N / G, hq = cpg >> 2, rps = p.nout.rps;
  // XCD-aware group order (block x runs on XCD x % 8): an XCD takes CONTIGUOUS groups -- neighbours share the cache lines of every row
  const int g = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int j = tid % hq, p0 = tid / hq, pstep = RGN_THREADS / hq;
  const int n = g * cpg + 4 * j;
  float s = 0.f, ss = 0.f;
  if (p0 < pstep) {
    // the launch is a pure gather of 16-byte pieces (a row of a slab is cpg x 4 bytes of a partial slab's row): what it needs is loads in
    // flight -- two rows x eight K slices per trip, predicated (a runtime trip count would serialise the slices: one memory round trip each)
    for (int r0 = p0; r0 < rps; r0 += 2 * pstep) {
      f32x4 v[2];
      v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < p.splitk; z0 += 8) {
        f32x4 t[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int zz = 0; zz < 8; ++zz) {
            t[u][zz] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int r = r0 + u * pstep;
            if (r < rps && z0 + zz < p.splitk) t[u][zz] = *reinterpret_cast<const f32x4*>(p.ws + ((long)(z0 + zz) * p.M + (long)b * rps + r) * p.N + n);
          }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int zz = 0; zz < 8; ++zz) v[u] += t[u][zz];  // (slices in z order, as splitk_reduce_kernel sums them)
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * pstep;
        if (r < rps) {
          const int m = b * rps + r;
          float w[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
          int bidx;
          epilogue_vals4(p, m, n, w, bidx);
          f16x4 h;
#pragma unroll
          for (int i = 0; i < 4; ++i) h[i] = (f16)w[i];
          *reinterpret_cast<f16x4*>(p.out + (long)m * p.ldo + n) = h;
          rgn_slab[r * hq + j] = *reinterpret_cast<const uint2*>(&h);
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float a = (float)h[i]; s += a; ss += a * a; }
        }
      }
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { red[wave] = s; red[RGN_THREADS / 64 + wave] = ss; }
  __syncthreads();
  float ts = 0.f, tss = 0.f;
#pragma unroll
  for (int w = 0; w < RGN_THREADS / 64; ++w) { ts += red[w]; tss += red[RGN_THREADS / 64 + w]; }
  const float cnt = (float)rps * (float)cpg;
  const float mean = ts / cnt;
  const float var = fmaxf(tss / cnt - mean * mean, 0.0f);
  const float rstd = rsqrtf(var + p.nout.eps);
  if (p0 < pstep) {
    const f16x4 gm = *reinterpret_cast<const f16x4*>(p.nout.gamma + n), bt = *reinterpret_cast<const f16x4*>(p.nout.beta + n);
    float a[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = rstd * (float)gm[i]; sh[i] = (float)bt[i] - mean * a[i]; }
    f16* dst = p.nout.y + (long)b * rps * p.N + n;
    for (int r = p0; r < rps; r += pstep) {
      const uint2 raw = rgn_slab[r * hq + j];
      const f16x4 v = *reinterpret_cast<const f16x4*>(&raw);
      f16x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y = (float)v[i] * a[i] + sh[i];
        if (p.nout.act == GN_ACT_SILU) y = act_silu(y);
        o[i] = (f16)y;
      }
      *reinterpret_cast<f16x4*>(dst + (long)r * p.N) = o;
    }
  }
}
constexpr long kRgnMaxSlab = 96 * 1024;

// the same reduce for a producer of the GroupNorm bridge (GemmParams.sink): a block owns an 8-row x 128-column patch of the output (512-byte
// row pieces of the partial slabs, as many blocks as the plain reduce), so the statistics of what it stores are a column sum through LDS + a
// handful of atomics per block (gn_bridge.h)
__global__ __launch_bounds__(256) void splitk_reduce_sink_kernel(const GemmParams p) {
  __shared__ float ls[8][129], lq[8][129];
  __shared__ float cs[2][128];
  const int t = threadIdx.x;
  const int r = t >> 5, c4 = (t & 31) * 4;
  const int m0 = blockIdx.y * 8, n0 = blockIdx.x * 128;
  const int m = m0 + r, nb = n0 + c4;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (m < p.M && nb < p.N) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splitk; ++z) s += *reinterpret_cast<const f32x4*>(p.ws + ((long)z * p.M + m) * p.N + nb);
    float v[4] = {s[0], s[1], s[2], s[3]};
    int bidx;
    epilogue_vals4(p, m, nb, v, bidx);
    f16x4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = (f16)v[i]; o[i] = (float)h[i]; }
    *reinterpret_cast<f16x4*>(p.out + out_row_off(p, m) + nb) = h;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { ls[r][c4 + i] = o[i]; lq[r][c4 + i] = o[i] * o[i]; }
  __syncthreads();
  const int mend = min(m0 + 8, p.M), nend = min(n0 + 128, p.N);
  for (int b = m0 / p.sink.rps; b <= (mend - 1) / p.sink.rps; ++b) {
    const int r0 = max(m0, b * p.sink.rps) - m0, r1 = min(mend, (b + 1) * p.sink.rps) - m0;
    if (t < 128) {
      float a = 0.f, q = 0.f;
      for (int rr = r0; rr < r1; ++rr) { a += ls[rr][t]; q += lq[rr][t]; }
      cs[0][t] = a;
      cs[1][t] = q;
    }
    __syncthreads();
    const int g0 = (p.sink.coff + n0) / p.sink.cpg, g1 = (p.sink.coff + nend - 1) / p.sink.cpg;
    if (t <= g1 - g0) {
      const int g = g0 + t;
      const int c0 = max(g * p.sink.cpg - p.sink.coff, n0) - n0, c1 = min((g + 1) * p.sink.cpg - p.sink.coff, nend) - n0;
      float a = 0.f, q = 0.f;
      for (int c = c0; c < c1; ++c) { a += cs[0][c]; q += cs[1][c]; }
      gn_stats_add(p.sink.stats + gn_stats_line((int)(blockIdx.y % p.sink.reps), b, g, p.sink.nb, p.sink.groups), a, q);
    }
    __syncthreads();
  }
}

template <int BM, int BN, int WM, int WN>
void launch_cfg(const GemmParams& p, bool conv, hipStream_t st) {
  dim3 grid(p.tiles_m * p.tiles_n, p.splitk, p.nbatch > 0 ? p.nbatch : 1);
  if (conv)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false>), grid, dim3(WM * WN * 64), 0, st, p);
}

template <int BM, int BN, int WM, int WN>
void launch_dma(const GemmParams& p, bool conv, hipStream_t st) {
  dim3 grid(p.tiles_m * p.tiles_n, p.splitk, p.nbatch > 0 ? p.nbatch : 1);
  if (conv)
    hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else if (p.ln_c1)
    hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, false, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, false>), grid, dim3(WM * WN * 64), 0, st, p);
}

struct Plan {
  int cfg;  // index into kCfg
  int bm, bn, splitk, kper;
};

// tile configurations: {BM, BN, relative MFMA rate on large problems (tools/bench_gemm.py), GEGLU-capable (wave tile >= 64
// columns)}.  Index + 1 is the public gn_gemm_desc::tile value.
struct Cfg { int bm, bn; double eff; bool geglu; bool dma; };
constexpr Cfg kCfg[] = {{256, 128, 1.00, true, false}, {128, 128, 1.00, true, false}, {128, 64, 0.85, false, false},
                        {64, 64, 0.65, false, false},  {256, 64, 0.90, true, false},  {128, 256, 1.00, true, false},
                        // LDS-DMA variants (never picked by the fallback heuristic: eff 0; the host autotuner times them)
                        {256, 256, 0.0, true, true},   {256, 128, 0.0, true, true},   {128, 128, 0.0, true, true},
                        {128, 64, 0.0, false, true},   {64, 64, 0.0, false, true},    {256, 64, 0.0, true, true},
                        // 320-wide N tiles: the 64x64-latent UNet level (N = 320) without padded-tile waste
                        {128, 320, 0.0, false, true},  {256, 320, 0.0, false, true},
                        // ping-pong 256x256 (gemm_pp.hip): counted-vmcnt 8-phase K loop
                        {256, 256, 0.0, false, true},
                        // 3-stage LDS-DMA ring (gemm_s3.hip): two K tiles in flight, counted vmcnt -- the latency-bound mid-size launches
                        {128, 128, 0.0, true, true},   {128, 64, 0.0, false, true},   {64, 64, 0.0, false, true},    {256, 64, 0.0, true, true},
                        // exact-fit ring tiles: N = 640 / 1280 / 320 problems whose 128x64 / 128x128 grids leave the 256 CUs 1.25 .. 2.5
                        // workgroups each (the K loop of those launches is bound by L2 -> LDS bytes per CU: bigger tile, fewer bytes)
                        {128, 160, 0.0, false, true},  {64, 160, 0.0, false, true},   {64, 320, 0.0, false, true},
                        // 2-stage 128x160 (two workgroups per CU)
                        {128, 160, 0.0, false, true},
                        // 2-stage 128x320 on EIGHT waves of 32x160 (round 5): the wave tile of the 128x160 tile with all of N = 320 in one workgroup --
                        // the A tile is staged once for both column halves (10.9 instead of 14 LDS-DMA bytes per kFLOP), one workgroup per CU
                        {128, 320, 0.0, false, true},
                        // persistent skewed ping-pong 256x256 (gemm_ppp.hip, round 6): one workgroup per CU walks the tile list; the next tile's ring is
                        // requested before the finished tile's epilogue, tile boundaries are skewed over the chip, the last partial round is split along K
                        {256, 256, 0.0, true, true}};
constexpr int kNumCfg = 25;
constexpr int kCfgPPP = 24;
constexpr int kCfgS3End = 22;  // one past the last 3-stage configuration
constexpr int kCfgPP = 14;
constexpr int kCfgS3 = 15;  // first of the four 3-stage configurations

// the ping-pong kernel's extra restrictions on top of dma_eligible (gemm_pp.hip header)
bool pp_eligible(const gn_gemm_desc* d) {
  if (d->act == GN_ACT_GEGLU || (d->batch > 1 && !d->up_phases) || d->fp8 || d->K % 64 != 0) return false;
  if (d->conv && (d->C1 % 64 != 0 || d->C2 % 64 != 0)) return false;
  return true;
}

// workgroups of the persistent kernel = CUs of the current device (256 on MI355X; also the answer where no device is visible: cross-compiling hosts)
int ppp_workgroups() {
  static int ncu[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!ncu[dev]) {
    int v = 0;
    ncu[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (v > 256 ? 256 : v) : 256;
  }
  return ncu[dev];
}
int64_t ppp_tiles(const gn_gemm_desc* d) { return cdiv64(d->M, 256) * cdiv64(d->N, 256) * (d->up_phases ? 4 : 1); }
// the persistent skewed ping-pong kernel's restrictions on top of pp_eligible (gemm_ppp.hip header)
bool ppp_eligible(const gn_gemm_desc* d) {
  const bool ff = d->act == GN_ACT_GEGLU || d->ln_c1 != nullptr;
  if (ff) {  // the feed-forward variant: LayerNorm fold AND GEGLU together (diffusers FeedForward.net[0] behind BasicTransformerBlock.norm3), dense, c2 in bias
    if (!(d->act == GN_ACT_GEGLU && d->ln_c1 && d->bias) || d->conv || d->batch > 1 || d->fp8 || d->K % 64 != 0 || d->residual || d->shift || d->out_scale != 1.0f) return false;
    if (((uintptr_t)d->ln_c1 & 15) != 0 || ((uintptr_t)d->bias & 7) != 0 || d->ln_eps <= 0.0f) return false;
  } else if (!pp_eligible(d)) {
    return false;
  }
  if (d->out_mode != GN_OUT_ROWMAJOR || d->out2 || d->sink.stats || d->norm_in.stats || d->norm_out.y) return false;
  if (d->splitk > 1 || (d->shift && d->residual) || d->k_append || d->a2) return false;
  // whole tiles only: no row mask in the loaders, the R part of an address is the buffer instruction's scalar offset
  if (d->M % 256 != 0 || d->N % 256 != 0 || d->K < 4 * 64) return false;
  if (d->ldo % 8 != 0 || ((uintptr_t)d->out & 15) != 0 || (d->out_row_width && d->ldo_hi % 8 != 0)) return false;
  if (d->residual && (d->ldr % 8 != 0 || ((uintptr_t)d->residual & 15) != 0)) return false;
  if (d->shift && ((d->ldshift > 0 ? d->ldshift : d->N) % 8 != 0 || ((uintptr_t)d->shift & 15) != 0 || d->rows_per_batch <= 0)) return false;
  if (d->bias && ((uintptr_t)d->bias & 7) != 0) return false;
  if (d->ldw >= (1 << 22) || (!d->conv && d->lda >= (1 << 22))) return false;  // 192 rows x the row pitch in bytes as a 32-bit scalar offset
  if (d->conv) {
    // a tile lies inside one sample and a lane's four rows are (oy_s + r0 / Wo, ox_s + r0 % Wo) with scalar (oy_s, ox_s) -- gemm_ppp.hip
    const int64_t hw = (int64_t)d->Ho * d->Wo;
    if (hw % 256 != 0 || !(d->Wo % 64 == 0 || 64 % d->Wo == 0)) return false;
    if (d->Ho * d->stride + 4 >= (1 << 15) || d->Wo * d->stride + 4 >= (1 << 15)) return false;  // packed 16-bit tap origins
  }
  const int64_t tiles = ppp_tiles(d);
  return tiles >= ppp_workgroups() && tiles < (1 << 24);
}

// buffer-descriptor extents of the LDS-DMA variant (32-bit byte offsets; kOOB must stay out of range)
struct DmaBytes { uint64_t a, a2, w, a3; };
DmaBytes dma_bytes(const gn_gemm_desc* d) {
  DmaBytes b;
  b.a3 = 0;
  if (d->conv) {
    const uint64_t px = (uint64_t)d->B * d->H * d->W;
    b.a = px * d->C1 * 2; b.a2 = px * d->C2 * 2;
    if (d->k_append && d->a3) b.a3 = px * d->C3 * 2;  // the second appended source has a buffer descriptor of its own
  } else {
    b.a = (uint64_t)d->M * d->lda * 2; b.a2 = d->k_append ? (uint64_t)d->M * d->lda2 * 2 : 0;
  }
  b.w = (uint64_t)d->N * d->ldw * 2;
  return b;
}
bool dma_eligible(const gn_gemm_desc* d) {
  const DmaBytes b = dma_bytes(d);
  const uint64_t lim = 0xFFFFFF00ull;
  if (b.a >= lim || b.a2 >= lim || b.w >= lim || b.a3 >= lim) return false;
  if (d->conv && d->a2 && !d->k_append && (d->C1 % 64 != 0 || (d->C1 + d->C2) % 64 != 0)) return false;  // a K tile must not straddle the concat
  return true;
}

int g_tile_override = -2;
int tile_override() {
  if (g_tile_override == -2) {
    const char* e = getenv("GN_GEMM_TILE");  // tuning aid: force a tile configuration index
    g_tile_override = e ? atoi(e) : -1;
  }
  return g_tile_override;
}

// Tile / split-K heuristic: maximise (config efficiency) x (useful fraction of the padded tile grid) x (fraction of the 256
// CUs' block slots the grid fills); split K when the grid would leave most CUs idle (8x8 / 16x16 latent levels, batch 1).
Plan plan_gemm(const gn_gemm_desc* d) {
  const int64_t M = d->M, N = d->N, K = d->K;
  Plan pl;
  const bool geglu = d->act == GN_ACT_GEGLU;
  int best = 1;
  {
    // fallback heuristic when the caller did not autotune (genima_amd/engine.py times every configuration per shape):
    // score = tile efficiency x useful fraction of the padded grid x how evenly the grid fills the 256 CUs
    double bs = -1.0;
    for (int c = 0; c < kNumCfg; ++c) {
      if (geglu && !kCfg[c].geglu) continue;
      const int64_t tm = cdiv64(M, kCfg[c].bm), tn = cdiv64(N, kCfg[c].bn);
      const double useful = (double)(M * N) / (double)(tm * kCfg[c].bm * tn * kCfg[c].bn);
      const double blocks = (double)(tm * tn);
      const double slots = 256.0 * (kCfg[c].bm * kCfg[c].bn > 128 * 128 ? 1.0 : (kCfg[c].bm * kCfg[c].bn == 128 * 128 ? 2.0 : 3.0));
      double fill = blocks >= slots ? blocks / (ceil(blocks / slots) * slots) : blocks / slots;
      if (fill < 0.05) fill = 0.05;
      const double score = kCfg[c].eff * useful * (0.35 + 0.65 * fill);
      if (score > bs) { bs = score; best = c; }
    }
  }
  const int ov = tile_override();
  if (ov >= 0 && ov < kNumCfg) best = ov;
  if (d->tile >= 1 && d->tile <= kNumCfg) best = d->tile - 1;
  if (geglu && !kCfg[best].geglu) best = 1;
  if (d->ln_c1) {  // LayerNorm fold: the LDS-DMA kernels (two-stage and ring) carry it
    static const int to_dma[kNumCfg] = {7, 8, 9, 10, 11, 8, 6, 7, 8, 9, 10, 11, 12, 13, 6, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24};
    best = to_dma[best];
  }
  if (d->k_append && !kCfg[best].dma) {  // the appended segment lives in the LDS-DMA loaders
    static const int to_dma[kNumCfg] = {7, 8, 9, 10, 11, 8, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24};
    best = to_dma[best];
  }
  if (d->norm_in.stats) {  // the normalising A path lives in the ring kernels (gemm_s3.hip): 15 .. 21 = {128x128, 128x64, 64x64, 256x64, 128x160, 64x160, 64x320}
    static const int to_s3[kNumCfg] = {18, 15, 16, 17, 18, 15, 15, 18, 15, 16, 17, 18, 19, 19, 15, 15, 16, 17, 18, 19, 20, 21, 19, 19, 15};
    best = to_s3[best];
    // a row tile spans whole samples or lies inside one (the kernel's scale / shift table covers <= 4 of them)
    const int64_t rps = d->conv ? (int64_t)d->Ho * d->Wo : d->norm_in.rows_per_sample;
    if (rps > 0 && kCfg[best].bm > 4 * rps) best = kCfg[best].bn >= 160 ? 20 : 17;  // -> a 64-row tile
    // the ring + the scale / shift table of the samples a row tile touches must fit the CU's 160 KB (launch_s3_gna): else the 64 x 64 ring
    const int64_t ct = d->conv ? (d->k_append ? d->C1 : d->C1 + d->C2) : (d->k_append ? d->K - d->C2 : d->K);
    auto lds_bytes = [&](int c) { return (int64_t)3 * (kCfg[c].bm + kCfg[c].bn) * 128 + (rps >= kCfg[c].bm ? 1 : kCfg[c].bm / (rps > 0 ? rps : 1)) * ct * 8; };
    if (lds_bytes(best) > 160 * 1024) best = 17;
  }
  if (best == kCfgPPP && !ppp_eligible(d)) best = (geglu || d->ln_c1) ? 6 : kCfgPP;
  if (best == kCfgPP && !pp_eligible(d)) best = 6;
  if (kCfg[best].dma && !dma_eligible(d)) {
    static const int fallback[kNumCfg] = {0, 1, 2, 3, 4, 5, 0, 0, 1, 2, 3, 4, 1, 0, 0, 1, 2, 3, 4, 1, 2, 2, 1, 1, 0};
    best = fallback[best];
  }
  {  // audit aid: GN_GEMM_LOG_FALLBACK=1 reports every launch whose requested tile (the tune table's) is not the tile that runs
    static const bool log_fb = getenv("GN_GEMM_LOG_FALLBACK") && atoi(getenv("GN_GEMM_LOG_FALLBACK")) != 0;
    if (log_fb && d->tile >= 1 && d->tile <= kNumCfg && best != d->tile - 1)
      fprintf(stderr, "[gn_gemm] tile %d requested, %d runs: conv %d M %ld N %ld K %ld act %d batch %d up_phases %d ln %d k_append %d fp8 %d C1 %d C2 %d\n", d->tile,
              best + 1, d->conv, (long)M, (long)N, (long)K, d->act, d->batch, d->up_phases, d->ln_c1 != nullptr, d->k_append, d->fp8, d->C1, d->C2);
  }
  pl.cfg = best;
  pl.bm = kCfg[best].bm; pl.bn = kCfg[best].bn;
  const int64_t blocks = cdiv64(M, pl.bm) * cdiv64(N, pl.bn);
  int sk = d->splitk;
  if (sk <= 0) {
    sk = 1;
    if (d->act != GN_ACT_GEGLU && d->out_mode != GN_OUT_BATCH_TRANSPOSED && d->batch <= 1 &&
        blocks < (d->out_mode == GN_OUT_F32 ? 512 : 192) && K >= 1024) {
      // f32 output = weight gradients: a handful of output tiles under a reduction over every pixel of the batch (K up to 2^21), so
      // the K split has to supply the parallelism; f16 outputs keep the inference limits (their summation order is part of the
      // recorded results)
      const bool wgrad = d->out_mode == GN_OUT_F32;
      sk = (int)cdiv64(wgrad ? 1024 : 384, blocks);
      const int maxsk = (int)(K / 512);
      if (sk > maxsk) sk = maxsk;
      if (sk > (wgrad ? 256 : 16)) sk = wgrad ? 256 : 16;
      if (sk < 1) sk = 1;
    }
  }
  if (d->act == GN_ACT_GEGLU || d->out_mode == GN_OUT_BATCH_TRANSPOSED || d->batch > 1 || d->fp8 || d->out2 || d->ln_c1 || best == kCfgPPP) sk = 1;
  int kper = (int)(cdiv64(cdiv64(K, sk), BK) * BK);
  sk = (int)cdiv64(K, kper);
  pl.splitk = sk;
  pl.kper = kper;
  return pl;
}

}  // namespace

extern "C" int32_t gn_set_gemm_tile_override(int32_t cfg) {
  g_tile_override = cfg < 0 ? -1 : cfg;
  return GN_OK;
}

extern "C" int32_t gn_gemm_norm_in_supported(const gn_gemm_desc* d) {
  if (!d || d->fp8 || d->ln_c1 || d->batch > 1 || d->up_phases || d->upsample2x || !dma_eligible(d)) return 0;
  if (d->norm_in.groups <= 0 || d->norm_in.cpg <= 0 || d->norm_in.cpg % 2 != 0 || d->norm_in.rows_per_sample <= 0) return 0;
  if (d->norm_in.act != GN_ACT_NONE && d->norm_in.act != GN_ACT_SILU) return 0;
  int64_t ct, rps;
  if (d->conv) {
    if (d->C1 % 64 != 0 || (d->C2 % 64 != 0)) return 0;  // a K tile (one tap x 64 channels) never straddles a source or a tap
    ct = d->k_append ? d->C1 : d->C1 + d->C2;
    rps = (int64_t)d->Ho * d->Wo;
    if ((int64_t)d->H * d->W != d->norm_in.rows_per_sample) return 0;
  } else {
    ct = d->k_append ? d->K - d->C2 : d->K;
    rps = d->norm_in.rows_per_sample;
    if (ct % 8 != 0) return 0;
  }
  if (ct != (int64_t)d->norm_in.groups * d->norm_in.cpg) return 0;
  if (rps <= 0 || d->M % rps != 0 || d->norm_in.samples != d->M / rps) return 0;
  // row tiles of 64 .. 256 rows must span whole samples or sit inside one: rps a power-of-two multiple / divisor of 64, at least 16 rows
  if (rps < 16 || (rps & (rps - 1)) != 0) return 0;
  if (4 * ct * 8 + 3 * (64 + 64) * 128 > 160 * 1024) return 0;  // table of up to 4 samples + the smallest ring
  return 1;
}

extern "C" int32_t gn_gemm_norm_out_supported(const gn_gemm_desc* d) {
  if (!d || !d->norm_out.y || !d->norm_out.gamma || !d->norm_out.beta) return 0;
  if (d->out_mode != GN_OUT_ROWMAJOR || d->out2 || d->act == GN_ACT_GEGLU || d->fp8 || d->batch > 1 || d->out_row_width || d->sink.stats || d->ln_c1) return 0;
  const int G = d->norm_out.groups, rps = d->norm_out.rows_per_sample;
  if (G <= 0 || rps <= 0 || d->N % G != 0 || d->M % rps != 0 || (d->norm_out.act != GN_ACT_NONE && d->norm_out.act != GN_ACT_SILU)) return 0;
  const int64_t cpg = d->N / G;
  if (cpg % 4 != 0 || cpg / 4 > RGN_THREADS || (int64_t)rps * cpg * 2 > kRgnMaxSlab || d->N % 4 != 0 || d->ldo % 4 != 0) return 0;
  if (((uintptr_t)d->norm_out.y & 7) != 0 || ((uintptr_t)d->out & 7) != 0 || ((uintptr_t)d->norm_out.gamma & 7) != 0 || ((uintptr_t)d->norm_out.beta & 7) != 0) return 0;
  if (d->residual && d->ldr % 4 != 0) return 0;
  if (d->shift && ((d->ldshift > 0 ? d->ldshift : d->N) % 4 != 0 || d->rows_per_batch <= 0)) return 0;
  return plan_gemm(d).splitk > 1 ? 1 : 0;  // the plan (d->tile / d->splitk) must split K: the fusion lives in the reduce launch
}

// Does the plan named in d (tile / splitk) carry the fusions attached to d?  gn_launch_gemm refuses such a launch with the same tests; callers that
// CHANGE a recorded op's plan (gn_program_set_gemm_plan, the in-call tuner) ask here first.  0 = no (gn_last_error says why).
extern "C" int32_t gn_gemm_plan_valid(const gn_gemm_desc* d) {
  if (!d) { gn_set_error("gn_gemm_plan_valid: null descriptor"); return 0; }
  const Plan pl = plan_gemm(d);
  if (d->norm_out.y && !gn_gemm_norm_out_supported(d)) {
    gn_set_error("norm_out lives in the split-K reduce: the plan (tile %d -> %d, splitk %d -> %d) does not split K or the problem is unsupported", d->tile, pl.cfg + 1, d->splitk, pl.splitk);
    return 0;
  }
  if (d->norm_in.stats) {
    if (!gn_gemm_norm_in_supported(d) || !d->norm_in.gamma || !d->norm_in.beta) { gn_set_error("norm_in: unsupported problem (gn_gemm_norm_in_supported) or missing gamma / beta"); return 0; }
    if (!(pl.cfg >= kCfgS3 && pl.cfg < kCfgS3End)) { gn_set_error("norm_in: the plan must be a ring tile (16 .. 22), tile %d runs", pl.cfg + 1); return 0; }
    const int64_t rps = d->conv ? (int64_t)d->Ho * d->Wo : d->norm_in.rows_per_sample;
    if (pl.bm > 4 * rps) { gn_set_error("norm_in: a %d-row tile would span more than 4 samples of %ld rows", pl.bm, (long)rps); return 0; }
  }
  return 1;
}

extern "C" int64_t gn_gemm_workspace_bytes(const gn_gemm_desc* d) {
  if (!d) return 0;
  Plan pl = plan_gemm(d);
  if (pl.cfg == kCfgPPP) {  // hand-off slabs of the tiles whose K range several workgroups share (gemm_ppp.hip)
    GemmParams p = {};
    p.K = (int)d->K;
    return (int64_t)gn_ppp_plan(&p, (int)ppp_tiles(d), ppp_workgroups()) * 256 * 256 * (int64_t)sizeof(float);
  }
  if (pl.splitk <= 1) return 0;
  return (int64_t)pl.splitk * d->M * d->N * (int64_t)sizeof(float);
}

int32_t gn_launch_gemm(gn_ctx* ctx, const gn_gemm_desc* d) {
  GN_REQUIRE(d && d->a && d->w && d->out, "gn_gemm: null a/w/out");
  GN_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gn_gemm: empty problem M=%ld N=%ld K=%ld", (long)d->M, (long)d->N, (long)d->K);
  GN_REQUIRE(d->K % 8 == 0 && d->ldw % 8 == 0, "gn_gemm: K (%ld) and ldw (%ld) must be multiples of 8", (long)d->K, (long)d->ldw);
  GN_REQUIRE(d->N % 4 == 0, "gn_gemm: N (%ld) must be a multiple of 4 (pad the weight rows)", (long)d->N);
  GN_REQUIRE(d->M < (1ll << 31) && d->N < (1 << 24) && d->K < (1 << 24), "gn_gemm: problem too large");
  GN_REQUIRE(((uintptr_t)d->a & 15) == 0 && ((uintptr_t)d->w & 15) == 0 && ((uintptr_t)d->out & 7) == 0,
             "gn_gemm: a/w must be 16-byte aligned, out 8-byte aligned");
  const bool geglu = d->act == GN_ACT_GEGLU;
  if (geglu) {
    GN_REQUIRE(d->N % 64 == 0 && !d->shift && !d->residual && d->out_mode == GN_OUT_ROWMAJOR,
               "gn_gemm: GEGLU needs N %% 64 == 0 and no shift/residual/transposed output");
  }
  GemmParams p;
  p.a = (const f16*)d->a; p.a2 = (const f16*)d->a2; p.w = (const f16*)d->w;
  p.bias = (const f16*)d->bias; p.shift = (const f16*)d->shift; p.res = (const f16*)d->residual;
  p.out = (f16*)d->out; p.ws = (float*)d->workspace;
  p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
  p.lda = d->lda; p.ldw = d->ldw; p.ldr = d->ldr; p.ldo = d->ldo;
  p.ldshift = d->ldshift > 0 ? d->ldshift : d->N;
  p.H = d->H; p.W = d->W; p.C1 = d->C1; p.C2 = d->C2; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride;
  p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.Ho = d->Ho; p.Wo = d->Wo; p.ups = d->upsample2x;
  p.act = d->act; p.out_mode = d->out_mode; p.res_first = d->residual_before_act;
  p.accumulate = d->accumulate;
  p.sa = nullptr; p.sw = nullptr;
  p.out2 = (f16*)d->out2; p.ldo2 = d->ldo2; p.split_n = d->split_n;
  if (d->out2) {
    GN_REQUIRE(d->out_mode == GN_OUT_ROWMAJOR && !geglu && d->batch <= 1 && !d->fp8, "gn_gemm: out2 needs a plain row-major f16 problem");
    GN_REQUIRE(d->split_n > 0 && d->split_n < d->N && d->split_n % 32 == 0, "gn_gemm: split_n (%d) must be a multiple of 32 inside (0, N)", d->split_n);
    GN_REQUIRE(d->rows_per_batch > 0 && d->M % d->rows_per_batch == 0 && d->ldo2 >= d->rows_per_batch, "gn_gemm: out2 needs rows_per_batch | M and ldo2 >= rows_per_batch");
    GN_REQUIRE(d->N % 8 == 0 && d->ldo % 8 == 0 && ((uintptr_t)d->out & 15) == 0, "gn_gemm: out2 needs the 16-byte row-major store path for out");
  }
  p.orw = d->out_row_width; p.ldo_hi = d->ldo_hi;
  if (d->out_row_width) {
    GN_REQUIRE(d->out_row_width > 0 && d->M % d->out_row_width == 0 && d->ldo_hi % 4 == 0, "gn_gemm: out_row_width must divide M, ldo_hi %% 4 == 0");
    GN_REQUIRE(d->out_mode == GN_OUT_ROWMAJOR && d->act != GN_ACT_GEGLU && (d->batch <= 1 || d->up_phases) && !d->out2,
               "gn_gemm: the two-level output row pitch is for plain row-major f16 outputs");
  }
  p.ln_c1 = d->ln_c1; p.ln_eps = d->ln_eps;
  if (d->ln_c1) {
    GN_REQUIRE(!d->conv && d->batch <= 1 && !d->fp8 && d->out_mode != GN_OUT_F32, "gn_gemm: ln_c1 (LayerNorm fold) is for dense f16 Linears");
    GN_REQUIRE(dma_eligible(d), "gn_gemm: ln_c1 needs an LDS-DMA eligible problem (16-byte aligned rows, 32-bit operand extents)");
    GN_REQUIRE(((uintptr_t)d->ln_c1 & 15) == 0 && d->ln_eps > 0.0f && d->bias, "gn_gemm: ln_c1 must be 16-byte aligned, ln_eps > 0, and bias must hold c2");
  }
  p.nbatch = d->batch > 1 ? d->batch : 0;
  p.binner = p.nbatch ? (d->batch_inner > 0 ? d->batch_inner : 1) : 0;
  p.up_ph = d->up_phases ? 1 : 0;
  if (d->up_phases) {
    GN_REQUIRE(d->conv && d->batch == 4 && d->KH == 2 && d->KW == 2 && d->stride == 1 && !d->upsample2x && d->out_row_width > 0 && !d->residual &&
               !d->shift && d->w_bs % 8 == 0 && d->ldo % 2 == 0 && d->ldo_hi % 2 == 0,
               "gn_gemm: up_phases is the four-phase form of an upsampling 3x3 conv (conv, batch = 4, 2x2 taps, two-level output row pitch)");
    p.binner = 0;  // the phase offsets replace the batch strides
  }
  p.a_bs = d->a_bs; p.a_bs2 = d->a_bs2; p.w_bs = d->w_bs; p.w_bs2 = d->w_bs2;
  p.o_bs = d->out_bs; p.o_bs2 = d->out_bs2; p.r_bs = d->res_bs; p.r_bs2 = d->res_bs2;
  if (p.nbatch && !d->up_phases) {
    GN_REQUIRE(!d->conv && !d->shift && d->out_mode != GN_OUT_BATCH_TRANSPOSED, "gn_gemm: batched mode is dense GEMM only (no conv/shift/transposed out)");
    GN_REQUIRE(d->a_bs % 8 == 0 && d->a_bs2 % 8 == 0 && d->w_bs % 8 == 0 && d->w_bs2 % 8 == 0 && d->out_bs % 4 == 0 && d->out_bs2 % 4 == 0 &&
               d->res_bs % 4 == 0 && d->res_bs2 % 4 == 0, "gn_gemm: batch strides must keep 16-byte (a/w) and 8-byte (out/res) alignment");
    GN_REQUIRE(d->batch % p.binner == 0, "gn_gemm: batch (%d) must be a multiple of batch_inner (%d)", d->batch, p.binner);
  }
  if (d->out_mode == GN_OUT_F32) GN_REQUIRE(d->act != GN_ACT_GEGLU && ((uintptr_t)d->out & 15) == 0, "gn_gemm: f32 output needs a 16-byte aligned out and no GEGLU");
  if (d->accumulate) GN_REQUIRE(d->out_mode == GN_OUT_F32, "gn_gemm: accumulate needs GN_OUT_F32");
  p.rpb = d->rows_per_batch > 0 ? d->rows_per_batch : (int)d->M;
  p.out_scale = d->out_scale;
  if (d->conv) {
    GN_REQUIRE(d->C1 % 8 == 0 && d->C2 % 8 == 0 && d->C1 > 0, "gn_gemm(conv): C1/C2 (%d/%d) must be multiples of 8", d->C1, d->C2);
    GN_REQUIRE((d->C2 == 0) == (d->a2 == nullptr), "gn_gemm(conv): a2 and C2 must be given together");
    if (d->k_append) {
      GN_REQUIRE(d->K == (int64_t)d->KH * d->KW * d->C1 + d->C2 + d->C3 && d->a2 && d->C2 > 0, "gn_gemm(conv, k_append): K != KH*KW*C1 + C2 + C3");
      GN_REQUIRE((d->C3 == 0) == (d->a3 == nullptr) && d->C3 % 8 == 0 && (d->C3 == 0 || d->C2 % 64 == 0), "gn_gemm(conv, k_append): a3 / C3 together, C3 %% 8 == 0, C2 %% 64 == 0 in front of a3");
      GN_REQUIRE(d->stride == 1 && !d->upsample2x && d->C1 % 64 == 0 && d->Ho == d->H && d->Wo == d->W && d->batch <= 1 && !d->up_phases,
                 "gn_gemm(conv, k_append): a stride-1 same-size conv with C1 %% 64 == 0 (the 1x1 segment reads the output pixel of the second source)");
      GN_REQUIRE(dma_eligible(d), "gn_gemm(conv, k_append): operands must fit 32-bit buffer offsets (LDS-DMA loaders)");
    } else
      GN_REQUIRE(d->K == (int64_t)d->KH * d->KW * (d->C1 + d->C2), "gn_gemm(conv): K != KH*KW*(C1+C2)");
    GN_REQUIRE(d->M == (int64_t)d->B * d->Ho * d->Wo, "gn_gemm(conv): M != B*Ho*Wo");
    GN_REQUIRE(d->stride >= 1 && d->KH >= 1 && d->KW >= 1 && d->H > 0 && d->W > 0, "gn_gemm(conv): bad geometry");
    GN_REQUIRE((int64_t)d->B * d->H * d->W < (1ll << 31), "gn_gemm(conv): source tensor has too many pixels");
  } else if (d->k_append) {
    GN_REQUIRE(d->a2 && d->C2 > 0 && d->C2 % 8 == 0 && (d->K - d->C2) % 64 == 0 && d->K > d->C2, "gn_gemm(k_append): a2 with C2 (%d) columns behind K - C2 (%% 64 == 0)", d->C2);
    GN_REQUIRE(d->lda % 8 == 0 && d->lda >= d->K - d->C2 && d->lda2 % 8 == 0 && d->lda2 >= d->C2 && ((uintptr_t)d->a2 & 15) == 0, "gn_gemm(k_append): lda / lda2 must be multiples of 8 covering their columns");
    GN_REQUIRE(!d->ln_c1 && !d->fp8 && d->batch <= 1 && dma_eligible(d), "gn_gemm(k_append): plain dense problems on the LDS-DMA tiles");
  } else {
    GN_REQUIRE(d->lda % 8 == 0 && d->lda >= d->K, "gn_gemm: lda (%ld) must be a multiple of 8 and >= K", (long)d->lda);
  }
  if (d->residual) GN_REQUIRE(d->ldr % 4 == 0 && ((uintptr_t)d->residual & 7) == 0, "gn_gemm: residual stride/alignment");
  if (d->out_mode != GN_OUT_BATCH_TRANSPOSED) GN_REQUIRE(d->ldo % 4 == 0, "gn_gemm: ldo (%ld) must be a multiple of 4", (long)d->ldo);
  if (d->out_mode == GN_OUT_BATCH_TRANSPOSED) GN_REQUIRE(d->rows_per_batch > 0 && d->M % d->rows_per_batch == 0, "gn_gemm: transposed output needs rows_per_batch | M");
  if (d->shift) GN_REQUIRE(d->rows_per_batch > 0 && p.ldshift % 4 == 0 && ((uintptr_t)d->shift & 7) == 0, "gn_gemm: shift needs rows_per_batch and 8-byte aligned rows");

  Plan pl = plan_gemm(d);
  {
    const DmaBytes db = dma_bytes(d);
    p.a_bytes = (unsigned)db.a; p.a2_bytes = (unsigned)(db.a2 ? db.a2 : db.a); p.w_bytes = (unsigned)db.w;
  }
  p.splitk = pl.splitk; p.kper = pl.kper;
  {
    // bytes each operand brings in from memory once: A = the source pixels (a conv re-reads them per tap out of cache), W = N x K
    static const int cm_env = [] { const char* e = getenv("GN_GEMM_CM_TILES"); return e ? atoi(e) : -1; }();  // A/B switch: 0 / 1 force
    const int64_t a_el = d->conv ? (int64_t)d->B * d->H * d->W * (d->C1 + d->C2) : d->M * d->K;
    const int64_t w_el = d->N * d->K;
    p.cm_tiles = cm_env >= 0 ? cm_env : (w_el > 2 * a_el && d->batch <= 1 ? 1 : 0);
  }
  p.tiles_m = (int)cdiv64(d->M, pl.bm); p.tiles_n = (int)cdiv64(d->N, pl.bn);
  p.kapp = d->k_append ? 1 : 0;
  p.kapp_k0 = d->conv ? d->KH * d->KW * d->C1 : (int)(d->K - d->C2);
  p.lda2 = d->lda2;
  p.a3 = d->k_append ? (const f16*)d->a3 : nullptr; p.C3 = d->k_append ? d->C3 : 0;
  p.a3_bytes = p.a3 ? (unsigned)dma_bytes(d).a3 : p.a_bytes;  // (< 0xFFFFFF00: k_append requires dma_eligible)
  if (pl.splitk > 1) GN_REQUIRE(d->workspace, "gn_gemm: split-K (%d) needs a workspace of gn_gemm_workspace_bytes()", pl.splitk);
  p.nout.y = (f16*)d->norm_out.y; p.nout.gamma = (const f16*)d->norm_out.gamma; p.nout.beta = (const f16*)d->norm_out.beta;
  p.nout.eps = d->norm_out.eps; p.nout.groups = d->norm_out.groups; p.nout.act = d->norm_out.act; p.nout.rps = d->norm_out.rows_per_sample;
  if (d->norm_out.y) GN_REQUIRE(gn_gemm_norm_out_supported(d), "gn_gemm(norm_out): unsupported problem, or a plan that does not split K (gn_gemm_norm_out_supported)");
  p.sink = gn_sink_params(d->sink);
  if (d->sink.stats) {
    GN_REQUIRE(d->out_mode == GN_OUT_ROWMAJOR && !d->out2 && !geglu && !d->fp8 && !d->ln_c1 && (d->batch <= 1 || d->up_phases),
               "gn_gemm(sink): GroupNorm statistics come from plain row-major f16 outputs");
    GN_REQUIRE(d->N % 8 == 0 && d->ldo % 8 == 0 && ((uintptr_t)d->out & 15) == 0 && (!d->out_row_width || d->ldo_hi % 8 == 0),
               "gn_gemm(sink): the statistics tail re-reads the output in 16-byte pieces (N, ldo %% 8 == 0, 16-byte aligned out)");
    GN_REQUIRE(d->sink.cpg > 0 && d->sink.groups > 0 && d->sink.coff >= 0 && d->sink.rows_per_sample > 0 && ((uintptr_t)d->sink.stats & 7) == 0 &&
                   (int64_t)d->sink.coff + d->N <= (int64_t)d->sink.cpg * d->sink.groups && d->M % d->sink.rows_per_sample == 0 &&
                   d->sink.samples == d->M / d->sink.rows_per_sample && d->sink.replicas >= 1 && ((uintptr_t)d->sink.stats & 127) == 0,
               "gn_gemm(sink): cpg %d / coff %d / groups %d / rows_per_sample %d do not cover this output [%ld x %ld]", d->sink.cpg, d->sink.coff,
               d->sink.groups, d->sink.rows_per_sample, (long)d->M, (long)d->N);
  }
  p.gin.stats = (const long long*)d->norm_in.stats; p.gin.gamma = (const f16*)d->norm_in.gamma; p.gin.beta = (const f16*)d->norm_in.beta;
  p.gin.eps = d->norm_in.eps; p.gin.groups = d->norm_in.groups; p.gin.cpg = d->norm_in.cpg; p.gin.act = d->norm_in.act;
  p.gin.rps = d->norm_in.rows_per_sample; p.gin.nb = d->norm_in.samples; p.gin.reps = d->norm_in.replicas > 0 ? d->norm_in.replicas : 1;
  if (d->norm_in.stats) {
    GN_REQUIRE(gn_gemm_norm_in_supported(d) && d->norm_in.gamma && d->norm_in.beta && ((uintptr_t)d->norm_in.stats & 7) == 0,
               "gn_gemm(norm_in): unsupported problem (gn_gemm_norm_in_supported) or missing gamma / beta");
    GN_REQUIRE(pl.cfg >= kCfgS3 && pl.cfg < kCfgS3End, "gn_gemm(norm_in): the plan must be a ring tile (16 .. 22)");
    const int64_t rps = d->conv ? (int64_t)d->Ho * d->Wo : d->norm_in.rows_per_sample;
    GN_REQUIRE(pl.bm <= 4 * rps, "gn_gemm(norm_in): a %d-row tile would span more than 4 samples of %ld rows", pl.bm, (long)rps);
  }

  if (d->fp8) {
    GN_REQUIRE(!d->conv && d->batch <= 1 && d->out_mode == GN_OUT_ROWMAJOR,
               "gn_gemm(fp8): dense row-major Linear only (no conv / batch / transposed or f32 output)");
    GN_REQUIRE(d->scale_a && d->scale_w && ((uintptr_t)d->scale_w & 15) == 0, "gn_gemm(fp8): scale_a [M] and 16-byte aligned scale_w [N] are required");
    GN_REQUIRE(d->K % 16 == 0 && d->lda % 16 == 0 && d->ldw % 16 == 0, "gn_gemm(fp8): K, lda, ldw must be multiples of 16 bytes");
    GN_REQUIRE((uint64_t)d->M * d->lda < 0xFFFFFF00ull && (uint64_t)d->N * d->ldw < 0xFFFFFF00ull, "gn_gemm(fp8): operand too large for 32-bit buffer offsets");
    p.sa = (const float*)d->scale_a; p.sw = (const float*)d->scale_w;
    p.a_bytes = (unsigned)((uint64_t)d->M * d->lda); p.w_bytes = (unsigned)((uint64_t)d->N * d->ldw);
    p.splitk = 1; p.kper = (int)d->K;
    static const int to_dma[kNumCfg] = {7, 8, 9, 10, 11, 8, 6, 7, 8, 9, 10, 11, 8, 7, 6, 8, 9, 10, 11, 8, 9, 9, 8, 8, 6};
    const int cfg = to_dma[pl.cfg];
    const int bm = kCfg[cfg].bm, bn = kCfg[cfg].bn;
    p.tiles_m = (int)cdiv64(d->M, bm); p.tiles_n = (int)cdiv64(d->N, bn);
    switch (cfg) {
      case 6: launch_fp8<256, 256, 2, 4>(p, ctx->stream); break;
      case 7: launch_fp8<256, 128, 4, 2>(p, ctx->stream); break;
      case 8: launch_fp8<128, 128, 2, 2>(p, ctx->stream); break;
      case 9: launch_fp8<128, 64, 2, 2>(p, ctx->stream); break;
      case 10: launch_fp8<64, 64, 2, 2>(p, ctx->stream); break;
      default: launch_fp8<256, 64, 4, 1>(p, ctx->stream); break;
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
  }
  const bool conv = d->conv != 0;
  switch (pl.cfg) {
    case 0: launch_cfg<256, 128, 4, 2>(p, conv, ctx->stream); break;
    case 1: launch_cfg<128, 128, 2, 2>(p, conv, ctx->stream); break;
    case 2: launch_cfg<128, 64, 2, 2>(p, conv, ctx->stream); break;
    case 3: launch_cfg<64, 64, 2, 2>(p, conv, ctx->stream); break;
    case 4: launch_cfg<256, 64, 4, 1>(p, conv, ctx->stream); break;
    case 5: launch_cfg<128, 256, 2, 4>(p, conv, ctx->stream); break;
    case 6: launch_dma<256, 256, 2, 4>(p, conv, ctx->stream); break;
    case 7: launch_dma<256, 128, 4, 2>(p, conv, ctx->stream); break;
    case 8: launch_dma<128, 128, 2, 2>(p, conv, ctx->stream); break;
    case 9: launch_dma<128, 64, 2, 2>(p, conv, ctx->stream); break;
    case 10: launch_dma<64, 64, 2, 2>(p, conv, ctx->stream); break;
    case 11: launch_dma<256, 64, 4, 1>(p, conv, ctx->stream); break;
    case 12: launch_dma<128, 320, 2, 2>(p, conv, ctx->stream); break;
    case 13: launch_dma<256, 320, 4, 2>(p, conv, ctx->stream); break;
    case 14: gn_launch_gemm_pp(&p, conv, p.tiles_m * p.tiles_n, p.splitk, p.nbatch > 0 ? p.nbatch : 1, ctx->stream); break;
    case kCfgPPP: {
      GN_REQUIRE(d->workspace, "gn_gemm: tile 25 (persistent ping-pong) needs a workspace of gn_gemm_workspace_bytes()");
      (void)gn_ppp_plan(&p, (int)ppp_tiles(d), ppp_workgroups());
      GN_REQUIRE(gn_ppp_pool_init(ctx->device) == GN_OK, "gn_gemm: tile 25 could not allocate its flag pool (first use inside a stream capture?): %s", gn_last_error());
      gn_launch_gemm_ppp(&p, conv, ctx->stream);
      break;
    }
    case 22: launch_dma<128, 160, 4, 1>(p, conv, ctx->stream); break;
    case 23: launch_dma<128, 320, 4, 2>(p, conv, ctx->stream); break;
    default: gn_launch_gemm_s3(&p, pl.cfg - kCfgS3, conv, p.tiles_m * p.tiles_n, p.splitk, p.nbatch > 0 ? p.nbatch : 1, ctx->stream); break;
  }
  GN_LAUNCH_CHECK();
  if (pl.splitk > 1) {
    const long total = (long)p.M * (p.N >> 2);
    if (p.nout.y) {
      static GnOncePerDevice attr_set;
      if (attr_set.first()) GN_HIP(hipFuncSetAttribute((const void*)splitk_reduce_gn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRgnMaxSlab));
      const int cpg = p.N / p.nout.groups;
      hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3((unsigned)p.nout.groups, (unsigned)(p.M / p.nout.rps)), dim3(RGN_THREADS),
                         (size_t)p.nout.rps * (cpg / 4) * 8, ctx->stream, p);
    } else if (p.sink.stats)
      hipLaunchKernelGGL(splitk_reduce_sink_kernel, dim3((unsigned)cdiv64(p.N, 128), (unsigned)cdiv64(p.M, 8)), dim3(256), 0, ctx->stream, p);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, ctx->stream, p);
    GN_LAUNCH_CHECK();
  }
  return GN_OK;
}
