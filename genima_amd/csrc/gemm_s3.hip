// 3-stage LDS-DMA ring for the LATENCY-BOUND mid-size launches (Linears / small convs with one workgroup per CU and 5..40 K tiles):
// the 2-stage kernel of gemm.hip has one K tile in flight and drains it (`vmcnt(0)` + barrier) every iteration, so each iteration costs
// a full L2 / MALL round trip; here two tiles are in flight and the wait is counted.  Same tiles, loaders, swizzle and epilogue as
// gemm_dma_kernel (gemm.hip); the per-lane validity selects become OR-masks so that every path issues the same number of VMEM
// instructions (the counted vmcnt depends on it).  Tiles 16..19 of gn_gemm_desc::tile.
#include <type_traits>

#include "gemm_common.h"

namespace {

// s_waitcnt takes an immediate: one asm statement per count
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 1 && N <= 16, "DMA instructions per wave per K tile");
#define GN_VMCNT_CASE(n) if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
  GN_VMCNT_CASE(1) GN_VMCNT_CASE(2) GN_VMCNT_CASE(3) GN_VMCNT_CASE(4) GN_VMCNT_CASE(5) GN_VMCNT_CASE(6) GN_VMCNT_CASE(7) GN_VMCNT_CASE(8)
  GN_VMCNT_CASE(9) GN_VMCNT_CASE(10) GN_VMCNT_CASE(11) GN_VMCNT_CASE(12) GN_VMCNT_CASE(13) GN_VMCNT_CASE(14) GN_VMCNT_CASE(15)
  GN_VMCNT_CASE(16)
#undef GN_VMCNT_CASE
}

// GNA (GroupNorm bridge, consumer side -- gn_gemm_desc.norm_in): A holds the RAW tensor a GroupNorm (+ SiLU) stands in front of.  Every lane
// normalises the 16-byte pieces IT staged, in LDS, right after the counted wait that retires them and before the barrier that publishes the
// tile -- no extra barrier, the MFMAs read the same f16 values a separate GroupNorm launch would have stored.  Pieces that the DMA zero-filled
// (conv padding, rows / K past the end) stay zero: the conv pads the NORMALISED tensor.  The per-(sample, channel) scale / shift come from a
// table in LDS behind the ring, built in the prologue from the producers' statistics block.  A tap is re-normalised for each of the KH*KW
// taps that stage it: the route is for the small-M launches whose SIMDs wait on the weight stream anyway (the host gates it by rows).
template <int BM, int BN, int WM, int WN, bool CONV, bool LNF = false, bool GNA = false>
__global__ __launch_bounds__(WM* WN * 64, gemm_waves_per_simd(3 * (BM + BN) * 128, WM* WN)) void gemm_s3_kernel(const GemmParams pin) {
  static_assert(!LNF || (!CONV && 4 % WN == 0), "LayerNorm fold: dense problems, K steps dealt over 1 / 2 / 4 column waves");
  static_assert(!(LNF && GNA), "one normalisation per launch");
  const GemmParams p = batch_offset(pin);
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile >= 32x32");
  static_assert(NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row DMA groups per wave");
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;  // DMA instructions per wave per tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;

  // ONE LDS object per instantiation (a second one makes hipcc drain vmcnt): static, or -- GNA: the scale / shift table's size follows the
  // input's channel count -- the dynamic one
  unsigned char* smem;
  if constexpr (GNA) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    smem = smem_dyn;
  } else {
    __shared__ __attribute__((aligned(16))) unsigned char smem_static[3 * (A_BYTES + B_BYTES) + (LNF ? BN * 4 : 0)];
    smem = smem_static;
  }

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
  unsigned aoff[GA], amask[GA];  // dense: byte offset of the row start (kOOB if the row is out of range)
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
      aoff[i] = (m < p.M) ? (unsigned)((long)m * p.lda * 2) : 0u;
      amask[i] = (m < p.M) ? 0u : kOOB;
    }
  }
  unsigned woff[GB], wmask[GB];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int n = n0 + 8 * (wave + NW * i) + lr;
    woff[i] = (n < p.N) ? (unsigned)((long)n * p.ldw * 2) : 0u;
    wmask[i] = (n < p.N) ? 0u : kOOB;
  }

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, (int)p.a2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a3 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a3 ? p.a3 : p.a), 0, (int)p.a3_bytes, 0x00020000);  // k_append: second appended source

  // ---- GNA: scale / shift table [S][Ct] (float2) behind the ring; per staged tile the lane's channel offset and the validity bits of its
  // pieces ride in a two-entry register FIFO from dma_tile (issue) to norm_tile (two tiles later)
  [[maybe_unused]] const float2* tab = reinterpret_cast<const float2*>(smem + 3 * (A_BYTES + B_BYTES));
  [[maybe_unused]] int trow[GA];
  [[maybe_unused]] int gi_c0 = 0, gi_c1 = 0;
  [[maybe_unused]] unsigned gi_m0 = 0u, gi_m1 = 0u;
  [[maybe_unused]] bool one_sample = true;
  if constexpr (GNA) {
    const int Ct = CONV ? Cin : (p.kapp ? p.kapp_k0 : p.K);  // channels the GroupNorm covers (an appended k_append segment stays raw)
    const int rps_out = CONV ? p.Ho * p.Wo : p.gin.rps;       // rows of this GEMM per sample
    const int b0 = m0 / rps_out, b1 = (min(m0 + BM, p.M) - 1) / rps_out;
    const int S = b1 - b0 + 1;
    one_sample = S == 1;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int m = min(m0 + 8 * (wave + NW * i) + lr, p.M - 1);
      trow[i] = (m / rps_out - b0) * Ct;
    }
    // (mean, rstd) of the S x groups slabs first -- into the head of the still empty ring --, then the per-channel pairs, 8 channels a thread
    float2* gst = reinterpret_cast<float2*>(smem);
    const double inv_count = 1.0 / ((double)p.gin.rps * (double)p.gin.cpg);
    for (int idx = tid; idx < S * p.gin.groups; idx += NW * 64) {
      const int sb = idx / p.gin.groups, g = idx - sb * p.gin.groups;
      float mean, rstd;
      gn_group_mean_rstd(p.gin.stats, b0 + sb, g, p.gin.nb, p.gin.groups, p.gin.reps, inv_count, p.gin.eps, mean, rstd);
      gst[idx] = make_float2(mean, rstd);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float2* wtab = reinterpret_cast<float2*>(smem + 3 * (A_BYTES + B_BYTES));
    const int C8 = Ct >> 3;
    for (int idx = tid; idx < S * C8; idx += NW * 64) {
      const int sb = idx / C8, c = (idx - sb * C8) * 8;
      const uint4 graw = *reinterpret_cast<const uint4*>(p.gin.gamma + c), braw = *reinterpret_cast<const uint4*>(p.gin.beta + c);
      const f16x8 gv = *reinterpret_cast<const f16x8*>(&graw), bv = *reinterpret_cast<const f16x8*>(&braw);
      int g = c / p.gin.cpg, gend = (g + 1) * p.gin.cpg;
      float2 ms = gst[sb * p.gin.groups + g];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (c + e >= gend) { ++g; gend += p.gin.cpg; ms = gst[sb * p.gin.groups + g]; }
        const float sc = ms.y * (float)gv[e];
        wtab[sb * Ct + c + e] = make_float2(sc, (float)bv[e] - ms.x * sc);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing of the table is in flight when the counted DMA ring starts
    __builtin_amdgcn_s_barrier();                                 // ... and gst (the ring's head) is free for the first DMA
  }
  // normalise the pieces of the tile in slot `buf` that THIS lane staged (call after the counted wait that retired them)
  auto norm_tile = [&](int buf) __attribute__((always_inline)) {
    if constexpr (GNA) {
      const unsigned vm = gi_m0;
      if (vm) {
        unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        const int c = gi_c0;
        const bool silu = p.gin.act == GN_ACT_SILU;
        float sc[8], sh[8];
        auto load_tab = [&](int base) __attribute__((always_inline)) {
          const f32x4* t4 = reinterpret_cast<const f32x4*>(tab + base + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 q = t4[e];
            sc[2 * e] = q[0]; sh[2 * e] = q[1]; sc[2 * e + 1] = q[2]; sh[2 * e + 1] = q[3];
          }
        };
        if (one_sample) load_tab(0);
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          if ((vm >> i) & 1u) {
            if (!one_sample) load_tab(trow[i]);
            f16x8* q = reinterpret_cast<f16x8*>(As + (wave + NW * i) * 1024 + lane * 16);
            *q = gn_apply8(*q, sc, sh, silu);
          }
        }
      }
    }
  };
  // dma_tile's side of the FIFO: the tile being issued has channel offset c (inside the GroupNorm's channel range) and live-piece bits m
  auto gna_push = [&](int c, unsigned m) __attribute__((always_inline)) {
    if constexpr (GNA) { gi_c0 = gi_c1; gi_m0 = gi_m1; gi_c1 = c; gi_m1 = m; }
  };

  auto dma_tile = [&](int buf) {
    const unsigned kmask = kcur < kend ? 0u : kOOB;  // OR-masks, not selects: every path must issue the same VMEM instructions
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
    if constexpr (CONV) {
      if (p.kapp && kt0 >= p.kapp_k0) {  // wave-uniform: the appended 1x1 segment (the same number of VMEM instructions on either side)
        int cs, cbase;
        const bool s2 = kapp_src(p, kt0, cs, cbase);
        const int co = kcur - cbase;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const unsigned voff = kapp_voff(p, iy0[i], ix0[i], pbase[i], cs, co) | kmask;
          lds_ptr_t dst = (lds_ptr_t)(As + (wave + NW * i) * 1024);
          if (s2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, dst, 16, voff, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a3, dst, 16, voff, 0, 0, 0);
        }
        gna_push(0, 0u);  // the appended segment is not under the GroupNorm
      } else {
        const bool first = p.kapp || cu < p.C1;  // wave-uniform: with two sources C1 % 64 == 0, so a K tile never straddles them
        const int cs = first ? p.C1 : p.C2;
        const int co = first ? cc : cc - p.C1;
        unsigned live = 0u;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
          const unsigned voff = ((unsigned)(pix[i] * cs + co) * 2u) | ((unsigned)(pix[i] >> 31) & kOOB) | kmask;
          lds_ptr_t dst = (lds_ptr_t)(As + (wave + NW * i) * 1024);
          if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, dst, 16, voff, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, dst, 16, voff, 0, 0, 0);
          if constexpr (GNA) live |= ((pix[i] >= 0 && kmask == 0u) ? 1u : 0u) << i;  // in the image and inside K: the piece holds data
        }
        gna_push(cc, live);
      }
    } else if (!LNF && p.kapp && kt0 >= p.kapp_k0) {  // dense k_append: the second operand's columns (wave-uniform: kapp_k0 % 64 == 0)
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const int m = m0 + 8 * (wave + NW * i) + lr;
        const unsigned voff = ((unsigned)((long)min(m, p.M - 1) * p.lda2 * 2) + (unsigned)(kcur - p.kapp_k0) * 2u) | amask[i] | kmask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(As + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
      }
      gna_push(0, 0u);
    } else {
      unsigned live = 0u;
#pragma unroll
      for (int i = 0; i < GA; ++i) {
        const unsigned voff = (aoff[i] + (unsigned)kcur * 2u) | amask[i] | kmask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(As + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
        if constexpr (GNA) live |= (((amask[i] | kmask) == 0u) ? 1u : 0u) << i;
      }
      gna_push(kcur, live);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const unsigned voff = (woff[i] + (unsigned)kcur * 2u) | wmask[i] | kmask;
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

  // 3-slot ring: tiles t+1 and t+2 are in flight while tile t is multiplied.  The wait is COUNTED -- vmcnt(GA + GB) retires tile t+1's
  // pieces and leaves tile t+2's in flight across the barrier -- and the barrier is the raw s_barrier (a __syncthreads() behind an
  // outstanding LDS-DMA drains vmcnt(0)).  WAR: slot (t+2) % 3 was last read in iteration t-1, before the barrier that ended it.
  constexpr int NIN = GA + GB;  // DMA instructions per wave per tile
  constexpr int kBudget = 512 / gemm_waves_per_simd(3 * (A_BYTES + B_BYTES), NW);
  constexpr bool RICH = epi_rich_fits(TM, TN, kBudget);
  constexpr bool PRE = RICH && epi_prefetch_fits(TM, TN, kBudget);
  EpiPre<TM, TN> pre;
  bool use_pre = false;
  if constexpr (PRE) use_pre = epilogue_prefetch<TM, TN>(p, pre, m0 + wm * WTM, n0 + wn * WTN, l31, hi);  // older than every DMA: retired by the first counted wait
  if constexpr (LNF) {  // the workgroup's c1 values -> the tail of the LDS object, before any DMA is in flight
    ln_c1_to_lds<BN>(p, reinterpret_cast<float*>(smem + 3 * (A_BYTES + B_BYTES)), n0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the raw s_barrier below does not wait for the ds_write)
  }
  dma_tile(0);
  dma_tile(1);  // past the last K tile the offsets are out of range: zero fill, no fetch -- the count stays the same on every path
  wait_vmcnt<NIN>();
  if constexpr (GNA) {
    norm_tile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the normalised pieces are in LDS before the barrier publishes the tile
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  LnStats<TM> lnst;  // LNF: row sums / sums of squares of the raw A rows (gemm_common.h ln_fold_apply)
  if constexpr (LNF) ln_stats_init(lnst);

  int cur = 0;
  // LNF: one specialised copy of the K loop per column wave, chosen ONCE outside it (wsel = whose share of the K steps the copy takes the
  // LayerNorm statistics on): conditional branches inside the loop cost issue slots even when they fall through (measured: DESIGN.md, round 3)
  auto k_loop = [&](auto wsel_c) __attribute__((always_inline)) {
    constexpr int WSEL = decltype(wsel_c)::value;
    for (int kt = 0; kt < nk; ++kt) {
      const int nxt2 = cur >= 1 ? cur - 1 : 2;  // (cur + 2) % 3
      dma_tile(nxt2);
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
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt<NIN>();
      if constexpr (GNA) {
        norm_tile(cur == 2 ? 0 : cur + 1);  // tile kt + 1 has landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == 2 ? 0 : cur + 1;
    }
  };
  if constexpr (LNF && WN > 1) {
    if (wn == 0) k_loop(std::integral_constant<int, 0>{});
    else if (WN > 2 && wn == 2) k_loop(std::integral_constant<int, 2 % WN>{});
    else if (WN > 2 && wn == 3) k_loop(std::integral_constant<int, 3 % WN>{});
    else k_loop(std::integral_constant<int, 1>{});
  } else {
    k_loop(std::integral_constant<int, 0>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead zero fills

  if constexpr (LNF) {
    __syncthreads();  // every wave's run-ahead fills have landed: the LDS ring is free for the row statistics
    ln_fold_apply<TM, TN, WN, BM>(p, acc, lnst, reinterpret_cast<float*>(smem), wm * WTM, wn,
                                  reinterpret_cast<const float*>(smem + 3 * (A_BYTES + B_BYTES)) + wn * WTN, l31, hi);
  }
  constexpr int SMEM = 3 * (A_BYTES + B_BYTES);
  if (!LNF && p.sink.stats) __syncthreads();  // (uniform) every wave's run-ahead zero fills have landed before the epilogue mirrors the tile into LDS
  gemm_epilogue<TM, TN, RICH>(p, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi, z, pre, PRE && use_pre, gemm_sink_lds<BM, BN, SMEM>(p, m0, n0, smem));
  gemm_sink_tail<NW * 64, BM, BN, SMEM>(p, m0, n0, smem);
}

// GNA launches: dynamic LDS = the ring + the scale / shift table of the samples a row tile can touch
template <int BM, int BN, int WM, int WN, bool CONV>
void launch_s3_gna(const GemmParams& p, dim3 grid, hipStream_t st) {
  const int rps_out = CONV ? p.Ho * p.Wo : p.gin.rps;
  const int S = rps_out >= BM ? 1 : BM / rps_out;
  const int Ct = CONV ? (p.kapp ? p.C1 : p.C1 + p.C2) : (p.kapp ? p.kapp_k0 : p.K);
  const size_t bytes = (size_t)3 * (BM + BN) * 128 + (size_t)S * Ct * sizeof(float2);
  static GnOncePerDevice attr_set;  // (per instantiation and device)
  if (attr_set.first())
    (void)hipFuncSetAttribute((const void*)gemm_s3_kernel<BM, BN, WM, WN, CONV, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((gemm_s3_kernel<BM, BN, WM, WN, CONV, false, true>), grid, dim3(WM * WN * 64), bytes, st, p);
}

template <int BM, int BN, int WM, int WN>
void launch_s3(const GemmParams& p, bool conv, dim3 grid, hipStream_t st) {
  if (p.gin.stats) {
    if (conv) launch_s3_gna<BM, BN, WM, WN, true>(p, grid, st);
    else launch_s3_gna<BM, BN, WM, WN, false>(p, grid, st);
  } else if (conv)
    hipLaunchKernelGGL((gemm_s3_kernel<BM, BN, WM, WN, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else if (p.ln_c1)
    hipLaunchKernelGGL((gemm_s3_kernel<BM, BN, WM, WN, false, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_s3_kernel<BM, BN, WM, WN, false>), grid, dim3(WM * WN * 64), 0, st, p);
}

}  // namespace

void gn_launch_gemm_s3(const void* params, int cfg, bool conv, int grid_x, int grid_y, int grid_z, hipStream_t st) {
  const GemmParams& p = *static_cast<const GemmParams*>(params);
  const dim3 grid(grid_x, grid_y, grid_z);
  switch (cfg) {
    case 0: launch_s3<128, 128, 2, 2>(p, conv, grid, st); break;
    case 1: launch_s3<128, 64, 2, 2>(p, conv, grid, st); break;
    case 2: launch_s3<64, 64, 2, 2>(p, conv, grid, st); break;
    case 3: launch_s3<256, 64, 4, 1>(p, conv, grid, st); break;
    // exact-fit tiles: N = 640 / 1280 / 320 problems whose 128x64 or 128x128 grids leave 256 CUs with 1.25 .. 2.5 workgroups each
    case 4: launch_s3<128, 160, 4, 1>(p, conv, grid, st); break;
    case 5: launch_s3<64, 160, 2, 1>(p, conv, grid, st); break;
    default: launch_s3<64, 320, 2, 2>(p, conv, grid, st); break;
  }
}
