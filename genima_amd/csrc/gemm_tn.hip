// Weight-gradient GEMM on operands in their NATURAL layout (no transposed copies, no im2col^T):
//   dW[n, k] += sum_r dY[r, n] * X[r, k]            (Linear; conv: X[r, k] = x[pixel(r) + tap(k), channel(k)], zero outside the image)
// Both operands are "reduction-major" -- the summed index r (a pixel / token) is the slow one -- which is the layout the forward pass
// leaves them in.  The previous path transposed dY and X (or materialised im2col^T) so that the forward kernel's K-contiguous
// loaders could read them: 435 transpose + 28 im2col^T launches and ~4.4 ms per train step.  Here the LDS image of a K tile is simply
// 64 rows of dY (BM channels each) and 64 rows of X (BN channels each), filled by LDS-DMA, and the MFMA operands -- 8 consecutive r
// for one channel -- come out of LDS through gfx950's transpose read: ds_read_b64_tr_b16 hands lane c of a 16-lane group column c
// of a 4-row x 16-column block whose rows the group's lanes address themselves (tools/probes/ds_read_tr_probe.hip pins the
// semantics), so two reads give a lane its v_mfma_f32_32x32x16_f16 fragment.
// Bank conflicts: a wave's transpose read touches 8 rows x two 32-byte column pairs; the rows of the image are swizzled at 32-byte
// pair granularity (applied on the SOURCE side of the lane-linear LDS-DMA) so that the 16 (row, pair) pieces spread over all banks.
// Reference op: the weight gradient autograd computes for nn.Linear / nn.Conv2d inside accelerator.backward(loss)
// (diffusion/train_controlnet_genima.py:1391).
#include <type_traits>

#include "gemm_common.h"

namespace {

typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) h4* lds_h4_ptr;
struct H8 { h4 lo, hi; };  // two transpose reads = one MFMA operand (8 consecutive reduction rows of one channel)

struct TnParams {
  GemmParams g;   // a = dY, w = X; M = dW rows (dY columns), N = dW columns, K = R (reduction rows); out = dW (f32, accumulate)
  int B, H, W, C, KH, KW, stride, pad, Ho, Wo;  // conv geometry (X = NHWC activations)
  float* sums_ws;  // optional [splitk][M]: column sums of dY over each row slice (bias / time-shift gradients ride on the A operand)
};

// chunk swizzle of a row of CPR 16-byte chunks.  A transpose read is served in two halves of 32 lanes, and one half is FOUR rows x 64
// contiguous bytes (rows 8 hb + 0 .. 3, the 32 columns of a fragment): conflict-free when the four 64-byte segments fall into the four
// 64-byte quarters of the 256 bytes the 64 banks span.  So the XOR moves whole quarters (chunk bits 2, 3), never pieces inside one.
// (Round 5: the first version XORed the 32-byte pair index -- rows r, r + 1 stayed in one quarter: SQ_LDS_BANK_CONFLICT = 50 % of
//  SQ_LDS_IDX_ACTIVE on both tiles, profiles/r05_v14_wgrad_pmc.txt; with four LDS instructions per MFMA on the 64 x 64 tile the LDS, not
//  the matrix pipe, was what the kernel waited for.)
template <int CPR>
__device__ __forceinline__ int tn_swz(int row) {
  // (only row bits 0 and 1 may enter: a fragment's second read is 4 rows further, the other half-wave 8, the next k16 step 16 rows
  //  further, and all must keep the swizzle so that one base address + immediates walks the tile)
  if constexpr (CPR >= 16) return (row & 3) << 2;   // 256-byte rows: every row starts at bank 0 -> row r's segment goes to quarter q ^ (r & 3)
  else return ((row >> 1) & 1) << 2;              // 128-byte rows: odd rows already sit on the other half of the banks
}

template <int BM, int BN, bool CONV, int WM = 2, int WN = 2>
__global__ __launch_bounds__(WM* WN * 64, gemm_waves_per_simd(2 * (BM + BN) * 128, WM* WN)) void gemm_tn_kernel(const TnParams tp) {
  const GemmParams& p = tp.g;
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int CPA = BM / 8, CPB = BN / 8;            // 16-byte chunks per LDS row
  constexpr int RPA = 64 / CPA, RPB = 64 / CPB;        // rows one DMA instruction (64 lanes x 16 B) fills
  constexpr int GA = 64 / RPA / NW, GB = 64 / RPB / NW;  // DMA instructions per wave per K tile
  constexpr int A_BYTES = 64 * BM * 2, B_BYTES = 64 * BN * 2;
  static_assert(TM >= 1 && TN >= 1 && GA >= 1 && GB >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;   // dW row / column origin of this workgroup
  const int z = blockIdx.y;
  const long rbeg = (long)z * p.kper;
  const long rend = min((long)p.K, rbeg + p.kper);
  const int nk = (int)((rend - rbeg + 63) / 64);

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

  // ---- loader state.  A piece i of this wave: LDS rows (wave + NW * i) * RPA + lane / CPA, physical chunk lane % CPA
  const int arow = lane / CPA, apc = lane % CPA;
  const int brow = lane / CPB, bpc = lane % CPB;
  // conv: every 16-byte piece (8 channels) lies inside one tap (C % 8 == 0); a piece's LDS row -- hence its swizzled column, tap and
  // channel -- is the same for every K tile
  int tdy[GB], tdx[GB], tcc[GB];
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int lrow = (wave + NW * i) * RPB + brow;
      const int col = n0 + ((bpc ^ tn_swz<CPB>(lrow)) << 3);
      const int tap = col / tp.C;
      tcc[i] = col - tap * tp.C;
      tdy[i] = tap / tp.KW;
      tdx[i] = tap - tdy[i] * tp.KW;
    }
  }
  // conv: (b, oy, ox) of the row each B piece loads, advanced by 64 rows per K tile without divisions
  int pb[GB], py[GB], px[GB];
  // 64 rows = qb64 whole images + qy64 image rows + qx64 pixels (mixed radix Ho * Wo, Wo): exact for ANY feature-map size -- a 4 x 4 map
  // (Ho * Wo = 16) wraps the image index four times per K tile
  const int hw64 = CONV ? tp.Ho * tp.Wo : 1;
  const int qb64 = CONV ? 64 / hw64 : 0, rem64 = CONV ? 64 % hw64 : 0;
  const int qy64 = CONV ? rem64 / tp.Wo : 0, qx64 = CONV ? rem64 % tp.Wo : 0;
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const long r = rbeg + (wave + NW * i) * RPB + brow;
      const int hw = tp.Ho * tp.Wo;
      const int b = (int)(r / hw), rem = (int)(r - (long)b * hw);
      pb[i] = b; py[i] = rem / tp.Wo; px[i] = rem - py[i] * tp.Wo;
    }
  }

  auto dma_tile = [&](int buf, long r0) {
    unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int lrow = (wave + NW * i) * RPA + arow;
      const long r = r0 + lrow;
      const int col = m0 + ((apc ^ tn_swz<CPA>(lrow)) << 3);
      unsigned voff = (r < rend && col < p.M) ? (unsigned)((r * p.lda + col) * 2) : kOOB;
      GN_PIN(voff);  // one value in one register: no exec-masked arms with a DMA each (gemm_common.h)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(As + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int lrow = (wave + NW * i) * RPB + brow;
      const long r = r0 + lrow;
      const int lc = (bpc ^ tn_swz<CPB>(lrow)) << 3;
      unsigned voff = kOOB;
      if constexpr (CONV) {
        const int iy = py[i] * tp.stride - tp.pad + tdy[i], ix = px[i] * tp.stride - tp.pad + tdx[i];
        const bool ok = r < rend && n0 + lc < p.N && (unsigned)iy < (unsigned)tp.H && (unsigned)ix < (unsigned)tp.W;
        voff = ok ? (unsigned)(((((long)pb[i] * tp.H + iy) * tp.W + ix) * tp.C + tcc[i]) * 2) : kOOB;
        // next K tile: 64 rows further -- selects, not branches: each digit carries at most once (qx64 < Wo, qy64 < Ho)
        px[i] += qx64;
        const int cx = px[i] >= tp.Wo ? 1 : 0;
        px[i] -= cx ? tp.Wo : 0;
        py[i] += qy64 + cx;
        const int cy = py[i] >= tp.Ho ? 1 : 0;
        py[i] -= cy ? tp.Ho : 0;
        pb[i] += qb64 + cy;
      } else {
        voff = (r < rend && n0 + lc < p.N) ? (unsigned)((r * p.ldw + n0 + lc) * 2) : kOOB;
      }
      GN_PIN(voff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(Bs + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
    }
  };

  // ---- fragment addresses: lane (i = lane & 15, g = (lane >> 4) & 1, hb = lane >> 5) of the transpose read supplies the 8 bytes at
  // row 8 hb + i / 4 (+ 4 for the second read, + 16 per k16 step), columns 16 g + 4 (i % 4) of a 32-column tile
  const int ti = lane & 15, tg = (lane >> 4) & 1;
  const int frow = 8 * hi + (ti >> 2);
  auto frag_off = [&](int cpr_swz, int col, int rowb) {  // byte offset inside a tile image
    const int chunk = (col >> 3) ^ cpr_swz;
    return frow * rowb + (chunk << 4) + ((col & 7) << 1);
  };
  int aoff[TM], boff[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) aoff[i] = frag_off(tn_swz<CPA>(frow), wm * WTM + i * 32 + 16 * tg + 4 * (ti & 3), BM * 2);
#pragma unroll
  for (int j = 0; j < TN; ++j) boff[j] = frag_off(tn_swz<CPB>(frow), wn * WTN + j * 32 + 16 * tg + 4 * (ti & 3), BN * 2);

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  const bool do_sums = tp.sums_ws != nullptr && tile_n == 0 && wn == 0;
  float csum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csum[i] = 0.0f;

  dma_tile(0, rbeg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  // one branch per K tile (the back edge): the last tile is peeled and the loop exists in two copies, with and without the column sums,
  // chosen once outside it -- conditional branches inside the loop cost issue slots even when they fall through (measured: DESIGN.md, round 3)
  auto k_tile = [&](auto sums_c) __attribute__((always_inline)) {
    constexpr bool SUMS = decltype(sums_c)::value;
    const unsigned char* As = smem + cur * (A_BYTES + B_BYTES);
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f16x8 fa[TM], fw[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned char* q = As + aoff[i] + kk * 16 * (BM * 2);
        const h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4_ptr)q);
        const h4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4_ptr)(q + 4 * (BM * 2)));
        fa[i] = __builtin_bit_cast(f16x8, H8{lo, hi4});
      }
      if constexpr (SUMS) {  // the first column tile's wn = 0 waves also sum the dY fragments they hold anyway
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[i] += (float)fa[i][e];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned char* q = Bs + boff[j] + kk * 16 * (BN * 2);
        const h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4_ptr)q);
        const h4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4_ptr)(q + 4 * (BN * 2)));
        fw[j] = __builtin_bit_cast(f16x8, H8{lo, hi4});
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[j][i], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  };
  auto k_loop = [&](auto sums_c) __attribute__((always_inline)) {
    for (int kt = 0; kt + 1 < nk; ++kt) {
      dma_tile(cur ^ 1, rbeg + (long)(kt + 1) * 64);
      k_tile(sums_c);
    }
    k_tile(sums_c);
  };
  if (do_sums) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  if (do_sums) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float v = csum[i] + __shfl_xor(csum[i], 32);  // lanes l and l + 32 hold the two 8-row halves of a k16 step
      const int m = m0 + wm * WTM + i * 32 + l31;
      if (hi == 0 && m < p.M) tp.sums_ws[(long)z * p.M + m] = v;
    }
  }
  GemmParams pe = p;
  gemm_epilogue<TM, TN>(pe, acc, m0 + wm * WTM, n0 + wn * WTN, l31, hi, z);
}

// One finishing launch: dW += sum of the f32 partial slabs (fixed order), and the column-sum slices folded into dbias (one group) /
// the per-sample time-shift gradient (shift_groups groups).  Blocks [0, nb_dw) reduce slabs, the next nb_bias the bias, the rest dshift.
struct TnFinish {
  const float* ws; float* dw; long M; int N; long ldo; int splitk;   // slabs (null: nothing to reduce)
  const float* sums; float* dbias; float* dshift; int Ns, groups, spg; // column sums [splitk][Ns]
  int nb_dw, nb_bias;
};
__global__ __launch_bounds__(256) void tn_finish_kernel(const TnFinish f) {
  int blk = blockIdx.x;
  if (blk < f.nb_dw) {
    const long n4 = f.N >> 2;
    const long idx = (long)blk * 256 + threadIdx.x;
    if (idx >= f.M * n4) return;
    const long m = idx / n4;
    const int nb = (int)(idx - m * n4) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int zz = 0; zz < f.splitk; ++zz) s += *reinterpret_cast<const f32x4*>(f.ws + ((long)zz * f.M + m) * f.N + nb);
    float* o = f.dw + m * f.ldo + nb;
    *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(o) + s;
    return;
  }
  blk -= f.nb_dw;
  float* out = f.dbias;
  int groups = 1, spg = f.splitk;
  if (blk >= f.nb_bias) { blk -= f.nb_bias; out = f.dshift; groups = f.groups; spg = f.spg; }
  const long idx = (long)blk * 256 + threadIdx.x;
  if (idx >= (long)groups * f.Ns) return;
  const int g = (int)(idx / f.Ns), n = (int)(idx - (long)g * f.Ns);
  float s = 0.f;
  for (int zz = 0; zz < spg; ++zz) s += f.sums[((long)g * spg + zz) * f.Ns + n];
  out[idx] += s;
}

struct TnPlan { int bm, bn, splitk; long kper; };
TnPlan tn_plan(const gn_wgrad_desc* d) {
  TnPlan pl;
  const bool small = d->tile == 2 || (d->tile == 0 && (d->N < 128 || d->K < 128));
  pl.bm = pl.bn = small ? 64 : 128;
  // tile 3 (round 6): 128 x 256 on eight waves of 64 x 64 -- one workgroup per CU (96 KB of LDS), a third more multiply-adds per byte moved into LDS than
  // 128 x 128; for the wide weight gradients (K = 9 C columns of a 3x3 conv, the feed-forward projections)
  if (d->tile == 3 && d->N >= 128 && d->K >= 256) pl.bn = 256;
  if (d->tile == 4 && d->N >= 256 && d->K >= 128) pl.bm = 256;  // 256 x 128: the same for tall gradients (N = 640 / 1280 rows are whole tiles)
  const long blocks = ((d->N + pl.bm - 1) / pl.bm) * ((d->K + pl.bn - 1) / pl.bn);
  int sk = d->splitk;
  if (sk <= 0) {  // a handful of output tiles under a reduction over every pixel of the batch: the row split supplies the parallelism
    sk = (int)((1024 + blocks - 1) / blocks);
    const long maxsk = d->R / 512;
    if (sk > maxsk) sk = (int)maxsk;
    if (sk > 256) sk = 256;
    if (sk < 1) sk = 1;
  }
  long kper = ((d->R + sk - 1) / sk + 63) / 64 * 64;
  if (d->dshift && d->shift_groups > 0) {  // per-sample sums: a row slice must not straddle two samples
    const long hw = d->R / d->shift_groups;
    long spg = (hw + kper - 1) / kper;  // slices per sample
    while (spg < hw / 64 && (hw % spg != 0 || (hw / spg) % 64 != 0)) ++spg;
    kper = hw / spg;
  }
  pl.splitk = (int)((d->R + kper - 1) / kper);
  pl.kper = kper;
  return pl;
}

}  // namespace

extern "C" int64_t gn_wgrad_workspace_bytes(const gn_wgrad_desc* d) {
  if (!d) return 0;
  const TnPlan pl = tn_plan(d);
  const int64_t slabs = pl.splitk > 1 ? (int64_t)pl.splitk * d->N * d->K : 0;
  const int64_t sums = (d->dbias || d->dshift) ? (int64_t)pl.splitk * d->N : 0;
  return (slabs + sums) * (int64_t)sizeof(float);
}

extern "C" int32_t gn_wgrad(gn_ctx* ctx, const gn_wgrad_desc* d) {
  GN_REQUIRE(ctx && d && d->dy && d->x && d->dw, "gn_wgrad: null pointer");
  GN_REQUIRE(d->R > 0 && d->N > 0 && d->K > 0 && d->N % 8 == 0 && d->K % 8 == 0 && d->ld_dy % 8 == 0 && d->ld_dy >= d->N && d->ld_dw % 4 == 0 &&
             d->ld_dw >= d->K, "gn_wgrad: N, K, ld_dy must be multiples of 8 and the strides cover the rows");
  GN_REQUIRE(((uintptr_t)d->dy & 15) == 0 && ((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->dw & 15) == 0, "gn_wgrad: 16-byte alignment");
  TnParams tp;
  GemmParams& p = tp.g;
  p = GemmParams();
  p.a = (const f16*)d->dy; p.w = (const f16*)d->x; p.out = (f16*)d->dw; p.ws = (float*)d->workspace;
  p.M = (int)d->N; p.N = (int)d->K; p.K = (int)d->R;
  p.lda = d->ld_dy; p.ldo = d->ld_dw;
  p.out_mode = GN_OUT_F32; p.accumulate = 1; p.act = GN_ACT_NONE; p.out_scale = 1.0f; p.rpb = (int)d->N;
  GN_REQUIRE((uint64_t)d->R * d->ld_dy * 2 < 0xFFFFFF00ull, "gn_wgrad: dY too large for 32-bit buffer offsets");
  p.a_bytes = (unsigned)((uint64_t)d->R * d->ld_dy * 2);
  if (d->conv) {
    GN_REQUIRE(d->C > 0 && d->C % 8 == 0 && d->K == (int64_t)d->KH * d->KW * d->C && d->R == (int64_t)d->B * d->Ho * d->Wo && d->stride >= 1,
               "gn_wgrad(conv): C %% 8 == 0, K = KH*KW*C, R = B*Ho*Wo");
    const uint64_t xb = (uint64_t)d->B * d->H * d->W * d->C * 2;
    GN_REQUIRE(xb < 0xFFFFFF00ull, "gn_wgrad(conv): x too large for 32-bit buffer offsets");
    p.w_bytes = (unsigned)xb;
    tp.B = d->B; tp.H = d->H; tp.W = d->W; tp.C = d->C; tp.KH = d->KH; tp.KW = d->KW; tp.stride = d->stride; tp.pad = d->pad; tp.Ho = d->Ho; tp.Wo = d->Wo;
  } else {
    GN_REQUIRE(d->ld_x % 8 == 0 && d->ld_x >= d->K && (uint64_t)d->R * d->ld_x * 2 < 0xFFFFFF00ull, "gn_wgrad: ld_x must be a multiple of 8, >= K, x within 4 GB");
    p.ldw = d->ld_x;
    p.w_bytes = (unsigned)((uint64_t)d->R * d->ld_x * 2);
    tp.B = tp.H = tp.W = tp.C = tp.KH = tp.KW = tp.stride = tp.pad = tp.Ho = tp.Wo = 0;
  }
  const TnPlan pl = tn_plan(d);
  p.splitk = pl.splitk; p.kper = (int)pl.kper;
  const bool want_sums = d->dbias || d->dshift;
  if (pl.splitk > 1 || want_sums) GN_REQUIRE(d->workspace, "gn_wgrad: split over rows (%d) / column sums need a workspace of gn_wgrad_workspace_bytes()", pl.splitk);
  if (d->dshift) {
    GN_REQUIRE(d->shift_groups > 0 && d->R % d->shift_groups == 0 && (d->R / d->shift_groups) % 64 == 0,
               "gn_wgrad: dshift needs R / shift_groups (%ld / %d) to be a multiple of 64", (long)d->R, d->shift_groups);
    GN_REQUIRE((d->R / d->shift_groups) % pl.kper == 0, "gn_wgrad: row slices (%ld) do not tile a sample", pl.kper);
  }
  tp.sums_ws = want_sums ? (float*)d->workspace + (pl.splitk > 1 ? (int64_t)pl.splitk * d->N * d->K : 0) : nullptr;
  p.tiles_m = (int)((d->N + pl.bm - 1) / pl.bm); p.tiles_n = (int)((d->K + pl.bn - 1) / pl.bn);
  const dim3 grid(p.tiles_m * p.tiles_n, pl.splitk, 1);
  if (pl.bn == 256) {
    if (d->conv) hipLaunchKernelGGL((gemm_tn_kernel<128, 256, true, 2, 4>), grid, dim3(512), 0, ctx->stream, tp);
    else hipLaunchKernelGGL((gemm_tn_kernel<128, 256, false, 2, 4>), grid, dim3(512), 0, ctx->stream, tp);
  } else if (pl.bm == 256) {
    if (d->conv) hipLaunchKernelGGL((gemm_tn_kernel<256, 128, true, 4, 2>), grid, dim3(512), 0, ctx->stream, tp);
    else hipLaunchKernelGGL((gemm_tn_kernel<256, 128, false, 4, 2>), grid, dim3(512), 0, ctx->stream, tp);
  } else if (pl.bm == 128) {
    if (d->conv) hipLaunchKernelGGL((gemm_tn_kernel<128, 128, true>), grid, dim3(256), 0, ctx->stream, tp);
    else hipLaunchKernelGGL((gemm_tn_kernel<128, 128, false>), grid, dim3(256), 0, ctx->stream, tp);
  } else {
    if (d->conv) hipLaunchKernelGGL((gemm_tn_kernel<64, 64, true>), grid, dim3(256), 0, ctx->stream, tp);
    else hipLaunchKernelGGL((gemm_tn_kernel<64, 64, false>), grid, dim3(256), 0, ctx->stream, tp);
  }
  GN_LAUNCH_CHECK();
  TnFinish f;
  f.ws = (const float*)d->workspace; f.dw = d->dw; f.M = (long)d->N; f.N = (int)d->K; f.ldo = (long)d->ld_dw; f.splitk = pl.splitk;
  f.sums = tp.sums_ws; f.dbias = d->dbias; f.dshift = d->dshift; f.Ns = (int)d->N; f.groups = d->dshift ? d->shift_groups : 1;
  f.spg = d->dshift ? pl.splitk / d->shift_groups : pl.splitk;
  f.nb_dw = pl.splitk > 1 ? (int)(((long)d->N * (d->K >> 2) + 255) / 256) : 0;
  f.nb_bias = d->dbias ? (int)((d->N + 255) / 256) : 0;
  const int nb_shift = d->dshift ? (int)(((long)d->shift_groups * d->N + 255) / 256) : 0;
  if (f.nb_dw + f.nb_bias + nb_shift > 0) {
    hipLaunchKernelGGL(tn_finish_kernel, dim3((unsigned)(f.nb_dw + f.nb_bias + nb_shift)), dim3(256), 0, ctx->stream, f);
    GN_LAUNCH_CHECK();
  }
  return GN_OK;
}
