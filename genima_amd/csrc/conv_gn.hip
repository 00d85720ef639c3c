// 3x3 convolution with its input patch resident in LDS and GroupNorm-apply + SiLU fused into the prologue (gfx950).
//
//   out[b, y, x, :] = bias + sum_{tap, c} act(x[b, y + dy - 1, x + dx - 1, c] * scale[b, c] + shift[b, c]) * w[:, tap * Cin + c]  (+ residual)
//
// The diffusers ResnetBlock2D runs  GroupNorm -> SiLU -> conv3x3  twice (the VAE decoder inside `self.pipe(...)`,
// controller/agent/sd_controlnet_agent.py:67-76; SURVEY.md section 8 a6.7 / north_star "Conv2d/GroupNorm/SiLU ... fusions with LDS-staged
// input tiles").  As separate launches the normalised tensor makes a round trip through HBM (one write + one read of the conv's whole
// input: 0.54 GB per conv at the VAE's 512^2 x 128-channel level) and the implicit-GEMM conv stages every input pixel nine times.  Here:
//   * a workgroup owns an 8 x 16 tile of output pixels (128 GEMM rows) x 128 output channels;
//   * its 10 x 18 input patch (halo included) of a 128-channel group is DMA'd into LDS ONCE (`buffer_load ... lds`, 256-byte pixels, the
//     sixteen 16-byte channel chunks of a pixel XOR-swizzled by the pixel's patch column so that the MFMA fragment reads of 16 neighbouring
//     pixels are conflict-free; pixels outside the image land as zeros = the conv's zero padding);
//   * the GroupNorm's per-(sample, channel) scale / shift (gn_groupnorm_fwd with y == NULL: statistics only) and the SiLU are applied to the
//     patch IN LDS, once per staged element (1.4x the tile's pixels instead of 9x), in f32 as gn_apply_kernel (norm.hip) does and rounded
//     to f16 where that launch rounds (SiLU on v_rcp_f32 instead of the IEEE division: equal after the rounding but for rare ties);
//   * the nine taps then read their A fragments straight out of the patch (a tap is an address offset), only the weights stream:
//     [128 x 64] K tiles through a 2-stage LDS-DMA ring;
//   * inputs wider than 128 channels walk their channel groups one patch at a time under the same accumulators.
// Two workgroups of four waves share a CU (78 KB of LDS each), so one's patch load / normalisation runs under the other's MFMAs.
// K order: channel group, tap, channel -- a different summation order than gn_gemm's (tap, channel): results agree to f32 rounding.
#include "gemm_common.h"

namespace {

constexpr int CG_TH = 8, CG_TW = 16;                      // output tile (pixels)
constexpr int CG_PH = CG_TH + 2, CG_PW = CG_TW + 2;       // patch with halo
constexpr int CG_NPIX = CG_PH * CG_PW;                    // 180
constexpr int CG_CG = 128;                                // channels per patch (one channel group)
constexpr int CG_PATCH = CG_NPIX * CG_CG * 2;             // 46080 bytes
constexpr int CG_W_OFF = CG_PATCH;                        // behind the patch: two weight K tiles [BN rows x 64 k] (BN = 128: 2 x 16 KB)
static_assert(2 * (CG_W_OFF + 2 * 128 * 128) <= 160 * 1024, "two workgroups per CU");

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct CgParams {
  const f16* x;
  const float* scsh;   // [B][Cin][2] (scale, shift) or nullptr: plain conv
  const f16* w;        // [Cout][9 * Cin]
  const f16* bias;     // [Cout] or nullptr
  const f16* res;      // [B * H * W, ldr] or nullptr
  f16* out;            // [B * H * W, ldo]
  long ldr, ldo;
  int B, H, W, Cin, Cout, silu;
  int tiles_x, tiles_y, tiles_n;
  unsigned x_bytes, w_bytes;
};

// BN = 128: the ResNet convs (2 x 2 waves, 64 x 64 wave tiles).  BN = 32: convs with a handful of output channels -- the VAE's conv_out (3 of 8
// padded channels) behind conv_norm_out + SiLU: 4 x 1 waves of one 32 x 32 tile each, the launch is the patch load + normalisation + one read of x
template <int BN>
__device__ __forceinline__ void conv3x3_gn_body(const CgParams& p, unsigned char* smem) {
  constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN, TM = 128 / (32 * WM), TN = BN / (32 * WN);
  constexpr int CG_WT = BN * 128;  // one weight K tile [BN rows x 64 k]
  constexpr int GB = BN / 32;      // weight DMA instructions per wave and K tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware block -> tile map (block b runs on XCD b % 8): an XCD walks a contiguous run of tiles, the N tiles of one patch side by
  // side, then the next patch along the image row -- neighbours share their halo and their weights in one L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid % p.tiles_n;
  int t = bid / p.tiles_n;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int y0 = ty * CG_TH, x0 = tx * CG_TW, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

  // ---- weight loader (gemm_dma_kernel's: 8 rows x 128 bytes per instruction, the XOR swizzle of lds_swz<128> on the source side)
  const int lr = lane >> 3;
  const int wchunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  const long ldw = 9l * p.Cin;
  unsigned woff[GB];  // (rows past Cout start past the buffer's end: they land as zeros)
#pragma unroll
  for (int i = 0; i < GB; ++i) woff[i] = (unsigned)(((long)(n0 + 8 * (wave + 4 * i) + lr) * ldw + wchunk * 8) * 2);
  auto dma_w = [&](int stage, int kelem) __attribute__((always_inline)) {
    unsigned char* Ws = smem + CG_W_OFF + stage * CG_WT;
#pragma unroll
    for (int i = 0; i < GB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Ws + (wave + 4 * i) * 1024), 16, woff[i] + (unsigned)kelem * 2u, 0, 0, 0);
  };

  // ---- A fragment addressing: GEMM row m = 16 py + px of the tile; its tap (dy, dx) pixel sits at patch index (py + dy) * 18 + px + dx
  int pidx0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = wm * (32 * TM) + i * 32 + l31;
    pidx0[i] = (m >> 4) * CG_PW + (m & 15);
  }

  f32x16 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  // residual rows of this wave's tiles, requested ahead of the K loop (raw 16-byte pieces; the lane swap waits for the data: epilogue)
  long orow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = wm * (32 * TM) + i * 32 + l31;
    orow[i] = ((long)b * p.H + y0 + (m >> 4)) * p.W + x0 + (m & 15);
  }
  u32x4v rraw[TM][TN][2];
  if (BN == 128 && p.res) {  // (the narrow variant takes no residual)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          rraw[i][j][h] = *reinterpret_cast<const u32x4v*>(p.res + orow[i] * p.ldr + n0 + wn * (32 * TN) + j * 32 + 16 * h + 8 * hi);
  }

  const int ngroups = p.Cin / CG_CG;
  for (int cg = 0; cg < ngroups; ++cg) {
    if (cg > 0) {  // every wave is done with the previous group's patch
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // ---- the patch: 45 instructions of 4 pixels x 256 bytes; lane q of one lands at physical chunk q & 15 of pixel 4 t + (q >> 4) and
    // therefore fetches logical chunk (q & 15) ^ (patch column & 15).  The key is the pixel's COLUMN, not its linear index: a fragment
    // read's 16-lane group is pixels x = 0..3, 12..15 of one tile row and x = 4..11 of the next, whose linear indices (row pitch 18)
    // collide mod 16 at two lanes -- SQ_LDS_BANK_CONFLICT was 31 % of the kernel's LDS cycles (round 5, tools/probes/lds_conflict_pmc.sh);
    // by column the group's keys are 16 consecutive values for every tap
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int tt = wave + 4 * i;
      if (tt < CG_NPIX / 4) {
        const int idx = 4 * tt + (lane >> 4);
        const int pr = idx / CG_PW, pc = idx - pr * CG_PW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int lc = (lane & 15) ^ (pc & 15);
        unsigned voff = ok ? (unsigned)(((((long)b * p.H + gy) * p.W + gx) * p.Cin + cg * CG_CG + lc * 8) * 2) : kOOB;
        GN_PIN(voff);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + tt * 1024), 16, voff, 0, 0, 0);
      }
    }
    dma_w(0, cg * CG_CG);  // tap 0, first 64 channels of the group
    float gsc[8], gsh[8];  // (scale, shift) of this thread's 8 channels of the group (normalisation pass below)
    if (p.scsh) {
      const f32x4* sp = reinterpret_cast<const f32x4*>(p.scsh + ((long)b * p.Cin + cg * CG_CG + (tid & 15) * 8) * 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 s4 = sp[e];
        gsc[2 * e] = s4[0]; gsh[2 * e] = s4[1]; gsc[2 * e + 1] = s4[2]; gsh[2 * e + 1] = s4[3];
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- GroupNorm-apply (+ SiLU) on the patch, in place; pixels outside the image stay zero (the conv pads the NORMALISED tensor).
    // A thread keeps ONE logical 16-byte chunk (8 channels: their scale / shift live in registers, requested before the patch wait) and
    // walks the pixels 16 apart; the 16 threads of a pixel cover its 256 bytes (conflict-free whatever the swizzle).
    if (p.scsh) {
      const int lcn = tid & 15;
#pragma unroll 2
      for (int idx = tid >> 4; idx < CG_NPIX; idx += 16) {
        const int pr = idx / CG_PW, pcx = idx - pr * CG_PW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pcx;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
          unsigned char* q = smem + idx * 256 + ((lcn ^ (pcx & 15)) << 4);
          const f16x8 v = *reinterpret_cast<const f16x8*>(q);
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float yv = (float)v[e] * gsc[e] + gsh[e];
            // SiLU as x * rcp(1 + exp(-x)) on v_rcp_f32 (1 ulp): the IEEE division of act_silu is ~10 VALU per element, and this pass runs beside
            // the other workgroup's MFMAs on the same SIMDs; after the rounding to f16 the two agree except on a 1e-4 fraction of ties
            if (p.silu) yv = yv * __builtin_amdgcn_rcpf(1.0f + __expf(-yv));
            o[e] = (f16)yv;
          }
          *reinterpret_cast<f16x8*>(q) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 9 taps x 2 K tiles of 64 channels; weights double-buffered (the next tile lands under this tile's MFMAs)
    for (int kt = 0; kt < 18; ++kt) {
      const int tap = kt >> 1, kc = kt & 1;
      if (kt + 1 < 18) dma_w((kt + 1) & 1, ((kt + 1) >> 1) * p.Cin + cg * CG_CG + ((kt + 1) & 1) * 64);
      const int dy = tap / 3, dx = tap - dy * 3;
      const int toff = dy * CG_PW + dx;
      const int key = ((l31 & 15) + dx) & 15;  // swizzle key of the tapped pixel = its patch COLUMN (below)
      const unsigned char* Ws = smem + CG_W_OFF + (kt & 1) * CG_WT;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cl = kc * 8 + kk * 2 + hi;  // logical 16-byte chunk of the pixel's 128 channels
        f16x8 fa[TM], fw[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int idx = pidx0[i] + toff;
          fa[i] = *reinterpret_cast<const f16x8*>(smem + idx * 256 + ((cl ^ key) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) fw[j] = *reinterpret_cast<const f16x8*>(Ws + lds_swz<128>(wn * (32 * TN) + j * 32 + l31, kk * 2 + hi));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa[i], acc[j][i], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: + bias (+ residual) -> f16, 16-byte row stores (lanes l / l + 32 trade halves, as gemm_common.h's wide path)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    f16* orw = p.out + orow[i] * p.ldo;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + wn * (32 * TN) + j * 32;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[j][i][e];
      if (p.bias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (BN == 128 || nb + 8 * g + 4 * hi + 4 <= p.Cout) {
            const f16x4 bb = *reinterpret_cast<const f16x4*>(p.bias + nb + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * g + e] += (float)bb[e];
          }
        }
      }
      if (BN == 128 && p.res) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4v r = rraw[i][j][h];
          const auto r0 = __builtin_amdgcn_permlane32_swap(r[0], r[2], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(r[1], r[3], false, false);
          const uint2 lo = make_uint2(r0[0], r1[0]), hi2 = make_uint2(r0[1], r1[1]);
          const f16x4 ga = *reinterpret_cast<const f16x4*>(&lo), gb = *reinterpret_cast<const f16x4*>(&hi2);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[8 * h + e] += (float)ga[e]; v[8 * h + 4 + e] += (float)gb[e]; }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        f16x4 ha, hb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ha[e] = (f16)v[4 * g + e]; hb[e] = (f16)v[4 * g + 4 + e]; }
        const uint2 ua = *reinterpret_cast<const uint2*>(&ha), ub = *reinterpret_cast<const uint2*>(&hb);
        const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
        if (BN == 128 || nb + 8 * g + 8 * hi + 8 <= p.Cout) *reinterpret_cast<u32x4v*>(orw + nb + 8 * g + 8 * hi) = u32x4v{r0[0], r1[0], r0[1], r1[1]};
      }
    }
  }
}

// (two plain kernels around the templated body: each owns its ONE LDS object -- see tblock.hip on hipcc's vmcnt drains)
__global__ __launch_bounds__(256, 2) void conv3x3_gn_kernel(const CgParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CG_W_OFF + 2 * 128 * 128];
  conv3x3_gn_body<128>(p, smem);
}
__global__ __launch_bounds__(256, 2) void conv3x3_gn_narrow_kernel(const CgParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[CG_W_OFF + 2 * 32 * 128];
  conv3x3_gn_body<32>(p, smem);
}

}  // namespace

extern "C" int32_t gn_conv3x3_gn_supported(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
  return B > 0 && H > 0 && W > 0 && H % CG_TH == 0 && W % CG_TW == 0 && Cin >= CG_CG && Cin % CG_CG == 0 && (Cout % 128 == 0 || (Cout % 8 == 0 && Cout > 0 && Cout <= 32)) &&
                 (int64_t)B * H * W * Cin * 2 < 0xFFFFFF00ll && (int64_t)Cout * 9 * Cin * 2 < 0xFFFFFF00ll
             ? 1 : 0;
}

extern "C" int32_t gn_conv3x3_gn(gn_ctx* ctx, const gn_conv3x3_gn_desc* d) {
  GN_REQUIRE(ctx && d && d->x && d->w && d->out, "gn_conv3x3_gn: null ctx / x / w / out");
  GN_REQUIRE(gn_conv3x3_gn_supported(d->B, d->H, d->W, d->Cin, d->Cout),
             "gn_conv3x3_gn: needs H %% 8 == 0, W %% 16 == 0, Cin %% 128 == 0, Cout %% 128 == 0 or Cout in {8, 16, 24, 32} (got %dx%d, %d -> %d); use gn_gemm otherwise", d->H, d->W,
             d->Cin, d->Cout);
  GN_REQUIRE(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->w & 15) == 0 && ((uintptr_t)d->out & 15) == 0 && d->ldo % 8 == 0 && d->ldo >= d->Cout,
             "gn_conv3x3_gn: x / w / out must be 16-byte aligned, ldo a multiple of 8 and >= Cout");
  if (d->residual) GN_REQUIRE(((uintptr_t)d->residual & 15) == 0 && d->ldr % 8 == 0 && d->ldr >= d->Cout, "gn_conv3x3_gn: residual alignment / stride");
  if (d->scsh) GN_REQUIRE(((uintptr_t)d->scsh & 15) == 0, "gn_conv3x3_gn: scsh must be 16-byte aligned");
  CgParams p;
  p.x = (const f16*)d->x; p.scsh = (const float*)d->scsh; p.w = (const f16*)d->w; p.bias = (const f16*)d->bias; p.res = (const f16*)d->residual;
  p.out = (f16*)d->out; p.ldr = d->ldr; p.ldo = d->ldo;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.silu = d->act == GN_ACT_SILU ? 1 : 0;
  GN_REQUIRE(d->act == GN_ACT_NONE || d->act == GN_ACT_SILU, "gn_conv3x3_gn: act (applied after the affine, before the conv) must be NONE or SILU");
  const bool narrow = d->Cout % 128 != 0;
  GN_REQUIRE(!narrow || !d->residual, "gn_conv3x3_gn: the narrow (Cout <= 32) variant takes no residual");
  p.tiles_x = d->W / CG_TW; p.tiles_y = d->H / CG_TH; p.tiles_n = narrow ? 1 : d->Cout / 128;
  p.x_bytes = (unsigned)((uint64_t)d->B * d->H * d->W * d->Cin * 2); p.w_bytes = (unsigned)((uint64_t)d->Cout * 9 * d->Cin * 2);
  const long nblocks = (long)d->B * p.tiles_x * p.tiles_y * p.tiles_n;
  if (narrow) hipLaunchKernelGGL(conv3x3_gn_narrow_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, p);
  else hipLaunchKernelGGL(conv3x3_gn_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
