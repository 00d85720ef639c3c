// Fused chains of a BasicTransformerBlock's Linears for the 64x64-latent level of the SD-Turbo UNet / ControlNet (C = 320), gfx950.
//
// The short-K Linears of a transformer block (K = 320: to_out, attn2.to_q, attn2.to_out, ff, proj_out) are 60 % fixed cost as separate
// launches (per-launch fill / drain against 1.3 us per K tile: DESIGN.md section 3, profiles/r03_v5_gemm_ksweep_n320.txt) and each one
// re-reads and re-writes the [M, C] residual stream.  Here ONE workgroup owns 128 rows of the stream for a whole chain of GEMMs:
//
//   * the rows live in LDS as the MFMA A operand (ten [128 x 32] sub-tiles, 64-byte rows, XOR-swizzled: 80 KB); every GEMM of the chain
//     writes its f16 output back into that image, so the [M, 4C] GEGLU intermediate and the residual stream between the chain's Linears
//     never reach HBM -- the values are rounded to f16 exactly where the unfused launches round them;
//   * the WEIGHTS of the whole chain come from a "weight tape": one contiguous buffer packed at load time (packing.pack_tblock_tape) that
//     holds, in consumption order, the LDS image of every 20 KB weight slot ([320 rows x 32 k] for an N = C GEMM, two [128 x 32] pieces of
//     the GEGLU projection, ...).  The kernel streams it through a 3-slot LDS ring with `buffer_load ... lds` (two slots in flight, counted
//     vmcnt, one raw s_barrier per slot) and never drains it between GEMMs -- the next GEMM's weights do not depend on anything;
//   * feed-forward: per 64-column chunk of the 4C hidden dimension GEGLU(LN(x) W1^T) is computed into LDS (16 KB) and multiplied into the
//     [128 x 320] f32 output tile held in registers (80 VGPRs over 512 threads);
//   * LayerNorm is folded as in gn_gemm_desc.ln_c1 (rstd * (x W'^T - mean * c1) + c2); the row statistics are taken from the LDS image
//     after the producing GEMM wrote it.
//
// Kinds (include/genima_hip.h gn_tblock_desc):
//   GN_TBLOCK_MID : h1 = a Wo^T + bo + res1 -> out;  q = LN2(h1) Wq^T -> out2                     (attn1.to_out.0, norm2 + attn2.to_q)
//   GN_TBLOCK_TAIL: h2 = a Wo^T + bo + res1;  h3 = GEGLU(LN3(h2) W1^T + b1) W2^T + b2 + h2;  out = h3 Wp^T + bp + res2
//                                                                     (attn2.to_out.0, norm3 + ff.net.0.proj + ff.net.2, proj_out)
// Replaces the corresponding gn_gemm launches of graphs.emit_transformer (the transformer blocks inside `self.pipe(...)`,
// controller/agent/sd_controlnet_agent.py:67-76).  Summation order per output element is that of the unfused launches (K walked in
// 16-wide steps, ascending), so the results agree with them to the rounding of the LayerNorm statistics.
#include "gemm_common.h"

namespace {

constexpr int TB_C = 320;                                  // channels of the level this kernel is built for
constexpr int TB_BM = 128;                                 // rows of the residual stream per workgroup
constexpr int TB_NT = 512;                                 // 8 waves: 4 row bands x 2 column halves
constexpr int TB_KT = TB_C / 32;                           // sub-tiles [BM x 32] of the A image
constexpr int TB_SUB = TB_BM * 64;                         // bytes of one sub-tile (64-byte rows)
constexpr int TB_SLOT = 20480;                             // bytes of one weight slot
constexpr int TB_RING_OFF = TB_KT * TB_SUB;                // 81920
constexpr int TB_H_OFF = TB_RING_OFF + 3 * TB_SLOT;        // 143360: GEGLU chunk [BM x 64] f16 as two sub-tiles
constexpr int TB_STATS_OFF = TB_H_OFF + 2 * TB_SUB;        // 159744: per row (-rstd * mean, rstd)
constexpr int TB_VEC_OFF = TB_STATS_OFF + TB_BM * 8;       // 160768: bias / c1 / c2 vectors of the N = C GEMMs
constexpr int TB_VEC_BYTES = 3072;
constexpr int TB_LDS = TB_VEC_OFF + TB_VEC_BYTES;          // 163840 = the CU's 160 KB
constexpr int TB_FF_CHUNKS = 4 * TB_C / 64;                // 20 chunks of 64 hidden columns
constexpr int TB_FRONT_VEC_BYTES = 7168;                   // FRONT's vector block (in the H region): b_in f16 [320] | c1 f32 [960] | c2 f16 [960], padded to 7 KB;
constexpr int TB_FRONT_SCSH_OFF = TB_FRONT_VEC_BYTES;      // behind it the sample's GroupNorm (scale, shift) pairs f32 [320][2]
constexpr int TB_C1C2_OFF = 2 * TB_SUB;                    // inside the LAST GEGLU-projection slot of a chunk: c1 f32 [128], then c2 f16 [128]
static_assert(TB_LDS <= 160 * 1024, "LDS budget");

struct TbParams {
  const f16* a;
  const f16* res1;
  const f16* res2;
  f16* out;
  f16* out2;
  const unsigned char* tape;
  const float* scsh;   // FRONT: GroupNorm (scale, shift) pairs [B][C][2]
  f16* out3;           // FRONT: V^T [B][C][ldo3]
  long lda, ldr1, ldr2, ldo, ldo2, ldo3;
  unsigned a_bytes, tape_bytes;
  int M, nslots, rpb;  // rpb: rows (tokens) per sample
  float eps;
};

__device__ __forceinline__ int swz64(int row, int chunk) { return lds_swz<64>(row, chunk); }

// LDS accesses inside the ring's cadence use ext_vector types ONLY: hipcc puts a `s_waitcnt vmcnt(0)` in front of every LDS access that
// carries no TBAA tag once an LDS-DMA is in flight (it cannot tell the access from the DMA's destination), and the HIP struct vectors
// (float2 / float4 / uint4) lose theirs in SROA -- a drained ring per GEGLU chunk in the first build of this kernel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 f32 of one 32x32 accumulator tile (lane: row l31, columns 8g + 4hi + e) -> f16 -> lanes l / l + 32 trade halves so that each owns
// 8 consecutive columns (8g + 8hi ..): two 16-byte pieces per lane, for chunks (0 + hi) and (2 + hi) of the 32-column tile
__device__ __forceinline__ void pack_tile(const float (&v)[16], u32x4 (&q)[2]) {
#pragma unroll
  for (int g = 0; g < 4; g += 2) {
    f16x4 ha, hb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ha[e] = (f16)v[4 * g + e]; hb[e] = (f16)v[4 * g + 4 + e]; }
    const uint2 ua = *reinterpret_cast<const uint2*>(&ha), ub = *reinterpret_cast<const uint2*>(&hb);
    const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
    q[g >> 1] = u32x4{r0[0], r1[0], r0[1], r1[1]};
  }
}

template <int KIND>
__global__ __launch_bounds__(TB_NT, 2) void tblock_kernel(const TbParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TB_LDS];  // ONE LDS object (a second one makes hipcc drain vmcnt around the DMAs)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * TB_BM;
  const int arow = wm * 32 + l31;  // this lane's row of the workgroup's 128 (A operand and accumulator rows alike)
  constexpr int VEC = KIND == GN_TBLOCK_FRONT ? TB_H_OFF : TB_VEC_OFF;  // the chain's bias / c1 / c2 vectors (FRONT has no GEGLU chunk: it takes that region)
  constexpr int VEC_BYTES = KIND == GN_TBLOCK_FRONT ? TB_FRONT_VEC_BYTES : TB_VEC_BYTES;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.tape, 0, (int)p.tape_bytes, 0x00020000);

  // ---- prologue: the workgroup's rows of `a` -> the A image (wave w: rows 16w .. 16w + 15, all ten sub-tiles; the XOR swizzle is applied
  // on the source side: lane q lands at physical chunk q & 3 of row q >> 2 and therefore fetches logical chunk (q & 3) ^ ((row >> 2) & 3))
  {
    const int row = 16 * wave + (lane >> 2);
    const int lc = (lane & 3) ^ ((row >> 2) & 3);
    const unsigned base = (unsigned)(((long)(m0 + row) * p.lda + lc * 8) * 2);
#pragma unroll
    for (int kt = 0; kt < TB_KT; ++kt)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + kt * TB_SUB + wave * 1024), 16, base + kt * 64, 0, 0, 0);
  }
  // the chain's bias / c1 / c2 vectors sit behind the last slot of the tape
  for (int i = wave; i < VEC_BYTES / 1024; i += 8)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)(smem + VEC + i * 1024), 16, (unsigned)p.nslots * TB_SLOT + i * 1024 + lane * 16, 0, 0, 0);
  if constexpr (KIND == GN_TBLOCK_FRONT) {
    // the sample's GroupNorm (scale, shift) pairs behind the vectors: 320 x 8 bytes (all rows of a workgroup belong to one sample)
    if (tid < TB_C / 2)
      *reinterpret_cast<f32x4*>(smem + VEC + TB_FRONT_SCSH_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(p.scsh + ((long)(m0 / p.rpb) * TB_C + 2 * tid) * 2);
  }

  // ---- the weight ring: slot s of the tape -> ring buffer s % 3; every wave moves two or three of its twenty 1 KB pieces
  // (the workgroups of an XCD walk the tape in step: each starts a slot at a different one of its 20 pieces, so that they do not all ask the
  // same L2 channel for the same line at the same moment)
  const int rot = __builtin_amdgcn_readfirstlane(((int)blockIdx.x >> 3) % 20);
  const int pc0 = (wave + rot) % 20, pc1 = (wave + 8 + rot) % 20, pc2 = (wave + 16 + rot) % 20;
  int n_issued = 0, ibuf = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    // past the last slot: offsets out of range = zero fill, no fetch -- every wave keeps issuing the same number of VMEM instructions
    const unsigned off = (unsigned)n_issued * TB_SLOT + lane * 16 + (n_issued < p.nslots ? 0u : 0x80000000u);
    unsigned char* dst = smem + TB_RING_OFF + ibuf * TB_SLOT;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)(dst + pc0 * 1024), 16, off + pc0 * 1024, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)(dst + pc1 * 1024), 16, off + pc1 * 1024, 0, 0, 0);
    if (wave < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lds_ptr_t)(dst + pc2 * 1024), 16, off + pc2 * 1024, 0, 0, 0);
    ++n_issued;
    ibuf = ibuf == 2 ? 0 : ibuf + 1;
  };
  int cbuf = 0;
  int stores_in_flight = 0;  // consumes for which an epilogue's 10 global stores per wave may still be outstanding (see consume)
  // consume one slot: its pieces have landed (the younger slot's stay in flight), every wave is past the previous slot (its buffer is free
  // for slot + 2) and past whatever it wrote to LDS before this call
  auto consume = [&](auto&& f) __attribute__((always_inline)) {
    // gfx9's single vmcnt retires loads and stores in issue order.  The ten global stores of an epilogue (`stores_in_flight` consumes ago)
    // are YOUNGER than this slot's pieces for two consumes: allowing them to stay outstanding lets them drain under the next GEMM's first
    // slots instead of stalling the ring behind the HBM write latency.
    if (stores_in_flight > 0) {
      if (wave < 4) asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      --stores_in_flight;
    } else {
      if (wave < 4) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue();
    f(smem + TB_RING_OFF + cbuf * TB_SLOT);
    cbuf = cbuf == 2 ? 0 : cbuf + 1;
  };
  auto block_barrier = [&]() __attribute__((always_inline)) {  // LDS hand-over between the waves outside the ring's cadence
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- MFMA pieces.  Operands swapped as in gemm.hip (D = W_tile A_tile^T): the lane holds D[n = 8g + 4hi + (r & 3)][m = l31].
  // one [320 x 32] weight slot against one [128 x 32] sub-tile of the A image: this wave's 32 rows x 160 columns
  auto mma_nc = [&](f32x16 (&acc)[5], const unsigned char* As, const unsigned char* Ws) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 2 + hi;
      const f16x8 fa = *reinterpret_cast<const f16x8*>(As + swz64(arow, c));
      f16x8 fw[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) fw[j] = *reinterpret_cast<const f16x8*>(Ws + swz64(wn * 160 + j * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fa, acc[j], 0, 0, 0);
    }
  };
  auto zero5 = [&](f32x16 (&acc)[5]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  };
  // residual rows of a global [M, ld] tensor for this wave's five tiles, requested a whole GEMM ahead of their use: RAW 16-byte pieces
  // (lane l: columns 8g .. 8g + 7, lane l + 32: the next eight, g = 0, 2) -- the lane swap that turns them into accumulator order would
  // wait for the data, so it happens in the epilogue (res_groups)
  auto prefetch_res = [&](const f16* res, long ld, uint4 (&r)[5][2]) __attribute__((always_inline)) {
    const f16* row = res + (long)(m0 + arow) * ld + wn * 160 + 8 * hi;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      r[j][0] = *reinterpret_cast<const uint4*>(row + j * 32);
      r[j][1] = *reinterpret_cast<const uint4*>(row + j * 32 + 16);
    }
  };
  auto res_groups = [&](const uint4 (&r)[2], f16x4 (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const auto r0 = __builtin_amdgcn_permlane32_swap(r[h].x, r[h].z, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(r[h].y, r[h].w, false, false);
      const uint2 lo = make_uint2(r0[0], r1[0]), hi2 = make_uint2(r0[1], r1[1]);
      a[2 * h] = *reinterpret_cast<const f16x4*>(&lo);
      a[2 * h + 1] = *reinterpret_cast<const f16x4*>(&hi2);
    }
  };

  // epilogue of an N = C GEMM: (LayerNorm fold) + bias + residual -> f16 -> the A image (in place) and / or global memory.
  //   bias_off: byte offset of the f16 bias (or c2) vector in the VEC region; c1_off: f32 c1 vector (LNF); RES: 0 none, 1 registers, 2 the A image
  auto epilogue_nc = [&](f32x16 (&acc)[5], int bias_off, auto lnf_c, int c1_off, auto res_c, const uint4 (&rpre)[5][2], auto lds_c, f16* gout,
                         long ldo, auto vt_c) __attribute__((always_inline)) {
    constexpr bool TO_VT = decltype(vt_c)::value;  // f16 values -> [n][m] in the (free) A image, for the transposed V^T store
    constexpr bool LNF = decltype(lnf_c)::value;
    constexpr int RES = decltype(res_c)::value;
    constexpr bool TO_LDS = decltype(lds_c)::value;
    float nrm = 0.0f, rstd = 1.0f;
    if constexpr (LNF) {
      const f32x2 st = *reinterpret_cast<const f32x2*>(smem + TB_STATS_OFF + arow * 8);
      nrm = st[0]; rstd = st[1];
    }
    static_for<5>::run([&](auto J) {
      constexpr int j = decltype(J)::value;
      const int n0 = wn * 160 + j * 32;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[j][e];
      if constexpr (LNF) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 c1 = *reinterpret_cast<const f32x4*>(smem + VEC + c1_off + (n0 + 8 * g + 4 * hi) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * g + e] = rstd * v[4 * g + e] + nrm * c1[e];
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f16x4 b = *reinterpret_cast<const f16x4*>(smem + VEC + bias_off + (n0 + 8 * g + 4 * hi) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] += (float)b[e];
      }
      if constexpr (RES == 1) {
        f16x4 rg[4];
        res_groups(rpre[j], rg);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += (float)rg[e >> 2][e & 3];
      } else if constexpr (RES == 2) {
        const unsigned char* At = smem + (wn * 5 + j) * TB_SUB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f16x4 r = *reinterpret_cast<const f16x4*>(At + swz64(arow, g) + hi * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * g + e] += (float)r[e];
        }
      }
      if constexpr (TO_VT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) *reinterpret_cast<f16*>(smem + ((n0 + 8 * (e >> 2) + 4 * hi + (e & 3)) * TB_BM + arow) * 2) = (f16)v[e];
        return;
      }
      u32x4 q[2];
      pack_tile(v, q);
      if constexpr (TO_LDS) {
        unsigned char* At = smem + (wn * 5 + j) * TB_SUB;
        *reinterpret_cast<u32x4*>(At + swz64(arow, 0 + hi)) = q[0];
        *reinterpret_cast<u32x4*>(At + swz64(arow, 2 + hi)) = q[1];
      }
      if (gout) {
        f16* orow = gout + (long)(m0 + arow) * ldo + n0 + 8 * hi;
        *reinterpret_cast<u32x4*>(orow) = q[0];
        *reinterpret_cast<u32x4*>(orow + 16) = q[1];
      }
    });
  };
  // LayerNorm statistics of the rows in the A image: four threads per row (sub-tiles part, part + 4, part + 8), f32 sums of the f16 values
  auto row_stats = [&]() __attribute__((always_inline)) {
    const int row = tid >> 2, part = tid & 3;
    const f16x2 ones = {(f16)1.0f, (f16)1.0f};
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int kt = part + 4 * i;
      if (kt < TB_KT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f16x8 x = *reinterpret_cast<const f16x8*>(smem + kt * TB_SUB + swz64(row, c));
          const f16x2 h0 = __builtin_shufflevector(x, x, 0, 1), h1 = __builtin_shufflevector(x, x, 2, 3);
          const f16x2 h2 = __builtin_shufflevector(x, x, 4, 5), h3 = __builtin_shufflevector(x, x, 6, 7);
          s1 = __builtin_amdgcn_fdot2(h0, ones, s1, false); s2 = __builtin_amdgcn_fdot2(h0, h0, s2, false);
          s1 = __builtin_amdgcn_fdot2(h1, ones, s1, false); s2 = __builtin_amdgcn_fdot2(h1, h1, s2, false);
          s1 = __builtin_amdgcn_fdot2(h2, ones, s1, false); s2 = __builtin_amdgcn_fdot2(h2, h2, s2, false);
          s1 = __builtin_amdgcn_fdot2(h3, ones, s1, false); s2 = __builtin_amdgcn_fdot2(h3, h3, s2, false);
        }
      }
    }
    s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);  // the four threads of a row are one DPP quad
    s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);
    if (part == 0) {
      const float mean = s1 * (1.0f / TB_C);
      const float var = fmaxf(s2 * (1.0f / TB_C) - mean * mean, 0.0f);
      const float rstd = __frsqrt_rn(var + p.eps);
      *reinterpret_cast<f32x2*>(smem + TB_STATS_OFF + row * 8) = f32x2{-rstd * mean, rstd};
    }
  };

  // ---- the chain -------------------------------------------------------------------------------------------------------------------
  uint4 rpre[5][2];
  if constexpr (KIND != GN_TBLOCK_FRONT) prefetch_res(p.res1, p.ldr1, rpre);  // older than every ring DMA: retired by the first counted wait
  issue();
  issue();
  if constexpr (KIND == GN_TBLOCK_FRONT) {
    // GroupNorm-apply on the rows in place (no activation: Transformer2DModel.norm): thread (row, quarter) walks its chunks of the ten sub-tiles
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int row = tid >> 2, lc = tid & 3;
#pragma unroll 2
    for (int kt = 0; kt < TB_KT; ++kt) {
      unsigned char* q = smem + kt * TB_SUB + swz64(row, lc);
      const f16x8 v = *reinterpret_cast<const f16x8*>(q);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(smem + VEC + TB_FRONT_SCSH_OFF + (kt * 32 + lc * 8 + e) * 8);
        o[e] = (f16)((float)v[e] * sc[0] + sc[1]);
        o[e + 1] = (f16)((float)v[e + 1] * sc[2] + sc[3]);
      }
      *reinterpret_cast<f16x8*>(q) = o;
    }
  }

  f32x16 acc[5];
  zero5(acc);
  // GEMM 1: a Wo^T + bo (+ res1) -> the A image (in place)
  for (int kt = 0; kt < TB_KT; ++kt)
    consume([&](const unsigned char* Ws) __attribute__((always_inline)) { mma_nc(acc, smem + kt * TB_SUB, Ws); });
  block_barrier();  // every wave is done reading `a` out of the image
  if constexpr (KIND == GN_TBLOCK_FRONT) {
    epilogue_nc(acc, 0, std::false_type{}, 0, std::integral_constant<int, 0>{}, rpre, std::true_type{}, p.out, p.ldo, std::false_type{});
    stores_in_flight = 2;
  } else if constexpr (KIND == GN_TBLOCK_MID) {
    epilogue_nc(acc, 0, std::false_type{}, 0, std::integral_constant<int, 1>{}, rpre, std::true_type{}, p.out, p.ldo, std::false_type{});
    stores_in_flight = 2;
  }
  else
    epilogue_nc(acc, 0, std::false_type{}, 0, std::integral_constant<int, 1>{}, rpre, std::true_type{}, (f16*)nullptr, 0, std::false_type{});
  block_barrier();
  row_stats();  // (visible to the epilogues that read them: at least one slot barrier lies between)

  if constexpr (KIND == GN_TBLOCK_FRONT) {
    // q, k, v = LN1(h) W'^T with the same raw rows (VEC: b_in f16 [320] | c1 f32 [960] | c2 f16 [960]); q | k row-major into out2 [M, 2C],
    // V transposed per sample (the attention kernels' V^T operand) through the A image once every wave is done with it
    for (int g3 = 0; g3 < 3; ++g3) {
      zero5(acc);
      for (int kt = 0; kt < TB_KT; ++kt)
        consume([&](const unsigned char* Ws) __attribute__((always_inline)) { mma_nc(acc, smem + kt * TB_SUB, Ws); });
      if (g3 < 2) {
        epilogue_nc(acc, 4480 + g3 * 640, std::true_type{}, 640 + g3 * 1280, std::integral_constant<int, 0>{}, rpre, std::false_type{},
                    p.out2 + g3 * TB_C, p.ldo2, std::false_type{});
        stores_in_flight = 2;
      } else {
        block_barrier();
        epilogue_nc(acc, 4480 + 2 * 640, std::true_type{}, 640 + 2 * 1280, std::integral_constant<int, 0>{}, rpre, std::false_type{}, (f16*)nullptr, 0,
                    std::true_type{});
        block_barrier();
        // [320 n][128 m] f16 in LDS -> V^T[b][n][m_local ..]: 16 lanes cover the 256 contiguous bytes of one n
        const int b = m0 / p.rpb, ml = m0 - b * p.rpb;
        f16* vt = p.out3 + (long)b * TB_C * p.ldo3 + ml;
#pragma unroll
        for (int i = 0; i < TB_C * 16 / TB_NT; ++i) {
          const int c = tid + TB_NT * i, n = c >> 4, pc = c & 15;
          *reinterpret_cast<u32x4*>(vt + (long)n * p.ldo3 + pc * 8) = *reinterpret_cast<const u32x4*>(smem + n * (TB_BM * 2) + pc * 16);
        }
      }
    }
  }
  if constexpr (KIND == GN_TBLOCK_FRONT) {
  } else if constexpr (KIND == GN_TBLOCK_MID) {
    // GEMM 2: LN2(h1) Wq^T -> out2 (VEC: bo f16 [320] | c1 f32 [320] | c2 f16 [320])
    zero5(acc);
    for (int kt = 0; kt < TB_KT; ++kt)
      consume([&](const unsigned char* Ws) __attribute__((always_inline)) { mma_nc(acc, smem + kt * TB_SUB, Ws); });
    epilogue_nc(acc, 640 + 1280, std::true_type{}, 640, std::integral_constant<int, 0>{}, rpre, std::false_type{}, p.out2, p.ldo2, std::false_type{});
  } else {
    // feed-forward: per 64-column chunk of the hidden dimension, hid = GEGLU(LN3(h2) W1'^T) -> LDS, acc += hid W2[:, chunk]^T
    zero5(acc);
    for (int ch = 0; ch < TB_FF_CHUNKS; ++ch) {
      f32x16 hacc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[0][r] = hacc[1][r] = 0.0f;
      for (int js = 0; js < 5; ++js) {
        consume([&](const unsigned char* Ws) __attribute__((always_inline)) {
          // slot = two [128 x 32] pieces of the chunk's 128 packed rows [hidden 0..31 | gate 0..31 | hidden 32..63 | gate 32..63]
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const unsigned char* As = smem + (2 * js + sub) * TB_SUB;
            const unsigned char* Wp = Ws + sub * TB_SUB;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              const int c = kk * 2 + hi;
              const f16x8 fa = *reinterpret_cast<const f16x8*>(As + swz64(arow, c));
              const f16x8 fh = *reinterpret_cast<const f16x8*>(Wp + swz64(wn * 64 + l31, c));
              const f16x8 fg = *reinterpret_cast<const f16x8*>(Wp + swz64(wn * 64 + 32 + l31, c));
              hacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fa, hacc[0], 0, 0, 0);
              hacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fg, fa, hacc[1], 0, 0, 0);
            }
          }
          if (js == 4) {
            // GEGLU epilogue; the chunk's c1 / c2 ride in this slot.  hid -> the H image (sub-tile wn: hidden columns 32 wn .. + 31)
            const f32x2 st = *reinterpret_cast<const f32x2*>(smem + TB_STATS_OFF + arow * 8);
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nh = wn * 64 + 8 * g + 4 * hi;  // packed row of the hidden value; its gate is 32 further
              const f32x4 c1h = *reinterpret_cast<const f32x4*>(Ws + TB_C1C2_OFF + nh * 4);
              const f32x4 c1g = *reinterpret_cast<const f32x4*>(Ws + TB_C1C2_OFF + (nh + 32) * 4);
              const f16x4 c2h = *reinterpret_cast<const f16x4*>(Ws + TB_C1C2_OFF + 512 + nh * 2);
              const f16x4 c2g = *reinterpret_cast<const f16x4*>(Ws + TB_C1C2_OFF + 512 + (nh + 32) * 2);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float hv = (st[1] * hacc[0][4 * g + e] + st[0] * c1h[e]) + (float)c2h[e];
                const float gv = (st[1] * hacc[1][4 * g + e] + st[0] * c1g[e]) + (float)c2g[e];
                v[4 * g + e] = hv * gelu_fast(gv);
              }
            }
            u32x4 q[2];
            pack_tile(v, q);
            unsigned char* Ht = smem + TB_H_OFF + wn * TB_SUB;
            *reinterpret_cast<u32x4*>(Ht + swz64(arow, 0 + hi)) = q[0];
            *reinterpret_cast<u32x4*>(Ht + swz64(arow, 2 + hi)) = q[1];
          }
        });
      }
      for (int j2 = 0; j2 < 2; ++j2)
        consume([&](const unsigned char* Ws) __attribute__((always_inline)) { mma_nc(acc, smem + TB_H_OFF + j2 * TB_SUB, Ws); });
    }
    // + b2 + h2 (read back from the image) -> h3, in place (no other wave reads the image any more: the last A-operand reads were five
    // slot barriers ago, and the waves' residual tiles are disjoint)
    epilogue_nc(acc, 640, std::false_type{}, 0, std::integral_constant<int, 2>{}, rpre, std::true_type{}, (f16*)nullptr, 0, std::false_type{});
    // proj_out: h3 Wp^T + bp + res2 -> out
    prefetch_res(p.res2, p.ldr2, rpre);
    zero5(acc);
    for (int kt = 0; kt < TB_KT; ++kt)
      consume([&](const unsigned char* Ws) __attribute__((always_inline)) { mma_nc(acc, smem + kt * TB_SUB, Ws); });
    epilogue_nc(acc, 1280, std::false_type{}, 0, std::integral_constant<int, 1>{}, rpre, std::false_type{}, p.out, p.ldo, std::false_type{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead zero fills must not outlive the workgroup's LDS allocation
}

}  // namespace

extern "C" int64_t gn_tblock_tape_bytes(int32_t kind, int32_t C) {
  if (C != TB_C) return 0;
  const int64_t nslots = kind == GN_TBLOCK_TAIL ? 2 * TB_KT + TB_FF_CHUNKS * 7 : (kind == GN_TBLOCK_MID ? 2 * TB_KT : (kind == GN_TBLOCK_FRONT ? 4 * TB_KT : 0));
  return nslots ? nslots * TB_SLOT + (kind == GN_TBLOCK_FRONT ? TB_FRONT_VEC_BYTES : TB_VEC_BYTES) : 0;
}

extern "C" int32_t gn_tblock_supported(int32_t kind, int64_t M, int32_t C) {
  return (kind == GN_TBLOCK_FRONT || kind == GN_TBLOCK_MID || kind == GN_TBLOCK_TAIL) && C == TB_C && M > 0 && M % TB_BM == 0 ? 1 : 0;
}

int32_t gn_launch_tblock(gn_ctx* ctx, const gn_tblock_desc* d) {
  GN_REQUIRE(d && d->a && d->out && d->tape, "gn_tblock: null a / out / tape");
  GN_REQUIRE(d->kind == GN_TBLOCK_FRONT || d->kind == GN_TBLOCK_MID || d->kind == GN_TBLOCK_TAIL, "gn_tblock: unknown kind %d", d->kind);
  GN_REQUIRE(d->kind == GN_TBLOCK_FRONT || d->res1, "gn_tblock: res1 is required for the MID / TAIL chains");
  GN_REQUIRE(d->C == TB_C, "gn_tblock: built for C = %d (got %d); use the gn_gemm launches for other widths", TB_C, d->C);
  GN_REQUIRE(d->M > 0 && d->M % TB_BM == 0, "gn_tblock: M (%ld) must be a positive multiple of %d", (long)d->M, TB_BM);
  GN_REQUIRE(d->tape_bytes == gn_tblock_tape_bytes(d->kind, d->C), "gn_tblock: tape_bytes %ld != gn_tblock_tape_bytes() = %ld", (long)d->tape_bytes,
             (long)gn_tblock_tape_bytes(d->kind, d->C));
  GN_REQUIRE(d->lda >= TB_C && d->lda % 8 == 0 && d->ldr1 % 8 == 0 && d->ldo % 8 == 0, "gn_tblock: row strides must be multiples of 8 elements");
  GN_REQUIRE(((uintptr_t)d->a & 15) == 0 && ((uintptr_t)d->res1 & 15) == 0 && ((uintptr_t)d->out & 15) == 0 && ((uintptr_t)d->tape & 15) == 0,
             "gn_tblock: a / res1 / out / tape must be 16-byte aligned");
  GN_REQUIRE((uint64_t)d->M * d->lda * 2 < 0xFFFFFF00ull, "gn_tblock: `a` too large for 32-bit buffer offsets");
  GN_REQUIRE(d->ln_eps > 0.0f, "gn_tblock: ln_eps must be positive");
  TbParams p;
  p.a = (const f16*)d->a; p.res1 = (const f16*)d->res1; p.res2 = (const f16*)d->res2;
  p.out = (f16*)d->out; p.out2 = (f16*)d->out2; p.tape = (const unsigned char*)d->tape;
  p.scsh = (const float*)d->scsh; p.out3 = (f16*)d->out3; p.ldo3 = d->ldo3; p.rpb = d->rows_per_batch;
  p.lda = d->lda; p.ldr1 = d->ldr1; p.ldr2 = d->ldr2; p.ldo = d->ldo; p.ldo2 = d->ldo2;
  p.a_bytes = (unsigned)((uint64_t)d->M * d->lda * 2); p.tape_bytes = (unsigned)d->tape_bytes;
  p.M = (int)d->M; p.eps = d->ln_eps;
  const dim3 grid((unsigned)(d->M / TB_BM)), block(TB_NT);
  if (d->kind == GN_TBLOCK_FRONT) {
    GN_REQUIRE(d->scsh && d->out2 && d->out3 && ((uintptr_t)d->scsh & 15) == 0 && ((uintptr_t)d->out2 & 15) == 0 && ((uintptr_t)d->out3 & 15) == 0,
               "gn_tblock(front): scsh, out2 (q | k) and out3 (V^T) must be given, 16-byte aligned");
    GN_REQUIRE(d->ldo2 % 8 == 0 && d->ldo2 >= 2 * TB_C && d->ldo3 % 8 == 0, "gn_tblock(front): ldo2 >= 2 C, strides multiples of 8");
    GN_REQUIRE(d->rows_per_batch > 0 && d->rows_per_batch % TB_BM == 0 && d->M % d->rows_per_batch == 0 && d->ldo3 >= d->rows_per_batch,
               "gn_tblock(front): rows_per_batch (%d) must be a multiple of %d dividing M, ldo3 >= rows_per_batch", d->rows_per_batch, TB_BM);
    p.nslots = 4 * TB_KT;
    hipLaunchKernelGGL((tblock_kernel<GN_TBLOCK_FRONT>), grid, block, 0, ctx->stream, p);
  } else if (d->kind == GN_TBLOCK_MID) {
    GN_REQUIRE(d->out2 && d->ldo2 % 8 == 0 && ((uintptr_t)d->out2 & 15) == 0, "gn_tblock(mid): out2 (q) must be given, 16-byte aligned, ldo2 %% 8 == 0");
    p.nslots = 2 * TB_KT;
    hipLaunchKernelGGL((tblock_kernel<GN_TBLOCK_MID>), grid, block, 0, ctx->stream, p);
  } else {
    GN_REQUIRE(d->res2 && d->ldr2 % 8 == 0 && ((uintptr_t)d->res2 & 15) == 0, "gn_tblock(tail): res2 (the block input) must be given, 16-byte aligned");
    p.nslots = 2 * TB_KT + TB_FF_CHUNKS * 7;
    hipLaunchKernelGGL((tblock_kernel<GN_TBLOCK_TAIL>), grid, block, 0, ctx->stream, p);
  }
  GN_LAUNCH_CHECK();
  return GN_OK;
}
