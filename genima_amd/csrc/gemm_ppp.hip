// Persistent, SKEWED ping-pong implicit GEMM for gfx950 (tile 25): the 256 x 256 x 64 ping-pong K loop of gemm_pp.hip inside a workgroup that
// stays on its CU and walks a list of tiles, so that a tile's fill and drain stop being a chip-wide event.
//
// What the one-launch-round form costs (profiles/r05_v15_pp_ksweep.txt, r05_v16_pp_epilogue_ablation.txt): a round of 256 one-per-CU tiles spends
// 14.3 us beside its K loop -- ~2 us until the first K tile has landed, ~11 us in an epilogue in which all 256 workgroups store 33.5 MB at the same
// moment -- and a grid of 2.5 rounds pays three of them.  Here:
//   * grid = one workgroup per CU (ppG); workgroup c runs the tiles c, c + G, c + 2 G ... (XCD-aware order inside a round, as gemm_pp.hip).
//   * tile i + 1's first 1.75 K tiles (the whole LDS ring) are requested BEFORE tile i's epilogue: the epilogue's vector loads, its arithmetic
//     and the ring fill share one memory round trip.  Then `s_waitcnt vmcnt(0)`, then ALL of the tile's stores back to back, then the next K loop
//     starts at once: its first K iteration reads only what the prologue brought and carries no wait at all, so the stores have a whole K
//     iteration (~1.8 us) to be acknowledged before the first counted wait that (gfx9's vmcnt retires in order) would have to sit them out.
//   * SKEW: workgroup c enters its first tile at K iteration k_c = c * nk / G and hands the partial sums to workgroup c - 1, which computes the K
//     prefix [0, k_c) of that tile as the LAST thing it does and writes the tile.  Every workgroup does the same amount of work, their tile
//     boundaries are spread evenly over a tile period, and the chip never drains in one write burst (the stagger experiment of round 5 with the
//     offset made of useful work).  Hand-off = f32 slab (256 KB, register order: fully coalesced 16-byte write-through `sc1` stores), drained
//     vmcnt, one agent-scope flag; the consumer polls that one word, one agent acquire, plain loads (cdna_hip_programming.md Guideline 16, R1).
//     The producer publishes at the very start of the launch, the consumer needs it at the very end: nobody waits in practice.
//   * TAIL: the tiles of the last, partial round are split along K over floor(G / tail tiles) workgroups each through the same hand-off (the part
//     that holds k = 0 owns the tile and adds the others' slabs in a fixed order): 640 tiles cost 2.5 tile times instead of 3, 320 tiles 1.25 of 2.
//   Flags are self-cleaning (their one consumer resets them) inside a library-owned, zero-initialised pool: no memset node per launch.
//   Deadlock freedom without co-residency: a workgroup only ever waits for a segment that is the FIRST thing its producer does (skew) or that the
//   producer reaches without waiting on anything later-dispatched than itself (tail); waits are bounded and counted (gn_ppp_timeouts).
// Results: a tile whose K range is not shared is bit-identical to every other tile configuration (K walked alike); a shared tile adds its f32 partial
// sums in a fixed order (prefix + suffix; part 0 + part 1 + ...): deterministic run to run, a K split's rounding against the unsplit sum.
// Restrictions on top of gemm_pp.hip's (the planner falls back to tile 15): row-major f16 output through 16-byte stores (N, ldo % 8 == 0), bias /
// shift OR residual / activation / scale epilogues, >= one tile per CU; no split-K, out2, LayerNorm fold, GEGLU, GroupNorm bridge.
#include <atomic>

#include "gemm_common.h"

namespace {

constexpr int PP_STAGE = 65536;  // bytes per LDS stage: A tile 256 x 128 B, then W tile 256 x 128 B
constexpr int PP_HALF = 16384;   // one half-tile (128 rows)
constexpr int PPP_SLAB = 256 * 256;  // floats of one hand-off slab

constexpr int PPP_POOL_HEAD = 64;      // words in front of the regions; word 0 counts bounded waits that gave up
constexpr int PPP_REGION = 512;        // flag words of one launch: [0, 256) the skew hand-offs, [256, 512) the tail's
constexpr int PPP_REGIONS = 1024;
constexpr int PPP_PROF_WORDS = 256 * 8;  // behind the regions: per-workgroup cycle sums of a profiling build (-DGN_PPP_PROFILE, tools/probes/ppp_profile.py)

enum { ROLE_FULL = 0, ROLE_PRODUCER = 1, ROLE_OWNER = 2 };
struct Seg { int tile, k0, k1, role, slot, nslot; };

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The kernel's parameter block read through a laundered kernarg pointer: loads the compiler cannot hoist.  Everything the segment BOUNDARIES need
// (epilogue pointers and strides, the tile list's constants) is read there, where it is used, instead of sitting in SGPRs across the K loop -- the
// ping-pong loop runs at 96 - 104 SGPRs and 243 - 249 VGPRs on its own (gemm_pp.hip), and a spilled SGPR costs VGPR lanes, a spilled VGPR a scratch
// access that the loop's counted vmcnt would have to count.
// `s_waitcnt vmcnt(0)` the compiler can SEE (expcnt / lgkmcnt fields at their maxima = not waited for): its scoreboard then knows that every LDS-DMA
// piece issued so far has landed and it puts no wait of its own in front of the next K loop's first fragment reads -- an asm wait is invisible to it,
// and the waits it then adds (vmcnt(12) .. (5) in front of the reads of K iteration 0) sit out the tile's stores, which are younger than the ring
#define PPP_DRAIN()                              \
  do {                                           \
    __builtin_amdgcn_s_waitcnt(0x0F70);          \
    asm volatile("" ::: "memory");               \
  } while (0)
#ifdef GN_PPP_PROFILE  // wave 0's cycle sums per section (s_memtime), written behind the flag regions at the end of the launch
#define PPP_T(var) const unsigned var = (unsigned)__builtin_readcyclecounter()
#define PPP_ACC(i, a, b) prof[i] += (b) - (a)
#else
#define PPP_T(var) do {} while (0)
#define PPP_ACC(i, a, b) do {} while (0)
#endif
typedef const GemmParams __attribute__((address_space(4))) KArgs;
__device__ __forceinline__ KArgs* ppp_kargs() {
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return (KArgs*)kp;
}

// n / d by the host's magic pair: three scalar instructions instead of the ~30 (32-bit) or ~300 (64-bit) of a division the compiler expands
__device__ __forceinline__ int fdiv(int n, unsigned mul, unsigned shift) {
  return (int)(((unsigned long long)__umulhi((unsigned)n, mul) + (unsigned)n) >> shift);
}
#define PPP_DIV(n, fd) fdiv((n), q->fd.mul, q->fd.shift)

// FF (dense only): the transformer's feed-forward projection -- LayerNorm folded into the Linear (gn_gemm_desc.ln_c1: the row statistics come from
// the A fragments of the K loop) with the GEGLU epilogue (W rows packed in 32-row [hidden | gate] blocks).  The loader permutes the W rows of a
// tile so that W half 0 holds its four hidden blocks and half 1 its four gate blocks: wave (., wc) then owns hidden block wc in the W0 quadrants
// and gate block wc in the W1 quadrants of the same rows, and the GEGLU product is register-local.
template <bool CONV, bool FF>
__global__ __launch_bounds__(512) void gemm_ppp_kernel(const GemmParams p) {
  static_assert(!(CONV && FF), "the feed-forward variant is a dense problem");
  constexpr int kStatsBytes = FF ? 4 * 256 * 8 : 0;  // FF: (sum, sum of squares) of the 256 rows from each of the 4 waves that share them
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PP_STAGE + kStatsBytes];
#ifdef GN_PPP_PROFILE
  unsigned prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PPP_T(t_start);
#endif

  // The ONLY per-lane value that lives across the whole kernel is the lane id; everything derived from it (fragment read offsets, the loader's row and
  // chunk, the epilogue's row / column) is recomputed from an opaque copy where it is used, so that it does not sit in registers -- or worse, in
  // scratch, whose reloads would queue behind a tile's stores -- across the epilogue, where the accumulators and the packed tile need the file.
  const int lane_ = threadIdx.x & 63;
  const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave_;
  const int wr = wave >> 2;  // wr = ping-pong group = which 64 rows of each A half (wc = wave & 3 = which 32 rows of each W half)

  // ---- the workgroup's segment list (scalar arithmetic, recomputed at every segment boundary: nothing of it lives across a K loop) -----------------
  auto seg_count = [&]() __attribute__((always_inline)) {
    KArgs* q = ppp_kargs();
    int c = blockIdx.x;
    asm volatile("" : "+s"(c));
    const int G = q->ppG, nk = q->K / BK;
    const int kc1 = (q->ppSkew && c + 1 < G) ? PPP_DIV((c + 1) * nk, dG) : 0;
    return q->ppR + (kc1 > 0 ? 1 : 0) + ((q->ppTail > 0 && c < q->ppTail * q->ppS) ? 1 : 0);
  };
  auto get_seg = [&](int i) __attribute__((always_inline)) {
    KArgs* q = ppp_kargs();
    int c = blockIdx.x;
    asm volatile("" : "+s"(c));  // (opaque: the values derived from it are not hoisted out of the segment loop)
    const int G = q->ppG, nk = q->K / BK, R = q->ppR, S = q->ppS, skew = q->ppSkew;
    auto vid = [&](int b) __attribute__((always_inline)) {  // XCD-aware position of hardware workgroup b inside a round: an XCD's workgroups take a contiguous run of tiles
      const int qq = G >> 3, r = G & 7, xcd = b & 7, idx = b >> 3;
      return (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    };
    const int kc0 = skew ? PPP_DIV(c * nk, dG) : 0;  // (c * nk < 2^31: the planner keeps nk < 2^20)
    const int kc1 = (skew && c + 1 < G) ? PPP_DIV((c + 1) * nk, dG) : 0;
    Seg s;
    s.slot = 0; s.nslot = 0;
    if (i == 0) {  // round 0: the K suffix from this workgroup's skew offset
      s.tile = vid(c); s.k0 = kc0; s.k1 = nk;
      s.role = kc0 == 0 ? ROLE_FULL : ROLE_PRODUCER; s.slot = c;
    } else if (i < R) {
      s.tile = i * G + vid(c); s.k0 = 0; s.k1 = nk; s.role = ROLE_FULL;
    } else if (i == R && kc1 > 0) {  // the K prefix of the next workgroup's round-0 tile: this workgroup writes that tile
      s.tile = vid(c + 1); s.k0 = 0; s.k1 = kc1; s.role = ROLE_OWNER; s.slot = c + 1; s.nslot = 1;
    } else {  // a tile of the last partial round, split along K over ppS workgroups
      const int j = PPP_DIV(c, dS), part = c - j * S;
      s.tile = R * G + j;
      s.k0 = PPP_DIV(part * nk, dS); s.k1 = PPP_DIV((part + 1) * nk, dS);
      if (S == 1) s.role = ROLE_FULL;
      else if (part == 0) { s.role = ROLE_OWNER; s.slot = 256 + j * (S - 1); s.nslot = S - 1; }
      else { s.role = ROLE_PRODUCER; s.slot = 256 + j * (S - 1) + part - 1; }
    }
    return s;
  };
  // tile origin and (up_phases) the output phase of a tile index
  auto tile_origin = [&](int tile, int& tm0, int& tn0, int& z) __attribute__((always_inline)) {
    KArgs* q = ppp_kargs();
    const int tm = q->tiles_m, tn = q->tiles_n;
    int rem = tile;
    z = 0;
    if (q->ppNz > 1) { z = PPP_DIV(tile, dTmn); rem = tile - z * (tm * tn); }
    int tile_n, tile_m;
    if (q->cm_tiles) { tile_n = PPP_DIV(rem, dTm); tile_m = rem - tile_n * tm; }
    else { tile_m = PPP_DIV(rem, dTn); tile_n = rem - tile_m * tn; }
    tm0 = tile_m * 256; tn0 = tile_n * 256;
  };

  // ---- loader state ---------------------------------------------------------------------------------------------------------------------------
  // A lane stages the rows r0 + R of every half-tile, r0 = 8 * wave + lane / 8 in [0, 64) and R = 128 h + 64 i in {0, 64, 128, 192}: with M and N
  // multiples of 256 there is no row mask, the R part of an address is SCALAR (the buffer instruction's soffset) and one offset register per
  // operand serves all four pieces.  Conv: with (Ho * Wo) % 256 == 0 a tile lies inside one sample, and with Wo % 64 == 0 or 64 % Wo == 0 the pixel
  // of row r0 + R is (oy_s[R] + r0 / Wo, ox_s[R] + r0 % Wo) with scalar oy_s / ox_s: the lane keeps (r0 / Wo, r0 % Wo) x stride and the four pixel
  // indices of the current filter tap.
  const int Cin = p.C1;        // channels under each filter tap (one source: virtual concats and k_append stay on tile 15)
  const int Hin = p.ups ? 2 * p.H : p.H, Win = p.ups ? 2 * p.W : p.W;
  auto lane_kl2 = [&](int lane) __attribute__((always_inline)) { return (((lane & 7) ^ ((4 * wave_ + (lane >> 4)) & 7)) * 16); };  // the lane's K byte offset inside a tile
  int kl2 = 0;             // the lane's K byte offset inside a tile: set by setup() and again at the top of every segment's K loop (conv: stage_a adds it to the channel offset)
  int iyl = 0, ixl = 0;    // conv: (r0 / Wo, r0 % Wo) * stride, likewise
  int pix[2][2];           // conv, [half][piece]: source pixel of the current tap, -1 = outside the image
  int tap_s[2][2];         // conv (scalar): (oy_s * stride - pad_t) & 0xFFFF | (ox_s * stride - pad_l) << 16 of piece (h, i)
  int pbase_s = 0;         // conv (scalar): first pixel of the tile's sample
  unsigned aoffk = 0;      // dense: byte offset of (row m0 + r0, column kl) of A
  unsigned woffk = 0;      // byte offset of (row n0 + r0, column kl) of W
  int ccA[2], dyA[2], dxA[2];  // conv: channel / tap of the next K tile each A half-tile stream stages (wave-uniform)
  int kbeg = 0, kend = 0;      // the segment being computed

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);

  auto set_tap = [&](int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = (int)(short)(tap_s[h][i] & 0xFFFF) + iyl + dyA[h], ix = (tap_s[h][i] >> 16) + ixl + dxA[h];
      const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
      const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
      pix[h][i] = ok ? pbase_s + sy * p.W + sx : -1;
    }
  };

  // everything gemm_pp.hip computes once per workgroup, per segment: tile origin, the rows this lane stages, the K walk's start
  auto setup = [&](const Seg& s) __attribute__((always_inline)) {
    KArgs* q = ppp_kargs();
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));  // (opaque copies: what is derived from them is recomputed here, not carried across the K loop)
    const int r0 = 8 * wave + (lane >> 3);
    kl2 = lane_kl2(lane);  // (the first segment's ring is requested before the first lane_setup)
    int m0, n0, z;
    tile_origin(s.tile, m0, n0, z);
    kbeg = s.k0 * BK;
    kend = s.k1 * BK;
    // FF: LDS row 64 i + r0 of W half h is global row n0 + 128 i + 32 h + 64 (r0 / 32) + r0 % 32 (the hidden block of pair 2 i + r0 / 32 for h = 0, its gate block for h = 1)
    woffk = (unsigned)((long)(n0 + (FF ? 64 * (r0 >> 5) + (r0 & 31) : r0)) * q->ldw * 2) + (unsigned)kl2;
    if (q->up_ph)  // phase z = 2 dy + dx of an upsampling conv (gemm_common.h batch_offset): its weights here, its padding below, its output offset in the epilogue
      rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(q->w + (long)z * q->w_bs), 0, (int)q->w_bytes, 0x00020000);
    if constexpr (CONV) {
      int pad_t = q->pad_t, pad_l = q->pad_l;
      if (q->up_ph) { pad_t = 1 - (z >> 1); pad_l = 1 - (z & 1); }
      const int Wo = q->Wo, hw = q->Ho * Wo, stride = q->stride;
      const int b = PPP_DIV(m0, dHw), rem0 = m0 - b * hw;
      pbase_s = b * q->H * q->W;
      { const int ry = r0 / Wo; iyl = ry * stride; ixl = (r0 - ry * Wo) * stride; }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rr = rem0 + 128 * h + 64 * i;
          const int oy = PPP_DIV(rr, dWo), ox = rr - oy * Wo;
          tap_s[h][i] = ((oy * stride - pad_t) & 0xFFFF) | ((ox * stride - pad_l) << 16);
        }
        const int tap = PPP_DIV(kbeg, dCin);
        ccA[h] = kbeg - tap * Cin;
        dyA[h] = PPP_DIV(tap, dKW);
        dxA[h] = tap - dyA[h] * q->KW;
        set_tap(h);
      }
    } else {
      aoffk = (unsigned)((long)(m0 + r0) * q->lda * 2) + (unsigned)kl2;
    }
  };

  // stage the K tile at origin k0 of A half `h` into LDS stage `buf` (2 DMA instructions); conv: advance that stream's tap walk by one K tile
  auto stage_a = [&](int h, int buf, int k0) __attribute__((always_inline)) {
    unsigned char* dst = smem + buf * PP_STAGE + h * PP_HALF + wave * 1024;
    const unsigned kmask = k0 < kend ? 0u : kOOB;  // wave-uniform (K % 64 == 0): past the segment's K range the hardware writes zeros
    if constexpr (CONV) {
      const int co2 = ccA[h] * 2 + kl2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned voff = ((unsigned)(pix[h][i] * Cin) * 2u + (unsigned)co2) | ((unsigned)(pix[h][i] >> 31) & kOOB) | kmask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(dst + i * 8192), 16, voff, 0, 0, 0);
      }
      ccA[h] += BK;
      if (ccA[h] >= Cin) {
        ccA[h] = 0;
        if (++dxA[h] == p.KW) { dxA[h] = 0; ++dyA[h]; }
        set_tap(h);
      }
    } else {
      const unsigned voff = (aoffk + (unsigned)k0 * 2u) | kmask;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(dst + i * 8192), 16, voff, (128 * h + 64 * i) * (int)p.lda * 2, 0, 0);
    }
  };
  auto stage_w = [&](int h, int buf, int k0) __attribute__((always_inline)) {
    unsigned char* dst = smem + buf * PP_STAGE + 2 * PP_HALF + h * PP_HALF + wave * 1024;
    const unsigned voff = (woffk + (unsigned)k0 * 2u) | (k0 < kend ? 0u : kOOB);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(dst + i * 8192), 16, voff, (FF ? 32 * h + 128 * i : 128 * h + 64 * i) * (int)p.ldw * 2, 0, 0);
  };
  // the whole LDS ring of a segment: all of K tile 0, then A0 / W0 of tile 1 (the order the steady-state vmcnt counts assume)
  auto prologue = [&]() __attribute__((always_inline)) {
    stage_a(0, 0, kbeg);
    stage_w(0, 0, kbeg);
    stage_w(1, 0, kbeg);
    stage_a(1, 0, kbeg);
    stage_a(0, 1, kbeg + BK);
    stage_w(0, 1, kbeg + BK);
  };

  // fragment read offsets inside a half-tile: rows wr*64 + mt*32 + l31 (A) / wc*32 + l31 (W); the swizzle depends on l31 only.  Recomputed at the
  // top of every segment's K loop.
  int a_rd[4], w_rd[4];
  auto lane_setup = [&]() __attribute__((always_inline)) {
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cc = ((kk * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
      a_rd[kk] = ((wave >> 2) * 64 + l31) * 128 + cc;
      w_rd[kk] = ((wave & 3) * 32 + l31) * 128 + cc;
    }
    if constexpr (CONV) kl2 = lane_kl2(lane);
  };
  auto rd = [&](const unsigned char* ptr) __attribute__((always_inline)) -> f16x8 { return *reinterpret_cast<const f16x8*>(ptr); };
  auto mma = [&](const f16x8& w, const f16x8& a, f32x16& acc1) __attribute__((always_inline)) { acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, acc1, 0, 0, 0); };
  auto bar = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_barrier(); };

  // FF: this wave's share of the rows' LayerNorm sums, [A half][row band]: the four waves of a row band take one 16-wide K step of every K tile each
  float ls1[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, ls2[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
  auto ln_step = [&](int hA, const f16x8 (&fa)[2][4]) __attribute__((always_inline)) {
    if constexpr (FF) {
      const f16x2 ones = {(f16)1.0f, (f16)1.0f};
      static_for<4>::run([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        if ((wave & 3) == kk) {  // wave-uniform
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const f16x8 f = fa[mt][kk];
            const f16x2 h0 = __builtin_shufflevector(f, f, 0, 1), h1 = __builtin_shufflevector(f, f, 2, 3);
            const f16x2 h2 = __builtin_shufflevector(f, f, 4, 5), h3 = __builtin_shufflevector(f, f, 6, 7);
            float a = ls1[hA][mt], b = ls2[hA][mt];
            a = __builtin_amdgcn_fdot2(h0, ones, a, false); b = __builtin_amdgcn_fdot2(h0, h0, b, false);
            a = __builtin_amdgcn_fdot2(h1, ones, a, false); b = __builtin_amdgcn_fdot2(h1, h1, b, false);
            a = __builtin_amdgcn_fdot2(h2, ones, a, false); b = __builtin_amdgcn_fdot2(h2, h2, b, false);
            a = __builtin_amdgcn_fdot2(h3, ones, a, false); b = __builtin_amdgcn_fdot2(h3, h3, b, false);
            ls1[hA][mt] = a; ls2[hA][mt] = b;
          }
        }
      });
    }
  };

  f32x16 acc[4][1][2];  // [quadrant][TN = 1][TM = 2]; quadrants (A0,W0) (A0,W1) (A1,W1) (A1,W0)
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][0][i][r] = 0.0f;
  };

  // ---- hand-off slabs: [slot][wave][32][lane] float4, the accumulators in register order ----------------------------------------------------------
  auto publish = [&](int slot) __attribute__((always_inline)) {  // PRODUCER: write-through stores, every wave drains, one flag (Guideline 16 R1)
    KArgs* q = ppp_kargs();
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)q->ws, 0, (int)0x7FFFFFF0, 0x00020000);
    const unsigned base = ((unsigned)slot * PPP_SLAB + (unsigned)(wave * 32) * 256 + (unsigned)lane * 4) * 4u;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[qd][0][i][4 * g], acc[qd][0][i][4 * g + 1], acc[qd][0][i][4 * g + 2], acc[qd][0][i][4 * g + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_ws, base + (unsigned)(((qd * 2 + i) * 4 + g) * 1024), 0, 16);  // aux 16 = sc1
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave (and the next segment's ring has landed with it)
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(q->ppflags + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto wait_parts = [&](int slot, int nslot) __attribute__((always_inline)) {  // OWNER: the other parts' slabs are complete and visible
    KArgs* q = ppp_kargs();
    unsigned* flags = q->ppflags;
    if (threadIdx.x == 0) {
      for (int s = 0; s < nslot; ++s) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + slot + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > (1u << 24)) {  // bounded (seconds): count it and go on with what is there -- a hung launch would be worse
            __hip_atomic_fetch_add(q->pptmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __hip_atomic_store(flags + slot + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning: its one consumer resets it
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ONE acquire after the match: this CU's stale lines are dropped
    }
    __syncthreads();
  };

  // ---- the tile's epilogue ------------------------------------------------------------------------------------------------------------------
  // The bias vectors and (tiles with a residual or a time shift) the whole tile's per-element operand as raw 16-byte loads -- 16 + 64 registers beside
  // the 128 accumulators, which is what the file holds once the K loop's fragments are dead --, ONE drained wait, which is also the next segment's
  // ring, then quadrant by quadrant convert and store: every store of the tile behind every load of it.
  auto finish_tile = [&](int tm0, int tn0, int tz, int slot, int nslot) __attribute__((always_inline)) {  // nslot > 0: an OWNER's tile
    KArgs* q = ppp_kargs();
    int lane = lane_, wave = wave_;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 2, wc = wave & 3;
    const f16x4 zero = {(f16)0.0f, (f16)0.0f, (f16)0.0f, (f16)0.0f};
    const f16* bias = q->bias; const f16* shift = q->shift; const f16* res = q->res;
    const int N = q->N, act = q->act, rpb = q->rpb, res_first = q->res_first;
    const long ldr = q->ldr, ldshift = q->ldshift;
    const float out_scale = q->out_scale;
    const bool has_shift = shift != nullptr, has_res = res != nullptr;  // (never both: the planner keeps those on tile 15)
    const bool has_aux = has_shift || has_res;
    const bool a_pre = has_shift || (has_res && res_first), a_post = has_res && !res_first;
    f16* outb = q->out;
    const long ldo = q->ldo, ldo_hi = q->ldo_hi;
    const int orw = q->orw;
    const unsigned orw_mul = q->dOrw.mul, orw_shift = q->dOrw.shift;
    if (q->up_ph) outb += (long)(tz >> 1) * (ldo_hi >> 1) + (long)(tz & 1) * (ldo >> 1);
    const int mrow = tm0 + 64 * wr + l31, colb = tn0 + 32 * wc + 8 * hi;
    const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)q->ws, 0, (int)0x7FFFFFF0, 0x00020000);
    const unsigned pbase = ((unsigned)slot * PPP_SLAB + (unsigned)(wave * 32) * 256 + (unsigned)lane * 4) * 4u;  // the hand-off slabs' register order (publish)

    // the per-element operand of one quadrant as RAW 16-byte loads (lane l: columns 8 hi .. + 7 of each 16-column pair; the planner guarantees
    // 16-byte aligned rows) -- issued back to back; the lane-pair swap that turns them into this lane's accumulator columns happens at the use
    auto load_aux = [&](auto Q, uint4 (&ax)[2][2]) __attribute__((always_inline)) {
      constexpr int qd = decltype(Q)::value;
      constexpr int rowh = qd >= 2 ? 128 : 0, cs = (qd == 1 || qd == 2) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = tm0 + 64 * wr + rowh + 32 * i + l31;
        const f16* row = (has_res ? res + (long)m * ldr : shift + (long)(m / rpb) * ldshift) + tn0 + 32 * wc + 128 * cs + 8 * hi;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) ax[i][gp] = *reinterpret_cast<const uint4*>(row + 16 * gp);
      }
    };
    auto unpack_aux = [&](const uint4 (&raw)[2], f16x4 (&a)[4]) __attribute__((always_inline)) {  // gemm_common.h load_groups4's swap
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(raw[gp].x, raw[gp].z, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(raw[gp].y, raw[gp].w, false, false);
        const uint2 lo = make_uint2(r0[0], r1[0]), hi2 = make_uint2(r0[1], r1[1]);
        a[2 * gp] = *reinterpret_cast<const f16x4*>(&lo);
        a[2 * gp + 1] = *reinterpret_cast<const f16x4*>(&hi2);
      }
    };
    auto load_bias = [&](int cs, f16x4 (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = tn0 + 32 * wc + 128 * cs + 8 * g + 4 * hi;
        bv[g] = bias ? *reinterpret_cast<const f16x4*>(bias + nb) : zero;
      }
    };
    // one quadrant: bias / pre-activation operand / activation / scale / post operand, f16, 16-byte stores
    auto convert_store = [&](auto Q, auto AUX, const float (&bf)[16], const uint4 (&axr)[2][2]) __attribute__((always_inline)) {
      constexpr int qd = decltype(Q)::value;
      constexpr bool aux = decltype(AUX)::value;
      constexpr int rowh = qd >= 2 ? 128 : 0, cs = (qd == 1 || qd == 2) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = acc[qd][0][i][e];
        if (nslot > 0) {  // (wave-uniform) the other parts' f32 sums of this band, in slot order: one (quadrant, band) = 4 x 16 bytes per lane and slot
          for (int s = 0; s < nslot; ++s) {
            const unsigned base = pbase + (unsigned)s * (PPP_SLAB * 4u) + (unsigned)((qd * 2 + i) * 4096);
            f32x4 t[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) t[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, base + (unsigned)(g * 1024), 0, 0));
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += t[e >> 2][e & 3];
          }
        }
        if (bias) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += bf[e];
        }
        f16x4 ax[4];
        if constexpr (aux) {
          unpack_aux(axr[i], ax);
          if (a_pre) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += (float)ax[e >> 2][e & 3];
          }
        }
        switch (act) {  // one wave-uniform branch per pass (gemm_common.h epilogue_tile_math)
          case GN_ACT_SILU:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = act_silu(v[e]);
            break;
          case GN_ACT_GELU:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = gelu_fast(v[e]);
            break;
          case GN_ACT_QUICK_GELU:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = act_quick_gelu(v[e]);
            break;
          case GN_ACT_RELU:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.0f);
            break;
          default: break;
        }
        if (out_scale != 1.0f) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] *= out_scale;
        }
        if constexpr (aux) {
          if (a_post) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += (float)ax[e >> 2][e & 3];
          }
        }
        // (the row address from two 32-bit per-lane values and scalars, band by band: a 64-bit per-lane pointer kept across the bands is what the
        // allocator spills first, and its reloads would queue behind the stores already issued)
        const int m = mrow + rowh + 32 * i;
        long roff = (long)m * ldo;
        if (orw) { const int mh = fdiv(m, orw_mul, orw_shift); roff = (long)mh * ldo_hi + (long)(m - mh * orw) * ldo; }  // two-level row pitch (one phase of an upsampling conv)
        f16* orow = outb + (roff + (colb + 128 * cs));
#pragma unroll
        for (int g = 0; g < 4; g += 2) {  // lanes l / l + 32 trade channel groups: each owns 8 consecutive channels = one 16-byte store
          f16x4 ha, hb2;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ha[e] = (f16)v[4 * g + e]; hb2[e] = (f16)v[4 * g + 4 + e]; }
          const uint2 ua = *reinterpret_cast<const uint2*>(&ha), ub = *reinterpret_cast<const uint2*>(&hb2);
          const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
          *reinterpret_cast<uint4*>(orow + 8 * g) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
      }
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    f16x4 bv0[4], bv1[4];
    load_bias(0, bv0);
    load_bias(1, bv1);
    // (after a drained wait the bias vectors are USED on every path the compiler sees: a load it believes pending at the end of the epilogue becomes
    // a wait in front of the next K loop's first reuse of that register, and that wait sits out this tile's stores at run time.)  One f16 -> f32
    // conversion per column set, not per band: the four bands of a set share it.
    float bf[16];
    auto bias_f32 = [&](const f16x4 (&bv)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) bf[e] = (float)bv[e >> 2][e & 3];
    };
    auto touch_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(bv0[g]), "v"(bv1[g]));
    };
    if (!has_aux) {
      uint4 none[2][2];  // never read
      PPP_T(t_d0);
      PPP_DRAIN();  // the bias vectors AND the next segment's ring have landed; nothing of this tile is stored yet
      PPP_T(t_d1);
      PPP_ACC(7, t_d0, t_d1);
      touch_bias();
      bias_f32(bv0);
      convert_store(Q0{}, std::false_type{}, bf, none);
      convert_store(Q3{}, std::false_type{}, bf, none);
      bias_f32(bv1);
      convert_store(Q1{}, std::false_type{}, bf, none);
      convert_store(Q2{}, std::false_type{}, bf, none);
    } else {
      uint4 ax0[2][2], ax1[2][2], ax2[2][2], ax3[2][2];  // 64 registers beside the accumulators: the whole tile's operand in ONE round trip
      load_aux(Q0{}, ax0);
      load_aux(Q3{}, ax3);
      load_aux(Q1{}, ax1);
      load_aux(Q2{}, ax2);
      PPP_DRAIN();
      touch_bias();
      bias_f32(bv0);
      convert_store(Q0{}, std::true_type{}, bf, ax0);
      convert_store(Q3{}, std::true_type{}, bf, ax3);
      bias_f32(bv1);
      convert_store(Q1{}, std::true_type{}, bf, ax1);
      convert_store(Q2{}, std::true_type{}, bf, ax2);
    }
  };

  // ---- FF epilogue: finish the rows' LayerNorm statistics, then  out = (rstd (h - mean c1h) + bh) * gelu(rstd (g - mean c1g) + bg)  ------------------
  // h / g = the hidden / gate accumulators of a row band (quadrants (A, W0) / (A, W1)), c1 = the column sums of the gamma-scaled weight, b = c2
  // (gn_gemm_desc.ln_c1; gemm_common.h ln_fold_apply + the GEGLU branch of gemm_epilogue are the launches this replaces).  Loads (c1, bias) first,
  // one drained wait = the next segment's ring, then band by band compute and store: 8 x 16-byte stores per wave (the output is N / 2 wide).
  auto finish_tile_ff = [&](int tm0, int tn0) __attribute__((always_inline)) {
    if constexpr (FF) {
      KArgs* q = ppp_kargs();
      int lane = lane_, wave = wave_;
      asm volatile("" : "+v"(lane), "+s"(wave));
      const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 2, wc = wave & 3;
      // (1) the statistics: the two lane halves hold the two 8-element halves of every K step; the four waves of a row band one K step in four each
      float* st = reinterpret_cast<float*>(smem + 2 * PP_STAGE);
#pragma unroll
      for (int hA = 0; hA < 2; ++hA)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const float a = ls1[hA][mt] + __shfl_xor(ls1[hA][mt], 32), b = ls2[hA][mt] + __shfl_xor(ls2[hA][mt], 32);
          if (hi == 0) *reinterpret_cast<float2*>(st + ((wc * 256) + hA * 128 + wr * 64 + mt * 32 + l31) * 2) = make_float2(a, b);
          ls1[hA][mt] = 0.0f; ls2[hA][mt] = 0.0f;
        }
      // (2) this wave's columns: c1 (f32) and c2 (f16, in `bias`) of hidden block wc and gate block wc of the tile
      const float* c1 = q->ln_c1 + tn0 + 64 * wc + 4 * hi;
      const f16* c2 = q->bias + tn0 + 64 * wc + 4 * hi;
      f32x4 c1h[4], c1g[4];
      f16x4 bh[4], bg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        c1h[g] = *reinterpret_cast<const f32x4*>(c1 + 8 * g);
        c1g[g] = *reinterpret_cast<const f32x4*>(c1 + 32 + 8 * g);
        bh[g] = *reinterpret_cast<const f16x4*>(c2 + 8 * g);
        bg[g] = *reinterpret_cast<const f16x4*>(c2 + 32 + 8 * g);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave's partial sums are in LDS
      float nmr[2][2], rstd[2][2];   // -rstd * mean, rstd of this lane's rows
      const float invk = 1.0f / (float)q->K, eps = q->ln_eps;
#pragma unroll
      for (int hA = 0; hA < 2; ++hA)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float a = 0.0f, b = 0.0f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {  // fixed order: every wave of a row band computes the same bits
            const float2 v = *reinterpret_cast<const float2*>(st + ((w * 256) + hA * 128 + wr * 64 + mt * 32 + l31) * 2);
            a += v.x; b += v.y;
          }
          const float mean = a * invk;
          const float var = fmaxf(b * invk - mean * mean, 0.0f);
          rstd[hA][mt] = __frsqrt_rn(var + eps);
          nmr[hA][mt] = -rstd[hA][mt] * mean;
        }
      PPP_DRAIN();  // c1 / c2 AND the next segment's ring have landed; nothing of this tile is stored yet
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(c1h[g]), "v"(c1g[g]), "v"(bh[g]), "v"(bg[g]));
      f16* outp = q->out + (tn0 >> 1) + 32 * wc + 8 * hi;
      const long ldo = q->ldo;
      const int mrow = tm0 + 64 * wr + l31;
      static_for<4>::run([&](auto U) {
        constexpr int u = decltype(U)::value, hA = u >> 1, mt = u & 1;
        constexpr int qh = hA == 0 ? 0 : 3, qg = hA == 0 ? 1 : 2;
        const float rs = rstd[hA][mt], nm = nmr[hA][mt];
        f16x4 o[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = rs * acc[qh][0][mt][4 * g + e] + nm * c1h[g][e] + (float)bh[g][e];
            const float gv = rs * acc[qg][0][mt][4 * g + e] + nm * c1g[g][e] + (float)bg[g][e];
            o[g][e] = (f16)(hv * gelu_fast(gv));
          }
        f16* orow = outp + (long)(mrow + 128 * hA + 32 * mt) * ldo;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          const uint2 ua = *reinterpret_cast<const uint2*>(&o[g]), ub = *reinterpret_cast<const uint2*>(&o[g + 1]);
          const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
          *reinterpret_cast<uint4*>(orow + 8 * g) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
      });
    }
  };

  // ================================================================ the walk ==================================================================
  int nks;
  {
    const Seg first = get_seg(0);
    nks = first.k1 - first.k0;
    setup(first);
  }
  prologue();
  zero_acc();
  PPP_DRAIN();
  __builtin_amdgcn_s_barrier();

  f16x8 fa[2][4], fw0[4], fw1[4];
  for (int si = 0;; ++si) {
    lane_setup();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) bar();  // group 1 runs one barrier behind inside the K loop
    __builtin_amdgcn_sched_barrier(0);

    int kt = kbeg;  // K origin of the tile being multiplied
    PPP_T(t_k0);
    for (int t = 0; t < nks; ++t, kt += BK) {
      const int b = t & 1;
      const unsigned char* Ab = smem + b * PP_STAGE;
      const unsigned char* Wb = Ab + 2 * PP_HALF;
      const bool waits = t > 0;  // K iteration 0 reads only what the segment's prologue brought (landed before the barrier above): no counted wait,
                                 // so the previous tile's stores are not waited for until a whole K iteration has passed

      // ---------------- phase 0: quadrant (A0, W0) ----------------
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fw0[kk] = rd(Wb + w_rd[kk]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[mt][kk] = rd(Ab + a_rd[kk] + mt * 4096);
      ln_step(0, fa);
      stage_w(1, b ^ 1, kt + BK);
#ifdef GN_PPP_PROFILE
      if (t == 1) { PPP_T(tw0); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); PPP_T(tw1); PPP_ACC(4, tw0, tw1); }
#endif
      if (waits) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma(fw0[kk], fa[mt][kk], acc[0][0][mt]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);

      // ---------------- phase 1: quadrant (A0, W1) ----------------
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fw1[kk] = rd(Wb + PP_HALF + w_rd[kk]);
      stage_a(1, b ^ 1, kt + BK);
      if (waits) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma(fw1[kk], fa[mt][kk], acc[1][0][mt]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);

      // ---------------- phase 2: quadrant (A1, W1) ----------------
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) fa[mt][kk] = rd(Ab + PP_HALF + a_rd[kk] + mt * 4096);
      ln_step(1, fa);
      stage_a(0, b, kt + 2 * BK);
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma(fw1[kk], fa[mt][kk], acc[2][0][mt]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);

      // ---------------- phase 3: quadrant (A1, W0): no fragment reads ----------------
      stage_w(0, b, kt + 2 * BK);
      if (waits) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma(fw0[kk], fa[mt][kk], acc[3][0][mt]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      bar();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wr == 0) bar();  // pairs with group 1's extra barrier: every fragment read of the segment has been consumed, the LDS ring is free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's run-ahead (out-of-range, zero-fill) stages
    PPP_T(t_k1);
    PPP_ACC(0, t_k0, t_k1);
    __builtin_amdgcn_sched_barrier(0);

    // the finished segment (recomputed: only `si` crossed the K loop), then the NEXT segment's ring is requested before anything of this one is
    // loaded or stored
    const Seg done = get_seg(si);
    int tm0, tn0, tz;
    tile_origin(done.tile, tm0, tn0, tz);
    const int role = done.role;
    // a shared tile's partial sums go out first (once or twice per workgroup and launch): the accumulators are dead before the next segment's
    // loader state and ring come in.  The OWNER of a shared tile adds the other parts' slabs inside its epilogue (the accumulators are only READ there:
    // as values modified on one path they would need a second copy of themselves at the join, i.e. scratch)
    if constexpr (!FF) {
      if (role == ROLE_PRODUCER) publish(done.slot);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool more = si + 1 < seg_count();
    int nks_next = 0;
    if (more) {
      const Seg nxt = get_seg(si + 1);
      nks_next = nxt.k1 - nxt.k0;
      setup(nxt);
      prologue();
    }
    __builtin_amdgcn_sched_barrier(0);
    PPP_T(t_b1);
    PPP_ACC(1, t_k1, t_b1);
    if (role == ROLE_PRODUCER) {
      PPP_DRAIN();  // the ring of the next segment
    } else {
      if constexpr (FF) {
        finish_tile_ff(tm0, tn0);  // (the planner neither skews nor splits these problems: a tile's statistics want its whole K range)
      } else {
        if (role == ROLE_OWNER) wait_parts(done.slot, done.nslot);
        finish_tile(tm0, tn0, tz, done.slot, role == ROLE_OWNER ? done.nslot : 0);
      }
    }
    PPP_T(t_f1);
    PPP_ACC(role == ROLE_FULL ? 2 : 5, t_b1, t_f1);
    if (!more) break;
    nks = nks_next;
    zero_acc();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // every wave has waited for its pieces of the next segment's ring (the drained waits above)
    PPP_T(t_bar);
    PPP_ACC(3, t_f1, t_bar);
  }
#ifdef GN_PPP_PROFILE
  if (threadIdx.x == 0) {
    PPP_T(t_end);
    prof[6] = t_end - t_start;
    unsigned* dst = p.pptmo + PPP_POOL_HEAD + PPP_REGION * PPP_REGIONS + blockIdx.x * 8;
    for (int i = 0; i < 8; ++i) dst[i] = prof[i];
  }
#endif
}

// ---- the flag pool: zero-initialised once per device, self-cleaning afterwards ----------------------------------------------------------------------
unsigned* g_pool[64] = {};

}  // namespace

int32_t gn_ppp_pool_init(int device) {
  if (device < 0 || device >= 64) { gn_set_error("gn_ppp_pool_init: device %d out of range", device); return GN_ERR_INVALID; }
  if (g_pool[device]) return GN_OK;
  unsigned* ptr = nullptr;
  const size_t bytes = (size_t)(PPP_POOL_HEAD + PPP_REGION * PPP_REGIONS + PPP_PROF_WORDS) * sizeof(unsigned);
  GN_HIP(hipMalloc((void**)&ptr, bytes));
  GN_HIP(hipMemset(ptr, 0, bytes));
  GN_HIP(hipDeviceSynchronize());
  g_pool[device] = ptr;
  return GN_OK;
}

// rounds / tail split / skew of a problem of `tiles` 256 x 256 tiles and nk K iterations on G workgroups; -> hand-off slabs the launch may use
static GemmParams::FastDiv fast_div(unsigned d) {
  GemmParams::FastDiv f;
  if (d == 0) d = 1;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mul = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.shift = l;
  return f;
}

int gn_ppp_plan(void* params, int tiles, int G) {
  GemmParams& p = *static_cast<GemmParams*>(params);
  const int nk = p.K / BK;
  p.ppG = G;
  p.ppNz = p.up_ph ? 4 : 1;
  p.ppR = tiles / G;
  p.ppTail = tiles - p.ppR * G;
  // The tiles of the last partial round: split s ways along K where that pays.  A K iteration of a tile is ~1.8 us, a tile boundary ~9 us and a
  // hand-off (256 KB of partial sums out, drained, back in through the owner's epilogue) ~22 us (profiles/r06_ppp_*): a 16-iteration tile gains nothing.
  int s = 1;
  if (p.ppTail > 0) {
    static const int s_env = [] { const char* e = getenv("GN_PPP_TAIL_SPLIT"); return e ? atoi(e) : -1; }();  // A/B switch: force the split (1 = never)
    int smax = G / p.ppTail;
    if (smax > 8) smax = 8;
    while (smax > 1 && nk / smax < 2) --smax;  // every part walks at least two K iterations
    double best = 1.8 * nk;
    for (int c = 2; c <= smax; ++c) {
      const double cost = 1.8 * ((nk + c - 1) / c) + 22.0;
      if (cost < best) { best = cost; s = c; }
    }
    if (s_env >= 1) s = s_env < smax ? s_env : smax;
    if (s < 1) s = 1;
  }
  if (p.ln_c1) s = 1;  // the feed-forward variant takes a tile's LayerNorm statistics from its whole K range
  p.ppS = s;
  // SKEW (workgroup c enters its first tile at K iteration c * nk / G): measured NOT to pay -- with the next tile's ring requested ahead of the
  // epilogue and no wait behind the stores, the lock-step store bursts drain under the next K loop, and the skew's two hand-offs per workgroup cost
  // more than the desynchronisation wins (profiles/r06_ppp_ksweep*.txt: 67 / 155 / 505 us without against 79 / 164 / 548 with, K = 256 / 1024 / 4096 at
  // 1024 tiles).  GN_PPP_SKEW=1 turns it on (tests run both).
  static const int skew_env = [] { const char* e = getenv("GN_PPP_SKEW"); return e ? atoi(e) : 0; }();
  p.ppSkew = (skew_env && nk >= 4 && !p.ln_c1) ? 1 : 0;
  p.dG = fast_div((unsigned)G); p.dS = fast_div((unsigned)s);
  p.dTm = fast_div((unsigned)p.tiles_m); p.dTn = fast_div((unsigned)p.tiles_n); p.dTmn = fast_div((unsigned)(p.tiles_m * p.tiles_n));
  p.dOrw = fast_div((unsigned)p.orw);
  p.dHw = fast_div((unsigned)(p.Ho * p.Wo)); p.dWo = fast_div((unsigned)p.Wo); p.dCin = fast_div((unsigned)p.C1); p.dKW = fast_div((unsigned)p.KW);
  return 256 + p.ppTail * (s > 1 ? s - 1 : 0);
}

void gn_launch_gemm_ppp(const void* params, bool conv, hipStream_t st) {
  GemmParams p = *static_cast<const GemmParams*>(params);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !g_pool[dev]) { (void)gn_ppp_pool_init(dev); }
  // Flag regions.  Only launches that hand partial sums over use their region (a K-split tail, the skewed walk).  A launch being CAPTURED bakes its
  // region into the graph, which may be replayed at any later time beside anything else: it takes a region of the lower half of the pool for good
  // (never recycled; once those 512 are gone a captured launch runs without hand-offs: its tail unsplit, correct and slower).  Eager launches rotate
  // through the upper half: two of them could only collide with 512 flag-using launches in flight between them.
  static std::atomic<unsigned> next_captured{0}, next_eager{0};
  unsigned region = 0;
  const bool needs_flags = p.ppSkew || (p.ppTail > 0 && p.ppS > 1);
  if (needs_flags) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    if (capturing) {
      const unsigned r = next_captured.fetch_add(1);
      if (r < PPP_REGIONS / 2) {
        region = r;
      } else {  // pool exhausted: no hand-offs in this launch (every tile by one workgroup; the workspace named for the split plan is simply not used)
        p.ppSkew = 0;
        p.ppS = 1;
        p.dS.mul = 1; p.dS.shift = 0;
      }
    } else {
      region = PPP_REGIONS / 2 + next_eager.fetch_add(1) % (PPP_REGIONS / 2);
    }
  }
  p.ppflags = g_pool[dev] + PPP_POOL_HEAD + (size_t)region * PPP_REGION;
  p.pptmo = g_pool[dev];
  const dim3 grid(p.ppG);
  if (conv) hipLaunchKernelGGL((gemm_ppp_kernel<true, false>), grid, dim3(512), 0, st, p);
  else if (p.ln_c1) hipLaunchKernelGGL((gemm_ppp_kernel<false, true>), grid, dim3(512), 0, st, p);  // LayerNorm fold + GEGLU (the planner admits them together only)
  else hipLaunchKernelGGL((gemm_ppp_kernel<false, false>), grid, dim3(512), 0, st, p);
}

// profiling builds: the per-workgroup cycle sums of the LAST tile-25 launch (8 words per workgroup: K loops, boundary up to the ring request,
// plain epilogues, the barrier behind them, the first counted wait of a K loop, shared-tile epilogues, whole kernel, the drained wait)
extern "C" int32_t gn_ppp_profile_read(uint32_t* out, int32_t words) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_pool[dev] || !out || words < 0 || words > PPP_PROF_WORDS) return GN_ERR_INVALID;
  if (hipDeviceSynchronize() != hipSuccess) return GN_ERR_HIP;
  return hipMemcpy(out, g_pool[dev] + PPP_POOL_HEAD + PPP_REGION * PPP_REGIONS, (size_t)words * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? GN_OK : GN_ERR_HIP;
}

extern "C" int64_t gn_ppp_timeouts(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_pool[dev]) return 0;
  unsigned v = 0;
  if (hipMemcpy(&v, g_pool[dev], sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)v;
}
