// fp8 (OCP e4m3) flash-attention forward for D = 64 on v_mfma_scale_f32_32x32x64_f8f6f4 -- the opt-in attention of the fp8 training
// forward (SURVEY.md section 8 a15 / BASELINE configs[4] "fp8 MFMA"; the reference call site is xformers' attention under the SDXL
// ControlNet step, diffusion/train_controlnet_sdxl_genima.py:1448-1471).  NOT used by the f16 inference path: e4m3 probabilities
// sit outside its 2e-3 parity bar (tests/test_attention_fp8_gpu.py states the bounds this kernel is held to).
//
// Same transposed formulation as attention.hip (S^T = K . Q^T, O^T = V^T . P^T, the softmax statistics of a query row live in the
// lane pair {l, l + 32}), with both products on the K = 64 fp8 MFMA: ONE instruction per 32-key score tile (D = 64 is the whole
// reduction) and ONE per 32-row O^T tile and 64-key step -- 4 MFMAs of 16 passes per 64-key tile against 16 of 8 passes in f16.
//   * gn_attention_fp8_quantize makes the operands from the f16 q | k | v rows: Q8 = e4m3(q * scale * log2 e) and K8 = e4m3(k),
//     row-major bytes; V8T = e4m3(v) transposed to [b][h*64 + d][key] with the keys of every 64-key tile stored in MFMA ORDER:
//     byte 32*hi + 16*u + 4*g + i of a tile row is key 32*u + 8*g + 4*hi + i -- exactly the key whose probability the lane half `hi`
//     holds in accumulator 4*g + i of score sub-tile u (the 32x32 D layout), so the P^T B-operand is the 32 converted accumulators in
//     register order and the V^T A-operand is 32 contiguous bytes (two ds_read_b128).  No unscaled operand exceeds e4m3's range in
//     the networks this serves (|q|, |k|, |v| of a few units); the quantiser saturates at +-448 instead of scaling.
//   * P' = exp2(s - m) is converted with v_cvt_pk_fp8_f32; the reference m is kept 3 exponent units BELOW the row maximum (P' <= 8
//     at the reference, so a flat row sums to 256 per lane) and the optimistic path of attention.hip is kept: no row max, a lane's
//     32-key sum above 448 (the e4m3 maximum; also inf / NaN) sends the tile down the careful path, which re-references.  Whatever
//     passes has every P' <= 448.  Values below 2^-9 (2^-12 of the reference maximum) flush to zero.
//   * the row sum l is taken on the f32 exponentials (v_pk_add_f32), O^T and l carry the same factor.
// Measured (MI355X, tools/bench_attn.py, profiles/r03_v5_attention_microbench.txt): 875 / 993 TFLOP/s at 8 x 5 / 8 x 10 x 4096^2 against
// 848 / 910 for the f16 kernel -- +3..9 % for the kernel alone, LESS than f16 once the operand pass is counted.  The matrix pipe's
// share halves (256 of its cycles per 64-key wave tile instead of 512) but the softmax gets heavier, and it is what bounds both kernels
// (tools/probes/valu_rates.hip, SIMD cycles per wave instruction: v_exp_f32 8.1 -- half rate, no overlap with other VALU work --,
// v_cvt_pk_fp8_f32 8.2 against 4.1 for v_cvt_pkrtz_f16_f32, v_pk_add_f32 6.2 against 4.1 for v_dot2c_f32_f16): per score 8 + 4.1 + 3.1
// = 15.2 VALU cycles here against 8 + 2 + 2 = 12 in f16, beside 8 (fp8) or 16 (f16) MFMA cycles.  At D = 64 an e4m3 attention is bound
// by the conversion, not by the products.  Kept as a tested opt-in (Engine.attention_fp8); nothing routes through it by default.
// LDS: K tile 64 keys x 64 B and V^T tile 64 d x 64 B, each stored as 32 physical rows of 128 B (row r = logical rows r and r + 32
// side by side) in the 16-byte-chunk XOR swizzle of common.h, double-buffered, filled by LDS-DMA (one K and one V^T instruction per
// wave and tile).
#include "attention_common.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float P8_LIM = 448.0f;   // lane sum of 32 probabilities that sends a tile down the careful path (e4m3 maximum)
constexpr float P8_REF = 3.0f;     // the reference sits this many exponent units below the row maximum

struct Attn8Params {
  const unsigned char* q; const unsigned char* k; const unsigned char* vt; f16* o;
  long q_bs, k_bs, vt_bs, o_bs;
  int q_rs, k_rs, vt_rs, o_rs;
  int heads, Nq, Nk, causal;
  float* lse;
};

// two waves per SIMD: the kernel wants 182 registers (at three, 168 + spills of Q and O^T inside the key loop: 389 against 875 TFLOP/s)
__global__ __launch_bounds__(256, 2) void attn_fp8_fwd_kernel(const Attn8Params p) {
  constexpr int NW = 4, NT = 256, QB = 128;
  constexpr int TILE = 4096;  // bytes of a K (or V^T) tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = (p.Nq + QB - 1) / QB, total = gridDim.x;
  const int slot = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;  // XCD-aware (attention.hip)
  const int bh = __builtin_amdgcn_readfirstlane(slot / nqb);
  const int b = __builtin_amdgcn_readfirstlane(bh / p.heads), h = bh - b * p.heads;
  const int q0 = (slot - bh * nqb) * QB;
  const int qrow = q0 + wave * 32 + l31;

  const unsigned char* qp = p.q + (long)b * p.q_bs + (long)h * 64;
  const unsigned char* kp = p.k + (long)b * p.k_bs + (long)h * 64;
  const unsigned char* vp = p.vt + (long)b * p.vt_bs + (long)h * 64 * p.vt_rs;

  // Q fragment (B operand): the lane holds Q8[qrow][32*hi .. 32*hi + 31]
  i32x8 qf = {0, 0, 0, 0, 0, 0, 0, 0};
  if (qrow < p.Nq) {
    const uint4 lo = *reinterpret_cast<const uint4*>(qp + (long)qrow * p.q_rs + hi * 32);
    const uint4 up = *reinterpret_cast<const uint4*>(qp + (long)qrow * p.q_rs + hi * 32 + 16);
    qf = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)up.x, (int)up.y, (int)up.z, (int)up.w};
  }

  f32x16 oacc[2], negm;
  float m_run = 0.0f, l_run = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { negm[r] = 0.0f; oacc[0][r] = 0.0f; oacc[1][r] = 0.0f; }

  int nk_eff = p.Nk;
  if (p.causal) nk_eff = min(p.Nk, q0 + QB);
  const int ntiles = (nk_eff + KT - 1) / KT;

  // LDS-DMA: instruction `wv` of a tile fills physical rows 8*wv .. 8*wv + 7 lane-linearly; the lane that owns (physical row, physical
  // chunk) fetches logical chunk lc = chunk ^ ((row >> 1) & 7): logical row 32*(lc >> 2) + row, bytes 16*(lc & 3) of its 64
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned koff, voff;
  {
    const int prow = 8 * wv + (lane >> 3);
    const int lc = (lane & 7) ^ ((prow >> 1) & 7);
    const int lrow = 32 * (lc >> 2) + prow;
    koff = (unsigned)((long)lrow * p.k_rs + 16 * (lc & 3));
    voff = (unsigned)((long)lrow * p.vt_rs + 16 * (lc & 3));
  }
  const long kbytes = (long)(p.Nk - 1) * p.k_rs + 64;
  const long vbytes = (long)63 * p.vt_rs + (long)((p.Nk + KT - 1) / KT) * KT;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, (int)kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, (int)vbytes, 0x00020000);
  auto dma_tile = [&](int buf) {
    unsigned char* Ks = smem + buf * (2 * TILE);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (attn_lds_ptr_t)(Ks + wv * 1024), 16, koff, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (attn_lds_ptr_t)(Ks + TILE + wv * 1024), 16, voff, 0, 0, 0);
    koff += (unsigned)(KT * p.k_rs);
    voff += (unsigned)KT;
  };
  // fragment of logical row 32*half + l31: bytes 32*hi .. 32*hi + 31 = logical chunks 4*half + 2*hi, + 1 of physical row l31
  auto frag = [&](const unsigned char* T, int half) -> i32x8 {
    const uint4 lo = *reinterpret_cast<const uint4*>(T + lds_swz<128>(l31, 4 * half + 2 * hi));
    const uint4 up = *reinterpret_cast<const uint4*>(T + lds_swz<128>(l31, 4 * half + 2 * hi + 1));
    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)up.x, (int)up.y, (int)up.z, (int)up.w};
  };

  if (ntiles > 0) dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) dma_tile(cur ^ 1);
    const unsigned char* Ks = smem + cur * (2 * TILE);
    const unsigned char* Vs = Ks + TILE;
    const int j0 = t * KT;
    const bool need_mask = (j0 + KT > p.Nk) || (p.causal && j0 + KT - 1 > q0);  // block-uniform

    f32x16 s[2];
    auto scores = [&]() {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        s[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(Ks, u), qf, negm, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    };
    // P' = exp2(S') as e4m3 bytes in accumulator order (byte 16*u + r of the B operand); returns the lane's part of the row sum
    i32x8 pf;
    float psum;
    auto exps = [&]() {
      f32x2 acc = {0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x2 e0 = {__builtin_amdgcn_exp2f(s[u][4 * g]), __builtin_amdgcn_exp2f(s[u][4 * g + 1])};
          f32x2 e1 = {__builtin_amdgcn_exp2f(s[u][4 * g + 2]), __builtin_amdgcn_exp2f(s[u][4 * g + 3])};
          acc += e0;
          acc += e1;
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(e0[0], e0[1], w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(e1[0], e1[1], w, true);
          pf[4 * u + g] = w;
        }
      psum = acc[0] + acc[1];
    };

    scores();
    const bool careful = need_mask || t == 0;
    bool redo = careful;
    if (!careful) {
      exps();
      redo = __any(!(psum <= P8_LIM));
      if (redo) scores();
    }
    if (redo) {
      if (need_mask) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = j0 + 32 * u + 8 * (r >> 2) + 4 * hi + (r & 3);
            const bool dead = (key >= p.Nk) || (p.causal && key > qrow);
            s[u][r] = dead ? -INFINITY : s[u][r];
          }
      }
      float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[1][0]);
      mx = fmaxf(mx, s[1][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) {
        mx = fmaxf(fmaxf(mx, s[0][r]), s[0][r + 1]);
        mx = fmaxf(fmaxf(mx, s[1][r]), s[1][r + 1]);
      }
      mx = pair_max(mx);  // relative to the current reference
      // the reference is kept P8_REF below the largest score seen; it only grows, except on the first tile where it is set
      const float want = mx - P8_REF;
      const float delta = mx == -INFINITY ? 0.0f : (t == 0 ? want : fmaxf(want, 0.0f));
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m_run += delta;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m_run;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][r] -= delta;
      exps();
    }
    l_run += psum;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      oacc[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(Vs, dt), pf, oacc[dt], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  const float l_tot = pair_sum(l_run);
  const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (p.lse && hi == 0 && qrow < p.Nq)
    p.lse[((long)b * p.heads + h) * p.Nq + qrow] = l_tot > 0.0f ? m_run + __builtin_amdgcn_logf(l_tot) : INFINITY;
  if (qrow < p.Nq) {
    f16* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_rs + (long)h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (f16)(oacc[dt][4 * g + i] * inv);
        *reinterpret_cast<f16x4*>(op + dt * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

// ---- the operands: one block per (batch, 64-row tile, head) converts the 64 x 64 q, k and v tiles ------------------------------------
__device__ __forceinline__ float sat448(float x) { return fminf(fmaxf(x, -448.0f), 448.0f); }

__global__ __launch_bounds__(256) void attn_fp8_quantize_kernel(const f16* q, const f16* k, const f16* v, long q_rs, long k_rs, long v_rs,
                                                                long q_bs, long k_bs, long v_bs, int N, int heads, float qscale,
                                                                unsigned char* q8, unsigned char* k8, unsigned char* v8t, int Npad) {
  __shared__ f16 vs[64][66];  // the V tile, [key][d] (+2 pad: the transposed reads below walk a column)
  const int tid = threadIdx.x;
  const int h = blockIdx.x, tile = blockIdx.y, b = blockIdx.z;
  const int C = heads * 64;
  const int r0 = tile * 64;
  // q, k: thread -> (row = tid / 8 (+32), 8 columns)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = r0 + pass * 32 + (tid >> 3), col = h * 64 + (tid & 7) * 8;
    uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq;
    if (row < N) {
      vq = *reinterpret_cast<const uint4*>(q + (long)b * q_bs + (long)row * q_rs + col);
      vk = *reinterpret_cast<const uint4*>(k + (long)b * k_bs + (long)row * k_rs + col);
      vv = *reinterpret_cast<const uint4*>(v + (long)b * v_bs + (long)row * v_rs + col);
    }
    const f16x8 hq = *reinterpret_cast<const f16x8*>(&vq), hk = *reinterpret_cast<const f16x8*>(&vk), hv = *reinterpret_cast<const f16x8*>(&vv);
    int a0 = 0, a1 = 0, c0 = 0, c1 = 0;
    a0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hq[0] * qscale), sat448((float)hq[1] * qscale), a0, false);
    a0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hq[2] * qscale), sat448((float)hq[3] * qscale), a0, true);
    a1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hq[4] * qscale), sat448((float)hq[5] * qscale), a1, false);
    a1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hq[6] * qscale), sat448((float)hq[7] * qscale), a1, true);
    c0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hk[0]), sat448((float)hk[1]), c0, false);
    c0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hk[2]), sat448((float)hk[3]), c0, true);
    c1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hk[4]), sat448((float)hk[5]), c1, false);
    c1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)hk[6]), sat448((float)hk[7]), c1, true);
    if (row < N) {
      *reinterpret_cast<uint2*>(q8 + ((long)b * N + row) * C + col) = make_uint2((unsigned)a0, (unsigned)a1);
      *reinterpret_cast<uint2*>(k8 + ((long)b * N + row) * C + col) = make_uint2((unsigned)c0, (unsigned)c1);
    }
    const int kr = pass * 32 + (tid >> 3), dc = (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) vs[kr][dc + i] = hv[i];  // rows >= N hold zeros
  }
  __syncthreads();
  // v: thread -> (d = tid & 63, 16 positions of the tile row): position 32*hi + 16*u + 4*g + i is key 32*u + 8*g + 4*hi + i
  const int d = tid & 63, part = tid >> 6;  // part = 2*hi + u
  const int hi = part >> 1, u = part & 1;
  int w[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int key = 32 * u + 8 * g + 4 * hi;
    int x = 0;
    x = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)vs[key][d]), sat448((float)vs[key + 1][d]), x, false);
    x = __builtin_amdgcn_cvt_pk_fp8_f32(sat448((float)vs[key + 2][d]), sat448((float)vs[key + 3][d]), x, true);
    w[g] = x;
  }
  *reinterpret_cast<uint4*>(v8t + ((long)b * C + h * 64 + d) * Npad + r0 + part * 16) = make_uint4((unsigned)w[0], (unsigned)w[1], (unsigned)w[2], (unsigned)w[3]);
}

}  // namespace

extern "C" int32_t gn_attention_fp8_quantize(gn_ctx* ctx, const void* q, const void* k, const void* v, int64_t q_rs, int64_t k_rs,
                                             int64_t v_rs, int64_t q_bs, int64_t k_bs, int64_t v_bs, int32_t B, int32_t N, int32_t heads,
                                             float scale, void* q8, void* k8, void* v8t, int32_t Npad) {
  GN_REQUIRE(ctx && q && k && v && q8 && k8 && v8t, "gn_attention_fp8_quantize: null pointer");
  GN_REQUIRE(B > 0 && N > 0 && heads > 0 && scale > 0.0f, "gn_attention_fp8_quantize: empty problem");
  GN_REQUIRE(Npad % 64 == 0 && Npad >= N, "gn_attention_fp8_quantize: Npad must be round_up(N, 64) or more, a multiple of 64");
  GN_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 &&
                 (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q8 | (uintptr_t)k8 | (uintptr_t)v8t) & 15) == 0,
             "gn_attention_fp8_quantize: 16-byte alignment of pointers and strides");
  GN_REQUIRE(B <= 65535 && (N + 63) / 64 <= 65535, "gn_attention_fp8_quantize: grid too large");
  hipLaunchKernelGGL(attn_fp8_quantize_kernel, dim3(heads, (N + 63) / 64, B), dim3(256), 0, ctx->stream, (const f16*)q, (const f16*)k,
                     (const f16*)v, (long)q_rs, (long)k_rs, (long)v_rs, (long)q_bs, (long)k_bs, (long)v_bs, N, heads,
                     scale * 1.4426950408889634f, (unsigned char*)q8, (unsigned char*)k8, (unsigned char*)v8t, Npad);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

extern "C" int32_t gn_attention_fp8_fwd(gn_ctx* ctx, const gn_attn_desc* d) {
  GN_REQUIRE(ctx && d && d->q && d->k && d->vt && d->o, "gn_attention_fp8_fwd: null pointer");
  GN_REQUIRE(d->D == 64, "gn_attention_fp8_fwd: head dim %d unsupported (64)", d->D);
  GN_REQUIRE(d->B > 0 && d->heads > 0 && d->Nq > 0 && d->Nk > 0, "gn_attention_fp8_fwd: empty problem");
  GN_REQUIRE(!d->v_rowmajor, "gn_attention_fp8_fwd: vt is the transposed, tile-permuted V8T of gn_attention_fp8_quantize");
  GN_REQUIRE(d->q_rs % 16 == 0 && d->k_rs % 16 == 0 && d->vt_rs % 64 == 0 && d->o_rs % 4 == 0 && d->vt_rs >= ((d->Nk + 63) / 64) * 64,
             "gn_attention_fp8_fwd: row strides (bytes for q / k / vt; vt must cover round_up(Nk, 64))");
  GN_REQUIRE(((uintptr_t)d->q & 15) == 0 && ((uintptr_t)d->k & 15) == 0 && ((uintptr_t)d->vt & 15) == 0 && ((uintptr_t)d->o & 7) == 0 &&
                 d->q_bs % 16 == 0 && d->k_bs % 16 == 0 && d->vt_bs % 16 == 0 && d->o_bs % 4 == 0,
             "gn_attention_fp8_fwd: pointer / batch stride alignment");
  GN_REQUIRE((int64_t)d->Nk * d->k_rs < 0x7FFFFF00ll && (int64_t)64 * d->vt_rs < 0x7FFFFF00ll, "gn_attention_fp8_fwd: operand too large for 32-bit buffer offsets");
  Attn8Params p;
  p.q = (const unsigned char*)d->q; p.k = (const unsigned char*)d->k; p.vt = (const unsigned char*)d->vt; p.o = (f16*)d->o;
  p.q_bs = d->q_bs; p.k_bs = d->k_bs; p.vt_bs = d->vt_bs; p.o_bs = d->o_bs;
  p.q_rs = d->q_rs; p.k_rs = d->k_rs; p.vt_rs = d->vt_rs; p.o_rs = d->o_rs;
  p.heads = d->heads; p.Nq = d->Nq; p.Nk = d->Nk; p.causal = d->causal;
  p.lse = d->lse;
  const int nqb = (d->Nq + 127) / 128;
  hipLaunchKernelGGL(attn_fp8_fwd_kernel, dim3(nqb * d->heads * d->B), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
