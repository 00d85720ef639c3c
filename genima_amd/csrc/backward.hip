// Training-side kernels of the ControlNet fine-tune step (reference
// diffusion/train_controlnet_genima.py:1317-1408).  The matrix products of the backward pass reuse the forward MFMA GEMM
// (gemm.hip: dX = dY . W via a transposed weight copy, dW = dY^T . X via transposed activations with f32 output and split-K
// over the pixel dimension, conv dgrad = conv with rotated weights, conv wgrad = GEMM over an im2col^T image); this file holds
// the HBM-bound glue: tiled transposes (plain and im2col-gathering), column sums (bias / shift gradients), activation /
// GEGLU / softmax / LayerNorm / GroupNorm backward, zero-insertion and 2x2 sum pooling (strided / upsampled conv dgrad),
// MSE loss, and the flat-buffer optimizer kernels (fused AdamW, sum of squares, scale, f32 -> f16 cast).
// f16 storage, f32 math, f32 parameter gradients, deterministic reductions (no float atomics).
#include "common.h"

namespace {

inline unsigned nblk(long n, int t = 256) { return (unsigned)((n + t - 1) / t); }

// ---- tiled transpose: out[b][c][r] = in[b][r][c] -----------------------------------------------------------------------
// zrows: output columns [rows, zrows) are written as zeros (gn_transpose2d: zrows = rows -- nothing past the data is touched;
// gn_transpose2d_zpad: zrows = ld_out <= rup(rows, 64), the GEMM-operand padding without a fill launch in front)
__global__ __launch_bounds__(256) void transpose2d_kernel(const f16* __restrict__ in, f16* __restrict__ out, int rows, int cols,
                                                          long ld_in, long ld_out, long in_bs, long out_bs, int zrows) {
  __shared__ f16 tile[64][66];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  in += (long)b * in_bs;
  out += (long)b * out_bs;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : (f16)0.0f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < zrows) out[(long)c * ld_out + r] = tile[tx][i];
  }
}

// 16-byte variant (cols, ld_in, ld_out, batch strides multiples of 8; 16-byte aligned bases; ld_out >= rup(rows, 8)): a 64x64 tile is
// read as 8-element row chunks, parked in LDS with a 33-word row pitch (conflict-free column reads: lanes differ by 8 rows = 264
// words = 8 banks), and written back as 8-element chunks of the transposed rows.  Rows >= `rows` inside the last 8-row granule
// come out as zeros (the GEMM-operand padding).
__device__ __forceinline__ void tile_store_chunk(uint32_t (*tile)[33], int r, int chunk, uint4 v) {
  uint32_t* d = &tile[r][chunk * 4];
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}
__device__ __forceinline__ uint4 tile_load_column_chunk(uint32_t (*tile)[33], int r8, int col) {
  // elements tile[r8 .. r8+7][col] packed as 8 f16
  uint32_t h[8];
  const int w = col >> 1, sh = (col & 1) * 16;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (tile[r8 + i][w] >> sh) & 0xffffu;
  return make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}
// SUMS: also write the column sums of the tile's 64 rows -- part[blockIdx.y][col] -- the first stage of a bias / time-shift gradient
// (the weight-gradient path transposes dY anyway; reading it a second time for gn_colsum_f32 cost 1.5 ms per train step)
template <bool SUMS>
__global__ __launch_bounds__(256) void transpose2d_vec_kernel(const f16* __restrict__ in, f16* __restrict__ out, int rows, int cols, long ld_in,
                                                              long ld_out, long in_bs, long out_bs, float* __restrict__ part, int row_tiles,
                                                              int zrows) {  // zrows: see transpose2d_kernel (a multiple of 8 or == rows)
  // row_tiles (SUMS only): consecutive 64-row tiles one block walks, so that the partial-sum matrix stays short (<= 128 rows: the
  // second-stage reduction reads all of it)
  __shared__ uint32_t tile[64][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 64;
  const int t = threadIdx.x, lo = t & 7, hi = t >> 3;
  in += (long)b * in_bs;
  out += (long)b * out_bs;
  float colacc[2] = {0.f, 0.f};
  for (int rt = 0; rt < row_tiles; ++rt) {
    const int r0 = (blockIdx.y * row_tiles + rt) * 64;
    if (rt > 0) __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = r0 + hi + 32 * it, c = c0 + lo * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < rows && c < cols) v = *reinterpret_cast<const uint4*>(in + (long)r * ld_in + c);
      tile_store_chunk(tile, hi + 32 * it, lo, v);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int oc = hi + 32 * it, c = c0 + oc, r = r0 + lo * 8;
      const uint4 v = tile_load_column_chunk(tile, lo * 8, oc);  // rows past `rows` were stored as zeros
      if (c < cols && r < zrows) *reinterpret_cast<uint4*>(out + (long)c * ld_out + r) = v;
      if (SUMS) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(&v);
#pragma unroll
        for (int e = 0; e < 8; ++e) colacc[it] += (float)h[e];
      }
    }
  }
  if (SUMS) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float a = colacc[it];
      a += __shfl_xor(a, 1);  // the 8 threads of one output row are neighbouring lanes
      a += __shfl_xor(a, 2);
      a += __shfl_xor(a, 4);
      const int c = c0 + hi + 32 * it;
      if (lo == 0 && c < cols) part[(long)blockIdx.y * cols + c] = a;
    }
  }
}

// ---- many transposes in ONE launch: the trainable net's ~140 derived weight copies (W^T for the Linears' data gradients, the rotated
// conv weights) are rebuilt after every optimizer step -- as separate 6 us launches they sit in the step's dependent chain.  `items` is a
// device table (fixed pointers: the f16 working copy and the persistent outputs), block_begin ascending; a block finds its item by bisection.
__global__ __launch_bounds__(256) void transpose2d_multi_kernel(const gn_transpose_item* __restrict__ items, int n_items) {
  __shared__ uint32_t tile[64][33];
  int lo_i = 0, hi_i = n_items - 1;
  const int bid = blockIdx.x;
  while (lo_i < hi_i) {  // last item whose block_begin <= bid (block-uniform)
    const int mid = (lo_i + hi_i + 1) >> 1;
    if (items[mid].block_begin <= bid) lo_i = mid; else hi_i = mid - 1;
  }
  const gn_transpose_item it_ = items[lo_i];
  const int local = bid - it_.block_begin;
  const int tx = (it_.cols + 63) / 64, ty = (it_.rows + 63) / 64;
  const int b = local / (tx * ty), rem = local - b * (tx * ty);
  const int c0 = (rem % tx) * 64, r0 = (rem / tx) * 64;
  const int t = threadIdx.x, lo = t & 7, hi = t >> 3;
  const f16* in = (const f16*)it_.in + (long)b * it_.in_bs;
  f16* out = (f16*)it_.out + (long)b * it_.out_bs;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = r0 + hi + 32 * k, c = c0 + lo * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < it_.rows && c < it_.cols) v = *reinterpret_cast<const uint4*>(in + (long)r * it_.ld_in + c);
    tile_store_chunk(tile, hi + 32 * k, lo, v);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int oc = hi + 32 * k, c = c0 + oc, r = r0 + lo * 8;
    const uint4 v = tile_load_column_chunk(tile, lo * 8, oc);
    if (c < it_.cols && r < it_.rows) *reinterpret_cast<uint4*>(out + (long)c * it_.ld_out + r) = v;
  }
}

// ---- im2col^T: out[(tap*C + c)][m] = x[b, oy*stride - pad + dy, ox*stride - pad + dx, c]  (0 in the padding) ---------------------
// Same 64x64 tile / 16-byte scheme as transpose2d_vec_kernel, with the row (pixel) address computed per tap.  C % 8 == 0, M % 8 == 0.
struct Im2colP { const f16* x; f16* out; int B, H, W, C, KH, KW, stride, pad, Ho, Wo; long M; };
__global__ __launch_bounds__(256) void im2col_t_kernel(const Im2colP p) {
  __shared__ uint32_t tile[64][33];
  const int tap = blockIdx.z, dy = tap / p.KW, dx = tap - dy * p.KW;
  const long m0 = (long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int t = threadIdx.x, lo = t & 7, hi = t >> 3;
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const long m = m0 + hi + 32 * it;
    const int c = c0 + lo * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < p.M && c < p.C) {
      const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy = oy * p.stride - p.pad + dy, ix = ox * p.stride - p.pad + dx;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const uint4*>(p.x + (((long)b * p.H + iy) * p.W + ix) * p.C + c);
    }
    tile_store_chunk(tile, hi + 32 * it, lo, v);
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int oc = hi + 32 * it, c = c0 + oc;
    const long m = m0 + lo * 8;
    if (c < p.C && m < p.M) *reinterpret_cast<uint4*>(p.out + ((long)tap * p.C + c) * p.M + m) = tile_load_column_chunk(tile, lo * 8, oc);
  }
}

// ---- column sums: out[nb][cols] (+)= sum over each batch's rows; two deterministic stages ----------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const f16* __restrict__ x, float* __restrict__ part, int rpb, int cols,
                                                             long ld, int chunks) {
  // grid: (ceil(cols/128), chunks, nb); a thread owns 8 consecutive columns (one 16-byte load per row), 16 row lanes per block
  __shared__ float red[16][129];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 128 + tx * 8;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const int rows_per = (rpb + chunks - 1) / chunks;
  const int r0 = chunk * rows_per, r1 = min(rpb, r0 + rows_per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cols)
    for (int r = r0 + ty; r < r1; r += 16) {
      const uint4 raw = *reinterpret_cast<const uint4*>(x + ((long)b * rpb + r) * ld + c);
      const f16x8 h = *reinterpret_cast<const f16x8*>(&raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)h[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = s[e];
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 128 && blockIdx.x * 128 + t < cols) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += red[i][t];
    part[((long)b * chunks + chunk) * cols + blockIdx.x * 128 + t] = a;
  }
}
__global__ void reduce_rows_f32_kernel(const float* __restrict__ part, float* __restrict__ out, int groups, int R, int cols, int accumulate) {
  // out[g][c] (+)= sum_{r < R} part[(g*R + r)][c].  A block of 4 waves owns 64 consecutive (g, c) outputs; wave w sums rows
  // w, w+4, ... with 4 independent accumulators (16 loads in flight per output), then the waves combine in a fixed order.
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (idx < (long)groups * cols) {
    const int g = (int)(idx / cols), c = (int)(idx - (long)g * cols);
    const float* src = part + (long)g * R * cols + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = w;
    for (; r + 12 < R; r += 16) {
      s0 += src[(long)r * cols];
      s1 += src[(long)(r + 4) * cols];
      s2 += src[(long)(r + 8) * cols];
      s3 += src[(long)(r + 12) * cols];
    }
    for (; r < R; r += 4) s0 += src[(long)r * cols];
    s = (s0 + s1) + (s2 + s3);
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && idx < (long)groups * cols) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[idx] = accumulate ? out[idx] + t : t;
  }
}

// ---- activation backward: dz = dy * act'(z) ---------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case GN_ACT_SILU: { const float s = 1.0f / (1.0f + __expf(-z)); return s * (1.0f + z * (1.0f - s)); }
    case GN_ACT_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
    case GN_ACT_QUICK_GELU: { const float s = 1.0f / (1.0f + __expf(-1.702f * z)); return s * (1.0f + 1.702f * z * (1.0f - s)); }
    case GN_ACT_RELU: return z > 0.0f ? 1.0f : 0.0f;
    default: return 1.0f;
  }
}
__global__ void act_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ z, uint4* __restrict__ dz, long n8, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 a = dy[i], b = z[i];
  const f16x8 va = *reinterpret_cast<const f16x8*>(&a), vb = *reinterpret_cast<const f16x8*>(&b);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)((float)va[e] * act_grad((float)vb[e], act));
  dz[i] = *reinterpret_cast<uint4*>(&o);
}

// ---- GEGLU (unfused form used in training): out = hidden * gelu(gate) -------------------------------------------------------------
// blk == 0: hg = [hidden | gate] halves;  blk > 0: alternating blk-column blocks [hidden | gate] (the packed ff.net.0.proj layout of
// the fused GN_ACT_GEGLU GEMM, packing.pack_geglu), so the training path shares the inference weight layout.
__device__ __forceinline__ long geglu_hidden_col(int j, int Hd, int blk, int* gate_off) {
  if (blk > 0) {
    *gate_off = blk;
    return (long)(j / blk) * 2 * blk + (j % blk);
  }
  *gate_off = Hd;
  return j;
}
// a thread handles 8 consecutive outputs (Hd and blk multiples of 8: the 8 hidden / gate inputs are one 16-byte chunk each)
__global__ void geglu_fwd_kernel(const f16* __restrict__ hg, f16* __restrict__ out, long M, int Hd, int blk) {
  const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (idx >= M * Hd) return;
  const long m = idx / Hd;
  int go;
  const long hc = m * 2 * Hd + geglu_hidden_col((int)(idx - m * Hd), Hd, blk, &go);
  const uint4 hr = *reinterpret_cast<const uint4*>(hg + hc), gr = *reinterpret_cast<const uint4*>(hg + hc + go);
  const f16x8 hh = *reinterpret_cast<const f16x8*>(&hr), gh = *reinterpret_cast<const f16x8*>(&gr);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)((float)hh[e] * act_gelu((float)gh[e]));
  *reinterpret_cast<uint4*>(out + idx) = *reinterpret_cast<uint4*>(&o);
}
__global__ void geglu_bwd_kernel(const f16* __restrict__ dy, const f16* __restrict__ hg, f16* __restrict__ dhg, long M, int Hd, int blk) {
  const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (idx >= M * Hd) return;
  const long m = idx / Hd;
  int go;
  const long hc = m * 2 * Hd + geglu_hidden_col((int)(idx - m * Hd), Hd, blk, &go);
  const uint4 hr = *reinterpret_cast<const uint4*>(hg + hc), gr = *reinterpret_cast<const uint4*>(hg + hc + go);
  const uint4 dr = *reinterpret_cast<const uint4*>(dy + idx);
  const f16x8 hh = *reinterpret_cast<const f16x8*>(&hr), gh = *reinterpret_cast<const f16x8*>(&gr), dh = *reinterpret_cast<const f16x8*>(&dr);
  f16x8 oh, og;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g = (float)gh[e], d = (float)dh[e];
    oh[e] = (f16)(d * act_gelu(g));
    og[e] = (f16)(d * (float)hh[e] * act_grad(g, GN_ACT_GELU));
  }
  *reinterpret_cast<uint4*>(dhg + hc) = *reinterpret_cast<uint4*>(&oh);
  *reinterpret_cast<uint4*>(dhg + hc + go) = *reinterpret_cast<uint4*>(&og);
}

// ---- softmax backward (attention): ds = scale * p * (dp - sum_j p*dp), in place over dp ----------------------------------------
constexpr int SB_MAXCH = 8;
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const f16* __restrict__ p, f16* __restrict__ dp, long rows, int cols, long ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f16* pr = p + row * ld;
  f16* dr = dp + row * ld;
  const int CC = cols >> 3;
  float pv[SB_MAXCH][8], dv[SB_MAXCH][8];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < SB_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
      const uint4 a = *reinterpret_cast<const uint4*>(pr + cx * 8), b = *reinterpret_cast<const uint4*>(dr + cx * 8);
      const f16x8 ha = *reinterpret_cast<const f16x8*>(&a), hb = *reinterpret_cast<const f16x8*>(&b);
#pragma unroll
      for (int e = 0; e < 8; ++e) { pv[i][e] = (float)ha[e]; dv[i][e] = (float)hb[e]; dot += pv[i][e] * dv[i][e]; }
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < SB_MAXCH; ++i) {
    const int cx = lane + 64 * i;
    if (cx < CC) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(scale * pv[i][e] * (dv[i][e] - dot));
      *reinterpret_cast<uint4*>(dr + cx * 8) = *reinterpret_cast<uint4*>(&o);
    }
  }
}

// ---- LayerNorm backward: wave per row; per-block f32 partials of dgamma / dbeta ----------------------------------------------
// CH 16-byte chunks per lane (C <= 512 CH), R rows per wave and trip: the x and dy loads of all R rows are issued before the first
// reduction (a wave that walks its rows one at a time pays two dependent memory round trips per row: 44 us at 32768 x 320), and the
// three reductions of the R rows interleave.
constexpr int LNB_MAXCH = 4;  // C <= 2048
template <int CH, int R>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const f16* __restrict__ x, const f16* __restrict__ gamma, const f16* __restrict__ dy,
                                                            f16* dx, float* __restrict__ part, long M, int C, float eps,
                                                            int rows_per_block, const f16* dx_add) {
  // block = 4 waves; each wave walks rows_per_block/4 rows; lanes own fixed column chunks so dgamma/dbeta accumulate in registers
  // dx_add (optional, may alias dx): the gradient x already holds, added before the store
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int CC = C >> 3;
  float dg[CH][8], db[CH][8], gm[CH][8];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int cx = lane + 64 * i;
    uint4 g4 = make_uint4(0, 0, 0, 0);
    if (cx < CC) g4 = *reinterpret_cast<const uint4*>(gamma + cx * 8);
    const f16x8 hg = *reinterpret_cast<const f16x8*>(&g4);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; gm[i][e] = (float)hg[e]; }
  }
  const long r0 = (long)blockIdx.x * rows_per_block, rend = min(M, r0 + rows_per_block);
  const float invC = 1.0f / (float)C;
  for (long rb = r0 + (long)w * R; rb < rend; rb += 4 * R) {
    uint4 xa[R][CH], da[R][CH];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int cx = lane + 64 * i;
        const bool ok = cx < CC && rb + r < rend;
        xa[r][i] = ok ? *reinterpret_cast<const uint4*>(x + (rb + r) * C + cx * 8) : make_uint4(0, 0, 0, 0);
        da[r][i] = ok ? *reinterpret_cast<const uint4*>(dy + (rb + r) * C + cx * 8) : make_uint4(0, 0, 0, 0);
      }
    float s[R], ss[R], mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(&xa[r][i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[r] += (float)h[e];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) * invC;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ss[r] = 0.f;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(&xa[r][i]);
        if (lane + 64 * i < CC)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = (float)h[e] - mean[r]; ss[r] += d * d; }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(wave_sum(ss[r]) * invC + eps);
    float m1[R], m2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      m1[r] = 0.f; m2[r] = 0.f;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const f16x8 hx = *reinterpret_cast<const f16x8*>(&xa[r][i]), hd = *reinterpret_cast<const f16x8*>(&da[r][i]);
        if (lane + 64 * i < CC)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = ((float)hx[e] - mean[r]) * rstd[r], d = (float)hd[e];
            dg[i][e] += d * xh;
            db[i][e] += d;
            const float gv = d * gm[i][e];
            m1[r] += gv;
            m2[r] += gv * xh;
          }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { m1[r] = wave_sum(m1[r]) * invC; m2[r] = wave_sum(m2[r]) * invC; }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int cx = lane + 64 * i;
        if (cx < CC && rb + r < rend) {
          const f16x8 hx = *reinterpret_cast<const f16x8*>(&xa[r][i]), hd = *reinterpret_cast<const f16x8*>(&da[r][i]);
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = ((float)hx[e] - mean[r]) * rstd[r];
            o[e] = (f16)(rstd[r] * ((float)hd[e] * gm[i][e] - m1[r] - xh * m2[r]));
          }
          if (dx_add) {
            const uint4 ar = *reinterpret_cast<const uint4*>(dx_add + (rb + r) * C + cx * 8);
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&ar);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)((float)o[e] + (float)ah[e]);
          }
          *reinterpret_cast<uint4*>(dx + (rb + r) * C + cx * 8) = *reinterpret_cast<uint4*>(&o);
        }
      }
  }
  if (part) {  // per-wave partial rows: part[(block*4 + wave)][2][C]
    float* o = part + ((long)blockIdx.x * 4 + w) * 2 * C;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int cx = lane + 64 * i;
      if (cx < CC)
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[cx * 8 + e] = dg[i][e]; o[C + cx * 8 + e] = db[i][e]; }
    }
  }
}

// ---- GroupNorm(+SiLU) backward, NHWC: per-channel sums -> per-(b, c) affine coefficients -> apply ------------------------------
struct GNBParams {
  const f16* x; const f16* x2; const f16* dy; const f16* gamma; const f16* beta;
  f16* dx; f16* dx2;
  const f16* dx_add; const f16* dx2_add;  // optional: gradients already held by x / x2, added before the store (may alias dx / dx2)
  float* part;    // [B][chunks][2][C] per-channel partial sums of dyh and dyh*x
  float* coef;    // [B][C][3]: dx = a1*dyh + a2*x + a3
  float* sums;    // [B][2][C]: per-channel totals over the whole slab (S1 | S2)
  float* dgamma; float* dbeta;  // f32 [C] accumulated (nullable)
  int B, HW, C1, C2, C, G, cpg, chunks, rows, act;
  float eps;
};
__device__ __forceinline__ float gnb_dyh(float dy, float x, float a, float s, int act) {
  // y = act(x*a + s): gradient w.r.t. the normalised-affine value
  return act == GN_ACT_SILU ? dy * act_grad(x * a + s, GN_ACT_SILU) : dy;
}
__global__ __launch_bounds__(256) void gnb_partial_kernel(const GNBParams p, const float* __restrict__ scsh) {
  // per-channel sums of dyh and dyh*x over the slab's rows.  A thread owns one 8-channel group (16-byte loads of x and dy) and one
  // of TY row lanes; the row lanes are folded through LDS in a fixed order.
  __shared__ float red[256][17];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * p.rows, r1 = min(p.HW, r0 + p.rows);
  const int CC = p.C >> 3;
  const int TX = CC < 256 ? CC : 256, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float* o = p.part + (((long)b * p.chunks + chunk) * 2) * p.C;
  for (int cb = 0; cb < CC; cb += TX) {
    const int cc = cb + tx, c8 = cc * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (ty < TY && cc < CC) {
      const f16* src; int cs, co;
      if (c8 < p.C1) { src = p.x; cs = p.C1; co = c8; } else { src = p.x2; cs = p.C2; co = c8 - p.C1; }
      float a[8], s[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] = scsh[((long)b * p.C + c8 + e) * 2];
        s[e] = scsh[((long)b * p.C + c8 + e) * 2 + 1];
      }
      auto body = [&](const uint4& xr, const uint4& dr) {
        const f16x8 xh = *reinterpret_cast<const f16x8*>(&xr), dh = *reinterpret_cast<const f16x8*>(&dr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = (float)xh[e];
          const float d = gnb_dyh((float)dh[e], xv, a[e], s[e], p.act);
          s1[e] += d;
          s2[e] += d * xv;
        }
      };
      int r = r0 + ty;
      for (; r + 3 * TY < r1; r += 4 * TY) {  // 8 loads in flight per thread: a row at a time is latency-bound
        uint4 xr[4], dr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long pix = (long)b * p.HW + r + u * TY;
          xr[u] = *reinterpret_cast<const uint4*>(src + pix * cs + co);
          dr[u] = *reinterpret_cast<const uint4*>(p.dy + pix * p.C + c8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) body(xr[u], dr[u]);
      }
      for (; r < r1; r += TY) {
        const long pix = (long)b * p.HW + r;
        body(*reinterpret_cast<const uint4*>(src + pix * cs + co), *reinterpret_cast<const uint4*>(p.dy + pix * p.C + c8));
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[threadIdx.x][e] = s1[e];
      red[threadIdx.x][8 + e] = s2[e];
    }
    __syncthreads();
    if (ty == 0 && cc < CC) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float acc = 0.f;
        for (int j = 0; j < TY; ++j) acc += red[j * TX + tx][e];
        o[(e < 8 ? 0 : p.C - 8) + c8 + e] = acc;
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void gnb_finalize_kernel(const GNBParams p, const float* __restrict__ stats) {
  // stats[b][g] = (mean, rstd) from the forward.  Per channel: S1 = sum dyh, S2 = sum dyh*x.
  // per group: c2 = mean(dyh*gamma), c1 = mean(dyh*gamma*xhat);  dx = r*gamma*dyh - r*c2 - r*xhat*c1
  //          = (r*gamma) dyh + (-r^2 c1) x + (r^2 c1 mu - r c2)
  // One workgroup per (group, sample) -- a workgroup per sample walked C x chunks partials with 8 workgroups on the chip (24 us);
  // thread (channel ci of the group, chunk lane cl) sums chunks cl, cl + L, ... and the lanes fold in a fixed order.
  __shared__ float P1[256], P2[256], S1[256], S2[256], T[2];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int L = 256 / p.cpg;  // chunk lanes (cpg <= 256)
  const int ci = tid % p.cpg, cl = tid / p.cpg;
  const int c = g * p.cpg + ci;
  float a1 = 0.f, a2 = 0.f;
  if (cl < L) {
    const float* o = p.part + ((long)b * p.chunks * 2) * p.C + c;
    for (int ch = cl; ch < p.chunks; ch += L) {
      a1 += o[(long)(2 * ch) * p.C];
      a2 += o[(long)(2 * ch + 1) * p.C];
    }
  }
  P1[tid] = a1;
  P2[tid] = a2;
  __syncthreads();
  if (tid < p.cpg) {
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < L; ++q) { s1 += P1[q * p.cpg + tid]; s2 += P2[q * p.cpg + tid]; }
    S1[tid] = s1;
    S2[tid] = s2;
    p.sums[((long)b * 2) * p.C + c] = s1;       // kept for gnb_param_kernel
    p.sums[((long)b * 2 + 1) * p.C + c] = s2;
  }
  __syncthreads();
  const float mu = stats[((long)b * p.G + g) * 2], r = stats[((long)b * p.G + g) * 2 + 1];
  if (tid == 0) {
    float t1 = 0.f, t2 = 0.f;  // sum over the group's channels of gamma*(S2 - mu*S1)*r and gamma*S1
    for (int q = 0; q < p.cpg; ++q) {
      const float gm = (float)p.gamma[g * p.cpg + q];
      t2 += gm * S1[q];
      t1 += gm * (S2[q] - mu * S1[q]) * r;
    }
    T[0] = t1;
    T[1] = t2;
  }
  __syncthreads();
  if (tid < p.cpg) {
    const float n = (float)p.HW * (float)p.cpg;
    const float c1 = T[0] / n, c2 = T[1] / n;
    float* o = p.coef + ((long)b * p.C + c) * 3;
    o[0] = r * (float)p.gamma[c];
    o[1] = -r * r * c1;
    o[2] = r * r * c1 * mu - r * c2;
  }
}
__global__ void gnb_param_kernel(const GNBParams p, const float* __restrict__ stats) {
  // dgamma[c] += sum_b (S2 - mu*S1)*r ; dbeta[c] += sum_b S1   (fixed order over b and chunks)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.C) return;
  const int g = c / p.cpg;
  float dg = 0.f, db = 0.f;
  for (int b = 0; b < p.B; ++b) {
    const float s1 = p.sums[((long)b * 2) * p.C + c], s2 = p.sums[((long)b * 2 + 1) * p.C + c];  // from gnb_finalize_kernel
    const float mu = stats[((long)b * p.G + g) * 2], r = stats[((long)b * p.G + g) * 2 + 1];
    dg += (s2 - mu * s1) * r;
    db += s1;
  }
  p.dgamma[c] += dg;
  p.dbeta[c] += db;
}
__global__ __launch_bounds__(256) void gnb_apply_kernel(const GNBParams p, const float* __restrict__ scsh) {
  // one 8-channel group (16 bytes of x, dy, dx) per thread
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int CC = p.C >> 3;
  if (idx >= (long)p.HW * CC) return;
  const int r = (int)(idx / CC), c = (int)(idx - (long)r * CC) * 8;
  const long pix = (long)b * p.HW + r;
  const f16* src; f16* dst; const f16* add; int cs, co;
  if (c < p.C1) { src = p.x; dst = p.dx; add = p.dx_add; cs = p.C1; co = c; } else { src = p.x2; dst = p.dx2; add = p.dx2_add; cs = p.C2; co = c - p.C1; }
  if (!dst) return;
  const uint4 xr = *reinterpret_cast<const uint4*>(src + pix * cs + co);
  const uint4 dr = *reinterpret_cast<const uint4*>(p.dy + pix * p.C + c);
  const f16x8 xh = *reinterpret_cast<const f16x8*>(&xr), dh = *reinterpret_cast<const f16x8*>(&dr);
  const float* ss = scsh + ((long)b * p.C + c) * 2;
  const float* k = p.coef + ((long)b * p.C + c) * 3;
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xv = (float)xh[e];
    const float d = gnb_dyh((float)dh[e], xv, ss[2 * e], ss[2 * e + 1], p.act);
    o[e] = (f16)(k[3 * e] * d + k[3 * e + 1] * xv + k[3 * e + 2]);
  }
  if (add) {  // the f16 sum the separate add launch would have produced: f16(f16(dx) + g)
    const uint4 ar = *reinterpret_cast<const uint4*>(add + pix * cs + co);
    const f16x8 ah = *reinterpret_cast<const f16x8*>(&ar);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)o[e] + (float)ah[e]);
  }
  *reinterpret_cast<uint4*>(dst + pix * cs + co) = *reinterpret_cast<uint4*>(&o);
}

// ---- strided / upsampled conv dgrad helpers ------------------------------------------------------------------------------
__global__ void zero_upsample2x_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int B, int H, int W, int C8) {
  // out[b, 2y, 2x, :] = x[b, y, x, :], zeros elsewhere (out is [B, 2H, 2W, C])
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * 2 * H * 2 * W * C8;
  if (idx >= total) return;
  const int c = (int)(idx % C8);
  long r = idx / C8;
  const int ox = (int)(r % (2 * W)); r /= 2 * W;
  const int oy = (int)(r % (2 * H));
  const int b = (int)(r / (2 * H));
  uint4 v = make_uint4(0, 0, 0, 0);
  if (!(ox & 1) && !(oy & 1)) v = x[(((long)b * H + (oy >> 1)) * W + (ox >> 1)) * C8 + c];
  out[idx] = v;
}
__global__ void sumpool2x2_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int B, int H, int W, int C8) {
  // out[b, y, x, :] = sum of the 2x2 block of x [B, 2H, 2W, C]
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W * C8;
  if (idx >= total) return;
  const int c = (int)(idx % C8);
  long r = idx / C8;
  const int ox = (int)(r % W); r /= W;
  const int oy = (int)(r % H);
  const int b = (int)(r / H);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const uint4 raw = x[(((long)b * 2 * H + 2 * oy + dy) * 2 * W + 2 * ox + dx) * C8 + c];
      const f16x8 v = *reinterpret_cast<const f16x8*>(&raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)s[e];
  out[idx] = *reinterpret_cast<uint4*>(&o);
}

// ---- loss + optimizer ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_partial_kernel(const f16* __restrict__ pred, const f16* __restrict__ target, f16* __restrict__ dpred,
                                                          float* __restrict__ part, long pixels, int C, int ldp, int ldt, float gscale) {
  // loss = mean((pred - target)^2) over pixels*C (f32, diffusion/train_controlnet_genima.py:1400); dpred = gscale*(pred - target)
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pixels * ldp; i += (long)gridDim.x * 256) {
    const long px = i / ldp;
    const int c = (int)(i - px * ldp);
    float d = 0.f;
    if (c < C) { d = (float)pred[i] - (float)target[px * ldt + c]; s += d * d; }
    if (dpred) dpred[i] = (f16)(gscale * d);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_small_kernel(const float* __restrict__ part, float* __restrict__ out, int n, float scale) {
  // one wave, fixed order (deterministic): lane l adds part[l], part[l + 64], ... in double, then a fixed butterfly over the lanes
  // (a single thread walking 1 024 partials took 40 us of the optimizer's tail)
  if (blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x == 0) out[0] = (float)(s * (double)scale);
}
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long n) {
  // 16-byte loads, four independent partial sums per thread (the scalar form held 2.4 TB/s over the 1.46 GB flat gradient); the
  // summation order is a function of (n, grid) only: deterministic
  __shared__ float red[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long stride = (long)gridDim.x * 256;
  if ((((uintptr_t)x) & 15) == 0) {
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      const float4 a = x4[i], b = x4[i + stride];
      s0 += a.x * a.x + b.x * b.x; s1 += a.y * a.y + b.y * b.y; s2 += a.z * a.z + b.z * b.z; s3 += a.w * a.w + b.w * b.w;
    }
    if (i < n4) {
      const float4 a = x4[i];
      s0 += a.x * a.x; s1 += a.y * a.y; s2 += a.z * a.z; s3 += a.w * a.w;
    }
    for (long j = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) s0 += x[j] * x[j];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s0 += x[i] * x[i];
  }
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n, float lr,
                             float b1, float b2, float eps, float wd, float bc1, float bc2, const float* __restrict__ gscale_dev, float gscale,
                             f16* __restrict__ half_out, int zero_grad) {
  // torch.optim.AdamW (diffusion/train_controlnet_genima.py:1178-1185): decoupled weight decay, bias-corrected moments.
  // gscale_dev (optional, device scalar): the global-norm clip coefficient computed on the device (no host sync).
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // half_out (optional): the f16 working copy of the parameters, refreshed here instead of by a cast pass over the fp32 master;
  // zero_grad: the gradient is cleared in the pass that consumed it (optimizer.zero_grad(), also when the step is skipped)
  if (i >= n) return;
  const float graw = g[i];
  if (zero_grad) g[i] = 0.0f;
  if (gscale_dev && gscale_dev[2] != 0.0f) return;  // non-finite gradients: skip the step (torch.cuda.amp.GradScaler.step)
  const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.0f);
  const float gi = graw * gs;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  const float pn = pi - (lr / bc1) * (mi / denom);
  p[i] = pn;
  if (half_out) half_out[i] = (f16)pn;
}
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float* __restrict__ out, float max_norm, float inv_scale) {
  // out[0] = min(1, max_norm / (norm + 1e-6)), out[1] = norm of the unscaled gradients (torch.nn.utils.clip_grad_norm_ after
  // GradScaler.unscale_), out[2] = 1 when the gradients hold inf/nan (the optimizer step is then skipped).
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float norm = sqrtf(sumsq[0]) * inv_scale;
    const bool bad = !(norm == norm) || norm > 3.0e38f;
    out[1] = norm;
    out[2] = bad ? 1.0f : 0.0f;
    out[0] = bad ? 0.0f : fminf(1.0f, max_norm / (norm + 1e-6f));
  }
}
// latents = (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps) * scaling_factor  -- DiagonalGaussianDistribution.sample()
// (diffusion/train_controlnet_genima.py:1329-1332); moments [p, 2*C] = (mean | logvar), out [p, ld] zero-padded
__global__ void latent_sample_kernel(const f16* __restrict__ mom, const f16* __restrict__ eps, f16* __restrict__ out, long pixels, int C, int ld_mom,
                                     int ld_eps, int ld_out, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * ld_out) return;
  const long p = idx / ld_out;
  const int c = (int)(idx - p * ld_out);
  float v = 0.f;
  if (c < C) {
    const float mean = (float)mom[p * ld_mom + c];
    const float logvar = fminf(fmaxf((float)mom[p * ld_mom + C + c], -30.0f), 20.0f);
    v = (mean + __expf(0.5f * logvar) * (float)eps[p * ld_eps + c]) * scale;
  }
  out[idx] = (f16)v;
}
__global__ void cast_f32_f16_kernel(const float* __restrict__ x, f16* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (f16)x[i];
}
__global__ void fill_f32_kernel(float* __restrict__ x, long n, float v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

}  // namespace

// EMAModel.step (diffusers training_utils; diffusion/train_instruct_pix2pix_genima.py:1271-1272): shadow -= (1 - decay) * (shadow - param)
__global__ void ema_flat_kernel(float* __restrict__ shadow, const float* __restrict__ param, long n4, float one_minus_decay) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = reinterpret_cast<float4*>(shadow)[i];
  const float4 q = reinterpret_cast<const float4*>(param)[i];
  s.x = __fsub_rn(s.x, __fmul_rn(one_minus_decay, __fsub_rn(s.x, q.x)));
  s.y = __fsub_rn(s.y, __fmul_rn(one_minus_decay, __fsub_rn(s.y, q.y)));
  s.z = __fsub_rn(s.z, __fmul_rn(one_minus_decay, __fsub_rn(s.z, q.z)));
  s.w = __fsub_rn(s.w, __fmul_rn(one_minus_decay, __fsub_rn(s.w, q.w)));
  reinterpret_cast<float4*>(shadow)[i] = s;
}

extern "C" {

static int32_t transpose2d_launch(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, int32_t batch,
                                  int64_t in_bs, int64_t out_bs, int zrows) {
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
  const bool vec = cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && in_bs % 8 == 0 && out_bs % 8 == 0 && ld_out >= (rows + 7) / 8 * 8 &&
                   ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(transpose2d_vec_kernel<false>, grid, dim3(256), 0, ctx->stream, (const f16*)in, (f16*)out, rows, cols, (long)ld_in, (long)ld_out,
                       (long)in_bs, (long)out_bs, (float*)nullptr, 1, zrows);
  else
    hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(256), 0, ctx->stream, (const f16*)in, (f16*)out, rows, cols, (long)ld_in, (long)ld_out,
                       (long)in_bs, (long)out_bs, zrows);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_transpose2d(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, int32_t batch,
                       int64_t in_bs, int64_t out_bs) {
  GN_REQUIRE(ctx && in && out && rows > 0 && cols > 0 && batch > 0 && ld_in >= cols && ld_out >= rows, "gn_transpose2d: bad arguments");
  return transpose2d_launch(ctx, in, out, rows, cols, ld_in, ld_out, batch, in_bs, out_bs, rows);
}

/* gn_transpose2d that also writes the padding: columns [rows, ld_out) of every output row come out as zeros (a GEMM operand whose reduction
 * length is padded: the cross-attention V^T of 77 tokens in a 128-column matrix).  ld_out <= round_up(rows, 64): the padding lies inside the
 * last 64-row tile the launch walks anyway. */
int32_t gn_transpose2d_zpad(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, int32_t batch,
                            int64_t in_bs, int64_t out_bs) {
  GN_REQUIRE(ctx && in && out && rows > 0 && cols > 0 && batch > 0 && ld_in >= cols && ld_out >= rows, "gn_transpose2d_zpad: bad arguments");
  GN_REQUIRE(ld_out <= ((int64_t)rows + 63) / 64 * 64, "gn_transpose2d_zpad: ld_out (%ld) must not exceed round_up(rows = %d, 64)", (long)ld_out, rows);
  return transpose2d_launch(ctx, in, out, rows, cols, ld_in, ld_out, batch, in_bs, out_bs, (int)ld_out);
}

/* n_items gn_transpose2d problems in one launch (csrc comment at transpose2d_multi_kernel): `items` lives in DEVICE memory, every item as
 * gn_transpose2d's 16-byte-vector form requires (cols, ld_in, ld_out, in_bs, out_bs multiples of 8, 16-byte aligned pointers, ld_out >=
 * round_up(rows, 8)); block_begin = the running sum of batch * ceil(rows / 64) * ceil(cols / 64), total_blocks its end. */
extern "C" int32_t gn_transpose2d_multi(gn_ctx* ctx, const gn_transpose_item* items, int32_t n_items, int32_t total_blocks) {
  GN_REQUIRE(ctx && items && n_items > 0 && total_blocks > 0, "gn_transpose2d_multi: bad arguments");
  hipLaunchKernelGGL(transpose2d_multi_kernel, dim3(total_blocks), dim3(256), 0, ctx->stream, items, n_items);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

/* out = in^T as gn_transpose2d (one matrix), and sums[g][cols] += column sums of rows [g * rows / groups, (g + 1) * rows / groups):
 * the bias gradient (groups = 1) and / or the per-sample time-shift gradient (groups = batch) from the pass that transposes dY for
 * the weight gradient.  sums2 / groups2: an optional second grouping of the same partial sums.  rows / groups must be multiples of
 * 64; workspace = ceil(rows / 64) * cols floats. */
int32_t gn_transpose2d_colsum(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out,
                              float* sums, int32_t groups, float* sums2, int32_t groups2, void* workspace) {
  GN_REQUIRE(ctx && in && out && sums && workspace && rows > 0 && cols > 0 && groups > 0 && ld_in >= cols && ld_out >= rows, "gn_transpose2d_colsum: bad arguments");
  if (!sums2) groups2 = groups;
  GN_REQUIRE(groups2 > 0 && rows % groups == 0 && (rows / groups) % 64 == 0 && rows % groups2 == 0 && (rows / groups2) % 64 == 0,
             "gn_transpose2d_colsum: rows / groups (%d / %d, %d) must be multiples of 64", rows, groups, groups2);
  GN_REQUIRE(cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0,
             "gn_transpose2d_colsum: cols / strides must be multiples of 8, buffers 16-byte aligned");
  int rt = 1;  // 64-row tiles per block: keep the partial matrix at <= 128 rows where the row blocks allow
  const int t1 = rows / groups / 64, t2 = rows / groups2 / 64;
  while (t1 % (rt * 2) == 0 && t2 % (rt * 2) == 0 && (long)rows / 64 / rt > 128) rt *= 2;
  const dim3 grid((cols + 63) / 64, rows / 64 / rt, 1);
  hipLaunchKernelGGL(transpose2d_vec_kernel<true>, grid, dim3(256), 0, ctx->stream, (const f16*)in, (f16*)out, rows, cols, (long)ld_in, (long)ld_out,
                     0l, 0l, (float*)workspace, rt, rows);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3(nblk((long)groups * cols, 64)), dim3(256), 0, ctx->stream, (const float*)workspace, sums, groups,
                     t1 / rt, cols, 1);
  GN_LAUNCH_CHECK();
  if (sums2) {
    hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3(nblk((long)groups2 * cols, 64)), dim3(256), 0, ctx->stream, (const float*)workspace, sums2, groups2,
                       t2 / rt, cols, 1);
    GN_LAUNCH_CHECK();
  }
  return GN_OK;
}

int32_t gn_im2col_t(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ksize, int32_t stride,
                    int32_t pad) {
  GN_REQUIRE(ctx && x && out && B > 0 && H > 0 && W > 0 && C > 0 && ksize > 0 && stride > 0, "gn_im2col_t: bad arguments");
  GN_REQUIRE(C % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, "gn_im2col_t: C (%d) must be a multiple of 8, buffers 16-byte aligned", C);
  Im2colP p;
  p.x = (const f16*)x; p.out = (f16*)out; p.B = B; p.H = H; p.W = W; p.C = C; p.KH = ksize; p.KW = ksize; p.stride = stride; p.pad = pad;
  p.Ho = (H + 2 * pad - ksize) / stride + 1; p.Wo = (W + 2 * pad - ksize) / stride + 1;
  p.M = (long)B * p.Ho * p.Wo;
  GN_REQUIRE(p.M % 8 == 0, "gn_im2col_t: B*Ho*Wo (%ld) must be a multiple of 8", p.M);
  hipLaunchKernelGGL(im2col_t_kernel, dim3((C + 63) / 64, (unsigned)((p.M + 63) / 64), ksize * ksize), dim3(256), 0, ctx->stream, p);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

static inline int colsum_chunks(int nb, int rows_per_batch) {
  // enough row chunks to fill the chip (~1024 blocks) without shrinking a chunk below 64 rows
  int chunks = rows_per_batch / 64; if (chunks < 1) chunks = 1;
  const int want = (1024 + nb - 1) / nb;
  if (chunks > want) chunks = want;
  if (chunks > 128) chunks = 128;
  return chunks;
}
int64_t gn_colsum_workspace_bytes(int32_t nb, int32_t rows_per_batch, int32_t cols) {
  return (int64_t)nb * colsum_chunks(nb, rows_per_batch) * cols * 4;
}
int32_t gn_colsum_f32(gn_ctx* ctx, const void* x, float* out, int32_t nb, int32_t rows_per_batch, int32_t cols, int64_t ld, void* workspace,
                      int32_t accumulate) {
  GN_REQUIRE(ctx && x && out && workspace && nb > 0 && rows_per_batch > 0 && cols > 0 && ld >= cols, "gn_colsum_f32: bad arguments");
  GN_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0, "gn_colsum_f32: cols (%d) / ld must be multiples of 8, x 16-byte aligned", cols);
  const int chunks = colsum_chunks(nb, rows_per_batch);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 127) / 128, chunks, nb), dim3(256), 0, ctx->stream, (const f16*)x, (float*)workspace,
                     rows_per_batch, cols, (long)ld, chunks);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3(nblk((long)nb * cols, 64)), dim3(256), 0, ctx->stream, (const float*)workspace, out, nb, chunks, cols, accumulate);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_reduce_rows_f32(gn_ctx* ctx, const float* part, float* out, int32_t groups, int32_t R, int32_t cols, int32_t accumulate) {
  GN_REQUIRE(ctx && part && out && groups > 0 && R > 0 && cols > 0, "gn_reduce_rows_f32: bad arguments");
  hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3(nblk((long)groups * cols, 64)), dim3(256), 0, ctx->stream, part, out, groups, R, cols, accumulate);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_act_bwd(gn_ctx* ctx, const void* dy, const void* z, void* dz, int64_t n, int32_t act) {
  GN_REQUIRE(ctx && dy && z && dz && n > 0 && n % 8 == 0, "gn_act_bwd: n must be a positive multiple of 8");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(n / 8)), dim3(256), 0, ctx->stream, (const uint4*)dy, (const uint4*)z, (uint4*)dz, (long)(n / 8), act);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_geglu_fwd(gn_ctx* ctx, const void* hg, void* out, int64_t M, int32_t Hd, int32_t block) {
  GN_REQUIRE(ctx && hg && out && M > 0 && Hd > 0 && Hd % 8 == 0 && block >= 0 && block % 8 == 0 && (block == 0 || Hd % block == 0),
             "gn_geglu_fwd: Hd and block must be multiples of 8, block | Hd");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(nblk(M * Hd / 8)), dim3(256), 0, ctx->stream, (const f16*)hg, (f16*)out, (long)M, Hd, block);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
int32_t gn_geglu_bwd(gn_ctx* ctx, const void* dy, const void* hg, void* dhg, int64_t M, int32_t Hd, int32_t block) {
  GN_REQUIRE(ctx && dy && hg && dhg && M > 0 && Hd > 0 && Hd % 8 == 0 && block >= 0 && block % 8 == 0 && (block == 0 || Hd % block == 0),
             "gn_geglu_bwd: Hd and block must be multiples of 8, block | Hd");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(nblk(M * Hd / 8)), dim3(256), 0, ctx->stream, (const f16*)dy, (const f16*)hg, (f16*)dhg, (long)M, Hd, block);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_softmax_bwd(gn_ctx* ctx, const void* p, void* dp, int64_t rows, int32_t cols, int64_t ld, float scale) {
  GN_REQUIRE(ctx && p && dp && rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 64 * 8 * SB_MAXCH && ld % 8 == 0, "gn_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(nblk(rows, 4)), dim3(256), 0, ctx->stream, (const f16*)p, (f16*)dp, (long)rows, cols, (long)ld, scale);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

// rows per workgroup: 64 on long inputs, fewer on short ones so that >= 256 workgroups exist (M = 2048 gave 32 workgroups: 46 us)
static int lnb_rows_per_block(int64_t M) {
  int rpb = 64;
  while (rpb > 8 && (M + rpb - 1) / rpb < 256) rpb >>= 1;
  return rpb;
}
int64_t gn_layernorm_bwd_workspace_bytes(int64_t M, int32_t C) {
  const int rpb = lnb_rows_per_block(M);
  const int64_t blocks = (M + rpb - 1) / rpb;
  return blocks * 4 * 2 * C * 4;
}
int32_t gn_layernorm_bwd(gn_ctx* ctx, const void* x, const void* gamma, const void* dy, void* dx, float* dgamma, float* dbeta, void* workspace,
                         int64_t M, int32_t C, float eps, const void* dx_add) {
  GN_REQUIRE(ctx && x && gamma && dy && dx && M > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LNB_MAXCH, "gn_layernorm_bwd: C must be a multiple of 8, <= %d", 64 * 8 * LNB_MAXCH);
  GN_REQUIRE((dgamma == nullptr) == (dbeta == nullptr) && (!dgamma || workspace), "gn_layernorm_bwd: dgamma/dbeta come together and need a workspace");
  const int rpb = lnb_rows_per_block(M);
  const long blocks = (M + rpb - 1) / rpb;
  float* part = dgamma ? (float*)workspace : nullptr;
  const int CC = C / 8;
#define GN_LNB(CH, R) hipLaunchKernelGGL((layernorm_bwd_kernel<CH, R>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const f16*)x, \
                                         (const f16*)gamma, (const f16*)dy, (f16*)dx, part, (long)M, C, eps, rpb, (const f16*)dx_add)
  if (CC <= 64) GN_LNB(1, 4);
  else if (CC <= 128) GN_LNB(2, 2);
  else if (CC <= 192) GN_LNB(3, 1);
  else GN_LNB(4, 1);
#undef GN_LNB
  GN_LAUNCH_CHECK();
  if (dgamma) {
    // partial rows are [blocks*4][2][C]: view as R = blocks*4 rows of 2C columns -> [2C] sums
    GN_REQUIRE(dbeta == dgamma + C, "gn_layernorm_bwd: dbeta must follow dgamma contiguously (flat gradient buffer layout)");
    hipLaunchKernelGGL(reduce_rows_f32_kernel, dim3(nblk(2L * C, 64)), dim3(256), 0, ctx->stream, (const float*)workspace, dgamma, 1, (int)(blocks * 4), 2 * C, 1);
    GN_LAUNCH_CHECK();
  }
  return GN_OK;
}

int64_t gn_groupnorm_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t C) {
  int chunks = HW / 16; if (chunks < 1) chunks = 1; if (chunks > 64) chunks = 64;
  return ((int64_t)B * chunks * 2 * C + (int64_t)B * C * 3 + (int64_t)B * C * 2) * 4;
}
/* fwd_ws: the forward's workspace (gn_groupnorm_workspace_bytes) still holding scsh[B][C][2]; stats: [B][G][2] (mean, rstd) */
int32_t gn_groupnorm_bwd(gn_ctx* ctx, const gn_groupnorm_desc* d, const void* dy, void* dx, void* dx2, const float* scsh, const float* stats,
                         float* dgamma, float* dbeta, void* workspace, const void* dx_add, const void* dx2_add) {
  GN_REQUIRE(ctx && d && d->x && dy && scsh && stats && workspace && (dx || dx2), "gn_groupnorm_bwd: null pointer");
  GNBParams p;
  p.x = (const f16*)d->x; p.x2 = (const f16*)d->x2; p.dy = (const f16*)dy; p.gamma = (const f16*)d->gamma; p.beta = (const f16*)d->beta;
  p.dx = (f16*)dx; p.dx2 = (f16*)dx2; p.dgamma = dgamma; p.dbeta = dbeta;
  p.dx_add = (const f16*)dx_add; p.dx2_add = (const f16*)dx2_add;
  p.B = d->B; p.HW = d->HW; p.C1 = d->C1; p.C2 = d->C2; p.C = d->C1 + d->C2; p.G = d->groups; p.cpg = p.C / p.G; p.act = d->act; p.eps = d->eps;
  GN_REQUIRE(p.C <= 4096 && p.C % 8 == 0 && p.C1 % 8 == 0, "gn_groupnorm_bwd: C <= 4096, C and C1 multiples of 8");
  int chunks = p.HW / 16; if (chunks < 1) chunks = 1; if (chunks > 64) chunks = 64;  // small maps: 16-row slabs keep > 100 workgroups
  p.rows = (p.HW + chunks - 1) / chunks;
  p.chunks = (p.HW + p.rows - 1) / p.rows;
  p.part = (float*)workspace;
  p.coef = p.part + (long)p.B * chunks * 2 * p.C;
  p.sums = p.coef + (long)p.B * p.C * 3;
  hipLaunchKernelGGL(gnb_partial_kernel, dim3(p.chunks, p.B), dim3(256), 0, ctx->stream, p, scsh);
  GN_LAUNCH_CHECK();
  GN_REQUIRE(p.cpg <= 256, "gn_groupnorm_bwd: at most 256 channels per group");
  hipLaunchKernelGGL(gnb_finalize_kernel, dim3(p.G, p.B), dim3(256), 0, ctx->stream, p, stats);
  GN_LAUNCH_CHECK();
  if (dgamma) {
    hipLaunchKernelGGL(gnb_param_kernel, dim3(nblk(p.C)), dim3(256), 0, ctx->stream, p, stats);
    GN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(gnb_apply_kernel, dim3(nblk((long)p.HW * (p.C / 8)), p.B), dim3(256), 0, ctx->stream, p, scsh);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_zero_upsample2x(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C) {
  GN_REQUIRE(ctx && x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gn_zero_upsample2x: bad arguments");
  hipLaunchKernelGGL(zero_upsample2x_kernel, dim3(nblk((long)B * 4 * H * W * (C / 8))), dim3(256), 0, ctx->stream, (const uint4*)x, (uint4*)out, B, H, W, C / 8);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
int32_t gn_sumpool2x2(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C) {
  GN_REQUIRE(ctx && x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gn_sumpool2x2: bad arguments (H, W are the OUTPUT size)");
  hipLaunchKernelGGL(sumpool2x2_kernel, dim3(nblk((long)B * H * W * (C / 8))), dim3(256), 0, ctx->stream, (const uint4*)x, (uint4*)out, B, H, W, C / 8);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

/* loss_out[0] = mean((pred[:, :C] - target)^2) in f32; dpred = grad_scale * 2/(pixels*C) * (pred - target) (0 in padded channels) */
int32_t gn_mse_loss(gn_ctx* ctx, const void* pred, const void* target, void* dpred, float* loss_out, void* workspace, int64_t pixels, int32_t C,
                    int32_t ld_pred, int32_t ld_target, float grad_scale) {
  GN_REQUIRE(ctx && pred && target && loss_out && workspace && pixels > 0 && C > 0 && ld_pred >= C && ld_target >= C, "gn_mse_loss: bad arguments");
  const int blocks = 256;
  const float gs = grad_scale * 2.0f / (float)((double)pixels * C);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const f16*)pred, (const f16*)target, (f16*)dpred, (float*)workspace,
                     (long)pixels, C, ld_pred, ld_target, gs);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float*)workspace, loss_out, blocks, 1.0f / (float)((double)pixels * C));
  GN_LAUNCH_CHECK();
  return GN_OK;
}

/* out[0] = sum(x^2) (deterministic two-stage); workspace >= 2048 floats */
int32_t gn_sumsq_f32(gn_ctx* ctx, const float* x, int64_t n, float* out, void* workspace) {
  GN_REQUIRE(ctx && x && out && workspace && n > 0, "gn_sumsq_f32: bad arguments");
  const int blocks = 2048;  // 8 workgroups per CU: enough loads in flight to hold the HBM rate
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, ctx->stream, x, (float*)workspace, (long)n);
  GN_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float*)workspace, out, blocks, 1.0f);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

/* norm = sqrt(sumsq[0]) * inv_scale;  clip[0] = min(1, max_norm / (norm + 1e-6)), clip[1] = norm, clip[2] = 1 if non-finite */
int32_t gn_clip_coef(gn_ctx* ctx, const float* sumsq, float* clip, float max_norm, float inv_scale) {
  GN_REQUIRE(ctx && sumsq && clip && max_norm > 0.f && inv_scale > 0.f, "gn_clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, sumsq, clip, max_norm, inv_scale);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

/* fused AdamW over flat f32 buffers; step >= 1; grad is multiplied by grad_scale * (clip_dev ? clip_dev[0] : 1) */
int32_t gn_adamw_flat(gn_ctx* ctx, float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, const float* clip_dev, float grad_scale, void* half_out, int32_t zero_grad) {
  GN_REQUIRE(ctx && param && grad && m && v && n > 0 && step >= 1, "gn_adamw_flat: bad arguments");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, param, grad, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                     clip_dev, grad_scale, (f16*)half_out, zero_grad);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

int32_t gn_latent_sample(gn_ctx* ctx, const void* moments, const void* eps, void* out, int64_t pixels, int32_t C, int32_t ld_moments, int32_t ld_eps,
                         int32_t ld_out, float scale) {
  GN_REQUIRE(ctx && moments && eps && out && pixels > 0 && C > 0 && ld_moments >= 2 * C && ld_eps >= C && ld_out >= C, "gn_latent_sample: bad arguments");
  hipLaunchKernelGGL(latent_sample_kernel, dim3(nblk(pixels * ld_out)), dim3(256), 0, ctx->stream, (const f16*)moments, (const f16*)eps, (f16*)out,
                     (long)pixels, C, ld_moments, ld_eps, ld_out, scale);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
int32_t gn_ema_flat(gn_ctx* ctx, float* shadow, const float* param, int64_t n, float one_minus_decay) {
  GN_REQUIRE(ctx && shadow && param && n > 0 && n % 4 == 0, "gn_ema_flat: n must be a positive multiple of 4");
  hipLaunchKernelGGL(ema_flat_kernel, dim3(nblk(n / 4)), dim3(256), 0, ctx->stream, shadow, param, (long)(n / 4), one_minus_decay);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
int32_t gn_cast_f32_f16(gn_ctx* ctx, const float* x, void* out, int64_t n) {
  GN_REQUIRE(ctx && x && out && n > 0, "gn_cast_f32_f16: bad arguments");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, x, (f16*)out, (long)n);
  GN_LAUNCH_CHECK();
  return GN_OK;
}
int32_t gn_fill_f32(gn_ctx* ctx, float* x, int64_t n, float v) {
  GN_REQUIRE(ctx && x && n > 0, "gn_fill_f32: bad arguments");
  hipLaunchKernelGGL(fill_f32_kernel, dim3(nblk(n)), dim3(256), 0, ctx->stream, x, (long)n, v);
  GN_LAUNCH_CHECK();
  return GN_OK;
}

}  // extern "C"
