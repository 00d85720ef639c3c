/*
 * genima_hip.h -- C ABI of libgenima_hip.so: the MI355X (gfx950 / CDNA4) kernels under the Genima hot path.
 *
 * The reference (MohitShridhar/genima) has no FFI of its own: its device arithmetic is reached through
 * Python seams.  This header is the boundary the build puts *underneath* those seams; each entry
 * point cites the reference call site whose third-party device op it replaces.  Conventions (binding):
 *   - extern "C", plain C types only; no C++ exceptions cross the boundary.
 *   - every function returns int32_t status: 0 = ok, negative = error (gn_last_error() has the thread-local message).
 *   - the caller owns ALL device memory (weights, activations, workspaces) and passes raw device pointers + shapes;
 *     the library never allocates device memory.  *_workspace_bytes() queries precede ops that need scratch.
 *   - functions enqueue on the gn_ctx's stream and never synchronise (asynchronous w.r.t. the host).
 *   - activations are NHWC ("[B, H*W, C]" row-major == the token layout of the transformer blocks); f16 storage with
 *     f32 accumulation (the reference runs torch_dtype=float16: controller/agent/sd_controlnet_agent.py:34,40).
 *   - weights are packed [Cout][KH*KW*Cin] (tap-major, channel-minor; K contiguous) by the host (genima_amd/packing.py).
 */
#ifndef GENIMA_HIP_H
#define GENIMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN_OK 0
#define GN_ERR_INVALID (-1)      /* bad argument / unsupported shape */
#define GN_ERR_HIP (-2)          /* a HIP runtime call failed */
#define GN_ERR_UNSUPPORTED (-3)

/* epilogue activations */
enum { GN_ACT_NONE = 0, GN_ACT_SILU = 1, GN_ACT_GELU = 2, GN_ACT_QUICK_GELU = 3, GN_ACT_RELU = 4, GN_ACT_GEGLU = 5,
       GN_ACT_TANH3 = 6 /* 3*tanh(x/3): AutoencoderTiny's latent clamp (controller/agent/sd_controlnet_agent.py:45-49); gn_act only */ };
/* output modes of the GEMM epilogue */
enum { GN_OUT_ROWMAJOR = 0, GN_OUT_BATCH_TRANSPOSED = 1, GN_OUT_F32 = 2 /* row-major float output, ldo in floats */ };

typedef struct gn_ctx gn_ctx;
typedef struct gn_program gn_program;

/* ---- context ------------------------------------------------------------------------------------------------ */
int32_t gn_version(void);
const char* gn_last_error(void);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the default stream. */
int32_t gn_ctx_create(int32_t device, void* stream, gn_ctx** out);
int32_t gn_ctx_destroy(gn_ctx* ctx);
int32_t gn_ctx_set_stream(gn_ctx* ctx, void* stream);

/* ---- K1/K3/K6/K7/K8: MFMA implicit-GEMM convolution and Linear (one kernel family) --------------------------------
 * out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )            M = B*Ho*Wo (conv) or rows (linear), K = KH*KW*(C1+C2)
 * Replaces the cuDNN conv / cuBLAS GEMM calls reached from `self.pipe(...)` (controller/agent/sd_controlnet_agent.py:67-76)
 * and `controlnet(...)` / `unet(...)` / `vae.encode` / `text_encoder` (diffusion/train_controlnet_genima.py:1324-1388).
 * Epilogue order:  v = acc + bias[n] + shift[m / rows_per_batch, n];  v = act(v);  v *= out_scale;  v += residual[m, n].
 * GN_ACT_GEGLU: W rows are packed in alternating 32-row blocks [hidden | gate]; out[m, j] = hidden * gelu(gate), out has N/2 cols.
 */
/* ---- the GroupNorm bridge (round 5): statistics out of the PRODUCER, GroupNorm-apply (+ SiLU) inside the CONSUMER ---------------------------
 * diffusers ResnetBlock2D runs  conv -> GroupNorm -> SiLU -> conv  and Transformer2DModel  conv -> GroupNorm -> proj_in  (inside `self.pipe(...)`,
 * controller/agent/sd_controlnet_agent.py:67-76): as separate launches every GroupNorm is a dependent pass over the tensor between two GEMMs.
 * Here the op that WRITES the tensor also adds its per-(sample, group) sum and sum of squares -- of the f16 values it stored -- into a
 * caller-zeroed statistics block (gn_stats_sink), and the op that READS it normalises its A operand on the way into the MFMA (gn_norm_in), or a
 * single coalesced apply pass does (gn_groupnorm_desc.stats_in).  Sums are fixed point (sum * 2^GN_STATS_SHIFT, sum of squares * 2^GN_STATS_SHIFT_SQ,
 * int64, device-scope integer atomics): integer addition is order-independent, so the statistics -- and everything downstream -- are
 * bit-reproducible run to run.  The second word's coarser scale is range: a slab's sum of squares may reach 2.2e15 before the word wraps. */
#define GN_STATS_SHIFT 24
#define GN_STATS_SHIFT_SQ 12
/* The statistics block: int64 [replicas][samples][groups][GN_STATS_LINE] -- one 128-byte line per (replica, sample, group) holding
 * (sum * 2^GN_STATS_SHIFT, sum of squares * 2^GN_STATS_SHIFT_SQ) in its first two words, ZEROED by the caller before the producers run.  Device-scope atomics
 * serialise per memory line (measured on MI355X: ~100 ns each; 2 240 adds onto the 8 lines of a packed 1 x 32 x 2 block cost a 28 us conv
 * another 30 us), hence a line per group, and `replicas` > 1 where few samples leave few lines: a producer workgroup adds into replica
 * (its row-tile index % replicas), the consumer sums the replicas (integer adds: any order, same bits). */
#define GN_STATS_LINE 16
typedef struct gn_stats_sink {
  void* stats;             /* the statistics block; NULL = off */
  int32_t cpg;             /* channels per group of the consuming GroupNorm */
  int32_t coff;            /* channel, in the consumer's (possibly concatenated) input, of this tensor's channel 0 */
  int32_t groups;          /* groups of the consuming GroupNorm */
  int32_t rows_per_sample; /* output rows (pixels) per sample */
  int32_t samples;         /* samples (the block's second extent) */
  int32_t replicas;        /* >= 1 */
} gn_stats_sink;
typedef struct gn_norm_in {
  const void* stats;       /* the statistics block the producers of the input filled (complete when this op starts); NULL = off */
  const void* gamma;       /* f16 [C] over the (concatenated) input channels */
  const void* beta;
  float eps;
  int32_t groups, cpg;
  int32_t act;             /* GN_ACT_NONE or GN_ACT_SILU applied after the affine */
  int32_t rows_per_sample; /* INPUT pixels (conv) / rows (dense) per sample */
  int32_t samples, replicas; /* extents of the statistics block */
} gn_norm_in;

/* GroupNorm fused into a split-K launch's REDUCE (round 5): the small-M convs of the 8x8 .. 32x32 latent levels split K and finish in a reduce
 * kernel anyway; with norm_out set that kernel owns one (sample, group) slab per workgroup -- it sums the partial slabs, applies the epilogue,
 * stores the raw f16 tensor to `out` as usual AND GroupNorm(+ SiLU) of it to `y`: diffusers ResnetBlock2D's conv1 -> norm2 -> SiLU and
 * conv2 -> next block's norm1 / Transformer2DModel.norm without the GroupNorm launch.  Row-major f16 output, rows_per_batch rows per sample,
 * N % groups == 0, an even group width, a slab (rows_per_batch x N / groups f16) of at most 96 KB; ignored (error) when the plan does not split K
 * (gn_gemm_workspace_bytes() == 0). */
typedef struct gn_norm_out {
  void* y;                 /* [M, N] f16 normalised output (row stride N); NULL = off */
  const void* gamma;       /* f16 [N] */
  const void* beta;
  float eps;
  int32_t groups;
  int32_t act;             /* GN_ACT_NONE or GN_ACT_SILU */
  int32_t rows_per_sample;
} gn_norm_out;

typedef struct gn_gemm_desc {
  const void* a;          /* A source 1: dense [M, lda] or NHWC [B, H, W, C1] */
  const void* a2;         /* conv only: second NHWC source [B, H, W, C2], virtually concatenated on channels; or NULL */
  const void* w;          /* [N, ldw] f16, K contiguous */
  const void* bias;       /* [N] f16 or NULL */
  const void* shift;      /* [M / rows_per_batch, ldshift] f16 or NULL (time-embedding shift, K3) */
  const void* residual;   /* [M, ldr] f16 or NULL */
  void* out;              /* f16 */
  void* workspace;        /* f32 split-K scratch (gn_gemm_workspace_bytes) or NULL when splitk <= 1 */
  int64_t M, N, K;
  int64_t lda, ldw, ldr, ldo;
  int64_t ldshift;        /* row stride of shift (0 = N) */
  int32_t conv;           /* 0 = dense A, 1 = implicit-GEMM gather */
  int32_t B, H, W, C1, C2;         /* conv: source tensor dims (before the optional nearest-2x upsample) */
  int32_t KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int32_t upsample2x;     /* conv: 1 = input is nearest-upsampled 2x on the fly (K8) */
  int32_t act;            /* GN_ACT_* */
  int32_t out_mode;       /* GN_OUT_*; BATCH_TRANSPOSED: out[b][n][m - b*rows_per_batch], row stride ldo, batch stride N*ldo */
  int32_t rows_per_batch; /* for shift / transposed output; 0 = M */
  int32_t splitk;         /* 0 = library heuristic, >=1 explicit */
  int32_t tile;           /* 0 = library heuristic; 1..6 = {256x128, 128x128, 128x64, 64x64, 256x64, 128x256} register-staged block tile,
                             7..14 = {256x256, 256x128, 128x128, 128x64, 64x64, 256x64, 128x320, 256x320} LDS-DMA block tile,
                             15 = 256x256 ping-pong (8-phase, counted vmcnt; K % 64 == 0, conv C1/C2 % 64 == 0, no GEGLU / batch),
                             16..22 = {128x128, 128x64, 64x64, 256x64, 128x160, 64x160, 64x320} with a 3-stage LDS-DMA ring (two K tiles
                             in flight, counted vmcnt); 20..22 are the exact-fit tiles of the N = 640 / 1280 / 320 launches,
                             23 = 128x160 two-stage LDS-DMA, 24 = 128x320 two-stage LDS-DMA on eight waves of 32x160,
                             25 = PERSISTENT skewed ping-pong 256x256 (round 6, csrc/gemm_ppp.hip): one workgroup per CU walks the tile list, the next
                             tile's LDS ring is requested before the finished tile's epilogue, tile boundaries are skewed over the chip, the last
                             partial round is split along K; needs >= one 256x256 tile per CU, a row-major f16 output with N, ldo % 8 == 0, no
                             split-K / out2 / ln_c1 / GEGLU / shift + residual together (else tile 15 runs), and a workspace of
                             gn_gemm_workspace_bytes() for the f32 hand-off slabs of the tiles two workgroups share
                             (the host autotunes this per shape: genima_amd/engine.py) */
  int32_t residual_before_act; /* 1: v = act(acc + bias + shift + residual) (ResNet basic block); 0: residual added last */
  float out_scale;        /* 1.0f = none */
  /* batched GEMM (attention backward, per-sample products): `batch` independent problems of the same shape; problem z uses
   * operand + (z / batch_inner) * *_bs + (z % batch_inner) * *_bs2 (strides in elements of that operand; batch_inner >= 1).
   * batch <= 1 = a single problem.  Split-K is disabled when batch > 1. */
  int32_t batch, batch_inner;
  int64_t a_bs, a_bs2, w_bs, w_bs2, out_bs, out_bs2, res_bs, res_bs2;
  int32_t accumulate;     /* GN_OUT_F32 only: out += result (gradient accumulation) */
  /* fp8 Linear (SURVEY section 8 a15 / BASELINE configs[4] "fp8 MFMA"; the reference side is the fp16 autocast Linear of
   * diffusion/train_controlnet_sdxl_genima.py:1448-1471): 1 = a and w hold OCP e4m3 bytes (K contiguous; K, lda, ldw count bytes
   * and are multiples of 16), out = epilogue(scale_a[m] * scale_w[n] * sum_k a*w) on v_mfma_scale_f32_32x32x64_f8f6f4.
   * gn_quantize_fp8_rows produces the bytes and the per-row scales of both operands.  Dense row-major only. */
  int32_t fp8;
  const void* scale_a;    /* fp8: f32 [M] */
  const void* scale_w;    /* fp8: f32 [N], 16-byte aligned */
  /* two-destination output: the q | k | v projections of a self-attention block as ONE launch (diffusers Attention.to_q / to_k /
   * to_v, three cuBLAS calls in the reference: SURVEY.md K7).  Columns [0, split_n) go to `out` (row-major, ldo) as usual; columns
   * [split_n, N) go to `out2` batch-transposed: out2[b][n - split_n][m - b * rows_per_batch], row stride ldo2 -- the V^T layout
   * gn_attention_fwd reads.  split_n % 32 == 0; GN_OUT_ROWMAJOR, no GEGLU / batch / fp8; K is never split.  NULL = off. */
  void* out2;
  int64_t ldo2;
  int32_t split_n;
  /* LayerNorm folded into the Linear that consumes it (diffusers BasicTransformerBlock.norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q,
   * norm3 -> ff.net[0].proj; CLIPEncoderLayer.layer_norm1/2 -> q/k/v, fc1: a LayerNorm kernel + a cuBLAS call each in the reference).
   * `a` holds the RAW rows; `w` the gamma-scaled weight W'[n, k] = W[n, k] * gamma[k] (f16); `ln_c1` f32 [N] = sum_k W'[n, k] over the
   * f16-rounded W'; `bias` holds c2[n] = sum_k W[n, k] * beta[k] + b[n].  The kernel takes each row's mean / rstd (biased variance,
   * ln_eps) from the A fragments of its K loop and emits rstd * (a . W'^T - mean * c1) + c2 through the usual epilogue (activation,
   * GEGLU, residual, out2 ...).  Dense f16 Linears on the LDS-DMA tiles (7..14, 16..23); K is never split.  NULL = off. */
  float ln_eps;
  const float* ln_c1;
  /* two-level output row pitch (row-major f16 outputs): row m is written at (m / out_row_width) * ldo_hi + (m % out_row_width) * ldo
   * elements from `out`.  One PHASE of a nearest-2x upsampling conv (diffusers Upsample2D: F.interpolate(scale 2, "nearest") + a 3x3
   * conv -- the UNet's up_blocks.*.upsamplers.0 and the VAE decoder's) is a 2x2 conv over the SOURCE pixels whose
   * results land on every other pixel of every other row of the output: out_row_width = W, ldo = 2 * C, ldo_hi = 4 * W * C,
   * out = base + (dy * 2 * W + dx) * C.  0 = off (row m at m * ldo). */
  int32_t out_row_width;
  int64_t ldo_hi;
  /* 1: the FOUR phase convs of an Upsample2D as one launch (batch = 4, conv, KH = KW = 2): w = [4][N][ldw] (phase 2 dy + dx at w + z * w_bs),
   * pad_t / pad_l / out / ldo / ldo_hi / out_row_width describe phase (0, 0); phase (dy, dx) uses padding 1 - dy / 1 - dx and writes
   * ldo_hi / 2 * dy + ldo / 2 * dx elements further. */
  int32_t up_phases;
  /* 1: a 1x1 conv of a SECOND tensor appended along K -- out = conv_KHxKW(a; w[:, :KH*KW*C1]) + conv_1x1(a2; w[:, KH*KW*C1:]) + bias:
   * diffusers ResnetBlock2D's  conv2(h) + conv_shortcut(x)  (the blocks whose channel count changes: every up block of the UNet reads a
   * concatenation) as ONE launch instead of a 1x1 launch, its output's round trip and a residual read.  a2 = [B, H, W, C2] (C2 channels, same
   * pixels as the output), K = KH*KW*C1 + C2, w = [N][K] with the 1x1 weight behind the packed KHxKW weight, bias = the two biases' sum.
   * A block whose input is a concatenation (the up blocks: cat(hidden, skip)) appends BOTH tensors: a2 (C2 channels, C2 % 64 == 0) then a3
   * (C3 channels), K = KH*KW*C1 + C2 + C3.  Stride 1, same-size output, C1 % 64 == 0, no fused upsample; LDS-DMA tiles (7..23; the others are
   * mapped onto them).
   * Dense problems (conv == 0): out = [A | A2] . W^T -- the last C2 of the K columns come from a2 (row stride lda2), K - C2 a multiple of 64.
   * That is how TWO Linears without a nonlinearity between them run as one: Transformer2DModel's  proj_out(ff.net.2(g) + h) + x  is
   * [W_po W_ff2 | W_po] . [g ; h] + (W_po b_ff2 + b_po) + x  (packing `ffo_pout`: the product in fp32, one rounding). */
  int32_t k_append;
  const void* a3;         /* k_append (conv): optional second appended source [B, H, W, C3] or NULL */
  int32_t C3;
  int64_t lda2;           /* k_append (dense): row stride of a2 */
  /* GroupNorm bridge, producer side: add the statistics of the f16 values this op stores (row-major f16 outputs; not GEGLU / out2 / batch) */
  gn_stats_sink sink;
  /* GroupNorm bridge, consumer side: a (and a2 of a virtual concat) hold the RAW tensor; each A tile is normalised (x * scale + shift, SiLU)
   * in LDS after it lands, taps in the zero padding stay zero, an appended k_append segment stays raw.  Ring tiles 16 .. 22, conv or dense,
   * C1 (and C2) % 64 == 0; a row tile may span at most 4 samples (gn_gemm_norm_in_supported). */
  gn_norm_in norm_in;
  gn_norm_out norm_out;   /* GroupNorm of the output inside the split-K reduce (above) */
} gn_gemm_desc;
int32_t gn_gemm_norm_in_supported(const gn_gemm_desc* d);
int32_t gn_gemm_norm_out_supported(const gn_gemm_desc* d); /* the problem AND its plan (tile / splitk as set in d) take norm_out */
int64_t gn_gemm_workspace_bytes(const gn_gemm_desc* d);
/* 1 when the plan named in d (tile / splitk) carries the fusions attached to d -- norm_out needs a plan that splits K, norm_in a ring tile whose
 * rows span at most 4 samples -- i.e. when gn_gemm would not refuse the launch for its plan; 0 otherwise (gn_last_error says why).
 * gn_program_set_gemm_plan asks here before it patches a recorded op. */
int32_t gn_gemm_plan_valid(const gn_gemm_desc* d);
#define GN_NUM_GEMM_TILES 25
/* tile 25's in-launch hand-offs wait with a bound; -> how many waits have given up on the current device since the library was loaded (0 in a
 * healthy process; tests assert it). */
int64_t gn_ppp_timeouts(void);
/* probe aid (a library built with -DGN_PPP_PROFILE): per-workgroup cycle sums of the last tile-25 launch, 8 words per workgroup; zeros otherwise */
int32_t gn_ppp_profile_read(uint32_t* out, int32_t words);
/* tuning hook: force tile configuration 0..3 = {256x128, 128x128, 128x64, 64x64} for every following gn_gemm; -1 = heuristic */
int32_t gn_set_gemm_tile_override(int32_t cfg);
int32_t gn_gemm(gn_ctx* ctx, const gn_gemm_desc* d);
/* fp8 operand preparation for gn_gemm_desc.fp8: q[r, k] = e4m3_rne(x[r, k] * 448 / amax_r), scales[r] = amax_r / 448 (1 for an
 * all-zero row); x f16 [rows, ldx], q bytes [rows, ldq] with ldq % 16 == 0 and >= round_up(K, 16) (the pad bytes are written as
 * zeros), scales f32 [rows].  Activations: rows = tokens; weights [N, K]: rows = output channels (done once per weight). */
int32_t gn_quantize_fp8_rows(gn_ctx* ctx, const void* x, int64_t ldx, int64_t rows, int32_t K, void* q, int64_t ldq, void* scales);

/* ---- fused chains of a BasicTransformerBlock's Linears (csrc/tblock.hip) --------------------------------------------------------
 * One workgroup keeps 128 rows of the [M, C] residual stream in LDS for a whole chain of GEMMs and streams the chain's weights through an
 * LDS ring from a "weight tape" (one contiguous buffer holding the LDS image of every 20 KB weight slot in consumption order,
 * genima_amd/packing.py pack_tblock_tape; gn_tblock_tape_bytes() long).  Replaces, inside the diffusers transformer blocks that
 * `self.pipe(...)` runs (controller/agent/sd_controlnet_agent.py:67-76; graphs.emit_transformer), the gn_gemm launches
 *   GN_TBLOCK_FRONT: out = (a * scale[b] + shift[b]) Wi^T + bi   (Transformer2DModel.norm applied from its statistics-only pass + proj_in)
 *                   out2 = LayerNorm1(out) [Wq | Wk]^T  row-major [M, 2C];  out3 = (LayerNorm1(out) Wv^T)^T per sample: V^T [B][C][ldo3]
 *                                                         (norm1 folded into attn1.to_q / to_k / to_v; the attention kernels' operands)
 *   GN_TBLOCK_MID : out  = a Wo^T + bo + res1            (attn1.to_out.0 + residual)
 *                   out2 = LayerNorm2(out) Wq^T           (norm2 folded into attn2.to_q, as gn_gemm_desc.ln_c1)
 *   GN_TBLOCK_TAIL: h2   = a Wo^T + bo + res1            (attn2.to_out.0 + residual)
 *                   h3   = GEGLU(LayerNorm3(h2) W1^T + b1) W2^T + b2 + h2   (norm3 folded into ff.net.0.proj; ff.net.2 + residual)
 *                   out  = h3 Wp^T + bp + res2            (proj_out + the transformer's input)
 * h2 / h3 and the [M, 4C] GEGLU intermediate never leave the chip; every intermediate is rounded to f16 where the separate launches round
 * it.  Built for C = 320 (the 64x64-latent level), M % 128 == 0; gn_tblock_supported() says whether a problem qualifies.
 * All tensors f16 row-major, rows 16-byte aligned. */
enum { GN_TBLOCK_MID = 1, GN_TBLOCK_TAIL = 2, GN_TBLOCK_FRONT = 3 };
typedef struct gn_tblock_desc {
  int32_t kind;           /* GN_TBLOCK_* */
  int32_t C;              /* channels (320) */
  int64_t M;              /* rows (tokens) */
  const void* a;          /* [M, lda]: the attention output the first Linear consumes */
  const void* res1;       /* [M, ldr1]: residual of the first Linear (MID / TAIL) */
  const void* res2;       /* TAIL: [M, ldr2] residual of proj_out (the transformer's input); MID: NULL */
  void* out;              /* [M, ldo]: MID: the residual stream after attn1; TAIL: the transformer's output */
  void* out2;             /* MID: [M, ldo2] cross-attention queries; TAIL: NULL */
  const void* tape;       /* the chain's weight tape */
  int64_t tape_bytes;
  int64_t lda, ldr1, ldr2, ldo, ldo2;
  float ln_eps;
  const void* scsh;       /* FRONT: f32 [B][C][2] GroupNorm (scale, shift) pairs (gn_groupnorm_fwd with y == NULL) */
  void* out3;             /* FRONT: V^T [B][C][ldo3] */
  int64_t ldo3;
  int32_t rows_per_batch; /* FRONT: tokens per sample (a multiple of 128) */
} gn_tblock_desc;
int64_t gn_tblock_tape_bytes(int32_t kind, int32_t C);          /* 0 = not built for this kind / width */
int32_t gn_tblock_supported(int32_t kind, int64_t M, int32_t C); /* 1 / 0 */
int32_t gn_tblock(gn_ctx* ctx, const gn_tblock_desc* d);
/* The chain's weight tape from the packed f16 tensors (what genima_amd/packing.py pack_tblock_*_tape does with torch ops, bit for bit): every
 * 20 KB slot is the LDS image the kernel copies verbatim -- [rows x 32] sub-tiles with 64-byte rows whose 16-byte chunks are XOR-swizzled by
 * (row >> 2) & 3 -- in consumption order, the bias / c1 / c2 vectors behind the last slot.  Run once per checkpoint load.
 *   w_a, b_a : the chain's first Linear [C, C], [C]     (FRONT proj_in; MID attn1.to_out.0; TAIL attn2.to_out.0)
 *   w_ln, c1, c2 : its LayerNorm-folded Linear (gn_pack_fold_layernorm): FRONT attn1 q | k | v [3C, C]; MID attn2.to_q [C, C];
 *                  TAIL ff.net.0.proj [8C, C] in the packed GEGLU row order (gn_pack_geglu_rows); c1 f32, c2 f16
 *   w2, b2 : TAIL ff.net.2 [C, 4C], [C];   w_p, b_p : TAIL proj_out [C, C], [C] */
typedef struct gn_tblock_tape_src {
  int32_t kind, C;
  const void* w_a; const void* b_a;
  const void* w_ln; const float* c1; const void* c2;
  const void* w2; const void* b2;
  const void* w_p; const void* b_p;
} gn_tblock_tape_src;
int32_t gn_pack_tblock_tape(gn_ctx* ctx, const gn_tblock_tape_src* src, void* tape, int64_t tape_bytes);

/* ---- K4/K5/K11: flash-style attention forward --------------------------------------------------------------------
 * o[b, i, h*D + :] = softmax_j(scale * q[b,i,h] . k[b,j,h]) v[b,j,h]; V is consumed TRANSPOSED (vt[b][h*D + d][j], produced
 * for free by the V projection's GN_OUT_BATCH_TRANSPOSED epilogue).  Replaces xformers memory_efficient_attention / torch SDPA
 * (controller/agent/diffusion_agent.py:35-36, diffusion/train_controlnet_genima.py:1125-1126).  D in {32, 64}.
 * vt row stride must be a multiple of 8 and cover round_up(Nk, 64) columns (pad columns must hold finite values). */
typedef struct gn_attn_desc {
  const void* q; const void* k; const void* vt; void* o;
  int64_t q_bs, k_bs, vt_bs, o_bs;   /* batch strides (elements) */
  int32_t q_rs, k_rs, vt_rs, o_rs;   /* row strides (elements) */
  int32_t B, heads, Nq, Nk, D;
  int32_t causal;
  float scale;
  float* lse;                        /* optional f32 [B][heads][Nq]: log2-domain log-sum-exp of the scaled scores, kept for
                                        gn_attention_bwd (training); NULL in inference */
  int32_t v_rowmajor;                /* 1 (D = 64): `vt` is V itself, row-major [B][Nk][vt_rs] with head h at column h*D (what a plain
                                        q | k | v projection writes); the kernel transposes out of its LDS tile (ds_read_b64_tr_b16) */
} gn_attn_desc;
int32_t gn_attention_fwd(gn_ctx* ctx, const gn_attn_desc* d);
/* Tuning / test aid: pick the D = 64 forward kernel for every later gn_attention_fwd (and recorded attention op) of this process.
 * -1 = the library's own choice (default; the GN_ATTN_VARIANT environment variable, read once, sets the initial value),
 * 0 = attention.hip (4 waves x 32 rows), 4 = attention_stream.hip wherever eligible, 5 = attention_pwg.hip wherever eligible
 * (non-causal, V^T given, Nk % 64 == 0, Nk >= 128).  Returns the previous value. */
int32_t gn_attention_set_variant(int32_t variant);

/* fp8 (OCP e4m3) attention forward, D = 64 -- the opt-in attention of the fp8 training forward (BASELINE configs[4] "fp8 MFMA";
 * xformers attention under diffusion/train_controlnet_sdxl_genima.py:1448-1471).  Both products run on the K = 64 fp8 MFMA; the
 * probabilities are e4m3 too, so the result sits ~1e-2 (relative, per element) from the f16 kernel: never used by the inference path.
 * gn_attention_fp8_quantize makes the operands from the f16 q | k | v rows ([B][N][*_rs] views, head h at column 64 h):
 *   q8 [B][N][heads*64] = e4m3(q * scale * log2 e), k8 [B][N][heads*64] = e4m3(k)  (bytes, saturating at +-448),
 *   v8t [B][heads*64][Npad] = e4m3(v) transposed, the keys of every 64-key tile in the MFMA operand order (csrc/attention_fp8.hip);
 *   Npad = a multiple of 64 >= N; keys >= N are written as zeros.
 * gn_attention_fp8_fwd takes a gn_attn_desc whose q / k / vt are those byte tensors (strides in BYTES; `scale` is ignored: it is
 * already in q8; v_rowmajor must be 0), o f16 and the optional lse as for gn_attention_fwd (same log2-domain value, so
 * gn_attention_bwd can run on the f16 q / k / v the operands were made from). */
int32_t gn_attention_fp8_quantize(gn_ctx* ctx, const void* q, const void* k, const void* v, int64_t q_rs, int64_t k_rs, int64_t v_rs,
                                  int64_t q_bs, int64_t k_bs, int64_t v_bs, int32_t B, int32_t N, int32_t heads, float scale,
                                  void* q8, void* k8, void* v8t, int32_t Npad);
int32_t gn_attention_fp8_fwd(gn_ctx* ctx, const gn_attn_desc* d);

/* Flash-attention backward (D = 64; xformers memory-efficient attention backward under accelerator.backward,
 * diffusion/train_controlnet_genima.py:1125-1126, :1402).  P is recomputed per tile from q, k and the forward's lse; two
 * deterministic kernels (dQ over key tiles; dK, dV over query tiles), no atomics.  q / k / v / o / d_o and the gradients are
 * row-major [B][rows][ld] with head h at column h*D of the given base pointer; qt / kt / dot (transposed copies of Q, K, dO that
 * round 1's kernels streamed) are IGNORED and may be NULL: the kernels read the transposed operand out of the row-major LDS tile
 * with ds_read_b64_tr_b16.  Rows Nk .. Nk_rows-1 of k / v must be zero padding (their dk / dv rows are
 * written as zeros up to round_up(Nk, 128), rows past that are left untouched); Nq and Nk_rows are multiples of 8.  delta: f32 [B][heads][Nq] scratch (sum_d dO*O, written here). */
typedef struct gn_attn_bwd_desc {
  const void* q; const void* k; const void* v; const void* o; const void* d_o;
  const void* qt; const void* kt; const void* dot;
  const float* lse; float* delta;
  void* dq; void* dk; void* dv;
  int64_t q_bs, k_bs, v_bs, o_bs, do_bs, qt_bs, kt_bs, dot_bs, dq_bs, dk_bs, dv_bs;   /* batch strides (elements) */
  int32_t q_rs, k_rs, v_rs, o_rs, do_rs, qt_rs, kt_rs, dot_rs, dq_rs, dk_rs, dv_rs;   /* row strides (elements) */
  int32_t B, heads, Nq, Nk, Nk_rows, D;
  float scale;
} gn_attn_bwd_desc;
int32_t gn_attention_bwd(gn_ctx* ctx, const gn_attn_bwd_desc* d);

/* ---- K2: GroupNorm(+SiLU), NHWC --------------------------------------------------------------------------------
 * y = act(GroupNorm_G(cat(x, x2)) * gamma + beta).  Three launches: coalesced partial statistics over pixel slabs, a finalize
 * that folds (mean, rstd, gamma, beta) into per-(b, c) scale/shift, and a coalesced apply.  Replaces torch
 * native_group_norm + silu (every ResnetBlock2D / Transformer2DModel entry / conv_norm_out). */
typedef struct gn_groupnorm_desc {
  const void* x; const void* x2;     /* NHWC f16; x2 optional concat source */
  const void* gamma; const void* beta; /* [C1+C2] f16 */
  void* y;                           /* [B, HW, C1+C2] f16 */
  void* workspace;                   /* gn_groupnorm_workspace_bytes() */
  int32_t B, HW, C1, C2, groups;
  int32_t act;                       /* GN_ACT_NONE or GN_ACT_SILU */
  float eps;
  void* save_stats;                  /* training: optional f32 [B][groups][2] (mean, rstd) kept for gn_groupnorm_bwd */
  void* save_scsh;                   /* training: optional f32 [B][C][2] per-(b, c) scale/shift kept for gn_groupnorm_bwd */
  const void* stats_in;              /* GroupNorm bridge: the statistics block the producers of x / x2 filled (gn_stats_sink; samples = B) --
                                        ONE coalesced apply launch, no statistics passes; NULL = compute them here */
  int32_t stats_replicas;            /* replicas of stats_in (>= 1) */
} gn_groupnorm_desc;
int64_t gn_groupnorm_workspace_bytes(const gn_groupnorm_desc* d);
int32_t gn_groupnorm_fwd(gn_ctx* ctx, const gn_groupnorm_desc* d);

/* gn_groupnorm_fwd with y == NULL and save_scsh set: STATISTICS ONLY -- the per-(sample, channel) scale / shift pairs are written and
 * nothing is normalised; gn_conv3x3_gn applies them (and the SiLU) to its input patch in LDS.
 *
 * ---- 3x3 convolution (stride 1, padding 1) with GroupNorm-apply + SiLU fused into its prologue (csrc/conv_gn.hip) ------------------------
 * out = conv3x3(act(x * scale[b, c] + shift[b, c])) + bias (+ residual): diffusers ResnetBlock2D's norm1 -> SiLU -> conv1 and
 * norm2 -> SiLU -> conv2 (+ shortcut) without the normalised tensor's round trip through HBM (the VAE decoder inside `self.pipe(...)`,
 * controller/agent/sd_controlnet_agent.py:67-76).  A workgroup keeps the 10 x 18 input patch of an 8 x 16 output tile in LDS, normalises it
 * there (the arithmetic of gn_groupnorm_fwd's apply pass: the MFMA sees the same f16 values) and reads all nine taps from it.
 * Needs H % 8 == 0, W % 16 == 0, Cin % 128 == 0, and Cout % 128 == 0 or -- the narrow variant, no residual: the VAE's conv_norm_out -> SiLU ->
 * conv_out with its 3 (padded 8) output channels -- Cout in {8, 16, 24, 32} (gn_conv3x3_gn_supported). */
typedef struct gn_conv3x3_gn_desc {
  const void* x;          /* NHWC f16 [B, H, W, Cin]: the RAW tensor the GroupNorm reads */
  const void* scsh;       /* f32 [B][Cin][2] (scale, shift) from the statistics-only GroupNorm call, or NULL: plain convolution */
  const void* w;          /* packed conv weight [Cout][9 * Cin] f16 (gn_pack_conv_weight) */
  const void* bias;       /* [Cout] f16 or NULL */
  const void* residual;   /* [B * H * W, ldr] f16 or NULL: added after the bias */
  void* out;              /* [B * H * W, ldo] f16 */
  int64_t ldr, ldo;
  int32_t B, H, W, Cin, Cout;
  int32_t act;            /* GN_ACT_NONE / GN_ACT_SILU on the normalised input (with scsh) */
} gn_conv3x3_gn_desc;
int32_t gn_conv3x3_gn_supported(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int32_t gn_conv3x3_gn(gn_ctx* ctx, const gn_conv3x3_gn_desc* d);

/* ---- K7: LayerNorm over the last dim of [M, C] (C % 8 == 0, C <= 4096) --------------------------------------------- */
int32_t gn_layernorm_fwd(gn_ctx* ctx, const void* x, const void* gamma, const void* beta, void* y,
                         int64_t M, int32_t C, float eps);

/* ---- K9: sinusoidal timestep embedding -> f16 [B, dim]  (t: device f32 [B]) ------------------------------------- */
int32_t gn_timestep_embedding(gn_ctx* ctx, const float* t, void* out, int32_t B, int32_t dim, int32_t flip_sin_to_cos,
                              float freq_shift);

/* ---- K10: scheduler elementwise ops on f16 latents [n] ------------------------------------------------------------
 * gn_scale_pad:  out[p, 0:C] = x[p, 0:C] * scale, out[p, C:Cpad] = 0       (scale_model_input / latents/scaling_factor,
 *                and the channel padding the implicit-GEMM conv wants for Cin = 4)
 * gn_euler_step: x <- x + eps * (sigma_next - sigma), f32 math, f16 storage; eps has row stride ld_eps
 * gn_add_noise:  out = a[b] * x0 + c[b] * noise  (DDPM add_noise, diffusion/train_controlnet_genima.py:1359) */
int32_t gn_scale_pad(gn_ctx* ctx, const void* x, void* out, int64_t pixels, int32_t C, int32_t Cpad, float scale);
int32_t gn_euler_step(gn_ctx* ctx, void* x, const void* eps, int64_t pixels, int32_t C, int32_t ld_eps, float sigma,
                      float sigma_next);
/* InstructPix2Pix channel concatenation (diffusers StableDiffusionInstructPix2PixPipeline: cat([scale_model_input(latents), image_latents],
 * dim=1); diffusion/train_instruct_pix2pix_genima.py:1236-1239): out[p, 0:C] = x[p*ld1 + 0:C] * scale, out[p, C:C+C2] = x2[p*ld2 + 0:C2] * scale2,
 * out[p, C+C2:Cpad] = 0 */
int32_t gn_scale_cat_pad(gn_ctx* ctx, const void* x, const void* x2, void* out, int64_t pixels, int32_t C, int32_t ld1, int32_t C2,
                         int32_t ld2, int32_t Cpad, float scale, float scale2);
int32_t gn_add_noise(gn_ctx* ctx, const void* x0, const void* noise, const float* sqrt_ac, const float* sqrt_1mac,
                     void* out, int32_t B, int64_t per_sample);

/* ---- K14: image pre/post-processing ------------------------------------------------------------------------------
 * gn_image_u8_to_f16: uint8 NHWC [pixels, 3] -> f16 [pixels, Cpad]: v/255*mul + add  (VaeImageProcessor.preprocess)
 * gn_image_f16_to_u8: f16 [pixels, ld] -> uint8 [pixels, 3]: round(clamp(v/2+0.5, 0, 1)*255)  (postprocess, Appendix D.6) */
int32_t gn_image_u8_to_f16(gn_ctx* ctx, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float mul, float add);
int32_t gn_image_f16_to_u8(gn_ctx* ctx, const void* in, uint8_t* out, int64_t pixels, int32_t ld);
/* ACT image path (controller/method/genima_act.py:146-148, :188): uint8 [pixels, 3] -> f16 [pixels, Cpad]: v * m_c + a_c
 * (m_c = 1/(255 std_c), a_c = -mean_c/std_c; channels >= 3 zero) */
int32_t gn_image_normalize_u8(gn_ctx* ctx, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float m0, float m1,
                              float m2, float a0, float a1, float a2);
/* ---- weight repacking (explicit gn_pack_* calls into caller-owned buffers; genima_amd/packing.py is the Python
 * host's torch restatement of the same layouts, bit-identical on the two copies: tests/test_kernels_gpu.py) -------------------------
 * diffusers Conv2d.weight OIHW (f32, or f16 when src_f16) -> dst f16 [round_up(O, 8)][KH * KW * round_up(I, 8)] with
 * dst[o][(kh * KW + kw) * Ip + c] = src[o][c][kh][kw], zero padding: the W operand of gn_gemm(conv) */
int32_t gn_pack_conv_weight(gn_ctx* ctx, const void* src_oihw, int32_t src_f16, void* dst, int32_t O, int32_t I, int32_t KH, int32_t KW);
/* diffusers GEGLU.proj weight [2H, K] (or its bias: K = 1), hidden rows first then gate rows -> alternating 32-row blocks
 * [hidden 0..31 | gate 0..31 | hidden 32..63 | ...], the order GN_ACT_GEGLU's epilogue pairs up.  H % 32 == 0 */
int32_t gn_pack_geglu_rows(gn_ctx* ctx, const void* src, int32_t src_f16, void* dst, int32_t H, int64_t K);
/* operands of gn_gemm_desc::ln_c1 from a packed f16 Linear weight w [N, K] (row stride ldw) and its LayerNorm's gamma / beta [K]
 * (+ the Linear's bias [N] or NULL): ln_weight = f16(w * gamma) [N, K] (row stride ldw), ln_c1[n] = sum_k ln_weight[n, k] (f32),
 * ln_c2[n] = f16(sum_k w[n, k] * beta[k] + bias[n]) */
int32_t gn_pack_fold_layernorm(gn_ctx* ctx, const void* w, const void* gamma, const void* beta, const void* bias, void* ln_weight,
                               float* ln_c1, void* ln_c2, int32_t N, int32_t K, int64_t ldw);
/* out[b, :] = x[b, idx[b], :]  (CLIP EOT-token pooling, controller/method/genima_act.py:337-343) */
int32_t gn_gather_rows(gn_ctx* ctx, const void* x, const int32_t* idx, void* out, int32_t B, int32_t L, int32_t D);
/* out[i] = index of the first maximum of row i of an int32 [rows, cols] matrix (EOT = highest token id) */
int32_t gn_argmax_rows_i32(gn_ctx* ctx, const int32_t* x, int32_t* out, int32_t rows, int32_t cols);
/* strided 4-D copy of contiguous f16 runs of L elements (L % 8 == 0; strides in elements, multiples of 8): the layout shuffles
 * between kernels (ACT: per-view feature maps -> views-along-width token rows of the encoder sequence) */
int32_t gn_copy4d(gn_ctx* ctx, const void* in, void* out, const int64_t* sizes, const int64_t* in_strides,
                  const int64_t* out_strides, int32_t L);

/* ---- misc elementwise / gather ------------------------------------------------------------------------------------ */
int32_t gn_add(gn_ctx* ctx, const void* a, const void* b, void* out, int64_t n);              /* f16, n % 8 == 0 */
/* count <= GN_ADD_MULTI_MAX independent out[i] = a[i] + b[i] (n[i] % 8 == 0) as ONE launch: UNet2DConditionModel.forward's
 * `down_block_res_samples + down_block_additional_residuals` / `sample + mid_block_additional_residual` additions (thirteen torch adds
 * inside `self.pipe(...)`, controller/agent/sd_controlnet_agent.py:67-76) */
#define GN_ADD_MULTI_MAX 16
int32_t gn_add_multi(gn_ctx* ctx, const void* const* a, const void* const* b, void* const* out, const int64_t* n, int32_t count);
/* the same adds with the GroupNorm bridge's producer side: tensor i is [n[i] / C[i]] rows of C[i] channels and adds the statistics of what it
 * stores to sinks[i] (sinks[i].stats == NULL: none) -- the UNet's skip + ControlNet-residual sums feed the decoder's concatenated GroupNorms */
int32_t gn_add_multi_stats(gn_ctx* ctx, const void* const* a, const void* const* b, void* const* out, const int64_t* n, const int32_t* C,
                           const gn_stats_sink* sinks, int32_t count);
int32_t gn_act(gn_ctx* ctx, const void* x, void* out, int64_t n, int32_t act);                /* f16, n % 8 == 0 */
/* FiLM (controller/method/genima_act.py:190: ``encoder_model(image, task_emb)`` with use_lang_cond, genima_act.yaml:39):
 * out[r, :] = act((1 + gamma[r / rows_per_film, :]) * x[r, :] + beta[r / rows_per_film, :]); x / out f16 [rows, C], gamma / beta f16 rows of
 * stride ld_film (slices of the per-layer FiLM feature buffer); out may alias x */
int32_t gn_film(gn_ctx* ctx, const void* x, void* out, const void* gamma, const void* beta, int64_t ld_film, int64_t rows_per_film,
                int64_t rows, int32_t C, int32_t act);
int32_t gn_embedding(gn_ctx* ctx, const int32_t* ids, const void* tok, const void* pos, void* out, int32_t B,
                     int32_t L, int32_t D);                                                   /* CLIP token + position */
int32_t gn_softmax_rows(gn_ctx* ctx, void* x, int64_t rows, int32_t cols, int32_t ld, float scale); /* in place, f16 */
/* same with key padding: columns >= valid get probability 0 (cols stays a multiple of 8) */
int32_t gn_softmax_rows_masked(gn_ctx* ctx, void* x, int64_t rows, int32_t cols, int32_t ld, float scale, int32_t valid);
int32_t gn_maxpool3x3s2(gn_ctx* ctx, const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C);

/* ---- training-side kernels (ControlNet fine-tune step, diffusion/train_controlnet_genima.py:1317-1408) ----------
 * The backward matrix products reuse gn_gemm: dX = dY.W through a transposed weight copy, dW = dY^T.X through transposed
 * activations with GN_OUT_F32 + split-K, conv dgrad = conv with rotated weights, conv wgrad = GEMM over gn_im2col_t's image.
 * Parameter gradients are f32 and accumulate; all reductions are deterministic. */
int32_t gn_transpose2d(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out,
                       int32_t batch, int64_t in_bs, int64_t out_bs);                 /* out[b][c][r] = in[b][r][c] (f16) */
/* the same, and columns [rows, ld_out) of every output row are written as zeros (ld_out <= round_up(rows, 64)): the padded reduction length
 * of a GEMM operand -- the cross-attention V^T of 77 tokens -- without a fill launch in front of the transpose */
int32_t gn_transpose2d_zpad(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out,
                            int32_t batch, int64_t in_bs, int64_t out_bs);
/* out[(tap*C + c)][m] = x[b, oy*stride - pad + dy, ox*stride - pad + dx, c] (0 in the padding); M = B*Ho*Wo contiguous */
int32_t gn_im2col_t(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ksize,
                    int32_t stride, int32_t pad);
/* ---- weight gradient on operands in their forward layout (no transposed copies, no im2col^T) ---------------------------------------
 * dw[n, k] += sum_r dy[r, n] * X[r, k];  dense: X = x [R, ld_x];  conv: X[r, tap * C + c] = x[b, oy*stride - pad + dy, ox*stride - pad + dx, c]
 * for r = (b, oy, ox), zero outside the image (x NHWC [B, H, W, C], C % 8 == 0).  What autograd computes for nn.Linear / nn.Conv2d weights
 * inside accelerator.backward(loss) (diffusion/train_controlnet_genima.py:1391).  f32 result accumulated into dw; the reduction over the
 * R rows is split across workgroups deterministically (workspace: gn_wgrad_workspace_bytes). */
typedef struct gn_wgrad_desc {
  const void* dy;        /* f16 [R, ld_dy] */
  const void* x;         /* f16 [R, ld_x] or NHWC [B, H, W, C] */
  float* dw;             /* f32 [N, ld_dw], accumulated */
  void* workspace;
  int64_t R, N, K;       /* K = columns of dw (conv: KH*KW*C) */
  int64_t ld_dy, ld_x, ld_dw;
  int32_t conv, B, H, W, C, KH, KW, stride, pad, Ho, Wo;
  int32_t tile;          /* 0 = heuristic, 1 = 128x128, 2 = 64x64, 3 = 128x256, 4 = 256x128 on eight waves (N / K at least a tile; else 128x128) */
  int32_t splitk;        /* 0 = heuristic */
  float* dbias;          /* optional f32 [N] += column sums of dy (the bias gradient), from the fragments the kernel loads anyway */
  float* dshift;         /* optional f32 [shift_groups, N] += column sums per block of R / shift_groups rows (per-sample time-shift gradient) */
  int32_t shift_groups;  /* R / shift_groups must be a multiple of 64 */
} gn_wgrad_desc;
int64_t gn_wgrad_workspace_bytes(const gn_wgrad_desc* d);
int32_t gn_wgrad(gn_ctx* ctx, const gn_wgrad_desc* d);

/* Many gn_transpose2d problems in ONE launch (the trainable network's derived weight copies -- W^T for the Linears' data gradients,
 * the tap-rotated conv weights -- are rebuilt after every optimizer step; accelerate / autograd get them for free from cuBLAS's
 * transposed operand forms, diffusion/train_controlnet_genima.py:1391).  `items`: DEVICE memory, each as the 16-byte-vector form of
 * gn_transpose2d requires; block_begin = running sum of batch * ceil(rows/64) * ceil(cols/64), total_blocks = its end. */
typedef struct gn_transpose_item {
  const void* in; void* out;
  int64_t ld_in, ld_out, in_bs, out_bs;
  int32_t rows, cols, batch, block_begin;
} gn_transpose_item;
int32_t gn_transpose2d_multi(gn_ctx* ctx, const gn_transpose_item* items, int32_t n_items, int32_t total_blocks);

/* gn_transpose2d of one matrix that also yields sums[g][cols] += the column sums of row block g (groups = 1: bias gradient;
 * groups = batch: per-sample time-shift gradient; sums2 / groups2: an optional second grouping from the same pass) -- the
 * weight-gradient path transposes dY anyway.  (rows / groups) % 64 == 0; workspace: ceil(rows / 64) * cols floats.  Replaces the
 * `.sum(0)` of autograd's Linear / conv bias backward. */
int32_t gn_transpose2d_colsum(gn_ctx* ctx, const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out,
                              float* sums, int32_t groups, float* sums2, int32_t groups2, void* workspace);
int64_t gn_colsum_workspace_bytes(int32_t nb, int32_t rows_per_batch, int32_t cols);
/* out[b][c] (+)= sum_r x[b*rows_per_batch + r][c]  (bias gradients nb = 1; time-shift gradients nb = batch) */
int32_t gn_colsum_f32(gn_ctx* ctx, const void* x, float* out, int32_t nb, int32_t rows_per_batch, int32_t cols, int64_t ld,
                      void* workspace, int32_t accumulate);
int32_t gn_reduce_rows_f32(gn_ctx* ctx, const float* part, float* out, int32_t groups, int32_t R, int32_t cols, int32_t accumulate);
int32_t gn_act_bwd(gn_ctx* ctx, const void* dy, const void* z, void* dz, int64_t n, int32_t act);   /* dz = dy * act'(z) */
/* unfused GEGLU out = hidden * gelu(gate); hg [M, 2*Hd]: block == 0 -> [hidden | gate] halves, block > 0 -> alternating
 * block-column groups [hidden | gate] (the packed ff.net.0.proj layout of GN_ACT_GEGLU, block = 32) */
int32_t gn_geglu_fwd(gn_ctx* ctx, const void* hg, void* out, int64_t M, int32_t Hd, int32_t block);
int32_t gn_geglu_bwd(gn_ctx* ctx, const void* dy, const void* hg, void* dhg, int64_t M, int32_t Hd, int32_t block);
/* attention backward softmax step, in place over dp: ds = scale * p * (dp - rowsum(p * dp)) */
int32_t gn_softmax_bwd(gn_ctx* ctx, const void* p, void* dp, int64_t rows, int32_t cols, int64_t ld, float scale);
int64_t gn_layernorm_bwd_workspace_bytes(int64_t M, int32_t C);
/* dx from (x, gamma, dy); dgamma/dbeta (f32, dbeta == dgamma + C, accumulated) optional */
/* dx_add (optional, may alias dx): the gradient x already holds from another branch; dx = f16(f16(dx) + dx_add) */
int32_t gn_layernorm_bwd(gn_ctx* ctx, const void* x, const void* gamma, const void* dy, void* dx, float* dgamma, float* dbeta,
                         void* workspace, int64_t M, int32_t C, float eps, const void* dx_add);
int64_t gn_groupnorm_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t C);
/* backward of gn_groupnorm_fwd(d) run with save_stats/save_scsh; dx / dx2 follow d->x / d->x2 (either may be NULL) */
int32_t gn_groupnorm_bwd(gn_ctx* ctx, const gn_groupnorm_desc* d, const void* dy, void* dx, void* dx2, const float* scsh,
                         const float* stats, float* dgamma, float* dbeta, void* workspace, const void* dx_add, const void* dx2_add);
int32_t gn_zero_upsample2x(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C); /* stride-2 dgrad */
int32_t gn_sumpool2x2(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C);       /* upsample dgrad */
int32_t gn_mse_loss(gn_ctx* ctx, const void* pred, const void* target, void* dpred, float* loss_out, void* workspace,
                    int64_t pixels, int32_t C, int32_t ld_pred, int32_t ld_target, float grad_scale);
int32_t gn_sumsq_f32(gn_ctx* ctx, const float* x, int64_t n, float* out, void* workspace);
/* global-norm clipping after loss-scale removal (accelerator.clip_grad_norm_, diffusion/train_controlnet_genima.py:1403-1405):
 * norm = sqrt(sumsq[0]) * inv_scale; clip[0] = min(1, max_norm / (norm + 1e-6)); clip[1] = norm; clip[2] = 1 when non-finite
 * (gn_adamw_flat given this clip buffer then leaves the parameters untouched, as GradScaler.step does) */
int32_t gn_clip_coef(gn_ctx* ctx, const float* sumsq, float* clip, float max_norm, float inv_scale);
/* half_out (optional): f16 [n] working copy of the parameters, refreshed in the same pass; zero_grad: grad is cleared in the same pass
 * (optimizer.zero_grad(), also when the step is skipped) */
int32_t gn_adamw_flat(gn_ctx* ctx, float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t step, const float* clip_dev, float grad_scale,
                      void* half_out, int32_t zero_grad);
/* VAE posterior sample of the train step (diffusion/train_controlnet_genima.py:1329-1332): moments [p, ld_moments] = (mean | logvar),
 * out[p, 0:C] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps) * scale, out[p, C:ld_out] = 0 */
int32_t gn_latent_sample(gn_ctx* ctx, const void* moments, const void* eps, void* out, int64_t pixels, int32_t C,
                         int32_t ld_moments, int32_t ld_eps, int32_t ld_out, float scale);
/* EMAModel.step of the InstructPix2Pix fine-tune (diffusion/train_instruct_pix2pix_genima.py:821-824, :1271-1272) on the flat fp32
 * buffers: shadow <- shadow - one_minus_decay * (shadow - param), f32 op order of diffusers' `s_param.sub_(one_minus_decay * (s_param - param))` */
int32_t gn_ema_flat(gn_ctx* ctx, float* shadow, const float* param, int64_t n, float one_minus_decay);
int32_t gn_cast_f32_f16(gn_ctx* ctx, const float* x, void* out, int64_t n);
int32_t gn_fill_f32(gn_ctx* ctx, float* x, int64_t n, float v);

/* ---- train-time augmentation on the device (diffusion/train_controlnet_genima.py:775-830, README `--augmentations=crop,colorjitter`) --
 * gn_color_jitter: torchvision ColorJitter's four adjust_* ops on [B, HW, ld] f16 RGB pixels in [0, 1], applied in `order`
 * (op ids 0 brightness, 1 contrast, 2 saturation, 3 hue = ColorJitter.get_params' fn_idx) with `factors[id]`; f32 colour math,
 * adjust_contrast's per-image grey mean by a deterministic two-stage sum.  out may alias x.
 * gn_reflect_pad_crop: F.pad(mode="reflect", pad on all sides) + crop of the original H x W at (crop_i, crop_j). */
int64_t gn_color_jitter_workspace_bytes(int32_t B);
int32_t gn_color_jitter(gn_ctx* ctx, const void* x, void* out, int32_t B, int64_t HW, int32_t ld, const int32_t* order,
                        const float* factors, void* workspace);
int32_t gn_reflect_pad_crop(gn_ctx* ctx, const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad,
                            int32_t crop_i, int32_t crop_j);

/* ---- op programs: record once, replay on the stream (eagerly or as a captured hipGraph) ---------------------------
 * The host classes (UNet2DConditionModel / ControlNetModel / AutoencoderKL / pipeline) lower a forward pass to a flat list of
 * the ops above with all buffers pre-allocated, so the 5-step denoise loop runs without returning to Python. */
int32_t gn_program_create(gn_ctx* ctx, gn_program** out);
int32_t gn_program_destroy(gn_program* p);
int32_t gn_program_add_gemm(gn_program* p, const gn_gemm_desc* d);
int32_t gn_program_add_attention(gn_program* p, const gn_attn_desc* d);
int32_t gn_program_add_tblock(gn_program* p, const gn_tblock_desc* d);
int32_t gn_program_add_conv3x3_gn(gn_program* p, const gn_conv3x3_gn_desc* d);
int32_t gn_program_add_groupnorm(gn_program* p, const gn_groupnorm_desc* d);
int32_t gn_program_add_layernorm(gn_program* p, const void* x, const void* gamma, const void* beta, void* y, int64_t M,
                                 int32_t C, float eps);
int32_t gn_program_add_timestep_embedding(gn_program* p, const float* t, void* out, int32_t B, int32_t dim,
                                          int32_t flip_sin_to_cos, float freq_shift);
int32_t gn_program_add_scale_pad(gn_program* p, const void* x, void* out, int64_t pixels, int32_t C, int32_t Cpad,
                                 float scale);
int32_t gn_program_add_euler_step(gn_program* p, void* x, const void* eps, int64_t pixels, int32_t C, int32_t ld_eps,
                                  float sigma, float sigma_next);
int32_t gn_program_add_scale_cat_pad(gn_program* p, const void* x, const void* x2, void* out, int64_t pixels, int32_t C, int32_t ld1,
                                     int32_t C2, int32_t ld2, int32_t Cpad, float scale, float scale2);
int32_t gn_program_add_image_f16_to_u8(gn_program* p, const void* in, uint8_t* out, int64_t pixels, int32_t ld);
int32_t gn_program_add_image_u8_to_f16(gn_program* p, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad,
                                       float mul, float add);
/* stream control inside a program: ops recorded after gn_program_add_fork run on the program's second HIP stream (which first
 * waits for everything recorded before the fork) until gn_program_add_main switches back; gn_program_add_join makes the main
 * stream wait for the side stream.  Used to run the ControlNet next to the UNet encoder inside a denoise step (both only read the
 * scaled latents); also valid under hipGraph capture (fork / join become graph edges).  Partial replays (gn_program_run of a
 * sub-range) ignore them and run serially. */
int32_t gn_program_add_fork(gn_program* p);
int32_t gn_program_add_main(gn_program* p);
int32_t gn_program_add_join(gn_program* p);
int32_t gn_program_add_add(gn_program* p, const void* a, const void* b, void* out, int64_t n);
int32_t gn_program_add_add_multi(gn_program* p, const void* const* a, const void* const* b, void* const* out, const int64_t* n, int32_t count);
int32_t gn_program_add_add_noise(gn_program* p, const void* x0, const void* noise, const float* sqrt_ac, const float* sqrt_1mac,
                                 void* out, int32_t B, int64_t per_sample);
int32_t gn_program_add_act(gn_program* p, const void* x, void* out, int64_t n, int32_t act);
int32_t gn_program_add_film(gn_program* p, const void* x, void* out, const void* gamma, const void* beta, int64_t ld_film,
                            int64_t rows_per_film, int64_t rows, int32_t C, int32_t act);
int32_t gn_program_add_embedding(gn_program* p, const int32_t* ids, const void* tok, const void* pos, void* out,
                                 int32_t B, int32_t L, int32_t D);
int32_t gn_program_add_softmax_rows(gn_program* p, void* x, int64_t rows, int32_t cols, int32_t ld, float scale);
int32_t gn_program_add_maxpool3x3s2(gn_program* p, const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C);
int32_t gn_program_add_image_normalize_u8(gn_program* p, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float m0,
                                          float m1, float m2, float a0, float a1, float a2);
int32_t gn_program_add_gather_rows(gn_program* p, const void* x, const int32_t* idx, void* out, int32_t B, int32_t L, int32_t D);
int32_t gn_program_add_copy4d(gn_program* p, const void* in, void* out, const int64_t* sizes, const int64_t* in_strides,
                              const int64_t* out_strides, int32_t L);
int32_t gn_program_add_argmax_rows_i32(gn_program* p, const int32_t* x, int32_t* out, int32_t rows, int32_t cols);
int64_t gn_program_num_ops(const gn_program* p);
/* In-call tile tuning.  The reference has no counterpart (diffusers leaves kernel selection to cuDNN's own autotuner inside
 * `self.pipe(...)`, controller/agent/sd_controlnet_agent.py:67-76); here the tile / K split of a recorded gn_gemm op can be read back and
 * replaced, so that candidates are timed INSIDE the recorded call (cold operands, the neighbours' cache state) instead of in an isolated
 * hot loop.  get: GN_ERR_INVALID when op is not a gn_gemm.  set: `workspace` replaces the op's split-K scratch pointer (NULL keeps the
 * recorded one; a plan that splits K with neither is refused); refused on a captured program. */
int32_t gn_program_get_gemm(const gn_program* p, int64_t op, gn_gemm_desc* out);
int32_t gn_program_set_gemm_plan(gn_program* p, int64_t op, int32_t tile, int32_t splitk, void* workspace);
/* GroupNorm bridge plumbing of a recorded program: the consumer of a tensor is recorded AFTER its producer, so the producer's sink is attached
 * afterwards -- op = a recorded gn_gemm (index 0) or gn_add_multi (index = which of its tensors); and the statistics arena is cleared by a
 * memset op at the top of every replay */
int32_t gn_program_set_sink(gn_program* p, int64_t op, int32_t index, const gn_stats_sink* sink, int32_t channels);
/* attach norm_out to a recorded gn_gemm (the GroupNorm that reads its output is recorded after it) */
int32_t gn_program_set_norm_out(gn_program* p, int64_t op, const gn_norm_out* n);
int32_t gn_program_add_memset(gn_program* p, void* ptr, int64_t bytes);
int32_t gn_program_set_memset_bytes(gn_program* p, int64_t op, int64_t bytes); /* shrink a recorded memset to the bytes the program came to use */
int32_t gn_memset(gn_ctx* ctx, void* ptr, int64_t bytes);
/* sizeof() of the descriptor structs as the library was compiled (0 gn_gemm_desc, 1 gn_attn_desc, 2 gn_groupnorm_desc, 3 gn_tblock_desc,
 * 4 gn_conv3x3_gn_desc, 5 gn_stats_sink, 6 gn_norm_in, 7 gn_norm_out): a host binding checks its own layout against these before the first call */
int64_t gn_desc_sizeof(int32_t which);
/* first..last (exclusive) op range; last < 0 = to the end */
int32_t gn_program_run(gn_program* p, int64_t first, int64_t last);
int32_t gn_program_capture(gn_program* p);   /* capture the whole program into a hipGraph on the ctx stream */
int32_t gn_program_launch(gn_program* p);    /* replay the captured graph */

/* ---- timing helper: HIP events on the ctx stream (bench.py's roofline leg) ------------------------------------- */
int32_t gn_event_create(void** ev);
int32_t gn_event_destroy(void* ev);
int32_t gn_event_record(gn_ctx* ctx, void* ev);
int32_t gn_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int32_t gn_stream_synchronize(gn_ctx* ctx);

/* ---- ACT controller update (controller/method/genima_act.py:27-139, :348-422) ----------------------------
 * gn_film_bwd: backward of gn_film (act NONE / RELU): dx = dz (1 + gamma), with dz = dy * act'(z); optional copies dz and dz * x whose
 *   per-(batch, channel) column sums (gn_colsum_f32) are dbeta / dgamma.
 * gn_dropout: out = x * keep_mask * scale (inverted dropout, mask drawn by the caller; the backward is the same call on dy).
 * gn_cvae_sample / gn_cvae_bwd: z = mu + exp(logvar / 2) eps over info = [mu | logvar] rows (reparametrize, :64-68) and its backward
 *   plus the KL term's gradient (kl_scale = loss_scale * kl_weight / B).
 * gn_act_loss: calculate_loss (:115-139): out4 = (loss, l1, gripper BCE x 0.05, kl); d_a_hat = grad_scale * d(l1 + gripper)/d(a_hat);
 *   a_hat f16 [B][T_rows][ld_hat] (rows >= T and columns >= A are padding), actions f32 [B][T][A], is_pad u8 [B][T] or NULL.
 * gn_add_f32_to_f16: dst[b, c] += src[b, c] (f32 column sums into an f16 feature gradient). */
int32_t gn_film_bwd(gn_ctx* ctx, const void* dy, const void* x, const void* gamma, const void* beta, int64_t ld_film, int64_t rows_per_film,
                    int64_t rows, int32_t C, int32_t act, void* dx, void* dz, void* dzx);
int32_t gn_dropout(gn_ctx* ctx, const void* x, const uint8_t* keep_mask, void* out, int64_t n, float scale);
int32_t gn_cvae_sample(gn_ctx* ctx, const void* info, int64_t ld_info, const float* eps, void* z, int64_t ld_z, int32_t B, int32_t L);
int32_t gn_cvae_bwd(gn_ctx* ctx, const void* info, int64_t ld_info, const float* eps, const void* dz, int64_t ld_z, void* dinfo, int32_t B, int32_t L,
                    float kl_scale);
int32_t gn_act_loss(gn_ctx* ctx, const void* a_hat, int64_t ld_hat, int64_t bs_hat, const float* actions, const uint8_t* is_pad, const void* info,
                    int64_t ld_info, int32_t B, int32_t T, int32_t T_rows, int32_t A, int32_t L, float kl_weight, float grad_scale, float* out4,
                    void* d_a_hat);
int32_t gn_add_f32_to_f16(gn_ctx* ctx, const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t B, int32_t C);
/* train-time ElasticTransform of the ACT policy (controller/method/genima_act.py:150-163): bilinear warp of NHWC f16 [B, H, W, C] by one
 * displacement field disp f32 [H][W][2] = (dx, dy) in pixels shared by the batch; samples outside the image read 0 */
int32_t gn_warp_bilinear(gn_ctx* ctx, const void* in, void* out, const float* disp, int32_t B, int32_t H, int32_t W, int32_t C);

/* ---- data-parallel gradient exchange (SURVEY.md section 8b "comm"; replaces accelerate's DDP all-reduce under accelerator.backward,
 * diffusion/train_controlnet_genima.py:1216-1218, :1402-1405).  One communicator per (process, GPU); the RCCL unique id (128 bytes) is
 * created on rank 0 and carried to the other ranks by the caller.  gn_comm_allreduce_grads SUMS one flat f32 buffer over the ranks in
 * place -- RCCL reduce-scatter + all-gather (+ a small all-reduce for a tail shorter than nranks) on the communicator's own HIP stream,
 * ordered behind everything launched so far on ctx's stream -- and returns at once; gn_comm_wait makes ctx's stream wait for it.
 * wire_bf16 = 1 sends bf16 on the links (the sum then rounds to bf16: opt-in).  scratch: device memory of gn_comm_scratch_bytes(). */
typedef struct gn_comm gn_comm;
int32_t gn_comm_unique_id(void* id128);
int32_t gn_comm_init(gn_ctx* ctx, int32_t rank, int32_t nranks, const void* id128, gn_comm** out);
int32_t gn_comm_destroy(gn_comm* comm);
int64_t gn_comm_scratch_bytes(const gn_comm* comm, int64_t count, int32_t wire_bf16);
int32_t gn_comm_allreduce_grads(gn_comm* comm, float* buf, int64_t count, int32_t wire_bf16, void* scratch);
int32_t gn_comm_wait(gn_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* GENIMA_HIP_H */
