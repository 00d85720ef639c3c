// C-ABI plumbing of libgenima_hip.so: context, thread-local error string, op programs (record / replay / hipGraph), events.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "common.h"

static thread_local char g_err[1024] = "";

void gn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

enum OpType {
  OP_GEMM = 0, OP_ATTN, OP_GROUPNORM, OP_LAYERNORM, OP_TEMB, OP_SCALE_PAD, OP_EULER, OP_F16_TO_U8, OP_U8_TO_F16, OP_ADD,
  OP_ACT, OP_EMBED, OP_SOFTMAX, OP_MAXPOOL, OP_NORMALIZE_U8, OP_GATHER_ROWS, OP_COPY4D, OP_ARGMAX, OP_ADD_NOISE, OP_FILM, OP_SCALE_CAT_PAD,
  OP_TBLOCK, OP_CONV_GN, OP_ADD_MULTI, OP_MEMSET,
  OP_FORK, OP_MAIN, OP_JOIN  // stream control: ops after FORK go to the program's side stream until MAIN; JOIN makes main wait for it
};

struct GenericArgs {  // argument block of the small ops
  const void* p0; const void* p1; const void* p2; void* p3;
  int64_t n0, n1;
  int32_t i0, i1, i2, i3;
  float f0, f1;
  int64_t m[12];  // copy4d: sizes[4], in_strides[4], out_strides[4]
  float g[6];     // normalize: mul[3], add[3]
};

struct AddMultiOp {
  const void* a[GN_ADD_MULTI_MAX]; const void* b[GN_ADD_MULTI_MAX]; void* out[GN_ADD_MULTI_MAX]; int64_t n[GN_ADD_MULTI_MAX]; int32_t count;
  int32_t C[GN_ADD_MULTI_MAX]; gn_stats_sink sink[GN_ADD_MULTI_MAX]; int32_t has_sink;  // GroupNorm bridge (gn_program_set_sink)
};

struct Op {
  int type;
  union {
    AddMultiOp addm;
    gn_gemm_desc gemm;
    gn_attn_desc attn;
    gn_groupnorm_desc gnorm;
    gn_tblock_desc tblock;
    gn_conv3x3_gn_desc convgn;
    GenericArgs g;
  };
};

}  // namespace

struct gn_program {
  gn_ctx* ctx;
  std::vector<Op> ops;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t side = nullptr;          // second stream for independent sub-graphs (ControlNet next to the UNet encoder)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static int32_t run_op(gn_ctx* ctx, const Op& op) {
  const GenericArgs& g = op.g;
  switch (op.type) {
    case OP_GEMM: return gn_launch_gemm(ctx, &op.gemm);
    case OP_ATTN: return gn_launch_attention(ctx, &op.attn);
    case OP_GROUPNORM: return gn_launch_groupnorm(ctx, &op.gnorm);
    case OP_TBLOCK: return gn_launch_tblock(ctx, &op.tblock);
    case OP_CONV_GN: return gn_conv3x3_gn(ctx, &op.convgn);
    case OP_LAYERNORM: return gn_layernorm_fwd(ctx, g.p0, g.p1, g.p2, g.p3, g.n0, g.i0, g.f0);
    case OP_TEMB: return gn_timestep_embedding(ctx, (const float*)g.p0, g.p3, g.i0, g.i1, g.i2, g.f0);
    case OP_SCALE_PAD: return gn_scale_pad(ctx, g.p0, g.p3, g.n0, g.i0, g.i1, g.f0);
    case OP_SCALE_CAT_PAD: return gn_scale_cat_pad(ctx, g.p0, g.p1, g.p3, g.n0, g.i0, (int32_t)g.n1, g.i1, g.i2, g.i3, g.f0, g.f1);
    case OP_EULER: return gn_euler_step(ctx, g.p3, g.p0, g.n0, g.i0, g.i1, g.f0, g.f1);
    case OP_F16_TO_U8: return gn_image_f16_to_u8(ctx, g.p0, (uint8_t*)g.p3, g.n0, g.i0);
    case OP_U8_TO_F16: return gn_image_u8_to_f16(ctx, (const uint8_t*)g.p0, g.p3, g.n0, g.i0, g.f0, g.f1);
    case OP_ADD: return gn_add(ctx, g.p0, g.p1, g.p3, g.n0);
    case OP_ADD_MULTI:
      if (op.addm.has_sink) return gn_add_multi_stats(ctx, op.addm.a, op.addm.b, op.addm.out, op.addm.n, op.addm.C, op.addm.sink, op.addm.count);
      return gn_add_multi(ctx, op.addm.a, op.addm.b, op.addm.out, op.addm.n, op.addm.count);
    case OP_MEMSET: return gn_memset(ctx, g.p3, g.n0);
    case OP_ACT: return gn_act(ctx, g.p0, g.p3, g.n0, g.i0);
    case OP_FILM: return gn_film(ctx, g.p0, g.p3, g.p1, g.p2, g.m[0], g.m[1], g.n0, g.i0, g.i1);
    case OP_EMBED: return gn_embedding(ctx, (const int32_t*)g.p0, g.p1, g.p2, g.p3, g.i0, g.i1, g.i2);
    case OP_SOFTMAX: return gn_softmax_rows(ctx, g.p3, g.n0, g.i0, g.i1, g.f0);
    case OP_MAXPOOL: return gn_maxpool3x3s2(ctx, g.p0, g.p3, g.i0, g.i1, g.i2, g.i3);
    case OP_NORMALIZE_U8: return gn_image_normalize_u8(ctx, (const uint8_t*)g.p0, g.p3, g.n0, g.i0, g.g[0], g.g[1], g.g[2], g.g[3], g.g[4], g.g[5]);
    case OP_GATHER_ROWS: return gn_gather_rows(ctx, g.p0, (const int32_t*)g.p1, g.p3, g.i0, g.i1, g.i2);
    case OP_COPY4D: return gn_copy4d(ctx, g.p0, g.p3, g.m, g.m + 4, g.m + 8, g.i0);
    case OP_ARGMAX: return gn_argmax_rows_i32(ctx, (const int32_t*)g.p0, (int32_t*)g.p3, g.i0, g.i1);
    case OP_ADD_NOISE: return gn_add_noise(ctx, g.p0, g.p1, (const float*)g.p2, (const float*)(uintptr_t)g.m[0], g.p3, g.i0, g.n0);
    default: gn_set_error("gn_program: unknown op type %d", op.type); return GN_ERR_INVALID;
  }
}

static int32_t push_generic(gn_program* p, int type, const void* p0, const void* p1, const void* p2, void* p3, int64_t n0,
                            int64_t n1, int32_t i0, int32_t i1, int32_t i2, int32_t i3, float f0, float f1) {
  GN_REQUIRE(p, "gn_program_add_*: null program");
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = type;
  op.g.p0 = p0; op.g.p1 = p1; op.g.p2 = p2; op.g.p3 = p3;
  op.g.n0 = n0; op.g.n1 = n1;
  op.g.i0 = i0; op.g.i1 = i1; op.g.i2 = i2; op.g.i3 = i3;
  op.g.f0 = f0; op.g.f1 = f1;
  p->ops.push_back(op);
  return GN_OK;
}

extern "C" {

int32_t gn_version(void) { return 101; }  // 101: gn_gemm_desc grew (sink, norm_in, norm_out; round 5) + tile 25, gn_gemm_plan_valid, GN_STATS_SHIFT_SQ (round 6)
const char* gn_last_error(void) { return g_err; }

int32_t gn_ctx_create(int32_t device, void* stream, gn_ctx** out) {
  GN_REQUIRE(out, "gn_ctx_create: null out");
  int n = 0;
  GN_HIP(hipGetDeviceCount(&n));
  GN_REQUIRE(device >= 0 && device < n, "gn_ctx_create: device %d out of range (%d visible)", device, n);
  GN_HIP(hipSetDevice(device));
  if (gn_ppp_pool_init(device) != GN_OK) return GN_ERR_HIP;  // the persistent GEMM's hand-off flags (2 MB, zeroed once; never allocated inside a capture)
  gn_ctx* c = new gn_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  *out = c;
  return GN_OK;
}
int32_t gn_ctx_destroy(gn_ctx* ctx) {
  delete ctx;
  return GN_OK;
}
int32_t gn_ctx_set_stream(gn_ctx* ctx, void* stream) {
  GN_REQUIRE(ctx, "gn_ctx_set_stream: null ctx");
  ctx->stream = (hipStream_t)stream;
  return GN_OK;
}

int32_t gn_gemm(gn_ctx* ctx, const gn_gemm_desc* d) {
  GN_REQUIRE(ctx, "gn_gemm: null ctx");
  return gn_launch_gemm(ctx, d);
}
int32_t gn_attention_fwd(gn_ctx* ctx, const gn_attn_desc* d) {
  GN_REQUIRE(ctx, "gn_attention_fwd: null ctx");
  return gn_launch_attention(ctx, d);
}
int32_t gn_tblock(gn_ctx* ctx, const gn_tblock_desc* d) {
  GN_REQUIRE(ctx, "gn_tblock: null ctx");
  return gn_launch_tblock(ctx, d);
}
int32_t gn_groupnorm_fwd(gn_ctx* ctx, const gn_groupnorm_desc* d) {
  GN_REQUIRE(ctx, "gn_groupnorm_fwd: null ctx");
  return gn_launch_groupnorm(ctx, d);
}

// ---- programs ---------------------------------------------------------------------------------------------------------
int32_t gn_program_create(gn_ctx* ctx, gn_program** out) {
  GN_REQUIRE(ctx && out, "gn_program_create: null argument");
  gn_program* p = new gn_program();
  p->ctx = ctx;
  *out = p;
  return GN_OK;
}
int32_t gn_program_destroy(gn_program* p) {
  if (p) {
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
  }
  return GN_OK;
}
int32_t gn_program_add_gemm(gn_program* p, const gn_gemm_desc* d) {
  GN_REQUIRE(p && d, "gn_program_add_gemm: null argument");
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = OP_GEMM;
  op.gemm = *d;
  p->ops.push_back(op);
  return GN_OK;
}
int32_t gn_program_add_attention(gn_program* p, const gn_attn_desc* d) {
  GN_REQUIRE(p && d, "gn_program_add_attention: null argument");
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = OP_ATTN;
  op.attn = *d;
  p->ops.push_back(op);
  return GN_OK;
}
int32_t gn_program_add_tblock(gn_program* p, const gn_tblock_desc* d) {
  GN_REQUIRE(p && d, "gn_program_add_tblock: null argument");
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = OP_TBLOCK;
  op.tblock = *d;
  p->ops.push_back(op);
  return GN_OK;
}
int32_t gn_program_add_conv3x3_gn(gn_program* p, const gn_conv3x3_gn_desc* d) {
  GN_REQUIRE(p && d, "gn_program_add_conv3x3_gn: null argument");
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = OP_CONV_GN;
  op.convgn = *d;
  p->ops.push_back(op);
  return GN_OK;
}
int32_t gn_program_add_groupnorm(gn_program* p, const gn_groupnorm_desc* d) {
  GN_REQUIRE(p && d, "gn_program_add_groupnorm: null argument");
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = OP_GROUPNORM;
  op.gnorm = *d;
  p->ops.push_back(op);
  return GN_OK;
}
int32_t gn_program_add_layernorm(gn_program* p, const void* x, const void* gamma, const void* beta, void* y, int64_t M, int32_t C, float eps) {
  return push_generic(p, OP_LAYERNORM, x, gamma, beta, y, M, 0, C, 0, 0, 0, eps, 0.f);
}
int32_t gn_program_add_timestep_embedding(gn_program* p, const float* t, void* out, int32_t B, int32_t dim, int32_t flip, float freq_shift) {
  return push_generic(p, OP_TEMB, t, nullptr, nullptr, out, 0, 0, B, dim, flip, 0, freq_shift, 0.f);
}
int32_t gn_program_add_scale_pad(gn_program* p, const void* x, void* out, int64_t pixels, int32_t C, int32_t Cpad, float scale) {
  return push_generic(p, OP_SCALE_PAD, x, nullptr, nullptr, out, pixels, 0, C, Cpad, 0, 0, scale, 0.f);
}
int32_t gn_program_add_scale_cat_pad(gn_program* p, const void* x, const void* x2, void* out, int64_t pixels, int32_t C, int32_t ld1,
                                     int32_t C2, int32_t ld2, int32_t Cpad, float scale, float scale2) {
  return push_generic(p, OP_SCALE_CAT_PAD, x, x2, nullptr, out, pixels, ld1, C, C2, ld2, Cpad, scale, scale2);
}
int32_t gn_program_add_euler_step(gn_program* p, void* x, const void* eps, int64_t pixels, int32_t C, int32_t ld_eps, float sigma, float sigma_next) {
  return push_generic(p, OP_EULER, eps, nullptr, nullptr, x, pixels, 0, C, ld_eps, 0, 0, sigma, sigma_next);
}
int32_t gn_program_add_image_f16_to_u8(gn_program* p, const void* in, uint8_t* out, int64_t pixels, int32_t ld) {
  return push_generic(p, OP_F16_TO_U8, in, nullptr, nullptr, out, pixels, 0, ld, 0, 0, 0, 0.f, 0.f);
}
int32_t gn_program_add_image_u8_to_f16(gn_program* p, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float mul, float add) {
  return push_generic(p, OP_U8_TO_F16, in, nullptr, nullptr, out, pixels, 0, Cpad, 0, 0, 0, mul, add);
}
int32_t gn_program_add_add(gn_program* p, const void* a, const void* b, void* out, int64_t n) {
  return push_generic(p, OP_ADD, a, b, nullptr, out, n, 0, 0, 0, 0, 0, 0.f, 0.f);
}
int32_t gn_program_add_add_multi(gn_program* p, const void* const* a, const void* const* b, void* const* out, const int64_t* n, int32_t count) {
  GN_REQUIRE(p && a && b && out && n && count >= 1 && count <= GN_ADD_MULTI_MAX, "gn_program_add_add_multi: 1 .. %d tensors", GN_ADD_MULTI_MAX);
  Op op;
  memset(&op, 0, sizeof(op));
  op.type = OP_ADD_MULTI;
  for (int i = 0; i < count; ++i) { op.addm.a[i] = a[i]; op.addm.b[i] = b[i]; op.addm.out[i] = out[i]; op.addm.n[i] = n[i]; }
  op.addm.count = count;
  p->ops.push_back(op);
  return GN_OK;
}
int32_t gn_program_add_film(gn_program* p, const void* x, void* out, const void* gamma, const void* beta, int64_t ld_film,
                            int64_t rows_per_film, int64_t rows, int32_t C, int32_t act) {
  const int32_t rc = push_generic(p, OP_FILM, x, gamma, beta, out, rows, 0, C, act, 0, 0, 0.f, 0.f);
  if (rc == GN_OK) { p->ops.back().g.m[0] = ld_film; p->ops.back().g.m[1] = rows_per_film; }
  return rc;
}
int32_t gn_program_add_act(gn_program* p, const void* x, void* out, int64_t n, int32_t act) {
  return push_generic(p, OP_ACT, x, nullptr, nullptr, out, n, 0, act, 0, 0, 0, 0.f, 0.f);
}
int32_t gn_program_add_embedding(gn_program* p, const int32_t* ids, const void* tok, const void* pos, void* out, int32_t B, int32_t L, int32_t D) {
  return push_generic(p, OP_EMBED, ids, tok, pos, out, 0, 0, B, L, D, 0, 0.f, 0.f);
}
int32_t gn_program_add_softmax_rows(gn_program* p, void* x, int64_t rows, int32_t cols, int32_t ld, float scale) {
  return push_generic(p, OP_SOFTMAX, nullptr, nullptr, nullptr, x, rows, 0, cols, ld, 0, 0, scale, 0.f);
}
int32_t gn_program_add_maxpool3x3s2(gn_program* p, const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C) {
  return push_generic(p, OP_MAXPOOL, x, nullptr, nullptr, y, 0, 0, B, H, W, C, 0.f, 0.f);
}
int32_t gn_program_add_image_normalize_u8(gn_program* p, const uint8_t* in, void* out, int64_t pixels, int32_t Cpad, float m0,
                                          float m1, float m2, float a0, float a1, float a2) {
  int32_t rc = push_generic(p, OP_NORMALIZE_U8, in, nullptr, nullptr, out, pixels, 0, Cpad, 0, 0, 0, 0.f, 0.f);
  if (rc == GN_OK) {
    float* g = p->ops.back().g.g;
    g[0] = m0; g[1] = m1; g[2] = m2; g[3] = a0; g[4] = a1; g[5] = a2;
  }
  return rc;
}
int32_t gn_program_add_gather_rows(gn_program* p, const void* x, const int32_t* idx, void* out, int32_t B, int32_t L, int32_t D) {
  return push_generic(p, OP_GATHER_ROWS, x, idx, nullptr, out, 0, 0, B, L, D, 0, 0.f, 0.f);
}
int32_t gn_program_add_copy4d(gn_program* p, const void* in, void* out, const int64_t* sizes, const int64_t* in_strides,
                              const int64_t* out_strides, int32_t L) {
  GN_REQUIRE(sizes && in_strides && out_strides, "gn_program_add_copy4d: null argument");
  int32_t rc = push_generic(p, OP_COPY4D, in, nullptr, nullptr, out, 0, 0, L, 0, 0, 0, 0.f, 0.f);
  if (rc == GN_OK) {
    int64_t* m = p->ops.back().g.m;
    for (int i = 0; i < 4; ++i) { m[i] = sizes[i]; m[4 + i] = in_strides[i]; m[8 + i] = out_strides[i]; }
  }
  return rc;
}
int32_t gn_program_add_argmax_rows_i32(gn_program* p, const int32_t* x, int32_t* out, int32_t rows, int32_t cols) {
  return push_generic(p, OP_ARGMAX, x, nullptr, nullptr, out, 0, 0, rows, cols, 0, 0, 0.f, 0.f);
}
int32_t gn_program_add_add_noise(gn_program* p, const void* x0, const void* noise, const float* sqrt_ac, const float* sqrt_1mac, void* out, int32_t B,
                                 int64_t per_sample) {
  const int32_t rc = push_generic(p, OP_ADD_NOISE, x0, noise, sqrt_ac, out, per_sample, 0, B, 0, 0, 0, 0.f, 0.f);
  if (rc == GN_OK) p->ops.back().g.m[0] = (int64_t)(uintptr_t)sqrt_1mac;
  return rc;
}
int32_t gn_program_add_fork(gn_program* p) { return push_generic(p, OP_FORK, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0.f, 0.f); }
int32_t gn_program_add_main(gn_program* p) { return push_generic(p, OP_MAIN, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0.f, 0.f); }
int32_t gn_program_add_join(gn_program* p) { return push_generic(p, OP_JOIN, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0.f, 0.f); }
int64_t gn_program_num_ops(const gn_program* p) { return p ? (int64_t)p->ops.size() : 0; }

// in-call tile tuning (genima_amd/incall_tune.py): read back a recorded gn_gemm op, and replace its tile / K split (+ the workspace its
// split-K partial sums use).  A tile never changes the summation order along K; a K split does (as documented on gn_gemm_desc).
int32_t gn_program_get_gemm(const gn_program* p, int64_t op, gn_gemm_desc* out) {
  GN_REQUIRE(p && out && op >= 0 && op < (int64_t)p->ops.size(), "gn_program_get_gemm: bad argument");
  if (p->ops[(size_t)op].type != OP_GEMM) return GN_ERR_INVALID;  // (no error text: callers probe every op)
  *out = p->ops[(size_t)op].gemm;
  return GN_OK;
}
int32_t gn_program_set_gemm_plan(gn_program* p, int64_t op, int32_t tile, int32_t splitk, void* workspace) {
  GN_REQUIRE(p && op >= 0 && op < (int64_t)p->ops.size() && p->ops[(size_t)op].type == OP_GEMM, "gn_program_set_gemm_plan: op %ld is not a gn_gemm", (long)op);
  GN_REQUIRE(!p->exec, "gn_program_set_gemm_plan: the program is captured (re-capture after tuning)");
  GN_REQUIRE(tile >= 0 && tile <= GN_NUM_GEMM_TILES && splitk >= 0, "gn_program_set_gemm_plan: tile %d (0 = heuristic, 1 .. %d) / splitk %d out of range", tile, GN_NUM_GEMM_TILES, splitk);
  gn_gemm_desc d = p->ops[(size_t)op].gemm;  // validated on a copy: a refused plan leaves the recorded op as it was
  d.tile = tile; d.splitk = splitk;
  if (workspace) d.workspace = workspace;
  GN_REQUIRE(gn_gemm_workspace_bytes(&d) == 0 || d.workspace, "gn_program_set_gemm_plan: this plan splits K and needs a workspace");
  // the fusions attached to the op must survive the new plan (gn_launch_gemm would refuse it at replay, possibly in the middle of a hipGraph capture):
  // norm_out lives in the split-K reduce, norm_in in the ring tiles, the plan must name the tile that will run
  if (!gn_gemm_plan_valid(&d)) {
    char why[600];
    snprintf(why, sizeof(why), "%s", g_err);  // (gn_set_error formats into g_err itself)
    gn_set_error("gn_program_set_gemm_plan: tile %d / splitk %d refused for op %ld: %s", tile, splitk, (long)op, why);
    return GN_ERR_INVALID;
  }
  p->ops[(size_t)op].gemm = d;
  return GN_OK;
}

int32_t gn_program_set_sink(gn_program* p, int64_t op, int32_t index, const gn_stats_sink* sink, int32_t channels) {
  GN_REQUIRE(p && sink && op >= 0 && op < (int64_t)p->ops.size(), "gn_program_set_sink: bad argument");
  GN_REQUIRE(!p->exec, "gn_program_set_sink: the program is captured");
  Op& o = p->ops[(size_t)op];
  if (o.type == OP_GEMM) {
    GN_REQUIRE(index == 0 && !o.gemm.sink.stats, "gn_program_set_sink: a gn_gemm op has one output and takes one sink");
    GN_REQUIRE(o.gemm.out_mode == GN_OUT_ROWMAJOR && !o.gemm.out2 && o.gemm.act != GN_ACT_GEGLU && !o.gemm.fp8 && (o.gemm.batch <= 1 || o.gemm.up_phases),
               "gn_program_set_sink: this gn_gemm does not write a plain row-major f16 tensor");
    o.gemm.sink = *sink;
    return GN_OK;
  }
  if (o.type == OP_ADD_MULTI) {
    GN_REQUIRE(index >= 0 && index < o.addm.count && !o.addm.sink[index].stats && channels > 0 && channels % 8 == 0 && o.addm.n[index] % channels == 0,
               "gn_program_set_sink: bad gn_add_multi tensor index / channel count");
    for (int i = 0; i < o.addm.count; ++i)
      if (o.addm.C[i] == 0) o.addm.C[i] = 8;  // tensors without a sink: any row shape
    o.addm.C[index] = channels;
    o.addm.sink[index] = *sink;
    o.addm.has_sink = 1;
    return GN_OK;
  }
  gn_set_error("gn_program_set_sink: op %ld is neither a gn_gemm nor a gn_add_multi", (long)op);
  return GN_ERR_INVALID;
}
int32_t gn_program_set_norm_out(gn_program* p, int64_t op, const gn_norm_out* n) {
  GN_REQUIRE(p && n && n->y && op >= 0 && op < (int64_t)p->ops.size() && p->ops[(size_t)op].type == OP_GEMM, "gn_program_set_norm_out: op %ld is not a gn_gemm", (long)op);
  GN_REQUIRE(!p->exec, "gn_program_set_norm_out: the program is captured");
  gn_gemm_desc d = p->ops[(size_t)op].gemm;
  GN_REQUIRE(!d.norm_out.y, "gn_program_set_norm_out: this gn_gemm already normalises its output");
  d.norm_out = *n;
  GN_REQUIRE(gn_gemm_norm_out_supported(&d), "gn_program_set_norm_out: the recorded problem / plan does not take norm_out");
  p->ops[(size_t)op].gemm = d;
  return GN_OK;
}
int32_t gn_program_set_memset_bytes(gn_program* p, int64_t op, int64_t bytes) {
  GN_REQUIRE(p && op >= 0 && op < (int64_t)p->ops.size() && p->ops[(size_t)op].type == OP_MEMSET && bytes > 0 && bytes <= p->ops[(size_t)op].g.n0,
             "gn_program_set_memset_bytes: op %ld is not a memset of at least %ld bytes", (long)op, (long)bytes);
  GN_REQUIRE(!p->exec, "gn_program_set_memset_bytes: the program is captured");
  p->ops[(size_t)op].g.n0 = bytes;
  return GN_OK;
}
int64_t gn_desc_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(gn_gemm_desc);
    case 1: return (int64_t)sizeof(gn_attn_desc);
    case 2: return (int64_t)sizeof(gn_groupnorm_desc);
    case 3: return (int64_t)sizeof(gn_tblock_desc);
    case 4: return (int64_t)sizeof(gn_conv3x3_gn_desc);
    case 5: return (int64_t)sizeof(gn_stats_sink);
    case 6: return (int64_t)sizeof(gn_norm_in);
    case 7: return (int64_t)sizeof(gn_norm_out);
    default: return -1;
  }
}
int32_t gn_program_add_memset(gn_program* p, void* ptr, int64_t bytes) {
  GN_REQUIRE(ptr && bytes > 0, "gn_program_add_memset: null / empty");
  return push_generic(p, OP_MEMSET, nullptr, nullptr, nullptr, ptr, bytes, 0, 0, 0, 0, 0, 0.f, 0.f);
}

static int32_t ensure_side_stream(gn_program* p) {
  if (!p->side) {
    GN_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    GN_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    GN_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
  }
  return GN_OK;
}

int32_t gn_program_run(gn_program* p, int64_t first, int64_t last) {
  GN_REQUIRE(p, "gn_program_run: null program");
  const int64_t n = (int64_t)p->ops.size();
  if (last < 0 || last > n) last = n;
  GN_REQUIRE(first >= 0 && first <= last, "gn_program_run: bad range [%ld, %ld)", (long)first, (long)last);
  // stream-control ops only act on whole-program replays; a sub-range (per-op timing) runs serially on the context's stream
  const bool two_streams = first == 0 && last == n;
  hipStream_t main_stream = p->ctx->stream;
  bool forked = false;  // side stream has work the main stream has not waited for
  int32_t rc = GN_OK;
  for (int64_t i = first; i < last && rc == GN_OK; ++i) {
    const Op& op = p->ops[(size_t)i];
    if (op.type == OP_FORK || op.type == OP_MAIN || op.type == OP_JOIN) {
      if (!two_streams) continue;
      if (op.type == OP_FORK) {
        rc = ensure_side_stream(p);
        if (rc != GN_OK) break;
        if (hipEventRecord(p->ev_fork, main_stream) != hipSuccess || hipStreamWaitEvent(p->side, p->ev_fork, 0) != hipSuccess) {
          gn_set_error("gn_program_run: fork failed"); rc = GN_ERR_HIP; break;
        }
        p->ctx->stream = p->side;
        forked = true;
      } else if (op.type == OP_MAIN) {
        p->ctx->stream = main_stream;
      } else if (forked) {
        p->ctx->stream = main_stream;
        if (hipEventRecord(p->ev_join, p->side) != hipSuccess || hipStreamWaitEvent(main_stream, p->ev_join, 0) != hipSuccess) {
          gn_set_error("gn_program_run: join failed"); rc = GN_ERR_HIP; break;
        }
        forked = false;
      }
      continue;
    }
    rc = run_op(p->ctx, op);
    if (rc != GN_OK) {
      char tmp[900];
      snprintf(tmp, sizeof(tmp), "%s", g_err);
      gn_set_error("op %ld (type %d): %s", (long)i, op.type, tmp);
    }
  }
  p->ctx->stream = main_stream;
  if (forked) {  // never leave the side stream dangling (also required to end a stream capture)
    (void)hipEventRecord(p->ev_join, p->side);
    (void)hipStreamWaitEvent(main_stream, p->ev_join, 0);
  }
  return rc;
}

int32_t gn_program_capture(gn_program* p) {
  GN_REQUIRE(p, "gn_program_capture: null program");
  GN_REQUIRE(p->ctx->stream != nullptr, "gn_program_capture: needs a non-default stream");
  if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
  GN_HIP(hipStreamBeginCapture(p->ctx->stream, hipStreamCaptureModeThreadLocal));
  int32_t rc = gn_program_run(p, 0, -1);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(p->ctx->stream, &g);
  if (rc != GN_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) { gn_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return GN_ERR_HIP; }
  p->graph = g;
  GN_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
  return GN_OK;
}

int32_t gn_program_launch(gn_program* p) {
  GN_REQUIRE(p && p->exec, "gn_program_launch: program not captured");
  GN_HIP(hipGraphLaunch(p->exec, p->ctx->stream));
  return GN_OK;
}

// ---- events ----------------------------------------------------------------------------------------------------------
int32_t gn_event_create(void** ev) {
  GN_REQUIRE(ev, "gn_event_create: null");
  hipEvent_t e;
  GN_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return GN_OK;
}
int32_t gn_event_destroy(void* ev) {
  if (ev) GN_HIP(hipEventDestroy((hipEvent_t)ev));
  return GN_OK;
}
int32_t gn_event_record(gn_ctx* ctx, void* ev) {
  GN_REQUIRE(ctx && ev, "gn_event_record: null");
  GN_HIP(hipEventRecord((hipEvent_t)ev, ctx->stream));
  return GN_OK;
}
int32_t gn_event_elapsed_ms(void* start, void* stop, float* ms) {
  GN_REQUIRE(start && stop && ms, "gn_event_elapsed_ms: null");
  GN_HIP(hipEventSynchronize((hipEvent_t)stop));
  GN_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return GN_OK;
}
int32_t gn_stream_synchronize(gn_ctx* ctx) {
  GN_REQUIRE(ctx, "gn_stream_synchronize: null ctx");
  GN_HIP(hipStreamSynchronize(ctx->stream));
  return GN_OK;
}

}  // extern "C"
