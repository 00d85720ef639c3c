// Data-parallel gradient exchange under the C ABI (SURVEY.md section 8b "comm"; reference: accelerate's DDP all-reduce of the ControlNet
// gradients under accelerator.backward, diffusion/train_controlnet_genima.py:1216-1218, :1402-1405).
//
//   gn_comm_allreduce_grads: in-place SUM over ranks of one flat f32 buffer as RCCL reduce-scatter + all-gather -- on the xGMI mesh
//   every one of a GPU's 7 links then carries 1/N of the buffer per phase, where a ring all-reduce is bound by one link -- issued on the
//   communicator's OWN HIP stream behind an event recorded on the caller's compute stream, so the caller can keep launching kernels
//   (the rest of the backward) and joins with gn_comm_wait.  Optional bf16 wire format (halves the bytes on the links; the sum then
//   runs in bf16 -- opt-in, the default is bit-faithful f32).
// RCCL is loaded lazily with dlopen: a single-GPU host never needs librccl.so, and libgenima_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace {

typedef void* rcclComm_t;
struct RcclUid { char b[128]; };  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
struct RcclApi {
  int (*GetUniqueId)(void*);
  int (*CommInitRank)(rcclComm_t*, int, RcclUid, int);
  int (*CommDestroy)(rcclComm_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  const char* (*GetErrorString)(int);
  bool ok = false;
};
constexpr int kFloat32 = 7, kBfloat16 = 9, kSum = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h

RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      *(void**)&api.GetUniqueId = dlsym(h, "ncclGetUniqueId");
      *(void**)&api.CommInitRank = dlsym(h, "ncclCommInitRank");
      *(void**)&api.CommDestroy = dlsym(h, "ncclCommDestroy");
      *(void**)&api.ReduceScatter = dlsym(h, "ncclReduceScatter");
      *(void**)&api.AllGather = dlsym(h, "ncclAllGather");
      *(void**)&api.AllReduce = dlsym(h, "ncclAllReduce");
      *(void**)&api.GetErrorString = dlsym(h, "ncclGetErrorString");
      api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.ReduceScatter && api.AllGather && api.AllReduce;
    }
  }
  return &api;
}

#define GN_RCCL(expr)                                                                                   \
  do {                                                                                                  \
    int _r = (expr);                                                                                    \
    if (_r != 0) {                                                                                      \
      gn_set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "?", __FILE__, __LINE__); \
      return GN_ERR_HIP;                                                                                \
    }                                                                                                   \
  } while (0)

typedef __bf16 bf16_t;

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = (bf16_t)x[i];
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = (float)x[i];
}

long round_up(long x, long m) { return (x + m - 1) / m * m; }

}  // namespace

struct gn_comm {
  gn_ctx* ctx;
  rcclComm_t comm;
  int rank, nranks;
  hipStream_t stream;
  hipEvent_t ready, done;
};

extern "C" int32_t gn_comm_unique_id(void* id128) {
  GN_REQUIRE(id128, "gn_comm_unique_id: null id");
  GN_REQUIRE(rccl()->ok, "gn_comm_unique_id: librccl.so could not be loaded");
  GN_RCCL(rccl()->GetUniqueId(id128));
  return GN_OK;
}

extern "C" int32_t gn_comm_init(gn_ctx* ctx, int32_t rank, int32_t nranks, const void* id128, gn_comm** out) {
  GN_REQUIRE(ctx && id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "gn_comm_init: bad arguments (rank %d of %d)", rank, nranks);
  GN_REQUIRE(rccl()->ok, "gn_comm_init: librccl.so could not be loaded");
  GN_HIP(hipSetDevice(ctx->device));
  gn_comm* c = new gn_comm();
  c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->comm = nullptr;
  RcclUid uid;
  memcpy(uid.b, id128, 128);
  int r = rccl()->CommInitRank(&c->comm, nranks, uid, rank);
  if (r != 0) {
    gn_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, rccl()->GetErrorString ? rccl()->GetErrorString(r) : "?");
    delete c;
    return GN_ERR_HIP;
  }
  GN_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  GN_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  GN_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  *out = c;
  return GN_OK;
}

extern "C" int32_t gn_comm_destroy(gn_comm* c) {
  if (!c) return GN_OK;
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) rccl()->CommDestroy(c->comm);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return GN_OK;
}

extern "C" int64_t gn_comm_scratch_bytes(const gn_comm* c, int64_t count, int32_t wire_bf16) {
  if (!c || count <= 0) return 0;
  const long main = count - count % c->nranks;
  const long slice = main / c->nranks;
  if (!wire_bf16) return round_up(slice * 4, 256);
  return round_up(main * 2, 256) + round_up(slice * 2, 256);
}

extern "C" int32_t gn_comm_allreduce_grads(gn_comm* c, float* buf, int64_t count, int32_t wire_bf16, void* scratch) {
  GN_REQUIRE(c && buf && count > 0, "gn_comm_allreduce_grads: bad arguments");
  GN_REQUIRE(((uintptr_t)buf & 15) == 0, "gn_comm_allreduce_grads: buf must be 16-byte aligned");
  const long main = count - count % c->nranks;
  const long slice = main / c->nranks;
  GN_REQUIRE(main == 0 || scratch, "gn_comm_allreduce_grads: scratch of gn_comm_scratch_bytes() is required");
  // the exchange starts once everything the caller has launched on its compute stream so far (the gradient writes) has finished
  GN_HIP(hipEventRecord(c->ready, c->ctx->stream));
  GN_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  if (main) {
    if (!wire_bf16) {
      float* mine = (float*)scratch;
      GN_RCCL(rccl()->ReduceScatter(buf, mine, (size_t)slice, kFloat32, kSum, c->comm, c->stream));
      GN_RCCL(rccl()->AllGather(mine, buf, (size_t)slice, kFloat32, c->comm, c->stream));
    } else {
      bf16_t* wire = (bf16_t*)scratch;
      bf16_t* mine = (bf16_t*)((char*)scratch + round_up(main * 2, 256));
      const unsigned blocks = (unsigned)std::min<long>(4096, (main + 1023) / 1024);
      hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, buf, wire, main);
      GN_RCCL(rccl()->ReduceScatter(wire, mine, (size_t)slice, kBfloat16, kSum, c->comm, c->stream));
      GN_RCCL(rccl()->AllGather(mine, wire, (size_t)slice, kBfloat16, c->comm, c->stream));
      hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, c->stream, wire, buf, main);
      GN_LAUNCH_CHECK();
    }
  }
  if (count != main) GN_RCCL(rccl()->AllReduce(buf + main, buf + main, (size_t)(count - main), kFloat32, kSum, c->comm, c->stream));
  GN_HIP(hipEventRecord(c->done, c->stream));
  return GN_OK;
}

extern "C" int32_t gn_comm_wait(gn_comm* c) {
  GN_REQUIRE(c, "gn_comm_wait: null comm");
  GN_HIP(hipStreamWaitEvent(c->ctx->stream, c->done, 0));  // the caller's stream resumes behind the exchange; the host does not block
  return GN_OK;
}
