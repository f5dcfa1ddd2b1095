// comm.cpp — the data-parallel exchange of the hot path as C-ABI entry points over RCCL (xGMI on an MI355X node).
// The reference has no multi-GPU code; the build adds exactly ONE collective per training step: an in-place
// sum all-reduce of the flat fp32 bucket (gradients + scalar tail, srl-zoo_amd/srlz/optim.py).  One communicator per
// process (one process per GPU), created from a 128-byte unique id that rank 0 generates and the host shares by whatever
// rendezvous it has (the Python side uses torch.distributed's store).
// RCCL is resolved at srlz_comm_init() time with dlopen — the copy already loaded into the process (PyTorch ships one) is
// preferred, so the two never coexist — and the library keeps BUILDING and loading where RCCL (headers or library) is
// absent: the handful of NCCL-ABI types this file needs are declared here instead of including <rccl/rccl.h>.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

// ---- the slice of the NCCL / RCCL C ABI used below (stable since NCCL 2.0; values as in rccl.h) ----
struct ncclComm;
typedef ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;      // enum in the header; ncclSuccess == 0
typedef int ncclDataType_t;    // enum in the header
typedef int ncclRedOp_t;       // enum in the header
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclFloat32 = 7;
constexpr ncclRedOp_t ncclSum = 0;

struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} api;

ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;

int load_api() {
  if (api.handle) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the process's own copy, if any
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  SRLZ_REQUIRE(h != nullptr, SRLZ_ERR_HIP, "srlz_comm: cannot load librccl.so.1 (%s)", dlerror());
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  SRLZ_REQUIRE(api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.GetErrorString, SRLZ_ERR_HIP,
               "srlz_comm: librccl lacks an expected symbol");
  api.handle = h;
  return 0;
}

#define SRLZ_NCCL(expr)                                                                      \
  do {                                                                                       \
    ncclResult_t _r = (expr);                                                                \
    SRLZ_REQUIRE(_r == ncclSuccess, SRLZ_ERR_HIP, "%s failed: %s", #expr, api.GetErrorString(_r)); \
  } while (0)

}  // namespace

extern "C" size_t srlz_comm_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

extern "C" int srlz_comm_unique_id(void* id_out_host) {
  SRLZ_REQUIRE(id_out_host, SRLZ_ERR_NULL, "srlz_comm_unique_id: null pointer");
  if (int rc = load_api()) return rc;
  ncclUniqueId id;
  SRLZ_NCCL(api.GetUniqueId(&id));
  memcpy(id_out_host, &id, sizeof(id));
  return 0;
}

extern "C" int srlz_comm_init(const void* id_host, int rank, int world) {
  SRLZ_REQUIRE(id_host, SRLZ_ERR_NULL, "srlz_comm_init: null pointer");
  SRLZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, SRLZ_ERR_BAD_DESC, "srlz_comm_init: rank %d of %d", rank, world);
  SRLZ_REQUIRE(g_comm == nullptr, SRLZ_ERR_BAD_DESC, "srlz_comm_init: the process already owns a communicator");
  if (int rc = load_api()) return rc;
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  SRLZ_NCCL(api.CommInitRank(&g_comm, world, id, rank));   // on the calling thread's current HIP device
  g_rank = rank; g_world = world;
  return 0;
}

extern "C" int srlz_comm_world(void) { return g_comm ? g_world : 0; }

extern "C" int srlz_comm_allreduce_f32(float* buf, long long n, srlz_stream_t stream) {
  SRLZ_REQUIRE(g_comm != nullptr, SRLZ_ERR_BAD_DESC, "srlz_comm_allreduce_f32: srlz_comm_init was not called");
  SRLZ_REQUIRE(buf && n > 0, SRLZ_ERR_NULL, "srlz_comm_allreduce_f32: empty buffer");
  SRLZ_NCCL(api.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, g_comm, as_stream(stream)));
  return 0;
}

extern "C" int srlz_comm_destroy(void) {
  if (g_comm) {
    SRLZ_NCCL(api.CommDestroy(g_comm));
    g_comm = nullptr;
    g_world = 1; g_rank = 0;
  }
  return 0;
}
