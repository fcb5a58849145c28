// RCCL call site of the library: the gradient all-reduce of the data-parallel step
// (reference: torch DistributedDataParallel's bucketed all-reduce, /root/reference/train.py:218-222, process group
// set up by utils.py:339-375 with backend 'nccl').
//
// One communicator per process (one process per GPU).  The bucket all-reduce is ENQUEUED on the caller's HIP stream
// (the reducer's side stream, event-chained to the backward kernels and to the optimizer kernel), never synchronised
// here.  RCCL is bound lazily with dlopen so that libicamd.so keeps loading on hosts without RCCL (the CPU-only build /
// symbol checks); a missing library is reported as ICAMD_ERR_UNSUPPORTED by every entry point below.
// The rendezvous blob (ncclUniqueId, 128 bytes) is produced by rank 0 and distributed by the host program through
// whatever control plane it has (torch.distributed's store in ddp.py).
#include "../../include/icamd.h"
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

// the slice of rccl.h this file needs (kept local: no build-time dependency on the RCCL headers)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;     // ncclSuccess == 0
enum { NCCL_INT32 = 2, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_BFLOAT16 = 9 };   // ncclDataType_t
enum { NCCL_SUM = 0, NCCL_PROD = 1, NCCL_MAX = 2, NCCL_MIN = 3 };                   // ncclRedOp_t

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl t;
    // an already-loaded copy first (PyTorch-ROCm maps its own librccl.so.1: share it), then the ROCm install
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    t.handle = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (int i = 0; t.handle == nullptr && i < 3; ++i) t.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (t.handle == nullptr) return t;
    t.GetUniqueId = (decltype(t.GetUniqueId))dlsym(t.handle, "ncclGetUniqueId");
    t.CommInitRank = (decltype(t.CommInitRank))dlsym(t.handle, "ncclCommInitRank");
    t.CommDestroy = (decltype(t.CommDestroy))dlsym(t.handle, "ncclCommDestroy");
    t.CommCount = (decltype(t.CommCount))dlsym(t.handle, "ncclCommCount");
    t.CommUserRank = (decltype(t.CommUserRank))dlsym(t.handle, "ncclCommUserRank");
    t.AllReduce = (decltype(t.AllReduce))dlsym(t.handle, "ncclAllReduce");
    t.Broadcast = (decltype(t.Broadcast))dlsym(t.handle, "ncclBroadcast");
    t.GetVersion = (decltype(t.GetVersion))dlsym(t.handle, "ncclGetVersion");
    t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.CommCount && t.CommUserRank && t.AllReduce && t.Broadcast;
    return t;
  }();
  return r;
}

int nccl_dtype(int dtype) {
  switch (dtype) {
    case ICAMD_DT_F32: return NCCL_FLOAT32;
    case ICAMD_DT_I32: return NCCL_INT32;
    case ICAMD_DT_F64: return NCCL_FLOAT64;
    case ICAMD_DT_BF16: return NCCL_BFLOAT16;
    default: return -1;
  }
}

}  // namespace

extern "C" {

int icamd_rccl_available(void) { return rccl().ok ? 1 : 0; }

int icamd_rccl_version(void) {
  int v = 0;
  if (!rccl().ok || rccl().GetVersion == nullptr || rccl().GetVersion(&v) != 0) return 0;
  return v;
}

int icamd_rccl_unique_id(void* id128) {
  if (id128 == nullptr) return ICAMD_ERR_BAD_ARG;
  if (!rccl().ok) return ICAMD_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != 0) return ICAMD_ERR_LAUNCH;
  memcpy(id128, id.internal, sizeof(id.internal));
  return ICAMD_OK;
}

int icamd_rccl_comm_init(const void* id128, int nranks, int rank, void** comm_out) {
  if (id128 == nullptr || comm_out == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return ICAMD_ERR_BAD_ARG;
  if (!rccl().ok) return ICAMD_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  ncclComm_t c = nullptr;
  if (rccl().CommInitRank(&c, nranks, id, rank) != 0 || c == nullptr) return ICAMD_ERR_LAUNCH;
  *comm_out = (void*)c;
  return ICAMD_OK;
}

int icamd_rccl_comm_info(void* comm, int* nranks, int* rank) {
  if (comm == nullptr) return ICAMD_ERR_BAD_ARG;
  if (!rccl().ok) return ICAMD_ERR_UNSUPPORTED;
  int n = 0, r = 0;
  if (rccl().CommCount((ncclComm_t)comm, &n) != 0 || rccl().CommUserRank((ncclComm_t)comm, &r) != 0) return ICAMD_ERR_LAUNCH;
  if (nranks != nullptr) *nranks = n;
  if (rank != nullptr) *rank = r;
  return ICAMD_OK;
}

int icamd_rccl_comm_destroy(void* comm) {
  if (comm == nullptr) return ICAMD_ERR_BAD_ARG;
  if (!rccl().ok) return ICAMD_ERR_UNSUPPORTED;
  return rccl().CommDestroy((ncclComm_t)comm) == 0 ? ICAMD_OK : ICAMD_ERR_LAUNCH;
}

int icamd_allreduce_bucket_launch(void* comm, void* buf, long long count, int dtype, int op, void* stream) {
  if (comm == nullptr || buf == nullptr || count <= 0) return ICAMD_ERR_BAD_ARG;
  const int dt = nccl_dtype(dtype);
  if (dt < 0 || (op != ICAMD_RED_SUM && op != ICAMD_RED_MIN && op != ICAMD_RED_MAX)) return ICAMD_ERR_BAD_ARG;
  if (!rccl().ok) return ICAMD_ERR_UNSUPPORTED;
  const int rop = op == ICAMD_RED_SUM ? NCCL_SUM : (op == ICAMD_RED_MIN ? NCCL_MIN : NCCL_MAX);
  // in place: the bucket IS a slice of the flat gradient arena
  return rccl().AllReduce(buf, buf, (size_t)count, dt, rop, (ncclComm_t)comm, (hipStream_t)stream) == 0 ? ICAMD_OK
                                                                                                          : ICAMD_ERR_LAUNCH;
}

int icamd_broadcast_launch(void* comm, void* buf, long long count, int dtype, int root, void* stream) {
  if (comm == nullptr || buf == nullptr || count <= 0 || root < 0) return ICAMD_ERR_BAD_ARG;
  const int dt = nccl_dtype(dtype);
  if (dt < 0) return ICAMD_ERR_BAD_ARG;
  if (!rccl().ok) return ICAMD_ERR_UNSUPPORTED;
  return rccl().Broadcast(buf, buf, (size_t)count, dt, root, (ncclComm_t)comm, (hipStream_t)stream) == 0 ? ICAMD_OK
                                                                                                          : ICAMD_ERR_LAUNCH;
}

}  // extern "C"
