// Data-parallel gradient exchange behind the C ABI (vu_dp_init / vu_dp_allreduce_bucket, "thin over RCCL").
// The reference has no collective at all (run_denoising.py:79,87: one 'cuda' device); the default exchange of this build stays
// torch.distributed (backend "nccl" = RCCL), and these entry points are the same sums for a host that does not bring torch:
// one communicator per process (one process per GPU), sum all-reduce of a contiguous range of the flat gradient arena on the
// caller's stream.  RCCL is NOT a link-time dependency of the library: it is opened at vu_dp_init (the copy the process already
// holds - torch's - or librccl.so.1 of the ROCm installation), so the compute path loads and runs without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm image without the RCCL development headers: the handful of types and enum values these entry points pass through (stable
// NCCL ABI: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0, ncclSuccess = 0).
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat = 7, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif
#include <stdio.h>
#include <string.h>
#include "vu_common.h"

namespace {
struct Dp {
  void* lib = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 0;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Dp g_dp;

int open_rccl() {
  if (g_dp.lib) return VU_OK;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // the copy this process already holds (torch's), if any
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { vu_set_error("vu_dp: cannot open librccl: %s", dlerror()); return VU_EUNSUPPORTED; }
  g_dp.GetUniqueId = (decltype(g_dp.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_dp.CommInitRank = (decltype(g_dp.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_dp.AllReduce = (decltype(g_dp.AllReduce))dlsym(h, "ncclAllReduce");
  g_dp.CommDestroy = (decltype(g_dp.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_dp.GetErrorString = (decltype(g_dp.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_dp.GetUniqueId || !g_dp.CommInitRank || !g_dp.AllReduce || !g_dp.CommDestroy || !g_dp.GetErrorString) {
    vu_set_error("vu_dp: librccl lacks an entry point");
    return VU_EUNSUPPORTED;
  }
  g_dp.lib = h;
  return VU_OK;
}
int fail(const char* what, ncclResult_t r) {
  vu_set_error("vu_dp: %s: %s", what, g_dp.GetErrorString ? g_dp.GetErrorString(r) : "?");
  return VU_ELAUNCH;
}
}  // namespace

// 128 bytes for vu_dp_init: rank 0 calls this once and hands the bytes to every rank (any channel: a file, MPI, torch's store)
extern "C" int vu_dp_unique_id(void* out128) {
  if (!out128) { vu_set_error("vu_dp_unique_id: null buffer"); return VU_EINVAL; }
  int rc = open_rccl();
  if (rc != VU_OK) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_dp.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, sizeof(id));
  return VU_OK;
}
// one communicator per process, on the CURRENT HIP device; collective over all `world` ranks (blocks until they have all called)
extern "C" int vu_dp_init(int rank, int world, const void* unique_id_128) {
  if (world < 1 || rank < 0 || rank >= world || !unique_id_128) { vu_set_error("vu_dp_init: bad rank / world / id"); return VU_EINVAL; }
  if (g_dp.comm) { vu_set_error("vu_dp_init: already initialised (vu_dp_finalize first)"); return VU_EINVAL; }
  int rc = open_rccl();
  if (rc != VU_OK) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id_128, sizeof(id));
  const ncclResult_t r = g_dp.CommInitRank(&g_dp.comm, world, id, rank);
  if (r != ncclSuccess) { g_dp.comm = nullptr; return fail("ncclCommInitRank", r); }
  g_dp.rank = rank; g_dp.world = world;
  return VU_OK;
}
// in-place sum over the ranks of ptr[0 .. count): dtype 0 = fp32, 1 = bf16; enqueued on `stream`, no host synchronisation.
// (The 1 / world average of the gradients is applied by vu_adamw's grad_scale, as with the torch.distributed exchange.)
extern "C" int vu_dp_allreduce_bucket(void* ptr, long long count, int dtype, void* stream) {
  if (!g_dp.comm) { vu_set_error("vu_dp_allreduce_bucket: vu_dp_init has not run"); return VU_EINVAL; }
  if (count <= 0) return VU_OK;
  if (!ptr || (dtype != 0 && dtype != 1)) { vu_set_error("vu_dp_allreduce_bucket: null pointer or dtype not in {0: fp32, 1: bf16}"); return VU_EINVAL; }
  const ncclResult_t r = g_dp.AllReduce(ptr, ptr, (size_t)count, dtype == 0 ? ncclFloat32 : ncclBfloat16, ncclSum, g_dp.comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail("ncclAllReduce", r);
  return VU_OK;
}
extern "C" int vu_dp_world(void) { return g_dp.comm ? g_dp.world : 0; }
extern "C" int vu_dp_finalize(void) {
  if (g_dp.comm) {
    const ncclResult_t r = g_dp.CommDestroy(g_dp.comm);
    g_dp.comm = nullptr; g_dp.world = 0; g_dp.rank = 0;
    if (r != ncclSuccess) return fail("ncclCommDestroy", r);
  }
  return VU_OK;
}
