// comm.hip — st355_comm_*: the gradient exchange of the data-parallel replicas as C-ABI entry points over RCCL.
//
// The reference has no such seam: its exchange is torch DDP's reducer over NCCL/RCCL (trainer.py:1034-1041, 4564-4571).  The Python host layer
// (training/grad_sync.py) drives the same collectives through torch.distributed; these entry points are the binding a non-torch host would use,
// with the conventions of the rest of the library: caller's hipStream_t, device pointers, int status, nothing allocated, no synchronisation.
//   * one communicator per process (one process per GPU); the 128-byte unique id is created on rank 0 (st355_comm_unique_id) and carried to the
//     other ranks by the host (any side channel: the launcher's TCP store, a file, MPI);
//   * in-place forms are allowed exactly as RCCL allows them (reduce_scatter: recv == send + rank*recv_count; all_gather: send == recv + rank*send_count),
//     which is what the flat gradient arena uses (grad_sync.py `rs_ag`);
//   * SUM only; the 1/world averaging is folded into the optimizer kernels' grad_scale.
// librccl is resolved at FIRST USE with dlopen (the copy already mapped into the process — e.g. torch's — if there is one, else the ROCm one): the
// library keeps no link-time dependency on RCCL, and every st355_comm_* call returns ST355_ENOSYS with a message when RCCL cannot be loaded.
#include <dlfcn.h>
#include <mutex>
#include "common.h"

namespace {
typedef struct { char internal[128]; } RcclUniqueId;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;
enum { kRcclFloat32 = 7, kRcclBfloat16 = 9, kRcclSum = 0 };      // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Api g_api;
std::once_flag g_once;

void load_api() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) {                                   // the copy that is already mapped (torch's own librccl), if any
    g_api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (g_api.lib) break;
  }
  if (!g_api.lib) {
    const char* paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : paths) {
      g_api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (g_api.lib) break;
    }
  }
  if (!g_api.lib) return;
#define SYM(field, name) *(void**)(&g_api.field) = dlsym(g_api.lib, name)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce"); SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce && g_api.ReduceScatter && g_api.AllGather;
}

int need_api(const char* what) {
  std::call_once(g_once, load_api);
  if (!g_api.ok) {
    st355_set_error("%s: librccl could not be loaded (%s)", what, g_api.lib ? "symbols missing" : "dlopen failed");
    return ST355_ENOSYS;
  }
  return ST355_OK;
}
int rc_of(int r, const char* what) {
  if (r == 0) return ST355_OK;
  st355_set_error("%s: RCCL error %d (%s)", what, r, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
  return ST355_EFAULT;
}
int dtype_of(int elem_kind, int* out) {                            // 0 = fp32, 1 = bf16 (the two gradient-arena types)
  if (elem_kind == 0) { *out = kRcclFloat32; return ST355_OK; }
  if (elem_kind == 1) { *out = kRcclBfloat16; return ST355_OK; }
  st355_set_error("st355_comm: element kind %d (0 = fp32, 1 = bf16)", elem_kind);
  return ST355_EINVAL;
}
}  // namespace

extern "C" int st355_comm_unique_id(void* id128) {
  ST_REQUIRE(id128, "comm_unique_id: null pointer");
  int rc = need_api("comm_unique_id");
  if (rc) return rc;
  return rc_of(g_api.GetUniqueId((RcclUniqueId*)id128), "comm_unique_id");
}
extern "C" int st355_comm_init(void** comm, const void* id128, int world, int rank) {
  ST_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args (world %d rank %d)", world, rank);
  int rc = need_api("comm_init");
  if (rc) return rc;
  RcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  RcclComm c = nullptr;
  rc = rc_of(g_api.CommInitRank(&c, world, id, rank), "comm_init");      // on the CURRENT device of the calling thread
  if (rc) return rc;
  *comm = c;
  return ST355_OK;
}
extern "C" int st355_comm_destroy(void* comm) {
  ST_REQUIRE(comm, "comm_destroy: null communicator");
  int rc = need_api("comm_destroy");
  if (rc) return rc;
  return rc_of(g_api.CommDestroy((RcclComm)comm), "comm_destroy");
}
extern "C" int st355_comm_all_reduce(void* comm, void* stream, void* buf, int64_t count, int elem_kind) {
  ST_REQUIRE(comm && buf && count > 0, "comm_all_reduce: bad args");
  int dt, rc = need_api("comm_all_reduce");
  if (rc || (rc = dtype_of(elem_kind, &dt))) return rc;
  return rc_of(g_api.AllReduce(buf, buf, (size_t)count, dt, kRcclSum, (RcclComm)comm, (hipStream_t)stream), "comm_all_reduce");
}
extern "C" int st355_comm_reduce_scatter(void* comm, void* stream, const void* send, void* recv, int64_t recv_count, int elem_kind) {
  ST_REQUIRE(comm && send && recv && recv_count > 0, "comm_reduce_scatter: bad args");
  int dt, rc = need_api("comm_reduce_scatter");
  if (rc || (rc = dtype_of(elem_kind, &dt))) return rc;
  return rc_of(g_api.ReduceScatter(send, recv, (size_t)recv_count, dt, kRcclSum, (RcclComm)comm, (hipStream_t)stream), "comm_reduce_scatter");
}
extern "C" int st355_comm_all_gather(void* comm, void* stream, const void* send, void* recv, int64_t send_count, int elem_kind) {
  ST_REQUIRE(comm && send && recv && send_count > 0, "comm_all_gather: bad args");
  int dt, rc = need_api("comm_all_gather");
  if (rc || (rc = dtype_of(elem_kind, &dt))) return rc;
  return rc_of(g_api.AllGather(send, recv, (size_t)send_count, dt, (RcclComm)comm, (hipStream_t)stream), "comm_all_gather");
}
