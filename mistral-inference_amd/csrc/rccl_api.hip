// RCCL transport of the pipeline-parallel exchange steps (reference transformer.py:196,214,237: torch.distributed
// recv / send / broadcast; main.py:110-118 process group).  Stream-ordered ncclSend / ncclRecv / ncclBroadcast on the
// caller's HIP stream - no host synchronisation, capturable in a hipGraph together with the decode step - over one
// communicator per process (one process per GPU; xGMI between the GPUs of a node).
//
// librccl is resolved with dlopen at the first mi_rccl_* call, not linked: libmistral_hip.so loads (and every
// single-GPU path works) on a box without RCCL, and a process that already carries a RCCL (PyTorch's) reuses it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mistral_hip.h"

namespace {

// The subset of rccl.h used here (stable NCCL 2 ABI).
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0 };

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char error[256] = "";
};

Api& api() {
  static Api a;
  return a;
}

thread_local char g_rccl_detail[384] = "";

int load() {
  Api& a = api();
  if (a.handle) return MI_OK;
  const char* names[] = {getenv("MI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (a.handle) break;
  }
  if (!a.handle) {
    snprintf(g_rccl_detail, sizeof(g_rccl_detail), "librccl not found (%s)", dlerror());
    return MI_ERR_UNSUPPORTED;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(a.handle, name);
    if (!p) ok = false;
    return p;
  };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.Send = (decltype(a.Send))sym("ncclSend");
  a.Recv = (decltype(a.Recv))sym("ncclRecv");
  a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
  a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  if (!ok) {
    snprintf(g_rccl_detail, sizeof(g_rccl_detail), "librccl lacks a required symbol");
    dlclose(a.handle);
    a.handle = nullptr;
    return MI_ERR_UNSUPPORTED;
  }
  return MI_OK;
}

int rc(int r, const char* what) {
  if (r == ncclSuccess) return MI_OK;
  snprintf(g_rccl_detail, sizeof(g_rccl_detail), "%s: %s", what, api().GetErrorString ? api().GetErrorString(r) : "rccl error");
  return MI_ERR_RCCL;
}

}  // namespace

struct mi_rccl_comm {
  ncclComm_t comm;
  int world, rank;
};

extern "C" {

const char* mi_rccl_last_error(void) { return g_rccl_detail; }

int mi_rccl_unique_id(void* id128) {
  if (!id128) return MI_ERR_ARG;
  if (int e = load()) return e;
  ncclUniqueId id;
  if (int e = rc(api().GetUniqueId(&id), "ncclGetUniqueId")) return e;
  memcpy(id128, id.internal, sizeof(id.internal));
  return MI_OK;
}

int mi_rccl_init(mi_rccl_t* comm, int world_size, int rank, const void* id128) {
  if (!comm || !id128 || world_size <= 0 || rank < 0 || rank >= world_size) return MI_ERR_ARG;
  if (int e = load()) return e;
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  mi_rccl_comm* c = new mi_rccl_comm{nullptr, world_size, rank};
  if (int e = rc(api().CommInitRank(&c->comm, world_size, id, rank), "ncclCommInitRank")) {
    delete c;
    return e;
  }
  *comm = c;
  return MI_OK;
}

int mi_rccl_destroy(mi_rccl_t comm) {
  if (!comm) return MI_ERR_ARG;
  const int e = rc(api().CommDestroy(comm->comm), "ncclCommDestroy");
  delete comm;
  return e;
}

int mi_rccl_send(mi_rccl_t comm, const void* buf, size_t bytes, int peer, mi_stream_t stream) {
  if (!comm || !buf || peer < 0 || peer >= comm->world) return MI_ERR_ARG;
  return rc(api().Send(buf, bytes, ncclInt8, peer, comm->comm, (hipStream_t)stream), "ncclSend");
}

int mi_rccl_recv(mi_rccl_t comm, void* buf, size_t bytes, int peer, mi_stream_t stream) {
  if (!comm || !buf || peer < 0 || peer >= comm->world) return MI_ERR_ARG;
  return rc(api().Recv(buf, bytes, ncclInt8, peer, comm->comm, (hipStream_t)stream), "ncclRecv");
}

int mi_rccl_bcast(mi_rccl_t comm, void* buf, size_t bytes, int root, mi_stream_t stream) {
  if (!comm || !buf || root < 0 || root >= comm->world) return MI_ERR_ARG;
  return rc(api().Broadcast(buf, buf, bytes, ncclInt8, root, comm->comm, (hipStream_t)stream), "ncclBroadcast");
}

int mi_rccl_group_start(void) {
  if (int e = load()) return e;
  return rc(api().GroupStart(), "ncclGroupStart");
}

int mi_rccl_group_end(void) {
  if (int e = load()) return e;
  return rc(api().GroupEnd(), "ncclGroupEnd");
}

}  // extern "C"
