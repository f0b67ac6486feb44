// transport_rccl.cpp -- in-library transport of a distributed context: RCCL (over xGMI inside a
// node) called directly on the context's HIP stream.  One grouped ncclSend/ncclRecv launch per
// ghost exchange, one ncclAllReduce per fused dot product; no host synchronisation and no
// callback into the host language.  Fills the same agmg_comm the host could bind itself
// (include/agmg.h); replaces the MPI calls of deal.II's distributed vectors on this path
// (SURVEY.md 8e).  RCCL is bound at run time (dlopen) so the library also loads where RCCL is absent.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "agmg_internal.h"

namespace agmg {
namespace {

struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi &api() {
  static RcclApi a;
  if (a.ok || a.h) return a;
  // prefer the RCCL the process already holds (e.g. the one PyTorch loaded): one instance per process
  for (const char *name : {"librccl.so.1", "librccl.so"}) {
    a.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (a.h) break;
  }
  if (!a.h)
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      a.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.h) break;
    }
  if (!a.h) return a;
  auto sym = [&](const char *n) { return dlsym(a.h, n); };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
  a.Send = (decltype(a.Send))sym("ncclSend");
  a.Recv = (decltype(a.Recv))sym("ncclRecv");
  a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv &&
         a.AllReduce && a.GetErrorString;
  return a;
}

struct RcclState {
  ncclComm_t comm = nullptr;
};

int fail(const char *what, ncclResult_t r) {
  set_error(std::string(what) + ": " + api().GetErrorString(r));
  return -102;
}

int rccl_exchange(void *user, int32_t n, const int32_t *peer, const double *const *send, const int64_t *n_send,
                  double *const *recv, const int64_t *n_recv, void *stream) {
  RcclState *st = static_cast<RcclState *>(user);
  RcclApi &a = api();
  hipStream_t s = static_cast<hipStream_t>(stream);
  ncclResult_t r = a.GroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart", r);
  for (int i = 0; i < n && r == ncclSuccess; ++i) {
    if (n_recv[i] > 0) r = a.Recv(recv[i], (size_t)n_recv[i], ncclFloat64, peer[i], st->comm, s);
    if (r == ncclSuccess && n_send[i] > 0) r = a.Send(send[i], (size_t)n_send[i], ncclFloat64, peer[i], st->comm, s);
  }
  const ncclResult_t e = a.GroupEnd();
  if (r != ncclSuccess) return fail("ncclSend/ncclRecv", r);
  if (e != ncclSuccess) return fail("ncclGroupEnd", e);
  return 0;
}

int rccl_allreduce(void *user, double *values, int32_t n, void *stream) {
  RcclState *st = static_cast<RcclState *>(user);
  const ncclResult_t r = api().AllReduce(values, values, (size_t)n, ncclFloat64, ncclSum, st->comm,
                                         static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : fail("ncclAllReduce", r);
}

}  // namespace

int rccl_unique_id(uint8_t id[128]) {
  RcclApi &a = api();
  if (!a.ok) { set_error("RCCL not available (librccl.so could not be bound)"); return -102; }
  static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
  ncclUniqueId u;
  const ncclResult_t r = a.GetUniqueId(&u);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  std::memcpy(id, &u, 128);
  return 0;
}

int rccl_comm_create(const uint8_t id[128], int rank, int size, agmg_comm *out, void **state) {
  RcclApi &a = api();
  if (!a.ok) { set_error("RCCL not available (librccl.so could not be bound)"); return -102; }
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  RcclState *st = new RcclState();
  const ncclResult_t r = a.CommInitRank(&st->comm, size, u, rank);
  if (r != ncclSuccess) { delete st; return fail("ncclCommInitRank", r); }
  out->rank = rank;
  out->size = size;
  out->device_buffers = 1;
  out->user = st;
  out->exchange = rccl_exchange;
  out->allreduce_sum = rccl_allreduce;
  *state = st;
  return 0;
}

void rccl_comm_destroy(void *state) {
  RcclState *st = static_cast<RcclState *>(state);
  if (!st) return;
  if (st->comm && api().ok) (void)api().CommDestroy(st->comm);
  delete st;
}

}  // namespace agmg
