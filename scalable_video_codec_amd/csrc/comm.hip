// comm.hip -- the one cross-rank step of the path: the neighbour shift of a packed Y
// pyramid, rank r -> r + 1, as an RCCL send/recv group on the caller's stream.
//
// The reference is a single process; its only cross-frame state is the previous source frame's
// pyramid (libs/encoder.cpp:661-663), which is exactly what crosses ranks here.  RCCL is bound
// at run time with dlopen: a process that already holds a copy (PyTorch ships one under the same
// soname) gets that copy, a plain C++ host gets ROCm's; a host that never shards never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "svc_common.hpp"

namespace svc {
namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = "";
};

template <typename F> bool bind(void* lib, const char* name, F* out, char* why, size_t why_sz) {
  *out = reinterpret_cast<F>(dlsym(lib, name));
  if (!*out) snprintf(why, why_sz, "librccl has no symbol %s", name);
  return *out != nullptr;
}

Rccl g_rccl;

Rccl* rccl() {
  Rccl& r = g_rccl;
  static std::once_flag once;
  std::call_once(once, [&r] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      snprintf(r.why, sizeof r.why, "cannot load librccl.so.1: %s", dlerror());
      return;
    }
    const bool ok = bind(r.lib, "ncclGetUniqueId", &r.GetUniqueId, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclCommInitRank", &r.CommInitRank, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclCommDestroy", &r.CommDestroy, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclCommCount", &r.CommCount, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclCommUserRank", &r.CommUserRank, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclCommCuDevice", &r.CommCuDevice, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclGroupStart", &r.GroupStart, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclGroupEnd", &r.GroupEnd, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclSend", &r.Send, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclRecv", &r.Recv, r.why, sizeof r.why) &&
                    bind(r.lib, "ncclGetErrorString", &r.GetErrorString, r.why, sizeof r.why);
    if (!ok) r.lib = nullptr;
  });
  return r.lib ? &r : nullptr;
}

int no_rccl() { return fail(SVC_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why); }

#define SVC_NCCL_TRY(r, expr)                                                                     \
  do {                                                                                            \
    ncclResult_t e_ = (expr);                                                                     \
    if (e_ != ncclSuccess)                                                                        \
      return ::svc::fail(SVC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (r)->GetErrorString(e_),    \
                         __FILE__, __LINE__);                                                     \
  } while (0)

}  // namespace
}  // namespace svc

using namespace svc;

extern "C" {

int svc_hip_comm_available(void) {
  return rccl() ? SVC_OK : no_rccl();
}

int svc_hip_comm_info(void* comm, uint32_t* ranks, uint32_t* rank, int32_t* device) {
  SVC_REQUIRE(comm, "comm_info: null communicator");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  int v = 0;
  if (ranks) { SVC_NCCL_TRY(r, r->CommCount(c, &v)); *ranks = (uint32_t)v; }
  if (rank) { SVC_NCCL_TRY(r, r->CommUserRank(c, &v)); *rank = (uint32_t)v; }
  if (device) { SVC_NCCL_TRY(r, r->CommCuDevice(c, &v)); *device = v; }
  return SVC_OK;
}

int svc_hip_comm_unique_id(uint8_t id[SVC_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == SVC_COMM_ID_BYTES, "ncclUniqueId size");
  SVC_REQUIRE(id, "comm_unique_id: null output");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  ncclUniqueId u;
  SVC_NCCL_TRY(r, r->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return SVC_OK;
}

int svc_hip_comm_create(const uint8_t id[SVC_COMM_ID_BYTES], uint32_t rank, uint32_t world, void** comm) {
  SVC_REQUIRE(id && comm, "comm_create: null pointer");
  SVC_REQUIRE(world >= 1 && rank < world, "comm_create: rank %u of %u", rank, world);
  Rccl* r = rccl();
  if (!r) return no_rccl();
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclComm_t c = nullptr;
  SVC_NCCL_TRY(r, r->CommInitRank(&c, (int)world, u, (int)rank));
  *comm = c;
  return SVC_OK;
}

int svc_hip_comm_destroy(void* comm) {
  if (!comm) return SVC_OK;
  Rccl* r = rccl();
  if (!r) return no_rccl();
  SVC_NCCL_TRY(r, r->CommDestroy(static_cast<ncclComm_t>(comm)));
  return SVC_OK;
}

int svc_hip_halo_shift(void* comm, const uint8_t* d_send, uint8_t* d_recv, uint64_t bytes, uint32_t rank,
                       uint32_t world, uint32_t flags, void* stream) {
  SVC_REQUIRE(world >= 1 && rank < world, "halo_shift: rank %u of %u", rank, world);
  const bool cyclic = (flags & SVC_SHIFT_CYCLIC) != 0;
  const bool sends = rank + 1 < world || cyclic, recvs = rank > 0 || cyclic;
  if (bytes == 0 || (!sends && !recvs)) return SVC_OK;  // a single rank has no neighbour
  SVC_REQUIRE(comm, "halo_shift: null communicator");
  SVC_REQUIRE((!sends || d_send) && (!recvs || d_recv), "halo_shift: null buffer");
  Rccl* r = rccl();
  if (!r) return no_rccl();
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one group: the send and the receive of a middle rank must progress together
  SVC_NCCL_TRY(r, r->GroupStart());
  ncclResult_t es = ncclSuccess, er = ncclSuccess;
  if (sends) es = r->Send(d_send, bytes, ncclUint8, (int)((rank + 1) % world), c, s);
  if (recvs) er = r->Recv(d_recv, bytes, ncclUint8, (int)((rank + world - 1) % world), c, s);
  ncclResult_t eg = r->GroupEnd();  // always closed, also after a failed enqueue
  SVC_NCCL_TRY(r, es);
  SVC_NCCL_TRY(r, er);
  SVC_NCCL_TRY(r, eg);
  return SVC_OK;
}

}  // extern "C"
