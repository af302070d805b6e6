// slu_comm_*: thin C-ABI wrappers over RCCL for the one collective of the data-parallel step — the gradient
// all-reduce over the 8 GPUs of a node (xGMI) — so that it can be enqueued on the TRAINING stream between the two
// captured graphs of a step (backward | all-reduce | Adam) without a detour through torch.distributed's own
// stream and Python dispatch.  The reference has no distributed code: new design.
//
// RCCL is not linked: the process already holds a copy (torch's bundled librccl.so, the one backend "nccl" uses),
// and two RCCL instances in one process would each build their own topology / IPC state.  The entry points are
// resolved at run time from the library that is already mapped (dlopen RTLD_NOLOAD only: never a second copy).
#include "slu_common.h"
#include <dlfcn.h>
#include <string.h>

namespace slu {

struct NcclId { char internal[128]; };                      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void* NcclComm;
typedef int (*GetVersionFn)(int*);
typedef int (*GetUniqueIdFn)(NcclId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*GroupFn)(void);

struct Rccl {
  void* handle = nullptr;
  GetVersionFn get_version = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn error_string = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
};

static Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  // ONLY a library that is already mapped (RTLD_NOLOAD): loading a second RCCL copy beside the one torch.distributed
  // uses is exactly what must not happen; a host without torch maps RCCL itself (dlopen RTLD_GLOBAL) before the
  // first slu_comm_* call, else every entry point reports SLU_ERR_UNSUPPORTED
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (r.handle) break;
  }
  if (!r.handle) return nullptr;
  r.get_version = (GetVersionFn)dlsym(r.handle, "ncclGetVersion");
  r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
  r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
  r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
  r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
  r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
  r.group_start = (GroupFn)dlsym(r.handle, "ncclGroupStart");
  r.group_end = (GroupFn)dlsym(r.handle, "ncclGroupEnd");
  if (!r.get_version || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) r.handle = nullptr;
  return r.handle ? &r : nullptr;
}

#define SLU_RCCL(call, what)                                                                        \
  do {                                                                                              \
    const int rc__ = (call);                                                                        \
    if (rc__ != 0) SLU_FAIL(SLU_ERR_HIP, "%s: RCCL error %d (%s)", what, rc__,                     \
                            R->error_string ? R->error_string(rc__) : "?");                         \
  } while (0)

}  // namespace slu

using namespace slu;

extern "C" int slu_comm_version(void) {
  Rccl* R = rccl();
  if (!R) return 0;
  int v = 0;
  return R->get_version(&v) == 0 ? v : 0;
}

extern "C" int slu_comm_unique_id(void* id128) {
  SLU_REQUIRE(id128, "slu_comm_unique_id: null pointer");
  Rccl* R = rccl();
  if (!R) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_comm_unique_id: no RCCL library in this process");
  NcclId id;
  SLU_RCCL(R->get_unique_id(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return SLU_OK;
}

extern "C" int slu_comm_init(void** comm_out, const void* id128, int64_t nranks, int64_t rank) {
  SLU_REQUIRE(comm_out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "slu_comm_init: bad argument");
  Rccl* R = rccl();
  if (!R) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_comm_init: no RCCL library in this process");
  NcclId id;
  memcpy(&id, id128, sizeof(id));
  NcclComm c = nullptr;
  SLU_RCCL(R->comm_init_rank(&c, (int)nranks, id, (int)rank), "ncclCommInitRank");
  *comm_out = c;
  return SLU_OK;
}

static int comm_allreduce(void* comm, void* buf, int64_t count, int dtype, void* stream, const char* who) {
  SLU_REQUIRE(comm && buf && count > 0, "%s: bad argument", who);
  Rccl* R = rccl();
  if (!R) SLU_FAIL(SLU_ERR_UNSUPPORTED, "%s: no RCCL library in this process", who);
  SLU_RCCL(R->all_reduce(buf, buf, (size_t)count, dtype, /*ncclSum*/ 0, (NcclComm)comm, (hipStream_t)stream), who);
  return SLU_OK;
}

extern "C" int slu_comm_allreduce_f32(void* comm, float* buf, int64_t count, void* stream) {
  return comm_allreduce(comm, buf, count, /*ncclFloat32*/ 7, stream, "slu_comm_allreduce_f32");
}

extern "C" int slu_comm_allreduce_f64(void* comm, double* buf, int64_t count, void* stream) {
  return comm_allreduce(comm, buf, count, /*ncclFloat64*/ 8, stream, "slu_comm_allreduce_f64");
}

// Both gradient buckets of a step (fp32 + the Sinc layer's 160 float64 values) as ONE grouped RCCL operation: one
// launch on the stream instead of two dependent collectives.  Either bucket may be absent (count 0).
extern "C" int slu_comm_allreduce_group(void* comm, float* f32, int64_t n32, double* f64, int64_t n64, void* stream) {
  SLU_REQUIRE(comm && n32 >= 0 && n64 >= 0 && n32 + n64 > 0 && (n32 == 0 || f32) && (n64 == 0 || f64),
              "slu_comm_allreduce_group: bad argument");
  Rccl* R = rccl();
  if (!R) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_comm_allreduce_group: no RCCL library in this process");
  const bool group = n32 > 0 && n64 > 0 && R->group_start && R->group_end;
  if (group) SLU_RCCL(R->group_start(), "ncclGroupStart");
  int rc = 0;
  if (n32 > 0) rc = R->all_reduce(f32, f32, (size_t)n32, /*ncclFloat32*/ 7, /*ncclSum*/ 0, (NcclComm)comm, (hipStream_t)stream);
  if (rc == 0 && n64 > 0) rc = R->all_reduce(f64, f64, (size_t)n64, /*ncclFloat64*/ 8, 0, (NcclComm)comm, (hipStream_t)stream);
  if (group) {
    const int rc_end = R->group_end();
    if (rc == 0) rc = rc_end;
  }
  SLU_RCCL(rc, "slu_comm_allreduce_group");
  return SLU_OK;
}

extern "C" int slu_comm_destroy(void* comm) {
  if (!comm) return SLU_OK;
  Rccl* R = rccl();
  if (!R) SLU_FAIL(SLU_ERR_UNSUPPORTED, "slu_comm_destroy: no RCCL library in this process");
  SLU_RCCL(R->comm_destroy((NcclComm)comm), "ncclCommDestroy");
  return SLU_OK;
}
