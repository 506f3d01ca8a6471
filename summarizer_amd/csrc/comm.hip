// RCCL behind the C ABI: the data-parallel gradient exchange as a library call (SURVEY.md section 8b lists `allreduce_flat` in
// the minimum export set).  One all-reduce (SUM, in place) of a flat fp32 / bf16 bucket on a caller-given HIP stream over a
// communicator the caller bootstraps: rank 0 asks for a 128-byte unique id, ships it to the other ranks by any means it has
// (the Python host uses its torch.distributed group), every rank calls sumk_comm_init.
// RCCL is loaded with dlopen on first use, NOT linked: a process that already carries an RCCL (PyTorch ships its own librccl)
// keeps exactly one copy, and a single-GPU user of libsumk.so never loads it at all.
#include "sumk_internal.h"
#include <dlfcn.h>
#include <mutex>

namespace {

typedef int (*fn_get_unique_id)(void*);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_get_error_string)(int);

struct UniqueId128 { char internal[128]; };                               // ncclUniqueId (passed BY VALUE to ncclCommInitRank)
typedef int (*fn_comm_init_rank_t)(void**, int, UniqueId128, int);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank_t comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_get_error_string error_string = nullptr;
  char load_error[256] = "symbols missing";   // dlerror() text of the failed dlopen, captured ONCE (dlerror clears itself when read)
};

Rccl g_rccl;

Rccl* rccl() {
  Rccl& r = g_rccl;
  static std::once_flag once;
  std::call_once(once, [&r] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
      if (const char* e = dlerror()) snprintf(r.load_error, sizeof(r.load_error), "%s", e);
    }
    if (!r.handle) return;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank_t)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (fn_get_error_string)dlsym(r.handle, "ncclGetErrorString");
  });
  return (r.handle && r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce) ? &r : nullptr;
}
const char* rccl_load_error() { rccl(); return g_rccl.load_error; }

int rccl_fail(int rc, const char* what) {
  Rccl* r = rccl();
  sumk::set_error("RCCL error %d (%s) in %s", rc, (r && r->error_string) ? r->error_string(rc) : "?", what);
  return SUMK_ERR_HIP;
}

// ncclDataType_t / ncclRedOp_t values of rccl.h (stable since NCCL 2.x): ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0
constexpr int kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclSum = 0;

}  // namespace

extern "C" int sumk_comm_unique_id(uint8_t* id128) {
  using namespace sumk;
  SUMK_ARG(id128 != nullptr, "comm_unique_id: null pointer");
  Rccl* r = rccl();
  SUMK_ARG(r != nullptr, "comm: librccl.so could not be loaded (%s)", rccl_load_error());
  UniqueId128 id;
  int rc = r->get_unique_id(&id);
  if (rc != 0) return rccl_fail(rc, "ncclGetUniqueId");
  for (int i = 0; i < 128; ++i) id128[i] = (uint8_t)id.internal[i];
  return SUMK_OK;
}

extern "C" int sumk_comm_init(const uint8_t* id128, int32_t rank, int32_t world, void** comm_out) {
  using namespace sumk;
  SUMK_ARG(id128 && comm_out, "comm_init: null pointer");
  SUMK_ARG(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
  Rccl* r = rccl();
  SUMK_ARG(r != nullptr, "comm: librccl.so could not be loaded (%s)", rccl_load_error());
  UniqueId128 id;
  for (int i = 0; i < 128; ++i) id.internal[i] = (char)id128[i];
  void* comm = nullptr;
  int rc = r->comm_init_rank(&comm, world, id, rank);     // binds the CURRENT HIP device, like every RCCL communicator
  if (rc != 0) return rccl_fail(rc, "ncclCommInitRank");
  *comm_out = comm;
  return SUMK_OK;
}

extern "C" int sumk_allreduce_flat(void* comm, void* buf, int64_t n, int32_t dtype, void* stream) {
  using namespace sumk;
  SUMK_ARG(comm && buf && n > 0, "allreduce_flat: bad argument");
  SUMK_ARG(dtype == 0 || dtype == 1, "allreduce_flat: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
  Rccl* r = rccl();
  SUMK_ARG(r != nullptr, "comm: librccl.so could not be loaded (%s)", rccl_load_error());
  int rc = r->all_reduce(buf, buf, (size_t)n, dtype == 0 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, comm, (hipStream_t)stream);
  if (rc != 0) return rccl_fail(rc, "ncclAllReduce");
  return SUMK_OK;
}

extern "C" int sumk_comm_destroy(void* comm) {
  using namespace sumk;
  if (!comm) return SUMK_OK;
  Rccl* r = rccl();
  SUMK_ARG(r != nullptr, "comm: librccl.so could not be loaded (%s)", rccl_load_error());
  int rc = r->comm_destroy(comm);
  if (rc != 0) return rccl_fail(rc, "ncclCommDestroy");
  return SUMK_OK;
}
