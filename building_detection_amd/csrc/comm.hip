// libsegengine: sg_comm_* — the one exchange step of the data-parallel path (SURVEY §8 b-2 last line, §8e), RCCL over
// xGMI behind the C ABI.  One process per GPU; rank 0 makes a 128-byte id (sg_comm_unique_id), the host hands it to
// every rank by any side channel (the Python host uses its rendezvous store), every rank calls sg_comm_init, then
// sg_comm_allreduce_sum queues an in-place sum on the stream it is given (the host runs it on a communication stream
// behind an event on the compute stream, so the gradient buckets overlap the rest of backward; dist.py).
//
// RCCL is resolved at run time (dlopen), not linked: a process that already holds a copy of librccl.so.1 (PyTorch
// links one) must not get a second one, and libsegengine stays loadable where no RCCL is installed (the single-GPU
// path never touches it).
#include "sg_common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

// the handful of RCCL declarations used here (rccl.h: ncclUniqueId 128 bytes; ncclInt64 4, ncclFloat32 7,
// ncclBfloat16 9; ncclSum 0; ncclSuccess 0)
struct UniqueId { char internal[SG_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_comm_destroy)(Comm);
typedef const char* (*fn_get_error_string)(int);
typedef int (*fn_get_version)(int*);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_get_error_string get_error_string = nullptr;
  fn_get_version get_version = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {  // a copy the process already holds wins
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return;
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.get_error_string = (fn_get_error_string)dlsym(h, "ncclGetErrorString");
  g_rccl.get_version = (fn_get_version)dlsym(h, "ncclGetVersion");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) return;
  g_rccl.handle = h;  // published last: a non-null handle means every entry point above is resolved
}

// Resolved exactly once per process, whichever thread asks first (std::call_once: concurrent callers wait for the one
// that loads); nullptr when no usable RCCL was found - every caller checks.
const Rccl* rccl() {
  std::call_once(g_rccl_once, rccl_load);
  return g_rccl.handle ? &g_rccl : nullptr;
}

int rccl_fail(const Rccl* r, const char* what, int code) {
  sg_set_error("%s: RCCL error %d (%s)", what, code, r->get_error_string ? r->get_error_string(code) : "?");
  return SG_ECOMM;
}

}  // namespace

struct sg_comm {
  Comm comm;
  int rank, nranks, device;
};

extern "C" {

int sg_comm_probe(int* version_out) {
  // "is there a usable RCCL in this process": dlopen + symbol resolution + ncclGetVersion - no bootstrap root, no listener
  // thread, no socket (ncclGetUniqueId creates all three; only the rank that hands its id out should pay for them)
  const Rccl* r = rccl();
  if (!r) {
    sg_set_error("sg_comm_probe: librccl.so.1 not found (dlopen): %s", dlerror());
    return SG_EUNSUPPORTED;
  }
  int v = 0;
  if (r->get_version) {
    const int rc = r->get_version(&v);
    if (rc != 0) return rccl_fail(r, "ncclGetVersion", rc);
  }
  if (version_out) *version_out = v;
  return 0;
}

int sg_comm_unique_id(void* id_out) {
  SG_CHECK_ARG(id_out != nullptr, "sg_comm_unique_id: null id");
  const Rccl* r = rccl();
  if (!r) {
    sg_set_error("sg_comm_unique_id: librccl.so.1 not found (dlopen): %s", dlerror());
    return SG_EUNSUPPORTED;
  }
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, SG_COMM_ID_BYTES);
  return 0;
}

int sg_comm_init(const void* id, int rank, int nranks, int device, sg_comm** out) {
  SG_CHECK_ARG(id && out, "sg_comm_init: null argument");
  *out = nullptr;
  SG_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "sg_comm_init: rank %d of %d", rank, nranks);
  const Rccl* r = rccl();
  if (!r) {
    sg_set_error("sg_comm_init: librccl.so.1 not found (dlopen): %s", dlerror());
    return SG_EUNSUPPORTED;
  }
  hipError_t e = hipSetDevice(device);  // RCCL binds the communicator to the calling thread's current device
  if (e != hipSuccess) {
    sg_set_error("sg_comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return (int)e;
  }
  UniqueId uid;
  memcpy(uid.internal, id, SG_COMM_ID_BYTES);
  Comm c = nullptr;
  const int rc = r->comm_init_rank(&c, nranks, uid, rank);
  if (rc != 0) return rccl_fail(r, "ncclCommInitRank", rc);
  sg_comm* s = new sg_comm();
  s->comm = c;
  s->rank = rank;
  s->nranks = nranks;
  s->device = device;
  *out = s;
  return 0;
}

int sg_comm_allreduce_sum(sg_comm* comm, void* stream, int dtype, void* buf, int64_t count) {
  SG_CHECK_ARG(comm && (buf || count == 0), "sg_comm_allreduce_sum: null argument");
  SG_CHECK_ARG(count >= 0, "sg_comm_allreduce_sum: count %lld", (long long)count);
  if (count == 0) return 0;
  int nccl_type;
  switch (dtype) {
    case SG_F32: nccl_type = 7; break;
    case SG_BF16: nccl_type = 9; break;
    case SG_I64: nccl_type = 4; break;
    default: sg_set_error("sg_comm_allreduce_sum: dtype %d", dtype); return SG_EINVAL;
  }
  const Rccl* r = rccl();
  if (!r) {  // cannot happen for a communicator sg_comm_init returned, but a null here must not be a crash
    sg_set_error("sg_comm_allreduce_sum: RCCL is not loaded");
    return SG_EUNSUPPORTED;
  }
  const int rc = r->all_reduce(buf, buf, (size_t)count, nccl_type, /*ncclSum*/ 0, comm->comm, (hipStream_t)stream);
  if (rc != 0) return rccl_fail(r, "ncclAllReduce", rc);
  return 0;
}

int sg_comm_rank(const sg_comm* comm) { return comm ? comm->rank : -1; }
int sg_comm_nranks(const sg_comm* comm) { return comm ? comm->nranks : 0; }

int sg_comm_destroy(sg_comm* comm) {
  if (!comm) return 0;
  const Rccl* r = rccl();
  int rc = 0;
  if (r && comm->comm) rc = r->comm_destroy(comm->comm);
  delete comm;
  if (rc != 0) return rccl_fail(r, "ncclCommDestroy", rc);
  return 0;
}

}  // extern "C"
