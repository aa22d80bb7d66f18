// bessx_comm.cpp -- a communicator behind the C ABI (round 6): the one collective the sharded paths need -- an all-gather
// of small fp64 records (the IC / CV curve, the chunks' last models, the fold fits' results) -- on RCCL
// directly, so that a plain C or R host can shard folds or k-chunks over the GPUs of a node without Python or
// torch.distributed.  RCCL is loaded at run time (dlopen: a copy that is already in the process -- torch's -- is reused;
// the library itself does not link against it, a single-GPU host never loads it).  The host distributes the 128-byte
// unique id of rank 0 by whatever it has (a file, MPI, a socket, torch's store).
#include <dlfcn.h>

#include <mutex>

#include "bessx_host.h"

namespace {

struct UniqueId {
  char internal[BESSX_COMM_ID_BYTES];
};
typedef int (*fn_get_id)(UniqueId *);
typedef int (*fn_init_rank)(void **comm, int nranks, UniqueId id, int rank);
typedef int (*fn_allgather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st);
typedef int (*fn_destroy)(void *comm);
typedef const char *(*fn_errstr)(int);

struct Rccl {
  void *lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allgather allgather = nullptr;
  fn_destroy destroy = nullptr, abort = nullptr;
  fn_errstr errstr = nullptr;
  std::string why;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) {  // a copy already loaded into the process first (RTLD_NOLOAD), then the search path
      r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (r.lib) break;
    }
    for (size_t i = 0; !r.lib && i < sizeof(names) / sizeof(names[0]); i++) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) {
      r.why = std::string("RCCL could not be loaded: ") + (dlerror() ? dlerror() : "librccl.so not found");
      return;
    }
    r.get_id = reinterpret_cast<fn_get_id>(dlsym(r.lib, "ncclGetUniqueId"));
    r.init_rank = reinterpret_cast<fn_init_rank>(dlsym(r.lib, "ncclCommInitRank"));
    r.allgather = reinterpret_cast<fn_allgather>(dlsym(r.lib, "ncclAllGather"));
    r.destroy = reinterpret_cast<fn_destroy>(dlsym(r.lib, "ncclCommDestroy"));
    r.abort = reinterpret_cast<fn_destroy>(dlsym(r.lib, "ncclCommAbort"));
    r.errstr = reinterpret_cast<fn_errstr>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.get_id || !r.init_rank || !r.allgather || !r.destroy) r.why = "RCCL: a symbol of the communicator API is missing";
  });
  return &r;
}

int rccl_fail(const char *what, int code) {
  Rccl *r = rccl();
  return bessx::fail(BESSX_ERR_HIP, std::string(what) + ": " + (r->errstr ? r->errstr(code) : "RCCL error") + " (" +
                                        std::to_string(code) + ")");
}

}  // namespace

struct bessx_comm {
  void *comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t st = nullptr;
  double *dsend = nullptr, *drecv = nullptr;
  size_t cap = 0;  // doubles per rank the device buffers hold
};

extern "C" {

int bessx_comm_unique_id(unsigned char *id) {
  using namespace bessx;
  if (!id) return fail(BESSX_ERR_ARG, "comm_unique_id: null buffer");
  Rccl *r = rccl();
  if (!r->why.empty()) return fail(BESSX_ERR_UNSUPPORTED, r->why);
  UniqueId u;
  if (int rc = r->get_id(&u)) return rccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id, u.internal, BESSX_COMM_ID_BYTES);
  return BESSX_OK;
}

int bessx_comm_init(bessx_comm **out, int rank, int world, const unsigned char *id, int device) {
  using namespace bessx;
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(BESSX_ERR_ARG, "comm_init: bad arguments");
  *out = nullptr;
  if (int rc = need_device()) return rc;
  Rccl *r = rccl();
  if (!r->why.empty()) return fail(BESSX_ERR_UNSUPPORTED, r->why);
  HIPX(hipSetDevice(device));
  bessx_comm *c = new bessx_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  UniqueId u;
  std::memcpy(u.internal, id, BESSX_COMM_ID_BYTES);
  if (int rc = r->init_rank(&c->comm, world, u, rank)) {
    delete c;
    return rccl_fail("ncclCommInitRank", rc);
  }
  if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    (void)(r->abort ? r->abort(c->comm) : r->destroy(c->comm));
    delete c;
    return fail(BESSX_ERR_HIP, "comm_init: no stream");
  }
  *out = c;
  return BESSX_OK;
}

int bessx_comm_rank(const bessx_comm *c) { return c ? c->rank : -1; }
int bessx_comm_world(const bessx_comm *c) { return c ? c->world : -1; }

int bessx_comm_allgather_f64(bessx_comm *c, const double *send, int count, double *recv) {
  using namespace bessx;
  if (!c || !send || !recv || count < 1) return fail(BESSX_ERR_ARG, "comm_allgather: bad arguments");
  HIPX(hipSetDevice(c->device));
  if ((size_t)count > c->cap) {
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    c->dsend = c->drecv = nullptr;
    c->cap = 0;
    const size_t cap = (size_t)count + (size_t)count / 2 + 64;
    HIPX(hipMalloc(reinterpret_cast<void **>(&c->dsend), cap * sizeof(double)));
    HIPX(hipMalloc(reinterpret_cast<void **>(&c->drecv), cap * (size_t)c->world * sizeof(double)));
    c->cap = cap;
  }
  HIPX(hipMemcpyAsync(c->dsend, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->st));
  if (int rc = rccl()->allgather(c->dsend, c->drecv, (size_t)count, 8 /* ncclFloat64 */, c->comm, c->st))
    return rccl_fail("ncclAllGather", rc);
  HIPX(hipMemcpyAsync(recv, c->drecv, (size_t)count * (size_t)c->world * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPX(hipStreamSynchronize(c->st));
  return BESSX_OK;
}

void bessx_comm_destroy(bessx_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->st) (void)hipStreamSynchronize(c->st);
  // (every collective this communicator queued has completed -- the stream is idle -- so nothing is lost by aborting;
  // ncclCommDestroy was seen to block for ever here, at world 1 on RCCL 2.27.7, where ncclCommAbort returns)
  if (c->comm) (void)(rccl()->abort ? rccl()->abort(c->comm) : rccl()->destroy(c->comm));
  if (c->dsend) (void)hipFree(c->dsend);
  if (c->drecv) (void)hipFree(c->drecv);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

}  // extern "C"
