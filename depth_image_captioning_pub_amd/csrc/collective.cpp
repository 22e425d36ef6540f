// Gradient exchange of the data-parallel step for callers that bind the C ABI directly (no torch.distributed): a thin wrapper
// over RCCL (include/dic.h, "data parallel").  The path has ONE exchange step - a sum all-reduce of the flat gradient buffer
//  - so this is all a pure-ctypes / cgo / JNI adopter needs next to dic_caption_loss (which already applies the
// 1/N and token-share scaling) and dic_adamw_step.
//
// RCCL is resolved at run time (dlopen), not linked: a process that already holds an RCCL - PyTorch ships its own librccl.so -
// keeps exactly that one, and processes that never exchange gradients do not load it at all.
#include "dic.h"
#include "common.h"
#include <dlfcn.h>
#include <mutex>

namespace dic {
namespace {

// the subset of rccl.h used here (ABI-stable NCCL 2 surface)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
char g_load_err[256] = "";
char g_rccl_path[512] = "";     // file the entry points were resolved from (dladdr): makes a second RCCL in the process visible
bool g_rccl_reused = false;     // true: the copy the process already held (RTLD_NOLOAD) - false: this library loaded one itself

void load_rccl() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)                       // an RCCL this process already holds wins (torch's, if torch is loaded)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) { g_rccl_reused = true; break; }
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) { snprintf(g_load_err, sizeof(g_load_err), "librccl.so not found: %s", dlerror()); return; }
  g_rccl.handle = h;
#define DIC_SYM(field, name) *(void**)(&g_rccl.field) = dlsym(h, name)
  DIC_SYM(GetUniqueId, "ncclGetUniqueId");
  DIC_SYM(CommInitRank, "ncclCommInitRank");
  DIC_SYM(AllReduce, "ncclAllReduce");
  DIC_SYM(CommDestroy, "ncclCommDestroy");
  DIC_SYM(GetErrorString, "ncclGetErrorString");
  DIC_SYM(CommCount, "ncclCommCount");
#undef DIC_SYM
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    snprintf(g_load_err, sizeof(g_load_err), "librccl.so lacks the NCCL 2 entry points");
    g_rccl.handle = nullptr;
    return;
  }
  Dl_info info{};
  if (dladdr((void*)g_rccl.AllReduce, &info) && info.dli_fname) snprintf(g_rccl_path, sizeof(g_rccl_path), "%s", info.dli_fname);
}

int rccl_ready() {
  std::call_once(g_once, load_rccl);
  DIC_REQUIRE(g_rccl.handle != nullptr, "RCCL unavailable: %s", g_load_err);
  return DIC_OK;
}

#define DIC_CHECK_RCCL(expr)                                                                         \
  do {                                                                                               \
    int _r = (expr);                                                                                 \
    if (_r != kNcclSuccess) {                                                                        \
      set_last_error("%s -> RCCL error %d (%s)", #expr, _r, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return DIC_ERR_HIP;                                                                            \
    }                                                                                                \
  } while (0)

}  // namespace
}  // namespace dic

using namespace dic;

struct dic_comm {
  ncclComm_t comm;
  int nranks, rank;
};

extern "C" {

int dic_comm_unique_id(void* id128) {
  DIC_REQUIRE(id128 != nullptr, "dic_comm_unique_id: null buffer (needs DIC_COMM_ID_BYTES bytes)");
  DIC_TRY(rccl_ready());
  ncclUniqueId id;
  DIC_CHECK_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return DIC_OK;
}

int dic_comm_create(const void* id128, int nranks, int rank, dic_comm** out) {
  DIC_REQUIRE(id128 && out && nranks >= 1 && rank >= 0 && rank < nranks, "dic_comm_create: bad arguments");
  DIC_TRY(rccl_ready());
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  DIC_CHECK_RCCL(g_rccl.CommInitRank(&c, nranks, id, rank));      // binds the calling thread's current HIP device
  *out = new dic_comm{c, nranks, rank};
  return DIC_OK;
}

int dic_allreduce_grads(dic_comm* comm, float* flat_grad, long long count, void* stream) {
  DIC_REQUIRE(comm && flat_grad && count > 0, "dic_allreduce_grads: bad arguments");
  DIC_TRY(rccl_ready());
  // in place, sum: the gradients arrive pre-scaled (dic_caption_loss grad_scale = N_r / sum N_r, reg_grad_scale = 1 / ranks),
  // so the sum over the ranks IS the gradient of the global loss.  Enqueued on `stream`, no host synchronisation.
  DIC_CHECK_RCCL(g_rccl.AllReduce(flat_grad, flat_grad, (size_t)count, kNcclFloat32, kNcclSum, comm->comm, (hipStream_t)stream));
  return DIC_OK;
}

int dic_comm_ranks(const dic_comm* comm, int* nranks, int* rank) {
  DIC_REQUIRE(comm != nullptr, "dic_comm_ranks: null communicator");
  if (nranks) *nranks = comm->nranks;
  if (rank) *rank = comm->rank;
  return DIC_OK;
}

int dic_comm_destroy(dic_comm* comm) {
  if (!comm) return DIC_OK;
  DIC_TRY(rccl_ready());
  const int r = g_rccl.CommDestroy(comm->comm);
  delete comm;                       // the handle is gone whatever RCCL answered (no leak on the error path)
  if (r != kNcclSuccess) {
    set_last_error("ncclCommDestroy -> RCCL error %d (%s)", r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return DIC_ERR_HIP;
  }
  return DIC_OK;
}

int dic_comm_info(char* buf, size_t bytes) {
  DIC_REQUIRE(buf && bytes > 0, "dic_comm_info: null buffer");
  DIC_TRY(rccl_ready());
  snprintf(buf, bytes, "rccl=%s;%s", g_rccl_path[0] ? g_rccl_path : "?",
           g_rccl_reused ? "reused (already loaded in this process)" : "loaded by libdic_hip.so (no RCCL was resident)");
  return DIC_OK;
}

}  // extern "C"
