// C-ABI collective entry points: the final-image all-gather of the ray-tile split for hosts that bind the
// library WITHOUT torch.distributed (a C / C++ caller).  RCCL is looked up at run time (dlopen): the Python host keeps using
// torch.distributed, whose backend "nccl" IS RCCL, and a process then holds exactly one copy of it -- the one already loaded.
//   unique id (rank 0) -> broadcast out of band by the caller -> pgrf_comm_init on every rank -> pgrf_allgather_image per frame
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "pgrf_internal.h"

namespace {

struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(void*) = nullptr;                                   // ncclGetUniqueId(ncclUniqueId*)
  int (*CommInitRank)(void**, int, PgrfUniqueId, int) = nullptr;          // ncclCommInitRank(ncclComm_t*, nranks, id BY VALUE, rank)
  int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;   // ncclAllGather(send, recv, count, dtype, comm, stream)
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;                                     // ncclGetVersion(int*)
};

Rccl g_rccl;
std::mutex g_mu;
char g_why[256] = "unknown";       // why load_rccl() failed (dlerror() may be NULL: never handed to a %s directly)

bool load_rccl() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rccl.so) return true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* so = nullptr;
  for (const char* n : names)                      // a copy that is already in the process (torch's) wins
    if ((so = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!so)
    for (const char* n : names)
      if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!so) {
    const char* e = dlerror();
    std::snprintf(g_why, sizeof(g_why), "librccl.so not found (%s)", e ? e : "unknown");
    return false;
  }
  Rccl r;
  r.so = so;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(so, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(so, "ncclCommInitRank"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(so, "ncclAllGather"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
    std::snprintf(g_why, sizeof(g_why), "the RCCL library in this process lacks %s",
                  !r.GetUniqueId ? "ncclGetUniqueId" : !r.CommInitRank ? "ncclCommInitRank" : !r.AllGather ? "ncclAllGather" : "ncclCommDestroy");
    dlclose(so);
    return false;
  }
  // the id is passed BY VALUE (128 bytes): refuse a library whose ABI this was not written against
  static_assert(sizeof(PgrfUniqueId) == 128, "ncclUniqueId is 128 bytes");
  r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(so, "ncclGetVersion"));
  if (auto ver = r.GetVersion) {
    int v = 0;
    if (ver(&v) == 0 && v > 0 && v < 20000) {          // NCCL_VERSION_CODE = major * 10000 + ...: RCCL 2.x and later
      std::snprintf(g_why, sizeof(g_why), "RCCL version code %d is older than 2.0", v);
      dlclose(so);
      return false;
    }
  }
  g_rccl = r;
  return true;
}

const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "unknown"; }

}  // namespace

using namespace pgrf;

extern "C" {

int pgrf_comm_unique_id(PgrfUniqueId* id) {
  if (!id) return set_error(PGRF_ERR_ARG, "pgrf_comm_unique_id: null pointer");
  if (!load_rccl()) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_comm_unique_id: %s", g_why);
  const int rc = g_rccl.GetUniqueId(id);
  if (rc) return set_error(PGRF_ERR_LAUNCH, "ncclGetUniqueId failed: %s", rccl_err(rc));
  return PGRF_OK;
}

int pgrf_comm_init(const PgrfUniqueId* id, int rank, int nranks, void** comm) {
  if (!id || !comm) return set_error(PGRF_ERR_ARG, "pgrf_comm_init: null pointer");
  if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(PGRF_ERR_ARG, "pgrf_comm_init: rank %d of %d", rank, nranks);
  if (!load_rccl()) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_comm_init: %s", g_why);
  void* c = nullptr;
  const int rc = g_rccl.CommInitRank(&c, nranks, *id, rank);
  if (rc) return set_error(PGRF_ERR_LAUNCH, "ncclCommInitRank failed: %s", rccl_err(rc));
  *comm = c;
  return PGRF_OK;
}

int pgrf_allgather_image(void* comm, const float* shard, float* full, size_t count, void* stream) {
  if (!comm || !shard || !full) return set_error(PGRF_ERR_ARG, "pgrf_allgather_image: null pointer");
  if (!g_rccl.so) return set_error(PGRF_ERR_ARG, "pgrf_allgather_image: no communicator (pgrf_comm_init first)");
  if (count == 0) return PGRF_OK;
  const int kFloat32 = 7;                                      // ncclFloat32
  const int rc = g_rccl.AllGather(shard, full, count, kFloat32, comm, stream);
  if (rc) return set_error(PGRF_ERR_LAUNCH, "ncclAllGather failed: %s", rccl_err(rc));
  return PGRF_OK;
}

int pgrf_comm_rccl_version(int* version) {
  if (!version) return set_error(PGRF_ERR_ARG, "pgrf_comm_rccl_version: null pointer");
  if (!load_rccl()) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_comm_rccl_version: %s", g_why);
  if (!g_rccl.GetVersion) return set_error(PGRF_ERR_UNSUPPORTED, "pgrf_comm_rccl_version: the RCCL library has no ncclGetVersion");
  const int rc = g_rccl.GetVersion(version);
  if (rc) return set_error(PGRF_ERR_LAUNCH, "ncclGetVersion failed: %s", rccl_err(rc));
  return PGRF_OK;
}

int pgrf_comm_destroy(void* comm) {
  if (!comm) return PGRF_OK;
  if (!g_rccl.so) return set_error(PGRF_ERR_ARG, "pgrf_comm_destroy: RCCL not loaded");
  const int rc = g_rccl.CommDestroy(comm);
  if (rc) return set_error(PGRF_ERR_LAUNCH, "ncclCommDestroy failed: %s", rccl_err(rc));
  return PGRF_OK;
}

}  // extern "C"
