// Gradient all-reduce over RCCL / xGMI behind the C-ABI (include/s2t_hip.h: s2t_comm_*, s2t_allreduce_bucket).
//
// Replaces the torch.distributed.all_reduce call of the reference's data-parallel wrapper
// (fairseq/distributed/legacy_distributed_data_parallel.py:107-120 via fairseq/distributed/utils.py all_reduce) with a
// process-global RCCL communicator owned by this library: a plain stream-ordered collective with no watchdog thread
// behind it, so it can be captured into the training step's hipGraph and run on a side stream beside backward.
//
// RCCL is bound at RUN time (dlopen): a process that already carries a librccl (PyTorch ships one) keeps using that
// copy — two different RCCL builds in one process would share symbol names — and a build box without RCCL still links.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                 // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);

enum { NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0, NCCL_AVG = 4 };

struct State {
  void* lib = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn destroy = nullptr;
  Comm comm = nullptr;
  int rank = -1, world = 0;
} g;

int bind() {
  if (g.lib) return S2T_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  // an already loaded copy first (RTLD_NOLOAD), then the search path
  for (int pass = 0; pass < 2 && !g.lib; ++pass)
    for (const char* n : names) {
      g.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g.lib) break;
    }
  if (!g.lib) return S2T_ERR_UNSUPPORTED;
  g.get_id = (GetUniqueIdFn)dlsym(g.lib, "ncclGetUniqueId");
  g.init_rank = (CommInitRankFn)dlsym(g.lib, "ncclCommInitRank");
  g.all_reduce = (AllReduceFn)dlsym(g.lib, "ncclAllReduce");
  g.destroy = (CommDestroyFn)dlsym(g.lib, "ncclCommDestroy");
  if (!g.get_id || !g.init_rank || !g.all_reduce || !g.destroy) {
    g.lib = nullptr;
    return S2T_ERR_UNSUPPORTED;
  }
  return S2T_OK;
}

int nccl_status(int r) { return r == 0 ? S2T_OK : 10000 + r; }  // positive, outside the hipError_t range

}  // namespace

extern "C" int s2t_comm_unique_id(void* out128) {
  if (!out128) return S2T_ERR_ARG;
  const int rc = bind();
  if (rc != S2T_OK) return rc;
  UniqueId id;
  const int r = g.get_id(&id);
  if (r == 0) memcpy(out128, &id, sizeof(id));
  return nccl_status(r);
}

extern "C" int s2t_comm_init(int rank, int world, const void* unique_id128) {
  if (!unique_id128 || world <= 0 || rank < 0 || rank >= world) return S2T_ERR_ARG;
  if (g.comm) return S2T_ERR_UNSUPPORTED;  // one communicator per process (one process per GPU)
  const int rc = bind();
  if (rc != S2T_OK) return rc;
  UniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  const int r = g.init_rank(&g.comm, world, id, rank);
  if (r != 0) {
    g.comm = nullptr;
    return nccl_status(r);
  }
  g.rank = rank;
  g.world = world;
  return S2T_OK;
}

extern "C" int s2t_comm_world(void) { return g.comm ? g.world : 0; }

extern "C" int s2t_allreduce_bucket(void* ptr, int64_t count, int dtype, int average, void* stream) {
  if (!g.comm) return S2T_ERR_UNSUPPORTED;
  if (!ptr || count < 0) return S2T_ERR_ARG;
  if (dtype != S2T_F32 && dtype != S2T_BF16) return S2T_ERR_DTYPE;
  if (count == 0) return S2T_OK;
  return nccl_status(g.all_reduce(ptr, ptr, (size_t)count, dtype == S2T_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16,
                                  average ? NCCL_AVG : NCCL_SUM, g.comm, (hipStream_t)stream));
}

extern "C" int s2t_comm_destroy(void) {
  if (!g.comm) return S2T_OK;
  const int r = g.destroy(g.comm);
  g.comm = nullptr;
  g.rank = -1;
  g.world = 0;
  return nccl_status(r);
}
