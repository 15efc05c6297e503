// Data-parallel gradient exchange behind the C-ABI (fdmi_allreduce_init / fdmi_allreduce): ONE in-place sum
// all-reduce (RCCL over xGMI) of the flat LoRA gradient per optimizer step -- what Lightning's DDP strategy does for the
// reference (examples/train_flash_sd.py:383-386) with bucketed NCCL calls.  One process per GPU, one communicator per
// process.  librccl is bound at run time (dlopen, preferring a copy that is already loaded -- a PyTorch host has one) so
// that libfdmi.so itself carries no load-time dependency on it and single-GPU users never touch it.
// The Python host (trainer.py) keeps using torch.distributed, whose "nccl" backend is the same RCCL; these entry points
// are for hosts without torch.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"
#include "../../include/fdmi.h"

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0;
std::mutex g_mu;

int bind_rccl() {
  if (g_rccl.h) return 0;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already holds (torch ships one)
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  const char* why = h ? nullptr : dlerror();   // (dlerror() clears the message: read it once)
  FDMI_CHECK(h != nullptr, std::string("allreduce: cannot load librccl.so: ") + (why ? why : "?"));
  Rccl r;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  FDMI_CHECK(r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GetErrorString,
             "allreduce: librccl.so lacks an expected symbol");
  g_rccl = r;
  return 0;
}

#define FDMI_NCCL(expr)                                                                             \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    FDMI_CHECK(_r == ncclSuccess, std::string(#expr) + ": " + g_rccl.GetErrorString(_r));           \
  } while (0)
}  // namespace

extern "C" {

int fdmi_comm_unique_id(void* out128) {
  std::lock_guard<std::mutex> lk(g_mu);
  FDMI_CHECK(out128 != nullptr, "comm_unique_id: null buffer");
  if (bind_rccl()) return -1;
  ncclUniqueId id;
  FDMI_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int fdmi_allreduce_init(int rank, int world, const void* uid128) {
  std::lock_guard<std::mutex> lk(g_mu);
  FDMI_CHECK(world >= 1 && rank >= 0 && rank < world && uid128, "allreduce_init: bad rank / world / id");
  FDMI_CHECK(g_comm == nullptr, "allreduce_init: communicator already initialised (call fdmi_allreduce_destroy first)");
  if (bind_rccl()) return -1;
  ncclUniqueId id;
  memcpy(id.internal, uid128, NCCL_UNIQUE_ID_BYTES);
  FDMI_NCCL(g_rccl.CommInitRank(&g_comm, world, id, rank));
  g_world = world;
  return 0;
}

int fdmi_allreduce(void* buf, int64_t count, int dtype, void* stream) {
  FDMI_CHECK(g_comm != nullptr, "allreduce: call fdmi_allreduce_init first");
  FDMI_CHECK(buf != nullptr && count >= 0 && (dtype == FDMI_F32 || dtype == FDMI_BF16), "allreduce: bad argument");
  if (count == 0) return 0;
  FDMI_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == FDMI_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, g_comm,
                             (hipStream_t)stream));
  return 0;
}

int fdmi_allreduce_world(void) { return g_world; }

int fdmi_allreduce_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_comm) {
    FDMI_NCCL(g_rccl.CommDestroy(g_comm));
    g_comm = nullptr;
    g_world = 0;
  }
  return 0;
}

}  // extern "C"
