// The single collective of the multi-GPU path: a gather of every rank's (boxes, scores, tokens) records on rank 0
// over RCCL point-to-point (one direct xGMI hop per peer).
//
// The reference is single-device (densecap/utils.lua:22-36 binds one GPU; run_model.lua:160-180 loops over images on
// it); images shard by index with no data-path exchange, so this is the only communication of the path
// (BASELINE.json north_star).  RCCL has no native gather: rank 0 posts world-1 ncclRecv, every other
// rank one ncclSend, inside one ncclGroupStart/ncclGroupEnd -- every peer rides its own link to rank 0.
//
// librccl is opened with dlopen() on the first dc_comm_create: the single-GPU path of libdensecap_hip.so keeps no
// dependency on it.  Only the types come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) { api.err = std::string("dlopen(librccl.so) failed: ") + dlerror(); return; }
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(api.handle, n);
      if (!p && api.err.empty()) api.err = std::string("librccl: missing symbol ") + n;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &api;
}

thread_local std::string g_comm_error;

}  // namespace

struct dc_comm {
  dc_ctx* ctx = nullptr;
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  void* dev_buf = nullptr;     // rank 0: world records blocks; others: one
  size_t dev_bytes = 0;
  void* host_buf = nullptr;    // pinned staging, same size
  size_t host_bytes = 0;
  std::string err;
  int fail(int code, const std::string& msg) {
    err = msg;
    g_comm_error = msg;
    if (ctx) dc_ctx_set_error(ctx, msg.c_str());
    return code;
  }
};

// One image's record: {int32 K, int32 T, int32 capacity, int32 0; float boxes[cap][4]; float scores[cap];
// int32 tokens[cap][T]} -- typed fields at fixed offsets, K inside the record (one message per rank).
static size_t record_bytes(int capacity, int T) { return 16 + (size_t)capacity * (16 + 4 + 4 * (size_t)T); }

extern "C" {

int dc_comm_unique_id(void* id_out) {
  if (!id_out) { g_comm_error = "dc_comm_unique_id: null out"; return DC_E_INVALID; }
  RcclApi* r = rccl();
  if (!r->err.empty()) { g_comm_error = r->err; return DC_E_UNSUPPORTED; }
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(e); return DC_E_HIP; }
  static_assert(sizeof(id) == DC_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof id);
  return DC_OK;
}

const char* dc_comm_last_error(const dc_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int dc_comm_create(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world) {
  if (!out || !ctx || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) {
    g_comm_error = "dc_comm_create: bad arguments";
    return DC_E_INVALID;
  }
  dc_comm* c = new dc_comm();
  c->ctx = ctx; c->rank = rank; c->world = world; c->device = dc_ctx_device(ctx);
  hipError_t he = hipSetDevice(c->device);
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (he != hipSuccess) {
    g_comm_error = std::string("dc_comm_create: ") + hipGetErrorString(he);
    delete c;
    return DC_E_HIP;
  }
  if (world > 1) {
    RcclApi* r = rccl();
    if (!r->err.empty()) { g_comm_error = r->err; hipStreamDestroy(c->stream); delete c; return DC_E_UNSUPPORTED; }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t e = r->CommInitRank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) {
      g_comm_error = std::string("ncclCommInitRank: ") + r->GetErrorString(e);
      hipStreamDestroy(c->stream);
      delete c;
      return DC_E_HIP;
    }
  }
  *out = c;
  return DC_OK;
}

void dc_comm_destroy(dc_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm) rccl()->CommDestroy(c->comm);
  if (c->dev_buf) hipFree(c->dev_buf);
  if (c->host_buf) hipHostFree(c->host_buf);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int dc_gather_results(dc_comm* c, const dc_result* local, int n_local, dc_result* gathered) {
  if (!c) return DC_E_INVALID;
  if (!local || n_local <= 0) return c->fail(DC_E_INVALID, "dc_gather_results: bad arguments");
  if (c->rank == 0 && !gathered) return c->fail(DC_E_INVALID, "dc_gather_results: rank 0 needs the output array");
  const int cap = local[0].capacity, T = local[0].T;
  if (cap <= 0 || T <= 0) return c->fail(DC_E_INVALID, "dc_gather_results: results carry no capacity / T (run a forward first)");
  for (int i = 0; i < n_local; ++i) {
    if (local[i].capacity != cap || local[i].T != T || local[i].K < 0 || local[i].K > cap || !local[i].boxes ||
        !local[i].scores || !local[i].tokens)
      return c->fail(DC_E_INVALID, "dc_gather_results: every record needs the same capacity and T and K <= capacity");
  }
  if (c->rank == 0)
    for (int i = 0; i < c->world * n_local; ++i)
      if (gathered[i].capacity < cap || !gathered[i].boxes || !gathered[i].scores || !gathered[i].tokens)
        return c->fail(DC_E_INVALID, "dc_gather_results: gathered[] entries need capacity >= the senders' capacity");
  if (hipSetDevice(c->device) != hipSuccess) return c->fail(DC_E_HIP, "hipSetDevice failed");
  const size_t rb = record_bytes(cap, T), block = rb * (size_t)n_local;
  const size_t need = c->rank == 0 ? block * (size_t)c->world : block;
  if (c->dev_bytes < need) {
    if (c->dev_buf) hipFree(c->dev_buf);
    if (c->host_buf) hipHostFree(c->host_buf);
    c->dev_buf = c->host_buf = nullptr; c->dev_bytes = c->host_bytes = 0;
    if (hipMalloc(&c->dev_buf, need) != hipSuccess || hipHostMalloc(&c->host_buf, need, hipHostMallocDefault) != hipSuccess)
      return c->fail(DC_E_NOMEM, "dc_gather_results: buffer allocation failed");
    c->dev_bytes = c->host_bytes = need;
  }
  // pack this rank's records (rank 0: into slot 0 of the gathered layout)
  char* hb = static_cast<char*>(c->host_buf);
  for (int i = 0; i < n_local; ++i) {
    char* p = hb + rb * (size_t)i;
    int32_t hdr[4] = {local[i].K, T, cap, 0};
    memcpy(p, hdr, 16);
    const size_t K = (size_t)local[i].K;
    memcpy(p + 16, local[i].boxes, K * 16);
    memcpy(p + 16 + (size_t)cap * 16, local[i].scores, K * 4);
    memcpy(p + 16 + (size_t)cap * 20, local[i].tokens, K * 4 * (size_t)T);
  }
  if (c->world > 1) {
    RcclApi* r = rccl();
#define HCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return c->fail(DC_E_HIP, std::string(#x ": ") + hipGetErrorString(_e)); } while (0)
#define NCHK(x) do { ncclResult_t _e = (x); if (_e != ncclSuccess) return c->fail(DC_E_HIP, std::string(#x ": ") + r->GetErrorString(_e)); } while (0)
    if (c->rank != 0) {
      HCHK(hipMemcpyAsync(c->dev_buf, hb, block, hipMemcpyHostToDevice, c->stream));
      NCHK(r->GroupStart());
      NCHK(r->Send(c->dev_buf, block, ncclInt8, 0, c->comm, c->stream));
      NCHK(r->GroupEnd());
      HCHK(hipStreamSynchronize(c->stream));
      return DC_OK;
    }
    NCHK(r->GroupStart());
    for (int peer = 1; peer < c->world; ++peer)
      NCHK(r->Recv(static_cast<char*>(c->dev_buf) + block * (size_t)peer, block, ncclInt8, peer, c->comm, c->stream));
    NCHK(r->GroupEnd());
    HCHK(hipMemcpyAsync(hb + block, static_cast<char*>(c->dev_buf) + block, block * (size_t)(c->world - 1),
                        hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
#undef HCHK
#undef NCHK
  }
  // rank 0: unpack world * n_local records
  for (int i = 0; i < c->world * n_local; ++i) {
    const char* p = hb + rb * (size_t)i;
    int32_t hdr[4];
    memcpy(hdr, p, 16);
    if (hdr[1] != T || hdr[2] != cap || hdr[0] < 0 || hdr[0] > cap)
      return c->fail(DC_E_STATE, "dc_gather_results: a peer sent records of another shape (capacity / T / n_local must agree on all ranks)");
    dc_result& g = gathered[i];
    g.K = hdr[0]; g.T = T;
    const size_t K = (size_t)hdr[0];
    memcpy(g.boxes, p + 16, K * 16);
    memcpy(g.scores, p + 16 + (size_t)cap * 16, K * 4);
    memcpy(g.tokens, p + 16 + (size_t)cap * 20, K * 4 * (size_t)T);
  }
  return DC_OK;
}

}  // extern "C"
