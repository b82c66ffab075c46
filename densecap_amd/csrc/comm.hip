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
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
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

// ---- transports -------------------------------------------------------------------------------------------------------
// dc_gather_results is written against this four-call point-to-point interface (the grouped send/recv subset of RCCL).
// Two carriers implement it: RCCL itself, and an in-process LOOPBACK hub used by the unit tests to drive the whole
// gather -- shape handshake, per-peer offsets, device staging -- at world > 1 on a single GPU (RCCL refuses two ranks
// on one device).  A loopback communicator is made by handing dc_comm_create an id that starts with "dc-loopback:"
// (the rest of the 128 bytes names the hub); its ranks are threads of one process.
struct Transport {
  virtual ~Transport() {}
  virtual const char* name() const = 0;
  virtual int group_start(std::string& err) = 0;
  virtual int send(const void* dev, size_t bytes, int peer, hipStream_t s, std::string& err) = 0;
  virtual int recv(void* dev, size_t bytes, int peer, hipStream_t s, std::string& err) = 0;
  virtual int group_end(hipStream_t s, std::string& err) = 0;    // returns with the group's traffic enqueued on s
};

struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  ~RcclTransport() override { if (comm) rccl()->CommDestroy(comm); }
  const char* name() const override { return "rccl"; }
  int chk(ncclResult_t e, const char* what, std::string& err) {
    if (e == ncclSuccess) return DC_OK;
    err = std::string(what) + ": " + rccl()->GetErrorString(e);
    return DC_E_HIP;
  }
  int group_start(std::string& err) override { return chk(rccl()->GroupStart(), "ncclGroupStart", err); }
  int send(const void* dev, size_t bytes, int peer, hipStream_t s, std::string& err) override {
    return chk(rccl()->Send(dev, bytes, ncclInt8, peer, comm, s), "ncclSend", err);
  }
  int recv(void* dev, size_t bytes, int peer, hipStream_t s, std::string& err) override {
    return chk(rccl()->Recv(dev, bytes, ncclInt8, peer, comm, s), "ncclRecv", err);
  }
  int group_end(hipStream_t, std::string& err) override { return chk(rccl()->GroupEnd(), "ncclGroupEnd", err); }
};

// In-process hub: one mailbox per (src, dst) pair, filled by send (device -> host copy), drained by the matching recv
// at group_end (host -> device copy).  Messages between a pair keep their order, like RCCL's.
struct LoopHub {
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<std::vector<char>>> box;
  int refs = 0;
};
std::mutex g_hub_mu;
std::map<std::string, LoopHub*> g_hubs;

struct LoopbackTransport : Transport {
  std::string key;
  LoopHub* hub = nullptr;
  int rank = 0;
  struct Pending { void* dev; size_t bytes; int peer; };
  std::vector<Pending> recvs;
  std::vector<std::vector<char>> landed;     // host copies stay alive until the stream has consumed them
  LoopbackTransport(const std::string& k, int r) : key(k), rank(r) {
    std::lock_guard<std::mutex> l(g_hub_mu);
    LoopHub*& h = g_hubs[key];
    if (!h) h = new LoopHub();
    h->refs += 1;
    hub = h;
  }
  ~LoopbackTransport() override {
    std::lock_guard<std::mutex> l(g_hub_mu);
    if (--hub->refs == 0) { g_hubs.erase(key); delete hub; }
  }
  const char* name() const override { return "loopback"; }
  int group_start(std::string&) override { recvs.clear(); return DC_OK; }
  int send(const void* dev, size_t bytes, int peer, hipStream_t s, std::string& err) override {
    std::vector<char> msg(bytes);
    hipError_t e = hipStreamSynchronize(s);              // the payload was staged on this stream
    if (e == hipSuccess) e = hipMemcpy(msg.data(), dev, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { err = std::string("loopback send: ") + hipGetErrorString(e); return DC_E_HIP; }
    {
      std::lock_guard<std::mutex> l(hub->mu);
      hub->box[{rank, peer}].push_back(std::move(msg));
    }
    hub->cv.notify_all();
    return DC_OK;
  }
  int recv(void* dev, size_t bytes, int peer, hipStream_t, std::string&) override {
    recvs.push_back({dev, bytes, peer});
    return DC_OK;
  }
  int group_end(hipStream_t s, std::string& err) override {
    landed.clear();
    landed.reserve(recvs.size());
    for (const Pending& p : recvs) {
      std::vector<char> msg;
      {
        std::unique_lock<std::mutex> l(hub->mu);
        auto& q = hub->box[{p.peer, rank}];
        if (!hub->cv.wait_for(l, std::chrono::seconds(60), [&] { return !q.empty(); })) {
          err = "loopback recv: peer " + std::to_string(p.peer) + " sent nothing within 60 s";
          return DC_E_STATE;
        }
        msg = std::move(q.front());
        q.pop_front();
      }
      if (msg.size() != p.bytes) {           // RCCL would hang or corrupt here; the loopback hub can tell
        err = "loopback recv: peer " + std::to_string(p.peer) + " sent " + std::to_string(msg.size()) + " bytes, " +
              std::to_string(p.bytes) + " expected";
        return DC_E_STATE;
      }
      landed.push_back(std::move(msg));
      hipError_t e = hipMemcpyAsync(p.dev, landed.back().data(), p.bytes, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) { err = std::string("loopback recv: ") + hipGetErrorString(e); return DC_E_HIP; }
    }
    recvs.clear();
    hipError_t e = hipStreamSynchronize(s);
    landed.clear();
    if (e != hipSuccess) { err = std::string("loopback recv: ") + hipGetErrorString(e); return DC_E_HIP; }
    return DC_OK;
  }
};

const char kLoopbackPrefix[] = "dc-loopback:";

}  // namespace

struct dc_comm {
  dc_ctx* ctx = nullptr;
  int device = 0, rank = 0, world = 1;
  Transport* tp = nullptr;     // null when world == 1
  hipStream_t stream = nullptr;
  void* dev_buf = nullptr;     // rank 0: world record blocks; others: one (world > 1 only)
  size_t dev_bytes = 0;
  void* host_buf = nullptr;    // pinned staging, same size
  size_t host_bytes = 0;
  int32_t* dev_hdr = nullptr;  // (world, 4) int32: the shape handshake before the payload
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

#define HCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return c->fail(DC_E_HIP, std::string(#x ": ") + hipGetErrorString(_e)); } while (0)
#define TCHK(x) do { int _r = (x); if (_r != DC_OK) return c->fail(_r, std::string(c->tp->name()) + " " + err); } while (0)

// Shape handshake before the payload: every rank must bring the same (n_local, capacity, T) -- a mismatch would post
// sends and receives of different byte counts, which hangs or corrupts under RCCL.  Peers send their 16-byte shape to
// rank 0, rank 0 answers every peer with its verdict {ok, n_local, capacity, T of rank 0}; only then does the payload
// move.  Two tiny messages per peer, once per gather.
// `local_ok`: this rank's own argument validation (advisor finding, round 3: a rank that returned BEFORE the handshake
// left its peers blocked in it).  A rank with bad arguments still takes part and says so in word 3; rank 0's verdict
// then fails the gather on EVERY rank together.
static const int32_t kShapeMagic = 0x44434752, kShapeBad = 0x44434244;   // 'DCGR', 'DCBD'
static int shape_handshake(dc_comm* c, int n_local, int cap, int T, bool local_ok) {
  std::string err;
  int32_t mine[4] = {n_local, cap, T, local_ok ? kShapeMagic : kShapeBad};
  if (c->rank != 0) {
    HCHK(hipMemcpyAsync(c->dev_hdr, mine, 16, hipMemcpyHostToDevice, c->stream));
    TCHK(c->tp->group_start(err));
    TCHK(c->tp->send(c->dev_hdr, 16, 0, c->stream, err));
    TCHK(c->tp->group_end(c->stream, err));
    TCHK(c->tp->group_start(err));
    TCHK(c->tp->recv(c->dev_hdr + 4, 16, 0, c->stream, err));
    TCHK(c->tp->group_end(c->stream, err));
    int32_t verdict[4];
    HCHK(hipMemcpyAsync(verdict, c->dev_hdr + 4, 16, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    if (verdict[0] == 2)
      return c->fail(DC_E_STATE, "dc_gather_results: another rank rejected its arguments; no payload was exchanged");
    if (verdict[0] != 1)
      return c->fail(DC_E_STATE, "dc_gather_results: ranks disagree on (n_local, capacity, T): rank 0 has (" +
                                     std::to_string(verdict[1]) + ", " + std::to_string(verdict[2]) + ", " +
                                     std::to_string(verdict[3]) + "), rank " + std::to_string(c->rank) + " has (" +
                                     std::to_string(n_local) + ", " + std::to_string(cap) + ", " + std::to_string(T) + ")");
    return DC_OK;
  }
  TCHK(c->tp->group_start(err));
  for (int peer = 1; peer < c->world; ++peer) TCHK(c->tp->recv(c->dev_hdr + 4 * peer, 16, peer, c->stream, err));
  TCHK(c->tp->group_end(c->stream, err));
  std::vector<int32_t> all((size_t)c->world * 4);
  HCHK(hipMemcpyAsync(all.data() + 4, c->dev_hdr + 4, (size_t)(c->world - 1) * 16, hipMemcpyDeviceToHost, c->stream));
  HCHK(hipStreamSynchronize(c->stream));
  int bad = -1, invalid = local_ok ? -1 : 0;
  for (int peer = 1; peer < c->world; ++peer) {
    if (all[4 * peer + 3] == kShapeBad) { if (invalid < 0) invalid = peer; continue; }
    if (bad < 0 && (all[4 * peer] != n_local || all[4 * peer + 1] != cap || all[4 * peer + 2] != T || all[4 * peer + 3] != kShapeMagic)) bad = peer;
  }
  int32_t verdict[4] = {invalid >= 0 ? 2 : (bad < 0 ? 1 : 0), n_local, cap, T};
  HCHK(hipMemcpyAsync(c->dev_hdr, verdict, 16, hipMemcpyHostToDevice, c->stream));
  TCHK(c->tp->group_start(err));
  for (int peer = 1; peer < c->world; ++peer) TCHK(c->tp->send(c->dev_hdr, 16, peer, c->stream, err));
  TCHK(c->tp->group_end(c->stream, err));
  HCHK(hipStreamSynchronize(c->stream));
  if (invalid > 0)
    return c->fail(DC_E_STATE, "dc_gather_results: rank " + std::to_string(invalid) + " rejected its arguments; no payload was exchanged");
  if (invalid == 0) return DC_E_INVALID;                       // rank 0's own arguments: the caller reports the message it recorded
  if (bad >= 0)
    return c->fail(DC_E_STATE, "dc_gather_results: rank " + std::to_string(bad) + " brought (n_local, capacity, T) = (" +
                                   std::to_string(all[4 * bad]) + ", " + std::to_string(all[4 * bad + 1]) + ", " +
                                   std::to_string(all[4 * bad + 2]) + "), rank 0 has (" + std::to_string(n_local) + ", " +
                                   std::to_string(cap) + ", " + std::to_string(T) + "): every rank must pass equal shards");
  return DC_OK;
}

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

int dc_comm_create_ex(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world, int flags) {
  if (!out || !ctx || world < 1 || rank < 0 || rank >= world || (world > 1 && !id) || (flags & ~DC_COMM_SELF_TRANSPORT)) {
    g_comm_error = "dc_comm_create: bad arguments";
    return DC_E_INVALID;
  }
  // world == 1 normally needs no carrier at all (a host-side copy).  DC_COMM_SELF_TRANSPORT makes the single rank build
  // the carrier anyway -- ncclCommInitRank with one rank (its own id when the caller passes none) -- and every gather
  // then travels through the SAME staging, ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd calls as at world > 1,
  // with rank 0 as its own peer: the multi-GPU code path can be executed (and is, by the tests and bench.py) on one GPU.
  const bool self_tp = world == 1 && (flags & DC_COMM_SELF_TRANSPORT) != 0;
  ncclUniqueId self_id;
  if (self_tp && !id) {
    if (int rc = dc_comm_unique_id(&self_id); rc != DC_OK) return rc;
    id = &self_id;
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
  if (world > 1 || self_tp) {
    if (memcmp(id, kLoopbackPrefix, sizeof(kLoopbackPrefix) - 1) == 0) {
      const char* p = static_cast<const char*>(id);
      c->tp = new LoopbackTransport(std::string(p, strnlen(p, DC_COMM_ID_BYTES)), rank);
    } else {
      RcclApi* r = rccl();
      if (!r->err.empty()) { g_comm_error = r->err; hipStreamDestroy(c->stream); delete c; return DC_E_UNSUPPORTED; }
      ncclUniqueId uid;
      memcpy(&uid, id, sizeof uid);
      RcclTransport* t = new RcclTransport();
      ncclResult_t e = r->CommInitRank(&t->comm, world, uid, rank);
      if (e != ncclSuccess) {
        g_comm_error = std::string("ncclCommInitRank: ") + r->GetErrorString(e);
        t->comm = nullptr;
        delete t;
        hipStreamDestroy(c->stream);
        delete c;
        return DC_E_HIP;
      }
      c->tp = t;
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->dev_hdr), (size_t)(world + 1) * 16) != hipSuccess) {
      g_comm_error = "dc_comm_create: hipMalloc failed";
      delete c->tp;
      hipStreamDestroy(c->stream);
      delete c;
      return DC_E_NOMEM;
    }
  }
  *out = c;
  return DC_OK;
}

// DC_COMM_FORCE_RCCL=1 in the environment: every world == 1 communicator builds the carrier (see dc_comm_create_ex) --
// lets an unmodified host (bench.py under torch.distributed.run with one rank, a LuaJIT script) exercise it.
int dc_comm_create(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world) {
  const char* e = getenv("DC_COMM_FORCE_RCCL");
  return dc_comm_create_ex(out, ctx, id, rank, world, (e && e[0] == '1') ? DC_COMM_SELF_TRANSPORT : 0);
}

const char* dc_comm_transport(const dc_comm* c) {
  if (!c) return "";
  if (!c->tp) return "host copy";
  return c->world == 1 ? (c->tp->name()[0] == 'r' ? "rccl, self" : "loopback, self") : c->tp->name();
}

void dc_comm_destroy(dc_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  delete c->tp;
  if (c->dev_buf) hipFree(c->dev_buf);
  if (c->dev_hdr) hipFree(c->dev_hdr);
  if (c->host_buf) hipHostFree(c->host_buf);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int dc_gather_results(dc_comm* c, const dc_result* local, int n_local, dc_result* gathered) {
  if (!c) return DC_E_INVALID;
  // Local validation first -- but with more than one rank nobody returns before the handshake: the verdict of the
  // handshake carries a rank's failure to all the others, so that they fail together instead of waiting for it.
  std::string why;
  int cap = 0, T = 0;
  if (!local || n_local <= 0) why = "dc_gather_results: bad arguments";
  else if (c->rank == 0 && !gathered) why = "dc_gather_results: rank 0 needs the output array";
  else {
    cap = local[0].capacity; T = local[0].T;
    if (cap <= 0 || T <= 0) why = "dc_gather_results: results carry no capacity / T (run a forward first)";
    for (int i = 0; why.empty() && i < n_local; ++i)
      if (local[i].capacity != cap || local[i].T != T || local[i].K < 0 || local[i].K > cap || !local[i].boxes ||
          !local[i].scores || !local[i].tokens)
        why = "dc_gather_results: every record needs the same capacity and T and K <= capacity";
    if (why.empty() && c->rank == 0)
      for (int i = 0; why.empty() && i < c->world * n_local; ++i)
        if (gathered[i].capacity < cap || !gathered[i].boxes || !gathered[i].scores || !gathered[i].tokens)
          why = "dc_gather_results: gathered[] entries need capacity >= the senders' capacity";
  }
  if (c->world == 1 && !why.empty()) return c->fail(DC_E_INVALID, why);
  if (hipSetDevice(c->device) != hipSuccess) return c->fail(DC_E_HIP, "hipSetDevice failed");
  if (c->world > 1) {
    const int rc = shape_handshake(c, why.empty() ? n_local : 0, cap, T, why.empty());
    if (!why.empty()) return c->fail(DC_E_INVALID, why);         // this rank's own message wins over the handshake's
    if (rc != DC_OK) return rc;
  }
  const size_t rb = record_bytes(cap, T), block = rb * (size_t)n_local;
  const size_t need = c->rank == 0 ? block * (size_t)c->world : block;
  if (c->host_bytes < need) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);        // nothing may still read the buffers being replaced
    if (c->dev_buf) hipFree(c->dev_buf);
    if (c->host_buf) hipHostFree(c->host_buf);
    c->dev_buf = c->host_buf = nullptr; c->dev_bytes = c->host_bytes = 0;
    if (hipHostMalloc(&c->host_buf, need, hipHostMallocDefault) != hipSuccess)
      return c->fail(DC_E_NOMEM, "dc_gather_results: host staging allocation failed");
    c->host_bytes = need;
    if (c->tp != nullptr) {                                       // without a carrier world == 1 is a host-side copy: no device buffer
      const size_t dneed = c->world == 1 ? 2 * need : need;       // self transport: send block + receive block
      if (hipMalloc(&c->dev_buf, dneed) != hipSuccess) return c->fail(DC_E_NOMEM, "dc_gather_results: device buffer allocation failed");
      c->dev_bytes = dneed;
    }
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
  if (c->world == 1 && c->tp != nullptr) {
    // Self transport: the single rank is sender and receiver of one block.  The same calls in the same order as a peer
    // and rank 0 make between them at world > 1 -- shape words first (16 bytes there and back), then the payload: H2D
    // staging, ONE group holding the receive and the send, D2H of the received block -- and the records that come back are
    // what gets unpacked, so a carrier that drops, reorders or truncates bytes fails the byte-for-byte tests.
    std::string err;
    int32_t mine[4] = {n_local, cap, T, kShapeMagic};
    HCHK(hipMemcpyAsync(c->dev_hdr, mine, 16, hipMemcpyHostToDevice, c->stream));
    TCHK(c->tp->group_start(err));
    TCHK(c->tp->recv(c->dev_hdr + 4, 16, 0, c->stream, err));
    TCHK(c->tp->send(c->dev_hdr, 16, 0, c->stream, err));
    TCHK(c->tp->group_end(c->stream, err));
    int32_t back[4] = {0, 0, 0, 0};
    HCHK(hipMemcpyAsync(back, c->dev_hdr + 4, 16, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    if (memcmp(back, mine, 16) != 0) return c->fail(DC_E_STATE, "dc_gather_results: the self transport returned other shape words than were sent");
    char* rx = static_cast<char*>(c->dev_buf) + block;
    HCHK(hipMemsetAsync(rx, 0xff, block, c->stream));
    HCHK(hipMemcpyAsync(c->dev_buf, hb, block, hipMemcpyHostToDevice, c->stream));
    HCHK(hipStreamSynchronize(c->stream));                        // (the pinned staging is read by the copy engine until here)
    memset(hb, 0xee, block);                                      // what is unpacked below must have travelled
    TCHK(c->tp->group_start(err));
    TCHK(c->tp->recv(rx, block, 0, c->stream, err));
    TCHK(c->tp->send(c->dev_buf, block, 0, c->stream, err));
    TCHK(c->tp->group_end(c->stream, err));
    HCHK(hipMemcpyAsync(hb, rx, block, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
  } else if (c->world > 1) {
    std::string err;
    if (c->rank != 0) {
      HCHK(hipMemcpyAsync(c->dev_buf, hb, block, hipMemcpyHostToDevice, c->stream));
      TCHK(c->tp->group_start(err));
      TCHK(c->tp->send(c->dev_buf, block, 0, c->stream, err));
      TCHK(c->tp->group_end(c->stream, err));
      HCHK(hipStreamSynchronize(c->stream));
      return DC_OK;
    }
    // ONE group: world-1 receives, peer p's block lands at offset block*p (each peer rides its own xGMI link to rank 0)
    TCHK(c->tp->group_start(err));
    for (int peer = 1; peer < c->world; ++peer)
      TCHK(c->tp->recv(static_cast<char*>(c->dev_buf) + block * (size_t)peer, block, peer, c->stream, err));
    TCHK(c->tp->group_end(c->stream, err));
    HCHK(hipMemcpyAsync(hb + block, static_cast<char*>(c->dev_buf) + block, block * (size_t)(c->world - 1),
                        hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
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
