// libdensecap_hip.so -- C ABI (include/densecap.h) over the gfx950 kernels.
//
// A dc_ctx owns: the repacked weights, `lanes` (stream + workspace for one in-flight image,
// used to software-pipeline run_model.lua's image loop), and the per-stage HIP events.
// The whole forward of one image is enqueued on one stream without host round trips
// (box counts stay on the device); the host waits once, for the result copy.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <memory>

#include "common.h"

namespace {

thread_local std::string g_last_error;

struct VggItem { int cin, cout; bool pool_after; };
const VggItem kVgg[DC_NUM_VGG_CONVS] = {
    {3, 64, false},    {64, 64, true},    {64, 128, false},  {128, 128, true},  {128, 256, false},
    {256, 256, false}, {256, 256, true},  {256, 512, false}, {512, 512, false}, {512, 512, true},
    {512, 512, false}, {512, 512, false}, {512, 512, false}};  // no pool5 (DenseCapModel.lua:61-63)

enum Stage { ST_TRUNK = 0, ST_RPN, ST_NMS1, ST_ROIPOOL, ST_FC, ST_HEADS, ST_LSTM, ST_NMS2, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"vgg16_trunk", "rpn_conv_heads_decode", "rpn_nms",   "bilinear_roi_pool",
                                     "fc6_fc7",     "recog_heads",           "lstm_decode", "final_nms_gather"};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct Lane {
  hipStream_t stream = nullptr;
  hipEvent_t ev[ST_COUNT + 1] = {};
  int H = 0, W = 0, P = 0;  // sizes the workspace is built for
  int G = 1;                // images the workspace holds side by side (group capacity)
  int g = 1;                // images of the group in flight
  int fh = 0, fw = 0, A = 0;
  DevBuf arena;               // one allocation, carved below
  float *img = nullptr, *act[2] = {nullptr, nullptr}, *feat = nullptr, *rpn_hidden = nullptr, *heads = nullptr;
  float *rpn_boxes = nullptr, *rpn_xyxy = nullptr, *rpn_p = nullptr;
  uint8_t* rpn_valid = nullptr;
  NmsWorkspace nms;
  void* nms_base = nullptr;
  int32_t *picks1 = nullptr, *count1 = nullptr, *picks2 = nullptr, *count2 = nullptr;
  float *roi_boxes = nullptr, *roi_feats = nullptr, *fc6_out = nullptr, *codes = nullptr;
  float *obj = nullptr, *final_trans = nullptr, *final_boxes = nullptr, *final_xyxy = nullptr;
  float *enc = nullptr, *gates = nullptr, *hstate = nullptr, *cstate = nullptr, *logits = nullptr;
  int32_t *tok = nullptr, *seq = nullptr;
  int32_t* surv_total = nullptr;   // captions after the final NMS: rows the group's final NMS runs kept, all images together
  float* out_feats = nullptr;
  char* out_pack = nullptr;     // the group's packed result records (final_pack_kernel), copied to host_stage in one piece
  float* splitk_ws = nullptr;   // split-K partial tiles (<= 256 tiles of 128x128)
  int32_t* out_tokens = nullptr;
  // pinned host staging
  void* host_stage = nullptr;
  size_t host_stage_bytes = 0;
  bool busy = false;
  dc_result* pending = nullptr;      // results of the group in flight: pending[0..g)
  bool pending_feats = false;
  float* pending_feat_dst = nullptr; float* pending_box_dst = nullptr; int32_t* pending_k_dst = nullptr;
  int pending_capacity = 0;
  float stage_ms[ST_COUNT] = {};
  bool have_times = false;
  // beam search scratch (allocated on first use; beam_chunk() proposals x beam rows at a time)
  void* beam_base = nullptr;
  int beam_rows = 0, beam_chunk = 0, beam_width = 0;   // what the scratch was carved for (rows = chunk x beam)
  float *bm_enc = nullptr, *bm_gates = nullptr, *bm_h[2] = {nullptr, nullptr}, *bm_c[2] = {nullptr, nullptr};
  float *bm_logits = nullptr, *bm_top_lp = nullptr, *bm_lp[2] = {nullptr, nullptr};
  int32_t *bm_top_idx = nullptr, *bm_beams[2] = {nullptr, nullptr}, *bm_parent = nullptr, *bm_tok = nullptr;
  uint8_t* bm_fin = nullptr;
  hipStream_t aux = nullptr;            // single-image mode: second half of the decode rows runs here
  hipStream_t aux2 = nullptr;           // single-image mode: the final NMS runs here, beside the decode
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
  // graph replay (dc_set_graph_replay): the forward of one (shape, settings) key, captured once and relaunched
  uint64_t carve_epoch = 0;             // bumped whenever lane_prepare hands out new workspace pointers
  hipGraphExec_t gexec = nullptr;
  std::array<int64_t, 28> gkey{}, last_key{};
  bool last_key_valid = false;          // last_key = the key of the previous (eager) forward on this lane
  bool ran_graph = false;               // the group in flight was a graph launch (no stage events)
  bool nms_before_decode = false;       // the group in flight ran the final NMS before the decode (dc_set_caption_order(1))
};

struct ProfEvt { hipEvent_t a, b; double flops; };

}  // namespace

struct dc_ctx {
  int device = 0;
  std::string err;
  bool have_weights = false;
  float rpn_nms_thresh = 0.7f, final_nms_thresh = 0.3f;
  int max_lanes = 3;
  bool captions_after_final_nms = false;
  int group = 0;             // images per lane group (dc_set_group): 0 = default (1), 1 .. kGemmMaxGroup
  int arena_allocs = 0;      // lane workspace (re)allocations so far (dc_debug_fetch "arena_allocs")
  double host_enqueue_ms = 0;  // host ms per image spent enqueueing in the last dc_forward_batch
  int beam_size = 0;         // 0 = greedy LM:sample; > 0 = LM:beamsearch (LanguageModel.lua:129-131)
  int64_t beam_chunk_floats = (int64_t)1 << 28;   // cap of the beam search's full-logits buffer (dc_debug_set)
  uint32_t* fault_dev = nullptr;   // sticky device word: a stream-K owner gave up waiting for its partner (checked with the results)
  int force_cfg = 0;         // measurement hook: tile configuration of plain launches (dc_debug_set "force_cfg")
  int stagger = 0;           // measurement hook: start-up stagger of a launch's workgroups (dc_debug_set "stagger")
  int walk = 0;              // measurement hook: 128x64 launches as one workgroup per slot walking its tiles (dc_debug_set "walk")
  int epi_wide = 1;          // plain interior epilogues as 16-byte stores staged through LDS (dc_debug_set "epi_wide")
  int plan_mode = -1;        // measurement hook "plan_mode": -1 = planning follows the lane count, 0 = multi-lane planning, 1 = single-image planning
  int v2_stages = 0;         // LDS ring depth of the 128x64-tile kernel (dc_debug_set "v2_stages": 0 by tile count, 2 or 3 forced)
  int tail_mode = 0;         // partial last round in single-image mode: 0 stream-K, 1 K-split tail plan, 2 whole tiles (dc_debug_set)
  bool serial_mode = false;  // lanes == 1: idle CUs in a layer's last round are worth a tail split-K (dc_set_lanes)
  int num_proposals = 300;  // LocalizationLayer default (LocalizationLayer.lua:237); run_model sets 1000
  bool clip_boxes = true;   // LocalizationLayer.test_clip_boxes (LocalizationLayer.lua:235)
  int math_mode = 0;        // dc_set_math_mode: 0 = fp32 MFMA (default), 1 = split-bf16 (three planes, six products, fp32 accumulate)
  bool graphs = false;      // dc_set_graph_replay: repeated forwards of one shape are relaunched as a captured hipGraph
  uint64_t weights_epoch = 0;
  int graph_launches = 0, graph_captures = 0;    // dc_debug_fetch "graph_launches" / "graph_captures"
  std::string graph_note;                        // why replay was dropped, if it was (dc_debug_fetch "graph_replay_on" == 0)
  // dims
  int k = 0, R = 0, V = 0, T = 0, E = 0, Hd = 0, D = 0;
  float fc[4] = {0, 0, 0, 0};
  // device weights
  std::vector<void*> owned;
  float* conv_w[DC_NUM_VGG_CONVS] = {};
  float* conv_b[DC_NUM_VGG_CONVS] = {};
  float *rpn_w = nullptr, *rpn_b = nullptr, *heads_w = nullptr, *heads_b = nullptr;
  float *fc6_w = nullptr, *fc6_b = nullptr, *fc7_w = nullptr, *fc7_b = nullptr, *head5_w = nullptr, *head5_b = nullptr;
  float *enc_w = nullptr, *enc_b = nullptr, *wxT = nullptr, *whT = nullptr, *lstm_b = nullptr, *xg = nullptr;
  float *out_w = nullptr, *out_b = nullptr, *anchors = nullptr;
  float* dec_w = nullptr;   // (V1pad + 4Hd, Hd): rows [0,V+1) = lm_out_w, zero rows up to V1pad (multiple of 64), then Wh^T
  int V1pad = 0;
  std::vector<std::unique_ptr<Lane>> lanes;
  // split-bf16 mode: weight matrices that can take it, and their bf16 planes (made when the mode is first switched on)
  struct PlaneEnt { const float* W; size_t rows; int K; uint16_t* planes; };
  std::vector<PlaneEnt> planes;
  int bf3_presplit = 1;             // dc_debug_set "bf3_presplit": 0 = split the weights in registers too (mode 1 of the kernels)
  int bf3_all = 0;                  // dc_debug_set "bf3_all": 1 = math mode 1 takes EVERY contraction, not only those mfma_gemm_bf3_pays
                                    // names (test hook: small and ragged problems then exercise the split-bf16 kernels)
  DevBuf pre_src, pre_scratch;      // dc_preprocess_u8: uploaded bytes, width-pass plane (grow only)
  DevBuf pre_taps;                  // ... and the tap tables of the sizes in pre_key, kept while the sizes repeat (webcam frames)
  std::vector<char> pre_taps_host;  // (the host copy outlives its asynchronous upload)
  int pre_key[4] = {0, 0, 0, 0};    // H0, W0, oh, ow the tables were made for
  // MFMA profile
  bool prof = false;
  std::vector<ProfEvt> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  int64_t prof_launches = 0;
  double prof_ms = 0, prof_flops = 0;

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    g_last_error = buf;
    return code;
  }
};

namespace {

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) return ctx->fail(DC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                           __FILE__, __LINE__);                                      \
  } while (0)
#define DCCHK(expr)            \
  do {                         \
    int _r = (expr);           \
    if (_r != DC_OK) return _r; \
  } while (0)

#define KCHK(expr)                                                                                      \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return ctx->fail(DC_E_HIP, "%s: %s", #expr, hipGetErrorString(_e));           \
  } while (0)

int dev_alloc(dc_ctx* ctx, void** p, size_t bytes) {
  HIPCHK(hipMalloc(p, bytes ? bytes : 16));
  ctx->owned.push_back(*p);
  return DC_OK;
}
int upload(dc_ctx* ctx, float** dst, const float* host, size_t n) {
  if (!host) return ctx->fail(DC_E_INVALID, "dc_load_weights: null weight pointer");
  DCCHK(dev_alloc(ctx, reinterpret_cast<void**>(dst), n * sizeof(float)));
  HIPCHK(hipMemcpy(*dst, host, n * sizeof(float), hipMemcpyHostToDevice));
  return DC_OK;
}

hipEvent_t prof_event(dc_ctx* ctx) {
  if (!ctx->prof_pool.empty()) {
    hipEvent_t e = ctx->prof_pool.back();
    ctx->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}

// every MFMA contraction goes through here (optionally bracketed by HIP events).  `ws` (optional) is a
// scratch buffer of ws_floats floats on the same stream: problems with few tiles and a long K are split
// along K over several workgroups per tile and finished by a small reduce kernel.
// Scratch of one stream's contractions: partial tiles of split-K launches and tail plans, stream-K slots.
struct Ws {
  float* p = nullptr;
  size_t floats = 0;
};

// the ctx's sticky device fault word (stream-K owners and the NMS band scan report a hand-off that never arrived there; the
// packed results carry it to the host): made on first need, before any capture of the forward
static void ensure_fault_word(dc_ctx* ctx) {
  if (ctx->fault_dev == nullptr && hipMalloc(reinterpret_cast<void**>(&ctx->fault_dev), 64) == hipSuccess) {
    ctx->owned.push_back(ctx->fault_dev);
    (void)hipMemset(ctx->fault_dev, 0, 64);
  }
}

int run_gemm(dc_ctx* ctx, const GemmDesc& d_in, hipStream_t s, const Ws& w = Ws()) {
  float* const ws = w.p;
  const size_t ws_floats = w.floats;
  GemmDesc d = d_in;
  d.stages = ctx->v2_stages;
  d.force_cfg = ctx->force_cfg;
  d.stagger = ctx->stagger;
  d.walk = ctx->walk;
  d.epi_wide = ctx->epi_wide;
  d.bf3 = ctx->math_mode == 1 && (ctx->bf3_all || mfma_gemm_bf3_pays(d)) ? 1 : 0;
  if (d.bf3 && ctx->bf3_presplit) {
    const uint16_t* pp = d_in.sk_slots != nullptr ? reinterpret_cast<const uint16_t*>(d_in.sk_slots) : nullptr;    // a caller's own planes (per-op entry points)
    int prow = d_in.sk_np;
    if (pp == nullptr)
      for (const auto& e : ctx->planes)
        if (e.W == d.W && e.K == d.K && e.planes != nullptr) { pp = e.planes; prow = (int)e.rows; break; }
    if (pp != nullptr) { d.bf3 = 2; d.sk_slots = reinterpret_cast<float*>(const_cast<uint16_t*>(pp)); d.sk_np = prow; }
  }
  if (d.bf3 != 2) { d.sk_slots = nullptr; d.sk_np = 0; }            // (a caller's planes mean nothing to the other routes)
  ProfEvt pe{nullptr, nullptr, gemm_flops(d)};
  if (ctx->prof) {
    pe.a = prof_event(ctx); pe.b = prof_event(ctx);
    (void)hipEventRecord(pe.a, s);
  }
  hipError_t e = hipSuccess;
  GemmPlan pl;                                                // mfma_gemm_plan decides; this function only acts on it
  const bool serial_plan = ctx->plan_mode < 0 ? ctx->serial_mode : ctx->plan_mode == 1;
  mfma_gemm_plan(d, serial_plan, ctx->tail_mode, ws != nullptr ? ws_floats : 0, &pl);
  const int m_split = pl.m_split;
  if (pl.kind == GEMM_PLAN_SPLITK) {
    // few tiles, long K: every tile is shared by `splitk` workgroups
    d.splitk = pl.splitk; d.splitk_ws = ws;
    e = launch_mfma_gemm(d, s);
    if (e == hipSuccess)
      e = d.pool ? launch_splitk_reduce_pool(ws, pl.splitk, d.bias, d.C, 0, d.M, d.N, d.ldc, d.H, d.Wd, d.relu, s)
                 : launch_splitk_reduce(ws, pl.splitk, d.bias, d.C, d.M, d.N, d.ldc, d.relu, s, d.m_dev);
  } else if (pl.kind == GEMM_PLAN_STREAMK) {
    // tile count not a multiple of the CU count: whole tiles for the full rounds, the last partial round shared evenly
    // along K by all CUs (stream-K with in-kernel fix-up: no reduce launch, one partial tile per cut)
    if (m_split > 0) {
      GemmDesc a = d;
      a.M = m_split; a.a_rows = d.M;
      e = launch_mfma_gemm_ks(a, s);
    }
    if (e == hipSuccess) {
      GemmDesc b = d;
      b.m_begin = m_split; b.a_rows = d.M;
      ensure_fault_word(ctx);
      b.sk_fault = ctx->fault_dev;
      e = launch_mfma_gemm_sk(b, pl.sk_wgs, pl.sk_np, ws, s);
    }
  } else if (pl.kind == GEMM_PLAN_TAIL) {
    // tile count not a multiple of the CU count: whole tiles for the full rounds, K-split for the last one
    const int tail_sp = pl.tail_splitk;
    if (m_split > 0) {
      GemmDesc a = d;
      a.M = m_split; a.a_rows = d.M;
      e = launch_mfma_gemm_ks(a, s);
    }
    if (e == hipSuccess) {
      GemmDesc b = d;
      b.m_begin = m_split; b.a_rows = d.M; b.splitk = tail_sp; b.splitk_ws = ws;
      e = launch_mfma_gemm_ks(b, s);
      if (e == hipSuccess)
        e = d.pool ? launch_splitk_reduce_pool(ws, tail_sp, d.bias, d.C, m_split, d.M - m_split, d.N, d.ldc, d.H, d.Wd, d.relu, s)
                   : launch_splitk_reduce(ws, tail_sp, d.bias, d.C + (size_t)m_split * d.ldc, d.M - m_split, d.N, d.ldc, d.relu, s);
    }
  } else {
    e = launch_mfma_gemm(d, s);
  }
  if (ctx->prof) {
    (void)hipEventRecord(pe.b, s);
    ctx->prof_pending.push_back(pe);
  }
  if (e != hipSuccess)
    return ctx->fail(DC_E_HIP, "mfma gemm launch failed: %s (M=%d N=%d K=%d conv=%d)", hipGetErrorString(e), d.M,
                     d.N, d.K, d.conv);
  return DC_OK;
}
void prof_collect(dc_ctx* ctx) {
  for (auto& pe : ctx->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
      ctx->prof_ms += ms;
      ctx->prof_flops += pe.flops;
      ctx->prof_launches += 1;
    }
    ctx->prof_pool.push_back(pe.a);
    ctx->prof_pool.push_back(pe.b);
  }
  ctx->prof_pending.clear();
}

// `planes` (optional): the caller's own bf16 planes of W for the split-bf16 mode (per-op entry points; the model's weights
// are looked up in ctx->planes)
int linear(dc_ctx* ctx, hipStream_t s, const float* A, const float* W, const float* bias, float* C, int M, int N,
           int K, int relu, const Ws& ws = Ws(), int plan_M = 0, const uint16_t* planes = nullptr) {
  GemmDesc d;
  d.A = A; d.W = W; d.bias = bias; d.C = C; d.M = M; d.N = N; d.K = K; d.ldc = N; d.relu = relu; d.plan_M = plan_M;
  d.sk_slots = reinterpret_cast<float*>(const_cast<uint16_t*>(planes)); d.sk_np = planes ? N : 0;
  return run_gemm(ctx, d, s, ws);
}
int conv3x3(dc_ctx* ctx, hipStream_t s, const float* in, const float* w, const float* b, float* out, int nimg, int H,
            int W, int Cin, int Cout, int relu, const Ws& ws = Ws(), const uint16_t* planes = nullptr) {
  GemmDesc d;
  d.sk_slots = reinterpret_cast<float*>(const_cast<uint16_t*>(planes)); d.sk_np = planes ? Cout : 0;
  d.A = in; d.W = w; d.bias = b; d.C = out; d.M = nimg * H * W; d.N = Cout; d.K = 9 * Cin; d.ldc = Cout;
  d.relu = relu; d.conv = 1; d.H = H; d.Wd = W; d.Cin = Cin; d.plan_M = H * W;
  return run_gemm(ctx, d, s, ws);
}
// conv3x3 + ReLU + nn.SpatialMaxPooling(2,2,2,2):ceil() (VGG layers conv1_2, conv2_2, conv3_3, conv4_3,
// DenseCapModel.lua:61-76): the pool rides in the conv's epilogue -- the full-resolution activation never reaches HBM
// (conv1_2: 110 MB store -> 28 MB) and four launches disappear.  out: (ceil(H/2), ceil(W/2), Cout).
int conv3x3_pool(dc_ctx* ctx, hipStream_t s, const float* in, const float* w, const float* b, float* out, int nimg, int H,
                 int W, int Cin, int Cout, int relu, const Ws& ws, const uint16_t* planes = nullptr) {
  GemmDesc d;
  d.sk_slots = reinterpret_cast<float*>(const_cast<uint16_t*>(planes)); d.sk_np = planes ? Cout : 0;
  const int slots = 4 * ((H + 1) / 2) * ((W + 1) / 2);          // window slots of one image
  d.A = in; d.W = w; d.bias = b; d.C = out; d.M = nimg * slots; d.N = Cout; d.K = 9 * Cin; d.plan_M = slots;
  d.ldc = Cout; d.relu = relu; d.conv = 1; d.H = H; d.Wd = W; d.Cin = Cin; d.pool = 1;
  if (!mfma_gemm_can_pool(d))
    return ctx->fail(DC_E_UNSUPPORTED, "conv3x3_pool: %dx%dx%d activation exceeds the kernels' 32-bit operand offsets", H, W, Cin);
  return run_gemm(ctx, d, s, ws);
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
// num_proposals = -1 (LocalizationLayer.lua:322-324: uncapped RPN NMS): capacity = every anchor of this image size
int effective_proposals(const dc_ctx* ctx, int H, int W);
constexpr size_t kSplitkWsFloats = (size_t)6400 * 128 * 128;  // 400 MiB per lane: split-K partial outputs (up to 8 x an eight-image group's 8 x 384 x 4096 fc6 rows), tail plans, stream-K slots

Ws lane_ws(const Lane& L) { return Ws{L.splitk_ws, L.splitk_ws ? kSplitkWsFloats : 0}; }

int effective_proposals(const dc_ctx* ctx, int H, int W) {
  if (ctx->num_proposals != -1) return ctx->num_proposals;
  int fh = H, fw = W;
  for (int i = 0; i < DC_NUM_VGG_CONVS; ++i)
    if (kVgg[i].pool_after) { fh = (fh + 1) / 2; fw = (fw + 1) / 2; }
  return ctx->k * fh * fw;
}

// bytes of one image's slot in the pinned result staging: {count; boxes; scores; tokens | fc7 codes}
size_t host_stage_stride(const dc_ctx* ctx, int P) {
  return al(256 + (size_t)P * (16 + 4 + (size_t)ctx->T * 4 + (size_t)ctx->D * 4));
}

// bytes of one image's PACKED record (device and pinned host staging share the layout): {K, fault | boxes | scores | tokens or codes}
size_t pack_stride(const dc_ctx* ctx, int P, bool feats) {
  return al(256 + (size_t)P * (16 + 4 + (feats ? (size_t)ctx->D * 4 : (size_t)ctx->T * 4)));
}

// (Re)build a lane's workspace for G images of size (H,W) side by side and proposal capacity P each.
int lane_prepare(dc_ctx* ctx, Lane& L, int H, int W, int P, int G) {
  if (L.stream == nullptr) {
    HIPCHK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&L.aux, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&L.aux2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&L.ev_fork2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_join2, hipEventDisableTiming));
    for (auto& e : L.ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming));
  }
  if (L.H == H && L.W == W && L.P == P && L.G == G && L.arena.p) return DC_OK;
  int fh = H, fw = W;
  for (int i = 0; i < DC_NUM_VGG_CONVS; ++i)
    if (kVgg[i].pool_after) { fh = (fh + 1) / 2; fw = (fw + 1) / 2; }
  const int A = ctx->k * fh * fw;
  const int Tn = ctx->T, V1 = ctx->V + 1, Dm = ctx->D, E = ctx->E, Hd = ctx->Hd;
  const int nms_n = std::max(A, P);
  struct Carve { void** p; size_t bytes; };
  const size_t act_bytes = (size_t)G * H * W * 64 * sizeof(float);
  const size_t GP = (size_t)G * P;          // rows of the per-RoI tensors: image i owns rows [i*P, (i+1)*P)
  std::vector<Carve> cv = {
      {(void**)&L.img, (size_t)G * 3 * H * W * 4},
      {(void**)&L.act[0], act_bytes},
      {(void**)&L.act[1], act_bytes},
      {(void**)&L.rpn_hidden, (size_t)G * fh * fw * ctx->R * 4},
      {(void**)&L.heads, (size_t)G * fh * fw * 6 * ctx->k * 4},
      {(void**)&L.rpn_boxes, (size_t)G * A * 16},
      {(void**)&L.rpn_xyxy, (size_t)G * A * 16},
      {(void**)&L.rpn_p, (size_t)G * A * 4},
      {(void**)&L.rpn_valid, (size_t)G * A},
      {(void**)&L.nms_base, nms_workspace_bytes(nms_n)},          // one: the images' NMS runs follow each other on the stream
      {(void**)&L.picks1, GP * 4},
      {(void**)&L.count1, (size_t)G * 256},
      {(void**)&L.picks2, GP * 4},
      {(void**)&L.count2, (size_t)G * 256},
      {(void**)&L.surv_total, 256},
      {(void**)&L.roi_boxes, GP * 16},
      {(void**)&L.roi_feats, GP * 49 * 512 * 4},
      {(void**)&L.fc6_out, GP * Dm * 4},
      {(void**)&L.codes, GP * Dm * 4},
      {(void**)&L.obj, GP * 4},
      {(void**)&L.final_trans, GP * 16},
      {(void**)&L.final_boxes, GP * 16},
      {(void**)&L.final_xyxy, GP * 16},
      {(void**)&L.enc, GP * E * 4},
      {(void**)&L.gates, GP * 4 * Hd * 4},
      {(void**)&L.hstate, GP * Hd * 4},
      {(void**)&L.cstate, GP * Hd * 4},
      {(void**)&L.logits, GP * (size_t)std::max(V1, ctx->V1pad / 16) * 4},   // full logits (beam) or 2 x V1pad/32 arg-max partials per row
      {(void**)&L.tok, GP * 4},
      {(void**)&L.seq, GP * Tn * 4},
      {(void**)&L.out_pack, (size_t)G * host_stage_stride(ctx, P)},   // packed result records of the group (either kind)
      {(void**)&L.out_tokens, GP * Tn * 4},
      {(void**)&L.out_feats, GP * Dm * 4},
      {(void**)&L.splitk_ws, kSplitkWsFloats * 4},
  };
  size_t total = 0;
  for (auto& c : cv) total += al(c.bytes);
  // The arena only grows: a directory of mixed sizes (720x480, 720x540, 480x720 ... the normal case for
  // run_model -input_dir) re-carves the existing allocation instead of a free + malloc of ~0.5 GB per image.
  if (L.arena.p == nullptr || total > L.arena.bytes) {
    if (L.arena.p) {
      HIPCHK(hipStreamSynchronize(L.stream));
      HIPCHK(hipFree(L.arena.p));
      L.arena = DevBuf();
    }
    HIPCHK(hipMalloc(&L.arena.p, total));
    L.arena.bytes = total;
    ctx->arena_allocs += 1;
  }
  char* p = static_cast<char*>(L.arena.p);
  for (auto& c : cv) { *c.p = p; p += al(c.bytes); }
  L.carve_epoch += 1;                  // every captured pointer is stale now
  nms_workspace_bind(L.nms, L.nms_base, nms_n);
  const size_t hs = (size_t)G * host_stage_stride(ctx, P);
  if (L.host_stage_bytes < hs) {
    if (L.host_stage) HIPCHK(hipHostFree(L.host_stage));
    HIPCHK(hipHostMalloc(&L.host_stage, hs, hipHostMallocDefault));
    L.host_stage_bytes = hs;
  }
  L.H = H; L.W = W; L.P = P; L.G = G; L.fh = fh; L.fw = fw; L.A = A;
  return DC_OK;
}


// One contiguous block of decode rows and the stream it runs on.  LSTM rows are independent, so a batch may be cut
// into blocks that advance on different streams: every row sees exactly the same arithmetic (the K order of a GEMM
// element does not depend on the tile it falls in), only the kernels of the blocks overlap in time.
struct LmPart { hipStream_t s; int r0, n; Ws ws; };

// LanguageModel:sample greedy decode (LanguageModel.lua:293-348) for the rows of `codes` covered by `parts`
// (n_dev: optional device-side row count <= n of a single part starting at row 0; rows past it are not computed).
int lm_sample_parts(dc_ctx* ctx, Lane& L, const float* codes, const LmPart* parts, int nparts, const int32_t* n_dev,
                    int32_t* seq_out, int plan = 0) {
  // Schedule of one decode (h_t = LSTM state after t steps past the image step, tok_0 = START):
  //   enc = ReLU(codes.Wenc^T + b)                      GEMM   (:27-30)
  //   G   = (b + enc.Wx)                                GEMM   step 0 of LM:sample: the image code is the first input
  //   tail: c_0, h_0 from G (c = 0)                     row kernel
  //   G   = h_0.Wh                                      GEMM
  //   tail: h_1 from xg[START] + G                      row kernel
  //   for t = 1..T-1:  [arg-max partials of h_t.Wout^T + b | G = h_t.Wh]   ONE GEMM launch (W = [Wout; pad; Wh])
  //                    tail: tok_t = arg-max; h_{t+1} from xg[tok_t] + G   row kernel
  //   arg-max partials of h_T.Wout^T + b                GEMM;  tail: tok_T
  // i.e. 2 launches per step.  The h.Wh product of the NEXT step rides in the vocabulary projection's launch (both
  // only need h_t) and fills its partial last round of tiles; the token-dependent half of the gates (a row of the
  // precomputed xg = b + Emb.Wx table) is added where the token is produced.  Per element the arithmetic and its order
  // are those of torch-rnn's nn.LSTM: (b + x.Wx) + h.Wh, sigmoid/tanh, c' = f*c + i*g, h' = o*tanh(c').
  const int E = ctx->E, Hd = ctx->Hd, V1 = ctx->V + 1, T = ctx->T, D = ctx->D;
  const int V1pad = ctx->V1pad, ntn = V1pad / 32;       // arg-max partials per row: one (value, column) per 32-column half of a 64-column tile
  for (int pi = 0; pi < nparts; ++pi) {
    const LmPart& p = parts[pi];
    hipStream_t s = p.s;
    float* gates = L.gates + (size_t)p.r0 * 4 * Hd;
    float* hstate = L.hstate + (size_t)p.r0 * Hd;
    float* cstate = L.cstate + (size_t)p.r0 * Hd;
    {  // image_encoder: Linear(4096,E)+ReLU (:27-30)
      GemmDesc g;
      g.A = codes + (size_t)p.r0 * D; g.W = ctx->enc_w; g.bias = ctx->enc_b; g.C = L.enc + (size_t)p.r0 * E;
      g.M = p.n; g.N = E; g.K = D; g.ldc = E; g.relu = 1; g.m_dev = n_dev; g.plan_M = plan;
      DCCHK(run_gemm(ctx, g, s, p.ws));
    }
    {  // step 0: gates = (b + enc.Wx) + 0.Wh ; c0 = 0 (output ignored, no vocab projection needed)
      GemmDesc g;
      g.A = L.enc + (size_t)p.r0 * E; g.W = ctx->wxT; g.bias = ctx->lstm_b; g.C = gates;
      g.M = p.n; g.N = 4 * Hd; g.K = E; g.ldc = 4 * Hd; g.m_dev = n_dev; g.plan_M = plan;
      DCCHK(run_gemm(ctx, g, s));
    }
    KCHK(launch_lstm_step_tail(nullptr, nullptr, 0, 0, 0, nullptr, gates, cstate, hstate, p.n, n_dev, Hd, 1, nullptr, T, 0, s));
    {  // h_0.Wh, then the START token's xg row (:32,320) joins it in the tail
      GemmDesc g;
      g.A = hstate; g.W = ctx->whT; g.C = gates; g.M = p.n; g.N = 4 * Hd; g.K = Hd; g.ldc = 4 * Hd; g.m_dev = n_dev;
      g.plan_M = plan;
      DCCHK(run_gemm(ctx, g, s));
    }
    KCHK(launch_lstm_step_tail(nullptr, nullptr, 0, 0, V1, ctx->xg, gates, cstate, hstate, p.n, n_dev, Hd, 0, nullptr, T, 0, s));
  }
  for (int t = 0; t < T; ++t) {
    const bool last = t == T - 1;
    for (int pi = 0; pi < nparts; ++pi) {
      const LmPart& p = parts[pi];
      hipStream_t s = p.s;
      float* gates = L.gates + (size_t)p.r0 * 4 * Hd;
      float* hstate = L.hstate + (size_t)p.r0 * Hd;
      // vocab projection with the row arg-max fused into the GEMM epilogue (logits never reach HBM); except after
      // the last step the same launch also produces h.Wh for the next step's gates
      GemmDesc v;
      v.A = hstate; v.W = ctx->dec_w; v.bias = ctx->out_b; v.M = p.n; v.K = Hd; v.m_dev = n_dev; v.plan_M = plan;
      v.amax_val = L.logits + (size_t)p.r0 * 2 * ntn;
      v.amax_idx = reinterpret_cast<int32_t*>(v.amax_val + (size_t)p.n * ntn);
      v.amax_ld = ntn;
      if (last) {
        v.N = V1; v.ldc = V1;
      } else {
        v.N = V1pad + 4 * Hd; v.amax_cols = V1pad; v.amax_n = V1; v.C = gates; v.ldc = 4 * Hd;
      }
      DCCHK(run_gemm(ctx, v, s));
      KCHK(launch_lstm_step_tail(v.amax_val, v.amax_idx, ntn, ntn, 0, ctx->xg, last ? nullptr : gates,
                                 L.cstate + (size_t)p.r0 * Hd, hstate, p.n, n_dev, Hd, 0, seq_out + (size_t)p.r0 * T, T, t, s));
    }
  }
  return DC_OK;
}

// `plan`: rows of one image when n covers a group (0 = n); see GemmDesc::plan_M
int lm_sample(dc_ctx* ctx, Lane& L, const float* codes, int n, int plan, const int32_t* n_dev, int32_t* seq_out) {
  const LmPart whole{L.stream, 0, n, lane_ws(L)};
  return lm_sample_parts(ctx, L, codes, &whole, 1, n_dev, seq_out, plan);
}
// LanguageModel:beamsearch (LanguageModel.lua:170-290), dispatched by LM:updateOutput when self.beam_size is set
// (:129-131; no reference script sets it).  The reference walks the proposals one by one with the beams in the
// minibatch dimension; here ALL proposals advance together (rows = proposals x beams; beam_chunk), row for row the same
// arithmetic: LSTM step (MFMA GEMM + point-wise), vocabulary projection (full logits this time), LogSoftMax + top-k per
// beam, beam x beam merge, states re-indexed by parent.  Ties: lower index first (docs/SEMANTICS.md).
// Proposals that advance together: all of them (rows = P x beam, one GEMM per step over every proposal) unless the
// full-logits buffer rows x (V+1) would pass 2^28 floats (1 GiB) -- e.g. 5,114 proposals at beam 5 / V = 10,497.
int beam_chunk(const dc_ctx* ctx, int n) {
  const long cap = (long)(ctx->beam_chunk_floats / ((int64_t)ctx->beam_size * (ctx->V + 1)));
  return (int)std::max<long>(1, std::min<long>(n, std::max<long>(64, cap)));
}
int beam_prepare(dc_ctx* ctx, Lane& L, int chunk) {
  const int beam = ctx->beam_size, rows = chunk * beam;
  // bm_enc is sized by the chunk, bm_top_lp / bm_top_idx by rows x beam: the scratch is reusable only when NONE of the
  // three grew (beam 2 x 1000 proposals and beam 20 x 100 have the same row count but not the same carve)
  if (L.beam_base && L.beam_rows >= rows && L.beam_chunk >= chunk && L.beam_width >= beam) return DC_OK;
  if (L.beam_base) { HIPCHK(hipStreamSynchronize(L.stream)); HIPCHK(hipFree(L.beam_base)); L.beam_base = nullptr; }
  const int E = ctx->E, Hd = ctx->Hd, V1 = ctx->V + 1, T = ctx->T;
  L.beam_base = nullptr; L.beam_rows = L.beam_chunk = L.beam_width = 0;
  struct Carve { void** p; size_t bytes; };
  std::vector<Carve> cv = {
      {(void**)&L.bm_enc, (size_t)chunk * E * 4},      {(void**)&L.bm_gates, (size_t)rows * 4 * Hd * 4},
      {(void**)&L.bm_h[0], (size_t)rows * Hd * 4},     {(void**)&L.bm_h[1], (size_t)rows * Hd * 4},
      {(void**)&L.bm_c[0], (size_t)rows * Hd * 4},     {(void**)&L.bm_c[1], (size_t)rows * Hd * 4},
      {(void**)&L.bm_logits, (size_t)rows * V1 * 4},   {(void**)&L.bm_top_lp, (size_t)rows * beam * 4},
      {(void**)&L.bm_top_idx, (size_t)rows * beam * 4}, {(void**)&L.bm_lp[0], (size_t)rows * 4},
      {(void**)&L.bm_lp[1], (size_t)rows * 4},         {(void**)&L.bm_beams[0], (size_t)rows * T * 4},
      {(void**)&L.bm_beams[1], (size_t)rows * T * 4},  {(void**)&L.bm_parent, (size_t)rows * 4},
      {(void**)&L.bm_tok, (size_t)rows * 4},           {(void**)&L.bm_fin, (size_t)rows},
  };
  size_t total = 0;
  for (auto& c : cv) total += al(c.bytes);
  HIPCHK(hipMalloc(&L.beam_base, total));
  char* p = static_cast<char*>(L.beam_base);
  for (auto& c : cv) { *c.p = p; p += al(c.bytes); }
  L.beam_rows = rows; L.beam_chunk = chunk; L.beam_width = beam;
  return DC_OK;
}

int lm_beamsearch(dc_ctx* ctx, Lane& L, const float* codes, int n, int32_t* seq_out, hipStream_t s) {
  const int kBeamChunk = beam_chunk(ctx, n);
  DCCHK(beam_prepare(ctx, L, kBeamChunk));
  const int beam = ctx->beam_size, E = ctx->E, Hd = ctx->Hd, V1 = ctx->V + 1, T = ctx->T, D = ctx->D;
  auto step = [&](float* h, float* c, int rows) -> int {        // one LSTM step on the words in bm_tok, in place
    GemmDesc d;
    d.A = h; d.W = ctx->whT; d.C = L.bm_gates; d.M = rows; d.N = 4 * Hd; d.K = Hd; d.ldc = 4 * Hd;
    d.rowterm = ctx->xg; d.rowidx = L.bm_tok; d.rowterm_ld = 4 * Hd;
    DCCHK(run_gemm(ctx, d, s));
    KCHK(launch_lstm_pointwise(L.bm_gates, c, h, rows, nullptr, Hd, 0, s));
    return DC_OK;
  };
  for (int p0 = 0; p0 < n; p0 += kBeamChunk) {
    const int c = std::min(kBeamChunk, n - p0);
    // image step (:198-201) and START step (:203-206), one state row per proposal
    DCCHK(linear(ctx, s, codes + (size_t)p0 * D, ctx->enc_w, ctx->enc_b, L.bm_enc, c, E, D, 1));
    DCCHK(linear(ctx, s, L.bm_enc, ctx->wxT, ctx->lstm_b, L.bm_gates, c, 4 * Hd, E, 0));
    KCHK(launch_lstm_pointwise(L.bm_gates, L.bm_c[0], L.bm_h[0], c, nullptr, Hd, 1, s));
    KCHK(launch_fill_i32(L.bm_tok, V1, c, s));
    DCCHK(step(L.bm_h[0], L.bm_c[0], c));
    DCCHK(linear(ctx, s, L.bm_h[0], ctx->out_w, ctx->out_b, L.bm_logits, c, V1, Hd, 0));
    KCHK(launch_beam_logsoftmax_topk(L.bm_logits, c, V1, V1, nullptr, beam, L.bm_top_lp, L.bm_top_idx, s));
    KCHK(launch_beam_init(L.bm_top_lp, L.bm_top_idx, c, beam, T, V1, L.bm_lp[0], L.bm_beams[0], L.bm_parent, L.bm_tok,
                          L.bm_fin, s));
    // LanguageModel.lua:221-226 duplicates the states for the beams with `layer.output = layer.cell:expand(...):clone()`:
    // BOTH the cell and the hidden state of every beam start from the CELL state of the START step (torch-rnn's
    // nn.LSTM with remember_states reads h0 from self.output).  Replicated as written: h rows := c rows.
    KCHK(launch_beam_gather_state(L.bm_c[0], L.bm_c[0], L.bm_parent, c * beam, beam, 1, Hd, L.bm_h[1], L.bm_c[1], s));
    int cur = 1, bcur = 0;
    const int rows = c * beam;
    for (int t = 1; t < T; ++t) {
      DCCHK(step(L.bm_h[cur], L.bm_c[cur], rows));
      DCCHK(linear(ctx, s, L.bm_h[cur], ctx->out_w, ctx->out_b, L.bm_logits, rows, V1, Hd, 0));
      KCHK(launch_beam_logsoftmax_topk(L.bm_logits, rows, V1, V1, L.bm_fin, beam, L.bm_top_lp, L.bm_top_idx, s));
      KCHK(launch_beam_merge(L.bm_top_lp, L.bm_top_idx, L.bm_lp[bcur], L.bm_beams[bcur], c, beam, T, t, V1,
                             L.bm_lp[bcur ^ 1], L.bm_beams[bcur ^ 1], L.bm_parent, L.bm_tok, L.bm_fin, s));
      KCHK(launch_beam_gather_state(L.bm_h[cur], L.bm_c[cur], L.bm_parent, rows, beam, beam, Hd, L.bm_h[cur ^ 1],
                                    L.bm_c[cur ^ 1], s));
      cur ^= 1; bcur ^= 1;
    }
    KCHK(launch_beam_best(L.bm_beams[bcur], c, beam, T, seq_out + (size_t)p0 * T, s));
  }
  return DC_OK;
}

// Single-image mode (lanes == 1): nothing else is in flight, so the small kernels and partial tile rounds of the 15
// decode steps leave the chip idle (~25 % of the decode).  The rows are cut into two blocks on two streams; their
// kernels fill each other's gaps.  Same outputs bit for bit (tests/test_gpu_e2e.py::test_single_lane_mode_parity).
int lm_sample_two_streams(dc_ctx* ctx, Lane& L, const float* codes, int n, int plan, int32_t* seq_out) {
  const int h = std::min(n, ((n / 2 + 127) / 128) * 128);
  if (h >= n || L.aux == nullptr) return lm_sample(ctx, L, codes, n, plan, nullptr, seq_out);
  const size_t wsf = L.splitk_ws ? kSplitkWsFloats / 2 : 0;     // each block its own half of the partial-tile scratch
  const LmPart parts[2] = {{L.stream, 0, h, Ws{L.splitk_ws, wsf}},
                           {L.aux, h, n - h, Ws{L.splitk_ws ? L.splitk_ws + wsf : nullptr, wsf}}};
  HIPCHK(hipEventRecord(L.ev_fork, L.stream));
  HIPCHK(hipStreamWaitEvent(L.aux, L.ev_fork, 0));
  DCCHK(lm_sample_parts(ctx, L, codes, parts, 2, nullptr, seq_out, plan));
  HIPCHK(hipEventRecord(L.ev_join, L.aux));
  HIPCHK(hipStreamWaitEvent(L.stream, L.ev_join, 0));
  return DC_OK;
}

// Enqueue the whole forward of a GROUP of g images (laid out back to back at `img`) on the lane's stream (no host
// sync).  The dense stages run once for the whole group -- the convolutions over g images, fc6/fc7, heads and the
// decode over g*P RoI rows -- so their launches carry g times the tiles (fuller last rounds, half the launches per
// image); the per-image stages (RPN decode, NMS, RoI pooling, final NMS, gathers) loop over the images.  Every routing
// decision that changes a sum's order is planned per image (GemmDesc::plan_M), so an image's numbers do not depend on
// the group it travels in.
// `events`: record the stage events (an eager enqueue; a captured graph carries none -- dc_stage_times then has nothing)
int enqueue_body(dc_ctx* ctx, Lane& L, int g, bool features_only, bool events) {
  hipStream_t s = L.stream;
  const int H = L.H, W = L.W, P = L.P;
#define STAGE_EVENT(i) do { if (events) HIPCHK(hipEventRecord(L.ev[i], s)); } while (0)
  STAGE_EVENT(0);
  // ---- VGG-16 trunk (DenseCapModel.lua:73-76) -------------------------------------------
  int h = H, w = W, cur = 0;
  KCHK(launch_conv3x3_c3(L.img, ctx->conv_w[0], ctx->conv_b[0], L.act[0], g, H, W, 64, 1, s));
  for (int i = 1; i < DC_NUM_VGG_CONVS; ++i) {
    if (kVgg[i].pool_after) {
      // conv + ReLU + ceil-mode 2x2 pool in one launch
      DCCHK(conv3x3_pool(ctx, s, L.act[cur], ctx->conv_w[i], ctx->conv_b[i], L.act[cur ^ 1], g, h, w, kVgg[i].cin,
                         kVgg[i].cout, 1, lane_ws(L)));
      h = (h + 1) / 2; w = (w + 1) / 2;
    } else {
      DCCHK(conv3x3(ctx, s, L.act[cur], ctx->conv_w[i], ctx->conv_b[i], L.act[cur ^ 1], g, h, w, kVgg[i].cin,
                    kVgg[i].cout, 1, lane_ws(L)));
    }
    cur ^= 1;
  }
  L.feat = L.act[cur];
  const size_t feat_elems = (size_t)h * w * 512;
  STAGE_EVENT(1);
  // ---- RPN (LocalizationLayer.lua:265, build_rpn :609-690) ----------------------------------
  DCCHK(conv3x3(ctx, s, L.feat, ctx->rpn_w, ctx->rpn_b, L.rpn_hidden, g, h, w, 512, ctx->R, 1, lane_ws(L)));
  DCCHK(linear(ctx, s, L.rpn_hidden, ctx->heads_w, ctx->heads_b, L.heads, g * h * w, 6 * ctx->k, ctx->R, 0, Ws(), h * w));
  KCHK(launch_rpn_decode(L.heads, g, h, w, ctx->k, ctx->anchors, ctx->fc[0], ctx->fc[1], ctx->fc[2], ctx->fc[3], H, W,
                         L.rpn_boxes, nullptr, nullptr, L.rpn_xyxy, L.rpn_p, L.rpn_valid, ctx->clip_boxes ? 1 : 0, s));   // the whole group in one launch
  STAGE_EVENT(2);
  // ---- RPN NMS (LocalizationLayer.lua:318-338) ------------------------------------------------
  ensure_fault_word(ctx);
  for (int i = 0; i < g; ++i)
    KCHK(launch_nms(L.nms, L.rpn_xyxy + (size_t)i * L.A * 4, L.rpn_p + (size_t)i * L.A, L.rpn_valid + (size_t)i * L.A, L.A,
                    nullptr, ctx->rpn_nms_thresh, P, L.picks1 + (size_t)i * P, L.count1 + i * 64, s, ctx->fault_dev));
  STAGE_EVENT(3);
  // ---- bilinear RoI pooling (LocalizationLayer.lua:338-349) -----------------------------------
  // one launch for the group; the picked RPN boxes are gathered by the kernel itself (and left in roi_boxes for the heads)
  KCHK(launch_bilinear_roi_pool_group(L.feat, feat_elems, g, h, w, 512, L.roi_boxes, P, L.count1, 64, L.picks1, L.rpn_boxes,
                                      (size_t)L.A * 4, H, W, 7, 7, L.roi_feats, 1, s));
  STAGE_EVENT(4);
  // ---- recog_base fc6/fc7 (DenseCapModel.lua:133) ------------------------------------------------
  const int R = g * P;                      // RoI rows of the group
  DCCHK(linear(ctx, s, L.roi_feats, ctx->fc6_w, ctx->fc6_b, L.fc6_out, R, ctx->D, 49 * 512, 1, lane_ws(L), P));
  DCCHK(linear(ctx, s, L.fc6_out, ctx->fc7_w, ctx->fc7_b, L.codes, R, ctx->D, ctx->D, 1, lane_ws(L), P));
  STAGE_EVENT(5);
  // ---- objectness / box regression / final boxes (DenseCapModel.lua:134,139-140) -----------------
  KCHK(launch_recog_heads(L.codes, ctx->head5_w, ctx->head5_b, L.roi_boxes, L.obj, L.final_trans, L.final_boxes,
                          L.final_xyxy, R, ctx->D, s));
  STAGE_EVENT(6);
  const bool survivors_only = ctx->captions_after_final_nms && !features_only;
  // single-image mode, reference order: decode (two row blocks on two streams) and final NMS (a third stream) are
  // independent consumers of the heads' outputs.  Per-launch HIP-event profiling wants kernels that do not overlap:
  // everything stays on one stream while it is on.
  const bool side_streams = ctx->serial_mode && !ctx->prof && !features_only && !survivors_only && R >= 256 &&
                            ctx->beam_size == 0;
  hipStream_t sn = side_streams ? L.aux2 : s;       // stream of the final NMS
  if (side_streams) {
    HIPCHK(hipEventRecord(L.ev_fork2, s));
    HIPCHK(hipStreamWaitEvent(L.aux2, L.ev_fork2, 0));
  }
  // ---- language model (reference order: all P proposals, DenseCapModel.lua:127-162) -----------------
  if (!features_only && !survivors_only) {
    if (ctx->beam_size > 0) DCCHK(lm_beamsearch(ctx, L, L.codes, R, L.seq, s));
    else if (side_streams) DCCHK(lm_sample_two_streams(ctx, L, L.codes, R, P, L.seq));
    else DCCHK(lm_sample(ctx, L, L.codes, R, P, nullptr, L.seq));
  }
  if (!survivors_only) STAGE_EVENT(7);
  // ---- final NMS + gather (DenseCapModel.lua:261-275) ----------------------------------------------
  for (int i = 0; i < g; ++i) {
    const size_t r0 = (size_t)i * P;
    // forward_test skips the final NMS when final_nms_thresh <= 0 (DenseCapModel.lua:261); extractFeatures calls
    // box_utils.nms unconditionally (DenseCapModel.lua:285-304)
    if (ctx->final_nms_thresh > 0.f || features_only) {
      KCHK(launch_nms(L.nms, L.final_xyxy + r0 * 4, L.obj + r0, nullptr, P, L.count1 + i * 64, ctx->final_nms_thresh, -1,
                      L.picks2 + r0, L.count2 + i * 64, sn, ctx->fault_dev));
    } else {
      // DenseCapModel.lua:261: no final NMS when final_nms_thresh <= 0 -> all RoIs, in RPN order
      KCHK(launch_iota_count(L.picks2 + r0, L.count2 + i * 64, L.count1 + i * 64, P, sn));
    }
  }
  if (side_streams) {
    HIPCHK(hipEventRecord(L.ev_join2, L.aux2));
    HIPCHK(hipStreamWaitEvent(s, L.ev_join2, 0));
  }
  // captions after the final NMS: event 7 sits between the two stages here as well, in the order they ran (harvest swaps the names)
  if (survivors_only) STAGE_EVENT(7);
  L.nms_before_decode = survivors_only;
  const bool packed_decode = survivors_only && ctx->beam_size == 0;
  if (packed_decode) {
    // Identical outputs, less work: LSTM rows are independent, so only the rows the final NMS kept are decoded (~a quarter at
    // 1000 proposals / 0.3).  Round 6: ONCE PER GROUP -- the kept fc7 rows of all g images packed into one row block
    // (survivor_compact_kernel), ONE decode over it with the device-side row count (row tiles past it exit at once), routes
    // planned on one image's P rows as in the reference order: an element's arithmetic is the same in either order and in any
    // group (tests/test_gpu_e2e.py::test_caption_order_is_output_invariant).  final_pack reads the packed token rows back
    // image by image.
    KCHK(launch_survivor_compact(L.codes, L.picks2, L.count2, 64, g, P, ctx->D, L.out_feats, L.surv_total, s));
    DCCHK(lm_sample(ctx, L, L.out_feats, R, P, L.surv_total, L.out_tokens));
  } else if (survivors_only) {
    // beam search after the final NMS: image by image (the beam rows of one image advance together)
    for (int i = 0; i < g; ++i) {
      const size_t r0 = (size_t)i * P;
      const int32_t *pk = L.picks2 + r0, *cnt = L.count2 + i * 64;
      KCHK(launch_gather_rows(L.codes + r0 * ctx->D, pk, cnt, P, ctx->D, L.out_feats + r0 * ctx->D, s));
      DCCHK(lm_beamsearch(ctx, L, L.out_feats + r0 * ctx->D, P, L.out_tokens + r0 * ctx->T, s));   // rows past K: zero codes, ignored
    }
  }
  // ---- results: ONE gather launch for the group into packed records, ONE copy to the pinned host staging ---------------
  const size_t stride = pack_stride(ctx, P, features_only);
  KCHK(launch_final_pack(L.final_boxes, L.obj, survivors_only ? L.out_tokens : L.seq, packed_decode ? 2 : survivors_only ? 0 : 1,
                         features_only ? L.codes : nullptr, L.picks2, L.count2, 64, ctx->fault_dev, g, P, ctx->T, ctx->D,
                         L.out_pack, stride, s));
  STAGE_EVENT(8);
  HIPCHK(hipMemcpyAsync(L.host_stage, L.out_pack, (size_t)g * stride, hipMemcpyDeviceToHost, s));
#undef STAGE_EVENT
  return DC_OK;
}

// Everything a captured forward bakes in: workspace pointers (carve epoch, arena, staging), weights, shape, every setting
// that reaches a kernel argument, a launch decision or the stream layout.
std::array<int64_t, 28> graph_key(const dc_ctx* ctx, const Lane& L, int g, bool features_only) {
  auto f2i = [](float f) { int32_t i; memcpy(&i, &f, 4); return (int64_t)i; };
  return {1, (int64_t)L.carve_epoch, (int64_t)ctx->weights_epoch, L.H, L.W, L.P, g, features_only ? 1 : 0,
          f2i(ctx->rpn_nms_thresh), f2i(ctx->final_nms_thresh), ctx->num_proposals, ctx->clip_boxes ? 1 : 0,
          ctx->captions_after_final_nms ? 1 : 0, ctx->serial_mode ? 1 : 0, ctx->plan_mode, ctx->tail_mode, ctx->force_cfg,
          ctx->v2_stages, ctx->stagger, ctx->walk + 2 * ctx->epi_wide, ctx->beam_size, (int64_t)(uintptr_t)ctx->fault_dev,
          (int64_t)(uintptr_t)L.arena.p, (int64_t)(uintptr_t)L.host_stage, (int64_t)(uintptr_t)L.splitk_ws, ctx->math_mode + 2 * ctx->bf3_presplit + 4 * ctx->bf3_all, 0, 0};
}

// `img`: the g images back to back; `sep` (optional) = g separate images instead (a run of equal-sized images of a mixed list)
int enqueue_forward(dc_ctx* ctx, Lane& L, const float* img, int g, int img_on_device, bool features_only,
                    const float* const* sep = nullptr) {
  hipStream_t s = L.stream;
  const size_t img_elems = (size_t)3 * L.H * L.W;
  const hipMemcpyKind kind = img_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (sep == nullptr) HIPCHK(hipMemcpyAsync(L.img, img, g * img_elems * 4, kind, s));
  else
    for (int i = 0; i < g; ++i) HIPCHK(hipMemcpyAsync(L.img + (size_t)i * img_elems, sep[i], img_elems * 4, kind, s));
  L.g = g;
  const size_t stride = pack_stride(ctx, L.P, features_only);
  for (int i = 0; i < g; ++i) *reinterpret_cast<uint32_t*>(static_cast<char*>(L.host_stage) + i * stride + 68) = 0;
  L.ran_graph = false;
  // Graph replay (dc_set_graph_replay): the FIRST forward of a key runs eagerly (lazy allocations, kernel attributes and
  // the stream-K fault word are all in place afterwards), the second is captured and instantiated, later ones are one
  // hipGraphLaunch.  Per-launch profiling and beam search (which allocates on first use) stay eager.
  const bool eligible = ctx->graphs && !ctx->prof && ctx->beam_size == 0;
  if (eligible) {
    const auto key = graph_key(ctx, L, g, features_only);
    if (L.gexec != nullptr && key == L.gkey) {
      HIPCHK(hipGraphLaunch(L.gexec, s));
      ctx->graph_launches += 1;
      L.ran_graph = true;
    } else if (L.last_key_valid && key == L.last_key) {
      if (L.gexec != nullptr) { (void)hipGraphExecDestroy(L.gexec); L.gexec = nullptr; }
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      const int rc = enqueue_body(ctx, L, g, features_only, false);
      const hipError_t e = hipStreamEndCapture(s, &graph);
      hipError_t e2 = hipSuccess;
      if (rc == DC_OK && e == hipSuccess && graph != nullptr) e2 = hipGraphInstantiate(&L.gexec, graph, nullptr, nullptr, 0);
      if (graph != nullptr) (void)hipGraphDestroy(graph);
      if (rc != DC_OK || e != hipSuccess || e2 != hipSuccess || L.gexec == nullptr) {
        // a forward that cannot be captured on this runtime: say so once (stderr + dc_last_error text, no error code:
        // the forward itself still runs, eagerly), stay eager on this ctx; dc_debug_fetch "graph_replay_on" reads 0 from here on
        const hipError_t why = e != hipSuccess ? e : e2;
        (void)hipGetLastError();
        L.gexec = nullptr;
        ctx->graphs = false;
        char note[256];
        snprintf(note, sizeof note, "dc_set_graph_replay: capture of a %dx%d forward failed (%s); graph replay is now OFF for this ctx",
                 L.W, L.H, rc != DC_OK ? ctx->err.c_str() : hipGetErrorString(why));
        fprintf(stderr, "libdensecap_hip: %s\n", note);
        ctx->graph_note = note;
        if (rc != DC_OK) return rc;
        DCCHK(enqueue_body(ctx, L, g, features_only, true));
      } else {
        L.gkey = key;
        ctx->graph_captures += 1;
        HIPCHK(hipGraphLaunch(L.gexec, s));
        ctx->graph_launches += 1;
        L.ran_graph = true;
      }
    } else {
      DCCHK(enqueue_body(ctx, L, g, features_only, true));
      L.last_key = key;
      L.last_key_valid = true;
    }
  } else {
    DCCHK(enqueue_body(ctx, L, g, features_only, true));
    L.last_key_valid = false;
  }
  L.busy = true;
  L.pending_feats = features_only;
  return DC_OK;
}


// Wait for the lane's in-flight group and hand the results to the caller's buffers.
int harvest(dc_ctx* ctx, Lane& L) {
  if (!L.busy) return DC_OK;
  HIPCHK(hipStreamSynchronize(L.stream));
  L.busy = false;
  if (!L.ran_graph) {
    for (int i = 0; i < ST_COUNT; ++i) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, L.ev[i], L.ev[i + 1]);
      L.stage_ms[i] = ms / (float)std::max(L.g, 1);       // per image of the group
    }
    if (L.nms_before_decode) std::swap(L.stage_ms[ST_LSTM], L.stage_ms[ST_NMS2]);    // events 6..7 timed the NMS, 7..8 the decode
  }
  L.have_times = !L.ran_graph;                            // a replayed graph carries no stage events
  const int P = L.P;
  const size_t stride = pack_stride(ctx, P, L.pending_feats);
  for (int i = 0; i < L.g; ++i) {
    const char* hs = static_cast<const char*>(L.host_stage) + i * stride;
    if (const uint32_t fw = *reinterpret_cast<const uint32_t*>(hs + 68); fw != 0u) {
      (void)hipMemset(ctx->fault_dev, 0, 64);
      L.pending = nullptr;
      if (fw == 2u) {
        nms_set_scan_band(0);       // every NMS window back on the chunk scan (process-wide)
        return ctx->fail(DC_E_HIP, "NMS band scan: a hand-off between the waves of nms_scan_band_kernel did not arrive within the spin "
                                   "bound; the band scan is now off (chunk scan for every window) -- repeat the call");
      }
      ctx->tail_mode = 1;           // stop sharing tiles between workgroups on this ctx
      return ctx->fail(DC_E_HIP, "stream-K: a workgroup's partner never published its partial tile within the spin bound (GPU "
                                 "shared with another job?); this ctx now uses the K-split tail plan -- repeat the call");
    }
    int K = *reinterpret_cast<const int32_t*>(hs);
    if (L.pending_feats) {
      K = std::min(K, L.pending_capacity);
      if (L.pending_k_dst) L.pending_k_dst[i] = K;
      if (L.pending_box_dst) memcpy(L.pending_box_dst + (size_t)i * L.pending_capacity * 4, hs + 256, (size_t)K * 16);
      if (L.pending_feat_dst)
        memcpy(L.pending_feat_dst + (size_t)i * L.pending_capacity * ctx->D, hs + 256 + (size_t)P * 20, (size_t)K * ctx->D * 4);
    } else if (L.pending) {
      dc_result* r = L.pending + i;
      K = std::min(K, (int)r->capacity);
      r->K = K;
      r->T = ctx->T;
      if (r->boxes) memcpy(r->boxes, hs + 256, (size_t)K * 16);
      if (r->scores) memcpy(r->scores, hs + 256 + (size_t)P * 16, (size_t)K * 4);
      if (r->tokens) memcpy(r->tokens, hs + 256 + (size_t)P * 20, (size_t)K * ctx->T * 4);
    }
  }
  L.pending = nullptr;
  return DC_OK;
}

Lane& lane0(dc_ctx* ctx) {
  if (ctx->lanes.empty()) ctx->lanes.emplace_back(new Lane());
  return *ctx->lanes[0];
}
int lane0_stream(dc_ctx* ctx, hipStream_t* s) {
  Lane& L = lane0(ctx);
  if (L.stream == nullptr) {
    HIPCHK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&L.aux, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&L.aux2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&L.ev_fork2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_join2, hipEventDisableTiming));
    for (auto& e : L.ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming));
  }
  *s = L.stream;
  return DC_OK;
}

}  // namespace

int dc_ctx_device(const dc_ctx* ctx) { return ctx->device; }
void dc_ctx_set_error(dc_ctx* ctx, const char* msg) { ctx->err = msg; g_last_error = msg; }

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

int dc_create(dc_ctx** out, int hip_device) {
  if (!out) { g_last_error = "dc_create: null out"; return DC_E_INVALID; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_last_error = std::string("dc_create: no HIP device available (") + hipGetErrorString(e) +
                   "); this library has no CPU fallback";
    return DC_E_HIP;
  }
  if (hip_device < 0 || hip_device >= ndev) { g_last_error = "dc_create: bad device index"; return DC_E_INVALID; }
  e = hipSetDevice(hip_device);
  if (e != hipSuccess) { g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return DC_E_HIP; }
  dc_ctx* ctx = new dc_ctx();
  ctx->device = hip_device;
  *out = ctx;
  return DC_OK;
}

void dc_destroy(dc_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  for (auto& lp : ctx->lanes) {
    Lane& L = *lp;
    if (L.arena.p) hipFree(L.arena.p);
    if (L.beam_base) hipFree(L.beam_base);
    if (L.gexec) (void)hipGraphExecDestroy(L.gexec);
    if (L.host_stage) hipHostFree(L.host_stage);
    for (auto& ev : L.ev) if (ev) hipEventDestroy(ev);
    if (L.ev_fork) hipEventDestroy(L.ev_fork);
    if (L.ev_join) hipEventDestroy(L.ev_join);
    if (L.ev_fork2) hipEventDestroy(L.ev_fork2);
    if (L.ev_join2) hipEventDestroy(L.ev_join2);
    if (L.aux) hipStreamDestroy(L.aux);
    if (L.aux2) hipStreamDestroy(L.aux2);
    if (L.stream) hipStreamDestroy(L.stream);
  }
  for (void* p : ctx->owned) hipFree(p);
  if (ctx->pre_src.p) hipFree(ctx->pre_src.p);
  if (ctx->pre_scratch.p) hipFree(ctx->pre_scratch.p);
  if (ctx->pre_taps.p) hipFree(ctx->pre_taps.p);
  for (auto e : ctx->prof_pool) hipEventDestroy(e);
  delete ctx;
}

const char* dc_last_error(const dc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

// LocalizationLayer:setTestArgs (LocalizationLayer.lua:233-238)
int dc_set_localization_test_args(dc_ctx* ctx, int clip_boxes, float nms_thresh, int max_proposals) {
  if (!ctx) return DC_E_INVALID;
  if (max_proposals != -1 && (max_proposals <= 0 || max_proposals > (1 << 20)))
    return ctx->fail(DC_E_UNSUPPORTED, "num_proposals must be -1 (uncapped) or in [1,1048576] (got %d)", max_proposals);
  ctx->clip_boxes = clip_boxes != 0;
  ctx->rpn_nms_thresh = nms_thresh;
  ctx->num_proposals = max_proposals;
  return DC_OK;
}

// DenseCapModel:setTestArgs (DenseCapModel.lua:185-191): the layer's setTestArgs WITHOUT a clip_boxes key (-> true) + opt.final_nms_thresh
int dc_set_test_args(dc_ctx* ctx, float rpn_nms_thresh, float final_nms_thresh, int num_proposals) {
  if (!ctx) return DC_E_INVALID;
  DCCHK(dc_set_localization_test_args(ctx, 1, rpn_nms_thresh, num_proposals));
  ctx->final_nms_thresh = final_nms_thresh;
  return DC_OK;
}

int dc_set_lanes(dc_ctx* ctx, int lanes) {
  if (!ctx) return DC_E_INVALID;
  if (lanes < 1 || lanes > 4) return ctx->fail(DC_E_INVALID, "dc_set_lanes: lanes must be in [1,4]");
  ctx->max_lanes = lanes;
  // numerics depend only on this setting, never on how many images a call happens to carry: with one lane the
  // last partial round of a layer is K-split (different fp32 summation order for those rows)
  ctx->serial_mode = lanes == 1;
  return DC_OK;
}

int dc_set_group(dc_ctx* ctx, int images) {
  if (!ctx) return DC_E_INVALID;
  if (images < 0 || images > kGemmMaxGroup) return ctx->fail(DC_E_INVALID, "dc_set_group: 0 (default = 1) .. %d images per group", kGemmMaxGroup);
  ctx->group = images;
  return DC_OK;
}

// beam search needs one vocabulary row in the top-k kernel's LDS and at least `beam` words to choose from
static int check_beam_fits(dc_ctx* ctx, int beam_size) {
  if (beam_size <= 0 || !ctx->have_weights) return DC_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(DC_E_HIP, "hipSetDevice failed");
  const size_t vmax = beam_topk_max_vocab();
  if ((size_t)(ctx->V + 1) > vmax)
    return ctx->fail(DC_E_UNSUPPORTED, "beam search: a vocabulary of %d words does not fit the top-k kernel's LDS row on this device (max %zu)",
                     ctx->V, vmax > 0 ? vmax - 1 : 0);
  if (beam_size > ctx->V + 1)
    return ctx->fail(DC_E_UNSUPPORTED, "beam search: beam_size %d exceeds the %d output words", beam_size, ctx->V + 1);
  return DC_OK;
}

int dc_set_beam_size(dc_ctx* ctx, int beam_size) {
  if (!ctx) return DC_E_INVALID;
  if (beam_size < 0 || beam_size > 32) return ctx->fail(DC_E_UNSUPPORTED, "dc_set_beam_size: beam_size must be in [0,32] (got %d)", beam_size);
  DCCHK(check_beam_fits(ctx, beam_size));          // before dc_load_weights the check runs there instead
  ctx->beam_size = beam_size;
  return DC_OK;
}

// the bf16 planes of every registered weight matrix (once; ~0.85 GB next to the 0.7 GB of fp32 weights)
static int make_weight_planes(dc_ctx* ctx) {
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s;
  DCCHK(lane0_stream(ctx, &s));
  for (auto& e : ctx->planes) {
    if (e.planes != nullptr) continue;
    DCCHK(dev_alloc(ctx, reinterpret_cast<void**>(&e.planes), (size_t)3 * e.rows * e.K * 2));
    KCHK(launch_split_planes(e.W, e.planes, e.rows, e.K, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return DC_OK;
}

int dc_set_math_mode(dc_ctx* ctx, int mode) {
  if (!ctx) return DC_E_INVALID;
  if (mode != DC_MATH_FP32 && mode != DC_MATH_SPLIT_BF16)
    return ctx->fail(DC_E_INVALID, "dc_set_math_mode: 0 (fp32 MFMA) or 1 (split-bf16), got %d", mode);
  if (mode == DC_MATH_SPLIT_BF16 && ctx->have_weights) DCCHK(make_weight_planes(ctx));
  ctx->math_mode = mode;
  return DC_OK;
}

int dc_set_graph_replay(dc_ctx* ctx, int on) {
  if (!ctx) return DC_E_INVALID;
  ctx->graphs = on != 0;
  return DC_OK;
}

int dc_set_caption_order(dc_ctx* ctx, int after_final_nms) {
  if (!ctx) return DC_E_INVALID;
  ctx->captions_after_final_nms = after_final_nms != 0;
  return DC_OK;
}

int dc_load_weights(dc_ctx* ctx, const dc_weights* w) {
  if (!ctx || !w) return DC_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->have_weights) return ctx->fail(DC_E_STATE, "weights already loaded; create a new ctx");
  if (w->num_anchors <= 0 || w->rpn_hidden % 32 || w->fc_dim % 256 || w->enc_size % 32 || w->rnn_size % 32 ||
      w->vocab_size <= 0 || w->seq_length <= 0)
    return ctx->fail(DC_E_INVALID, "dc_load_weights: unsupported dimensions");
  ctx->k = w->num_anchors; ctx->R = w->rpn_hidden; ctx->V = w->vocab_size; ctx->T = w->seq_length;
  ctx->E = w->enc_size; ctx->Hd = w->rnn_size; ctx->D = w->fc_dim;
  memcpy(ctx->fc, w->field_centers, sizeof ctx->fc);
  hipStream_t s;
  DCCHK(lane0_stream(ctx, &s));
  const int k = ctx->k, R = ctx->R, V = ctx->V, E = ctx->E, Hd = ctx->Hd, D = ctx->D;
  float* tmp = nullptr;
  // VGG convs: conv1_1 stays OIHW (direct kernel); the rest are repacked to (Cout, 9*Cin)
  DCCHK(upload(ctx, &ctx->conv_w[0], w->conv_w[0], (size_t)64 * 27));
  DCCHK(upload(ctx, &ctx->conv_b[0], w->conv_b[0], 64));
  for (int i = 1; i < DC_NUM_VGG_CONVS; ++i) {
    const size_t n = (size_t)kVgg[i].cout * kVgg[i].cin * 9;
    if (!w->conv_w[i] || !w->conv_b[i]) return ctx->fail(DC_E_INVALID, "null conv weight %d", i);
    HIPCHK(hipMalloc(&tmp, n * 4));
    HIPCHK(hipMemcpy(tmp, w->conv_w[i], n * 4, hipMemcpyHostToDevice));
    DCCHK(dev_alloc(ctx, (void**)&ctx->conv_w[i], n * 4));
    KCHK(launch_pack_conv3x3(tmp, ctx->conv_w[i], kVgg[i].cout, kVgg[i].cin, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(tmp));
    DCCHK(upload(ctx, &ctx->conv_b[i], w->conv_b[i], kVgg[i].cout));
  }
  {  // RPN conv
    const size_t n = (size_t)R * 512 * 9;
    if (!w->rpn_conv_w) return ctx->fail(DC_E_INVALID, "null rpn_conv_w");
    HIPCHK(hipMalloc(&tmp, n * 4));
    HIPCHK(hipMemcpy(tmp, w->rpn_conv_w, n * 4, hipMemcpyHostToDevice));
    DCCHK(dev_alloc(ctx, (void**)&ctx->rpn_w, n * 4));
    KCHK(launch_pack_conv3x3(tmp, ctx->rpn_w, R, 512, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(tmp));
    DCCHK(upload(ctx, &ctx->rpn_b, w->rpn_conv_b, R));
  }
  {  // fused 1x1 heads: rows [0,4k) box, [4k,6k) score
    if (!w->rpn_box_w || !w->rpn_score_w || !w->rpn_box_b || !w->rpn_score_b)
      return ctx->fail(DC_E_INVALID, "null rpn head weight");
    std::vector<float> hw((size_t)6 * k * R), hb((size_t)6 * k);
    memcpy(hw.data(), w->rpn_box_w, (size_t)4 * k * R * 4);
    memcpy(hw.data() + (size_t)4 * k * R, w->rpn_score_w, (size_t)2 * k * R * 4);
    memcpy(hb.data(), w->rpn_box_b, (size_t)4 * k * 4);
    memcpy(hb.data() + 4 * k, w->rpn_score_b, (size_t)2 * k * 4);
    DCCHK(upload(ctx, &ctx->heads_w, hw.data(), hw.size()));
    DCCHK(upload(ctx, &ctx->heads_b, hb.data(), hb.size()));
  }
  {  // fc6: permute K from (c,i,j) to (i,j,c) to match the channels-last RoI features
    const size_t n = (size_t)D * 512 * 49;
    if (!w->fc6_w) return ctx->fail(DC_E_INVALID, "null fc6_w");
    HIPCHK(hipMalloc(&tmp, n * 4));
    HIPCHK(hipMemcpy(tmp, w->fc6_w, n * 4, hipMemcpyHostToDevice));
    DCCHK(dev_alloc(ctx, (void**)&ctx->fc6_w, n * 4));
    KCHK(launch_permute_fc6(tmp, ctx->fc6_w, D, 512, 49, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(tmp));
    DCCHK(upload(ctx, &ctx->fc6_b, w->fc6_b, D));
  }
  DCCHK(upload(ctx, &ctx->fc7_w, w->fc7_w, (size_t)D * D));
  DCCHK(upload(ctx, &ctx->fc7_b, w->fc7_b, D));
  {
    if (!w->obj_w || !w->boxreg_w || !w->obj_b || !w->boxreg_b) return ctx->fail(DC_E_INVALID, "null head weight");
    std::vector<float> h5((size_t)5 * D), b5(5);
    memcpy(h5.data(), w->obj_w, (size_t)D * 4);
    memcpy(h5.data() + D, w->boxreg_w, (size_t)4 * D * 4);
    b5[0] = w->obj_b[0];
    memcpy(b5.data() + 1, w->boxreg_b, 16);
    DCCHK(upload(ctx, &ctx->head5_w, h5.data(), h5.size()));
    DCCHK(upload(ctx, &ctx->head5_b, b5.data(), 5));
  }
  DCCHK(upload(ctx, &ctx->enc_w, w->lm_enc_w, (size_t)E * D));
  DCCHK(upload(ctx, &ctx->enc_b, w->lm_enc_b, E));
  DCCHK(upload(ctx, &ctx->lstm_b, w->lstm_b, (size_t)4 * Hd));
  {  // torch-rnn weight (E+Hd, 4Hd): rows [0,E) = Wx, [E,E+Hd) = Wh; kernels want (N,K)
    float* lw = nullptr;
    DCCHK(upload(ctx, &lw, w->lstm_w, (size_t)(E + Hd) * 4 * Hd));
    DCCHK(dev_alloc(ctx, (void**)&ctx->wxT, (size_t)4 * Hd * E * 4));
    DCCHK(dev_alloc(ctx, (void**)&ctx->whT, (size_t)4 * Hd * Hd * 4));
    KCHK(launch_transpose2d(lw, ctx->wxT, E, 4 * Hd, s));
    KCHK(launch_transpose2d(lw + (size_t)E * 4 * Hd, ctx->whT, Hd, 4 * Hd, s));
    // xg[v] = b + Emb[v].Wx for every token of the LookupTable (V+2 rows): the input half of the
    // gate pre-activation of every decode step becomes a row gather.
    float* emb = nullptr;
    DCCHK(upload(ctx, &emb, w->lm_emb, (size_t)(V + 2) * E));
    DCCHK(dev_alloc(ctx, (void**)&ctx->xg, (size_t)(V + 2) * 4 * Hd * 4));
    DCCHK(linear(ctx, s, emb, ctx->wxT, ctx->lstm_b, ctx->xg, V + 2, 4 * Hd, E, 0));
    HIPCHK(hipStreamSynchronize(s));
  }
  {  // decode-step operand: [Wout (V+1 rows); zero rows to a multiple of 64; Wh^T (4Hd rows)], all with K = Hd -- one GEMM
     // launch per step produces the vocabulary arg-max of h_t and h_t.Wh for the next step's gates
    if (!w->lm_out_w || !w->lm_out_b) return ctx->fail(DC_E_INVALID, "null lm_out weight");
    ctx->V1pad = (V + 1 + 63) / 64 * 64;
    const size_t rows = (size_t)ctx->V1pad + 4 * Hd;
    DCCHK(dev_alloc(ctx, (void**)&ctx->dec_w, rows * Hd * 4));
    HIPCHK(hipMemset(ctx->dec_w, 0, rows * Hd * 4));
    HIPCHK(hipMemcpy(ctx->dec_w, w->lm_out_w, (size_t)(V + 1) * Hd * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->dec_w + (size_t)ctx->V1pad * Hd, ctx->whT, (size_t)4 * Hd * Hd * 4, hipMemcpyDeviceToDevice));
    ctx->out_w = ctx->dec_w;
  }
  DCCHK(upload(ctx, &ctx->out_b, w->lm_out_b, (size_t)V + 1));
  DCCHK(upload(ctx, &ctx->anchors, w->anchors, (size_t)2 * k));
  HIPCHK(hipStreamSynchronize(s));
  prof_collect(ctx);
  // the matrices the split-bf16 mode can take (one image's problem fills the chip): their planes are made when the mode is on
  for (int i = 1; i < DC_NUM_VGG_CONVS; ++i)
    ctx->planes.push_back({ctx->conv_w[i], (size_t)kVgg[i].cout, 9 * kVgg[i].cin, nullptr});
  ctx->planes.push_back({ctx->fc6_w, (size_t)D, 49 * 512, nullptr});
  ctx->planes.push_back({ctx->fc7_w, (size_t)D, D, nullptr});
  ctx->planes.push_back({ctx->dec_w, (size_t)ctx->V1pad + 4 * Hd, Hd, nullptr});
  if (ctx->math_mode == DC_MATH_SPLIT_BF16) DCCHK(make_weight_planes(ctx));
  ctx->have_weights = true;
  ctx->weights_epoch += 1;                 // captured graphs hold the old weight pointers
  if (int rc = check_beam_fits(ctx, ctx->beam_size); rc != DC_OK) {   // dc_set_beam_size came first: validate it now
    ctx->beam_size = 0;
    return rc;
  }
  return DC_OK;
}

// A failed enqueue/harvest must not leave lanes that still point at the caller's result buffers (the caller frees them
// when the error surfaces) or work in flight on a workspace that the next call may rebuild: wait for every lane and
// forget its pending destination without copying anything out.
static void drain_lanes(dc_ctx* ctx) {
  for (auto& lp : ctx->lanes) {
    Lane& L = *lp;
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    if (L.aux) (void)hipStreamSynchronize(L.aux);
    if (L.aux2) (void)hipStreamSynchronize(L.aux2);
    L.busy = false;
    L.pending = nullptr;
    L.pending_box_dst = nullptr; L.pending_feat_dst = nullptr; L.pending_k_dst = nullptr;
  }
  (void)hipGetLastError();
}
#define DCCHK_DRAIN(expr)                 \
  do {                                    \
    int _r = (expr);                      \
    if (_r != DC_OK) { drain_lanes(ctx); return _r; } \
  } while (0)

// The reference puts no limit on the image size (box_utils.lua:154-256 handles any number of boxes).  What bounds an
// image here is the 32-bit operand addressing of the convolution kernels: the largest activation, conv1_x's
// (H, W, 64) fp32 map, must stay below 4 GiB -- about 16 Mpx.
static int check_image_size(dc_ctx* ctx, int H, int W, const char* who) {
  if ((size_t)H * W * 64 * 4 >= 0xffffe000ull)
    return ctx->fail(DC_E_UNSUPPORTED, "%s: a %dx%d image exceeds the conv kernels' 4 GiB activation limit (~16 Mpx)", who, W, H);
  return DC_OK;
}

// images of a group share one 32-bit operand offset space in conv1_x (the pooled conv counts window slots)
static int clamp_group(const dc_ctx* ctx, int G, int H, int W) {
  if (ctx->plan_mode < 0 ? ctx->serial_mode : ctx->plan_mode == 1) return 1;       // single-image planning: images travel alone
  const size_t rows1 = std::max((size_t)H * W, (size_t)4 * ((H + 1) / 2) * ((W + 1) / 2));
  while (G > 1 && (size_t)G * rows1 * 64 * 4 >= 0xffffe000ull) --G;
  return std::max(G, 1);
}
// length of the run of equal-sized images starting at i that may travel as one group
static int group_run(const dc_ctx* ctx, const int* H, const int* W, int i, int n) {
  const int G = clamp_group(ctx, std::max(1, ctx->group), H[i], W[i]);
  int g = 1;
  while (g < G && i + g < n && H[i + g] == H[i] && W[i + g] == W[i]) ++g;
  return g;
}

static int forward_common(dc_ctx* ctx, const float* imgs, int n, int H, int W, int on_dev, dc_result* outs) {
  if (!ctx) return DC_E_INVALID;
  if (!ctx->have_weights) return ctx->fail(DC_E_STATE, "dc_forward_*: weights not loaded");
  if (!imgs || !outs || n <= 0 || H < 32 || W < 32) return ctx->fail(DC_E_INVALID, "dc_forward_*: bad arguments");
  DCCHK(check_image_size(ctx, H, W, "dc_forward_*"));
  HIPCHK(hipSetDevice(ctx->device));
  const int P = effective_proposals(ctx, H, W);
  for (int i = 0; i < n; ++i)
    if (outs[i].capacity <= 0) return ctx->fail(DC_E_INVALID, "dc_result.capacity must be > 0");
  // images travel in groups of G through a lane (dc_set_group): the group's dense stages share launches
  // Single-image planning (dc_set_lanes(1)) shares a layer's partial last round along K -- plans made for ONE image's tile
  // count, which a group does not have: images travel alone there, so that results never depend on the group.
  // A group's conv1_x activation shares one 32-bit offset space; the pooled conv counts window slots (4 per 2x2 window: a
  // pixel more per odd side) -- the same count the launch itself checks.
  const int G = clamp_group(ctx, std::max(1, std::min(ctx->group > 0 ? ctx->group : 1, n)), H, W);
  const int ngroups = (n + G - 1) / G;
  const int nl = std::min(ngroups, ctx->max_lanes);
  while ((int)ctx->lanes.size() < nl) ctx->lanes.emplace_back(new Lane());
  for (int l = 0; l < nl; ++l) DCCHK(lane_prepare(ctx, *ctx->lanes[l], H, W, P, G));
  const size_t img_elems = (size_t)3 * H * W;
  double enq_ms = 0;
  for (int gi = 0; gi < ngroups; ++gi) {
    const int i = gi * G, g = std::min(G, n - i);
    Lane& L = *ctx->lanes[gi % nl];
    DCCHK_DRAIN(harvest(ctx, L));
    L.pending = &outs[i];
    const auto t0 = std::chrono::steady_clock::now();
    DCCHK_DRAIN(enqueue_forward(ctx, L, imgs + img_elems * i, g, on_dev, false));
    enq_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  for (int l = 0; l < nl; ++l) DCCHK_DRAIN(harvest(ctx, *ctx->lanes[l]));
  ctx->host_enqueue_ms = enq_ms / n;       // host time spent enqueueing, per image (dc_debug_fetch "host_enqueue_us")
  prof_collect(ctx);
  return DC_OK;
}

int dc_forward_images(dc_ctx* ctx, const float* const* imgs, const int* H, const int* W, int n, int on_dev, dc_result* outs) {
  if (!ctx) return DC_E_INVALID;
  if (!ctx->have_weights) return ctx->fail(DC_E_STATE, "dc_forward_images: weights not loaded");
  if (!imgs || !H || !W || !outs || n <= 0) return ctx->fail(DC_E_INVALID, "dc_forward_images: bad arguments");
  for (int i = 0; i < n; ++i) {
    if (!imgs[i] || H[i] < 32 || W[i] < 32 || outs[i].capacity <= 0)
      return ctx->fail(DC_E_INVALID, "dc_forward_images: image %d: null pointer, side below 32 px or capacity <= 0", i);
    DCCHK(check_image_size(ctx, H[i], W[i], "dc_forward_images"));
  }
  HIPCHK(hipSetDevice(ctx->device));
  const int nl = std::min(n, ctx->max_lanes);
  while ((int)ctx->lanes.size() < nl) ctx->lanes.emplace_back(new Lane());
  // runs of consecutive images of ONE size travel as groups (dc_set_group), like the images of dc_forward_batch
  int gi = 0;
  for (int i = 0; i < n; ++gi) {
    const int g = group_run(ctx, H, W, i, n);
    Lane& L = *ctx->lanes[gi % nl];
    DCCHK_DRAIN(harvest(ctx, L));                 // the lane's previous images leave before its workspace is re-carved
    DCCHK_DRAIN(lane_prepare(ctx, L, H[i], W[i], effective_proposals(ctx, H[i], W[i]), std::max(g, L.H == H[i] && L.W == W[i] ? L.G : 1)));
    L.pending = &outs[i];
    DCCHK_DRAIN(enqueue_forward(ctx, L, nullptr, g, on_dev, false, imgs + i));
    i += g;
  }
  for (int l = 0; l < nl; ++l) DCCHK_DRAIN(harvest(ctx, *ctx->lanes[l]));
  prof_collect(ctx);
  return DC_OK;
}

int dc_forward_test(dc_ctx* ctx, const float* img_chw, int H, int W, int img_on_device, dc_result* out) {
  return forward_common(ctx, img_chw, 1, H, W, img_on_device, out);
}
int dc_forward_batch(dc_ctx* ctx, const float* imgs, int n, int H, int W, int imgs_on_device, dc_result* outs) {
  return forward_common(ctx, imgs, n, H, W, imgs_on_device, outs);
}

int dc_extract_features(dc_ctx* ctx, const float* img_chw, int H, int W, int img_on_device, int capacity,
                        float* boxes, float* feats, int32_t* K) {
  if (!ctx) return DC_E_INVALID;
  if (!ctx->have_weights) return ctx->fail(DC_E_STATE, "dc_extract_features: weights not loaded");
  if (!img_chw || capacity <= 0 || H < 32 || W < 32) return ctx->fail(DC_E_INVALID, "dc_extract_features: bad arguments");
  DCCHK(check_image_size(ctx, H, W, "dc_extract_features"));
  HIPCHK(hipSetDevice(ctx->device));
  Lane& L = lane0(ctx);
  DCCHK(harvest(ctx, L));
  DCCHK(lane_prepare(ctx, L, H, W, effective_proposals(ctx, H, W), std::max(L.G, 1)));
  L.pending = nullptr;
  L.pending_capacity = capacity; L.pending_box_dst = boxes; L.pending_feat_dst = feats; L.pending_k_dst = K;
  DCCHK_DRAIN(enqueue_forward(ctx, L, img_chw, 1, img_on_device, true));
  DCCHK_DRAIN(harvest(ctx, L));
  prof_collect(ctx);
  return DC_OK;
}

int dc_extract_features_images(dc_ctx* ctx, const float* const* imgs, const int* H, const int* W, int n, int on_dev,
                               int capacity, float* boxes, float* feats, int32_t* K) {
  if (!ctx) return DC_E_INVALID;
  if (!ctx->have_weights) return ctx->fail(DC_E_STATE, "dc_extract_features_images: weights not loaded");
  if (!imgs || !H || !W || n <= 0 || capacity <= 0 || !boxes || !feats || !K)
    return ctx->fail(DC_E_INVALID, "dc_extract_features_images: bad arguments");
  for (int i = 0; i < n; ++i) {
    if (!imgs[i] || H[i] < 32 || W[i] < 32)
      return ctx->fail(DC_E_INVALID, "dc_extract_features_images: image %d: null pointer or side below 32 px", i);
    DCCHK(check_image_size(ctx, H[i], W[i], "dc_extract_features_images"));
  }
  HIPCHK(hipSetDevice(ctx->device));
  const int nl = std::min(n, ctx->max_lanes);
  while ((int)ctx->lanes.size() < nl) ctx->lanes.emplace_back(new Lane());
  // runs of equal-sized images travel as groups here too (round-4 verdict: extractFeatures always ran groups of one)
  int gi = 0;
  for (int i = 0; i < n; ++gi) {
    const int g = group_run(ctx, H, W, i, n);
    Lane& L = *ctx->lanes[gi % nl];
    DCCHK_DRAIN(harvest(ctx, L));
    DCCHK_DRAIN(lane_prepare(ctx, L, H[i], W[i], effective_proposals(ctx, H[i], W[i]), std::max(g, L.H == H[i] && L.W == W[i] ? L.G : 1)));
    L.pending = nullptr;
    L.pending_capacity = capacity;
    L.pending_box_dst = boxes + (size_t)i * capacity * 4;           // image j of the group: j * capacity rows further on
    L.pending_feat_dst = feats + (size_t)i * capacity * ctx->D;
    L.pending_k_dst = K + i;
    DCCHK_DRAIN(enqueue_forward(ctx, L, nullptr, g, on_dev, true, imgs + i));
    i += g;
  }
  for (int l = 0; l < nl; ++l) DCCHK_DRAIN(harvest(ctx, *ctx->lanes[l]));
  prof_collect(ctx);
  return DC_OK;
}

// run_model.lua:67-74 on the device.  Synchronous; runs on the ctx's primary stream.
int dc_preprocess_size(int H0, int W0, int image_size, int* H, int* W) {
  if (H0 <= 0 || W0 <= 0 || image_size <= 0 || !H || !W) return DC_E_INVALID;
  preprocess_scaled_size(H0, W0, image_size, H, W);
  return (*H >= 1 && *W >= 1) ? DC_OK : DC_E_INVALID;
}

int dc_preprocess_u8(dc_ctx* ctx, const uint8_t* rgb_hwc, int H0, int W0, int on_device, int image_size, float* out_chw_dev,
                     uint8_t* scaled_rgb_dev) {
  if (!ctx) return DC_E_INVALID;
  if (!rgb_hwc || !out_chw_dev || H0 <= 0 || W0 <= 0 || image_size <= 0 || (size_t)H0 * W0 > ((size_t)1 << 28))
    return ctx->fail(DC_E_INVALID, "dc_preprocess_u8: bad arguments");
  int oh = 0, ow = 0;
  preprocess_scaled_size(H0, W0, image_size, &oh, &ow);
  if (oh < 1 || ow < 1) return ctx->fail(DC_E_INVALID, "dc_preprocess_u8: a %dx%d image scaled to %d leaves no pixels", W0, H0, image_size);
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s;
  DCCHK(lane0_stream(ctx, &s));
  auto grow = [&](DevBuf& b, size_t bytes) -> int {
    if (b.bytes >= bytes) return DC_OK;
    if (b.p) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipFree(b.p)); b = DevBuf(); }
    HIPCHK(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return DC_OK;
  };
  const uint8_t* src = rgb_hwc;
  if (!on_device) {
    const size_t nb = (size_t)H0 * W0 * 3;
    DCCHK(grow(ctx->pre_src, nb));
    HIPCHK(hipMemcpyAsync(ctx->pre_src.p, rgb_hwc, nb, hipMemcpyHostToDevice, s));
    src = static_cast<const uint8_t*>(ctx->pre_src.p);
  }
  DCCHK(grow(ctx->pre_scratch, preprocess_scratch_bytes(H0, W0, oh, ow)));
  const int key[4] = {H0, W0, oh, ow};
  if (ctx->pre_taps.p == nullptr || memcmp(key, ctx->pre_key, sizeof key) != 0) {
    // new sizes: rebuild the tables (round-5 advisor finding: they were rebuilt, uploaded from pageable memory and waited
    // for on every frame).  The host copy is replaced only after the stream has drained its previous upload.
    HIPCHK(hipStreamSynchronize(s));
    ctx->pre_taps_host.resize(preprocess_taps_bytes(oh, ow));
    preprocess_make_taps(H0, W0, oh, ow, ctx->pre_taps_host.data());
    DCCHK(grow(ctx->pre_taps, ctx->pre_taps_host.size()));
    HIPCHK(hipMemcpyAsync(ctx->pre_taps.p, ctx->pre_taps_host.data(), ctx->pre_taps_host.size(), hipMemcpyHostToDevice, s));
    memcpy(ctx->pre_key, key, sizeof key);
  }
  const float mean_bgr[3] = {103.939f, 116.779f, 123.68f};          // run_model.lua:73
  KCHK(launch_preprocess_u8(src, H0, W0, oh, ow, mean_bgr, ctx->pre_scratch.p, ctx->pre_taps.p, out_chw_dev, scaled_rgb_dev, s));
  HIPCHK(hipStreamSynchronize(s));
  return DC_OK;
}

int dc_stage_times(dc_ctx* ctx, const char** names, float* ms, int max_stages) {
  if (!ctx) return DC_E_INVALID;
  if (ctx->lanes.empty() || !ctx->lanes[0]->have_times) return 0;
  const int n = std::min(max_stages, (int)ST_COUNT);
  for (int i = 0; i < n; ++i) {
    if (names) names[i] = kStageNames[i];
    if (ms) ms[i] = ctx->lanes[0]->stage_ms[i];
  }
  return n;
}

int dc_mfma_profile(dc_ctx* ctx, int reset, int64_t* launches, double* total_ms, double* total_flops) {
  if (!ctx) return DC_E_INVALID;
  if (launches) *launches = ctx->prof_launches;
  if (total_ms) *total_ms = ctx->prof_ms;
  if (total_flops) *total_flops = ctx->prof_flops;
  if (reset) {
    ctx->prof_launches = 0; ctx->prof_ms = 0; ctx->prof_flops = 0;
    ctx->prof = reset > 0;   // reset=1: (re)start profiling; reset=-1: stop
  }
  return DC_OK;
}

int64_t dc_debug_fetch(dc_ctx* ctx, const char* name, void* host_buf, int64_t capacity_bytes) {
  if (!ctx || !name || !host_buf) return DC_E_INVALID;
  if (ctx->lanes.empty() || !ctx->lanes[0]->arena.p) return ctx->fail(DC_E_STATE, "no forward has run yet");
  Lane& L = *ctx->lanes[0];
  const int P = L.P;
  struct Ent { const char* n; const void* p; int64_t elems; int esize; };
  const Ent tab[] = {
      {"feat_hwc", L.feat, (int64_t)L.fh * L.fw * 512, 4},
      {"rpn_heads", L.heads, (int64_t)L.fh * L.fw * 6 * ctx->k, 4},
      {"rpn_boxes", L.rpn_boxes, (int64_t)L.A * 4, 4},
      {"rpn_x1y1x2y2", L.rpn_xyxy, (int64_t)L.A * 4, 4},
      {"rpn_p", L.rpn_p, (int64_t)L.A, 4},
      {"rpn_valid", L.rpn_valid, (int64_t)L.A, 1},
      {"rpn_nms_idx", L.picks1, (int64_t)P, 4},
      {"rpn_nms_count", L.count1, 1, 4},
      {"roi_boxes", L.roi_boxes, (int64_t)P * 4, 4},
      {"roi_feats", L.roi_feats, (int64_t)P * 49 * 512, 4},
      {"codes", L.codes, (int64_t)P * ctx->D, 4},
      {"obj", L.obj, (int64_t)P, 4},
      {"final_trans", L.final_trans, (int64_t)P * 4, 4},
      {"final_boxes", L.final_boxes, (int64_t)P * 4, 4},
      {"seq", L.seq, (int64_t)P * ctx->T, 4},
      {"final_nms_idx", L.picks2, (int64_t)P, 4},
      {"final_nms_count", L.count2, 1, 4},
      // language-model state of image 0's rows (reference order: row = RoI; captions after the final NMS: row = final rank)
      {"lm_enc", L.enc, (int64_t)P * ctx->E, 4},
      {"lm_h", L.hstate, (int64_t)P * ctx->Hd, 4},
      {"lm_c", L.cstate, (int64_t)P * ctx->Hd, 4},
      {"survivor_rows", L.surv_total, 1, 4},
  };
  if (strcmp(name, "host_enqueue_us") == 0) {
    if (capacity_bytes < 4) return ctx->fail(DC_E_INVALID, "dc_debug_fetch: buffer too small");
    *static_cast<int32_t*>(host_buf) = (int32_t)(ctx->host_enqueue_ms * 1000.0);
    return 1;
  }
  if (strcmp(name, "graph_launches") == 0 || strcmp(name, "graph_captures") == 0) {
    if (capacity_bytes < 4) return ctx->fail(DC_E_INVALID, "dc_debug_fetch: buffer too small");
    *static_cast<int32_t*>(host_buf) = name[6] == 'l' ? ctx->graph_launches : ctx->graph_captures;
    return 1;
  }
  if (strcmp(name, "graph_replay_on") == 0) {
    if (capacity_bytes < 4) return ctx->fail(DC_E_INVALID, "dc_debug_fetch: buffer too small");
    *static_cast<int32_t*>(host_buf) = ctx->graphs ? 1 : 0;
    if (!ctx->graphs && !ctx->graph_note.empty()) ctx->err = ctx->graph_note;      // readable through dc_last_error
    return 1;
  }
  if (strcmp(name, "arena_allocs") == 0) {
    if (capacity_bytes < 4) return ctx->fail(DC_E_INVALID, "dc_debug_fetch: buffer too small");
    *static_cast<int32_t*>(host_buf) = ctx->arena_allocs;
    return 1;
  }
  for (const Ent& e : tab) {
    if (strcmp(e.n, name) == 0) {
      const int64_t bytes = e.elems * e.esize;
      if (bytes > capacity_bytes) return ctx->fail(DC_E_INVALID, "dc_debug_fetch: buffer too small (%lld needed)", (long long)bytes);
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(hipStreamSynchronize(L.stream));
      HIPCHK(hipMemcpy(host_buf, e.p, bytes, hipMemcpyDeviceToHost));
      return e.elems;
    }
  }
  return ctx->fail(DC_E_INVALID, "dc_debug_fetch: unknown name '%s'", name);
}

int dc_debug_plan_gemm(int64_t M, int64_t N, int64_t K, int64_t plan_M, int conv_cin, int argmax, int serial_mode,
                       int32_t* out8) {
  if (!out8 || M <= 0 || N <= 0 || K <= 0 || K % 32 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || plan_M < 0 ||
      plan_M > M || (conv_cin != 0 && (conv_cin % 32 || K != 9 * (int64_t)conv_cin)))
    return DC_E_INVALID;
  GemmDesc d;
  static float dummy;                                        // only the POINTER's presence matters to the planner
  d.M = (int)M; d.N = (int)N; d.K = (int)K; d.ldc = (int)N; d.plan_M = (int)plan_M;
  if (conv_cin) { d.conv = 1; d.Cin = conv_cin; }
  if (argmax) d.amax_val = &dummy;
  GemmPlan pl;
  // (DC_PLAN_CU_COUNT: "what would a part with this many CUs be given" -- honoured by THIS query only, never by a launch)
  int cu_override = 0;
  if (const char* e = getenv("DC_PLAN_CU_COUNT")) {
    const int nc = atoi(e);
    if (nc >= 8 && nc <= 4096) cu_override = nc;
  }
  set_planning_cu_override(cu_override);
  mfma_gemm_plan(d, serial_mode != 0, 0, kSplitkWsFloats, &pl);
  set_planning_cu_override(0);
  const int32_t v[8] = {pl.kind, pl.route, pl.stages, pl.splitk, pl.m_split, pl.sk_wgs, pl.sk_np, pl.tail_splitk};
  memcpy(out8, v, sizeof(v));
  return DC_OK;
}

int dc_debug_set(dc_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return DC_E_INVALID;
  if (strcmp(name, "beam_chunk_floats") == 0) {
    if (value < 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: beam_chunk_floats must be >= 1");
    ctx->beam_chunk_floats = value;
    return DC_OK;
  }
  if (strcmp(name, "nms_band") == 0) {                 // process-wide: 0 = nms_scan_kernel for every NMS window
    if (value < 0 || value > 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: nms_band must be 0 or 1");
    nms_set_scan_band((int)value);
    return DC_OK;
  }
  if (strcmp(name, "v2_stages") == 0) {
    if (value != 0 && value != 2 && value != 3) return ctx->fail(DC_E_INVALID, "dc_debug_set: v2_stages must be 0, 2 or 3");
    ctx->v2_stages = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "force_cfg") == 0) {
    if (value < 0 || value > 6) return ctx->fail(DC_E_INVALID, "dc_debug_set: force_cfg must be 0..6");
    ctx->force_cfg = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "plan_mode") == 0) {
    if (value < -1 || value > 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: plan_mode must be -1, 0 or 1");
    ctx->plan_mode = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "stagger") == 0) {
    if (value < 0 || value > 4096) return ctx->fail(DC_E_INVALID, "dc_debug_set: stagger must be 0..4096 (64-cycle sleeps)");
    ctx->stagger = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "epi_wide") == 0) {
    if (value < 0 || value > 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: epi_wide must be 0 or 1");
    ctx->epi_wide = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "walk") == 0) {
    if (value < 0 || value > 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: walk must be 0 or 1");
    ctx->walk = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "bf3_all") == 0) {
    if (value < 0 || value > 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: bf3_all must be 0 or 1");
    ctx->bf3_all = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "bf3_presplit") == 0) {
    if (value < 0 || value > 1) return ctx->fail(DC_E_INVALID, "dc_debug_set: bf3_presplit must be 0 or 1");
    ctx->bf3_presplit = (int)value;
    return DC_OK;
  }
  if (strcmp(name, "tail_mode") == 0) {
    if (value < 0 || value > 2) return ctx->fail(DC_E_INVALID, "dc_debug_set: tail_mode must be 0, 1 or 2");
    ctx->tail_mode = (int)value;
    return DC_OK;
  }
  return ctx->fail(DC_E_INVALID, "dc_debug_set: unknown name '%s'", name);
}

// ---- memory helpers ----------------------------------------------------------------------
int dc_malloc(dc_ctx* ctx, void** dev_ptr, size_t bytes) {
  if (!ctx || !dev_ptr) return DC_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMalloc(dev_ptr, bytes ? bytes : 16));
  return DC_OK;
}
int dc_free(dc_ctx* ctx, void* dev_ptr) {
  if (!ctx) return DC_E_INVALID;
  HIPCHK(hipFree(dev_ptr));
  return DC_OK;
}
int dc_memcpy_h2d(dc_ctx* ctx, void* d, const void* h, size_t bytes) {
  if (!ctx) return DC_E_INVALID;
  HIPCHK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
  return DC_OK;
}
int dc_memcpy_d2h(dc_ctx* ctx, void* h, const void* d, size_t bytes) {
  if (!ctx) return DC_E_INVALID;
  HIPCHK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
  return DC_OK;
}
int dc_synchronize(dc_ctx* ctx) {
  if (!ctx) return DC_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  return DC_OK;
}


// stream-K fault word after a synchronised per-op call
static int check_sk_fault(dc_ctx* ctx, const char* who) {
  if (ctx->fault_dev == nullptr) return DC_OK;
  uint32_t f = 0;
  if (hipMemcpy(&f, ctx->fault_dev, 4, hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(DC_E_HIP, "%s: fault word read failed", who);
  if (f == 0) return DC_OK;
  (void)hipMemset(ctx->fault_dev, 0, 64);
  if (f == 2u) {
    nms_set_scan_band(0);
    return ctx->fail(DC_E_HIP, "%s: a hand-off between the waves of nms_scan_band_kernel did not arrive within the spin bound; the band "
                               "scan is now off -- repeat the call", who);
  }
  ctx->tail_mode = 1;
  return ctx->fail(DC_E_HIP, "%s: stream-K partner never published its partial tile within the spin bound", who);
}

// split-bf16 mode on a per-op call: the caller's weight matrix gets temporary planes (the model's own are made once)
static int op_planes(dc_ctx* ctx, hipStream_t s, const float* W, int N, int K, uint16_t** out) {
  *out = nullptr;
  if (ctx->math_mode != DC_MATH_SPLIT_BF16 || !ctx->bf3_presplit || K % 32) return DC_OK;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(out), (size_t)3 * N * K * 2));
  if (hipError_t e = launch_split_planes(W, *out, (size_t)N, K, s); e != hipSuccess) {
    (void)hipFree(*out);                     // (round-5 advisor finding: the planes leaked when the launch failed)
    *out = nullptr;
    return ctx->fail(DC_E_HIP, "launch_split_planes: %s", hipGetErrorString(e));
  }
  return DC_OK;
}

// ---- per-op entry points --------------------------------------------------------------------
#define OP_PROLOGUE()                         \
  if (!ctx) return DC_E_INVALID;              \
  HIPCHK(hipSetDevice(ctx->device));          \
  hipStream_t s;                              \
  DCCHK(lane0_stream(ctx, &s));
#define OP_EPILOGUE()                 \
  HIPCHK(hipStreamSynchronize(s));    \
  prof_collect(ctx);                  \
  return DC_OK;

int dc_op_chw_to_hwc(dc_ctx* ctx, const float* in, float* out, int C, int H, int W) {
  OP_PROLOGUE(); KCHK(launch_chw_to_hwc(in, out, C, H, W, s)); OP_EPILOGUE();
}
int dc_op_hwc_to_chw(dc_ctx* ctx, const float* in, float* out, int C, int H, int W) {
  OP_PROLOGUE(); KCHK(launch_hwc_to_chw(in, out, C, H, W, s)); OP_EPILOGUE();
}
int dc_op_pack_conv3x3_weights(dc_ctx* ctx, const float* w, float* out, int Cout, int Cin) {
  OP_PROLOGUE(); KCHK(launch_pack_conv3x3(w, out, Cout, Cin, s)); OP_EPILOGUE();
}
int dc_op_conv3x3(dc_ctx* ctx, const float* in, const float* w, const float* b, float* out, int n_img, int H, int W,
                  int Cin, int Cout, int relu) {
  OP_PROLOGUE();
  if (Cin % 32 || n_img <= 0 || H <= 0 || W <= 0 || Cout <= 0)
    return ctx->fail(DC_E_INVALID, "dc_op_conv3x3: need Cin %% 32 == 0 and positive sizes");
  float* ws = nullptr;
  HIPCHK(hipMalloc((void**)&ws, kSplitkWsFloats * 4));
  uint16_t* pl = nullptr;
  if (int rc0 = op_planes(ctx, s, w, Cout, 9 * Cin, &pl); rc0 != DC_OK) { (void)hipFree(ws); return rc0; }
  int rc = conv3x3(ctx, s, in, w, b, out, n_img, H, W, Cin, Cout, relu, Ws{ws, kSplitkWsFloats}, pl);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(ws);
  if (pl) (void)hipFree(pl);
  prof_collect(ctx);
  if (rc != DC_OK) return rc;
  if (e2 != hipSuccess) return ctx->fail(DC_E_HIP, "dc_op_conv3x3 sync: %s", hipGetErrorString(e2));
  return check_sk_fault(ctx, "dc_op_conv3x3");
}
int dc_op_conv3x3_relu_pool(dc_ctx* ctx, const float* in, const float* w, const float* b, float* out, int H, int W,
                            int Cin, int Cout) {
  OP_PROLOGUE();
  if (Cin % 32 || Cout % 4 || H <= 0 || W <= 0 || Cout <= 0)
    return ctx->fail(DC_E_INVALID, "dc_op_conv3x3_relu_pool: need Cin %% 32 == 0, Cout %% 4 == 0 and positive sizes");
  float* ws = nullptr;
  HIPCHK(hipMalloc((void**)&ws, kSplitkWsFloats * 4));
  uint16_t* pl = nullptr;
  if (int rc0 = op_planes(ctx, s, w, Cout, 9 * Cin, &pl); rc0 != DC_OK) { (void)hipFree(ws); return rc0; }
  int rc = conv3x3_pool(ctx, s, in, w, b, out, 1, H, W, Cin, Cout, 1, Ws{ws, kSplitkWsFloats}, pl);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(ws);
  if (pl) (void)hipFree(pl);
  prof_collect(ctx);
  if (rc != DC_OK) return rc;
  if (e2 != hipSuccess) return ctx->fail(DC_E_HIP, "dc_op_conv3x3_relu_pool sync: %s", hipGetErrorString(e2));
  return check_sk_fault(ctx, "dc_op_conv3x3_relu_pool");
}
int dc_op_conv3x3_c3(dc_ctx* ctx, const float* in, const float* w, const float* b, float* out, int H, int W, int Cout,
                     int relu) {
  OP_PROLOGUE();
  if (Cout != 64) return ctx->fail(DC_E_UNSUPPORTED, "dc_op_conv3x3_c3: Cout must be 64");
  KCHK(launch_conv3x3_c3(in, w, b, out, 1, H, W, Cout, relu, s));
  OP_EPILOGUE();
}
int dc_op_maxpool2x2_ceil(dc_ctx* ctx, const float* in, float* out, int n_img, int H, int W, int C) {
  OP_PROLOGUE(); KCHK(launch_maxpool2x2_ceil(in, out, n_img, H, W, C, s)); OP_EPILOGUE();
}
int dc_op_linear(dc_ctx* ctx, const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                 int relu) {
  OP_PROLOGUE();
  if (K % 32 || M <= 0 || N <= 0) return ctx->fail(DC_E_INVALID, "dc_op_linear: need K %% 32 == 0");
  float* ws = nullptr;
  HIPCHK(hipMalloc((void**)&ws, kSplitkWsFloats * 4));
  uint16_t* pl = nullptr;
  if (int rc0 = op_planes(ctx, s, W, N, K, &pl); rc0 != DC_OK) { (void)hipFree(ws); return rc0; }
  int rc = linear(ctx, s, A, W, bias, C, M, N, K, relu, Ws{ws, kSplitkWsFloats}, 0, pl);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(ws);
  if (pl) (void)hipFree(pl);
  prof_collect(ctx);
  if (rc != DC_OK) return rc;
  if (e2 != hipSuccess) return ctx->fail(DC_E_HIP, "dc_op_linear sync: %s", hipGetErrorString(e2));
  return check_sk_fault(ctx, "dc_op_linear");
}
int dc_op_make_anchors(dc_ctx* ctx, float* out, int h, int w, float x0, float y0, float sx, float sy,
                       const float* anchors_dev, int k) {
  OP_PROLOGUE(); KCHK(launch_make_anchors(out, h, w, x0, y0, sx, sy, anchors_dev, k, s)); OP_EPILOGUE();
}
int dc_op_apply_box_transform(dc_ctx* ctx, const float* boxes, const float* trans, float* out, int n) {
  OP_PROLOGUE(); KCHK(launch_apply_box_transform(boxes, trans, out, n, s)); OP_EPILOGUE();
}
int dc_op_clip_boxes(dc_ctx* ctx, const float* boxes, float* clipped, uint8_t* valid, int n, float x_min, float y_min,
                     float x_max, float y_max) {
  OP_PROLOGUE(); KCHK(launch_clip_boxes(boxes, clipped, valid, n, x_min, y_min, x_max, y_max, s)); OP_EPILOGUE();
}
int dc_op_xcycwh_to_x1y1x2y2(dc_ctx* ctx, const float* boxes, float* out, int n) {
  OP_PROLOGUE(); KCHK(launch_xcycwh_to_x1y1x2y2(boxes, out, n, s)); OP_EPILOGUE();
}
int dc_op_box_iou(dc_ctx* ctx, const float* b1, const float* b2, float* out, int B1, int B2, int convention) {
  OP_PROLOGUE(); KCHK(launch_box_iou(b1, b2, out, B1, B2, convention, s)); OP_EPILOGUE();
}
int dc_op_rpn_decode(dc_ctx* ctx, const float* heads, int h, int w, int k, const float* anchors_dev, float x0,
                     float y0, float sx, float sy, int img_h, int img_w, float* boxes, float* anchors_out,
                     float* trans, float* x1y1x2y2, float* p, uint8_t* valid) {
  OP_PROLOGUE();
  KCHK(launch_rpn_decode(heads, 1, h, w, k, anchors_dev, x0, y0, sx, sy, img_h, img_w, boxes, anchors_out, trans,
                         x1y1x2y2, p, valid, 1, s));
  OP_EPILOGUE();
}
int dc_op_nms(dc_ctx* ctx, const float* boxes, const float* scores, const uint8_t* valid, int n, float thresh,
              int max_boxes, int32_t* picks, int32_t* count) {
  OP_PROLOGUE();
  if (n < 0) return ctx->fail(DC_E_INVALID, "dc_op_nms: n must be >= 0");
  if (n == 0) { HIPCHK(hipMemsetAsync(count, 0, 4, s)); OP_EPILOGUE(); }
  void* base = nullptr;
  HIPCHK(hipMalloc(&base, nms_workspace_bytes(n)));
  NmsWorkspace ws;
  nms_workspace_bind(ws, base, n);
  ensure_fault_word(ctx);
  hipError_t e = launch_nms(ws, boxes, scores, valid, n, nullptr, thresh, max_boxes, picks, count, s, ctx->fault_dev);
  hipError_t e2 = hipStreamSynchronize(s);
  hipFree(base);
  if (e != hipSuccess) return ctx->fail(DC_E_HIP, "nms launch: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return ctx->fail(DC_E_HIP, "nms sync: %s", hipGetErrorString(e2));
  return check_sk_fault(ctx, "dc_op_nms");
}
int dc_op_bilinear_roi_pool(dc_ctx* ctx, const float* feat_hwc, int h, int w, int C, const float* boxes, int B,
                            int img_h, int img_w, int HH, int WW, float* out, int out_layout) {
  OP_PROLOGUE();
  if (C % 4 || B <= 0 || HH < 2 || WW < 2) return ctx->fail(DC_E_INVALID, "dc_op_bilinear_roi_pool: bad shape");
  KCHK(launch_bilinear_roi_pool(feat_hwc, h, w, C, boxes, B, nullptr, img_h, img_w, HH, WW, out, out_layout, s));
  OP_EPILOGUE();
}
int dc_op_lm_sample(dc_ctx* ctx, const float* codes, int n, int32_t* tokens) {
  OP_PROLOGUE();
  if (!ctx->have_weights) return ctx->fail(DC_E_STATE, "dc_op_lm_sample: weights not loaded");
  if (n <= 0) return ctx->fail(DC_E_INVALID, "dc_op_lm_sample: n must be > 0");
  Lane& L = lane0(ctx);
  // private scratch for n rows
  const int E = ctx->E, Hd = ctx->Hd, V1 = ctx->V + 1;
  struct Sav { float *enc, *gates, *h, *c, *logits; int32_t* tok; } sv{L.enc, L.gates, L.hstate, L.cstate, L.logits, L.tok};
  const size_t bytes = al((size_t)n * E * 4) + al((size_t)n * 4 * Hd * 4) + 2 * al((size_t)n * Hd * 4) +
                       al((size_t)n * std::max(V1, ctx->V1pad / 16) * 4) + al((size_t)n * 4);
  char* base = nullptr;
  HIPCHK(hipMalloc((void**)&base, bytes));
  char* p = base;
  L.enc = (float*)p; p += al((size_t)n * E * 4);
  L.gates = (float*)p; p += al((size_t)n * 4 * Hd * 4);
  L.hstate = (float*)p; p += al((size_t)n * Hd * 4);
  L.cstate = (float*)p; p += al((size_t)n * Hd * 4);
  L.logits = (float*)p; p += al((size_t)n * std::max(V1, ctx->V1pad / 16) * 4);
  L.tok = (int32_t*)p;
  int rc = ctx->beam_size > 0 ? lm_beamsearch(ctx, L, codes, n, tokens, s) : lm_sample(ctx, L, codes, n, 0, nullptr, tokens);
  hipError_t e2 = hipStreamSynchronize(s);
  L.enc = sv.enc; L.gates = sv.gates; L.hstate = sv.h; L.cstate = sv.c; L.logits = sv.logits; L.tok = sv.tok;
  hipFree(base);
  prof_collect(ctx);
  if (rc != DC_OK) return rc;
  if (e2 != hipSuccess) return ctx->fail(DC_E_HIP, "lm_sample sync: %s", hipGetErrorString(e2));
  return DC_OK;
}

}  // extern "C"
