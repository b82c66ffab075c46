// Shared declarations for libdensecap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/densecap.h"
#include "../../include/densecap_debug.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// TH / THNN of the reference's era compute exp, sigmoid and tanh of a FloatTensor through the C DOUBLE functions and cast
// the result to float (TH: LAB_IMPLEMENT_BASIC_FUNCTION(exp, exp), TH_sigmoid(double) = 1.0 / (1.0 + exp(-x)), tanh;
// docs/SEMANTICS.md).  Both sides of the parity tests use this form: a double result is within an ulp of the true value
// in any libm, so the float it rounds to is the same on the device and in the oracle (bar one case in ~2^29).
__device__ __forceinline__ float th_expf(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float th_sigmoidf(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float th_tanhf(float x) { return (float)tanh((double)x); }

// (shared by boxes.hip and the recognition heads: one definition of the conversion every NMS input goes through)
__device__ __forceinline__ void corners(float xc, float yc, float w, float h, float& x1, float& y1, float& x2,
                                        float& y2) {
  // box_utils.xcycwh_to_x1y1x2y2 (box_utils.lua:288-291): x0 = ((w-1)/2)*-1 + xc ; x1 = (w-1)/2 + xc
  const float hw = __fdiv_rn(__fsub_rn(w, 1.f), 2.f);
  const float hh = __fdiv_rn(__fsub_rn(h, 1.f), 2.f);
  x1 = __fadd_rn(-hw, xc);
  y1 = __fadd_rn(-hh, yc);
  x2 = __fadd_rn(hw, xc);
  y2 = __fadd_rn(hh, yc);
}

// ---- ctx accessors for the other translation units (densecap.hip) ------------------
int dc_ctx_device(const dc_ctx* ctx);
void dc_ctx_set_error(dc_ctx* ctx, const char* msg);

// ---- MFMA contraction engine (mfma_gemm.hip) --------------------------------
struct GemmDesc {
  const float* A = nullptr;   // dense: (M,K) row-major; conv: (nimg,H,W,Cin) HWC activations
  const float* W = nullptr;   // (N,K) row-major, K contiguous
  const float* bias = nullptr;  // (N) or null
  float* C = nullptr;         // (M,ldc)
  int M = 0, N = 0, K = 0, ldc = 0;
  int relu = 0;
  // conv3x3 implicit GEMM (mode 1): M = nimg*H*Wd, K = 9*Cin
  int conv = 0, H = 0, Wd = 0, Cin = 0;
  // conv + fused 2x2/2 ceil-mode max-pool (one image): the M index walks POOL WINDOWS -- m = 4*window + 2*dy + dx with
  // window = wy*ceil(Wd/2) + wx in raster order of the pooled map, pixel (2wy+dy, 2wx+dx) -- so that the four pixels of
  // a window are four consecutive accumulator registers of one lane; M = 4*ceil(H/2)*ceil(Wd/2); slots outside the
  // image (odd H or Wd) read zeros and are left out of the max.  C is the POOLED map: C[window*ldc + n].
  int pool = 0;
  // Rows of ONE image when the launch carries a group of images (0 = M): every routing decision that changes the fp32
  // summation order (K-split kernel vs the sequential-K kernels, split-K factor) and the tile shape are planned on this
  // count, so an image's numbers do not depend on how many images share its launches.
  int plan_M = 0;
  // arithmetic: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, the default and the only mode results are bit-compared in);
  // 1 = split-bf16 (dc_set_math_mode(1)): both operands split into three bf16 planes in registers, six of the nine partial
  // products accumulated in fp32 on v_mfma_f32_32x32x16_bf16 -- fp32-class accuracy at 2.67x the matrix rate
  // 2 = the same with the B operand (weights) split ONCE into planes in HBM (launch_split_planes): `sk_slots` then carries the
  // planes pointer and `sk_np` the rows of the plane matrix (stream-K, whose fields these are, does not exist in this mode)
  int bf3 = 0;                // (sits in what was the alignment hole in front of `rowterm`: no other field moved)
  // optional gathered row term (LSTM input gates): C[m][n] += rowterm[rowidx[m]*rowterm_ld + n]
  const float* rowterm = nullptr;
  const int32_t* rowidx = nullptr;   // values are 1-based token ids -> row = id-1
  int rowterm_ld = 0;
  // optional fused row arg-max (vocab projection): instead of storing C, every (row, 32-column half of a 64-column tile)
  // writes its best (value, column) to amax_val/amax_idx[m * amax_ld + 2 * tile_n + half] (columns ascend with the slot);
  // amax_ld >= 2 * ceil(N / 64); C may be null.
  float* amax_val = nullptr;
  int32_t* amax_idx = nullptr;
  int amax_ld = 0;
  // arg-max over a PREFIX of the columns (decode step: W = [vocab rows; pad; Wh rows], both products share A = h_t):
  // columns [0, amax_cols) (a multiple of 64) get the arg-max epilogue, of which [0, amax_n) are real vocabulary
  // entries; columns [amax_cols, N) are stored raw to C[m*ldc + (n - amax_cols)].  amax_cols = 0: every column.
  int amax_cols = 0;
  int amax_n = 0;
  // optional device-side row count: effective M = min(M, *m_dev); workgroups past it exit at once
  const int32_t* m_dev = nullptr;
  // split-K (K-split 128x128 kernel only): `splitk` workgroups share one tile, each sums a contiguous K range
  // and writes its raw partial tile to splitk_ws[(slice*M + m)*N + n]; launch_splitk_reduce finishes the job
  int splitk = 1;
  float* splitk_ws = nullptr;
  // stream-K over the tiles of rows [m_begin, M) (K-split kernel, launch_mfma_gemm_sk): sk_lo[0..sk_wgs] = unit offsets
  // of the workgroups on the line of K units (unit = 2 K-tiles, sk_np units per tile, tiles n-fastest); sk_slots =
  // sk_wgs partial tiles of 128x128 floats; sk_flags[0..sk_wgs) zeroed before the launch; sk_fault = sticky device word
  // raised when an owner gives up waiting for a partner (checked by the host with the results)
  int stages = 0;             // LDS ring depth of the 128x64-tile kernel: 0 = by tile count, 3 (two workgroups per CU) or 2 (three per CU)
  int force_cfg = 0;          // measurement hook (dc_debug_set "force_cfg"): 0 = planned, 1 = 128x128, 2 = 128x64, 3 = 64x64 tiles; 4 = planned tiles, no split-K;
                              // 5 = 128x128 tiles on the v2 kernel with a two-stage ring (two workgroups per CU); 6 = the K-split 128x128 kernel whatever K
  int stagger = 0;            // measurement hook (dc_debug_set "stagger"): workgroups start after a pseudo-random pause of up to this many 64-cycle sleeps
  int epi_wide = 1;           // interior tiles of plain epilogues leave as 16-byte stores staged through the wave's LDS (dc_debug_set "epi_wide": 0 = dword stores)
  int walk = 0;               // measurement hook (dc_debug_set "walk"): 128x64 launches run one workgroup per slot that walks its tiles
  const int* sk_lo = nullptr;
  int sk_np = 0;
  float* sk_slots = nullptr;
  unsigned* sk_flags = nullptr;
  unsigned* sk_fault = nullptr;
  // row window (K-split kernel only): tiles cover rows [m_begin, M); a_rows = rows of the whole A operand
  // (extent of the conv input for the buffer descriptor) when M is only a prefix, 0 = M
  int m_begin = 0;
  int a_rows = 0;
};
// C = act(sum_s ws[s] + bias): fixed summation order s = 0..S-1
// m_dev (optional): device-side row count, rows >= *m_dev are left alone (their partials were never written)
hipError_t launch_splitk_reduce(const float* ws, int S, const float* bias, float* C, int M, int N, int ldc, int relu,
                                hipStream_t s, const int32_t* m_dev = nullptr);
// same for a pooled conv (GemmDesc::pool): ws rows are window-ordered slots [m_begin, m_begin+M); C_pooled is the base
// of the whole pooled map
hipError_t launch_splitk_reduce_pool(const float* ws, int S, const float* bias, float* C_pooled, int m_begin, int M, int N,
                                     int ldc, int H, int Wd, int relu, hipStream_t s);
// can this conv problem take GemmDesc::pool (operands within the kernels' 32-bit buffer offsets)?
bool mfma_gemm_can_pool(const GemmDesc& d);
// Largest image group a launch may carry (dc_set_group).  Split-K factors are functions of ONE image's problem, so the
// partial-output workspace test assumes a group of this size: an image gets the same factor in every group.
constexpr int kGemmMaxGroup = 8;
// split factor launch_mfma_gemm would like for this problem (1 = none)
int mfma_gemm_splitk(const GemmDesc& d, size_t ws_floats);      // ws_floats: capacity of the partial-output workspace
// Tail plan for problems whose 128x128 tile count is not a multiple of the 256 CUs: rows [0, m_split) run as
// whole tiles (full rounds), rows [m_split, M) are split `tail_splitk` ways along K so the last partial round
// fills the chip.  Returns false when it does not pay.
bool mfma_gemm_tail_plan(const GemmDesc& d, int* m_split, int* tail_splitk);
// Stream-K plan for the last, partial round of 128x128 tiles: rows [0, m_split) as whole tiles, the tiles of rows
// [m_split, M) shared along K by `wgs` workgroups (np = K units per tile).  false when it does not pay.
bool mfma_gemm_sk_plan(const GemmDesc& d, int* m_split, int* wgs, int* np);
size_t mfma_gemm_sk_ws_floats(int wgs);
hipError_t launch_mfma_gemm_sk(const GemmDesc& d, int wgs, int np, float* ws, hipStream_t stream);
// force the K-split 128x128 kernel (honours m_begin / a_rows / splitk)
hipError_t launch_mfma_gemm_ks(const GemmDesc& d, hipStream_t stream);
// How a contraction is carried out (mfma_gemm_plan): kind = what run_gemm does around the launch, route = the kernel family.
enum { GEMM_PLAN_PLAIN = 0, GEMM_PLAN_SPLITK = 1, GEMM_PLAN_STREAMK = 2, GEMM_PLAN_TAIL = 3 };
enum { GEMM_ROUTE_KS = 0, GEMM_ROUTE_V2_128x64 = 1, GEMM_ROUTE_V2_128x128 = 2, GEMM_ROUTE_V2_64x64 = 3 };
struct GemmPlan {
  int kind = GEMM_PLAN_PLAIN, route = GEMM_ROUTE_KS;
  int stages = 0;            // LDS ring depth of a 128x64 launch (2 or 3), 0 otherwise
  int splitk = 1;            // GEMM_PLAN_SPLITK
  int m_split = 0;           // GEMM_PLAN_STREAMK / GEMM_PLAN_TAIL: rows [0, m_split) run as whole tiles
  int sk_wgs = 0, sk_np = 0; // GEMM_PLAN_STREAMK
  int tail_splitk = 1;       // GEMM_PLAN_TAIL
};
// Pure function of the problem, the scheduling mode and the workspace size (no device needed beyond its CU count).
void mfma_gemm_plan(const GemmDesc& d, bool serial_mode, int tail_mode, size_t ws_floats, GemmPlan* plan);
// Launches the fp32 MFMA kernel on `stream`; returns hipSuccess or the launch error.
hipError_t launch_mfma_gemm(const GemmDesc& d, hipStream_t stream);
double gemm_flops(const GemmDesc& d);
// does the split-bf16 mode pay for this contraction (enough tiles for ONE image to fill the chip without K sharing)?
bool mfma_gemm_bf3_pays(const GemmDesc& d);

// ---- element-wise / layout kernels (elementwise.hip) ---------------------------
hipError_t launch_chw_to_hwc(const float* in, float* out, int C, int H, int W, hipStream_t s);
hipError_t launch_hwc_to_chw(const float* in, float* out, int C, int H, int W, hipStream_t s);
hipError_t launch_pack_conv3x3(const float* w_oihw, float* w_packed, int Cout, int Cin, hipStream_t s);
int device_cu_count();      // compute units of the current device (256 on MI355X); cached per device
void set_planning_cu_override(int cus);   // dc_debug_plan_gemm only: this thread's planners see `cus` CUs until reset with 0
hipError_t launch_conv3x3_c3(const float* in_chw, const float* w_oihw, const float* bias, float* out_hwc, int nimg,
                             int H, int W, int Cout, int relu, hipStream_t s);
hipError_t launch_maxpool2x2_ceil(const float* in, float* out, int nimg, int H, int W, int C, hipStream_t s);
hipError_t launch_transpose2d(const float* in, float* out, int rows, int cols, hipStream_t s);
// fc6 weight (N, C*HH*WW) with k = c*HW + p  ->  k' = p*C + c
hipError_t launch_permute_fc6(const float* in, float* out, int N, int C, int HW, hipStream_t s);
// LSTM pointwise: gates (n,4Hd) [i f o g], c (n,Hd) in/out, h (n,Hd) out
hipError_t launch_lstm_pointwise(const float* gates, float* c, float* h, int n, const int32_t* n_dev, int Hd,
                                  int zero_c, hipStream_t s);
hipError_t launch_fill_i32(int32_t* p, int32_t v, int n, hipStream_t s);
// idx[i] = i for i < cap, *count_out = *count_in
hipError_t launch_iota_count(int32_t* idx, int32_t* count_out, const int32_t* count_in, int cap, hipStream_t s);
// One decode step's row-wise tail (LanguageModel.lua:316-335 between two GEMMs), one workgroup per row
// (a wave per row was measured slower: 11.4 vs 8.3 us -- the row's 2048 gate values want 256 lanes in flight):
//   pval != null: tok = 1 + argmax over the row's `ntiles` partials (first max on ties), seq[m*T+t] = tok;
//                 else tok = fixed_tok (START, or 0 = no input-gate row term);
//   gates = (tok ? xg[(tok-1)*4Hd ..] : 0) + gates_pre[m]  (same association as torch-rnn: (b + x.Wx) + h.Wh);
//   [i f o g] -> c' = f*c + i*g (c = 0 if zero_c), h' = o*tanh(c').  gates_pre == null: arg-max only.
hipError_t launch_lstm_step_tail(const float* pval, const int32_t* pidx, int ntiles, int ld, int fixed_tok,
                                 const float* xg, const float* gates_pre, float* c, float* h, int n,
                                 const int32_t* n_dev, int Hd, int zero_c, int32_t* seq, int T, int t, hipStream_t s);
// split-bf16 mode: W (N, K) fp32 -> 3 x N x K bf16 planes, k permuted per 32-tile as the kernels read them (elementwise.hip)
hipError_t launch_split_planes(const float* W, uint16_t* planes, size_t N, int K, hipStream_t s);
// objectness + box regression heads + final ApplyBoxTransform (DenseCapModel.lua:134,139-140)
// final_xyxy (optional): the final boxes as corners too (box_utils.xcycwh_to_x1y1x2y2), what the final NMS reads
hipError_t launch_recog_heads(const float* codes, const float* w5 /*(5,D): obj, 4 boxreg*/, const float* b5,
                              const float* roi_boxes, float* obj, float* trans, float* final_boxes, float* final_xyxy,
                              int n, int D, hipStream_t s);

// hipFuncAttributeMaxDynamicSharedMemorySize per (device, kernel) -- mfma_gemm.hip
hipError_t ensure_dyn_lds(const void* fn, size_t bytes);

// ---- beam search row kernels (beam.hip; LanguageModel.lua:170-290) --------------------------
size_t beam_topk_max_vocab();     // largest V+1 whose row fits the top-k kernel's LDS on the current device
hipError_t launch_beam_logsoftmax_topk(const float* logits, int rows, int V1, int ld, const uint8_t* finished, int k,
                                       float* top_lp, int32_t* top_idx, hipStream_t s);
hipError_t launch_beam_init(const float* top_lp, const int32_t* top_idx, int nprop, int beam, int T, int END,
                            float* beam_lp, int32_t* beams, int32_t* parent, int32_t* cur_tok, uint8_t* finished,
                            hipStream_t s);
hipError_t launch_beam_merge(const float* top_lp, const int32_t* top_idx, const float* beam_lp_in,
                             const int32_t* beams_in, int nprop, int beam, int T, int t, int END, float* beam_lp_out,
                             int32_t* beams_out, int32_t* parent, int32_t* cur_tok, uint8_t* finished, hipStream_t s);
hipError_t launch_beam_gather_state(const float* h_in, const float* c_in, const int32_t* parent, int rows, int beam,
                                    int src_per_prop, int Hd, float* h_out, float* c_out, hipStream_t s);
hipError_t launch_beam_best(const int32_t* beams, int nprop, int beam, int T, int32_t* seq, hipStream_t s);

// ---- box pipeline (boxes.hip) ------------------------------------------------------
hipError_t launch_make_anchors(float* out, int h, int w, float x0, float y0, float sx, float sy,
                               const float* anchors, int k, hipStream_t s);
hipError_t launch_apply_box_transform(const float* boxes, const float* trans, float* out, int n, hipStream_t s);
hipError_t launch_clip_boxes(const float* boxes, float* clipped, uint8_t* valid, int n, float x_min, float y_min,
                             float x_max, float y_max, hipStream_t s);
hipError_t launch_xcycwh_to_x1y1x2y2(const float* boxes, float* out, int n, hipStream_t s);
hipError_t launch_box_iou(const float* b1, const float* b2, float* out, int B1, int B2, int convention,
                          hipStream_t s);
// (nimg images of a group side by side: every per-image tensor follows the previous image's)
hipError_t launch_rpn_decode(const float* heads, int nimg, int h, int w, int k, const float* anchors, float x0, float y0,
                             float sx, float sy, int img_h, int img_w, float* boxes, float* anchors_out,
                             float* trans, float* x1y1x2y2, float* p, uint8_t* valid, int clip, hipStream_t s);
struct NmsWorkspace {
  // device scratch, sized for n_cap boxes (see boxes.hip)
  int n_cap = 0;
  uint32_t* keys = nullptr;       // (n) order-preserving sort keys
  int32_t* tmp_idx = nullptr;     // (n) bucketed (unordered inside a bucket) indices
  uint32_t* tmp_key = nullptr;    // (n) their keys, same order
  int32_t* order = nullptr;       // (n) sorted position -> original index
  float* sboxes = nullptr;        // (n,4) boxes in sorted order
  float* sarea = nullptr;         // (n)
  int32_t* pick_pos = nullptr;    // (n) sorted positions of the picks so far
  int32_t* hist = nullptr;        // (NMS_BUCKETS)
  int32_t* cursor = nullptr;      // (NMS_BUCKETS)
  int32_t* off = nullptr;         // (NMS_BUCKETS)
  int32_t* state = nullptr;       // {count, done}
  int32_t* nvalid = nullptr;      // (1)
  unsigned long long* removed0 = nullptr;  // (ceil(n/64)) bits suppressed by picks of earlier windows
  unsigned long long* mask = nullptr;      // window bit mask
  unsigned long long* nearband = nullptr;  // (4096, 4) words c, c-1, c-2, c-3 of every row of a window of <= 4096 rows
  size_t mask_words = 0;
  size_t zero_bytes = 0;          // hist..removed0 are contiguous and zeroed per call
};
size_t nms_workspace_bytes(int n);
hipError_t nms_workspace_bind(NmsWorkspace& ws, void* base, int n);
// n_dev (optional device int32) overrides n at run time (n is then the capacity)
// fault (optional): the ctx's sticky device word; nms_scan_band_kernel stores 2 there when a hand-off between its waves did not
// arrive within the spin bound (the host then fails the call and switches the band scan off)
hipError_t launch_nms(NmsWorkspace& ws, const float* boxes, const float* scores, const uint8_t* valid, int n,
                      const int32_t* n_dev, float thresh, int max_boxes, int32_t* picks, int32_t* count,
                      hipStream_t s, uint32_t* fault = nullptr);
void nms_set_scan_band(int on);          // test hook: 0 = the per-chunk scan kernel for every window
// out[i] = src[idx[i]] rows of `width` floats for i < *count (rows >= *count zero-filled up to cap)
hipError_t launch_gather_rows(const float* src, const int32_t* idx, const int32_t* count, int cap, int width,
                              float* out, hipStream_t s);
hipError_t launch_gather_rows_i32(const int32_t* src, const int32_t* idx, const int32_t* count, int cap, int width,
                                  int32_t* out, hipStream_t s);

// the fc7 rows the final NMS kept, of all images of a group, packed into one row block in pick order; *total = their number
hipError_t launch_survivor_compact(const float* codes, const int32_t* picks, const int32_t* count, int count_stride, int nimg,
                                   int P, int D, float* out, int32_t* total, hipStream_t s);

// the results of a group of images gathered by their final-NMS picks into packed records (see final_pack_kernel);
// tok_gather: 1 = token rows are per RoI (gathered by pick), 0 = already in final order per image, 2 = final order, packed over the group
hipError_t launch_final_pack(const float* final_boxes, const float* obj, const int32_t* tokens, int tok_gather,
                             const float* codes, const int32_t* picks, const int32_t* count, int count_stride,
                             const uint32_t* fault, int nimg, int P, int T, int D, void* pack, size_t stride, hipStream_t s);

// ---- bilinear RoI pooling (roipool.hip) ---------------------------------------------
// group form: nimg feature maps feat_stride floats apart, B rows of boxes / output per image, live counts
// B_dev[image * bdev_stride]; pick != null: box b of an image = src_boxes[image * src_stride + pick[b]], also written to boxes
hipError_t launch_bilinear_roi_pool_group(const float* feat_hwc, size_t feat_stride, int nimg, int h, int w, int C,
                                          float* boxes, int B, const int32_t* B_dev, int bdev_stride,
                                          const int32_t* pick, const float* src_boxes, size_t src_stride, int img_h,
                                          int img_w, int HH, int WW, float* out, int out_layout, hipStream_t s);
hipError_t launch_bilinear_roi_pool(const float* feat_hwc, int h, int w, int C, const float* boxes, int B,
                                    const int32_t* B_dev, int img_h, int img_w, int HH, int WW, float* out,
                                    int out_layout, hipStream_t s);

// ---- image preprocessing (preprocess.hip; run_model.lua:67-74) -------------------------------------------------------
void preprocess_scaled_size(int H0, int W0, int image_size, int* oh, int* ow);
size_t preprocess_scratch_bytes(int H0, int W0, int oh, int ow);     // the width pass's double plane
size_t preprocess_taps_bytes(int oh, int ow);                        // the tap tables of a (H0, W0) -> (oh, ow) scaling ...
void preprocess_make_taps(int H0, int W0, int oh, int ow, void* host_out);   // ... built on the host (they depend on the sizes only)
hipError_t launch_preprocess_u8(const uint8_t* src_dev, int H0, int W0, int oh, int ow, const float mean_bgr[3],
                                void* scratch, const void* taps_dev, float* out_chw, uint8_t* rgb_hwc, hipStream_t s);
