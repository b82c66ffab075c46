// Box pipeline of the LocalizationLayer test path (gfx950): anchors, box transform, clipping,
// fused RPN decode, greedy NMS (rank sort + IoU bit-mask + chunked wavefront scan), gathers.
//
// Bit-exactness: every fp32 expression that feeds an integer decision (valid flags, NMS picks)
// is written with __f*_rn intrinsics in the reference's operation order (no FMA contraction),
// so that, fed the oracle's inputs, the picks are identical to box_utils.nms
// (densecap/box_utils.lua:154-256).
#include "common.h"

// every fp32 op rounds once, in source order (integer decisions depend on it)
#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

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
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void make_anchors_kernel(float* __restrict__ out, int h, int w, float x0, float y0, float sx, float sy,
                                    const float* __restrict__ anchors, int k) {
  const int total = k * h * w;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total) return;
  const int a = b / (h * w), rem = b - a * h * w, y = rem / w, x = rem - y * w;
  f32x4 v;
  v[0] = __fadd_rn(__fmul_rn((float)x, sx), x0);   // MakeAnchors.lua:44-47
  v[1] = __fadd_rn(__fmul_rn((float)y, sy), y0);
  v[2] = anchors[a];
  v[3] = anchors[k + a];
  *reinterpret_cast<f32x4*>(out + (size_t)b * 4) = v;
}

__global__ void apply_box_transform_kernel(const float* __restrict__ boxes, const float* __restrict__ trans,
                                           float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)i * 4);
  const f32x4 t = *reinterpret_cast<const f32x4*>(trans + (size_t)i * 4);
  f32x4 o;
  o[0] = __fadd_rn(__fmul_rn(t[0], b[2]), b[0]);   // ApplyBoxTransform.lua:85-88
  o[1] = __fadd_rn(__fmul_rn(t[1], b[3]), b[1]);
  o[2] = __fmul_rn(expf(t[2]), b[2]);
  o[3] = __fmul_rn(expf(t[3]), b[3]);
  *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = o;
}

__device__ __forceinline__ bool clip_one(const f32x4& b, float x_min, float y_min, float x_max, float y_max,
                                         f32x4& o) {
  float x1, y1, x2, y2;
  corners(b[0], b[1], b[2], b[3], x1, y1, x2, y2);
  x1 = clampf(x1, x_min, x_max - 1.f);               // box_utils.lua:505-508
  y1 = clampf(y1, y_min, y_max - 1.f);
  x2 = clampf(x2, x_min + 1.f, x_max);
  y2 = clampf(y2, y_min + 1.f, y_max);
  o[0] = __fdiv_rn(__fadd_rn(x1, x2), 2.f);          // x1y1x2y2_to_xcycwh (box_utils.lua:400-403)
  o[1] = __fdiv_rn(__fadd_rn(y1, y2), 2.f);
  o[2] = __fsub_rn(x2, x1);
  o[3] = __fsub_rn(y2, y1);
  return (x2 > x1) && (y2 > y1);
}

__global__ void clip_boxes_kernel(const float* __restrict__ boxes, float* __restrict__ clipped,
                                  uint8_t* __restrict__ valid, int n, float x_min, float y_min, float x_max,
                                  float y_max) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)i * 4);
  f32x4 o;
  const bool v = clip_one(b, x_min, y_min, x_max, y_max, o);
  *reinterpret_cast<f32x4*>(clipped + (size_t)i * 4) = o;
  if (valid) valid[i] = v ? 1 : 0;
}

__global__ void xcycwh_to_x1y1x2y2_kernel(const float* __restrict__ boxes, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)i * 4);
  float o0, o1, o2, o3;
  corners(b[0], b[1], b[2], b[3], o0, o1, o2, o3);
  *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = f32x4{o0, o1, o2, o3};
}

// nn.BoxIoU (BoxIoU.lua:40-73); convention 1 = NMS inline (+1) form on xcycwh inputs.
__global__ void box_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out,
                               int B1, int B2, int convention) {
  const size_t total = (size_t)B1 * B2;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / B2), j = (int)(idx % B2);
    const f32x4 p = *reinterpret_cast<const f32x4*>(b1 + (size_t)i * 4);
    const f32x4 q = *reinterpret_cast<const f32x4*>(b2 + (size_t)j * 4);
    float px1, py1, px2, py2, qx1, qy1, qx2, qy2;
    corners(p[0], p[1], p[2], p[3], px1, py1, px2, py2);
    corners(q[0], q[1], q[2], q[3], qx1, qy1, qx2, qy2);
    const float one = convention ? 1.f : 0.f;
    const float a1 = convention ? __fmul_rn(__fadd_rn(__fsub_rn(px2, px1), 1.f), __fadd_rn(__fsub_rn(py2, py1), 1.f))
                                : __fmul_rn(p[2], p[3]);
    const float a2 = convention ? __fmul_rn(__fadd_rn(__fsub_rn(qx2, qx1), 1.f), __fadd_rn(__fsub_rn(qy2, qy1), 1.f))
                                : __fmul_rn(q[2], q[3]);
    const float x0 = fmaxf(px1, qx1), y0 = fmaxf(py1, qy1), x1 = fminf(px2, qx2), y1 = fminf(py2, qy2);
    float w = __fadd_rn(__fsub_rn(x1, x0), one), h = __fadd_rn(__fsub_rn(y1, y0), one);
    w = w > 0.f ? w : 0.f;
    h = h > 0.f ? h : 0.f;
    const float inter = __fmul_rn(w, h);
    out[idx] = __fdiv_rn(inter, __fsub_rn(__fadd_rn(a1, a2), inter));
  }
}

// Fused MakeAnchors + ReshapeBoxFeatures + ApplyBoxTransform + clip_boxes + corners + p(pos)
// (LocalizationLayer.lua:265-308).  heads: (h,w,6k) channels-last, box channels a*4+d, then
// score channels 4k + a*2 + d.  Row b = a*h*w + y*w + x (ReshapeBoxFeatures.lua:24-33).
__global__ void rpn_decode_kernel(const float* __restrict__ heads, int h, int w, int k,
                                  const float* __restrict__ anchors, float x0, float y0, float sx, float sy,
                                  float img_h, float img_w, float* __restrict__ boxes, float* __restrict__ anc_out,
                                  float* __restrict__ trans_out, float* __restrict__ xyxy, float* __restrict__ p_out,
                                  uint8_t* __restrict__ valid) {
  const int total = k * h * w;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total) return;
  const int a = b / (h * w), rem = b - a * h * w, y = rem / w, x = rem - y * w;
  const float* px = heads + (size_t)rem * (6 * k);
  const f32x4 t = *reinterpret_cast<const f32x4*>(px + a * 4);
  const float s1 = px[4 * k + a * 2], s2 = px[4 * k + a * 2 + 1];
  const float xa = __fadd_rn(__fmul_rn((float)x, sx), x0);
  const float ya = __fadd_rn(__fmul_rn((float)y, sy), y0);
  const float wa = anchors[a], ha = anchors[k + a];
  f32x4 bx;
  bx[0] = __fadd_rn(__fmul_rn(t[0], wa), xa);
  bx[1] = __fadd_rn(__fmul_rn(t[1], ha), ya);
  bx[2] = __fmul_rn(expf(t[2]), wa);
  bx[3] = __fmul_rn(expf(t[3]), ha);
  f32x4 cb;
  const bool v = clip_one(bx, 1.f, 1.f, img_w, img_h, cb);
  if (boxes) *reinterpret_cast<f32x4*>(boxes + (size_t)b * 4) = cb;
  if (anc_out) *reinterpret_cast<f32x4*>(anc_out + (size_t)b * 4) = f32x4{xa, ya, wa, ha};
  if (trans_out) *reinterpret_cast<f32x4*>(trans_out + (size_t)b * 4) = t;
  if (xyxy) {
    float c0, c1, c2, c3;
    corners(cb[0], cb[1], cb[2], cb[3], c0, c1, c2, c3);
    *reinterpret_cast<f32x4*>(xyxy + (size_t)b * 4) = f32x4{c0, c1, c2, c3};
  }
  if (p_out) {
    const float e1 = expf(s1), e2 = expf(s2);
    p_out[b] = __fmul_rn(__fdiv_rn(1.f, __fadd_rn(e1, e2)), e1);   // pow(-1):cmul (LocalizationLayer.lua:308)
  }
  if (valid) valid[b] = v ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// NMS stage 1: rank sort.  rank[i] = #boxes that come before i in (valid desc, score desc,
// index asc) order.  O(n^2) compares spread over (n/256) x SPLIT workgroups; embarrassingly
// parallel and deterministic, which a multi-pass radix sort is not needed for at n <= 64k.
// ---------------------------------------------------------------------------------------
constexpr int RANK_SPLIT = 16;
__device__ __forceinline__ float sort_score(float s, bool v) {
  if (!v) return -INFINITY;
  return (s != s) ? -INFINITY : s;
}
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ valid,
                                                       int n_cap, const int32_t* __restrict__ n_dev,
                                                       uint32_t* __restrict__ rank, int32_t* __restrict__ nvalid) {
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  __shared__ float ss[256];
  __shared__ uint8_t sv[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool vi = i < n && (valid == nullptr || valid[i] != 0);
  const float si = i < n ? sort_score(scores[i], vi) : 0.f;
  const int per = (n + RANK_SPLIT - 1) / RANK_SPLIT;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per);
  uint32_t cnt = 0;
  for (int base = j0; base < j1; base += 256) {
    const int j = base + threadIdx.x;
    __syncthreads();
    if (j < j1) {
      const bool vj = valid == nullptr || valid[j] != 0;
      ss[threadIdx.x] = sort_score(scores[j], vj);
      sv[threadIdx.x] = vj;
    }
    __syncthreads();
    const int lim = min(256, j1 - base);
    for (int q = 0; q < lim; ++q) {
      const float sj = ss[q];
      const bool vj = sv[q] != 0;
      const int jj = base + q;
      const bool before = (vj && !vi) || (vj == vi && (sj > si || (sj == si && jj < i)));
      cnt += before ? 1u : 0u;
    }
  }
  if (i < n && cnt) atomicAdd(&rank[i], cnt);
  if (blockIdx.y == 0 && vi) atomicAdd(nvalid, 1);
}

__global__ void nms_scatter_kernel(const float* __restrict__ boxes, const uint32_t* __restrict__ rank, int n_cap,
                                   const int32_t* __restrict__ n_dev, int32_t* __restrict__ order,
                                   float* __restrict__ sboxes, float* __restrict__ sarea) {
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = rank[i];
  const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)i * 4);
  order[r] = i;
  *reinterpret_cast<f32x4*>(sboxes + (size_t)r * 4) = b;
  // box_utils.lua:178-181: area = (x2-x1+1) * (y2-y1+1)
  sarea[r] = __fmul_rn(__fadd_rn(__fsub_rn(b[2], b[0]), 1.f), __fadd_rn(__fsub_rn(b[3], b[1]), 1.f));
}

// NMS stage 2: suppression bit-mask over sorted boxes.  Workgroup = 4 waves; wave q of block
// (cg, rc) computes, for the 64 sorted rows of chunk rc, the 64-bit word against column chunk
// cc = cg*4+q (only cc >= rc is needed; bit j set <=> NOT(iou(i,j) <= thresh) and j > i).
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ sboxes, const float* __restrict__ sarea,
                                                       const int32_t* __restrict__ nvalid, float thresh, int nwords_ld,
                                                       u64* __restrict__ mask) {
  const int n = *nvalid;
  const int rc = blockIdx.y, q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cc = blockIdx.x * 4 + q;
  if (rc * 64 >= n) return;
  __shared__ float cb[4][64][5];
  const int cj = cc * 64 + lane;
  if (cj < n) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(sboxes + (size_t)cj * 4);
    cb[q][lane][0] = b[0]; cb[q][lane][1] = b[1]; cb[q][lane][2] = b[2]; cb[q][lane][3] = b[3];
    cb[q][lane][4] = sarea[cj];
  }
  __syncthreads();
  if (cc < rc || cc * 64 >= n) return;
  const int row = rc * 64 + lane;
  if (row >= n) return;
  const f32x4 bi = *reinterpret_cast<const f32x4*>(sboxes + (size_t)row * 4);
  const float ai = sarea[row];
  const int jn = min(64, n - cc * 64);
  u64 word = 0;
  for (int j = (cc == rc ? lane + 1 : 0); j < jn; ++j) {
    // box_utils.lua:219-227 (j = candidate, i = picked box)
    const float xx1 = fmaxf(cb[q][j][0], bi[0]);
    const float yy1 = fmaxf(cb[q][j][1], bi[1]);
    const float xx2 = fminf(cb[q][j][2], bi[2]);
    const float yy2 = fminf(cb[q][j][3], bi[3]);
    float w = __fadd_rn(__fsub_rn(xx2, xx1), 1.f);
    float h = __fadd_rn(__fsub_rn(yy2, yy1), 1.f);
    w = w > 0.f ? w : 0.f;
    h = h > 0.f ? h : 0.f;
    const float inter = __fmul_rn(w, h);
    const float uni = __fsub_rn(__fadd_rn(cb[q][j][4], ai), inter);
    const float iou = __fdiv_rn(inter, uni);
    if (!(iou <= thresh)) word |= (1ull << j);
  }
  mask[(size_t)row * nwords_ld + cc] = word;
}

// NMS stage 3: greedy scan in sorted order, 64 candidates (one mask word) per step.
// Wave 0 resolves the intra-chunk dependencies with readlane on the diagonal words; all four
// waves then OR the picked rows into the LDS-resident `removed` bit set for later chunks.
constexpr int NMS_MAX_WORDS = 1024;  // up to 65536 boxes
__global__ __launch_bounds__(256) void nms_scan_kernel(const u64* __restrict__ mask, int nwords_ld,
                                                       const int32_t* __restrict__ nvalid,
                                                       const int32_t* __restrict__ order, int max_boxes,
                                                       int32_t* __restrict__ picks, int32_t* __restrict__ count) {
  __shared__ u64 removed[NMS_MAX_WORDS];
  __shared__ int s_cnt, s_npick;
  __shared__ int s_rows[64];
  const int n = *nvalid;
  const int nw = (n + 63) >> 6;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int v = tid; v < nw; v += 256) removed[v] = 0;
  if (tid == 0) { s_cnt = 0; s_npick = 0; }
  __syncthreads();
  for (int w = 0; w < nw; ++w) {
    if (tid < 64) {
      u64 alive = ~removed[w];
      if (w == nw - 1 && (n & 63)) alive &= (1ull << (n & 63)) - 1ull;
      const int row = w * 64 + lane;
      const u64 diag = row < n ? mask[(size_t)row * nwords_ld + w] : 0ull;
      u64 rem = alive;
      while (rem) {
        const int b = __builtin_ctzll(rem);
        const u64 d = __shfl(diag, b, 64);
        alive &= ~d;
        rem = alive & ~((2ull << b) - 1ull);
      }
      const int cnt = s_cnt;
      if (max_boxes >= 0) {
        int room = max_boxes - cnt;
        while (__builtin_popcountll(alive) > room) alive &= ~(1ull << (63 - __builtin_clzll(alive)));
      }
      if ((alive >> lane) & 1ull) {
        const int pos = __builtin_popcountll(alive & ((1ull << lane) - 1ull));
        picks[cnt + pos] = order[row];
        s_rows[pos] = row;
      }
      if (lane == 0) {
        s_npick = __builtin_popcountll(alive);
        s_cnt = cnt + __builtin_popcountll(alive);
      }
    }
    __syncthreads();
    const int npick = s_npick;
    if (max_boxes >= 0 && s_cnt >= max_boxes) break;
    if (npick > 0) {
      for (int v = w + 1 + tid; v < nw; v += 256) {
        u64 acc = 0;
        for (int p = 0; p < npick; ++p) acc |= mask[(size_t)s_rows[p] * nwords_ld + v];
        removed[v] |= acc;
      }
    }
    __syncthreads();
  }
  if (tid == 0) *count = s_cnt;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                   const int32_t* __restrict__ count, int cap, int width, float* __restrict__ out) {
  const int total = cap * width;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i = t / width, c = t - i * width;
  out[t] = i < *count ? src[(size_t)idx[i] * width + c] : 0.f;
}
__global__ void gather_rows_i32_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                       const int32_t* __restrict__ count, int cap, int width,
                                       int32_t* __restrict__ out) {
  const int total = cap * width;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i = t / width, c = t - i * width;
  out[t] = i < *count ? src[(size_t)idx[i] * width + c] : 0;
}

}  // namespace

#define LAUNCH1D(kern, n, s, ...)                                                        \
  hipLaunchKernelGGL(kern, dim3(((n) + 255) / 256 > 0 ? ((n) + 255) / 256 : 1), dim3(256), 0, s, __VA_ARGS__); \
  return hipGetLastError();

hipError_t launch_make_anchors(float* out, int h, int w, float x0, float y0, float sx, float sy,
                               const float* anchors, int k, hipStream_t s) {
  LAUNCH1D(make_anchors_kernel, k * h * w, s, out, h, w, x0, y0, sx, sy, anchors, k);
}
hipError_t launch_apply_box_transform(const float* boxes, const float* trans, float* out, int n, hipStream_t s) {
  LAUNCH1D(apply_box_transform_kernel, n, s, boxes, trans, out, n);
}
hipError_t launch_clip_boxes(const float* boxes, float* clipped, uint8_t* valid, int n, float x_min, float y_min,
                             float x_max, float y_max, hipStream_t s) {
  LAUNCH1D(clip_boxes_kernel, n, s, boxes, clipped, valid, n, x_min, y_min, x_max, y_max);
}
hipError_t launch_xcycwh_to_x1y1x2y2(const float* boxes, float* out, int n, hipStream_t s) {
  LAUNCH1D(xcycwh_to_x1y1x2y2_kernel, n, s, boxes, out, n);
}
hipError_t launch_box_iou(const float* b1, const float* b2, float* out, int B1, int B2, int convention,
                          hipStream_t s) {
  size_t total = (size_t)B1 * B2;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(box_iou_kernel, dim3(grid), dim3(256), 0, s, b1, b2, out, B1, B2, convention);
  return hipGetLastError();
}
hipError_t launch_rpn_decode(const float* heads, int h, int w, int k, const float* anchors, float x0, float y0,
                             float sx, float sy, int img_h, int img_w, float* boxes, float* anchors_out,
                             float* trans, float* x1y1x2y2, float* p, uint8_t* valid, hipStream_t s) {
  LAUNCH1D(rpn_decode_kernel, k * h * w, s, heads, h, w, k, anchors, x0, y0, sx, sy, (float)img_h, (float)img_w,
           boxes, anchors_out, trans, x1y1x2y2, p, valid);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t nms_workspace_bytes(int n) {
  const size_t nw = (n + 63) / 64;
  return align256((size_t)n * 4) * 3 + align256((size_t)n * 16) + align256(256) + align256((size_t)n * nw * 8);
}
hipError_t nms_workspace_bind(NmsWorkspace& ws, void* base, int n) {
  char* p = static_cast<char*>(base);
  ws.n_cap = n;
  ws.mask_words = (n + 63) / 64;
  ws.rank = reinterpret_cast<uint32_t*>(p); p += align256((size_t)n * 4);
  ws.order = reinterpret_cast<int32_t*>(p); p += align256((size_t)n * 4);
  ws.sarea = reinterpret_cast<float*>(p); p += align256((size_t)n * 4);
  ws.sboxes = reinterpret_cast<float*>(p); p += align256((size_t)n * 16);
  ws.nvalid = reinterpret_cast<int32_t*>(p); p += align256(256);
  ws.mask = reinterpret_cast<u64*>(p);
  return hipSuccess;
}

hipError_t launch_nms(NmsWorkspace& ws, const float* boxes, const float* scores, const uint8_t* valid, int n,
                      const int32_t* n_dev, float thresh, int max_boxes, int32_t* picks, int32_t* count,
                      hipStream_t s) {
  if (n > ws.n_cap || n > NMS_MAX_WORDS * 64) return hipErrorInvalidValue;
  hipError_t e;
  if (n <= 0) return hipMemsetAsync(count, 0, 4, s);
  if ((e = hipMemsetAsync(ws.rank, 0, (size_t)n * 4, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(ws.nvalid, 0, 4, s)) != hipSuccess) return e;
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(nms_rank_kernel, dim3(nb, RANK_SPLIT), dim3(256), 0, s, scores, valid, n, n_dev, ws.rank,
                     ws.nvalid);
  hipLaunchKernelGGL(nms_scatter_kernel, dim3(nb), dim3(256), 0, s, boxes, ws.rank, n, n_dev, ws.order, ws.sboxes,
                     ws.sarea);
  const int nchunks = (n + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3((nchunks + 3) / 4, nchunks), dim3(256), 0, s, ws.sboxes, ws.sarea,
                     ws.nvalid, thresh, (int)ws.mask_words, ws.mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), 0, s, ws.mask, (int)ws.mask_words, ws.nvalid, ws.order,
                     max_boxes, picks, count);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const float* src, const int32_t* idx, const int32_t* count, int cap, int width,
                              float* out, hipStream_t s) {
  LAUNCH1D(gather_rows_kernel, cap * width, s, src, idx, count, cap, width, out);
}
hipError_t launch_gather_rows_i32(const int32_t* src, const int32_t* idx, const int32_t* count, int cap, int width,
                                  int32_t* out, hipStream_t s) {
  LAUNCH1D(gather_rows_i32_kernel, cap * width, s, src, idx, count, cap, width, out);
}
