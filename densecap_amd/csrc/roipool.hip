// Bilinear RoI pooling (gfx950), HBM-bound.
//
// Fuses nn.BoxToAffine (BoxToAffine.lua:69-93) -> stnbhwd AffineGridGeneratorBHWD(HH,WW) ->
// nn.BatchBilinearSamplerBHWD (BatchBilinearSamplerBHWD.lua:104-122) -> nn.Transpose of
// nn.BilinearRoiPooling (BilinearRoiPooling.lua:42-60).  The sampling grid is never
// materialised.  The feature map is channels-last (h,w,C): each of the 4 taps is a contiguous
// C*4-byte run, read as float4 per lane (fully coalesced, L2-resident: 3.5 MB at 38x45x512);
// the output is written once, float4 per lane, in (B,HH,WW,C) order, which is the K-order the
// repacked fc6 weight expects -- so the reference's two Transpose copies vanish.
// Algorithmic bytes = 4*C*(h*w + B*HH*WW) + 16*B.
#include <stdlib.h>

#include "common.h"

// every fp32 op rounds once, in source order (integer decisions depend on it)
#pragma clang fp contract(off)

namespace {

// One workgroup per box: phase 1 computes the HH*WW sampling positions once (BoxToAffine + grid + floor +
// weights: ~6 IEEE divisions each) into LDS; phase 2 streams (point, 4-channel) items: four 16-byte taps
// from the L2-resident map, blend, one 16-byte store.  A point's C channels are 2 KiB contiguous in the
// output, so every store instruction of a wave writes 1 KiB linearly.
constexpr int ROI_MAX_PTS = 256;  // phase 1 uses one thread per point
// Batched over the images of a group (round 5): virtual box v / split runs over nimg * B boxes, image = box / B; every image
// has its own feature map (feat + image * feat_stride), live count (B_dev[image * bdev_stride]) and B rows of boxes / output.
// `pick` (optional): the boxes are gathered on the fly -- box b of an image is src_boxes[image * src_stride + pick[b]]
// (box_utils.nms picks into the RPN boxes, LocalizationLayer.lua:338-343) and is also written to `boxes` (zeros for the rows
// past the live count), which replaces the separate gather launch.
__global__ __launch_bounds__(256) void bilinear_roi_pool_kernel(const float* __restrict__ feat_base, int h, int w, int C,
                                                                float* __restrict__ boxes, int B, int nimg,
                                                                size_t feat_stride, const int32_t* __restrict__ B_dev,
                                                                int bdev_stride, const int32_t* __restrict__ pick,
                                                                const float* __restrict__ src_boxes, size_t src_stride,
                                                                float img_h, float img_w, int HH, int WW,
                                                                float* __restrict__ out, int out_layout, int split) {
  __shared__ int s_off[ROI_MAX_PTS][4];     // element offset of each tap (or -1 when outside the map)
  __shared__ float s_w[ROI_MAX_PTS][4];     // w00, w01, w10, w11
  const int C4 = C >> 2;
  const int npts = HH * WW;
  // `split` workgroups share a box (each takes a contiguous slice of its (point, channel-chunk) items): with few boxes
  // the kernel is a latency chain of ~25 dependent load rounds per thread, not a bandwidth problem
  // XCD-aware order for a group: blocks x, x + 8, ... share an XCD and its 4 MiB L2.  With 2, 4 or 8 images every image gets
  // its own XCDs, so that an L2 holds ONE feature map (3.5 MB at 720x600) instead of streaming all of them (measured with
  // 4 x 300 boxes: 2.1 TB/s on the interleaved order against 2.9 for a single image's 1000 boxes).
  const int per_img = B * split;                           // work items of one image
  int v0 = blockIdx.x, vstride = gridDim.x, v_end = nimg * per_img, img_fixed = -1;
  if (nimg > 1 && (8 % nimg) == 0 && (gridDim.x & 7) == 0) {
    const int xpi = 8 / nimg, xcd = blockIdx.x & 7;        // XCDs per image
    img_fixed = xcd / xpi;
    v0 = (xcd % xpi) + xpi * (blockIdx.x >> 3);            // index among the blocks that serve this image
    vstride = xpi * (gridDim.x >> 3);
    v_end = per_img;
  }
  for (int vv = v0; vv < v_end; vv += vstride) {
    const int v = img_fixed >= 0 ? img_fixed * per_img + vv : vv;
    const int b = v / split, part = v - b * split;         // b: row over all images of the group
    const int img = b / B, bi = b - img * B;
    const int b_live = B_dev ? min(B_dev[(size_t)img * bdev_stride], B) : B;
    const bool live = bi < b_live;
    const float* feat = feat_base + (size_t)img * feat_stride;
    __shared__ f32x4 s_box;
    if (pick != nullptr) {                                  // (the previous box's readers of s_box are behind that box's second barrier)
      if (threadIdx.x == 0) {
        f32x4 bx = {0.f, 0.f, 0.f, 0.f};
        if (live) bx = *reinterpret_cast<const f32x4*>(src_boxes + (size_t)img * src_stride + (size_t)pick[b] * 4);
        s_box = bx;
        if (part == 0) *reinterpret_cast<f32x4*>(boxes + (size_t)b * 4) = bx;
      }
    }
    // boxes past the RPN NMS count are not sampled but their rows ARE zero-filled: fc6 / fc7, the heads and the decode
    // run over all B rows in the reference caption order (only the NMS and the gathers are bounded by the device-side
    // count), so a dead row must hold defined values -- zeros, as the gathers before it write (advisor finding, round 3)
    __syncthreads();
    if (live && threadIdx.x < npts) {
      const int i = threadIdx.x / WW, j = threadIdx.x - i * WW;
      const f32x4 bx = pick != nullptr ? s_box : *reinterpret_cast<const f32x4*>(boxes + (size_t)b * 4);
      // BoxToAffine.lua:88-91
      const float th23 = __fdiv_rn(__fadd_rn(__fmul_rn(bx[0], 2.f), -1.f - img_w), img_w - 1.f);
      const float th13 = __fdiv_rn(__fadd_rn(__fmul_rn(bx[1], 2.f), -1.f - img_h), img_h - 1.f);
      const float th22 = __fdiv_rn(bx[2], img_w);
      const float th11 = __fdiv_rn(bx[3], img_h);
      // AffineGridGeneratorBHWD base grid: -1 + 2*i/(HH-1), computed in double then rounded
      const float yb = (float)(-1.0 + ((double)i / (double)(HH - 1)) * 2.0);
      const float xb = (float)(-1.0 + ((double)j / (double)(WW - 1)) * 2.0);
      const float gy = __fadd_rn(__fadd_rn(__fmul_rn(yb, th11), __fmul_rn(xb, 0.f)), th13);
      const float gx = __fadd_rn(__fadd_rn(__fmul_rn(yb, 0.f), __fmul_rn(xb, th22)), th23);
      // BilinearSamplerBHWD_updateOutput
      const float xcoord = __fdiv_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)(w - 1)), 2.f);
      const float ycoord = __fdiv_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)(h - 1)), 2.f);
      const float xfl = floorf(xcoord), yfl = floorf(ycoord);
      // clamp before the int cast (far-away boxes): anything outside [-1, dim] is invalid either way
      const int x0 = (int)fminf(fmaxf(xfl, -2.f), (float)w + 1.f), y0 = (int)fminf(fmaxf(yfl, -2.f), (float)h + 1.f);
      const float wx = __fsub_rn(1.f, __fsub_rn(xcoord, xfl));
      const float wy = __fsub_rn(1.f, __fsub_rn(ycoord, yfl));
      const bool xin0 = x0 >= 0 && x0 <= w - 1, xin1 = x0 + 1 >= 0 && x0 + 1 <= w - 1;
      const bool yin0 = y0 >= 0 && y0 <= h - 1, yin1 = y0 + 1 >= 0 && y0 + 1 <= h - 1;
      const int p = threadIdx.x;
      s_off[p][0] = (xin0 && yin0) ? (y0 * w + x0) * C : -1;
      s_off[p][1] = (xin1 && yin0) ? (y0 * w + x0 + 1) * C : -1;
      s_off[p][2] = (xin0 && yin1) ? ((y0 + 1) * w + x0) * C : -1;
      s_off[p][3] = (xin1 && yin1) ? ((y0 + 1) * w + x0 + 1) * C : -1;
      s_w[p][0] = __fmul_rn(wx, wy);
      s_w[p][1] = __fmul_rn(__fsub_rn(1.f, wx), wy);
      s_w[p][2] = __fmul_rn(wx, __fsub_rn(1.f, wy));
      s_w[p][3] = __fmul_rn(__fsub_rn(1.f, wx), __fsub_rn(1.f, wy));
    }
    __syncthreads();
    const int items = npts * C4;
    const int it_lo = (int)((long)items * part / split), it_hi = (int)((long)items * (part + 1) / split);
#pragma unroll 4
    for (int it = it_lo + threadIdx.x; it < it_hi; it += 256) {
      const int p = it / C4, c4 = it - p * C4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const float* fp = feat + (size_t)c4 * 4;
        const int o0 = s_off[p][0], o1 = s_off[p][1], o2 = s_off[p][2], o3 = s_off[p][3];
        const f32x4 tl = o0 >= 0 ? *reinterpret_cast<const f32x4*>(fp + o0) : z;
        const f32x4 tr = o1 >= 0 ? *reinterpret_cast<const f32x4*>(fp + o1) : z;
        const f32x4 bl = o2 >= 0 ? *reinterpret_cast<const f32x4*>(fp + o2) : z;
        const f32x4 br = o3 >= 0 ? *reinterpret_cast<const f32x4*>(fp + o3) : z;
        const float w00 = s_w[p][0], w01 = s_w[p][1], w10 = s_w[p][2], w11 = s_w[p][3];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w00, tl[e]), __fmul_rn(w01, tr[e])), __fmul_rn(w10, bl[e])),
                           __fmul_rn(w11, br[e]));
      }
      if (out_layout == 1) {
        // written once, consumed by fc6 much later: keep it out of the L2 that holds the feature map
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + ((size_t)b * npts + p) * C + (size_t)c4 * 4));
      } else {
        const int i = p / WW, j = p - i * WW;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(((size_t)b * C + c4 * 4 + e) * HH + i) * WW + j] = v[e];
      }
    }
  }
}

}  // namespace

hipError_t launch_bilinear_roi_pool_group(const float* feat_hwc, size_t feat_stride, int nimg, int h, int w, int C,
                                          float* boxes, int B, const int32_t* B_dev, int bdev_stride,
                                          const int32_t* pick, const float* src_boxes, size_t src_stride, int img_h,
                                          int img_w, int HH, int WW, float* out, int out_layout, hipStream_t s) {
  if (C % 4 || B <= 0 || nimg <= 0 || HH * WW > ROI_MAX_PTS || (pick != nullptr && src_boxes == nullptr)) return hipErrorInvalidValue;
  const long Bt = (long)B * nimg;
  long split = (4096 + Bt / 2) / Bt;          // ~4096 workgroups (16 per CU) measured best for 300 .. 2000 boxes
  split = split < 1 ? 1 : (split > 8 ? 8 : split);
  const long want = (Bt * split + 7) / 8 * 8;     // a multiple of 8: the XCD-aware order of a group
  const int grid = want < 256 * 16 ? (int)want : 256 * 16;
  hipLaunchKernelGGL(bilinear_roi_pool_kernel, dim3((unsigned)grid), dim3(256), 0, s, feat_hwc, h, w, C, boxes, B, nimg,
                     feat_stride, B_dev, bdev_stride, pick, src_boxes, src_stride, (float)img_h, (float)img_w, HH, WW, out,
                     out_layout, (int)split);
  return hipGetLastError();
}

hipError_t launch_bilinear_roi_pool(const float* feat_hwc, int h, int w, int C, const float* boxes, int B,
                                    const int32_t* B_dev, int img_h, int img_w, int HH, int WW, float* out,
                                    int out_layout, hipStream_t s) {
  return launch_bilinear_roi_pool_group(feat_hwc, 0, 1, h, w, C, const_cast<float*>(boxes), B, B_dev, 0, nullptr, nullptr, 0,
                                        img_h, img_w, HH, WW, out, out_layout, s);
}
