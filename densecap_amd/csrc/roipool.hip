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
#include "common.h"

// every fp32 op rounds once, in source order (integer decisions depend on it)
#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void bilinear_roi_pool_kernel(const float* __restrict__ feat, int h, int w, int C,
                                                                const float* __restrict__ boxes, int B,
                                                                const int32_t* __restrict__ B_dev, float img_h,
                                                                float img_w, int HH, int WW, float* __restrict__ out,
                                                                int out_layout) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * HH * WW * C4;
  const int b_live = B_dev ? min(*B_dev, B) : B;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c4 = (int)(idx % C4);
    size_t t = idx / C4;
    const int j = (int)(t % WW); t /= WW;
    const int i = (int)(t % HH);
    const int b = (int)(t / HH);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (b < b_live) {
      const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + (size_t)b * 4);
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
      const int x0 = (int)xfl, y0 = (int)yfl;
      const float wx = __fsub_rn(1.f, __fsub_rn(xcoord, xfl));
      const float wy = __fsub_rn(1.f, __fsub_rn(ycoord, yfl));
      const bool xin0 = x0 >= 0 && x0 <= w - 1, xin1 = x0 + 1 >= 0 && x0 + 1 <= w - 1;
      const bool yin0 = y0 >= 0 && y0 <= h - 1, yin1 = y0 + 1 >= 0 && y0 + 1 <= h - 1;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const float* fp = feat + (size_t)c4 * 4;
      const f32x4 tl = (xin0 && yin0) ? *reinterpret_cast<const f32x4*>(fp + ((size_t)y0 * w + x0) * C) : z;
      const f32x4 tr = (xin1 && yin0) ? *reinterpret_cast<const f32x4*>(fp + ((size_t)y0 * w + x0 + 1) * C) : z;
      const f32x4 bl = (xin0 && yin1) ? *reinterpret_cast<const f32x4*>(fp + ((size_t)(y0 + 1) * w + x0) * C) : z;
      const f32x4 br = (xin1 && yin1) ? *reinterpret_cast<const f32x4*>(fp + ((size_t)(y0 + 1) * w + x0 + 1) * C) : z;
      const float w00 = __fmul_rn(wx, wy), w01 = __fmul_rn(__fsub_rn(1.f, wx), wy);
      const float w10 = __fmul_rn(wx, __fsub_rn(1.f, wy)), w11 = __fmul_rn(__fsub_rn(1.f, wx), __fsub_rn(1.f, wy));
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w00, tl[e]), __fmul_rn(w01, tr[e])), __fmul_rn(w10, bl[e])),
                         __fmul_rn(w11, br[e]));
    }
    if (out_layout == 1) {
      *reinterpret_cast<f32x4*>(out + idx * 4) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(((size_t)b * C + c4 * 4 + e) * HH + i) * WW + j] = v[e];
    }
  }
}

}  // namespace

hipError_t launch_bilinear_roi_pool(const float* feat_hwc, int h, int w, int C, const float* boxes, int B,
                                    const int32_t* B_dev, int img_h, int img_w, int HH, int WW, float* out,
                                    int out_layout, hipStream_t s) {
  if (C % 4 || B <= 0) return hipErrorInvalidValue;
  const size_t total = (size_t)B * HH * WW * (C / 4);
  size_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(bilinear_roi_pool_kernel, dim3((unsigned)grid), dim3(256), 0, s, feat_hwc, h, w, C, boxes, B,
                     B_dev, (float)img_h, (float)img_w, HH, WW, out, out_layout);
  return hipGetLastError();
}
