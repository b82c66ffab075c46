// Image preprocessing on the device (run_model.lua:67-74): image.load's byte -> [0,1] conversion, image.scale(img, size)
// (torch/image generic/image.c: scaleBilinear = scaleLinear_rowcol along the width, then along the height -- linear
// interpolation when a side grows, AREA AVERAGING when it shrinks, a copy when it stays), `:float()`, RGB -> BGR, x 255,
// minus the VGG mean.
//
// Arithmetic (round 6, advisor finding): run_model.lua never calls torch.setdefaulttensortype, so image.load hands
// image.scale a DoubleTensor -- pixels byte/255 in double, a double intermediate plane -- while the C loops keep their
// `float scale`, `float acc`, `float n` locals; `:float()` comes after the scaling.  This file follows that chain: products
// and sums in double where the library's operands are double, every assignment to `acc` rounded to float, `acc / n` a float
// division, the interpolation `(1 - f) * a + f * b` left in double; one rounding to float at the end.  (Rounds 4-5 ran an
// all-fp32 chain, within an ulp of this one.)  Bit-equal to the oracle's scalar restatement oracle.preprocess and to the
// host restatement densecap_amd/run_model.py::image_scale (tests/test_gpu_preprocess.py): the per-sample source ranges and
// float weights are computed once on the host exactly as the library computes them inside its loops, and travel as two
// small tables.
//
// Round-4 verdict, item 4: the host restatement (a Python loop per output row / column) took 94-190 ms per photograph and
// starved a device that needs 5.5 ms per image.  Here a 1600x1200 photograph costs its 5.8 MB upload and two launches.
#include <algorithm>
#include <vector>

#include "common.h"

// every operation rounds once, in source order (bit-equality with the restatements depends on it)
#pragma clang fp contract(off)

namespace {

// One output sample of scaleLinear_rowcol along an axis of length src_len -> dst_len:
//   grow (dst > src):   out = (1 - f0) * src[i0] + f0 * src[i0 + 1]          (i1 = -1; the last sample copies src[src_len-1]: f0 = 0, i0 = src_len-1 ... see make_taps)
//   shrink (dst < src): acc = (1 - f0) * src[i0]; acc += src[i] for i0 < i < i1; acc += f1 * src[i1] if i1 < src_len; out = acc / n
//   copy:               out = src[i0]
struct Tap { int i0, i1; float f0, f1, n; int kind; };      // kind: 0 copy, 1 grow, 2 shrink

std::vector<Tap> make_taps(int src_len, int dst_len) {
  std::vector<Tap> t((size_t)dst_len);
  if (dst_len == src_len) {
    for (int d = 0; d < dst_len; ++d) t[d] = Tap{d, -1, 0.f, 0.f, 1.f, 0};
  } else if (dst_len > src_len) {
    if (src_len == 1) {
      for (int d = 0; d < dst_len; ++d) t[d] = Tap{0, -1, 0.f, 0.f, 1.f, 0};
    } else {
      const float scale = (float)(src_len - 1) / (float)(dst_len - 1);
      for (int d = 0; d < dst_len - 1; ++d) {
        float si_f = (float)d * scale;
        const int si_i = (int)si_f;
        si_f = si_f - (float)si_i;
        t[d] = Tap{si_i, -1, si_f, 0.f, 1.f, 1};
      }
      t[dst_len - 1] = Tap{src_len - 1, -1, 0.f, 0.f, 1.f, 0};
    }
  } else {
    const float scale = (float)src_len / (float)dst_len;
    int si0_i = 0;
    float si0_f = 0.f;
    for (int d = 0; d < dst_len; ++d) {
      float si1_f = (float)(d + 1) * scale;
      const int si1_i = (int)si1_f;
      si1_f = si1_f - (float)si1_i;
      float n = 1.f - si0_f;
      for (int si = si0_i + 1; si < si1_i; ++si) n = n + 1.f;
      if (si1_i < src_len) n = n + si1_f;
      t[d] = Tap{si0_i, si1_i, si0_f, si1_f, n, 2};
      si0_i = si1_i;
      si0_f = si1_f;
    }
  }
  return t;
}

// One output sample of scaleLinear_rowcol over double samples px(i) (see make_taps for the cases)
template <typename PX>
__device__ __forceinline__ double scale_sample(const Tap& tp, int src_len, PX px) {
  if (tp.kind == 0) return px(tp.i0);
  if (tp.kind == 1)      // (1 - si_f) * src[a] + si_f * src[b]: float weights, double samples, the sum stays double
    return __dadd_rn(__dmul_rn((double)__fsub_rn(1.f, tp.f0), px(tp.i0)), __dmul_rn((double)tp.f0, px(tp.i0 + 1)));
  // float acc: every assignment rounds the double expression to float
  float acc = (float)__dmul_rn((double)__fsub_rn(1.f, tp.f0), px(tp.i0));
  for (int i = tp.i0 + 1; i < tp.i1; ++i) acc = (float)__dadd_rn((double)acc, px(i));
  if (tp.i1 < src_len) acc = (float)__dadd_rn((double)acc, __dmul_rn((double)tp.f1, px(tp.i1)));
  return (double)__fdiv_rn(acc, tp.n);
}

// pass 1: along the width.  src: (H0, W0, 3) uint8 -> double byte/255 (image.load into a DoubleTensor); tmp: (3, H0, ow) double
__global__ void scale_width_u8_kernel(const uint8_t* __restrict__ src, int H0, int W0, const Tap* __restrict__ taps, int ow,
                                      double* __restrict__ tmp) {
  const long total = (long)3 * H0 * ow;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % ow);
    const long r = t / ow;
    const int y = (int)(r % H0), c = (int)(r / H0);
    const uint8_t* row = src + ((size_t)y * W0) * 3 + c;
    tmp[t] = scale_sample(taps[ox], W0, [&](int x) { return __ddiv_rn((double)row[(size_t)x * 3], 255.0); });
  }
}

// pass 2: along the height, `:float()`, then run_model.lua:70-74: out[k] = scaled[2 - k] * 255 - mean_bgr[k] in float;
// optionally the scaled RGB image as bytes for the visualiser (image.save: clamp to [0,1], x 255, truncate).
__global__ void scale_height_finish_kernel(const double* __restrict__ tmp, int H0, int ow, const Tap* __restrict__ taps, int oh,
                                           float m0, float m1, float m2, float* __restrict__ out_chw,
                                           uint8_t* __restrict__ rgb_hwc) {
  const long total = (long)3 * oh * ow;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % ow);
    const long r = t / ow;
    const int oy = (int)(r % oh), c = (int)(r / oh);          // c: RGB channel of the scaled image
    const double* col = tmp + (size_t)c * H0 * ow + ox;
    const float v = (float)scale_sample(taps[oy], H0, [&](int y) { return col[(size_t)y * ow]; });      // :float()
    const int k = 2 - c;                                      // BGR plane
    const float mean = k == 0 ? m0 : (k == 1 ? m1 : m2);
    out_chw[((size_t)k * oh + oy) * ow + ox] = __fsub_rn(__fmul_rn(v, 255.f), mean);
    if (rgb_hwc != nullptr) {
      const float cl = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      rgb_hwc[((size_t)oy * ow + ox) * 3 + c] = (uint8_t)__fmul_rn(cl, 255.f);
    }
  }
}

}  // namespace

// image.scale(img, size) with a number (run_model.lua:68): the longer side becomes `size`, the other keeps the aspect ratio,
// truncated like a Lua number handed to Tensor:resize (floating-point division in double, as Lua numbers are)
void preprocess_scaled_size(int H0, int W0, int image_size, int* oh, int* ow) {
  const int imax = H0 > W0 ? H0 : W0;
  *oh = (int)((double)H0 * (double)image_size / (double)imax);
  *ow = (int)((double)W0 * (double)image_size / (double)imax);
}

// scratch of one call: the width pass's (3, H0, ow) double plane
size_t preprocess_scratch_bytes(int H0, int W0, int oh, int ow) {
  (void)W0; (void)oh;
  return (size_t)3 * H0 * ow * sizeof(double);
}

// The two tap tables of a (H0, W0) -> (oh, ow) scaling: ow width taps, then oh height taps.  They depend on the four sizes
// only: dc_preprocess_u8 keeps the device copy for as long as the sizes repeat (every frame of the webcam daemon).
size_t preprocess_taps_bytes(int oh, int ow) { return (size_t)(oh + ow) * sizeof(Tap); }
void preprocess_make_taps(int H0, int W0, int oh, int ow, void* host_out) {
  const std::vector<Tap> tw = make_taps(W0, ow), th = make_taps(H0, oh);
  Tap* o = static_cast<Tap*>(host_out);
  std::copy(tw.begin(), tw.end(), o);
  std::copy(th.begin(), th.end(), o + ow);
}

// src_dev: (H0, W0, 3) uint8 on the device; scratch: preprocess_scratch_bytes on the device; taps_dev: the tables of these sizes
hipError_t launch_preprocess_u8(const uint8_t* src_dev, int H0, int W0, int oh, int ow, const float mean_bgr[3],
                                void* scratch, const void* taps_dev, float* out_chw, uint8_t* rgb_hwc, hipStream_t s) {
  if (H0 <= 0 || W0 <= 0 || oh <= 0 || ow <= 0 || taps_dev == nullptr) return hipErrorInvalidValue;
  double* tmp = static_cast<double*>(scratch);
  const Tap* taps = static_cast<const Tap*>(taps_dev);
  const long n1 = (long)3 * H0 * ow, n2 = (long)3 * oh * ow;
  hipLaunchKernelGGL(scale_width_u8_kernel, dim3((unsigned)std::min<long>((n1 + 255) / 256, 65535)), dim3(256), 0, s, src_dev, H0,
                     W0, taps, ow, tmp);
  hipLaunchKernelGGL(scale_height_finish_kernel, dim3((unsigned)std::min<long>((n2 + 255) / 256, 65535)), dim3(256), 0, s, tmp, H0,
                     ow, taps + ow, oh, mean_bgr[0], mean_bgr[1], mean_bgr[2], out_chw, rgb_hwc);
  return hipGetLastError();
}
