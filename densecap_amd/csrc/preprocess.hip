// Image preprocessing on the device (run_model.lua:67-74): image.load's byte -> float conversion, image.scale(img, size)
// (torch/image generic/image.c: scaleBilinear = scaleLinear_rowcol along the width, then along the height -- linear
// interpolation when a side grows, AREA AVERAGING when it shrinks, a copy when it stays), RGB -> BGR, x 255, minus the
// VGG mean.  Bit-equal to the host restatement densecap_amd/run_model.py::image_scale (tests/test_gpu_preprocess.py): every
// output sample is the same chain of single fp32 operations -- the per-sample source ranges and weights are computed once
// on the host, in fp32, exactly as the library computes them inside its loops, and travel as two small tables.
//
// Round-4 verdict, item 4: the host restatement (a Python loop per output row / column) took 94-190 ms per photograph and
// starved a device that needs 5.5 ms per image.  Here a 1600x1200 photograph costs its 5.8 MB upload and two launches.
#include <vector>

#include "common.h"

// every fp32 op rounds once, in source order (bit-equality with the host restatement depends on it)
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

// pass 1: along the width.  src: (H0, W0, 3) uint8 -> float byte/255 (image.load); tmp: (3, H0, ow) fp32
__global__ void scale_width_u8_kernel(const uint8_t* __restrict__ src, int H0, int W0, const Tap* __restrict__ taps, int ow,
                                      float* __restrict__ tmp) {
  const long total = (long)3 * H0 * ow;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % ow);
    const long r = t / ow;
    const int y = (int)(r % H0), c = (int)(r / H0);
    const uint8_t* row = src + ((size_t)y * W0) * 3 + c;
    auto px = [&](int x) { return __fdiv_rn((float)row[(size_t)x * 3], 255.f); };
    const Tap tp = taps[ox];
    float v;
    if (tp.kind == 0) v = px(tp.i0);
    else if (tp.kind == 1) v = __fadd_rn(__fmul_rn(__fsub_rn(1.f, tp.f0), px(tp.i0)), __fmul_rn(tp.f0, px(tp.i0 + 1)));
    else {
      float acc = __fmul_rn(__fsub_rn(1.f, tp.f0), px(tp.i0));
      for (int x = tp.i0 + 1; x < tp.i1; ++x) acc = __fadd_rn(acc, px(x));
      if (tp.i1 < W0) acc = __fadd_rn(acc, __fmul_rn(tp.f1, px(tp.i1)));
      v = __fdiv_rn(acc, tp.n);
    }
    tmp[t] = v;
  }
}

// pass 2: along the height, then run_model.lua:70-74: out[k] = scaled[2 - k] * 255 - mean_bgr[k]; optionally the scaled RGB
// image as bytes for the visualiser (image.save: clamp to [0,1], x 255, truncate).
__global__ void scale_height_finish_kernel(const float* __restrict__ tmp, int H0, int ow, const Tap* __restrict__ taps, int oh,
                                           float m0, float m1, float m2, float* __restrict__ out_chw,
                                           uint8_t* __restrict__ rgb_hwc) {
  const long total = (long)3 * oh * ow;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % ow);
    const long r = t / ow;
    const int oy = (int)(r % oh), c = (int)(r / oh);          // c: RGB channel of the scaled image
    const float* col = tmp + (size_t)c * H0 * ow + ox;
    auto px = [&](int y) { return col[(size_t)y * ow]; };
    const Tap tp = taps[oy];
    float v;
    if (tp.kind == 0) v = px(tp.i0);
    else if (tp.kind == 1) v = __fadd_rn(__fmul_rn(__fsub_rn(1.f, tp.f0), px(tp.i0)), __fmul_rn(tp.f0, px(tp.i0 + 1)));
    else {
      float acc = __fmul_rn(__fsub_rn(1.f, tp.f0), px(tp.i0));
      for (int y = tp.i0 + 1; y < tp.i1; ++y) acc = __fadd_rn(acc, px(y));
      if (tp.i1 < H0) acc = __fadd_rn(acc, __fmul_rn(tp.f1, px(tp.i1)));
      v = __fdiv_rn(acc, tp.n);
    }
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

// src_dev: (H0, W0, 3) uint8 on the device; scratch: >= 3*H0*ow floats + (oh + ow) Taps on the device (see preprocess_scratch_bytes)
size_t preprocess_scratch_bytes(int H0, int W0, int oh, int ow) {
  return ((size_t)3 * H0 * ow * sizeof(float) + 255) / 256 * 256 + (size_t)(oh + ow) * sizeof(Tap);
}

hipError_t launch_preprocess_u8(const uint8_t* src_dev, int H0, int W0, int oh, int ow, const float mean_bgr[3],
                                void* scratch, float* out_chw, uint8_t* rgb_hwc, hipStream_t s) {
  if (H0 <= 0 || W0 <= 0 || oh <= 0 || ow <= 0) return hipErrorInvalidValue;
  float* tmp = static_cast<float*>(scratch);
  Tap* taps_dev = reinterpret_cast<Tap*>(static_cast<char*>(scratch) + ((size_t)3 * H0 * ow * sizeof(float) + 255) / 256 * 256);
  std::vector<Tap> tw = make_taps(W0, ow), th = make_taps(H0, oh);
  // (pageable host vectors: the two small copies complete before the call returns to the vectors' destruction)
  hipError_t e = hipMemcpyAsync(taps_dev, tw.data(), tw.size() * sizeof(Tap), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(taps_dev + ow, th.data(), th.size() * sizeof(Tap), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  const long n1 = (long)3 * H0 * ow, n2 = (long)3 * oh * ow;
  hipLaunchKernelGGL(scale_width_u8_kernel, dim3((unsigned)std::min<long>((n1 + 255) / 256, 65535)), dim3(256), 0, s, src_dev, H0,
                     W0, taps_dev, ow, tmp);
  hipLaunchKernelGGL(scale_height_finish_kernel, dim3((unsigned)std::min<long>((n2 + 255) / 256, 65535)), dim3(256), 0, s, tmp, H0,
                     ow, taps_dev + ow, oh, mean_bgr[0], mean_bgr[1], mean_bgr[2], out_chw, rgb_hwc);
  return hipGetLastError();
}
