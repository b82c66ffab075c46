// Layout, pooling, conv1_1, LSTM point-wise, arg-max and recognition-head kernels (gfx950).
// All of these are HBM- or latency-bound byte movers: coalesced 16-byte accesses along the
// channels-last axis, wavefront (64-lane) reductions, no MFMA.
#include <algorithm>

#include "common.h"

// every fp32 op rounds once, in source order (integer decisions depend on it)
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- (C,H,W) <-> (H,W,C) ---------------------------------------------------------------
// Tiled 32x32 transpose through LDS of the (C, H*W) matrix.
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    if (r < rows && c < cols) tile[j][tx] = in[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;  // out is (cols, rows)
    if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][j];
  }
}

// OIHW (Cout,Cin,3,3) -> (Cout, 9*Cin), k = (kh*3+kw)*Cin + c
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
  const size_t total = (size_t)Cout * Cin * 9;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const size_t t = i / Cin;
    const int tap = (int)(t % 9);
    const int o = (int)(t / 9);
    out[i] = w[((size_t)o * Cin + c) * 9 + tap];
  }
}

// fc6: (N, C*HW) k = c*HW + p  ->  k' = p*C + c  (matches the (B,HH,WW,C) RoI-pool output)
__global__ void permute_fc6_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t t = i / C;
    const int p = (int)(t % HW);
    const size_t n = t / HW;
    out[i] = in[(n * C + c) * HW + p];
  }
}

// ---- conv1_1: Cin = 3 (K = 27), CHW boundary image in -> HWC out, on the matrix cores -------------------------------
// 1.49 GFLOP against 110 MB of output: HBM-store-bound (~14 us at 8 TB/s) once the arithmetic leaves the VALU
// (the scalar version was VALU-bound at 65 us).  A tile is 4 image rows x 32 columns; its (3, 6, 34) input patch is
// staged in LDS (zero padding outside the image); wave w computes row w: M = 32 pixels, N = 64 channels (two 32x32
// accumulator blocks), K = 28 = 14 x v_mfma_f32_32x32x2_f32 with k = c*9 + kh*3 + kw (k = 27: zero weight).
// A fragments are single ds_read_b32 from the patch (lane = pixel, lane half = k parity), B fragments (the weights,
// staged through LDS once per workgroup) live in 28 registers.  Each store instruction writes two pixels x 128
// contiguous bytes.  A workgroup walks `tpw` consecutive tiles of its 4-row strip: weights, bias and the patch
// addressing are set up once, the next tile's patch is in flight (registers) while this one is computed and stored, so
// the stores of a CU's three workgroups stream instead of arriving in bursts behind a load -> barrier -> compute chain
// per 32 KiB of output.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int H, int W, int relu, int tpw) {
  static_assert(COUT == 64, "two 32-channel accumulator blocks");
  constexpr int TR = 4, TC = 32, PR = TR + 2, PC = TC + 2, KP = 28;
  constexpr int NP = 3 * PR * PC, PPT = (NP + 255) / 256;
  __shared__ float patch[2][NP];
  constexpr int OP = 72;                             // transpose-tile pitch in floats: the two lane halves write pixel rows 4
                                                     // apart, 4*72 = 32 (mod 64) banks apart -> no write conflicts (68 had 2-way)
  __shared__ float wsm[4 * 32 * OP];                 // weights (64 x 27 floats) first, then the output transpose tiles
  static_assert(4 * 32 * OP >= COUT * 27, "weights fit in the transpose buffer");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hsel = lane >> 5;
  const int y0 = blockIdx.y * TR;
  const int ntx = (W + TC - 1) / TC, tx0 = blockIdx.x * tpw, tx1 = min(ntx, tx0 + tpw);
  in += (size_t)blockIdx.z * 3 * H * W;                       // image of the group
  out += (size_t)blockIdx.z * H * W * COUT;
  // this thread's (up to) three patch elements: row / channel part of the address once, the column moves with the tile
  int p_ofs[PPT], p_dx[PPT];
  bool p_row[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int i = tid + u * 256;
    const int c = i / (PR * PC), rem = i - c * (PR * PC), py = rem / PC, px = rem - py * PC;
    const int y = y0 + py - 1;
    p_row[u] = i < NP && (unsigned)y < (unsigned)H;
    p_dx[u] = px - 1;
    p_ofs[u] = p_row[u] ? (c * H + y) * W + px - 1 : 0;
  }
  float pre[PPT];
  auto fetch = [&](int tx) {
    const int x0 = tx * TC;
#pragma unroll
    for (int u = 0; u < PPT; ++u)
      pre[u] = (p_row[u] && (unsigned)(x0 + p_dx[u]) < (unsigned)W) ? in[p_ofs[u] + x0] : 0.f;
  };
  fetch(tx0);
  // weights (64 x 27, 6.9 KB): one coalesced pass into LDS (a per-lane gather from global costs 28 scattered loads per
  // thread -- measured 2x the whole kernel), then lane (n = r, half hsel) takes W[n + 32 j][k = 2 s + hsel]; the odd row
  // stride 27 keeps the LDS reads conflict-free
  for (int i = tid; i < COUT * 27; i += 256) wsm[i] = w[i];
  __syncthreads();
  float bw[2][KP / 2], bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bv[j] = bias[r + 32 * j];
#pragma unroll
    for (int st = 0; st < KP / 2; ++st) {
      const int k = 2 * st + hsel;
      bw[j][st] = k < 27 ? wsm[(r + 32 * j) * 27 + k] : 0.f;
    }
  }
  __syncthreads();                                   // the weights are in registers: wsm becomes the transpose tiles
  float* ot = wsm + wid * (32 * OP);
  const int y = y0 + wid;
  for (int tx = tx0; tx < tx1; ++tx) {
    float* pt = patch[(tx - tx0) & 1];
#pragma unroll
    for (int u = 0; u < PPT; ++u)
      if (tid + u * 256 < NP) pt[tid + u * 256] = pre[u];
    __syncthreads();     // (the other buffer is still being read by slower waves; this one was last read two tiles ago)
    if (tx + 1 < tx1) fetch(tx + 1);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const float* prow = pt + wid * PC + r;          // pixel (row wid, column r) of the tile; tap (kh, kw) at +kh*PC + kw
#pragma unroll
    for (int st = 0; st < KP / 2; ++st) {
      const int k0 = 2 * st, k1 = k0 + 1 < 27 ? k0 + 1 : 26;               // k = 27 meets a zero weight: any finite input does
      const int o0 = (k0 / 9) * (PR * PC) + ((k0 % 9) / 3) * PC + (k0 % 3);
      const int o1 = (k1 / 9) * (PR * PC) + ((k1 % 9) / 3) * PC + (k1 % 3);
      const float a = prow[hsel ? o1 : o0];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[0][st], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[1][st], acc[1], 0, 0, 0);
    }
    // Epilogue.  C/D map of the 32x32 MFMA: col (channel) = lane&31, row (pixel) = (e&3) + 8*(e>>2) + 4*hsel.  The wave's
    // result -- 32 consecutive pixels x 64 channels -- is ONE contiguous 8 KiB run of the channels-last output, so it is
    // transposed through LDS ([pixel][64 + 8 pad] floats) and leaves as 16-byte stores, 1 KiB contiguous per instruction
    // (dword stores in MFMA order reach ~2.5 TB/s on this 110 MB store-bound kernel).  The tile is the wave's own: its
    // LDS operations complete in order, no barrier between the writes, the reads and the next tile's writes.
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[j][e] + bv[j];
        if (relu) v = v > 0.f ? v : 0.f;
        ot[((e & 3) + 8 * (e >> 2) + 4 * hsel) * OP + r + 32 * j] = v;
      }
    if (y < H) {
      const int x0 = tx * TC;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane;                  // float4 index inside the wave's 32 x 64 block
        const int px = idx >> 4, c4 = idx & 15;
        const int x = x0 + px;
        if (x < W) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(ot + px * OP + c4 * 4);
          // (non-temporal stores: 27.7 instead of 29.9 us for this kernel alone, nothing end to end -- conv1_2 re-reads the tensor)
          *reinterpret_cast<f32x4*>(out + ((size_t)y * W + x) * COUT + c4 * 4) = v;
        }
      }
    }
  }
}

// ---- 2x2/2 max-pool, ceil mode, HWC -----------------------------------------------------
__global__ void maxpool2x2_ceil_kernel(const float* __restrict__ in, float* __restrict__ out, int nimg, int H,
                                       int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C / 4;
  const size_t total = (size_t)nimg * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int y0 = 2 * yo, x0 = 2 * xo;
    const float* base = in + ((size_t)n * H * W) * C + (size_t)c4 * 4;
    f32x4 m = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * C);
    const bool hx = x0 + 1 < W, hy = y0 + 1 < H;
    auto mx = [&](const f32x4& a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = a[e] > m[e] ? a[e] : m[e];
    };
    if (hx) mx(*reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0 + 1) * C));
    if (hy) mx(*reinterpret_cast<const f32x4*>(base + ((size_t)(y0 + 1) * W + x0) * C));
    if (hx && hy) mx(*reinterpret_cast<const f32x4*>(base + ((size_t)(y0 + 1) * W + x0 + 1) * C));
    *reinterpret_cast<f32x4*>(out + i * 4) = m;
  }
}

// ---- LSTM point-wise (torch-rnn nn.LSTM step; gate order i,f,o,g) ----------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return th_sigmoidf(x); }
__global__ void lstm_pointwise_kernel(const float* __restrict__ gates, float* __restrict__ c, float* __restrict__ h,
                                      int n, const int32_t* __restrict__ n_dev, int Hd, int zero_c) {
  if (n_dev) n = min(n, *n_dev);
  const size_t total = (size_t)n * Hd;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / Hd;
    const int j = (int)(i % Hd);
    const float* g = gates + m * 4 * Hd;
    const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[Hd + j]), og = sigmoidf_(g[2 * Hd + j]);
    const float gg = th_tanhf(g[3 * Hd + j]);
    const float cp = zero_c ? 0.f : c[i];
    const float cn = fg * cp + ig * gg;
    c[i] = cn;
    h[i] = og * th_tanhf(cn);
  }
}

// ---- decode step tail: arg-max finalize + LSTM point-wise, one workgroup (256 threads) per row ---------------------
// Replaces argmax_finalize + the gate row-term epilogue + lstm_pointwise of one step (3 launches and an 8 MB gate
// round trip) by one launch: the token a row just produced selects its xg row here, so the h.Wh product of the NEXT
// step's gates can run inside the same GEMM launch as the vocabulary projection (both only need h_t).
__global__ __launch_bounds__(256) void lstm_step_tail_kernel(const float* __restrict__ pval,
                                                             const int32_t* __restrict__ pidx, int ntiles, int ld,
                                                             int fixed_tok, const float* __restrict__ xg,
                                                             const float* __restrict__ gates_pre,
                                                             float* __restrict__ c, float* __restrict__ h, int n,
                                                             const int32_t* __restrict__ n_dev, int Hd, int zero_c,
                                                             int32_t* __restrict__ seq, int T, int t) {
  if (n_dev) n = min(n, *n_dev);
  const int m = blockIdx.x;
  if (m >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  __shared__ float sv[4];
  __shared__ int si[4];
  // the token-independent operands are requested first: they travel while the arg-max is being reduced
  constexpr int UPT = 2;                            // hidden units per thread and pass (Hd = 512: one pass)
  float gpre[UPT][4], cprev[UPT];
  const float* g = gates_pre ? gates_pre + (size_t)m * 4 * Hd : nullptr;
  if (g != nullptr) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int j = tid + u * 256;
      if (j < Hd) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gpre[u][q] = g[q * Hd + j];
        cprev[u] = zero_c ? 0.f : c[(size_t)m * Hd + j];
      }
    }
  }
  int tok = fixed_tok;
  if (pval != nullptr) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < ntiles; j += 256) {
      const float v = pval[(size_t)m * ld + j];
      const int i = pidx[(size_t)m * ld + j];
      if (bi == 0x7fffffff || v > best) { best = v; bi = i; }      // ascending j = ascending column: first max stays
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = sv[w];
      const int oi = si[w];
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    tok = bi + 1;
    if (tid == 0) seq[(size_t)m * T + t] = tok;
  }
  if (g == nullptr) return;
  const float* x = tok > 0 ? xg + (size_t)(tok - 1) * 4 * Hd : nullptr;
  for (int j0 = 0; j0 < Hd; j0 += 256 * UPT) {
    if (j0 > 0) {                                   // Hd > 512: further passes load in place
#pragma unroll
      for (int u = 0; u < UPT; ++u) {
        const int j = j0 + tid + u * 256;
        if (j < Hd) {
#pragma unroll
          for (int q = 0; q < 4; ++q) gpre[u][q] = g[q * Hd + j];
          cprev[u] = zero_c ? 0.f : c[(size_t)m * Hd + j];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int j = j0 + tid + u * 256;
      if (j >= Hd) continue;
      float gi = gpre[u][0], gf = gpre[u][1], go = gpre[u][2], gg = gpre[u][3];
      if (x != nullptr) { gi = x[j] + gi; gf = x[Hd + j] + gf; go = x[2 * Hd + j] + go; gg = x[3 * Hd + j] + gg; }
      const float ig = sigmoidf_(gi), fg = sigmoidf_(gf), og = sigmoidf_(go);
      const float gt = th_tanhf(gg);
      const size_t i = (size_t)m * Hd + j;
      const float cn = fg * cprev[u] + ig * gt;
      c[i] = cn;
      h[i] = og * th_tanhf(cn);
    }
  }
}

// split-K finish: C = act(sum_s ws[s] + bias), fixed order
// (m_dev: device-side row count -- the slices sit M rows apart, only rows < *m_dev were written and are finished)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                     float* __restrict__ C, int M, int N, int ldc, int relu,
                                     const int32_t* __restrict__ m_dev) {
  const int N4 = N >> 2;
  const size_t total = (size_t)(m_dev != nullptr ? min(M, *m_dev) : M) * N4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / N4;
    const int n = (int)(i - m * N4) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(ws + m * N + n);
    for (int s = 1; s < S; ++s) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(ws + ((size_t)s * M + m) * N + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] + v[e];
    }
    if (bias) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] + b[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.f ? acc[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(C + m * ldc + n) = acc;
  }
}

// split-K finish of a pooled conv (GemmDesc::pool): slots 4*wl .. 4*wl+3 of the workspace are one pool window
__global__ void splitk_reduce_pool_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                          float* __restrict__ C, int m_begin, int M, int N, int ldc, int H, int Wd,
                                          int relu) {
  const int N4 = N >> 2, Wo = (Wd + 1) >> 1, per = ((H + 1) >> 1) * Wo;
  const size_t total = (size_t)(M >> 2) * N4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t wl = i / N4;
    const int n = (int)(i - wl * N4) * 4;
    const int win = (m_begin >> 2) + (int)wl, wi = win % per, wy = wi / Wo, wx = wi - wy * Wo;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = *reinterpret_cast<const f32x4*>(bias + n);
    f32x4 best = {0.f, 0.f, 0.f, 0.f};
    bool have = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (2 * wy + (c >> 1) >= H || 2 * wx + (c & 1) >= Wd) continue;
      const size_t m = wl * 4 + c;
      f32x4 acc = *reinterpret_cast<const f32x4*>(ws + m * N + n);
      for (int s = 1; s < S; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + ((size_t)s * M + m) * N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + v[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[e] + b[e];
        if (relu) t = t > 0.f ? t : 0.f;
        best[e] = (!have || t > best[e]) ? t : best[e];
      }
      have = true;
    }
    *reinterpret_cast<f32x4*>(C + (size_t)win * ldc + n) = best;
  }
}

__global__ void iota_count_kernel(int32_t* idx, int32_t* count_out, const int32_t* count_in, int cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) idx[i] = i;
  if (i == 0) *count_out = *count_in;
}
__global__ void fill_i32_kernel(int32_t* p, int32_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- recognition heads: objectness (4096->1), box regression (4096->4), final box transform ------
// One wave per RoI row: 5 dot products of length D, wave reduction.
__global__ __launch_bounds__(256) void recog_heads_kernel(const float* __restrict__ codes, const float* __restrict__ w5,
                                                          const float* __restrict__ b5, const float* __restrict__ roi,
                                                          float* __restrict__ obj, float* __restrict__ trans,
                                                          float* __restrict__ fin, float* __restrict__ fin_xyxy, int n,
                                                          int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* x = codes + (size_t)row * D;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = lane * 4; k < D; k += 256) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
#pragma unroll
    for (int o = 0; o < 5; ++o) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w5 + (size_t)o * D + k);
      s[o] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
    }
  }
#pragma unroll
  for (int o = 0; o < 5; ++o) s[o] = wave_sum(s[o]) + b5[o];
  if (lane == 0) {
    obj[row] = s[0];
    const float xa = roi[row * 4 + 0], ya = roi[row * 4 + 1], wa = roi[row * 4 + 2], ha = roi[row * 4 + 3];
    trans[row * 4 + 0] = s[1]; trans[row * 4 + 1] = s[2]; trans[row * 4 + 2] = s[3]; trans[row * 4 + 3] = s[4];
    // nn.ApplyBoxTransform (ApplyBoxTransform.lua:85-88); no FMA contraction to keep op order
    const float fx = __fadd_rn(__fmul_rn(s[1], wa), xa), fy = __fadd_rn(__fmul_rn(s[2], ha), ya);
    const float fw = __fmul_rn(th_expf(s[3]), wa), fh = __fmul_rn(th_expf(s[4]), ha);
    fin[row * 4 + 0] = fx;
    fin[row * 4 + 1] = fy;
    fin[row * 4 + 2] = fw;
    fin[row * 4 + 3] = fh;
    if (fin_xyxy != nullptr) {            // box_utils.xcycwh_to_x1y1x2y2 on the stored values (DenseCapModel.lua:262): one launch less
      float c0, c1, c2, c3;
      corners(fx, fy, fw, fh, c0, c1, c2, c3);
      *reinterpret_cast<f32x4*>(fin_xyxy + (size_t)row * 4) = f32x4{c0, c1, c2, c3};
    }
  }
}

inline int grid_for(size_t total, int block = 256, int cap = 256 * 8) {
  size_t g = (total + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

hipError_t launch_transpose2d(const float* in, float* out, int rows, int cols, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, in, out, rows, cols);
  return hipGetLastError();
}
hipError_t launch_chw_to_hwc(const float* in, float* out, int C, int H, int W, hipStream_t s) {
  return launch_transpose2d(in, out, C, H * W, s);
}
hipError_t launch_hwc_to_chw(const float* in, float* out, int C, int H, int W, hipStream_t s) {
  return launch_transpose2d(in, out, H * W, C, s);
}
hipError_t launch_pack_conv3x3(const float* w, float* out, int Cout, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for((size_t)Cout * Cin * 9)), dim3(256), 0, s, w, out, Cout, Cin);
  return hipGetLastError();
}
hipError_t launch_permute_fc6(const float* in, float* out, int N, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(permute_fc6_kernel, dim3(grid_for((size_t)N * C * HW, 256, 256 * 16)), dim3(256), 0, s, in, out,
                     N, C, HW);
  return hipGetLastError();
}
hipError_t launch_conv3x3_c3(const float* in, const float* w, const float* bias, float* out, int nimg, int H, int W,
                             int Cout, int relu, hipStream_t s) {
  if (Cout != 64 || nimg < 1 || (size_t)3 * H * W >= 0x7fffffffull) return hipErrorInvalidValue;
  // tiles per workgroup: ONE round of three workgroups per CU (41 KiB of LDS each) covers the image (720x600: 5 tiles each,
  // 750 workgroups, 29.9 us = 3.7 TB/s of stores; 3 or 8 tiles per workgroup: 33-34 us; one tile per workgroup, the
  // round-2 kernel: 40.4 us; the same walk without the MFMAs: 23.1 us)
  const int ntx = (W + 31) / 32, nty = (H + 3) / 4;
  const long tiles = (long)ntx * nty * nimg, slots = 3L * device_cu_count();
  int tpw = (int)((tiles + slots - 1) / slots);
  tpw = tpw < 1 ? 1 : (tpw > ntx ? ntx : tpw);
  hipLaunchKernelGGL((conv3x3_c3_kernel<64>), dim3((ntx + tpw - 1) / tpw, nty, nimg), dim3(256), 0, s, in, w, bias, out,
                     H, W, relu, tpw);
  return hipGetLastError();
}
hipError_t launch_maxpool2x2_ceil(const float* in, float* out, int nimg, int H, int W, int C, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const size_t total = (size_t)nimg * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2x2_ceil_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, in, out, nimg, H,
                     W, C);
  return hipGetLastError();
}
hipError_t launch_lstm_pointwise(const float* gates, float* c, float* h, int n, const int32_t* n_dev, int Hd,
                                  int zero_c, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(grid_for((size_t)n * Hd)), dim3(256), 0, s, gates, c, h, n, n_dev, Hd,
                     zero_c);
  return hipGetLastError();
}
hipError_t launch_lstm_step_tail(const float* pval, const int32_t* pidx, int ntiles, int ld, int fixed_tok,
                                 const float* xg, const float* gates_pre, float* c, float* h, int n,
                                 const int32_t* n_dev, int Hd, int zero_c, int32_t* seq, int T, int t, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(lstm_step_tail_kernel, dim3(n), dim3(256), 0, s, pval, pidx, ntiles, ld, fixed_tok, xg, gates_pre,
                     c, h, n, n_dev, Hd, zero_c, seq, T, t);
  return hipGetLastError();
}
hipError_t launch_splitk_reduce(const float* ws, int S, const float* bias, float* C, int M, int N, int ldc, int relu,
                                hipStream_t s, const int32_t* m_dev) {
  if (N % 4 || ldc % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for((size_t)M * (N / 4))), dim3(256), 0, s, ws, S, bias, C, M, N, ldc,
                     relu, m_dev);
  return hipGetLastError();
}
hipError_t launch_splitk_reduce_pool(const float* ws, int S, const float* bias, float* C_pooled, int m_begin, int M, int N,
                                     int ldc, int H, int Wd, int relu, hipStream_t s) {
  if (N % 4 || ldc % 4 || M % 4 || m_begin % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(splitk_reduce_pool_kernel, dim3(grid_for((size_t)(M / 4) * (N / 4))), dim3(256), 0, s, ws, S, bias,
                     C_pooled, m_begin, M, N, ldc, H, Wd, relu);
  return hipGetLastError();
}
hipError_t launch_iota_count(int32_t* idx, int32_t* count_out, const int32_t* count_in, int cap, hipStream_t s) {
  hipLaunchKernelGGL(iota_count_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, idx, count_out, count_in, cap);
  return hipGetLastError();
}
hipError_t launch_fill_i32(int32_t* p, int32_t v, int n, hipStream_t s) {
  hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
  return hipGetLastError();
}
// Weight planes of the split-bf16 mode (mfma_gemm.hip, v2_tile<.., BF3 = 2>): W (N, K) fp32 -> three planes of N x K bf16,
// x = p0 + p1 + p2 exactly (round to nearest at every level, the same chain as the in-register split), the k of every
// 32-tile permuted so that 16-byte chunk c = 2s + h holds k = 16s + 4h + {0..3}, 16s + 8 + 4h + {0..3}.
__global__ void split_planes_kernel(const float* __restrict__ W, uint16_t* __restrict__ P, size_t N, int K) {
  const size_t total = N * (size_t)K;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t n = t / K;
    const int kk = (int)(t - n * K), kt = kk >> 5, o = kk & 31;      // o: position inside the plane's 32-tile
    const int c = o >> 3, e = o & 7, s_ = c >> 1, h = c & 1;
    const int kl = 16 * s_ + 4 * h + (e < 4 ? e : 8 + (e - 4));
    const float x = W[n * K + kt * 32 + kl];
    const __bf16 p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    const __bf16 p1 = (__bf16)r1;
    const float r2 = r1 - (float)p1;
    const __bf16 p2 = (__bf16)r2;
    P[t] = __builtin_bit_cast(uint16_t, p0);
    P[total + t] = __builtin_bit_cast(uint16_t, p1);
    P[2 * total + t] = __builtin_bit_cast(uint16_t, p2);
  }
}
hipError_t launch_split_planes(const float* W, uint16_t* planes, size_t N, int K, hipStream_t s) {
  if (K % 32) return hipErrorInvalidValue;
  const size_t total = N * (size_t)K;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, W, planes, N, K);
  return hipGetLastError();
}

hipError_t launch_recog_heads(const float* codes, const float* w5, const float* b5, const float* roi_boxes,
                              float* obj, float* trans, float* final_boxes, float* final_xyxy, int n, int D, hipStream_t s) {
  if (D % 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(recog_heads_kernel, dim3((n + 3) / 4), dim3(256), 0, s, codes, w5, b5, roi_boxes, obj, trans,
                     final_boxes, final_xyxy, n, D);
  return hipGetLastError();
}
