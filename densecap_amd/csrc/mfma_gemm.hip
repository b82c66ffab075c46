// fp32 MFMA contraction engine for gfx950 (CDNA4).
//
// One kernel family serves every dense contraction on the densecap hot path:
//   * nn.SpatialConvolution 3x3/s1/p1 (+bias+ReLU) of the VGG-16 trunk and the RPN conv
//     (reference: DenseCapModel.lua:73-76, LocalizationLayer.lua:627-636) as an IMPLICIT GEMM:
//     M = pixels, N = Cout, K = 9*Cin; activations are channels-last so one im2col row
//     fragment is a contiguous 128-byte run; im2col patches are built straight into LDS
//     and never exist in HBM;
//   * nn.Linear (+bias+ReLU) for fc6/fc7 (DenseCapModel.lua:133), the LSTM/vocab projections
//     (LanguageModel.lua:27-61) and the fused 1x1 RPN heads.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 64 cycles/SIMD, 157 TF chip peak).
// Tiling: 256-thread workgroup = 4 waves in a 2x2 grid; wave tile (32*TM)x(32*TN);
// block tile BM=64*TM, BN=64*TN, BK=32.  A and B tiles live in LDS as [rows][BK+4] fp32
// (16-byte row pad => conflict-free ds_read_b128 / ds_write_b128).  Each lane fetches 4
// consecutive k with ONE ds_read_b128 and feeds 4 MFMAs: lane-half h=lane>>5 supplies
// k = 8g+4h+j to the j-th MFMA of group g, so the hardware's k-pair is (8g+j, 8g+4+j) --
// a fixed permutation of the summation order, identical for A and B.
// Pipeline: register-staged double buffering, one s_barrier per K-tile: global loads of
// tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer after.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;  // floats per LDS row

template <int TM, int TN, bool CONV>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(GemmDesc d, int ntm, int ntn, int m_fastest) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int PA = BM / 32, PB = BN / 32;  // load passes (32 rows x 8 float4 per pass)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                         // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

  // ---- XCD-aware tile mapping: blocks b, b+8, b+16.. share an XCD (b % 8); give each XCD a
  // contiguous run of logical tile ids so neighbours in the fast dimension share its L2.
  const int nblk = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, o = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
  }
  int tile_m, tile_n;
  if (m_fastest) { tile_m = bid % ntm; tile_n = bid / ntm; }
  else           { tile_n = bid % ntn; tile_m = bid / ntn; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lrow = tid >> 3, lchunk = tid & 7;

  // ---- per-thread load descriptors -------------------------------------------------
  const float* a_ptr[PA];
  int a_y[PA], a_x[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int m = m0 + lrow + 32 * i;
    a_ok[i] = m < d.M;
    if (m >= d.M) m = d.M - 1;
    if constexpr (CONV) {
      const int hw = d.H * d.Wd;
      const int img = m / hw, rem = m - img * hw;
      a_y[i] = rem / d.Wd;
      a_x[i] = rem - a_y[i] * d.Wd;
      a_ptr[i] = d.A + (size_t)m * d.Cin + lchunk * 4;
    } else {
      a_y[i] = a_x[i] = 0;
      a_ptr[i] = d.A + (size_t)m * d.K + lchunk * 4;
    }
  }
  const float* b_ptr[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int n = n0 + lrow + 32 * i;
    if (n >= d.N) n = d.N - 1;
    b_ptr[i] = d.W + (size_t)n * d.K + lchunk * 4;
  }

  f32x4 ra[PA], rb[PB];
  // conv K-walk state: tap (dy,dx) and channel offset
  int tap = 0, c0 = 0;

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + (size_t)kt * BK);
    if constexpr (CONV) {
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int off = (dy * d.Wd + dx) * d.Cin + c0;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bool ok = a_ok[i] && (unsigned)(a_y[i] + dy) < (unsigned)d.H && (unsigned)(a_x[i] + dx) < (unsigned)d.Wd;
        const float* p = ok ? a_ptr[i] + off : a_ptr[i];
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        ra[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      c0 += BK;
      if (c0 >= d.Cin) { c0 = 0; ++tap; }
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (size_t)kt * BK);
    }
  };
  auto store_tile = [&](int buf) {
    float* as = As + buf * BM * LDS_LD;
    float* bs = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(as + (lrow + 32 * i) * LDS_LD + lchunk * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(bs + (lrow + 32 * i) * LDS_LD + lchunk * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nkt = d.K / BK;
  const int r = lane & 31, hsel = lane >> 5;
  const int a_frag_off = (wm * 32 * TM + r) * LDS_LD + hsel * 4;
  const int b_frag_off = (wn * 32 * TN + r) * LDS_LD + hsel * 4;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const float* as = As + buf * BM * LDS_LD + a_frag_off;
    const float* bs = Bs + buf * BN * LDS_LD + b_frag_off;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_LD + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_LD + g * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias (+ gathered row term) + ReLU, channels-last store -------------------
  // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 32 * TN + j * 32 + r;
    const bool n_ok = n < d.N;
    const float bv = (d.bias != nullptr && n_ok) ? d.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * 32 * TM + i * 32 + 4 * hsel;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (n_ok && m < d.M) {
          float v;
          if (d.rowterm != nullptr) v = d.rowterm[(size_t)(d.rowidx[m] - 1) * d.rowterm_ld + n] + acc[i][j][e];
          else v = acc[i][j][e] + bv;
          if (d.relu) v = v > 0.f ? v : 0.f;
          d.C[(size_t)m * d.ldc + n] = v;
        }
      }
    }
  }
}

template <int TM, int TN, bool CONV>
hipError_t launch_cfg(const GemmDesc& d, hipStream_t stream) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_gemm_kernel<TM, TN, CONV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int m_fastest = ntm <= ntn ? 1 : 0;
  hipLaunchKernelGGL((mfma_gemm_kernel<TM, TN, CONV>), dim3(ntm * ntn), dim3(256), lds, stream, d, ntm, ntn,
                     m_fastest);
  return hipGetLastError();
}

template <bool CONV>
hipError_t launch_pick(const GemmDesc& d, hipStream_t stream) {
  // Tile choice: largest tile that still yields >= ~1.5 workgroups per CU (256 CUs).
  auto blocks = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
  if (d.N > 64 && blocks(128, 128) >= 384) return launch_cfg<2, 2, CONV>(d, stream);
  if (d.N <= 64 && blocks(128, 64) >= 384) return launch_cfg<2, 1, CONV>(d, stream);
  if (d.N > 64 && blocks(128, 128) >= 200 && blocks(128, 128) <= 256) return launch_cfg<2, 2, CONV>(d, stream);
  if (blocks(128, 64) >= 384) return launch_cfg<2, 1, CONV>(d, stream);
  return launch_cfg<1, 1, CONV>(d, stream);
}

}  // namespace

double gemm_flops(const GemmDesc& d) { return 2.0 * (double)d.M * (double)d.N * (double)d.K; }

hipError_t launch_mfma_gemm(const GemmDesc& d, hipStream_t stream) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K % BK) != 0) return hipErrorInvalidValue;
  if (d.conv) {
    if (d.Cin % BK != 0 || d.K != 9 * d.Cin) return hipErrorInvalidValue;
    return launch_pick<true>(d, stream);
  }
  return launch_pick<false>(d, stream);
}
