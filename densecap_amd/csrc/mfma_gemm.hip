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
// Two kernels share the interface (GemmDesc) and the K order of every output element:
//   * mfma_gemm_v2_kernel<TM,TN,CONV,NS[,AMAX]>  -- the workhorse: 2x2 waves, wave tile (32*TM)x(32*TN), operands
//     HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds) into an NS-stage ring of unpadded, XOR-swizzled 128-byte rows,
//     one barrier per K-tile in the middle of the tile's MFMAs.  AMAX = fused row arg-max epilogue (vocabulary
//     projection) with transposed accumulator blocks; with GemmDesc::amax_cols it also carries the h.Wh half of the
//     next decode step's gates in the same launch;
//   * mfma_gemm_ks_kernel<CONV>                  -- 128x128 tile, the four waves split K instead of the tile (half the
//     LDS->VGPR traffic), cross-wave reduction through LDS at the end; optional split-K over workgroups and row
//     windows (tail plans).
// Both address their operands through 32-bit buffer offsets: an operand tile set beyond 4 GiB (an image of more than
// ~16 Mpx) is refused with an error, not routed elsewhere.
// Each lane fetches 4 consecutive k with ONE ds_read_b128 and feeds 4 MFMAs: lane-half h=lane>>5 supplies
// k = 8g+4h+j to the j-th MFMA of group g, so the hardware's k-pair is (8g+j, 8g+4+j) -- a fixed permutation of the
// summation order, identical for A and B and for both kernels.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

// timing-only ablation builds of the split-bf16 K loop (tools/lab_session.sh; wrong results on purpose, never the product)
#ifdef BF3_ABL_A            // no split, no LDS-DMA after the first tiles
#define BF3_ABL_NO_SPLIT
#define ABL_NO_DMA
#endif
#ifdef BF3_ABL_B            // A + no rendezvous
#define BF3_ABL_NO_SPLIT
#define ABL_NO_DMA
#define BF3_ABL_NO_BARRIER
#endif
#ifdef BF3_ABL_C            // B + no fragment reads: the MFMAs alone
#define BF3_ABL_NO_SPLIT
#define ABL_NO_DMA
#define BF3_ABL_NO_BARRIER
#define BF3_ABL_NO_READS
#endif
#ifdef BF3_ABL_D            // the real loop without its rendezvous
#define BF3_ABL_NO_BARRIER
#endif
constexpr int BK = 32;
// timing-only ablation build ABL_EPI_NO_STORE: the 2x2-wave kernel computes its epilogue but does not store it
#if defined(ABL_EPI_NO_STORE) && defined(__HIP_DEVICE_COMPILE__)
#define EPI_STORE(lhs, val) asm volatile("" ::"v"(val))
#elif defined(EXP_EPI_NT)                                  // experiment build: results leave as non-temporal stores
#define EPI_STORE(lhs, val) __builtin_nontemporal_store((val), &(lhs))
#else
#define EPI_STORE(lhs, val) (lhs) = (val)
#endif
// K-tiles from which the K-split 128x128 kernel (one workgroup per CU) beats the 128x64 kernel.  32 while the latter ran two
// workgroups per CU; with the two-stage ring's three (round 3) the crossover moved past K = 2304: conv2_2 (K = 1152)
// 340 -> 285 us, conv3_1 185 -> 152, conv3_2 / conv3_3 (K = 2304) 332 -> 285 / 315 -> 277 in the multi-lane planning,
// 312 -> 284 / 299 -> 272 in single-image mode; K = 4608 (conv4_2.., conv5_x) still belongs to the K-split kernel
// (single-image conv4_2 299 vs 309 us, conv5_1 98 vs 104).
constexpr int KS_MIN_KTILES = 96;
constexpr unsigned CV_PAD = 0xffffe000u;   // conv padding marker: voffset (+ up to 8 KiB of channel offset) past any descriptor range

// ---- split-bf16 arithmetic (GemmDesc::bf3, dc_set_math_mode(1); opt-in, never the default) -------------------------------
// An fp32 value is the exact sum of three bf16 values: x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (8 + 8 + 8
// significand bits, round-to-nearest at every level; both residuals are exact in fp32).  A product a*b is then the sum of
// nine bf16 x bf16 products, each exact in fp32; the six of order <= 2 (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0) carry it to
// 2^-26 relative -- below the rounding of one fp32 multiply-add -- and run on v_mfma_f32_32x32x16_bf16 at sixteen times the
// fp32 MFMA rate: 6 instructions of 32 cycles per 16 k against 8 of 64, i.e. 2.67x the matrix throughput at fp32-class
// accuracy.  Operands stay fp32 in HBM and in LDS (same loads, same im2col assembly); the split happens in registers, on
// the fragments a lane has just read, with the vector unit that the fp32 loop leaves idle.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Bf3 { u32x4 p[3]; };        // three planes of 8 bf16 (packed pairs), ready as MFMA operands

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32: lo = bf16(a), hi = bf16(b), RNE
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// elements 2i, 2i+1 of the 8 values (lo[0..3], hi[0..3]) -> pair i of the three planes
__device__ __forceinline__ void split3_pair(const float xa, const float xb, Bf3& out, const int i) {
  unsigned a0 = pk_bf16(xa, xb);
  const float ra = xa - __builtin_bit_cast(float, a0 << 16), rb = xb - __builtin_bit_cast(float, a0 & 0xffff0000u);
  unsigned a1 = pk_bf16(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, a1 << 16), sb = rb - __builtin_bit_cast(float, a1 & 0xffff0000u);
  unsigned a2 = pk_bf16(sa, sb);
  // The split stays where it is written.  Instruction selection places a value without side effects as late as its first
  // use allows -- every split of a step would land in one block in front of the NEXT step's MFMAs, whatever the program
  // order and the sched_barriers say (measured: 160 vector instructions, then 16 MFMAs back to back).  An empty volatile
  // asm that "rewrites" the three results is ordered like a memory operation and costs no instruction.
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));
  out.p[0][i] = a0;
  out.p[1][i] = a1;
  out.p[2][i] = a2;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Implicit-GEMM row m -> input pixel (y, x) and its byte offset in the channels-last activation.  Plain convs walk the
// pixels in raster order; pooled convs (GemmDesc::pool) walk pool windows, four consecutive m per window.
__device__ __forceinline__ void conv_pixel(const GemmDesc& d, int m, int hw, int& y, int& x, bool& ok, unsigned& off) {
  if (d.pool) {
    const int Wo = (d.Wd + 1) >> 1, per = ((d.H + 1) >> 1) * Wo;       // windows per image
    const int win = m >> 2, img = win / per, wi = win - img * per;
    const int wy = wi / Wo, wx = wi - wy * Wo;
    y = 2 * wy + ((m >> 1) & 1);
    x = 2 * wx + (m & 1);
    ok = ok && y < d.H && x < d.Wd;
    off = ok ? (unsigned)(img * hw + y * d.Wd + x) * (unsigned)d.Cin * 4u : 0u;
  } else {
    const int img = m / hw, rem = m - img * hw;
    y = rem / d.Wd;
    x = rem - y * d.Wd;
    off = (unsigned)m * (unsigned)d.Cin * 4u;
  }
}
// input pixels behind a conv problem (extent of the activation buffer): pooled problems count window slots in M
__device__ __forceinline__ size_t conv_input_pixels(const GemmDesc& d, int rows) {
  if (!d.pool) return (size_t)rows;
  const int per = ((d.H + 1) >> 1) * ((d.Wd + 1) >> 1);
  const int slots = d.a_rows ? d.a_rows : d.M;          // the whole problem's slots when this launch is a row window
  return (size_t)(slots / (4 * per)) * d.H * d.Wd;
}
// Pooled epilogue of one lane's four consecutive rows (one pool window): max over the window's in-image pixels of
// act(v + bias).  win = first row >> 2 (window index over all images of the launch).
__device__ __forceinline__ float pool_window(const GemmDesc& d, int win, float v0, float v1, float v2, float v3, float bv) {
  const int Wo = (d.Wd + 1) >> 1, per = ((d.H + 1) >> 1) * Wo;
  const int wi = win % per;
  const int wy = wi / Wo, wx = wi - wy * Wo;
  const bool hx = 2 * wx + 1 < d.Wd, hy = 2 * wy + 1 < d.H;
  float t0 = v0 + bv, t1 = v1 + bv, t2 = v2 + bv, t3 = v3 + bv;
  if (d.relu) {
    t0 = t0 > 0.f ? t0 : 0.f; t1 = t1 > 0.f ? t1 : 0.f; t2 = t2 > 0.f ? t2 : 0.f; t3 = t3 > 0.f ? t3 : 0.f;
  }
  float best = t0;
  if (hx) best = t1 > best ? t1 : best;
  if (hy) best = t2 > best ? t2 : best;
  if (hx && hy) best = t3 > best ? t3 : best;
  return best;
}

// =========================================================================================
// v2: LDS-DMA (buffer_load ... lds) ring of three stages (two for the 128x64 tiles of launches with >= 3 tiles per CU:
// three workgroups per CU), fragment double-buffering, ONE barrier per K-tile placed in the MIDDLE of the tile's MFMA
// sequence (two-stage ring: behind its third group).
//
//  * operands go HBM/L2 -> LDS directly (no staging VGPRs, no ds_write); conv zero padding
//    comes for free from the buffer descriptor's out-of-range rule (offset >= num_records
//    returns 0), so the im2col patch is assembled by the memory unit;
//  * LDS rows are 128 B, unpadded (an LDS-DMA wave instruction writes 1 KiB linearly = 8 rows);
//    bank conflicts are avoided by XOR-swizzling the 16-byte chunk index with (row>>1)&7 on the
//    SOURCE address and on the fragment read;
//  * tile t+2 is requested right after the barrier of tile t, i.e. a full K-tile (64 MFMAs per
//    wave = 4096 cycles) before it is needed; the barrier sits after MFMA group 1 so the first
//    fragments of tile t+1 are fetched from LDS while groups 2-3 of tile t execute: the MFMA
//    pipe never waits for an LDS round trip.
// =========================================================================================
// AMAX: the fused row arg-max variant (vocab projection + torch.max).  Its MFMAs take the two operands in swapped
// roles, so each 32x32 accumulator block is held TRANSPOSED: a lane owns one output row m (lane&31) and its registers
// walk 16 of the block's 32 columns n.  The row arg-max is then a compare chain inside the lane -- no LDS transpose.
// Each element is the same fp32 fmaf chain over k either way.
// One output tile [m0, m0+BM) x [n0, n0+BN) (tile_n = n0 / BN indexes the arg-max partials); Meff = rows of the problem.
// BF3: 0 = fp32 MFMA; 1 = split-bf16, both operands split in registers; 2 = split-bf16 with the B operand (weights) split ONCE
// at load: three bf16 planes in HBM (GemmDesc::sk_slots carries the pointer in this mode), fetched by LDS-DMA as they are
template <int TM, int TN, bool CONV, int NS, bool AMAX, int BF3 = 0>
__device__ __forceinline__ void v2_tile(const GemmDesc& d, const int m0, const int n0, const int tile_n, const int Meff,
                                        float* const smem) {
  static_assert(!(AMAX && CONV), "arg-max epilogue is for dense GEMMs");
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int PA = BM / 32, PB = BN / 32;
  // floats per ring stage; BF3 == 2: the B tile is three bf16 planes of BN rows x 32 k (64 bytes a row) instead of fp32 rows
  constexpr int STAGE = BF3 == 2 ? BM * BK + 3 * BN * (BK / 2) : (BM + BN) * BK;
  constexpr int PB3 = BF3 == 2 ? (BN / 64 > 0 ? BN / 64 : 1) : 1;     // 64-row passes of a plane: 4 waves x 16 rows x 64 B = one LDS-DMA piece each
  constexpr int NPIECE = BF3 == 2 ? PA + 3 * PB3 : PA + PB;            // LDS-DMA pieces of a K-tile per wave

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int lrow = tid >> 3;                              // 0..31: row within a 32-row pass
  const int lchunk = (tid & 7) ^ ((lrow >> 1) & 7);        // LOGICAL 16-B chunk this lane fetches

  // ---- buffer descriptors (wave-uniform) + per-lane byte offsets ---------------------------
  __amdgpu_buffer_rsrc_t rsrcA, rsrcB;
  unsigned a_off[PA];
  int a_y[PA], a_x[PA];
  bool a_ok[PA];
  if constexpr (CONV) {
    const size_t bytes = conv_input_pixels(d, Meff) * d.Cin * 4;
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
    const int hw = d.H * d.Wd;
    // Round 4: an instruction issued at a tile's edges costs the launch about as much as one in its K loop (idle time is
    // covered by the co-resident workgroups, issued instructions are not), and a row's pixel costs two integer divisions
    // (~100 VALU instructions).  Interior tiles divide once per lane: the next row pass is 32 rows on -- a step of 32
    // pixels (8 pool windows) that wraps the image row at most once.
    const bool step_ok = m0 + BM <= Meff && (d.pool ? ((d.Wd + 1) >> 1) >= 8 : d.Wd >= 32);
    if (step_ok) {
      if (d.pool) {
        const int Wo = (d.Wd + 1) >> 1, Ho = (d.H + 1) >> 1, per = Ho * Wo;
        const int m = m0 + lrow, win = m >> 2;
        int img = win / per;
        const int wi = win - img * per;
        int wy = wi / Wo, wx = wi - wy * Wo;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const int y = 2 * wy + ((m >> 1) & 1), x = 2 * wx + (m & 1);
          a_y[i] = y; a_x[i] = x;
          a_ok[i] = y < d.H && x < d.Wd;
          a_off[i] = (a_ok[i] ? (unsigned)(img * hw + y * d.Wd + x) * (unsigned)d.Cin * 4u : 0u) + (unsigned)lchunk * 16u;
          wx += 8;
          if (wx >= Wo) {
            wx -= Wo;
            if (++wy >= Ho) { wy = 0; ++img; }
          }
        }
      } else {
        const int m = m0 + lrow, img = m / hw, rem = m - img * hw;
        int y = rem / d.Wd, x = rem - y * d.Wd;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          a_y[i] = y; a_x[i] = x; a_ok[i] = true;
          a_off[i] = (unsigned)(m + 32 * i) * (unsigned)d.Cin * 4u + (unsigned)lchunk * 16u;
          x += 32;
          if (x >= d.Wd) {
            x -= d.Wd;
            if (++y >= d.H) y = 0;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        int m = m0 + lrow + 32 * i;
        a_ok[i] = m < Meff;
        if (m >= Meff) m = Meff - 1;
        conv_pixel(d, m, hw, a_y[i], a_x[i], a_ok[i], a_off[i]);
        a_off[i] += (unsigned)lchunk * 16u;
      }
    }
  } else {
    const float* baseA = d.A + (size_t)m0 * d.K;
    const size_t bytes = (size_t)(Meff - m0) * d.K * 4;
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int rr = lrow + 32 * i;
      if (m0 + rr >= Meff) rr = Meff - 1 - m0;
      a_off[i] = (unsigned)rr * (unsigned)d.K * 4u + (unsigned)lchunk * 16u;
      a_ok[i] = true; a_y[i] = a_x[i] = 0;
    }
  }
  unsigned b_off[PB];
  unsigned b3_off[PB3];
  unsigned b3_plane = 0;                 // bytes of one plane
  if constexpr (BF3 == 2) {
    // Weight planes (split once at load, densecap.hip::weight_planes): plane p = N rows of K bf16, the k of every 32-tile
    // permuted so that 16-byte chunk 2s+h holds k = 16s + 4h + {0..3}, 16s + 8 + 4h + {0..3} -- the k a lane half h meets
    // in step s on the A side.  LDS rows are 64 bytes; chunk c of row r sits at slot c ^ ((r >> 2) & 3) (conflict-free
    // ds_read_b128 over the instruction's 16-lane groups), the swizzle applied here on the SOURCE chunk a lane fetches.
    const size_t prow = d.sk_np > 0 ? d.sk_np : d.N;          // rows of the plane matrix (>= N: the decode's last step uses a row prefix)
    const size_t bytes = (size_t)3 * prow * d.K * 2;
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)d.sk_slots, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
    b3_plane = (unsigned)(prow * d.K * 2);
#pragma unroll
    for (int j = 0; j < PB3; ++j) {
      const int row = 64 * j + 16 * wid + (lane >> 2);
      int grow = n0 + row;
      if (grow >= d.N) grow = d.N - 1;
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      b3_off[j] = (unsigned)grow * (unsigned)d.K * 2u + (unsigned)c * 16u;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) b_off[i] = 0;
  } else {
    const float* baseB = d.W + (size_t)n0 * d.K;
    const size_t bytes = (size_t)(d.N - n0) * d.K * 4;
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int rr = lrow + 32 * i;
      if (n0 + rr >= d.N) rr = d.N - 1 - n0;
      b_off[i] = (unsigned)rr * (unsigned)d.K * 4u + (unsigned)lchunk * 16u;
    }
#pragma unroll
    for (int j = 0; j < PB3; ++j) b3_off[j] = 0;
  }

  int tap = 0, c0 = 0;  // conv K-walk
  // Request K-tile `kt` into ring stage `st`, one LDS-DMA instruction (1 KiB) per call: piece p in
  // [0,PA) is an A row-pass, [PA,PA+PB) a B row-pass.  An LDS-DMA instruction occupies the issuing
  // wave for ~60+ cycles, so the pieces are interleaved 1:1 with MFMAs by the caller.
  // im2col addressing: the per-lane byte offsets (tap displacement and the padding test folded in; CV_PAD = beyond the
  // descriptor's range -> the memory unit returns zeros) are recomputed only when the K walk enters a new tap, i.e. every
  // Cin/32 K-tiles; within a tap a K-tile only moves the scalar channel offset.  The per-piece issue is then a bare
  // buffer_load (the earlier per-piece compare/select chain cost ~3 % of the loop).
  unsigned cv_vo[PA];
  int cv_soff = 0;
  bool cv_new_tap = true;
  auto issue_begin = [&]() {
    if constexpr (CONV) {
      if (cv_new_tap) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const unsigned toff = (unsigned)((dy * d.Wd + dx) * d.Cin * 4);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const bool ok = a_ok[i] && (unsigned)(a_y[i] + dy) < (unsigned)d.H && (unsigned)(a_x[i] + dx) < (unsigned)d.Wd;
          cv_vo[i] = ok ? a_off[i] + toff : CV_PAD;
        }
      }
      cv_soff = c0 * 4;
      c0 += BK;
      cv_new_tap = c0 >= d.Cin;
      if (cv_new_tap) { c0 = 0; ++tap; }
    }
  };
  auto issue_piece = [&](int kt, int st, int p) {
    float* sa = smem + st * STAGE + (8 * wid) * BK;       // this wave's 8-row slice of pass 0
    float* sb = sa + BM * BK;
#ifdef ABL_NO_DMA
    if (kt >= 2) return;                                   // timing-only ablation build: operands of later K-tiles never arrive
#endif
    if (p < PA) {
      if constexpr (CONV) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, sa + p * 32 * BK, 16, (int)cv_vo[p], cv_soff, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, sa + p * 32 * BK, 16, (int)a_off[p], kt * (BK * 4), 0, 0);
      }
    } else if constexpr (BF3 == 2) {
      const int q = p - PA, plane = q / PB3, j = q - plane * PB3;     // this wave's 16 rows of pass j of one plane
      float* dst = smem + st * STAGE + BM * BK + plane * (BN * (BK / 2)) + (64 * j + 16 * wid) * (BK / 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, dst, 16, (int)b3_off[j], kt * (BK * 2) + plane * (int)b3_plane, 0, 0);
    } else {
      const int i = p - PA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, sb + i * 32 * BK, 16, (int)b_off[i], kt * (BK * 4), 0, 0);
    }
  };
  auto issue = [&](int kt, int st) {
    issue_begin();
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) issue_piece(kt, st, p);
  };

  // ---- fragment addressing --------------------------------------------------------------------
  const int r = lane & 31, hsel = lane >> 5;
  int foff[4];  // float offset of logical chunk (2g+hsel) of row r inside a 32-row block
#pragma unroll
  for (int g = 0; g < 4; ++g) foff[g] = r * BK + (((2 * g + hsel) ^ ((r >> 1) & 7)) << 2);
  const int a_base = (wm * 32 * TM) * BK;
  const int b_base = BM * BK + (wn * 32 * TN) * BK;

  f32x4 fa[2][TM], fb[2][TN];
#ifdef ABL_NO_READS
  bool abl_reads_on = true;
#endif
  auto read_piece = [&](int st, int g, int set, int p) {   // p in [0,TM): A fragment, [TM,TM+TN): B fragment
#ifdef ABL_NO_READS
    if (!abl_reads_on) return;                             // timing-only ablation build: fragments are read once, then reused
#endif
    const float* base = smem + st * STAGE + foff[g];
    if (p < TM) fa[set][p] = *reinterpret_cast<const f32x4*>(base + a_base + p * 32 * BK);
    else fb[set][p - TM] = *reinterpret_cast<const f32x4*>(base + b_base + (p - TM) * 32 * BK);
  };
  auto read_frag = [&](int st, int g, int set) {
#pragma unroll
    for (int p = 0; p < TM + TN; ++p) read_piece(st, g, set, p);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int NM = 4 * TM * TN;        // MFMAs per group
  auto mfma_slot = [&](int set, int q) {
    const int e = q / (TM * TN), rem = q % (TM * TN), i = rem / TN, j = rem % TN;
#ifdef V2_AGPR_ACC
    // experiment: accumulators pinned to the AGPR half of the register file (the K-split kernel's are there by pressure)
    if constexpr (AMAX) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[set][j][e]), "v"(fa[set][i][e]));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[set][i][e]), "v"(fb[set][j][e]));
#else
    if constexpr (AMAX) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j][e], fa[set][i][e], acc[i][j], 0, 0, 0);
    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][e], fb[set][j][e], acc[i][j], 0, 0, 0);
#endif
  };
  // one MFMA group with `nside` side operations spread evenly between its MFMAs (pinned order)
  auto group_with = [&](int set, int nside, auto&& side) {
#pragma unroll
    for (int q = 0; q < NM; ++q) {
      mfma_slot(set, q);
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < nside && (k * NM) / nside == q) {
          side(k);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };

  const int nkt = d.K / BK;
  // prologue: tiles 0..NS-2 in flight (two-stage ring: both stages; split-bf16 loop: all NS stages, see below)
#pragma unroll
  for (int p = 0; p < (NS == 2 || BF3 ? NS : NS - 1); ++p)
    if (p < nkt) issue(p, p);
  // Arg-max tiles: this lane's 4 x 4 bias values per column block are requested NOW, behind the first operand tiles, and
  // wait in registers: the epilogue then starts with nothing to fetch.  (Round 4: a tile's epilogue is NOT hidden by its two
  // co-resident workgroups -- a build without epilogues ran the decode step 10 % faster -- so every dependent memory round
  // trip and barrier in it costs the launch.)
  f32x4 amax_bias[AMAX ? TN : 1][4];
  if constexpr (AMAX) {
    const int an = d.amax_cols > 0 ? d.amax_n : d.N;
    const bool amax_tile = !(d.amax_cols > 0 && n0 >= d.amax_cols);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int nb = n0 + wn * 32 * TN + j * 32 + 8 * q4 + 4 * (lane >> 5);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (amax_tile && d.bias != nullptr) {
          if (nb + 3 < an) bv = *reinterpret_cast<const f32x4*>(d.bias + nb);
          else
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = nb + c < an ? d.bias[nb + c] : 0.f;
        }
        amax_bias[j][q4] = bv;
      }
  }
  // the other epilogues' bias values (one per column block and lane) travel the same way
  float bias_r[AMAX ? 1 : TN];
  if constexpr (!AMAX) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 32 * TN + j * 32 + r;
      bias_r[j] = (d.bias != nullptr && n < d.N) ? d.bias[n] : 0.f;
    }
  }
#ifndef ABL_NO_PROLOGUE_WAIT                              // timing-only ablation: the first K-tile is multiplied before it has landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#endif
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!BF3) read_frag(0, 0, 0);
#ifdef ABL_NO_READS
  read_frag(0, 1, 1);
  abl_reads_on = false;
#endif

  // One K-tile whose ring stage ST is a compile-time constant: every LDS address of its fragment reads and LDS-DMA
  // destinations is an immediate (the ring walk costs no address arithmetic between the MFMAs).
  auto body = [&](int kt, auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    constexpr int st1 = st == NS - 1 ? 0 : st + 1;
    constexpr int stn = st == 0 ? NS - 1 : st - 1;     // stage of tile kt-1 == stage of tile kt+NS-1
    if constexpr (NS == 2) {
      // Two-stage ring (48 KiB for a 128x64 tile: THREE workgroups per CU instead of two): the rendezvous sits after
      // group 2, when every fragment of tile kt has been requested -- and, since round 6, is waited for (lgkmcnt) -- so that its
      // stage is free for tile kt+2, requested a full tile ahead of its first use; tile kt+1 (requested during tile kt-1) is
      // waited for at the same point.
      group_with(0, TM + TN, [&](int k) { read_piece(st, 1, 1, k); });
      group_with(1, TM + TN, [&](int k) { read_piece(st, 2, 0, k); });
      group_with(0, TM + TN, [&](int k) { read_piece(st, 3, 1, k); });
      __builtin_amdgcn_sched_barrier(0);
#ifndef ABL_NO_WAITCNT
      // Round 6: lgkmcnt(0) too.  The group-3 fragments just REQUESTED from this stage are consumed after the rendezvous, and right
      // behind it tile kt+2 is requested INTO this stage: a fragment read still queued in a congested LDS (three workgroups per CU,
      // plus whatever another stream's kernel does there) could be overtaken by LDS-DMA data that comes back from the L1 in ~150 ns
      // -- one 16-byte chunk of tile kt+2 multiplied in place of tile kt's.  The three-stage ring requests into the PREVIOUS
      // tile's stage and the K-split kernel has consumed its fragments by then; only this ring had the window (tools/soak.py:
      // a wrong element about once in 100,000 images, only with kernels of several streams on the chip).  The reads were issued
      // >= 3 MFMAs ago: the wait is normally already satisfied.
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifndef ABL_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
      const bool more2 = kt + 2 < nkt, next2 = kt + 1 < nkt;
      if (more2) issue_begin();
      group_with(1, PA + PB + TM + TN, [&](int k) {
        if (k < PA + PB) { if (more2) issue_piece(kt + 2, st, k); }
        else if (next2) read_piece(st1, 0, 0, k - (PA + PB));
      });
      return;
    }
    group_with(0, TM + TN, [&](int k) { read_piece(st, 1, 1, k); });
    group_with(1, TM + TN, [&](int k) { read_piece(st, 2, 0, k); });
    // ---- mid-tile rendezvous: tile kt+1 has landed everywhere; the stage of tile kt-1 is free
    __builtin_amdgcn_sched_barrier(0);
#ifndef ABL_NO_WAITCNT
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * (PA + PB)) : "memory");
#endif
#ifndef ABL_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
    const bool more = kt + NS - 1 < nkt;
    if (more) issue_begin();
    // group 2: the PA+PB LDS-DMA pieces of tile kt+NS-1 and the fragment reads of group 3
    group_with(0, PA + PB + TM + TN, [&](int k) {
      if (k < PA + PB) { if (more) issue_piece(kt + NS - 1, stn, k); }
      else read_piece(st, 3, 1, k - (PA + PB));
    });
    const bool next = kt + 1 < nkt;
    group_with(1, TM + TN, [&](int k) { if (next) read_piece(st1, 0, 0, k); });
  };
  auto ring_round = [&](int kt, bool guarded) __attribute__((always_inline)) {
    if (!guarded || kt + 0 < nkt) body(kt + 0, std::integral_constant<int, 0>{});
    if (!guarded || kt + 1 < nkt) body(kt + 1, std::integral_constant<int, 1>{});
    if constexpr (NS >= 3) {
      if (!guarded || kt + 2 < nkt) body(kt + 2, std::integral_constant<int, 2>{});
    }
    if constexpr (NS == 4) {
      if (!guarded || kt + 3 < nkt) body(kt + 3, std::integral_constant<int, 3>{});
    }
  };
  if constexpr (!BF3) {
    int kt = 0;
    for (; kt + NS <= nkt; kt += NS) ring_round(kt, false);
    if (kt < nkt) ring_round(kt, true);
  } else {
    // ---- split-bf16 K loop ------------------------------------------------------------------------------------------
    // A K-tile (32 k) is two STEPS of 16 k; step s takes the fragment chunks of groups 2s and 2s+1 (a lane's 8 values:
    // k = 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3} -- the same set for A and B, which is all a contraction needs).
    // Software pipeline, one step deep at each level:   step j:  MFMAs on planes[j & 1]
    //                                                            || split raw (step j+1, read during step j-1) -> planes[(j+1) & 1]
    //                                                            || then read raw <- step j+2
    // so ALL LDS reads of tile kt+1 happen during tile kt, and a tile's ring stage is free from the rendezvous of the NEXT
    // tile on: at the rendezvous of tile kt (middle of its step 0) tile kt+1 must have landed and tile kt+NS is requested
    // into stage kt % NS -- NS - 1 tiles of lead.
    constexpr int NMF = 6 * TM * TN;        // MFMAs of a step
    constexpr int NSPL = BF3 == 2 ? TM : TM + TN;   // blocks split in the loop (BF3 == 2: the weights arrive split)
    constexpr int NCV = 4 * NSPL;           // pair conversions of a step (4 per 32-row block)
    constexpr int NRD = 2 * NSPL;           // fp32 fragment reads of a step (2 chunks per block)
    constexpr int NBR = BF3 == 2 ? 3 * TN : 0;      // plane reads of a step (BF3 == 2: three 16-byte operands per B block)
    constexpr int NDM = NPIECE;             // LDS-DMA pieces of a K-tile
    // TWO accumulators per 32x32 block: `acc` takes the leading products a0.b0 only, `lo` the five lower-order ones, and the
    // two meet once, after the K loop.  The bf16 MFMA aligns its sixteen products to the largest exponent among them and C and
    // keeps three guard bits (tools/probes/mfma_bf16_numerics.hip: sixteen products of 1/16 ulp(C) vanish, of 1/8 ulp add up):
    // added straight into a large running sum, the a1.b0 / a0.b1 terms would lose their low bits at EVERY step -- measured
    // 3.4x the fp32 path's error at K = 4608.  Kept apart, they are aligned against a sum 2^-8 as large, and the main
    // accumulator is rounded once per 16 k instead of once per k: the error comes out below the fp32 path's.
    f32x16 lo[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) lo[i][j][e] = 0.f;
    Bf3 pl[2][TM + TN];                     // planes: [set][A blocks 0..TM-1, B blocks TM..TM+TN-1]
    f32x4 raw[TM + TN][2];                  // the fragments of ONE step as read
    auto read_raw = [&](int st, int s, int k) {           // k in [0, NRD): block k >> 1, chunk k & 1
#ifdef BF3_ABL_NO_READS
      if (st != 0 || s != 0) return;                      // timing-only ablation build: fragments are read in the prologue only
#endif
      const int p = k >> 1, c = k & 1;
      const float* base = smem + st * STAGE + foff[2 * s + c];
      if (p < TM) raw[p][c] = *reinterpret_cast<const f32x4*>(base + a_base + p * 32 * BK);
      else raw[p][c] = *reinterpret_cast<const f32x4*>(base + b_base + (p - TM) * 32 * BK);
    };
    auto convert = [&](int set, int k) {                  // k in [0, NCV): block k >> 2, pair k & 3
      const int p = k >> 2, i = k & 3;
#ifdef BF3_ABL_NO_SPLIT                                   // timing-only ablation build: the raw bits go to the MFMAs unsplit (wrong results)
      pl[set][p].p[0][i] = __builtin_bit_cast(unsigned, raw[p][i >> 1][2 * (i & 1)]);
      pl[set][p].p[1][i] = __builtin_bit_cast(unsigned, raw[p][i >> 1][2 * (i & 1) + 1]);
      pl[set][p].p[2][i] = pl[set][p].p[0][i];
      return;
#endif
      split3_pair(raw[p][i >> 1][2 * (i & 1)], raw[p][i >> 1][2 * (i & 1) + 1], pl[set][p], i);
    };
    auto read_b3 = [&](int st, int s, int set, int k) {   // BF3 == 2; k in [0, NBR): B block k / 3, plane k % 3 -> straight into the operand
      const int j = k / 3, pp = k - 3 * j;
      const float* src = smem + st * STAGE + BM * BK + pp * (BN * (BK / 2)) + (wn * 32 * TN + j * 32 + r) * (BK / 2) +
                         (((2 * s + hsel) ^ ((r >> 2) & 3)) << 2);
      pl[set][TM + j].p[pp] = *reinterpret_cast<const u32x4*>(src);
    };
    auto mfma3 = [&](int set, int q) {                    // q in [0, NMF): product q / (TM TN) of block q % (TM TN)
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};      // a2b0 a0b2 a1b1 a1b0 a0b1 -> lo;  a0b0 -> acc
      const int e = q / (TM * TN), rem = q % (TM * TN), i = rem / TN, j = rem % TN;
      f32x16& dst = e == 5 ? acc[i][j] : lo[i][j];
      if constexpr (AMAX) dst = mfma_bf16(pl[set][TM + j].p[PB_[e]], pl[set][i].p[PA_[e]], dst);
      else dst = mfma_bf16(pl[set][i].p[PA_[e]], pl[set][TM + j].p[PB_[e]], dst);
    };
    // A step = NMF MFMAs.  Part 1: the first QA MFMAs, the NCV pair conversions spread evenly behind them (about one
    // conversion = 11 vector instructions per MFMA for the 64x64 wave tile).  Part 2: the other MFMAs, one LDS read or one
    // LDS-DMA piece behind each.  The order is the program's: memory operations and the pinned splits keep it.
    constexpr int QA = (2 * NMF) / 3;                     // MFMAs that cover the conversions
    static_assert(NMF - QA >= 2, "part 2 needs MFMAs");
    // (BF3 == 2: the plane reads of the NEXT step's B operands -- ring stage st_b, step s_b -- go first, the splits of its A
    // fragments behind them)
    auto region1 = [&](int set, int cset, int st_b, int s_b) {
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        mfma3(set, q);
#pragma unroll
        for (int k = 0; k < NBR + NCV; ++k)
          if ((k * QA) / (NBR + NCV) == q) {
            if (k < NBR) read_b3(st_b, s_b, cset, k);
            else convert(cset, k - NBR);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // region 2: MFMA q (QA <= q < NMF) followed by its side operations; NRD reads first, then `ndma` LDS-DMA pieces
    auto region2 = [&](int set, int st_rd, int s_rd, bool more, int kt_dma, int st_dma, int dma0, int ndma) {
      constexpr int QB = NMF - QA;
      const int nside = NRD + ndma;
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        mfma3(set, QA + q);
#pragma unroll
        for (int k = 0; k < NRD + NDM; ++k)
          if (k < nside && (k * QB) / nside == q) {
            if (k < NRD) read_raw(st_rd, s_rd, k);
            else if (more) issue_piece(kt_dma, st_dma, dma0 + k - NRD);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    };
    // prologue: planes of step 0, raw of step 1
#pragma unroll
    for (int k = 0; k < NRD; ++k) read_raw(0, 0, k);
#pragma unroll
    for (int k = 0; k < NCV; ++k) convert(0, k);
#pragma unroll
    for (int k = 0; k < NBR; ++k) read_b3(0, 0, 0, k);
#pragma unroll
    for (int k = 0; k < NRD; ++k) read_raw(0, 1, k);
    __builtin_amdgcn_sched_barrier(0);
    auto body3 = [&](int kt, auto ST) __attribute__((always_inline)) {
      constexpr int st = decltype(ST)::value;
      constexpr int st1 = st == NS - 1 ? 0 : st + 1;
      // (The reads and splits that run ahead are NOT guarded at the end of K: in the last tile they fetch and split whatever
      // the next ring stage holds and nobody uses the result.  A guard would be a branch per side operation, and the
      // compiler merges such same-condition blocks across the MFMAs between them -- the interleave is gone.)
      const bool more = kt + NS < nkt;                  // tile kt+NS exists (requested into this tile's stage)
      // step 0: planes[0]; raw holds (kt, step 1) -> planes[1]; rendezvous; raw <- (kt+1, step 0); first half of the LDS-DMA pieces
      region1(0, 1, st, 1);
      if constexpr (BF3 == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the plane reads just issued still target this ring
#ifndef BF3_ABL_NO_BARRIER
      if (kt + NS - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NDM) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue_begin();
      region2(0, st1, 0, more, kt + NS, st, 0, NDM / 2);
      // step 1: planes[1]; raw holds (kt+1, step 0) -> planes[0]; raw <- (kt+1, step 1); the other LDS-DMA pieces
      region1(1, 0, st1, 0);
      region2(1, st1, 1, more, kt + NS, st, NDM / 2, NDM - NDM / 2);
    };
    auto ring3 = [&](int kt, bool guarded) __attribute__((always_inline)) {
      if (!guarded || kt + 0 < nkt) body3(kt + 0, std::integral_constant<int, 0>{});
      if (!guarded || kt + 1 < nkt) body3(kt + 1, std::integral_constant<int, 1>{});
      if constexpr (NS >= 3) {
        if (!guarded || kt + 2 < nkt) body3(kt + 2, std::integral_constant<int, 2>{});
      }
    };
    static_assert(NS == 2 || NS == 3, "split-bf16 loop: two- or three-stage ring");
    int kt = 0;
    for (; kt + NS <= nkt; kt += NS) ring3(kt, false);
    if (kt < nkt) ring3(kt, true);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += lo[i][j][e];
  }

#ifdef ABL_EPI_SLEEP                                       // timing-only ablation: the wave idles ~2 us (4800 cycles) before its epilogue
  __builtin_amdgcn_s_sleep(75);
#endif
#ifdef ABL_NO_EPILOGUE                                     // timing-only ablation: nothing after the K loop (the asm keeps the MFMAs alive)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::"v"(acc[i][j]));
#endif
    }
  return;
#endif
  if constexpr (AMAX) {
    // ---- fused row arg-max epilogue (vocab projection + torch.max, LanguageModel.lua:326-329) ----
    // acc[i][j] is the transposed block: this lane's row is m = m0 + wm*32*TM + i*32 + (lane&31); register e is column
    // n = n0 + wn*32*TN + j*32 + 8*(e>>2) + 4*hsel + (e&3), ascending in (j, e).  Ties: lower column (first max).
    const int an = d.amax_cols > 0 ? d.amax_n : d.N;           // real vocabulary columns
    if (d.amax_cols > 0 && n0 >= d.amax_cols) {
      // ---- columns past the arg-max prefix (the h.Wh half of the next step's gates): raw store of the transposed
      // blocks; a lane owns row m and writes 16-byte runs of 4 consecutive columns
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 32 * TM + i * 32 + r;
        if (m >= Meff) continue;
        float* crow = d.C + (size_t)m * d.ldc - d.amax_cols;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int nb = n0 + wn * 32 * TN + j * 32 + 8 * q4 + 4 * hsel;
            if (nb + 3 < d.N) {
              EPI_STORE(*reinterpret_cast<f32x4*>(crow + nb),
                        (f32x4{acc[i][j][q4 * 4 + 0], acc[i][j][q4 * 4 + 1], acc[i][j][q4 * 4 + 2], acc[i][j][q4 * 4 + 3]}));
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (nb + c < d.N) crow[nb + c] = acc[i][j][q4 * 4 + c];
            }
          }
      }
      return;
    }
    // One partial per (row, 32-column half of the tile): partial slot 2 * tile_n + wn, columns ascending with the slot.
    // Every wave finishes on its own: a compare chain inside the lane, one exchange between the two lane halves, two stores
    // -- no LDS, no workgroup barrier (the row kernel that reduces the partials reads twice as many of them: 2.6 KB a row).
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int nb = n0 + wn * 32 * TN + j * 32 + 8 * q4 + 4 * hsel;     // 4 consecutive columns, 16-byte aligned
          const f32x4 bv = amax_bias[j][q4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = nb + c < an ? acc[i][j][q4 * 4 + c] + bv[c] : -INFINITY;
            if (v > best) { best = v; bi = nb + c; }
          }
        }
      // the other lane half holds the interleaved columns of the same row
      const float ov = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      const int m = m0 + wm * 32 * TM + i * 32 + r;
      if (hsel == 0 && m < Meff) {
        const size_t slot = (size_t)m * d.amax_ld + 2 * tile_n + wn;
        EPI_STORE(d.amax_val[slot], best);
        EPI_STORE(d.amax_idx[slot], bi);
      }
    }
    return;
  }
  // Interior tiles (every row and column live, no gathered row term) leave through straight-line code: the row part of a
  // store's address is wave-uniform (scalar unit), the lane part one 32-bit offset for the whole tile, bounds tests gone.
  // The general path below tests and addresses every element on its own (~25 instructions and two branches per store).
  const bool interior = m0 + BM <= Meff && n0 + BN <= d.N;
  const int mw = m0 + wm * 32 * TM, nw = n0 + wn * 32 * TN;          // wave-uniform: this wave's first row / column
  if constexpr (CONV) {
    if (d.pool) {
      // ---- fused 2x2/2 ceil-mode max-pool: registers 4q..4q+3 of a lane are the four pixels of pool window (mb+8q)>>2
      const int Wo = (d.Wd + 1) >> 1, Ho = (d.H + 1) >> 1;
      if (interior && Wo >= 2) {
        // this lane's windows: (mw >> 2) + hsel + 2 * (4 * i + q) -- a walk in steps of two windows, one division pair
        const int per = Ho * Wo, win0 = (mw >> 2) + hsel;
        const int wi = win0 % per;
        int wy = wi / Wo, wx = wi - wy * Wo;
        float* cw = d.C + (size_t)(mw >> 2) * d.ldc + nw;
        const unsigned lane_off = (unsigned)(hsel * d.ldc + r);
        auto walk = [&](auto RELU) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bool hx = 2 * wx + 1 < d.Wd, hy = 2 * wy + 1 < d.H;
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const float bv = bias_r[j];
                float t0 = acc[i][j][4 * q] + bv, t1 = acc[i][j][4 * q + 1] + bv, t2 = acc[i][j][4 * q + 2] + bv,
                      t3 = acc[i][j][4 * q + 3] + bv;
                if constexpr (decltype(RELU)::value) {
                  t0 = t0 > 0.f ? t0 : 0.f; t1 = t1 > 0.f ? t1 : 0.f; t2 = t2 > 0.f ? t2 : 0.f; t3 = t3 > 0.f ? t3 : 0.f;
                }
                float best = t0;
                if (hx) best = t1 > best ? t1 : best;
                if (hy) best = t2 > best ? t2 : best;
                if (hx && hy) best = t3 > best ? t3 : best;
                EPI_STORE(cw[(size_t)(2 * (4 * i + q)) * d.ldc + j * 32 + lane_off], best);
              }
              wx += 2;
              if (wx >= Wo) {
                wx -= Wo;
                if (++wy >= Ho) wy = 0;
              }
            }
        };
        if (d.relu) walk(std::true_type{}); else walk(std::false_type{});
        return;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + r;
        const bool n_ok = n < d.N;
        const float bv = bias_r[j];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = mw + i * 32 + 4 * hsel;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int mrow = mb + 8 * q;
            if (n_ok && mrow < Meff)
              EPI_STORE(d.C[(size_t)(mrow >> 2) * d.ldc + n], pool_window(d, mrow >> 2, acc[i][j][4 * q], acc[i][j][4 * q + 1],
                                                                            acc[i][j][4 * q + 2], acc[i][j][4 * q + 3], bv));
          }
        }
      }
      return;
    }
  }
  // ---- epilogue: bias (+ gathered row term) + ReLU, channels-last store -------------------
  // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  if (interior && d.rowterm == nullptr) {
    float* cw = d.C + (size_t)mw * d.ldc + nw;
    const unsigned lane_off = (unsigned)(4 * hsel * d.ldc + r);
    // Round 4, last finding of the lab notes: what an epilogue costs the launch is its number of STORE instructions.  A 32x32
    // block held one column per lane leaves as 16 dword stores; passed through 4 KB of LDS that belong to this wave alone
    // (the ring stage no K-tile will use again: every wave is past the last rendezvous) it leaves as 4 stores of 16 bytes
    // per lane -- 8 full 128-byte lines each.  Same values, same addresses.
    if (d.epi_wide && (d.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(d.C) & 15) == 0) {
      float* const scr = smem + (nkt % NS) * STAGE + wid * 1024;
      auto flush_wide = [&](auto RELU) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float v = acc[i][j][e] + bias_r[j];
              if constexpr (decltype(RELU)::value) v = v > 0.f ? v : 0.f;
              scr[((e & 3) + 8 * (e >> 2) + 4 * hsel) * 32 + r] = v;
            }
            __builtin_amdgcn_wave_barrier();             // program order is LDS order inside a wave; keep the compiler to it
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(scr + (8 * q + (lane >> 3)) * 32 + (lane & 7) * 4);
              EPI_STORE(*reinterpret_cast<f32x4*>(cw + (size_t)(i * 32 + 8 * q + (lane >> 3)) * d.ldc + j * 32 + (lane & 7) * 4), v);
            }
            __builtin_amdgcn_wave_barrier();
          }
      };
      if (d.relu) flush_wide(std::true_type{}); else flush_wide(std::false_type{});
      return;
    }
    auto flush = [&](auto RELU) {
#ifdef ABL_EPI_WIDE                                        // timing-only ablation: the same bytes and lines leave as 16-byte stores
#pragma unroll                                             // (a quarter of the store instructions; values land in the wrong places)
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float t = acc[i][j][4 * q + c] + bias_r[j];
              if constexpr (decltype(RELU)::value) t = t > 0.f ? t : 0.f;
              v[c] = t;
            }
            *reinterpret_cast<f32x4*>(cw + (size_t)(i * 32 + 8 * q + (lane >> 3)) * d.ldc + j * 32 + (lane & 7) * 4) = v;
          }
      return;
#endif
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][e] + bias_r[j];
            if constexpr (decltype(RELU)::value) v = v > 0.f ? v : 0.f;
            EPI_STORE(cw[(size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * d.ldc + j * 32 + lane_off], v);
          }
    };
    if (d.relu) flush(std::true_type{}); else flush(std::false_type{});
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + r;
    const bool n_ok = n < d.N;
    const float bv = bias_r[j];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = mw + i * 32 + 4 * hsel;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (n_ok && m < Meff) {
          float v;
          if (d.rowterm != nullptr) v = d.rowterm[(size_t)(d.rowidx[m] - 1) * d.rowterm_ld + n] + acc[i][j][e];
          else v = acc[i][j][e] + bv;
          if (d.relu) v = v > 0.f ? v : 0.f;
          EPI_STORE(d.C[(size_t)m * d.ldc + n], v);
        }
      }
    }
  }
}


// Measurement hook (GemmDesc::stagger): a launch's workgroups all start within a microsecond and then march through their
// K loops in lockstep -- every CU asks for its next K-tile at the same moment.  A one-off pseudo-random pause of up to
// `stagger` x 64 cycles spreads them over a K-tile's period.  Results do not change.
__device__ __forceinline__ void start_stagger(const GemmDesc& d) {
  if (d.stagger <= 0) return;
  const unsigned n = ((blockIdx.x * 2654435761u) >> 12) % (unsigned)d.stagger;
  for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

// XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (b % 8); each XCD gets a contiguous run of logical tile ids so
// that neighbours in the fast dimension share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, o = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
}

template <int TM, int TN, bool CONV, int NS, bool AMAX = false, int BF3 = 0>
__global__ __launch_bounds__(256) void mfma_gemm_v2_kernel(GemmDesc d, int ntm, int ntn, int m_fastest) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  start_stagger(d);
  int Meff = d.M;
  if (d.m_dev != nullptr) {             // device-side row count (e.g. boxes surviving the final NMS)
    // The grid was sized for d.M rows; only the row tiles below the count are live.  Round 6: the LIVE tiles take the first
    // block ids (the tile map is rebuilt for the live row-tile count), so they start at once and spread over all CUs; the
    // blocks past them exit.  (Before, live and dead tiles alternated in id order -- with m-fastest ids one in four at 221 of
    // 1000 rows -- and the live ones landed on a fraction of the CUs: 64 us per decode step against 35 for a 221-row launch.)
    const int me = *d.m_dev;
    if (me < Meff) Meff = me;
    ntm = (Meff + 64 * TM - 1) / (64 * TM);
    if ((int)blockIdx.x >= ntm * ntn) return;
  }
  const int bid = xcd_remap(blockIdx.x, ntm * ntn);
  int tile_m, tile_n;
  if (m_fastest) { tile_m = bid % ntm; tile_n = bid / ntm; }
  else           { tile_n = bid % ntn; tile_m = bid / ntn; }
  v2_tile<TM, TN, CONV, NS, AMAX, BF3>(d, tile_m * (64 * TM), tile_n * (64 * TN), tile_n, Meff, smem);
}

// 128x64-tile launches with a FINER LAST ROUND (the decode-step GEMM, conv1_2 .. conv3_3).  The 128x64 tiles of one launch
// (decode step: 1576 at 1000 rows x 12,608 columns) do not
// fill a whole number of rounds on the chip's 2 or 3 x 256 workgroup slots; the leftover tiles used to run one per CU while the
// rest of the chip idled (the critical CU does 7 tiles against a mean of 6.16).  Here the first `nbig` tiles (whole rounds)
// are 128x64 and each leftover tile is cut into two 64x64 tiles on the SAME launch -- twice the workgroups at half the
// duration in the ragged round.  An element's K order does not depend on the tile it falls in (same fragment/lane walk
// for every v2 shape), so results are bit-identical to the plain launch.
template <bool CONV, int NS, bool AMAX, int BF3 = 0>
__global__ __launch_bounds__(256) void mfma_gemm_v2_mixed_kernel(GemmDesc d, int ntm, int ntn, int m_fastest, int nbig,
                                                                 int nwalk, int slots) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  start_stagger(d);
  int Meff = d.M;
  if (d.m_dev != nullptr) {
    // device-side row count: the tile map -- live row tiles, whole rounds, split last round -- is rebuilt for the LIVE tile
    // count with the host's rule (launch_mixed), inside the grid the host sized for d.M rows (see mfma_gemm_v2_kernel)
    const int me = *d.m_dev;
    if (me < Meff) Meff = me;
    ntm = (Meff + 127) / 128;
    const int total = ntm * ntn;
    nbig = total / slots * slots;
    int tail = total - nbig;
    if (!(nbig > 0 && tail > 0 && 4 * tail <= 3 * slots)) { nbig = total; tail = 0; }
    const int G = (int)gridDim.x;
    if (nbig + 2 * tail <= G) {
      nwalk = nbig;                                     // one workgroup per live tile
    } else {
      // fewer workgroups than live tiles: the host launched a tile WALK (dc_debug_set "walk": nwalk slots + the split round) --
      // the live tiles are walked by as many workgroups as there are, a multiple of 8 so that a workgroup stays on its XCD
      nbig = total; tail = 0;
      nwalk = G >= 8 ? (G & ~7) : G;
    }
    if ((int)blockIdx.x >= nwalk + 2 * tail) return;
  }
  const int b = blockIdx.x;
  if (b < nwalk) {
    // nwalk == nbig: one 128x64 tile per workgroup.  nwalk < nbig (tile walk, a multiple of 8): workgroup b takes tiles b,
    // b + nwalk, ... -- the same XCD every time, walking on through that XCD's run of tile ids.
    for (int t = b; t < nbig; t += nwalk) {
      if (t != b) __syncthreads();                    // every wave is done with the previous tile's LDS stages
      const int bid = xcd_remap(t, nbig);
      int tile_m, tile_n;
      if (m_fastest) { tile_m = bid % ntm; tile_n = bid / ntm; }
      else           { tile_n = bid % ntn; tile_m = bid / ntn; }
      if (tile_m * 128 >= Meff) continue;
      // a row tile with <= 64 live rows (the last 44 of a 300-proposal decode): the 64x64 variant does half the MFMAs
      if (Meff - tile_m * 128 <= 64) v2_tile<1, 1, CONV, NS, AMAX, BF3>(d, tile_m * 128, tile_n * 64, tile_n, Meff, smem);
      else v2_tile<2, 1, CONV, NS, AMAX, BF3>(d, tile_m * 128, tile_n * 64, tile_n, Meff, smem);
    }
  } else {
    const int r = b - nwalk, bid = nbig + (r >> 1), half = r & 1;
    int tile_m, tile_n;
    if (m_fastest) { tile_m = bid % ntm; tile_n = bid / ntm; }
    else           { tile_n = bid % ntn; tile_m = bid / ntn; }
    const int m0 = tile_m * 128 + half * 64;
    if (m0 >= Meff) return;
    v2_tile<1, 1, CONV, NS, AMAX, BF3>(d, m0, tile_n * 64, tile_n, Meff, smem);
  }
}

// Split-bf16 launches of 128x128 tiles: the same tile function on a two-stage ring (64 KiB), compiled for TWO workgroups per CU.
// The two accumulator sets of that mode (128 registers) put the wave at ~264 registers; two waves per SIMD need 256, which the
// compiler reaches by parking the im2col bookkeeping that is touched once per tap (a handful of spills outside the K-tile
// body).  Worth it: one wave's split (vector unit) runs under the other wave's MFMAs.
template <bool CONV, int BF3>
__global__ __launch_bounds__(256, 2) void mfma_gemm_bf3_128_kernel(GemmDesc d, int ntm, int ntn, int m_fastest) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int Meff = d.M;
  if (d.m_dev != nullptr) {             // (live tiles first: see mfma_gemm_v2_kernel)
    const int me = *d.m_dev;
    if (me < Meff) Meff = me;
    ntm = (Meff + 127) / 128;
    if ((int)blockIdx.x >= ntm * ntn) return;
  }
  const int bid = xcd_remap(blockIdx.x, ntm * ntn);
  int tile_m, tile_n;
  if (m_fastest) { tile_m = bid % ntm; tile_n = bid / ntm; }
  else           { tile_n = bid % ntn; tile_m = bid / ntn; }
  v2_tile<2, 2, CONV, 2, false, BF3>(d, tile_m * 128, tile_n * 128, tile_n, Meff, smem);
}

// =========================================================================================
// K-split variant of the 128x128 tile ("ks"): the four waves do not partition the tile, they partition K.
//
// Measured on v2 (profiles/r01_mfma_ablation.md): the fragment ds_read_b128s cost ~11 % of the loop.
// With 2x2 waves of 64x64 every wave reads (2 A + 2 B) fragments per 16 MFMAs.  Here every wave owns the
// WHOLE 128x128 tile (16 accumulator tiles = 256 registers; one wave per SIMD may use 512) for one quarter
// of each K-tile (k = 8w .. 8w+7): (4 A + 4 B) fragments per 64 MFMAs -- half the LDS->VGPR traffic for
// the same MFMA count.  The four partial tiles are summed once, after the K loop, through LDS
// (fixed order ((w0+w1)+w2)+w3), one 64x64 quadrant per wave, and leave through the usual epilogue.
// Because a tile's fragments are read during the previous tile's second half, a 2-stage ring is enough.
// =========================================================================================
// What a K segment of a 128x128 tile does with its result (stream-K, see mfma_gemm_sk_kernel): KS_NORMAL = the usual
// epilogue (or the split-K partial store); KS_PUBLISH = the summed partial tile goes to this workgroup's slot in register
// order, write-through, then its flag; KS_OWNER = the partial tiles of the `sk_npartner` workgroups after this one
// (ascending K) are added to the own result in order, then the usual epilogue.
enum { KS_NORMAL = 0, KS_PUBLISH = 1, KS_OWNER = 2 };
constexpr int SK_SLOT_FLOATS = 128 * 128;
constexpr unsigned SK_SPIN_LIMIT = 1u << 22;       // bounded: a partner that never shows up is reported, not waited for

// One K segment [kt0, kt0+nkt) (nkt even) of the tile at (m0, n0).
// HALF: the tile has at most 64 live rows (a 50-proposal batch, the last row tile of a 300-proposal one): the two upper
// 32-row blocks are neither fetched nor multiplied -- half the MFMAs of a K-tile, the weight panel streams at the same
// rate, so a <= 64-row fc6 moves from the matrix pipe's bound to HBM's.  Same K walk per element: a row's result does not
// depend on which variant its tile ran.
template <bool CONV, bool HALF = false>
__device__ __forceinline__ void ks_segment(const GemmDesc& d, const int m0, const int n0, const int Meff, const int slice,
                                           const int kt0, const int nkt, const int mode, const int sk_wg,
                                           const int sk_npartner, float* const smem) {
  constexpr int BM = 128, BN = 128;
  constexpr int PA = BM / 32, PB = BN / 32;
  constexpr int NI = HALF ? 2 : 4;         // live 32-row blocks of A
  constexpr int STAGE = (BM + BN) * BK;    // floats per ring stage
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = tid >> 3;
  const int lchunk = (tid & 7) ^ ((lrow >> 1) & 7);

  // ---- LDS-DMA descriptors / offsets: identical to v2 (8 pieces per wave per K-tile) ------------------
  __amdgpu_buffer_rsrc_t rsrcA, rsrcB;
  unsigned a_off[PA];
  int a_y[PA], a_x[PA];
  bool a_ok[PA];
  if constexpr (CONV) {
    const size_t bytes = conv_input_pixels(d, d.a_rows ? d.a_rows : d.M) * d.Cin * 4;
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
    const int hw = d.H * d.Wd;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int m = m0 + lrow + 32 * i;
      a_ok[i] = m < Meff;
      if (m >= Meff) m = Meff - 1;
      conv_pixel(d, m, hw, a_y[i], a_x[i], a_ok[i], a_off[i]);
      a_off[i] += (unsigned)lchunk * 16u;
    }
  } else {
    const float* baseA = d.A + (size_t)m0 * d.K;
    const size_t bytes = (size_t)(Meff - m0) * d.K * 4;
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int rr = lrow + 32 * i;
      if (m0 + rr >= Meff) rr = Meff - 1 - m0;
      a_off[i] = (unsigned)rr * (unsigned)d.K * 4u + (unsigned)lchunk * 16u;
      a_ok[i] = true; a_y[i] = a_x[i] = 0;
    }
  }
  unsigned b_off[PB];
  {
    const float* baseB = d.W + (size_t)n0 * d.K;
    const size_t bytes = (size_t)(d.N - n0) * d.K * 4;
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)baseB, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : bytes), 0x00020000);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int rr = lrow + 32 * i;
      if (n0 + rr >= d.N) rr = d.N - 1 - n0;
      b_off[i] = (unsigned)rr * (unsigned)d.K * 4u + (unsigned)lchunk * 16u;
    }
  }
  int tap = 0, c0 = 0;
  // im2col addressing: the per-lane byte offsets (tap displacement and the padding test folded in; CV_PAD = beyond the
  // descriptor's range -> the memory unit returns zeros) are recomputed only when the K walk enters a new tap, i.e. every
  // Cin/32 K-tiles; within a tap a K-tile only moves the scalar channel offset.  The per-piece issue is then a bare
  // buffer_load (the earlier per-piece compare/select chain cost ~3 % of the loop).
  unsigned cv_vo[PA];
  int cv_soff = 0;
  bool cv_new_tap = true;
  auto issue_begin = [&]() {
    if constexpr (CONV) {
      if (cv_new_tap) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const unsigned toff = (unsigned)((dy * d.Wd + dx) * d.Cin * 4);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const bool ok = a_ok[i] && (unsigned)(a_y[i] + dy) < (unsigned)d.H && (unsigned)(a_x[i] + dx) < (unsigned)d.Wd;
          cv_vo[i] = ok ? a_off[i] + toff : CV_PAD;
        }
      }
      cv_soff = c0 * 4;
      c0 += BK;
      cv_new_tap = c0 >= d.Cin;
      if (cv_new_tap) { c0 = 0; ++tap; }
    }
  };
  auto issue_piece = [&](int kt, int st, int p) {
    float* sa = smem + st * STAGE + (8 * wid) * BK;
    float* sb = sa + BM * BK;
    if (HALF && p >= NI && p < PA) return;              // rows 64..127 of the tile do not exist
    if (p < PA) {
      if constexpr (CONV) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, sa + p * 32 * BK, 16, (int)cv_vo[p], cv_soff, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, sa + p * 32 * BK, 16, (int)a_off[p], kt * (BK * 4), 0, 0);
      }
    } else {
      const int i = p - PA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, sb + i * 32 * BK, 16, (int)b_off[i], kt * (BK * 4), 0, 0);
    }
  };

  // ---- fragments: this wave's k-slice (logical chunks 2*wid, 2*wid+1) of all 4+4 row blocks -----------
  const int r = lane & 31, hsel = lane >> 5;
  const int foff = r * BK + (((2 * wid + hsel) ^ ((r >> 1) & 7)) << 2);
  f32x4 fa[2][4], fb[2][4];
  auto read_piece = [&](int st, int set, int p) {     // p in [0,4): A row block, [4,8): B row block
    const float* base = smem + st * STAGE + foff;
    if (HALF && p >= NI && p < 4) return;
    if (p < 4) fa[set][p] = *reinterpret_cast<const f32x4*>(base + p * 32 * BK);
    else fb[set][p - 4] = *reinterpret_cast<const f32x4*>(base + BM * BK + (p - 4) * 32 * BK);
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int NQ = 16 * NI;                          // MFMAs of one K-tile
  auto mfma_slot = [&](int set, int q) {               // q in [0,NQ): e = q/(4 NI), (i,j) = q%(4 NI)
    const int e = q / (4 * NI), i = (q >> 2) % NI, j = q & 3;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][e], fb[set][j][e], acc[i][j], 0, 0, 0);
  };

  // this segment's K range [kt0, kt0 + nkt)
  if constexpr (CONV) {
    const int cpt = d.Cin / BK;
    tap = kt0 / cpt;
    c0 = (kt0 - tap * cpt) * BK;
  }
  issue_begin();
#pragma unroll
  for (int p = 0; p < PA + PB; ++p) issue_piece(kt0, 0, p);
  if (nkt > 1) {
    issue_begin();
#pragma unroll
    for (int p = 0; p < PA + PB; ++p) issue_piece(kt0 + 1, 1, p);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 8; ++p) read_piece(0, 0, p);

  // one K-tile: 32 MFMAs, rendezvous, 32 MFMAs interleaved with the 8 LDS-DMA pieces of tile kt+2 and the
  // 8 fragment reads of tile kt+1 (HALF: 16 + 16 MFMAs, 6 pieces, 6 reads)
  constexpr int SIDE = NI + 4;                          // live LDS-DMA pieces = live fragment reads of a K-tile
  auto side_piece = [&](int sidx) { return sidx < NI ? sidx : sidx - NI + 4; };   // skip the dead A blocks
  auto tile_body = [&](int kt, int set) {
    const int st = kt & 1;
#pragma unroll
    for (int q = 0; q < NQ / 2; ++q) mfma_slot(set, q);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt+1 landed (issued one tile ago)
    __builtin_amdgcn_s_barrier();                        // ... everywhere; tile kt's stage is free (already in registers)
    __builtin_amdgcn_sched_barrier(0);
    const bool more = kt + 2 < nkt, next = kt + 1 < nkt;
    if (more) issue_begin();
#pragma unroll
    for (int q = NQ / 2; q < NQ; ++q) {
      mfma_slot(set, q);
      const int k = q - NQ / 2;                          // side operation s follows MFMA floor(s * (NQ/2) / (2 SIDE))
#pragma unroll
      for (int sdx = 0; sdx < 2 * SIDE; ++sdx) {
        if ((sdx * (NQ / 2)) / (2 * SIDE) != k) continue;
        if (sdx < SIDE) { if (more) issue_piece(kt0 + kt + 2, st, side_piece(sdx)); }
        else if (next) read_piece(st ^ 1, set ^ 1, side_piece(sdx - SIDE));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  for (int kt = 0; kt < nkt; kt += 2) {   // nkt is even (launcher guarantees K % 64 == 0): no phi over the 256 accumulators
    tile_body(kt, 0);
    tile_body(kt + 1, 1);
  }

  // ---- cross-wave reduction of the four K-partials + epilogue, one 64x64 quadrant per phase -------------
  // phase q: every wave stores its partial of quadrant q in MFMA register order (wave w -> slot w, 16 KiB);
  // then ALL waves sum the four slots (fixed order ((w0+w1)+w2)+w3): wave w' takes register group e4 = w' of
  // the quadrant's four 32x32 tiles and stores bias/row-term/ReLU results (128-byte row segments).
  float* const slots = smem;                         // [4 waves][4 tiles][4 e4][64 lanes][4]
  __amdgpu_buffer_rsrc_t rsrcP;                      // stream-K: this workgroup's partial-tile slot
  if (mode == KS_PUBLISH)
    rsrcP = __builtin_amdgcn_make_buffer_rsrc((void*)(d.sk_slots + (size_t)sk_wg * SK_SLOT_FLOATS), 0, SK_SLOT_FLOATS * 4, 0x00020000);
  if (mode == KS_OWNER) {
    // the partners' partial tiles: ONE lane polls their flags (relaxed), ONE agent acquire, then plain loads
    if (tid == 0) {
      for (int p = 1; p <= sk_npartner; ++p) {
        unsigned spins = 0;
        while (__hip_atomic_load(d.sk_flags + sk_wg + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > SK_SPIN_LIMIT) {          // the partner never published: report it (sticky word the host checks), do not hang
            if (d.sk_fault != nullptr) __hip_atomic_store(d.sk_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  // stream-K owner: the first partner's partial tile (this thread's 16 pieces of it) is requested NOW, so that it travels
  // during the LDS reduction below instead of costing one memory round trip per phase
  f32x4 pre[16];
  if (mode == KS_OWNER && sk_npartner >= 1) {
    const float* ps = d.sk_slots + (size_t)(sk_wg + 1) * SK_SLOT_FLOATS + wid * 256 + lane * 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) pre[k] = *reinterpret_cast<const f32x4*>(ps + k * 1024);
  }
#pragma unroll
  for (int q = 0; q < (HALF ? 2 : 4); ++q) {
    const int qi = (q >> 1) * 2, qj = (q & 1) * 2;   // quadrant q = accumulator tiles [qi..qi+1][qj..qj+1]
    float* slot = slots + wid * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          f32x4 v = {acc[qi + i][qj + j][e4 * 4 + 0], acc[qi + i][qj + j][e4 * 4 + 1], acc[qi + i][qj + j][e4 * 4 + 2],
                     acc[qi + i][qj + j][e4 * 4 + 3]};
          *reinterpret_cast<f32x4*>(slot + ((i * 2 + j) * 4 + e4) * 256 + lane * 4) = v;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = ((i * 2 + j) * 4 + wid) * 256 + lane * 4;
        f32x4 s0 = *reinterpret_cast<const f32x4*>(slots + off);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const f32x4 sw = *reinterpret_cast<const f32x4*>(slots + w * 4096 + off);
#pragma unroll
          for (int c = 0; c < 4; ++c) s0[c] = s0[c] + sw[c];
        }
        if (mode != KS_NORMAL) {
          const int sidx = ((q * 4 + (i * 2 + j)) * 4 + wid) * 256 + lane * 4;     // float index inside a slot
          if (mode == KS_PUBLISH) {
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const __attribute__((ext_vector_type(4))) unsigned*>(&s0), rsrcP, sidx * 4, 0, 16);   // aux 16 = sc1: write-through
            continue;
          }
          // ascending K: own (head) + partners in order; the first partner's piece was prefetched
          if (sk_npartner >= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s0[c] = s0[c] + pre[q * 4 + i * 2 + j][c];
          }
          for (int p = 2; p <= sk_npartner; ++p) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(d.sk_slots + (size_t)(sk_wg + p) * SK_SLOT_FLOATS + sidx);
#pragma unroll
            for (int c = 0; c < 4; ++c) s0[c] = s0[c] + pv[c];
          }
        }
        // registers e = 4*wid + c of tile (qi+i, qj+j): row = (e&3) + 8*(e>>2) + 4*hsel, col = lane&31
        const int col_l = (qj + j) * 32 + r;
        const int n = n0 + col_l;
        const bool n_ok = n < d.N;
        const float bv = (d.bias != nullptr && n_ok) ? d.bias[n] : 0.f;
        if constexpr (CONV) {
          if (d.pool && d.splitk == 1) {       // the four summed rows are one pool window: store its max in the pooled map
            const int mrow = m0 + (qi + i) * 32 + 8 * wid + 4 * hsel;
            if (n_ok && mrow < Meff)
              d.C[(size_t)(mrow >> 2) * d.ldc + n] = pool_window(d, mrow >> 2, s0[0], s0[1], s0[2], s0[3], bv);
            continue;
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row_l = (qi + i) * 32 + c + 8 * wid + 4 * hsel;
          const int m = m0 + row_l;
          if (d.splitk > 1) {
            if (n_ok && m < Meff) d.splitk_ws[((size_t)slice * (d.M - d.m_begin) + (m - d.m_begin)) * d.N + n] = s0[c];
          } else if (n_ok && m < Meff) {
            float v;
            if (d.rowterm != nullptr) v = d.rowterm[(size_t)(d.rowidx[m] - 1) * d.rowterm_ld + n] + s0[c];
            else v = s0[c] + bv;
            if (d.relu) v = v > 0.f ? v : 0.f;
            d.C[(size_t)m * d.ldc + n] = v;
          }
        }
      }
    __syncthreads();
  }
  if (mode == KS_PUBLISH) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_store(d.sk_flags + sk_wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool CONV>
__global__ __launch_bounds__(256) void mfma_gemm_ks_kernel(GemmDesc d, int ntm, int ntn, int m_fastest) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  start_stagger(d);
  int Meff = d.M;
  if (d.m_dev != nullptr) {             // (live tiles first: see mfma_gemm_v2_kernel)
    const int me = *d.m_dev;
    if (me < Meff) Meff = me;
    ntm = Meff > d.m_begin ? (Meff - d.m_begin + 127) / 128 : 0;
    if ((int)blockIdx.x >= ntm * ntn * d.splitk) return;
  }
  int bid = xcd_remap(blockIdx.x, ntm * ntn * d.splitk);
  const int slice = bid % d.splitk;                  // K slice of this workgroup (split-K), fastest index
  bid /= d.splitk;
  int tile_m, tile_n;
  if (m_fastest) { tile_m = bid % ntm; tile_n = bid / ntm; }
  else           { tile_n = bid % ntn; tile_m = bid / ntn; }
  const int m0 = d.m_begin + tile_m * 128, n0 = tile_n * 128;
  const int nkt = d.K / BK / d.splitk;               // this workgroup's K range (the whole K unless split-K)
  if constexpr (!CONV) {
    if (Meff - m0 <= 64) {                             // a <= 64-row tile (50-proposal batch, last tile of 300 rows)
      ks_segment<false, true>(d, m0, n0, Meff, slice, slice * nkt, nkt, KS_NORMAL, 0, 0, smem);
      return;
    }
  }
  ks_segment<CONV>(d, m0, n0, Meff, slice, slice * nkt, nkt, KS_NORMAL, 0, 0, smem);
}

// =========================================================================================
// Stream-K over the LAST, partial round of 128x128 tiles (single-image mode).  A layer whose tile count is not a multiple
// of the CU count leaves CUs idle in its last round (conv4_x: 212 tiles on 256 CUs; conv3_x: 422 = 256 + 166).  The tiles
// of that round (rows [m_begin, M), tile order n-fastest) are laid end to end as a line of K units (one unit = two K-tiles);
// the line is cut into sk_wgs equal contiguous ranges, one per workgroup (d.sk_lo).  A range covers the tail of one tile
// and/or the head of the next:
//   * a segment that starts at K = 0 OWNS its tile: it adds the partial tiles of the workgroups that hold the rest of the
//     tile's K range -- they come right after it on the line and computed their part EARLIER in their own time line, so
//     the wait is normally empty -- in ascending K order and runs the epilogue;
//   * every other segment publishes its partial tile (write-through stores, then a flag).
// A tile's sum is thus a fixed function of the problem shape: own K range first, then the following ranges in order.
// One workgroup per CU (the grid never exceeds the CU count), spins bounded (a timeout raises the sticky word d.sk_fault,
// which the host checks with the results: DC_E_HIP instead of a hang or silently wrong tiles).
// =========================================================================================
template <bool CONV>
__global__ __launch_bounds__(256) void mfma_gemm_sk_kernel(GemmDesc d, int ntn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // XCD-aware range order (round 4): consecutive ranges -- the K pieces of one tile, then the tile's n-neighbours, which
  // read the same im2col rows -- go to ONE XCD's L2 instead of eight.  Without it every XCD streamed nearly every tile's
  // operands through its 4 MiB L2 (conv4_2: 884 MB of fabric requests per launch against 37 MB of operands).  The owner /
  // partner protocol works on the LOGICAL range index w; all ranges are co-resident (grid <= CU count).
  const int w = xcd_remap(blockIdx.x, gridDim.x), np = d.sk_np;
  const int lo = d.sk_lo[w], hi = d.sk_lo[w + 1];
  int u = lo;
  while (u < hi) {
    const int t = u / np, p0 = u - t * np;
    const int p1 = min(np, p0 + (hi - u));
    const int tile_n = t % ntn, tile_m = t / ntn;
    const int m0 = d.m_begin + tile_m * 128, n0 = tile_n * 128;
    int mode = KS_NORMAL, npartner = 0;
    if (p0 > 0) mode = KS_PUBLISH;
    else if (p1 < np) {
      mode = KS_OWNER;                                  // workgroups w+1.. hold [p1, np) of this tile
      const int tile_end = (t + 1) * np;
      while (d.sk_lo[w + 1 + npartner] < tile_end) ++npartner;
    }
    ks_segment<CONV>(d, m0, n0, d.M, 0, 2 * p0, 2 * (p1 - p0), mode, w, npartner, smem);
    u += p1 - p0;
    __syncthreads();                                    // the LDS ring / slots are reused by the next segment
  }
}

}  // namespace

// compute units of the current device (256 on MI355X); cached per device
// Planning tests (tests/test_gemm_plan.py): the planners are pure functions of the problem AND the CU count.
// dc_debug_plan_gemm -- and nothing else -- may ask what a part with another CU count would be given: the override lives in
// the calling thread for the duration of that one query (advisor finding, round 4: an environment variable read here
// reached the launch paths and the stream-K co-residency guard as well).
static thread_local int g_planning_cus = 0;
void set_planning_cu_override(int cus) { g_planning_cus = cus; }
int device_cu_count() {
  if (g_planning_cus > 0) return g_planning_cus;
  static std::mutex mu;
  static std::vector<int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)cus.size() <= dev) cus.resize(dev + 1, 0);
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: a process that drives several
// devices (N contexts in N host threads, include/densecap.h) must raise it on each of them.  (device, kernel) pairs
// already raised are remembered; a host mutex makes the table safe for one ctx per thread.
hipError_t ensure_dyn_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  struct Ent { int dev; const void* fn; size_t bytes; };
  static std::mutex mu;
  static std::vector<Ent> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (Ent& t : done)
    if (t.dev == dev && t.fn == fn) {
      if (t.bytes >= bytes) return hipSuccess;
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e == hipSuccess) t.bytes = bytes;
      return e;
    }
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.push_back({dev, fn, bytes});
  return e;
}

// 128x64 launches: whole rounds as 128x64 tiles, the ragged last round as twice as many 64x64 tiles (see the mixed kernel;
// without a ragged round worth splitting every tile is a "whole round" tile -- the same kernel, so that a row tile with
// <= 64 live rows always finds its 64x64 variant).
namespace {

// Round model of a 128x64 launch, in tile times: whole rounds x workgroups per CU + the last round (split 64x64 halves count
// 1/2), over the loop efficiency measured at that co-residency (0.80 with three workgroups per CU, 0.76 with two).  Which
// ring depth wins depends on how the tile count falls into rounds (720x480 conv2_2, 1350 tiles: 1.76 rounds of 768 slots
// with an unsplit 582-tile remainder, 242 us, against 2 rounds of 512 + a split remainder, 229 us).
inline bool v2_split_tail(int full_rounds, int tail, int slots) { return full_rounds > 0 && tail > 0 && 4 * tail <= 3 * slots; }
inline double v2_cost_units(int total, int stages, int cus) {
  const int per_cu = stages == 2 ? 3 : 2, slots = per_cu * cus, full = total / slots, tail = total - full * slots;
  double units = (double)full * per_cu;
  if (tail > 0) units += v2_split_tail(full, tail, slots) ? 0.5 * ((2 * tail + cus - 1) / cus) : (double)((tail + cus - 1) / cus);
  return units / (stages == 2 ? 0.80 : 0.76);
}
// the same for 64x64 tiles (three workgroups per CU, half a 128x64 tile's work each, loop efficiency 0.72)
inline double v2_cost_units64(int total, int cus) {
  const int slots = 3 * cus, full = total / slots, tail = total - full * slots;
  return 0.5 * ((double)full * 3 + (double)((tail + cus - 1) / cus)) / 0.72;
}
inline int v2_pick_stages(int total, int cus) {
  return 2 * total >= 5 * cus && v2_cost_units(total, 2, cus) < v2_cost_units(total, 3, cus) ? 2 : 3;
}

template <bool CONV, bool AMAX, int BF3 = 0>
hipError_t launch_mixed(const GemmDesc& d, hipStream_t stream, int ntm, int ntn, int m_fastest, size_t lds) {
  // Ring depth: two stages (48 KiB) put three workgroups on a CU instead of two (72 KiB), which hides more of each tile's
  // prologue / epilogue behind its neighbours' K loops -- measured +5% on conv1_2, conv2_1 and the vocabulary projection --
  // but only once there are (nearly) three tiles for every CU: at equal co-residency the deeper ring wins (300-row
  // decode: 495 tiles, 0.87 vs 0.78 ms), so the three-stage ring keeps those launches.
  // (BF3 == 2: a stage holds the weight tile as three bf16 planes, 28 KiB for 128x64 -- three stages would leave ONE workgroup
  // per CU: always two stages there; `lds` is then the two-stage size.)
  const int total = ntm * ntn, cus = device_cu_count();
  const int stages = BF3 == 2 ? 2 : (d.stages == 2 || d.stages == 3 ? d.stages : v2_pick_stages(total, cus));
  const int wg_per_cu = BF3 == 2 ? 2 : (stages == 2 ? 3 : 2);
  const int slots = wg_per_cu * cus;
  int nbig = total / slots * slots, tail = total - nbig;
  if (!v2_split_tail(nbig / slots, tail, slots)) { nbig = total; tail = 0; }     // no ragged round worth splitting
  const int nwalk = d.walk > 0 && nbig > slots ? slots : nbig;     // measurement hook: one workgroup per slot walks its tiles
  if (stages == 2) {
    const size_t lds2 = BF3 == 2 ? lds : lds / 3 * 2;
    const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_mixed_kernel<CONV, 2, AMAX, BF3>);
    if (hipError_t e = ensure_dyn_lds(fn, lds2); e != hipSuccess) return e;
    hipLaunchKernelGGL((mfma_gemm_v2_mixed_kernel<CONV, 2, AMAX, BF3>), dim3(nwalk + 2 * tail), dim3(256), lds2, stream, d, ntm, ntn,
                       m_fastest, nbig, nwalk, slots);
    return hipGetLastError();
  }
  if constexpr (BF3 != 2) {
    const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_mixed_kernel<CONV, 3, AMAX, BF3>);
    if (hipError_t e = ensure_dyn_lds(fn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((mfma_gemm_v2_mixed_kernel<CONV, 3, AMAX, BF3>), dim3(nwalk + 2 * tail), dim3(256), lds, stream, d, ntm, ntn,
                       m_fastest, nbig, nwalk, slots);
    return hipGetLastError();
  }
  return hipErrorInvalidValue;
}

// Split-bf16 launches (GemmDesc::bf3): BF3V = 1 both operands split in registers, 2 = weight planes from HBM
template <int TM, int TN, bool CONV, int BF3V>
hipError_t launch_bf3(const GemmDesc& d, hipStream_t stream, int ntm, int ntn, int m_fastest) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr size_t stage = BF3V == 2 ? ((size_t)BM * BK + 3 * (size_t)BN * (BK / 2)) * sizeof(float) : (size_t)(BM + BN) * BK * sizeof(float);
  if constexpr (TM == 2 && TN == 2) {
    // 128x128 tiles on a two-stage ring (64 / 80 KiB of LDS): two workgroups per CU -- the split's vector work of one wave runs
    // under the MFMAs of the other
    if (d.amax_val != nullptr) return hipErrorInvalidValue;
    const void* fn = reinterpret_cast<const void*>(&mfma_gemm_bf3_128_kernel<CONV, BF3V>);
    if (hipError_t e = ensure_dyn_lds(fn, 2 * stage); e != hipSuccess) return e;
    hipLaunchKernelGGL((mfma_gemm_bf3_128_kernel<CONV, BF3V>), dim3(ntm * ntn), dim3(256), 2 * stage, stream, d, ntm, ntn, m_fastest);
    return hipGetLastError();
  } else {
    const size_t lds = (BF3V == 2 && TM == 2 ? 2 : 3) * stage;
    if (d.amax_val != nullptr) {
      if constexpr (!CONV) {
        if (d.amax_cols % BN != 0 || (d.amax_cols > 0 && (d.C == nullptr || d.amax_n > d.amax_cols || d.amax_cols > d.N)))
          return hipErrorInvalidValue;
        if constexpr (TM == 2) return launch_mixed<false, true, BF3V>(d, stream, ntm, ntn, m_fastest, lds);
        else {
          const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_kernel<1, 1, false, 3, true, BF3V>);
          if (hipError_t e = ensure_dyn_lds(fn, lds); e != hipSuccess) return e;
          hipLaunchKernelGGL((mfma_gemm_v2_kernel<1, 1, false, 3, true, BF3V>), dim3(ntm * ntn), dim3(256), lds, stream, d, ntm, ntn, m_fastest);
          return hipGetLastError();
        }
      } else {
        return hipErrorInvalidValue;
      }
    }
    if constexpr (TM == 2) return launch_mixed<CONV, false, BF3V>(d, stream, ntm, ntn, m_fastest, lds);
    else {
      const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_kernel<1, 1, CONV, 3, false, BF3V>);
      if (hipError_t e = ensure_dyn_lds(fn, lds); e != hipSuccess) return e;
      hipLaunchKernelGGL((mfma_gemm_v2_kernel<1, 1, CONV, 3, false, BF3V>), dim3(ntm * ntn), dim3(256), lds, stream, d, ntm, ntn, m_fastest);
      return hipGetLastError();
    }
  }
}

bool cfg128_uses_ks(const GemmDesc& d);

template <int TM, int TN, bool CONV>
hipError_t launch_cfg(const GemmDesc& d, hipStream_t stream) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int ntm = (d.M - d.m_begin + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
  const int m_fastest = ntm <= ntn ? 1 : 0;
  // operands are addressed through 32-bit buffer offsets
  const bool fits = CONV ? ((size_t)(d.a_rows > d.M ? d.a_rows : d.M) * d.Cin * 4 < CV_PAD && d.Cin <= 2048) : ((size_t)BM * d.K * 4 < 0xfffffff0ull);
  if (!fits || (size_t)BN * d.K * 4 >= 0xfffffff0ull) return hipErrorInvalidValue;
  if (d.bf3) {
    // ---- split-bf16 arithmetic (opt-in): the 2x2-wave kernels only, plain launches only (no K sharing between workgroups)
    if (d.splitk > 1 || d.m_begin != 0 || d.a_rows != 0) return hipErrorInvalidValue;
    // mode 2 (weights split once at load) needs the planes and 32-bit offsets over all three of them
    const bool planes = d.bf3 == 2 && d.sk_slots != nullptr && (size_t)3 * (d.sk_np > 0 ? d.sk_np : d.N) * d.K * 2 < 0xfffffff0ull;
    return planes ? launch_bf3<TM, TN, CONV, 2>(d, stream, ntm, ntn, m_fastest) : launch_bf3<TM, TN, CONV, 1>(d, stream, ntm, ntn, m_fastest);
  }
  if constexpr (TM == 2 && TN == 2) {
    if (d.amax_val != nullptr) return hipErrorInvalidValue;    // the arg-max epilogue lives in the 64-column v2 variant
    if (d.force_cfg == 5 && d.splitk == 1 && d.m_begin == 0 && d.a_rows == 0) {
      // measurement route: 128x128 tiles on the v2 kernel with a TWO-stage ring -- 64 KiB of LDS, two workgroups per CU
      const size_t lds2 = (size_t)2 * (BM + BN) * BK * sizeof(float);
      const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_kernel<2, 2, CONV, 2>);
      if (hipError_t e = ensure_dyn_lds(fn, lds2); e != hipSuccess) return e;
      hipLaunchKernelGGL((mfma_gemm_v2_kernel<2, 2, CONV, 2>), dim3(ntm * ntn), dim3(256), lds2, stream, d, ntm, ntn, m_fastest);
      return hipGetLastError();
    }
    if (cfg128_uses_ks(d)) {                                   // short K loops do not amortise the 4-phase reduction
      // 64 KiB operand ring, reused as the 4 x 16 KiB reduction slots; leaves 96 KiB of the CU's LDS to co-resident workgroups
      const size_t lds_ks = (size_t)2 * (BM + BN) * BK * sizeof(float);
      const void* fn = reinterpret_cast<const void*>(&mfma_gemm_ks_kernel<CONV>);
      if (hipError_t e = ensure_dyn_lds(fn, lds_ks); e != hipSuccess) return e;
      hipLaunchKernelGGL((mfma_gemm_ks_kernel<CONV>), dim3(ntm * ntn * d.splitk), dim3(256), lds_ks, stream, d, ntm, ntn,
                         m_fastest);
      return hipGetLastError();
    }
  }
  if (d.splitk > 1 || d.m_begin != 0 || d.a_rows != 0) return hipErrorInvalidValue;   // K-split kernel features
  if constexpr (!CONV && TN == 1) {
    if (d.amax_val != nullptr) {
      if (d.amax_cols % BN != 0 || (d.amax_cols > 0 && (d.C == nullptr || d.amax_n > d.amax_cols || d.amax_cols > d.N)))
        return hipErrorInvalidValue;
      const size_t lds3 = (size_t)3 * (BM + BN) * BK * sizeof(float);
      if constexpr (TM == 2) {
        return launch_mixed<false, true>(d, stream, ntm, ntn, m_fastest, lds3);
      } else {
        const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_kernel<TM, TN, false, 3, true>);
        if (hipError_t e = ensure_dyn_lds(fn, lds3); e != hipSuccess) return e;
        hipLaunchKernelGGL((mfma_gemm_v2_kernel<TM, TN, false, 3, true>), dim3(ntm * ntn), dim3(256), lds3, stream, d, ntm,
                           ntn, m_fastest);
        return hipGetLastError();
      }
    }
  }
  if (d.amax_val != nullptr) return hipErrorInvalidValue;
  const size_t lds = (size_t)3 * (BM + BN) * BK * sizeof(float);
  if constexpr (TM == 2 && TN == 1) {
    return launch_mixed<CONV, false>(d, stream, ntm, ntn, m_fastest, lds);
  } else {
    const void* fn = reinterpret_cast<const void*>(&mfma_gemm_v2_kernel<TM, TN, CONV, 3>);
    if (hipError_t e = ensure_dyn_lds(fn, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((mfma_gemm_v2_kernel<TM, TN, CONV, 3>), dim3(ntm * ntn), dim3(256), lds, stream, d, ntm, ntn,
                       m_fastest);
    return hipGetLastError();
  }
}

// Tile configuration of a plain launch: the largest tile that still yields >= ~1.5 workgroups per CU (256 CUs) -- for ONE
// image of a group (plan_M), with the measured exceptions below.  Pure function of the problem (mfma_gemm_plan reports it).
enum TileCfg { CFG_128x128 = 0, CFG_128x64 = 1, CFG_64x64 = 2 };
TileCfg pick_cfg(const GemmDesc& d) {
  if (d.bf3) {
    // split-bf16 launches: the largest tile with >= 1.5 workgroups per CU for ONE image (plan_M), as below; arg-max and
    // narrow outputs take 64-column tiles.  No K sharing between workgroups in this mode.
    if (d.force_cfg >= 1 && d.force_cfg <= 3 && !(d.amax_val != nullptr && d.force_cfg == 1)) return (TileCfg)(d.force_cfg - 1);
    const int pm = d.plan_M > 0 ? d.plan_M : d.M;
    auto blocks = [&](int bm, int bn) { return (long)((pm + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
    const long cus = device_cu_count();
    // 128x128 tiles from three quarters of a round on: the operand traffic per FLOP of a 128x64 tile is 1.5x a 128x128 tile's,
    // and at this mode's MFMA rate the L2 -> LDS path is what a tile waits for (measured, 1000 x 4096 x 25088: 154 vs 134 TF
    // fp32-equivalent; 6840 x 512 x 4608: 148 vs 134; few-tile problems keep the fine tiles: 1710 x 512 x 4608 84 vs 44)
    if (d.amax_val == nullptr && d.N > 64 && 4 * blocks(128, 128) >= 3 * cus) return CFG_128x128;
    return 2 * blocks(128, 64) >= 3 * cus ? CFG_128x64 : CFG_64x64;
  }
  if (d.splitk > 1) return CFG_128x128;
  if (d.force_cfg >= 1 && d.force_cfg <= 3 && !(d.amax_val != nullptr && d.force_cfg == 1)) return (TileCfg)(d.force_cfg - 1);
  if ((d.force_cfg == 5 || d.force_cfg == 6) && d.amax_val == nullptr && d.N > 64) return CFG_128x128;
  const int pm = d.plan_M > 0 ? d.plan_M : d.M;
  auto blocks = [&](int bm, int bn) { return (long)((pm + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
  const int cus = device_cu_count();
  auto fills = [&](long t) { return 2 * t >= 3 * (long)cus; };        // >= 1.5 workgroups per CU (384 tiles on MI355X)
  // fused arg-max: 128x64 tiles (two or three workgroups per CU, see launch_mixed) overlap one tile's epilogue
  // with the others' K loops -- measured 2.43 vs 2.51 ms for the 15 decode steps at 1000 x 10498 with two
  // and the same holds for every K loop too short for the K-split kernel (conv2_1: 214 -> 178 us)
  if (d.amax_val != nullptr)             // arg-max epilogue: 64-column tiles only (the partial rows are indexed by tile_n)
    return fills(blocks(128, 64)) ? CFG_128x64 : CFG_64x64;
  if (d.K < KS_MIN_KTILES * BK && d.N > 64 && fills(blocks(128, 128))) return CFG_128x64;
  if (d.N > 64 && fills(blocks(128, 128))) {
    // long K, many tiles: the K-split kernel (one workgroup per CU: rounds x (2.08 us x K-tiles + 10 us), calibrated on
    // fc6 / fc7 / conv2_2 .. conv4_2) unless its rounds fall so badly that the 128x64 kernel's finer rounds win by 5 %
    // (1080x720 conv4_3: 384 tiles = 1.5 rounds, 587 us, against 768 tiles of 128x64, 473 us measured on its 380-tile twin)
    const int nkt = d.K / BK;
    const long t128 = blocks(128, 128), t64 = blocks(128, 64);
    const double t_ks = (double)((t128 + cus - 1) / cus) * (2.08 * nkt + 10.0);
    const double t_v2 = v2_cost_units((int)t64, v2_pick_stages((int)t64, cus), cus) * 0.853 * nkt;
    return t64 < (1 << 30) && t_v2 < 0.95 * t_ks ? CFG_128x64 : CFG_128x128;
  }
  if (d.N <= 64 && fills(blocks(128, 64))) return CFG_128x64;
  if (d.N > 64 && 32 * blocks(128, 128) >= 25 * (long)cus && blocks(128, 128) <= cus) return CFG_128x128;   // 200..256 tiles: one nearly full round
  if (fills(blocks(128, 64))) {
    // a round and a half of 128x64 tiles, or finer 64x64 tiles?  (700-row fc7: 384 tiles of 128x64 on 512 slots, 244 us,
    // against 704 of 64x64, 189 us; 480x320 conv2_2: 600 vs 1200 tiles, 105 vs 102 us) -- the round model decides
    const int t64 = (int)blocks(128, 64), t6464 = (int)blocks(64, 64);
    return v2_cost_units(t64, v2_pick_stages(t64, cus), cus) <= v2_cost_units64(t6464, cus) ? CFG_128x64 : CFG_64x64;
  }
  return CFG_64x64;
}
// a 128x128 launch runs the K-split kernel when its K loop is long enough (or the caller's row window / split-K needs it)
bool cfg128_uses_ks(const GemmDesc& d) {
  const bool forced = d.splitk > 1 || d.m_begin != 0 || d.a_rows != 0 || d.force_cfg == 6;
  if (d.force_cfg == 5 && !forced) return false;
  return d.amax_val == nullptr && (d.K % (2 * BK * d.splitk)) == 0 && (forced || d.K >= KS_MIN_KTILES * BK);
}

template <bool CONV>
hipError_t launch_pick(const GemmDesc& d, hipStream_t stream) {
  switch (pick_cfg(d)) {
    case CFG_128x128: return launch_cfg<2, 2, CONV>(d, stream);
    case CFG_128x64: return launch_cfg<2, 1, CONV>(d, stream);
    default: return launch_cfg<1, 1, CONV>(d, stream);
  }
}

}  // namespace

// Split-bf16 mode is taken contraction by contraction: only where ONE image's problem (plan_M) fills the chip with 128x64
// tiles (>= 1.5 per CU).  Problems with fewer tiles live on K sharing between workgroups -- split-K, stream-K, tail plans --
// which this mode's kernels do not have, and lose there (measured: conv5_x 74 vs 82 TF, RPN conv 40 vs 69, LM encoder 51 vs 74,
// the whole 480x320 / 50-proposal frame 2.62 vs 1.98 ms): they keep the fp32 route.
bool mfma_gemm_bf3_pays(const GemmDesc& d) {
  const int pm = d.plan_M > 0 ? d.plan_M : d.M;
  const long t64 = (long)((pm + 127) / 128) * ((d.N + 63) / 64);
  return 2 * t64 >= 3 * (long)device_cu_count();
}

// Few 128x128 tiles and a long K: split K so that ~224-256 workgroups exist (one round on 256 CUs).  Every slice
// keeps an even number (>= 16) of K-tiles for the K-split kernel.
int mfma_gemm_splitk(const GemmDesc& d, size_t ws_floats) {
  if ((size_t)128 * d.K * 4 >= 0xfffffff0ull || (d.conv && (size_t)d.M * d.Cin * 4 >= CV_PAD)) return 1;
  // (a device-side row count does not change the choice -- round 6: the factor fixes the summation order, and the rows a final
  // NMS kept must get the bits they get when all P rows are decoded; row tiles past the count exit, the reduce stops at it)
  if (d.amax_val != nullptr || d.rowterm != nullptr || (d.m_dev != nullptr && d.pool)) return 1;
  const int pm = d.plan_M > 0 ? d.plan_M : d.M;       // one image's tiles: the split factor fixes the summation order
  const long tiles = (long)((pm + 127) / 128) * ((d.N + 127) / 128);
  const int nkt = d.K / BK;
  const long G = device_cu_count();                   // every planner costs its rounds on the device's CU count (256 on MI355X)
  // (problems of 128..255 tiles leave up to half the chip idle in their single round: inside the K-split kernel's domain
  // they are costed below like the smaller ones -- 500 proposals' fc6: 128 tiles x 2; outside it the 128x64 kernel has them)
  if (tiles >= G || d.N < 128 || (2 * tiles >= G && nkt < KS_MIN_KTILES)) return 1;
  // the largest factor that still leaves every workgroup an even run of K-tiles: >= 16 of them when the chip can be
  // filled that way, down to 6 for problems of a handful of tiles (480x320: conv5_x 20 tiles, RPN conv 10 -- 9 x 16
  // K-tiles used 180 / 90 CUs: 55 / 52 us; 12 x 12 and 24 x 6 fill 240: measured below)
  // every choice is a function of ONE image's problem (plan_M): a group of two images must get the factor each image gets
  // alone, so the workspace test assumes the largest group the ABI allows (dc_set_group: kGemmMaxGroup)
  auto fits = [&](int sp) { return (size_t)sp * kGemmMaxGroup * pm * d.N <= ws_floats; };
  int best = 1;
  for (int sp = 2; sp <= 32; ++sp) {
    if (tiles * sp > G) break;
    if (nkt % sp || !fits(sp)) continue;
    const int per = nkt / sp;
    if ((per & 1) || per < 6) continue;
    if (per < 16 && 4 * tiles * best >= 3 * G) continue;     // the chip is (nearly) full already: do not shorten the K runs
    best = sp;
  }
  // Several rounds of shorter K runs, when one round leaves a quarter of the chip idle behind very long runs (fc6 at 300
  // rows: 96 tiles x 2 = 192 workgroups of 392 K-tiles, 764 us; x 8 = three rounds of 98).  Cost model in us: a round is
  // its K run at the K-split kernel's 2.08 us per K-tile + ~10 us of prologue and reduction phases; the reduce launch
  // reads `sp` partial outputs at ~3 TB/s.  Taken only when it beats the one-round choice by 10 %.
  auto est = [&](int sp) {
    const long rounds = (tiles * sp + G - 1) / G;
    const double reduce = sp > 1 ? 4.0 + (double)sp * pm * d.N * 4.0 / 3.0e6 : 0.0;
    return (double)rounds * ((double)(nkt / sp) * 2.08 + 10.0) + reduce;
  };
  // (`est` describes the K-split kernel: with no one-round factor the comparison only holds if the UNSPLIT launch is that
  // kernel too -- 720x480 conv4_2, 172 tiles, runs 64x64 tiles in 212 us unsplit where 4 x 3 rounds took 253 us)
  if (best == 1) {
    GemmDesc q = d;
    q.force_cfg = 0;
    if (pick_cfg(q) != CFG_128x128 || !cfg128_uses_ks(q)) return 1;
  }
  int multi = best;
  double t_multi = est(best);
  for (int sp = best + 1; sp <= 32 && nkt >= KS_MIN_KTILES; ++sp) {
    if (tiles * sp <= G || nkt % sp) continue;
    const int per = nkt / sp;
    if ((per & 1) || per < 16 || !fits(sp)) continue;
    if (est(sp) < t_multi) { multi = sp; t_multi = est(sp); }
  }
  return t_multi <= 0.9 * est(best) ? multi : best;
}

// 4096 cycles per K-tile at ~2.1 GHz: the unit of the tail cost model below
static const double kUsPerKtile = 1.95;
bool mfma_gemm_tail_plan(const GemmDesc& d, int* m_split, int* tail_splitk) {
  if (d.amax_val != nullptr || d.rowterm != nullptr || d.m_dev != nullptr || d.N < 128 || d.N % 4) return false;
  if (d.plan_M > 0 && d.plan_M != d.M) return false;      // groups of images: the doubled tile count quantises better as it is
  if ((size_t)128 * d.K * 4 >= 0xfffffff0ull || (d.conv && (size_t)d.M * d.Cin * 4 >= CV_PAD)) return false;
  const int ntm = (d.M + 127) / 128, ntn = (d.N + 127) / 128, nkt = d.K / BK;
  // the K-split kernel's domain only: below KS_MIN_KTILES the 128x64 kernel is the faster one (conv2_1, K = 576: a tail
  // plan used to force it onto the K-split kernel in single-image mode, 205 vs 167 us)
  const long G = device_cu_count();
  if ((nkt & 1) || nkt < KS_MIN_KTILES || G % ntn) return false;
  const long T = (long)ntm * ntn;
  const long rounds = T / G, r = T % G;
  // Measured (profiles/r03_streamk_ablation.md): the plan pays only behind at least one full round and with a sizeable
  // remainder -- 844 tiles (r = 76): 351 -> 317 us, 422 tiles (r = 166): 319 -> 300 us; 300 tiles (r = 44, 480x320
  // conv2_2): 108 -> 134 us, 150 tiles and no full round (480x320 conv3_2): 118 -> 130 us
  if (rounds < 1 || 4 * r < G) return false;
  const double t_tile = nkt * kUsPerKtile;
  const double base = t_tile;                       // the partial round as whole tiles
  double best = base;
  int best_s = 1;
  for (int sp = 2; sp <= 8; ++sp) {
    if (nkt % sp) continue;
    const int per = nkt / sp;
    if ((per & 1) || per < 6) continue;
    const long units = r * sp;
    const double t = (double)((units + G - 1) / G) * per * kUsPerKtile          // K loops
                     + (double)((units + G - 1) / G) * 4.0                       // per-round prologue + 4-phase reduction
                     + (double)(sp + 1) * r * 65536.0 / 4.0e6 + 6.0;             // partial tiles through HBM + reduce launch
    if (t < best) { best = t; best_s = sp; }
  }
  if (best_s == 1 || best > 0.93 * base) return false;
  *m_split = (int)(rounds * G / ntn) * 128;         // rows covered by the full rounds (whole row-tiles: ntn | G)
  *tail_splitk = best_s;
  return true;
}

// ---- stream-K plan -----------------------------------------------------------------------------------------------------
// Rows [0, m_split) run as whole tiles in full rounds (launch_mfma_gemm_ks); the tiles of the last, partial round --
// rows [m_split, M) -- are shared evenly by `wgs` workgroups along K (mfma_gemm_sk_kernel).  Returns false when the
// problem has no partial round worth sharing.
// Measured on MI355X (profiles/r03_streamk_ablation.md): a cut tile costs ~33 us per workgroup on top of its K loop --
// the second prologue and LDS reduction (~20 us) and, at the very end of the launch where nothing hides it, the owner's
// fetch of its partner's partial tile (~11 us even with the prefetch).  So the partial round is shared along K only
// where the K loop it saves is clearly longer than that: a layer with NO full round (conv4_2 / conv4_3 at 720x600:
// 212 tiles of 144 K-tiles -> 120 K-tiles per CU, 311 -> 292 us).  Layers with full rounds before the partial one keep the
// K-split tail plan (conv3_x: 304 vs 309 us), short K loops keep whole tiles (conv4_1: 165 vs 165 us).
static const double kUsPerKtileMeasured = 2.16;   // one K-tile of the K-split kernel, one workgroup per CU, incl. its share of prologue/epilogue
static const double kSkCutUs = 45.0;              // break-even saving of a cut (33 us measured + margin)
bool mfma_gemm_sk_plan(const GemmDesc& d, int* m_split, int* wgs, int* np_out) {
  if (d.amax_val != nullptr || d.rowterm != nullptr || d.m_dev != nullptr || d.N < 128 || d.N % 4) return false;
  if (d.plan_M > 0 && d.plan_M != d.M) return false;      // groups of images: the doubled tile count quantises better as it is
  if ((size_t)128 * d.K * 4 >= 0xfffffff0ull || (d.conv && (size_t)d.M * d.Cin * 4 >= CV_PAD)) return false;
  const int G = device_cu_count();
  const int ntm = (d.M + 127) / 128, ntn = (d.N + 127) / 128, nkt = d.K / BK;
  if ((nkt & 1) || nkt < KS_MIN_KTILES || G % ntn) return false;
  const long T = (long)ntm * ntn;
  if (T < G / 2 || T >= G) return false;            // few tiles: plain split-K; full rounds first: the K-split tail plan
  const int np = nkt / 2;
  const long U = T * np;
  const long g = std::min<long>(G, U / 4);          // at least 4 units (8 K-tiles) per workgroup
  if (g < 1) return false;
  if (4 * ((U + g - 1) / g) < 3 * (long)np) return false;   // ranges shorter than 3/4 of a tile cut tiles in three: the owner then
                                                    // waits on a partner that finishes with it (480x320 conv3_2: 118 -> 128 us)
  const double saved = ((double)nkt - 2.0 * (double)((U + g - 1) / g)) * kUsPerKtileMeasured;
  if (saved < kSkCutUs) return false;
  *m_split = 0;
  *wgs = (int)g;
  *np_out = np;
  return true;
}

// device-resident unit offsets of a plan (they depend on the shape only): created once per (device, U, wgs)
static const int* sk_plan_offsets(long U, int wgs) {
  struct Ent { int dev; long U; int wgs; int* ptr; };
  static std::mutex mu;
  static std::vector<Ent> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (const Ent& e : cache)
    if (e.dev == dev && e.U == U && e.wgs == wgs) return e.ptr;
  std::vector<int> lo(wgs + 2);
  for (int w = 0; w <= wgs; ++w) lo[w] = (int)((long)w * U / wgs);
  lo[wgs + 1] = lo[wgs];                           // sentinel: the partner walk of the last workgroup stops here
  int* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), lo.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(p, lo.data(), lo.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return nullptr; }
  cache.push_back({dev, U, wgs, p});
  return p;
}

size_t mfma_gemm_sk_ws_floats(int wgs) { return (size_t)wgs * SK_SLOT_FLOATS + 64 + (size_t)wgs; }

// d: the whole problem with m_begin = m_split (rows [m_begin, M) are shared along K); ws >= mfma_gemm_sk_ws_floats(wgs)
hipError_t launch_mfma_gemm_sk(const GemmDesc& d_in, int wgs, int np, float* ws, hipStream_t stream) {
  GemmDesc d = d_in;
  if (d.splitk != 1 || d.K % (2 * BK) || wgs < 1 || wgs > device_cu_count()) return hipErrorInvalidValue;
  const int ntn = (d.N + 127) / 128, ntm = (d.M - d.m_begin + 127) / 128;
  const long U = (long)ntm * ntn * np;
  d.sk_lo = sk_plan_offsets(U, wgs);
  if (d.sk_lo == nullptr) return hipErrorOutOfMemory;
  d.sk_np = np;
  d.sk_slots = ws;
  d.sk_flags = reinterpret_cast<unsigned*>(ws + (size_t)wgs * SK_SLOT_FLOATS) + 16;     // [-1] = fault word
  if (hipError_t e = hipMemsetAsync(d.sk_flags - 16, 0, (64 + (size_t)wgs) * sizeof(unsigned) - 0, stream); e != hipSuccess) return e;
  const size_t lds_ks = (size_t)2 * (128 + 128) * BK * sizeof(float);
  if (d.conv) {
    if (d.Cin % BK != 0 || d.K != 9 * d.Cin) return hipErrorInvalidValue;
    const void* fn = reinterpret_cast<const void*>(&mfma_gemm_sk_kernel<true>);
    if (hipError_t e = ensure_dyn_lds(fn, lds_ks); e != hipSuccess) return e;
    hipLaunchKernelGGL((mfma_gemm_sk_kernel<true>), dim3(wgs), dim3(256), lds_ks, stream, d, ntn);
  } else {
    const void* fn = reinterpret_cast<const void*>(&mfma_gemm_sk_kernel<false>);
    if (hipError_t e = ensure_dyn_lds(fn, lds_ks); e != hipSuccess) return e;
    hipLaunchKernelGGL((mfma_gemm_sk_kernel<false>), dim3(wgs), dim3(256), lds_ks, stream, d, ntn);
  }
  return hipGetLastError();
}

hipError_t launch_mfma_gemm_ks(const GemmDesc& d, hipStream_t stream) {
  if (d.M - d.m_begin <= 0 || d.N <= 0 || d.K <= 0 || (d.K % (2 * BK * d.splitk)) != 0) return hipErrorInvalidValue;
  if (d.conv) {
    if (d.Cin % BK != 0 || d.K != 9 * d.Cin) return hipErrorInvalidValue;
    return launch_cfg<2, 2, true>(d, stream);
  }
  return launch_cfg<2, 2, false>(d, stream);
}

// algorithmic FLOPs (zero rows that pad the arg-max prefix to a tile boundary and pool-window slots outside the image
// do not count)
double gemm_flops(const GemmDesc& d) {
  const double n = d.amax_cols > 0 ? (double)d.amax_n + (double)(d.N - d.amax_cols) : (double)d.N;
  const double m = d.pool ? (double)(d.M / (4 * ((d.H + 1) / 2) * ((d.Wd + 1) / 2))) * (double)d.H * (double)d.Wd : (double)d.M;
  return 2.0 * m * n * (double)d.K;
}

bool mfma_gemm_can_pool(const GemmDesc& d) {
  return d.conv && (size_t)d.M * d.Cin * 4 < CV_PAD && d.Cin <= 2048 && (size_t)128 * d.K * 4 < 0xfffffff0ull &&
         d.N % 4 == 0 && d.ldc % 4 == 0;
}

// How one contraction will be carried out -- the ONE place that decides (run_gemm in densecap.hip acts on it, and
// dc_debug_plan_gemm reports it so that the policy can be pinned by tests without a GPU).
void mfma_gemm_plan(const GemmDesc& d, bool serial_mode, int tail_mode, size_t ws_floats, GemmPlan* p) {
  *p = GemmPlan();
  const bool ws_ok = ws_floats > 0 && d.ldc % 4 == 0 && d.N % 4 == 0;
  GemmDesc q = d;                                            // the launch that follows the choice (its split factor fixes the route)
  if (d.bf3) {                                               // split-bf16 mode: plain launches of the 2x2-wave kernels
    switch (pick_cfg(d)) {
      case CFG_128x128: p->route = GEMM_ROUTE_V2_128x128; p->stages = 2; break;
      case CFG_128x64: {
        const int total = ((d.M + 127) / 128) * ((d.N + 63) / 64);
        p->route = GEMM_ROUTE_V2_128x64;
        p->stages = d.stages == 2 || d.stages == 3 ? d.stages : v2_pick_stages(total, device_cu_count());
        break;
      }
      default: p->route = GEMM_ROUTE_V2_64x64;
    }
    return;
  }
  const int sp = ws_ok && d.force_cfg == 0 ? mfma_gemm_splitk(d, ws_floats) : 1;
  if (sp > 1 && (size_t)sp * d.M * d.N <= ws_floats) {
    p->kind = GEMM_PLAN_SPLITK; p->splitk = sp; q.splitk = sp;
  } else if (ws_ok && serial_mode && tail_mode == 0 && mfma_gemm_sk_plan(d, &p->m_split, &p->sk_wgs, &p->sk_np) &&
             mfma_gemm_sk_ws_floats(p->sk_wgs) <= ws_floats) {
    p->kind = GEMM_PLAN_STREAMK; p->route = GEMM_ROUTE_KS;
    return;
  } else if (ws_ok && serial_mode && tail_mode <= 1 && mfma_gemm_tail_plan(d, &p->m_split, &p->tail_splitk) &&
             (size_t)p->tail_splitk * (d.M - p->m_split) * d.N <= ws_floats) {
    p->kind = GEMM_PLAN_TAIL; p->route = GEMM_ROUTE_KS; p->sk_wgs = p->sk_np = 0;
    return;
  } else {
    p->m_split = 0; p->sk_wgs = p->sk_np = 0; p->tail_splitk = 1;
  }
  switch (pick_cfg(q)) {
    case CFG_128x128: p->route = cfg128_uses_ks(q) ? GEMM_ROUTE_KS : GEMM_ROUTE_V2_128x128; break;
    case CFG_128x64: {
      const int pm = d.M, total = ((pm + 127) / 128) * ((d.N + 63) / 64);
      p->route = GEMM_ROUTE_V2_128x64;
      p->stages = d.stages == 2 || d.stages == 3 ? d.stages : v2_pick_stages(total, device_cu_count());
      break;
    }
    default: p->route = GEMM_ROUTE_V2_64x64;
  }
}

hipError_t launch_mfma_gemm(const GemmDesc& d, hipStream_t stream) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || (d.K % BK) != 0) return hipErrorInvalidValue;
  if (d.conv) {
    if (d.Cin % BK != 0 || d.K != 9 * d.Cin) return hipErrorInvalidValue;
    return launch_pick<true>(d, stream);
  }
  return launch_pick<false>(d, stream);
}
