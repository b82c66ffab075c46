// LanguageModel:sample for FEW rows (<= 64: the webcam regime, single_machine_demo.lua:25-26 runs 50 proposals) as ONE
// persistent launch with the decode weights resident in LDS.
//
// At <= 64 rows a decode step of the GEMM route is a weight-streaming problem: [Wout; Wh^T] = 12,608 x 512 fp32 = 25.8 MB
// pass through the LDS ring of 197 workgroups for ~0.8 GFLOP of arithmetic, 16 times per image, plus two launches per
// step.  Here every workgroup loads its 64 columns of that matrix (128 KiB) into LDS ONCE and keeps them for all T+1
// steps; a step is then 64 rows x 64 columns x K = 512 of MFMA work per CU on operands that are already on chip:
//   * 165 "vocabulary" workgroups (64 vocabulary columns each) produce per-row arg-max candidates and merge them with
//     one 64-bit atomic max per (row, wave) -- key = order-preserving bits of the logit, ~column: the winner is the
//     largest logit, the LOWEST column among equals, i.e. torch.max's first maximum (LanguageModel.lua:326-329);
//   * 32 "gate" workgroups (16 hidden units x 4 gates each: their 64 weight rows are picked so that one lane ends up
//     holding i, f, o, g of the same unit) compute h.Wh, wait for the step's tokens, add the token's row of the
//     xg = b + Emb.Wx table, apply the LSTM non-linearity (cell state lives in registers for the whole launch) and
//     publish their 16 units of h_{t+1} to every workgroup.
// Two hand-offs per step (tokens: vocabulary -> gate workgroups; h: gate -> all), written per the CDNA4 rules for
// cross-XCD visibility: payload by device-scope atomics / write-through (sc1) stores, every writing wave drains, ONE lane
// bumps a counter; consumers poll that word relaxed from one lane, ONE agent acquire, then plain loads.  Spins are
// bounded: a workgroup that never shows up raises the fault word instead of hanging the GPU.
//
// Arithmetic is the GEMM route's, element for element: the same v_mfma_f32_32x32x2_f32 chain over k in the same order
// (k = 8i + 4*(lane>>5) + e, i ascending), bias and gate row term added in the same association, the same sigmoid / tanh
// -- tokens are bit-identical to the GEMM route (tests/test_gpu_e2e.py::test_persistent_decode_equals_gemm_decode).
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int PD_ROWS = 64;                 // rows one launch decodes
constexpr int PD_COLS = 64;                 // weight rows (output columns) resident per workgroup
constexpr int PD_SHARDS = 8;                // arg-max merge targets per (step, row): 165 atomics on one word cost ~5 us a step, 21 do not
constexpr unsigned PD_SPIN_LIMIT = 1u << 21;

__device__ __forceinline__ float pd_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ unsigned long long pd_key(float v, int col) {
  v = v + 0.0f;                             // -0 -> +0: the float compare of the GEMM route treats them as equal
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(~(unsigned)col);
}

// wait until *cnt >= target: ONE lane polls (relaxed, device scope), ONE agent acquire, workgroup barrier.
// Returns false when the launch has been declared faulty (here or elsewhere).
template <bool ACQUIRE = true>
__device__ __forceinline__ bool pd_wait(unsigned* cnt, unsigned target, unsigned* fault, int* s_ok) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > PD_SPIN_LIMIT || __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    if (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}
// every wave has issued its payload (atomics / sc1 stores): drain, meet, ONE lane arrives
__device__ __forceinline__ void pd_arrive(unsigned* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef PD_TRACE
#define PD_STAMP(slot) do { if (tid == 0 && (wg == 0 || gw == 0)) a.trace[((size_t)(is_vocab ? 0 : 1) * 32 + it) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PD_STAMP(slot) do { } while (0)
#endif

template <int HD>
__global__ __launch_bounds__(256) void lm_decode_persistent_kernel(LmPersistArgs a) {
  constexpr int PITCH = HD + 8;             // LDS row pitch in floats: 16 lanes of a b128 read cover all 64 banks
  constexpr int NI = HD / 8;                // 8-k groups along K
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const wl = lds;                                     // [PD_COLS][PITCH] this workgroup's weight rows
  int* const s_ok = reinterpret_cast<int*>(lds + PD_COLS * PITCH);

  const int wg = blockIdx.x;
  const bool is_vocab = wg < a.nvocab_wg;
  const int gw = wg - a.nvocab_wg;                           // gate workgroup index (16 hidden units each)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, r = lane & 31, hsel = lane >> 5;
  int n = a.n;
  if (a.n_dev != nullptr) n = min(n, *a.n_dev);
  if (n <= 0) return;
  const int T = a.T, V1 = a.V1;
  unsigned* const fault = a.sync;
  unsigned* const cnt_tok = a.sync + 16;                     // [T+2]
  unsigned* const cnt_h = a.sync + 16 + (T + 2);             // [T+2]
  unsigned long long* const best = a.best;                   // [T+1][PD_SHARDS][PD_ROWS]: one shard per XCD (block b runs on XCD b % 8)

  // ---- this workgroup's 64 weight rows -> LDS, once ------------------------------------------------------------------
  for (int idx = tid; idx < PD_COLS * (HD / 4); idx += 256) {
    const int row = idx / (HD / 4), c4 = idx - row * (HD / 4);
    size_t grow;
    if (is_vocab) {
      grow = (size_t)wg * PD_COLS + row;                     // vocabulary rows (zero rows pad V+1 to a multiple of 64)
    } else {
      // local column n = wn*32 + 8*gate + 4*hsel + c  <->  Wh^T row gate*HD + unit, unit = 16*gw + 8*wn + 4*hsel + c:
      // the 16 registers of a lane (e = 4*gate + c) are then the four gates of its four hidden units
      const int rem = row & 31, gate = rem >> 3, unit = 16 * gw + 8 * (row >> 5) + 4 * ((rem >> 2) & 1) + (rem & 3);
      grow = (size_t)a.V1pad + (size_t)gate * HD + unit;
    }
    *reinterpret_cast<f32x4*>(wl + row * PITCH + c4 * 4) = *reinterpret_cast<const f32x4*>(a.dec_w + grow * HD + c4 * 4);
  }
  const int m = wm * 32 + r;                                  // the output row this lane owns
  const bool row_live = m < n;
  const bool wave_live = wm * 32 < n;
  const int arow = min(m, n - 1);
  float bias[16];
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  const int unit0 = 16 * gw + 8 * wn + 4 * hsel;             // first of this lane's four hidden units (gate workgroups)
  if (is_vocab) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int col = wg * PD_COLS + wn * 32 + 8 * (e >> 2) + 4 * hsel + (e & 3);
      bias[e] = col < V1 ? a.out_b[col] : 0.f;
    }
  } else if (row_live) {
    const f32x4 c4v = *reinterpret_cast<const f32x4*>(a.c0 + (size_t)m * HD + unit0);
    cst[0] = c4v[0]; cst[1] = c4v[1]; cst[2] = c4v[2]; cst[3] = c4v[3];
  }
  __amdgpu_buffer_rsrc_t rsrcH =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.hbuf, 0, 2 * PD_ROWS * HD * (int)sizeof(float), 0x00020000);
  __syncthreads();

  for (int it = 0; it <= T; ++it) {
    const bool compute = is_vocab ? it >= 1 : it < T;
    if (!is_vocab && it == T && gw != 0) break;               // only gate workgroup 0 stays to write the last token
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    PD_STAMP(0);
    if (compute) {
      if (it >= 1 && !pd_wait(cnt_h + it, (unsigned)a.ngate_wg, fault, s_ok)) return;      // h of this step has landed
      PD_STAMP(1);
      if (wave_live) {
        // it == 0: h_0 comes row-major from the image step's tail kernel (a lane's 16-byte pieces are HD floats apart:
        // 32 cache lines per wave load, paid once); later steps read hbuf in FRAGMENT order -- piece (row block wm,
        // k-group i, half hsel, row r) at ((wm*NI + i)*2 + hsel)*32 + r -- so that one wave load is 1 KiB contiguous
        const float* ap;
        int astride;                                          // floats between consecutive k-groups of this lane
#ifdef PD_ABL_ROWMAJOR
        if (true) {
#else
        if (it == 0) {
#endif
          ap = (it == 0 ? a.h0 : a.hbuf + (size_t)(it & 1) * PD_ROWS * HD) + (size_t)arow * HD + 4 * hsel;
          astride = 8;
        } else {
          ap = a.hbuf + (size_t)(it & 1) * PD_ROWS * HD + ((size_t)(wm * NI) * 2 + hsel) * 32 * 4 + r * 4;
          astride = 2 * 32 * 4;
        }
        const float* bp = wl + (wn * 32 + r) * PITCH + 4 * hsel;
        // The row's K = HD operand (HD/8 16-byte pieces per lane) arrives from L2 / the fabric (it was written by other
        // CUs a moment ago): pieces are requested two chunks (2 x 16 groups = ~3.5 us of MFMAs) ahead of their use.
        constexpr int CH = 16, NCH = NI / CH;
        static_assert(NI % CH == 0 && NCH >= 2, "K chunking");
        f32x4 abuf[2][CH];
        auto load_chunk = [&](int c, int slot) {
#pragma unroll
          for (int i = 0; i < CH; ++i) abuf[slot][i] = *reinterpret_cast<const f32x4*>(ap + (size_t)astride * (c * CH + i));
        };
        auto mfma_chunk = [&](int c, int slot) {
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(bp + 8 * (c * CH + i));
#pragma unroll
#ifdef PD_ABL_NOMFMA
            for (int e = 0; e < 1; ++e) acc[e] += bf[e] * abuf[slot][i][e];
#else
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[e], abuf[slot][i][e], acc, 0, 0, 0);
#endif
          }
        };
        load_chunk(0, 0);
        load_chunk(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          mfma_chunk(c, c & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (c + 2 < NCH) load_chunk(c + 2, c & 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    PD_STAMP(2);
    if (is_vocab) {
      if (!compute) continue;
      // ---- row arg-max over this wave's 32 columns, merged across workgroups by one atomic max per (row, wave) ---------
      if (wave_live) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int col = wg * PD_COLS + wn * 32 + 8 * (e >> 2) + 4 * hsel + (e & 3);
          const float v = col < V1 ? acc[e] + bias[e] : -INFINITY;
          if (v > bv) { bv = v; bi = col; }
        }
        const float ov = __shfl_xor(bv, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
#ifdef PD_ABL_NOATOMIC
        if (hsel == 0 && row_live && bi != 0x7fffffff && wg == 0 && wn == 0)
#else
        if (hsel == 0 && row_live && bi != 0x7fffffff)
#endif
          __hip_atomic_fetch_max(best + ((size_t)it * PD_SHARDS + (wg & (PD_SHARDS - 1))) * PD_ROWS + m, pd_key(bv, bi),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      PD_STAMP(3);
      pd_arrive(cnt_tok + it);
      PD_STAMP(4);
      continue;
    }
    // ---- gate workgroups ---------------------------------------------------------------------------------------------
    int tok = V1;                                             // it == 0: the START token (LanguageModel.lua:32,320)
    if (it >= 1) {
      // every candidate is in; the keys are read with device-scope atomic loads (written by device-scope atomics): no
      // acquire fence on this hop
      if (!pd_wait<false>(cnt_tok + it, (unsigned)a.nvocab_wg, fault, s_ok)) return;
      PD_STAMP(3);
      unsigned long long k = 0ull;
#pragma unroll
      for (int sh = 0; sh < PD_SHARDS; ++sh) {
        const unsigned long long ks = __hip_atomic_load(best + ((size_t)it * PD_SHARDS + sh) * PD_ROWS + min(m, n - 1),
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        k = ks > k ? ks : k;
      }
      tok = (int)(~(unsigned)(k & 0xffffffffull)) + 1;
      if (gw == 0 && wn == 0 && hsel == 0 && row_live) a.seq[(size_t)m * T + (it - 1)] = tok;
      if (it == T) break;
    }
    PD_STAMP(4);
    if (wave_live) {
      // gates = (b + x.Wx) + h.Wh (xg row of the token first, as the tail kernel of the GEMM route), [i f o g]
      const float* x = a.xg + (size_t)(tok - 1) * 4 * HD + unit0;
      const f32x4 xi = *reinterpret_cast<const f32x4*>(x), xf = *reinterpret_cast<const f32x4*>(x + HD);
      const f32x4 xo = *reinterpret_cast<const f32x4*>(x + 2 * HD), xgg = *reinterpret_cast<const f32x4*>(x + 3 * HD);
      f32x4 hv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gi = xi[c] + acc[c], gf = xf[c] + acc[4 + c], go = xo[c] + acc[8 + c], gg = xgg[c] + acc[12 + c];
        const float ig = pd_sigmoid(gi), fg = pd_sigmoid(gf), og = pd_sigmoid(go);
        const float gt = tanhf(gg);
        const float cn = fg * cst[c] + ig * gt;
        cst[c] = cn;
        hv[c] = og * tanhf(cn);
      }
      if (row_live) {                                         // write-through (sc1): readers on other XCDs need no release fence
#ifdef PD_ABL_ROWMAJOR
        const size_t piece = ((size_t)m * HD + unit0) / 4;
#else
        const size_t piece = (((size_t)wm * NI + (unit0 >> 3)) * 2 + ((unit0 >> 2) & 1)) * 32 + r;     // fragment order
#endif
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const __attribute__((ext_vector_type(4))) unsigned*>(&hv), rsrcH,
                                               (int)(((size_t)((it + 1) & 1) * PD_ROWS * HD + piece * 4) * sizeof(float)), 0, 16);
      }
    }
    PD_STAMP(5);
    pd_arrive(cnt_h + it + 1);
    PD_STAMP(6);
  }
}

}  // namespace

size_t lm_persistent_fault_offset(int Hd, int T);
size_t lm_persistent_scratch_bytes(int Hd, int T) {
  return (size_t)2 * PD_ROWS * Hd * sizeof(float) + (size_t)(T + 1) * PD_SHARDS * PD_ROWS * sizeof(unsigned long long) +
         (size_t)(16 + 2 * (T + 2) + 16) * sizeof(unsigned) + 2 * 32 * 8 * sizeof(unsigned long long);
}

size_t lm_persistent_trace_offset(int Hd, int T) {
  return lm_persistent_fault_offset(Hd, T) + (size_t)(16 + 2 * (T + 2) + 16) * sizeof(unsigned);
}

size_t lm_persistent_fault_offset(int Hd, int T) {
  return (size_t)2 * PD_ROWS * Hd * sizeof(float) + (size_t)(T + 1) * PD_SHARDS * PD_ROWS * sizeof(unsigned long long);
}

bool lm_persistent_supported(int Hd, int V1pad, int n) {
  if (Hd != 512 || n < 1 || n > PD_ROWS || V1pad % PD_COLS) return false;
  int dev = 0, lds = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  const size_t need = (size_t)PD_COLS * (Hd + 8) * sizeof(float) + 64;
  return (size_t)lds >= need && cus >= V1pad / PD_COLS + 4 * Hd / PD_COLS;      // every workgroup on its own CU
}

hipError_t launch_lm_decode_persistent(LmPersistArgs a, int Hd, void* scratch, hipStream_t s) {
  if (!lm_persistent_supported(Hd, a.V1pad, a.n)) return hipErrorInvalidValue;
  char* p = static_cast<char*>(scratch);
  a.hbuf = reinterpret_cast<float*>(p); p += (size_t)2 * PD_ROWS * Hd * sizeof(float);
  a.best = reinterpret_cast<unsigned long long*>(p); p += (size_t)(a.T + 1) * PD_SHARDS * PD_ROWS * sizeof(unsigned long long);
  a.sync = reinterpret_cast<unsigned*>(p); p += (size_t)(16 + 2 * (a.T + 2) + 16) * sizeof(unsigned);
  a.trace = reinterpret_cast<unsigned long long*>(p);
  a.nvocab_wg = a.V1pad / PD_COLS;
  a.ngate_wg = 4 * Hd / PD_COLS;
  // every polled word starts from zero on EVERY launch (best[] is an atomic-max target: zero = below every key)
  const size_t zero_bytes = (size_t)(a.T + 1) * PD_SHARDS * PD_ROWS * sizeof(unsigned long long) + (size_t)(16 + 2 * (a.T + 2) + 16) * sizeof(unsigned);
  if (hipError_t e = hipMemsetAsync(a.best, 0, zero_bytes, s); e != hipSuccess) return e;
  const size_t lds = (size_t)PD_COLS * (Hd + 8) * sizeof(float) + 64;
  const void* fn = reinterpret_cast<const void*>(&lm_decode_persistent_kernel<512>);
  if (hipError_t e = ensure_dyn_lds(fn, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((lm_decode_persistent_kernel<512>), dim3(a.nvocab_wg + a.ngate_wg), dim3(256), lds, s, a);
  return hipGetLastError();
}
