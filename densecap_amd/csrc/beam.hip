// LanguageModel:beamsearch (LanguageModel.lua:170-290) -- row kernels.  The dense steps (LSTM gates, vocabulary
// projection) reuse the MFMA engine; these kernels are the parts between them: LogSoftMax + per-beam top-k, the
// beam x beam merge, and the re-indexing of LSTM states by parent beam.
//
// The reference runs one proposal at a time with the beams in the minibatch dimension; here every proposal of a chunk
// advances together (rows = proposals x beams), which computes the same numbers row by row.
// Tie rule (torch.topk's order on equal values is unspecified): the LOWER index first, everywhere.  It matters only for
// finished beams, whose next-word log-probabilities the reference zeroes (:243-247) -- all V+1 candidates tie.
#include "common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ void arg_better(float& bv, int& bi, float v, int i) {
  if (i >= 0 && (bi < 0 || v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
}

// block-wide arg-max with lowest-index ties over values supplied per thread; returns (value, index) to all threads
__device__ void block_argmax(float& bv, int& bi, float* sv, int* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    arg_better(bv, bi, ov, oi);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
  __syncthreads();
  bv = sv[0]; bi = si[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) arg_better(bv, bi, sv[w], si[w]);
}

// One workgroup per row: nn.LogSoftMax (THNN, FloatTensor on the CPU: exp and the running sum in double,
// logsum = max + log(sum), output = float(x - logsum)), the finished-beam mask (:243-247), torch.topk(k, sorted).
// top_idx is 1-based (Lua word ids).
__global__ __launch_bounds__(256) void beam_logsoftmax_topk_kernel(const float* __restrict__ logits, int V1, int ld,
                                                                   const uint8_t* __restrict__ finished, int k,
                                                                   float* __restrict__ top_lp,
                                                                   int32_t* __restrict__ top_idx) {
  extern __shared__ float row[];            // V1 log-probabilities
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ double sd[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (finished != nullptr && finished[r]) {
    // every next-word log-probability is multiplied by 0: k zeros, indices 1..k under the lowest-index tie rule
    for (int j = tid; j < k; j += 256) { top_lp[(size_t)r * k + j] = 0.f; top_idx[(size_t)r * k + j] = j + 1; }
    return;
  }
  const float* x = logits + (size_t)r * ld;
  float mx = -INFINITY;
  for (int j = tid; j < V1; j += 256) { const float v = x[j]; row[j] = v; mx = v > mx ? v : mx; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(mx, o, 64); mx = ov > mx ? ov : mx; }
  if (lane == 0) sv[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  double sum = 0.0;
  for (int j = tid; j < V1; j += 256) sum += exp((double)(row[j] - mx));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) sd[wid] = sum;
  __syncthreads();
  const double logsum = (double)mx + log(((sd[0] + sd[1]) + sd[2]) + sd[3]);
  for (int j = tid; j < V1; j += 256) row[j] = (float)((double)row[j] - logsum);
  __syncthreads();
  for (int q = 0; q < k; ++q) {
    float bv = 0.f;
    int bi = -1;
    for (int j = tid; j < V1; j += 256) arg_better(bv, bi, row[j], j);
    block_argmax(bv, bi, sv, si);
    if (tid == 0) {
      top_lp[(size_t)r * k + q] = bv;
      top_idx[(size_t)r * k + q] = bi + 1;
      row[bi] = -INFINITY;
    }
    __syncthreads();
  }
}

// First expansion (t = 1, :207-214): one state row per proposal; beams(beam,T) filled with 1, column 1 = the top-k words.
__global__ void beam_init_kernel(const float* __restrict__ top_lp, const int32_t* __restrict__ top_idx, int nprop,
                                 int beam, int T, int END, float* __restrict__ beam_lp, int32_t* __restrict__ beams,
                                 int32_t* __restrict__ parent, int32_t* __restrict__ cur_tok,
                                 uint8_t* __restrict__ finished) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (proposal, beam)
  if (i >= nprop * beam) return;
  const int w = top_idx[i];
  beam_lp[i] = top_lp[i];
  int32_t* row = beams + (size_t)i * T;
  row[0] = w;
  for (int t = 1; t < T; ++t) row[t] = 1;
  parent[i] = 0;
  cur_tok[i] = w;
  finished[i] = (w == END) ? 1 : 0;
}

// One workgroup per proposal (:249-264): all_next = top_next_word_logprobs + beam_logprobs (beam x beam candidates),
// torch.topk(beam, sorted) over them, beams re-indexed by parent with column t set to the chosen word.
__global__ __launch_bounds__(256) void beam_merge_kernel(const float* __restrict__ top_lp,
                                                         const int32_t* __restrict__ top_idx,
                                                         const float* __restrict__ beam_lp_in,
                                                         const int32_t* __restrict__ beams_in, int beam, int T, int t,
                                                         int END, float* __restrict__ beam_lp_out,
                                                         int32_t* __restrict__ beams_out, int32_t* __restrict__ parent,
                                                         int32_t* __restrict__ cur_tok, uint8_t* __restrict__ finished) {
  __shared__ float cand[1024];
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int pick[32];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int nc = beam * beam;
  for (int j = tid; j < nc; j += 256) cand[j] = top_lp[(size_t)p * nc + j] + beam_lp_in[(size_t)p * beam + j / beam];
  __syncthreads();
  for (int q = 0; q < beam; ++q) {
    float bv = 0.f;
    int bi = -1;
    for (int j = tid; j < nc; j += 256) arg_better(bv, bi, cand[j], j);
    block_argmax(bv, bi, sv, si);
    if (tid == 0) {
      pick[q] = bi;
      beam_lp_out[(size_t)p * beam + q] = bv;
      cand[bi] = -INFINITY;
    }
    __syncthreads();
  }
  for (int q = tid; q < beam; q += 256) {
    const int b = pick[q] / beam;
    const int w = top_idx[(size_t)p * nc + pick[q]];
    const int32_t* src = beams_in + ((size_t)p * beam + b) * T;
    int32_t* dst = beams_out + ((size_t)p * beam + q) * T;
    bool fin = false;
    for (int u = 0; u < T; ++u) {
      const int v = u == t ? w : src[u];
      dst[u] = v;
      fin |= v == END;
    }
    parent[(size_t)p * beam + q] = b;
    cur_tok[(size_t)p * beam + q] = w;
    finished[(size_t)p * beam + q] = fin ? 1 : 0;   // torch.eq(beams, END):sum(2) ~= 0 (:243)
  }
}

// new state row (p, q) = old state row (p, parent[p,q]) (:266-277); src_per_prop = 1 for the first expansion
__global__ void beam_gather_state_kernel(const float* __restrict__ h_in, const float* __restrict__ c_in,
                                         const int32_t* __restrict__ parent, int rows, int beam, int src_per_prop,
                                         int Hd, float* __restrict__ h_out, float* __restrict__ c_out) {
  const size_t total = (size_t)rows * (Hd / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (Hd / 4)), j4 = (int)(i % (Hd / 4));
    const int p = r / beam;
    const size_t src = (size_t)p * src_per_prop + parent[r];
    reinterpret_cast<f32x4*>(h_out)[(size_t)r * (Hd / 4) + j4] = reinterpret_cast<const f32x4*>(h_in)[src * (Hd / 4) + j4];
    reinterpret_cast<f32x4*>(c_out)[(size_t)r * (Hd / 4) + j4] = reinterpret_cast<const f32x4*>(c_in)[src * (Hd / 4) + j4];
  }
}

// seq[p] = beams[p][argmax beam_logprobs] (:281-282); beam_logprobs come back sorted, so that is beam 0
__global__ void beam_best_kernel(const int32_t* __restrict__ beams, int nprop, int beam, int T, int32_t* __restrict__ seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nprop * T) return;
  const int p = i / T, u = i % T;
  seq[(size_t)p * T + u] = beams[((size_t)p * beam) * T + u];
}

}  // namespace

// The top-k kernel keeps one vocabulary row in dynamic LDS: V+1 floats must fit what the CURRENT device grants a
// workgroup (160 KiB on gfx950; the static reduction scratch of the kernel takes a few hundred bytes of it).
size_t beam_topk_max_vocab() {
  int dev = 0, lds = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
  return lds > 1024 ? (size_t)(lds - 1024) / sizeof(float) : 0;
}

hipError_t launch_beam_logsoftmax_topk(const float* logits, int rows, int V1, int ld, const uint8_t* finished, int k,
                                       float* top_lp, int32_t* top_idx, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const size_t lds = (size_t)V1 * sizeof(float);
  if ((size_t)V1 > beam_topk_max_vocab() || k < 1 || k > V1) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&beam_logsoftmax_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(beam_logsoftmax_topk_kernel, dim3(rows), dim3(256), lds, s, logits, V1, ld, finished, k, top_lp,
                     top_idx);
  return hipGetLastError();
}
hipError_t launch_beam_init(const float* top_lp, const int32_t* top_idx, int nprop, int beam, int T, int END,
                            float* beam_lp, int32_t* beams, int32_t* parent, int32_t* cur_tok, uint8_t* finished,
                            hipStream_t s) {
  const int n = nprop * beam;
  hipLaunchKernelGGL(beam_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, top_lp, top_idx, nprop, beam, T, END,
                     beam_lp, beams, parent, cur_tok, finished);
  return hipGetLastError();
}
hipError_t launch_beam_merge(const float* top_lp, const int32_t* top_idx, const float* beam_lp_in,
                             const int32_t* beams_in, int nprop, int beam, int T, int t, int END, float* beam_lp_out,
                             int32_t* beams_out, int32_t* parent, int32_t* cur_tok, uint8_t* finished, hipStream_t s) {
  if (beam < 1 || beam > 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(beam_merge_kernel, dim3(nprop), dim3(256), 0, s, top_lp, top_idx, beam_lp_in, beams_in, beam, T, t,
                     END, beam_lp_out, beams_out, parent, cur_tok, finished);
  return hipGetLastError();
}
hipError_t launch_beam_gather_state(const float* h_in, const float* c_in, const int32_t* parent, int rows, int beam,
                                    int src_per_prop, int Hd, float* h_out, float* c_out, hipStream_t s) {
  if (Hd % 4) return hipErrorInvalidValue;
  size_t g = ((size_t)rows * (Hd / 4) + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(beam_gather_state_kernel, dim3((unsigned)g), dim3(256), 0, s, h_in, c_in, parent, rows, beam,
                     src_per_prop, Hd, h_out, c_out);
  return hipGetLastError();
}
hipError_t launch_beam_best(const int32_t* beams, int nprop, int beam, int T, int32_t* seq, hipStream_t s) {
  const int n = nprop * T;
  hipLaunchKernelGGL(beam_best_kernel, dim3((n + 255) / 256), dim3(256), 0, s, beams, nprop, beam, T, seq);
  return hipGetLastError();
}
