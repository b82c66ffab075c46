/* densecap_debug.h -- measurement and test hooks of libdensecap_hip.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a reference interface and a maintainer of jcjohnson/densecap
 * binds none of it (the LuaJIT cdef in lua/densecap_hip.lua does not).  bench.py, tools/ and tests/ use these entry
 * points to time kernels, to read intermediate tensors for stage-wise parity, and to pin the contraction planner
 * without a GPU.  No setting changes a result except where a hook says it picks among deterministic fp32 summation
 * orders (tail_mode, force_cfg).
 */
#ifndef DENSECAP_DEBUG_H
#define DENSECAP_DEBUG_H

#include "densecap.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Accumulated [contraction count, total ms, total algorithmic FLOPs] of the MFMA contraction kernel family
 * (HIP events around every contraction, including its split-K finish).  reset = 1 (re)starts the
 * measurement, reset = -1 stops it, 0 just reads.  Used by bench.py for the live roofline figure. */
int dc_mfma_profile(dc_ctx* ctx, int reset, int64_t* launches, double* total_ms, double* total_flops);
/* Copy an intermediate of the most recent forward to the host for stage-wise parity:
 * name in {"feat_hwc","rpn_heads","rpn_boxes","rpn_x1y1x2y2","rpn_p","rpn_valid",
 * "rpn_nms_idx","rpn_nms_count","roi_boxes","roi_feats","codes","obj","final_trans","final_boxes",
 * "seq","final_nms_idx","final_nms_count"} (lane 0; "seq" is only filled in the reference caption order),
 * "lm_enc","lm_h","lm_c" (image encoder output and final LSTM state, P rows: row = RoI in the reference caption order,
 * row = final rank with captions after the final NMS -- rows past "survivor_rows" (int32) are then undefined), or
 * "arena_allocs" (int32: how many times a lane workspace has been (re)allocated -- it only grows), or
 * "host_enqueue_us" (int32: host microseconds per image spent enqueueing in the last dc_forward_batch).
 * Returns the number of elements copied (or <0). */
int64_t dc_debug_fetch(dc_ctx* ctx, const char* name, void* host_buf, int64_t capacity_bytes);
/* Test hooks (never needed for correct results; every setting gives the same outputs bit for bit):
 *   "beam_chunk_floats"  cap, in floats, of the beam search's full-logits buffer (default 2^28): proposals advance in
 *                        chunks of max(64, cap / (beam * (V+1))) -- lets a test walk the chunk loop with few rows;
 *   "decode_route"       0 / 1 = the GEMM decode (default), 2 = the persistent LDS-resident decode (one launch for all
 *                        T+1 LSTM steps, [Wout; Wh^T] resident in LDS) wherever it applies: greedy decode of <= 64 rows,
 *                        rnn_size 512.  Tokens are bit-identical on both routes; measured no faster.
 *   "tail_mode"          single-image mode (dc_set_lanes(1)), layers whose 128x128 tile count is not a multiple of the CU
 *                        count: 0 = per layer, whichever of stream-K over the last round / K-split tail plan / whole tiles
 *                        was measured fastest for that shape class (default), 1 = never stream-K, 2 = whole tiles only.
 *                        The routes differ in the fp32 summation order of the affected rows (each one deterministic).
 *   "force_cfg"          measurement hook for tools/route_sweep.py: 0 = planned (default), 1 / 2 / 3 = plain launches use
 *                        128x128 / 128x64 / 64x64 tiles and no split-K, 4 = planned tiles, no split-K, 5 = 128x128 tiles on the 2x2-wave kernel
 *                        with a two-stage ring (two workgroups per CU), 6 = the K-split 128x128 kernel whatever K.  Changes the fp32
 *                        summation order with the kernel family; never set by the product path.
 *   "plan_mode"          -1 (default) = contraction planning follows dc_set_lanes (1 lane = single-image planning: stream-K /
 *                        tail plans over partial last rounds); 0 / 1 force multi-lane / single-image planning whatever the lane
 *                        count -- lets a one-stream profiler pass run exactly the kernels of the multi-lane schedule.
 *   "stagger"            0 (default) .. 4096: every workgroup of a contraction launch first sleeps a pseudo-random number (below
 *                        this value) of 64-cycle periods.  Measurement only (profiles/r04_kernel_lab.md).
 *   "epi_wide"           1 (default) / 0: interior tiles of the plain epilogues leave as 16-byte stores staged through the wave's
 *                        own 4 KB of LDS (8 full lines per instruction) / as dword stores.  Same values, same addresses.
 *   "walk"               0 (default) / 1: 128x64-tile launches run one workgroup per slot that walks its tiles (same XCD, same
 *                        tile order) instead of one workgroup per tile.  Bit-identical; measurement only (no gain measured).
 *   "v2_stages"          LDS ring depth of the 128x64-tile contraction kernel: 0 = by tile count (default: two stages, three
 *                        workgroups per CU, once a launch has >= 3 tiles per CU; three stages otherwise), 2 or 3 forced.
 *                        Same K order either way: bit-identical results.
 *   "nms_band"           1 (default) / 0, process-wide: NMS windows of <= 4096 sorted rows are scanned by nms_scan_band_kernel (the
 *                        near-diagonal words of the suppression mask resident in LDS) / every window by nms_scan_kernel (one
 *                        memory round trip per 64-row chunk).  Identical picks (tests/test_gpu_ops.py); A/B measurement only.
 * Returns DC_OK or DC_E_INVALID for an unknown name / bad value. */
int dc_debug_set(dc_ctx* ctx, const char* name, int64_t value);
/* Planning query -- pure (no context, no GPU: a missing device counts as 256 CUs; the environment variable DC_PLAN_CU_COUNT,
 * read by the planners, lets a test ask what a part with another CU count would be given): how the contraction engine carries out
 * C[M,N] = A[M,K] . W[N,K]^T (conv_cin != 0: the implicit GEMM of a 3x3 convolution with that many input channels,
 * K = 9*conv_cin; argmax != 0: the vocabulary projection with its fused row arg-max; plan_M = rows of ONE image when M
 * holds a group of images, 0 = M; serial_mode = the dc_set_lanes(1) scheduling).  out8 = {kind, route, stages, splitk,
 * m_split, sk_workgroups, sk_units, tail_splitk}: kind 0 plain launch, 1 split-K + reduce, 2 stream-K over the last
 * round, 3 K-split tail plan; route 0 K-split 128x128, 1 128x64 tiles, 2 128x128 tiles, 3 64x64 tiles; stages = LDS ring
 * depth of a 128x64 launch.  Exists so that the policy (and its invariance under image groups) is pinned by tests that
 * need no GPU.  Returns DC_OK or DC_E_INVALID. */
int dc_debug_plan_gemm(int64_t M, int64_t N, int64_t K, int64_t plan_M, int conv_cin, int argmax, int serial_mode,
                       int32_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* DENSECAP_DEBUG_H */
