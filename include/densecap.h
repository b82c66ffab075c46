/* densecap.h -- C ABI of libdensecap_hip.so
 *
 * MI355X (gfx950) native replacement for the test-time hot path of
 * jcjohnson/densecap:  image -> (boxes, scores, caption tokens), i.e. what
 * `DenseCapModel:forward_test()` (densecap/DenseCapModel.lua:319-327) computes
 * when driven by `run_model.lua:64-87`.
 *
 * The reference has no FFI/plugin registry: its "operator API" is the duck-typed
 * Lua nn.Module protocol.  Each entry point below names the reference interface
 * it replaces (file:line, relative to the reference repo).  The library has no
 * Lua, Python or torch dependency: plain pointers and sizes only.  Host-side
 * mirrors of the reference classes live in lua/ (LuaJIT FFI) and densecap_amd/
 * (Python ctypes); see INTEGRATION.md.
 *
 * Conventions
 *  - all tensors fp32, row-major; token ids int32, 1-based (END = START = V+1)
 *    exactly as the reference's LongTensor `seq` (LanguageModel.lua:30-33);
 *  - box coordinates are 1-based image pixels like the reference; INDEX outputs
 *    (NMS picks) are 0-based;
 *  - every function returns DC_OK (0) or a negative DC_E_* code and never aborts;
 *    dc_last_error() returns the message (replaces Lua assert/error());
 *  - a dc_ctx is bound to one HIP device, owns its stream(s), weights and
 *    workspaces, and is NOT thread-safe (the reference is single-threaded Lua);
 *  - "dev" pointers are HIP device pointers on the ctx's device.  Per-op entry
 *    points (dc_op_*) run on the ctx's primary stream and are synchronous on
 *    return unless stated.
 */
#ifndef DENSECAP_H
#define DENSECAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK 0
#define DC_E_INVALID (-1)     /* bad argument / shape            */
#define DC_E_HIP (-2)         /* HIP runtime error               */
#define DC_E_STATE (-3)       /* call order (e.g. no weights)    */
#define DC_E_NOMEM (-4)
#define DC_E_UNSUPPORTED (-5)

#define DC_NUM_VGG_CONVS 13

typedef struct dc_ctx dc_ctx;

/* Weights in the checkpoint's (Torch7) layouts, HOST pointers, fp32.
 * Shapes follow DenseCapModel.lua:61-67,93-100, LocalizationLayer.lua:627-673,
 * LanguageModel.lua:27-61.  The library repacks them into kernel layouts. */
typedef struct dc_weights {
  const float* conv_w[DC_NUM_VGG_CONVS]; /* OIHW (Cout,Cin,3,3): VGG-16 conv1_1..conv5_3 */
  const float* conv_b[DC_NUM_VGG_CONVS]; /* (Cout)                                         */
  const float* rpn_conv_w;  /* (R,512,3,3)  R = rpn_hidden (256)                           */
  const float* rpn_conv_b;  /* (R)                                                         */
  const float* rpn_box_w;   /* (4k,R,1,1)  channel = a*4+d                                 */
  const float* rpn_box_b;   /* (4k)                                                        */
  const float* rpn_score_w; /* (2k,R,1,1)  channel = a*2+{pos,neg}                         */
  const float* rpn_score_b; /* (2k)                                                        */
  const float* fc6_w;       /* (4096, 512*7*7) input index c*49+i*7+j                      */
  const float* fc6_b;
  const float* fc7_w;       /* (4096,4096) */
  const float* fc7_b;
  const float* obj_w;       /* (1,4096)  objectness_branch */
  const float* obj_b;       /* (1) */
  const float* boxreg_w;    /* (4,4096)  box_reg_branch */
  const float* boxreg_b;    /* (4) */
  const float* lm_enc_w;    /* (E,4096)  image_encoder Linear, E = 512 */
  const float* lm_enc_b;    /* (E) */
  const float* lm_emb;      /* (V+2,E)   LookupTable */
  const float* lstm_w;      /* (E+Hd,4*Hd) torch-rnn nn.LSTM weight, gate order i,f,o,g */
  const float* lstm_b;      /* (4*Hd) */
  const float* lm_out_w;    /* (V+1,Hd) */
  const float* lm_out_b;    /* (V+1) */
  const float* anchors;     /* (2,k): row 0 widths, row 1 heights (LocalizationLayer.lua:613-619) */
  float field_centers[4];   /* x0,y0,sx,sy (net_utils.lua:106-140) = 8.5,8.5,16,16 for VGG-16 */
  int32_t num_anchors;      /* k  */
  int32_t rpn_hidden;       /* R  */
  int32_t vocab_size;       /* V  */
  int32_t seq_length;       /* T  */
  int32_t enc_size;         /* E  */
  int32_t rnn_size;         /* Hd */
  int32_t fc_dim;           /* 4096 */
} dc_weights;

/* Result of one image.  Caller owns the buffers (HOST memory) and sets
 * `capacity` >= num_proposals; the library writes K <= capacity rows.
 * Replaces the three return values of DenseCapModel:forward_test
 * (DenseCapModel.lua:319-327): final_boxes (K,4) xcycwh, objectness_scores (K,1)
 * raw logits in decreasing order, and the token matrix `seq` (K,T) that
 * LanguageModel:decodeSequence (LanguageModel.lua:86-103) turns into strings. */
typedef struct dc_result {
  int32_t capacity;  /* in  */
  int32_t K;         /* out */
  int32_t T;         /* out */
  float* boxes;      /* out (capacity,4) xc,yc,w,h */
  float* scores;     /* out (capacity)   */
  int32_t* tokens;   /* out (capacity,T) */
} dc_result;

/* ---- lifecycle ---------------------------------------------------------- */
/* utils.setup_gpus(gpu, use_cudnn) (densecap/utils.lua:22-36): bind to a device. */
int dc_create(dc_ctx** out, int hip_device);
void dc_destroy(dc_ctx* ctx);
/* Lua error()/assert message equivalent. ctx may be NULL (last global error). */
const char* dc_last_error(const dc_ctx* ctx);
/* torch.load(checkpoint).model + model:convert(dtype) (run_model.lua:146-148,
 * DenseCapModel.lua:198-208): upload + repack weights for the kernels. */
int dc_load_weights(dc_ctx* ctx, const dc_weights* w);
/* DenseCapModel:setTestArgs{rpn_nms_thresh,final_nms_thresh,num_proposals}
 * (DenseCapModel.lua:185-191). num_proposals = -1 = uncapped RPN NMS (capacity = all anchors of the image,
 * LocalizationLayer.lua:322-324); final_nms_thresh <= 0 = no final NMS (DenseCapModel.lua:261). */
int dc_set_test_args(dc_ctx* ctx, float rpn_nms_thresh, float final_nms_thresh, int num_proposals);
/* LocalizationLayer:setTestArgs{clip_boxes=, nms_thresh=, max_proposals=} (LocalizationLayer.lua:233-238), which
 * train.lua:139-142 calls directly.  clip_boxes = 0: the RPN boxes are neither clipped to the image nor masked
 * (LocalizationLayer.lua:272-300 is skipped) -- every anchor stays a candidate of the RPN NMS.
 * dc_set_test_args IS this call with clip_boxes = 1 plus the final threshold: the reference's DenseCapModel:setTestArgs
 * passes no `clip_boxes` key, so every call of it turns clipping back on. */
int dc_set_localization_test_args(dc_ctx* ctx, int clip_boxes, float nms_thresh, int max_proposals);

/* ---- the hot path ------------------------------------------------------- */
/* DenseCapModel:forward_test(input) (DenseCapModel.lua:319-327) for one image
 * (3,H,W) BGR, mean-subtracted (run_model.lua:67-74).  img_on_device != 0 means
 * `img_chw` is a device pointer (inputs resident in HBM).  Synchronous. */
int dc_forward_test(dc_ctx* ctx, const float* img_chw, int H, int W, int img_on_device, dc_result* out);
/* run_model.lua:160-180 host loop over images, n images of identical size laid out
 * back to back; images are software-pipelined over the ctx's lanes (streams). */
int dc_forward_batch(dc_ctx* ctx, const float* imgs, int n, int H, int W, int imgs_on_device,
                     dc_result* outs);
/* The same loop over images of DIFFERENT sizes (a directory of photographs: run_model.lua -input_dir): imgs[i] is
 * image i, (3, H[i], W[i]); images are pipelined over the lanes exactly like dc_forward_batch, each lane's workspace
 * growing to the largest size it meets.  Results are those of dc_forward_test on each image. */
int dc_forward_images(dc_ctx* ctx, const float* const* imgs, const int* H, const int* W, int n, int imgs_on_device,
                      dc_result* outs);
/* Number of lanes (HIP streams with private workspaces, 1..4, default 3) dc_forward_batch
 * pipelines images over.  1 = single-image mode (lowest latency for one image at a time): a layer's last
 * partial round of tiles may be shared along K by the idle CUs (a different but fixed fp32 summation order) and the decode
 * rows advance as two blocks on two streams (same arithmetic per row); per-kernel profiling (dc_mfma_profile)
 * keeps every kernel on one stream.  Results are bit-identical for a given lanes setting however images are batched. */
int dc_set_lanes(dc_ctx* ctx, int lanes);
/* Images per GROUP inside dc_forward_batch, 1 .. 8 (0 or 1 = every image on its own, the default; up to 4 through round 5): the images of a
 * group share the launches of the dense stages (the convolutions run over all of them, fc6 / fc7 and the decode over all
 * their RoI rows: fuller tile rounds, a fraction of the launches per image) and of the batched per-image kernels (RPN
 * decode, RoI pooling, gathers); the NMS runs follow each other.  Every decision that changes a sum's order (kernel
 * route, split-K factor) is planned on ONE image's problem, so an image's results do not depend on the group it travels
 * in (bit-identical, like the lane count; tests/fuzz_groups.py).  In single-image mode (dc_set_lanes(1)) the setting is
 * ignored and images travel alone: that mode shares a layer's partial last tile round along K, a plan made for one
 * image's tile count.  Measured at 720x600 / 1000 proposals: 183 images/s with groups of four on two or four lanes against
 * 182 ungrouped; 317 against 285 at 300 proposals; round 6: groups of eight on two lanes 184 against 181 with groups of four,
 * with captions after the final NMS 239-240 against 238 (a packed decode of ~1800 rows a launch). */
int dc_set_group(dc_ctx* ctx, int images);
/* Caption order. 0 (default) = the reference's order: LanguageModel:sample runs on all num_proposals
 * RoIs and the final NMS then keeps K rows (DenseCapModel.lua:127-162,261-275).  1 = run the final NMS
 * first and decode only the K surviving rows: LSTM rows are independent, so boxes, scores and tokens
 * are bit-identical, with ~K/num_proposals of the decode work. */
int dc_set_caption_order(dc_ctx* ctx, int after_final_nms);
/* Arithmetic of the dense contractions (convolutions, nn.Linear, LSTM / vocabulary products).
 *   DC_MATH_FP32 (0, the default): fp32 MFMA, v_mfma_f32_32x32x2_f32 -- an exact fp32 multiply-add chain, the arithmetic the
 *     reference computes in (DenseCapModel.lua:73-76,133) and the only mode whose results are compared bit for bit.
 *   DC_MATH_SPLIT_BF16 (1, opt-in): every fp32 operand is split, in registers, into three bf16 values that sum to it
 *     exactly; six of the nine partial products (all but those below 2^-26 of the product) are accumulated in fp32 on
 *     v_mfma_f32_32x32x16_bf16, which runs at 16x the fp32 MFMA rate -- 2.67x the matrix throughput.  Error against an fp64
 *     result is of the fp32 path's size (tests: <= 1.5x), but the bits differ: NMS / arg-max decisions that hang on the
 *     last ulp may fall the other way, as between any two fp32 summation orders.  Non-finite operands give NaN where fp32
 *     gives inf.  Inputs, outputs and everything between the contractions stay fp32; conv1_1 (3 input channels), the
 *     objectness / box-regression heads and every contraction too small to fill the chip with whole tiles (one image's
 *     conv5_x, RPN conv, LM encoder; everything at webcam sizes) stay on the fp32 path -- the mode is taken layer by
 *     layer, by a rule that depends on ONE image's problem only (results do not depend on lanes or groups).
 * May be changed between forwards; weights need no reloading. */
#define DC_MATH_FP32 0
#define DC_MATH_SPLIT_BF16 1
int dc_set_math_mode(dc_ctx* ctx, int mode);
/* Graph replay (0 = off, the default).  1: a lane that is handed the same work again -- same image size, proposal
 * capacity, group size and settings -- captures its forward once (the second time the key is seen; the first runs
 * eagerly so that every lazy allocation has happened) and relaunches it afterwards as one hipGraph: ~95 kernel launches
 * and copies of an image become one call, the gaps between dependent kernels shrink.  A scheduling knob like the lane
 * count: the kernels and their arguments are the captured ones, results are bit-identical.  Pays in the latency regime
 * (one image in flight, small proposal counts: the webcam daemon); with two or more lanes the other lane already fills
 * the gaps.  Stage times (dc_stage_times) are not available for replayed forwards; beam search and per-launch
 * profiling stay eager.  A forward that this runtime cannot capture turns replay off for the ctx: the forward still runs
 * (eagerly, DC_OK), one warning goes to stderr, and dc_debug_fetch(ctx, "graph_replay_on") reads 0 with the reason
 * left in dc_last_error. */
int dc_set_graph_replay(dc_ctx* ctx, int on);
/* LanguageModel.beam_size (LanguageModel.lua:129-131): 0 (default) = greedy LM:sample; 1..32 = LM:beamsearch
 * (LanguageModel.lua:170-290) with that many beams.  Ties in torch.topk (unspecified in the reference; they occur for
 * finished beams, whose next-word log-probabilities are zeroed) resolve to the lower index. */
int dc_set_beam_size(dc_ctx* ctx, int beam_size);
/* DenseCapModel:extractFeatures (DenseCapModel.lua:285-304): boxes (K,4) and fc7
 * codes (K,fc_dim) after the final NMS; the LSTM decode is skipped. Host outputs. */
int dc_extract_features(dc_ctx* ctx, const float* img_chw, int H, int W, int img_on_device,
                        int capacity, float* boxes, float* feats, int32_t* K);

/* extract_features.lua's loop (extract_features.lua:79-91) over n images of possibly different sizes, pipelined over
 * the lanes like dc_forward_images: image i writes K[i] rows to boxes + i*capacity*4 and feats + i*capacity*fc_dim. */
int dc_extract_features_images(dc_ctx* ctx, const float* const* imgs, const int* H, const int* W, int n,
                               int imgs_on_device, int capacity, float* boxes, float* feats, int32_t* K);

/* run_model.lua:67-74 (`run_image` before the forward) on the device: image.load's byte -> float conversion (byte / 255),
 * image.scale(img, image_size) -- torch/image's scaleBilinear: scaleLinear_rowcol along the width, then the height; linear
 * interpolation where a side grows, area averaging where it shrinks; the longer side becomes image_size --, RGB -> BGR,
 * x 255, minus the VGG mean (103.939, 116.779, 123.68).  rgb_hwc: (H0, W0, 3) bytes as a JPEG decoder delivers them, host or
 * device memory (on_device); out_chw_dev: (3, H, W) fp32 on the device with (H, W) from dc_preprocess_size -- the tensor
 * dc_forward_test / dc_forward_images take with img_on_device = 1; scaled_rgb_dev (optional, device): the scaled image as
 * (H, W, 3) bytes, what run_model.lua:184 saves for the visualiser.  Every sample is the same chain of fp32 operations as the
 * library's C loops (bit-equal to the host restatement in densecap_amd/run_model.py).  Synchronous. */
int dc_preprocess_size(int H0, int W0, int image_size, int* H, int* W);
int dc_preprocess_u8(dc_ctx* ctx, const uint8_t* rgb_hwc, int H0, int W0, int on_device, int image_size, float* out_chw_dev,
                     uint8_t* scaled_rgb_dev);

/* Per-stage GPU time of the most recent dc_forward_test on this ctx, measured with
 * HIP events on the ctx's stream (replaces LocalizationLayer:timeit,
 * LocalizationLayer.lua:219-230).  names[i] are static strings.  Returns the
 * number of stages written (<= max_stages). */
int dc_stage_times(dc_ctx* ctx, const char** names, float* ms, int max_stages);
/* ---- multi-GPU: image shards + ONE gather ------------------------------------ */
/* The reference binds one device (densecap/utils.lua:22-36) and loops over images on it
 * (run_model.lua:160-180).  Here images shard by index over ranks (one process or thread and one
 * dc_ctx per GPU, weights replicated, no data-path exchange); the only communication is one gather
 * of the per-image results on rank 0: RCCL point-to-point over xGMI, rank 0 posting world-1 receives
 * and every peer one send inside a single group.  librccl is dlopen()ed by dc_comm_create (world > 1)
 * or dc_comm_unique_id; the single-GPU path does not depend on it. */
#define DC_COMM_ID_BYTES 128
#define DC_COMM_SELF_TRANSPORT 1 /* dc_comm_create_ex flag */
typedef struct dc_comm dc_comm;
/* Rank 0 creates the 128-byte rendezvous id (ncclUniqueId) and hands it to the other ranks out of
 * band (file, environment, launcher). */
int dc_comm_unique_id(void* id_out);
/* Collective over all ranks (blocks until everyone has joined).  world == 1 needs no id and no RCCL. */
int dc_comm_create(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world);
/* The same with flags.  DC_COMM_SELF_TRANSPORT (world == 1 only; ignored otherwise): build the carrier for the single
 * rank as well -- ncclCommInitRank with one rank (id may be NULL: the library makes one) -- and route every gather through
 * the staging buffers and one ncclGroupStart / ncclRecv / ncclSend / ncclGroupEnd with rank 0 as its own peer, the calls a
 * multi-GPU gather makes.  Lets the RCCL path be executed and checked byte for byte on a one-GPU machine.
 * dc_comm_create behaves like this when the environment holds DC_COMM_FORCE_RCCL=1. */
int dc_comm_create_ex(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world, int flags);
/* What carries this communicator's gathers: "host copy" (world == 1, no carrier), "rccl", "rccl, self", "loopback",
 * "loopback, self".  Static string. */
const char* dc_comm_transport(const dc_comm* comm);
void dc_comm_destroy(dc_comm* comm);
const char* dc_comm_last_error(const dc_comm* comm);
/* Gather `n_local` results (as filled by dc_forward_test / dc_forward_batch: same capacity and T on
 * every rank) from every rank on rank 0.  gathered: rank 0 passes world*n_local caller-allocated
 * results (capacity >= the senders'); entry r*n_local + i receives image i of rank r.  Other ranks
 * pass NULL.  One record per image travels as {K, T, capacity; boxes; scores; int32 tokens}. */
int dc_gather_results(dc_comm* comm, const dc_result* local, int n_local, dc_result* gathered);

/* ---- device memory helpers (for hosts without a GPU allocator, e.g. LuaJIT) -- */
int dc_malloc(dc_ctx* ctx, void** dev_ptr, size_t bytes);
int dc_free(dc_ctx* ctx, void* dev_ptr);
int dc_memcpy_h2d(dc_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int dc_memcpy_d2h(dc_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int dc_synchronize(dc_ctx* ctx);

/* ---- per-op entry points (device pointers) ------------------------------ */
/* layout changes: the kernels keep activations channels-last (HWC). */
int dc_op_chw_to_hwc(dc_ctx* ctx, const float* in_chw, float* out_hwc, int C, int H, int W);
int dc_op_hwc_to_chw(dc_ctx* ctx, const float* in_hwc, float* out_chw, int C, int H, int W);
/* Repack OIHW (Cout,Cin,3,3) -> (Cout, 9*Cin) with k = (kh*3+kw)*Cin + c. */
int dc_op_pack_conv3x3_weights(dc_ctx* ctx, const float* w_oihw, float* w_packed, int Cout, int Cin);
/* nn.SpatialConvolution(Cin,Cout,3,3,1,1,1,1) [+ nn.ReLU] (VGG layers,
 * DenseCapModel.lua:73-76; RPN conv LocalizationLayer.lua:627-636), fp32 MFMA
 * implicit GEMM.  in: (n_img,H,W,Cin) HWC, Cin % 32 == 0; w_packed from
 * dc_op_pack_conv3x3_weights; out (n_img,H,W,Cout). */
int dc_op_conv3x3(dc_ctx* ctx, const float* in_hwc, const float* w_packed, const float* bias,
                  float* out_hwc, int n_img, int H, int W, int Cin, int Cout, int relu);
/* The same conv + ReLU followed by nn.SpatialMaxPooling(2,2,2,2):ceil() (VGG conv1_2, conv2_2, conv3_3, conv4_3 ->
 * pool1..4, DenseCapModel.lua:61-76) in ONE launch: the pool is taken in the conv's epilogue, the full-resolution
 * activation never reaches HBM.  out (ceil(H/2), ceil(W/2), Cout); bit-identical to dc_op_conv3x3 + dc_op_maxpool2x2_ceil. */
int dc_op_conv3x3_relu_pool(dc_ctx* ctx, const float* in_hwc, const float* w_packed, const float* bias,
                            float* out_hwc, int H, int W, int Cin, int Cout);
/* conv1_1: Cin = 3, reads the (3,H,W) CHW boundary image, writes (H,W,Cout) HWC. */
int dc_op_conv3x3_c3(dc_ctx* ctx, const float* in_chw, const float* w_oihw, const float* bias,
                     float* out_hwc, int H, int W, int Cout, int relu);
/* nn.SpatialMaxPooling(2,2,2,2):ceil() as loadcaffe builds it: (H,W,C)->(ceil(H/2),ceil(W/2),C). */
int dc_op_maxpool2x2_ceil(dc_ctx* ctx, const float* in_hwc, float* out_hwc, int n_img, int H, int W, int C);
/* nn.Linear [+ReLU]: C(M,N) = A(M,K) . W(N,K)^T + bias(N); K % 32 == 0. fp32 MFMA. */
int dc_op_linear(dc_ctx* ctx, const float* A, const float* W, const float* bias, float* C,
                 int M, int N, int K, int relu);
/* nn.MakeAnchors (MakeAnchors.lua:40-67) + nn.ReshapeBoxFeatures order: out (k*h*w,4). */
int dc_op_make_anchors(dc_ctx* ctx, float* out, int h, int w, float x0, float y0, float sx, float sy,
                       const float* anchors_dev /*(2,k)*/, int k);
/* nn.ApplyBoxTransform (ApplyBoxTransform.lua:63-90): (n,4),(n,4)->(n,4). */
int dc_op_apply_box_transform(dc_ctx* ctx, const float* boxes, const float* trans, float* out, int n);
/* box_utils.clip_boxes(boxes,{x_min=1,y_min=1,x_max=W,y_max=H},'xcycwh') (box_utils.lua:486-523). */
int dc_op_clip_boxes(dc_ctx* ctx, const float* boxes, float* clipped, uint8_t* valid, int n,
                     float x_min, float y_min, float x_max, float y_max);
/* box_utils.xcycwh_to_x1y1x2y2 (box_utils.lua:270-298). */
int dc_op_xcycwh_to_x1y1x2y2(dc_ctx* ctx, const float* boxes, float* out, int n);
/* nn.BoxIoU (BoxIoU.lua:40-73): (B1,4),(B2,4) xcycwh -> (B1,B2).  convention:
 * DC_IOU_BOXIOU_MODULE (0) the module as written ((w-1)/2 corners, area w*h, no +1);
 * DC_IOU_NMS_PLUS1 (1) box_utils.nms inline form (box_utils.lua:178-181,219-227: +1 on every extent);
 * DC_IOU_LEGACY_HALF_W (2) the module's original converter (BoxIoU.lua:15-37, xc -/+ w/2), the one
 * test/BoxIoU_test.lua:13-94 was written for. */
#define DC_IOU_BOXIOU_MODULE 0
#define DC_IOU_NMS_PLUS1 1
#define DC_IOU_LEGACY_HALF_W 2
int dc_op_box_iou(dc_ctx* ctx, const float* b1, const float* b2, float* out, int B1, int B2, int convention);
/* Fused LocalizationLayer._forward_test lines 265-308 after the head convs: heads (h,w,6k)
 * HWC with channels [0,4k) box (a*4+d) and [4k,6k) score (a*2+d) -> per anchor-row
 * b = a*h*w + y*w + x: boxes (clipped xcycwh), anchors, trans, x1y1x2y2, p, valid. Any out may be NULL. */
int dc_op_rpn_decode(dc_ctx* ctx, const float* heads_hwc, int h, int w, int k, const float* anchors_dev,
                     float x0, float y0, float sx, float sy, int img_h, int img_w,
                     float* boxes, float* anchors_out, float* trans, float* x1y1x2y2, float* p, uint8_t* valid);
/* box_utils.nms (box_utils.lua:154-256).  boxes (n,4) x1y1x2y2, scores (n), valid (n) or
 * NULL; max_boxes < 0 = uncapped.  Writes 0-based picks (capacity >= min(n,max_boxes))
 * in decreasing score order (ties: lower index first) and their count (device int32). */
int dc_op_nms(dc_ctx* ctx, const float* boxes, const float* scores, const uint8_t* valid, int n,
              float thresh, int max_boxes, int32_t* picks, int32_t* count);
/* nn.BilinearRoiPooling forward (BilinearRoiPooling.lua:42-60): feat (h,w,C) HWC, boxes (B,4)
 * xcycwh image px -> out.  out_layout 0: (B,C,HH,WW) as the reference; 1: (B,HH,WW,C). */
int dc_op_bilinear_roi_pool(dc_ctx* ctx, const float* feat_hwc, int h, int w, int C, const float* boxes,
                            int B, int img_h, int img_w, int HH, int WW, float* out, int out_layout);
/* LanguageModel:sample, greedy (LanguageModel.lua:293-348) with the ctx's loaded language
 * model: codes (n,fc_dim) -> tokens (n,T) int32 1-based.  With dc_set_beam_size > 0: LanguageModel:beamsearch. */
int dc_op_lm_sample(dc_ctx* ctx, const float* codes, int n, int32_t* tokens);

#ifdef __cplusplus
}
#endif
#endif /* DENSECAP_H */
