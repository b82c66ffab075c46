--[[
LuaJIT FFI binding of libdensecap_hip.so (C ABI: include/densecap.h).

This is the binding a maintainer of jcjohnson/densecap adds next to
densecap/DenseCapModel.lua.  It needs only LuaJIT (already required by Torch7);
no cutorch / cunn / cudnn.  NOTE: no Lua runtime exists in the build container,
so this file is written against the header and reviewed by inspection only; the
identical ABI is exercised from Python (densecap_amd/_lib.py), which is what
the parity tests and bench.py execute.
--]]
local ffi = require 'ffi'

ffi.cdef[[
typedef struct dc_ctx dc_ctx;
typedef struct dc_weights {
  const float* conv_w[13]; const float* conv_b[13];
  const float* rpn_conv_w; const float* rpn_conv_b;
  const float* rpn_box_w;  const float* rpn_box_b;
  const float* rpn_score_w; const float* rpn_score_b;
  const float* fc6_w; const float* fc6_b; const float* fc7_w; const float* fc7_b;
  const float* obj_w; const float* obj_b; const float* boxreg_w; const float* boxreg_b;
  const float* lm_enc_w; const float* lm_enc_b; const float* lm_emb;
  const float* lstm_w; const float* lstm_b; const float* lm_out_w; const float* lm_out_b;
  const float* anchors;
  float field_centers[4];
  int32_t num_anchors, rpn_hidden, vocab_size, seq_length, enc_size, rnn_size, fc_dim;
} dc_weights;
typedef struct dc_result {
  int32_t capacity, K, T;
  float* boxes; float* scores; int32_t* tokens;
} dc_result;
int dc_create(dc_ctx** out, int hip_device);
void dc_destroy(dc_ctx* ctx);
const char* dc_last_error(const dc_ctx* ctx);
int dc_load_weights(dc_ctx* ctx, const dc_weights* w);
int dc_set_test_args(dc_ctx* ctx, float rpn_nms_thresh, float final_nms_thresh, int num_proposals);
int dc_set_localization_test_args(dc_ctx* ctx, int clip_boxes, float nms_thresh, int max_proposals);
int dc_set_lanes(dc_ctx* ctx, int lanes);
int dc_set_caption_order(dc_ctx* ctx, int after_final_nms);
int dc_set_math_mode(dc_ctx* ctx, int mode);
int dc_set_graph_replay(dc_ctx* ctx, int on);
int dc_set_beam_size(dc_ctx* ctx, int beam_size);
int dc_set_group(dc_ctx* ctx, int images);
int dc_forward_test(dc_ctx* ctx, const float* img_chw, int H, int W, int img_on_device, dc_result* out);
int dc_forward_batch(dc_ctx* ctx, const float* imgs, int n, int H, int W, int imgs_on_device, dc_result* outs);
int dc_forward_images(dc_ctx* ctx, const float* const* imgs, const int* H, const int* W, int n, int imgs_on_device,
                      dc_result* outs);
int dc_extract_features(dc_ctx* ctx, const float* img_chw, int H, int W, int img_on_device,
                        int capacity, float* boxes, float* feats, int32_t* K);
int dc_extract_features_images(dc_ctx* ctx, const float* const* imgs, const int* H, const int* W, int n,
                               int imgs_on_device, int capacity, float* boxes, float* feats, int32_t* K);
int dc_preprocess_size(int H0, int W0, int image_size, int* H, int* W);
int dc_preprocess_u8(dc_ctx* ctx, const uint8_t* rgb_hwc, int H0, int W0, int on_device, int image_size, float* out_chw_dev,
                     uint8_t* scaled_rgb_dev);
int dc_stage_times(dc_ctx* ctx, const char** names, float* ms, int max_stages);
typedef struct dc_comm dc_comm;
int dc_comm_unique_id(void* id_out);
int dc_comm_create(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world);
int dc_comm_create_ex(dc_comm** out, dc_ctx* ctx, const void* id, int rank, int world, int flags);
const char* dc_comm_transport(const dc_comm* comm);
void dc_comm_destroy(dc_comm* comm);
const char* dc_comm_last_error(const dc_comm* comm);
int dc_gather_results(dc_comm* comm, const dc_result* local, int n_local, dc_result* gathered);
]]

local M = {}
M.C = ffi.load(os.getenv('DENSECAP_HIP_LIB') or 'densecap_hip')

function M.check(ctx, rc, what)
  if rc < 0 then
    error(string.format('%s failed (%d): %s', what or 'densecap_hip', rc,
                        ffi.string(M.C.dc_last_error(ctx))))
  end
  return rc
end

return M
