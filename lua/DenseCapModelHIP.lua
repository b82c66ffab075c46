--[[
DenseCapModelHIP: drop-in for the TEST-TIME API of nn.DenseCapModel
(densecap/DenseCapModel.lua) backed by libdensecap_hip.so on an AMD MI355X.

Same method names / return values as the reference:
  model:convert(dtype, use_cudnn)            (DenseCapModel.lua:198-208)   no-op, fp32 HIP
  model:setTestArgs{rpn_nms_thresh=, final_nms_thresh=, num_proposals=}   (:185-191)
  model:evaluate()
  boxes, scores, captions = model:forward_test(img)                         (:319-327)
  boxes, feats = model:extractFeatures(img)                                 (:285-304)
  model.nets.language_model:decodeSequence(seq)                             (LanguageModel.lua:86-103)

Construction: DenseCapModelHIP.fromCheckpoint(checkpoint.model, gpu) walks the Torch7
module tree of a loaded reference checkpoint (run_model.lua:146-147) and hands the
FloatTensor storages to dc_load_weights.  Written against the header; not executable in
the build container (no LuaJIT / Torch7 there) -- see INTEGRATION.md.
--]]
local ffi = require 'ffi'
local hip = require 'densecap_hip'
local C = hip.C

local Model = {}
Model.__index = Model

-- utils.getopt (densecap/utils.lua:67-75): opt[key], or the default when the key is absent (nil)
local function getopt(opt, key, default_value)
  if default_value == nil and (opt == nil or opt[key] == nil) then
    error('error: required key ' .. key .. ' was not provided in an opt.')
  end
  if opt == nil then return default_value end
  local v = opt[key]
  if v == nil then v = default_value end
  return v
end

local function fptr(t)  -- FloatTensor -> const float*
  assert(t:type() == 'torch.FloatTensor' and t:isContiguous())
  return ffi.cast('const float*', torch.data(t))
end

-- ref: nn.DenseCapModel instance deserialised by torch.load; gpu: 0-based HIP device
function Model.fromCheckpoint(ref, gpu)
  local self = setmetatable({}, Model)
  local pctx = ffi.new('dc_ctx*[1]')
  hip.check(nil, C.dc_create(pctx, gpu or 0), 'dc_create')
  self.ctx = ffi.gc(pctx[0], C.dc_destroy)
  ref:float()
  local w = ffi.new('dc_weights')
  local keep = {}
  local function P(t) t = t:contiguous(); keep[#keep + 1] = t; return fptr(t) end
  -- VGG-16 convs: conv_net1 (layers 1-10) then conv_net2 (11-30), DenseCapModel.lua:61-76
  local convs = {}
  for _, net in ipairs{ref.nets.conv_net1, ref.nets.conv_net2} do
    for i = 1, #net do
      local m = net:get(i)
      if torch.isTypeOf(m, 'nn.SpatialConvolution') then convs[#convs + 1] = m end
    end
  end
  assert(#convs == 13, 'expected the 13 VGG-16 convolutions')
  for i, m in ipairs(convs) do
    w.conv_w[i - 1] = P(m.weight:view(m.nOutputPlane, m.nInputPlane, 3, 3))
    w.conv_b[i - 1] = P(m.bias)
  end
  -- RPN (LocalizationLayer.lua:609-690): rpn = Sequential{conv, ReLU, ConcatTable{box_branch, rpn_branch}, Flatten}
  local rpn = ref.nets.localization_layer.nets.rpn
  local rconv = rpn:get(1)
  local box_conv = rpn:get(3):get(1):get(1)
  local score_conv = rpn:get(3):get(2):get(1)
  w.rpn_conv_w, w.rpn_conv_b = P(rconv.weight), P(rconv.bias)
  w.rpn_box_w, w.rpn_box_b = P(box_conv.weight), P(box_conv.bias)
  w.rpn_score_w, w.rpn_score_b = P(score_conv.weight), P(score_conv.bias)
  local make_anchors = rpn:get(3):get(1):get(3):get(1):get(1)   -- nn.MakeAnchors
  w.anchors = P(make_anchors.anchors)
  w.field_centers[0], w.field_centers[1] = make_anchors.x0, make_anchors.y0
  w.field_centers[2], w.field_centers[3] = make_anchors.sx, make_anchors.sy
  w.num_anchors = make_anchors.anchors:size(2)
  w.rpn_hidden = rconv.nOutputPlane
  -- recog_base = VGG layers 32..38: View, fc6, ReLU, Dropout, fc7, ReLU, Dropout (DenseCapModel.lua:64,90)
  local fcs = {}
  for i = 1, #ref.nets.recog_base do
    local m = ref.nets.recog_base:get(i)
    if torch.isTypeOf(m, 'nn.Linear') then fcs[#fcs + 1] = m end
  end
  w.fc6_w, w.fc6_b, w.fc7_w, w.fc7_b = P(fcs[1].weight), P(fcs[1].bias), P(fcs[2].weight), P(fcs[2].bias)
  w.obj_w, w.obj_b = P(ref.nets.objectness_branch.weight), P(ref.nets.objectness_branch.bias)
  w.boxreg_w, w.boxreg_b = P(ref.nets.box_reg_branch.weight), P(ref.nets.box_reg_branch.bias)
  -- language model (LanguageModel.lua:27-61)
  local lm = ref.nets.language_model
  local enc = lm.image_encoder:get(1)
  w.lm_enc_w, w.lm_enc_b = P(enc.weight), P(enc.bias)
  w.lm_emb = P(lm.lookup_table.weight)
  local lstm, out
  for i = 1, #lm.rnn do
    local m = lm.rnn:get(i)
    if torch.isTypeOf(m, 'nn.LSTM') then lstm = m end
    if torch.isTypeOf(m, 'nn.Linear') then out = m end
  end
  w.lstm_w, w.lstm_b = P(lstm.weight), P(lstm.bias)
  w.lm_out_w, w.lm_out_b = P(out.weight), P(out.bias)
  w.vocab_size, w.seq_length = lm.vocab_size, lm.seq_length
  w.enc_size, w.rnn_size, w.fc_dim = lm.input_encoding_size, lm.rnn_size, fcs[2].weight:size(1)
  hip.check(self.ctx, C.dc_load_weights(self.ctx, w), 'dc_load_weights')
  keep = nil
  -- Test-time state: what the deserialised objects carry (LocalizationLayer.lua:155,233-238 writes the three layer
  -- fields, DenseCapModel.lua:31 opt.final_nms_thresh; train.lua:139-143 sets all four before torch.save) is what the
  -- model runs with until somebody calls setTestArgs.  forward reads these fields at call time, as the reference does.
  local rll = ref.nets.localization_layer
  local ll = {}
  function ll.setTestArgs(layer, args)                     -- LocalizationLayer:setTestArgs, as written: all three re-derived
    layer.test_clip_boxes = getopt(args, 'clip_boxes', true)
    layer.test_nms_thresh = getopt(args, 'nms_thresh', 0.7)
    layer.test_max_proposals = getopt(args, 'max_proposals', 300)
  end
  ll:setTestArgs()
  if rll.test_clip_boxes ~= nil then ll.test_clip_boxes = rll.test_clip_boxes end
  if rll.test_nms_thresh ~= nil then ll.test_nms_thresh = rll.test_nms_thresh end
  if rll.test_max_proposals ~= nil then ll.test_max_proposals = rll.test_max_proposals end
  self.opt = {final_nms_thresh = getopt(ref.opt, 'final_nms_thresh', 0.3)}
  self.vocab_size, self.seq_length, self.fc_dim = lm.vocab_size, lm.seq_length, fcs[2].weight:size(1)
  self.num_anchors = make_anchors.anchors:size(2)
  self.idx_to_token = lm.idx_to_token
  -- keep the reference's field layout for callers that reach into it
  -- `model.nets.language_model.beam_size = n` (LanguageModel.lua:129-131) keeps working: the assignment reaches the library
  local lm_state = {}           -- beam_size lives here so that EVERY assignment goes through __newindex
  local lm_proxy = setmetatable({decodeSequence = function(_, seq) return self:decodeSequence(seq) end}, {
    __index = lm_state,
    __newindex = function(t, k, v)
      if k == 'beam_size' then
        self:setBeamSize(v)
        lm_state.beam_size = v
      else
        rawset(t, k, v)
      end
    end})
  self.nets = {language_model = lm_proxy, localization_layer = ll}
  self:_push_test_args()
  return self
end

-- DenseCapModel:setTestArgs (DenseCapModel.lua:185-191), as written: EVERY call re-derives all three values (absent keys
-- -> 0.7 / 1000 / 0.3), unknown keys are ignored (evaluate_model.lua:39-43 passes `max_proposals=`: that caller runs
-- with 1000 proposals), and -- the layer's setTestArgs being called without `clip_boxes` -- box clipping is back on.
function Model:setTestArgs(kwargs)
  local ll = self.nets.localization_layer
  -- a value the library refuses must not stay behind in the object (every later forward would fail in _push_test_args):
  -- the previous state comes back before the error travels on
  local saved = {ll.test_clip_boxes, ll.test_nms_thresh, ll.test_max_proposals, self.opt.final_nms_thresh}
  local ok, err = pcall(function()
    self.nets.localization_layer:setTestArgs{
      nms_thresh = getopt(kwargs, 'rpn_nms_thresh', 0.7),
      max_proposals = getopt(kwargs, 'num_proposals', 1000)
    }
    self.opt.final_nms_thresh = getopt(kwargs, 'final_nms_thresh', 0.3)
    self:_push_test_args()
  end)
  if not ok then
    ll.test_clip_boxes, ll.test_nms_thresh, ll.test_max_proposals, self.opt.final_nms_thresh = unpack(saved)
    error(err, 0)
  end
end
-- the current values travel to the library before every forward (LocalizationLayer.lua:250-256 and DenseCapModel.lua:261
-- read the fields at call time; train.lua:139-143 writes them directly)
function Model:_push_test_args()
  local ll = self.nets.localization_layer
  hip.check(self.ctx, C.dc_set_test_args(self.ctx, ll.test_nms_thresh, self.opt.final_nms_thresh,
                                         ll.test_max_proposals), 'dc_set_test_args')
  if not ll.test_clip_boxes then
    hip.check(self.ctx, C.dc_set_localization_test_args(self.ctx, 0, ll.test_nms_thresh, ll.test_max_proposals),
              'dc_set_localization_test_args')
  end
end
-- rows the result buffers need: num_proposals, or every anchor of the image when it is -1 (uncapped RPN NMS,
-- LocalizationLayer.lua:322-324): k * ceil(H/16) * ceil(W/16) after the four ceil-mode pools
function Model:_capacity(H, W)
  local P = self.nets.localization_layer.test_max_proposals
  if P ~= -1 then return P end
  for _ = 1, 4 do H, W = math.floor((H + 1) / 2), math.floor((W + 1) / 2) end
  return self.num_anchors * H * W
end
-- language_model.beam_size: nil / 0 = greedy LM:sample, n = LM:beamsearch with n beams (1..32)
function Model:setBeamSize(n)
  hip.check(self.ctx, C.dc_set_beam_size(self.ctx, n or 0), 'dc_set_beam_size')
end
-- repeated forwards of one image size relaunched as a captured hipGraph (same results; the webcam daemon's regime)
function Model:setGraphReplay(on)
  hip.check(self.ctx, C.dc_set_graph_replay(self.ctx, on and 1 or 0), 'dc_set_graph_replay')
  return self
end
-- caption order (include/densecap.h: dc_set_caption_order): false = the reference's (LM:sample on all num_proposals rows, then
-- the final NMS, DenseCapModel.lua:127-162,261-275), true = the final NMS first and ONE packed decode of the rows it keeps --
-- the same boxes, scores and tokens bit for bit, about a quarter of the decode work at 1000 proposals
function Model:setCaptionOrder(after_final_nms)
  hip.check(self.ctx, C.dc_set_caption_order(self.ctx, after_final_nms and 1 or 0), 'dc_set_caption_order')
  return self
end
-- arithmetic of the large contractions: 0 = fp32 MFMA (default, the reference's arithmetic), 1 = split-bf16 (opt-in; include/densecap.h)
function Model:setMathMode(mode)
  hip.check(self.ctx, C.dc_set_math_mode(self.ctx, mode or 0), 'dc_set_math_mode')
  return self
end
-- run_model.lua:67-74 on the device: `img` = ByteTensor (H0, W0, 3) RGB as a decoder delivers it -> device pointer of the
-- (3, H, W) float tensor forward_test_device takes, plus H, W.  (image.load + image.scale + BGR, x255, mean; bit-equal to the
-- library's own C loops.)  The caller frees the pointer with C.dc_free(self.ctx, ptr).
function Model:preprocess(img_hwc_bytes, image_size)
  assert(torch.type(img_hwc_bytes) == 'torch.ByteTensor' and img_hwc_bytes:dim() == 3 and img_hwc_bytes:size(3) == 3,
         'preprocess: expected a ByteTensor of shape (H, W, 3)')
  local H0, W0 = img_hwc_bytes:size(1), img_hwc_bytes:size(2)
  local ph, pw = ffi.new('int[1]'), ffi.new('int[1]')
  assert(C.dc_preprocess_size(H0, W0, image_size, ph, pw) == 0, 'image.scale leaves no pixels')
  -- the contiguous copy stays referenced by a local until the call has returned (a temporary could be collected between
  -- the evaluation of the arguments and the C call: ffi.cast allocates)
  local bytes = img_hwc_bytes:contiguous()
  local pp = ffi.new('void*[1]')
  hip.check(self.ctx, C.dc_malloc(self.ctx, pp, 3 * ph[0] * pw[0] * 4), 'dc_malloc')
  local rc = C.dc_preprocess_u8(self.ctx, bytes:data(), H0, W0, 0, image_size, ffi.cast('float*', pp[0]), nil)
  if rc ~= 0 then
    C.dc_free(self.ctx, pp[0])                        -- hip.check raises: give the buffer back first
    hip.check(self.ctx, rc, 'dc_preprocess_u8')
  end
  bytes = nil
  return pp[0], ph[0], pw[0]
end
function Model:convert(dtype, use_cudnn) return self end
function Model:evaluate() return self end
function Model:type() return self end

function Model:decodeSequence(seq)   -- LanguageModel.lua:86-103
  local captions = {}
  local N, T = seq:size(1), seq:size(2)
  for i = 1, N do
    local caption = ''
    for t = 1, T do
      local idx = seq[{i, t}]
      if idx == self.vocab_size + 1 or idx == 0 then break end
      if t > 1 then caption = caption .. ' ' end
      caption = caption .. self.idx_to_token[idx]
    end
    table.insert(captions, caption)
  end
  return captions
end

function Model:forward_test(input)
  self:_push_test_args()
  assert(input:dim() == 4 and input:size(1) == 1 and input:size(2) == 3)  -- DenseCapModel.lua:244
  local img = input:float():contiguous()
  local H, W, T = img:size(3), img:size(4), self.seq_length
  local P = self:_capacity(H, W)
  local boxes, scores = torch.FloatTensor(P, 4), torch.FloatTensor(P, 1)
  local tokens = torch.IntTensor(P, T)
  local r = ffi.new('dc_result')
  r.capacity = P
  r.boxes, r.scores = torch.data(boxes), torch.data(scores)
  r.tokens = torch.data(tokens)
  hip.check(self.ctx, C.dc_forward_test(self.ctx, fptr(img), H, W, 0, r), 'dc_forward_test')
  local K = r.K
  if K == 0 then return torch.FloatTensor(), torch.FloatTensor(), {} end
  local seq = tokens[{{1, K}}]:long()
  return boxes[{{1, K}}]:clone(), scores[{{1, K}}]:clone(), self:decodeSequence(seq)
end

function Model:extractFeatures(input)
  self:_push_test_args()
  local img = input:float():contiguous()
  local H, W = img:size(3), img:size(4)
  local P = self:_capacity(H, W)
  local boxes, feats = torch.FloatTensor(P, 4), torch.FloatTensor(P, self.fc_dim)
  local K = ffi.new('int32_t[1]')
  hip.check(self.ctx, C.dc_extract_features(self.ctx, fptr(img), H, W, 0, P, torch.data(boxes),
                                            torch.data(feats), K), 'dc_extract_features')
  return boxes[{{1, K[0]}}]:clone(), feats[{{1, K[0]}}]:clone()
end

-- Multi-GPU (one LuaJIT process per GPU; the reference is single-device, densecap/utils.lua:22-36): every rank runs
-- forward_raw on its shard of the image list, then ONE gather on rank 0 (RCCL point-to-point over xGMI).
--   id = DenseCapModelHIP.commUniqueId()                       -- rank 0; pass the 128-byte string to the others
--   model:commInit(id, rank, world)
--   all = model:gatherResults(results)                         -- results: array of dc_result filled by forward_raw
function Model.commUniqueId()
  local id = ffi.new('uint8_t[128]')
  hip.check(nil, C.dc_comm_unique_id(id), 'dc_comm_unique_id')
  return ffi.string(id, 128)
end
-- self_transport (world == 1 only): build the RCCL carrier for the single rank too, so that gatherResults travels through
-- ncclSend / ncclRecv to itself (DC_COMM_SELF_TRANSPORT) -- the multi-GPU code path on one GPU
function Model:commInit(id, rank, world, self_transport)
  local pc = ffi.new('dc_comm*[1]')
  hip.check(self.ctx, C.dc_comm_create_ex(pc, self.ctx, id, rank, world, self_transport and 1 or 0), 'dc_comm_create')
  self.comm, self.rank, self.world = ffi.gc(pc[0], C.dc_comm_destroy), rank, world
end
-- forward_test without string decoding: returns a dc_result (and the tensors that own its buffers)
function Model:forward_raw(input)
  self:_push_test_args()
  local img = input:float():contiguous()
  local H, W, T = img:size(3), img:size(4), self.seq_length
  local P = self:_capacity(H, W)
  local keep = {torch.FloatTensor(P, 4), torch.FloatTensor(P), torch.IntTensor(P, T)}
  local r = ffi.new('dc_result')
  r.capacity, r.boxes, r.scores, r.tokens = P, torch.data(keep[1]), torch.data(keep[2]), torch.data(keep[3])
  hip.check(self.ctx, C.dc_forward_test(self.ctx, fptr(img), H, W, 0, r), 'dc_forward_test')
  return r, keep
end
function Model:gatherResults(results)   -- results: Lua array of dc_result with one capacity
  local n = #results
  local loc = ffi.new('dc_result[?]', n)
  for i = 1, n do loc[i - 1] = results[i] end
  local all, keep = nil, {}
  if self.rank == 0 then
    all = ffi.new('dc_result[?]', n * self.world)
    local P, T = loc[0].capacity, self.seq_length
    for i = 0, n * self.world - 1 do
      local k = {torch.FloatTensor(P, 4), torch.FloatTensor(P), torch.IntTensor(P, T)}
      keep[#keep + 1] = k
      all[i].capacity, all[i].boxes, all[i].scores, all[i].tokens = P, torch.data(k[1]), torch.data(k[2]), torch.data(k[3])
    end
  end
  local rc = C.dc_gather_results(self.comm, loc, n, all)
  if rc < 0 then error('dc_gather_results: ' .. ffi.string(C.dc_comm_last_error(self.comm))) end
  return all, keep
end

return Model
