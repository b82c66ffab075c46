--[[
run_model.lua with the model swapped for the MI355X path.  Differences from the reference's
run_model.lua are the three marked lines; preprocessing (run_model.lua:67-74), result JSON
(:89-95,:182-188) and flags are the reference's own.
--]]
require 'torch'
require 'nn'
require 'image'
require 'densecap.DenseCapModel'            -- needed to deserialise the checkpoint's classes
local utils = require 'densecap.utils'
local box_utils = require 'densecap.box_utils'
local DenseCapModelHIP = require 'DenseCapModelHIP'   -- (1) new

local cmd = torch.CmdLine()
cmd:option('-checkpoint', 'data/models/densecap/densecap-pretrained-vgg16.t7')
cmd:option('-image_size', 720)
cmd:option('-rpn_nms_thresh', 0.7)
cmd:option('-final_nms_thresh', 0.3)
cmd:option('-num_proposals', 1000)
cmd:option('-input_image', '')
cmd:option('-gpu', 0)
local opt = cmd:parse(arg)

local checkpoint = torch.load(opt.checkpoint)
local model = DenseCapModelHIP.fromCheckpoint(checkpoint.model, opt.gpu)   -- (2) was: checkpoint.model
model:convert('torch.FloatTensor', false)                                  -- (3) dtype is irrelevant
model:setTestArgs{rpn_nms_thresh = opt.rpn_nms_thresh, final_nms_thresh = opt.final_nms_thresh,
                  num_proposals = opt.num_proposals}
model:evaluate()

local img = image.load(opt.input_image, 3)
img = image.scale(img, opt.image_size):float()
local H, W = img:size(2), img:size(3)
local img_caffe = img:view(1, 3, H, W)
img_caffe = img_caffe:index(2, torch.LongTensor{3, 2, 1}):mul(255)
local vgg_mean = torch.FloatTensor{103.939, 116.779, 123.68}
img_caffe:add(-1, vgg_mean:view(1, 3, 1, 1):expand(1, 3, H, W))

local boxes_xcycwh, scores, captions = model:forward_test(img_caffe)
local boxes_xywh = box_utils.xcycwh_to_xywh(boxes_xcycwh)
utils.write_json('vis/data/results.json', {
  results = {{img_name = paths.basename(opt.input_image), boxes = boxes_xywh:totable(),
              scores = scores:float():view(-1):totable(), captions = captions}},
  opt = opt})
