--[[
run_model.lua on the MI355X path, with EVERY flag of the reference script (-input_image, -input_dir, -input_split,
-max_images, -output_dir, -output_vis, -output_vis_dir, -image_size, ... run_model.lua:26-61).

Nothing of run_model.lua is restated here: the reference's own script is loaded as text, the two statements that
choose the device and take the model out of the checkpoint are replaced, and the result is run with the caller's
arguments.  These two substitutions are the whole integration (INTEGRATION.md section 2):

  run_model.lua:145   local dtype, use_cudnn = utils.setup_gpus(opt.gpu, opt.use_cudnn)
                  ->  local dtype, use_cudnn = 'torch.FloatTensor', false          -- host tensors stay float; no cutorch
  run_model.lua:147   local model = checkpoint.model
                  ->  local model = require('DenseCapModelHIP').fromCheckpoint(checkpoint.model, opt.gpu)
                                            :setCaptionOrder(os.getenv('DENSECAP_CAPTION_ORDER') ~= '0')
      (captions after the final NMS, as `python -m densecap_amd.run_model` defaults to: identical outputs, the decode runs on the
       rows the final NMS keeps; DENSECAP_CAPTION_ORDER=0 restores the reference's order)

usage (from the densecap checkout, lua/ on package.path, libdensecap_hip.so on the loader path or in DENSECAP_HIP_LIB):
    th /path/to/lua/run_model_hip.lua -input_dir imgs -max_images 10 -output_vis_dir vis/data
(No Lua runtime exists in the build container: the substitutions are checked against the reference text by
tests/test_abi_and_host.py::test_lua_run_model_substitutions_match_the_reference.)
--]]
local path = os.getenv('DENSECAP_RUN_MODEL') or 'run_model.lua'
local f = assert(io.open(path, 'r'), 'run_model_hip.lua: cannot open ' .. path .. ' (run from the densecap checkout or set DENSECAP_RUN_MODEL)')
local src = f:read('*a')
f:close()

local SUBSTITUTIONS = {
  {"local dtype, use_cudnn = utils%.setup_gpus%(opt%.gpu, opt%.use_cudnn%)",
   "local dtype, use_cudnn = 'torch.FloatTensor', false"},
  {"local model = checkpoint%.model",
   "local model = require('DenseCapModelHIP').fromCheckpoint(checkpoint.model, opt.gpu):setCaptionOrder(os.getenv('DENSECAP_CAPTION_ORDER') ~= '0')"},
}
for _, s in ipairs(SUBSTITUTIONS) do
  local n
  src, n = src:gsub(s[1], (s[2]:gsub('%%', '%%%%')))
  assert(n == 1, 'run_model_hip.lua: expected exactly one match of "' .. s[1] .. '" in ' .. path .. ', found ' .. n)
end

local chunk = assert((loadstring or load)(src, '@' .. path .. ' (model swapped for DenseCapModelHIP)'))
return chunk(...)
