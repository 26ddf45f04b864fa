--[[ cunn.lua -- required by src/train.lua:245 next to cutorch; the CUDA nn kernels it would register are replaced by libaocr's
     HIP kernels behind lua/model.lua, so there is nothing to load. ]]
require 'nn'
return {}
