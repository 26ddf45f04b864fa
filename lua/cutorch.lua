--[[ cutorch.lua -- stands in for the CUDA tensor package the reference requires when -gpu_id > 0 (src/train.lua:244-247).
     On MI355X there is no CudaTensor: device memory belongs to lua/model.lua (hipMalloc through lua/aocr_ffi.lua) and tensors that
     reach `localize` (src/utils/utils.lua:96-102) stay host tensors -- `:cuda()` is the identity here; model:step uploads them. ]]
local A = require 'aocr_ffi'
cutorch = cutorch or {}
function cutorch.setDevice(id)              -- 1-based like Torch7 (train.lua:246)
    A.hip_ok(A.hip.hipSetDevice(id - 1), 'hipSetDevice'); cutorch._device = id
end
function cutorch.getDevice() return cutorch._device or 1 end
function cutorch.getDeviceCount()
    local n = require('ffi').new('int[1]'); A.hip_ok(A.hip.hipGetDeviceCount(n), 'hipGetDeviceCount'); return n[0]
end
function cutorch.manualSeed(seed) cutorch._seed = seed end     -- train.lua:247; the fused path draws its dropout masks from (seed, step, layer, index)
function cutorch.synchronize() A.hip_ok(A.hip.hipDeviceSynchronize(), 'hipDeviceSynchronize') end
-- `thing:cuda()` (utils.lua:99): tensors and nn modules stay where they are
local function identity(self) return self end
for _, name in ipairs({'DoubleTensor', 'FloatTensor', 'IntTensor', 'LongTensor', 'ByteTensor'}) do
    local mt = torch.getmetatable('torch.' .. name)
    if mt and not mt.cuda then mt.cuda = identity end
end
if nn and nn.Module and not nn.Module.cuda then nn.Module.cuda = identity end
if nn and nn.Criterion and not nn.Criterion.cuda then nn.Criterion.cuda = identity end
return cutorch
