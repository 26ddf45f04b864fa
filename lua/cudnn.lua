--[[ cudnn.lua -- the package name src/model/cnn.lua:12-42 builds its layers from.  On MI355X the arithmetic of those layers runs in
     libaocr (implicit-GEMM MFMA convolutions with fused bias / ReLU / pooling epilogues), driven by lua/model.lua; the classes
     below are the CPU nn modules under the cudnn.* type names, so that createCNNModel() still returns the nn.Sequential the
     reference serializes: they carry the parameters of a checkpoint (model:save / model:load, model.lua:45-80,720-725) and are
     never run.  Constructor signatures as in cudnn.torch [upstream]. ]]
require 'nn'
cudnn = cudnn or {}
local SC, scparent = torch.class('cudnn.SpatialConvolution', 'nn.SpatialConvolution')
function SC:__init(nIn, nOut, kW, kH, dW, dH, padW, padH, groups) scparent.__init(self, nIn, nOut, kW, kH, dW, dH, padW, padH) end
local RL, rlparent = torch.class('cudnn.ReLU', 'nn.ReLU')
function RL:__init(inplace) rlparent.__init(self, inplace) end
local MP, mpparent = torch.class('cudnn.SpatialMaxPooling', 'nn.SpatialMaxPooling')
function MP:__init(kW, kH, dW, dH, padW, padH) mpparent.__init(self, kW, kH, dW, dH, padW, padH) end
cudnn.benchmark = false; cudnn.fastest = false; cudnn.verbose = false
function cudnn.convert(net, dst) return net end
return cudnn
