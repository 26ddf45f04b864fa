--[[ aocr_nn.lua -- the nn.Module surface of src/model/*.lua (SURVEY.md 8(b) B2) on libaocr's MODULE-level entry points.

     `require 'aocr_nn'.install()` gives the classes the reference builds its graphs from an updateOutput / updateGradInput /
     accGradParameters that run on the MI355X through the LuaJIT FFI (lua/aocr_ffi.lua), so that a reference-style
         local y = cnn_model:forward(x); ... ; cnn_model:backward(x, dy)            (cnn.lua:12-45, model.lua:285,692)
     executes HIP kernels instead of the CPU nn code the classes fall back to:
         cudnn.SpatialConvolution / cudnn.ReLU / cudnn.SpatialMaxPooling   -> aocr_conv1_* / aocr_conv2d_* / aocr_unpool_relu_backward
         nn.SpatialBatchNormalization (+ the ReLU behind it)               -> aocr_batchnorm_relu_*
         nn.Linear / nn.LinearNoBias (model_utils.lua:57-116)              -> aocr_gemm
         nn.LookupTable                                                    -> aocr_lookup_*
         the LSTM cell of createLSTM (LSTM.lua:79-105)                     -> aocr_lstm_cell_forward[_zx] / aocr_lstm_cell_backward
         create_decoder_attn (LSTM.lua:124-162)                            -> aocr_attention_* + aocr_gemm(+tanh) + aocr_pointwise
         nn.LogSoftMax + nn.ClassNLLCriterion (output_projector.lua:6, criterion.lua:3-9) -> aocr_logsoftmax_nll
     Values travel between modules as DeviceTensors (fp32 in HBM, maps channels-last (B,H,W,C)); a torch.FloatTensor handed to the
     first module is uploaded (NCHW -> NHWC), `:float()` on a DeviceTensor downloads it.

     MI355X-first, not module-for-module: the reference's Sequential calls conv, ReLU and pooling one after the other, the library
     has ONE kernel for the three.  So a convolution's updateOutput only RECORDS the product (a "pending" value); a following ReLU /
     max-pooling / BatchNorm extends the record, and the first module that needs real data (the next convolution, View, a criterion)
     makes the single fused call.  The backward pass mirrors it: pooling and ReLU record, the convolution's updateGradInput makes
     the un-pool + data-gradient calls, accGradParameters the filter-gradient call.

     UNEXECUTED in the build container (no Lua / LuaJIT / Torch7: SURVEY.md 8(c)).  Its executed twin is the module-level section
     of tests/abi_harness.cc, which makes the same calls in the same order with the same arguments against the same library and is
     checked against the fused step and the golden logits; tests/test_lua_cdef_cpu.py pins every prototype used here to
     include/aocr.h and checks that this file calls declared entry points only. ]]
require 'nn'
local ffi = require 'ffi'
local A = require 'aocr_ffi'
local L = A.lib

local M = {compute = 0}                       -- AOCR_COMPUTE_F32 (exact fp32 MFMA); 1 = bf16 operands
local F = 'float*'

-- ------------------------------------------------------------------------------------------------ DeviceTensor
local DT = {}
DT.__index = DT
local function numel(sizes) local n = 1; for _, v in ipairs(sizes) do n = n * v end; return n end
function M.tensor(sizes)
    local n = numel(sizes)
    local self = setmetatable({buf = A.device_bytes(n * 4), sizes = sizes, n = n}, DT)
    self.buf:zero()
    return self
end
function DT:ptr(offset) return self.buf:as(F) + (offset or 0) end
function DT:size(i) if i then return self.sizes[i] end; return self.sizes end
function DT:nElement() return self.n end
function DT:zero() self.buf:zero(); return self end
function DT:view(sizes) assert(numel(sizes) == self.n); return setmetatable({buf = self.buf, sizes = sizes, n = self.n}, DT) end
function DT:add(other)                        -- nngraph sums the gradients of a shared input with :add
    A.check(L.aocr_pointwise(nil, 0, self:ptr(), other:ptr(), self:ptr(), self.n), 'aocr_pointwise')
    return self
end
-- maps: host (B,C,H,W) <-> device (B,H,W,C); everything else is copied as it is
function M.from_host(t)
    t = t:float()
    if t:dim() == 4 then t = t:permute(1, 3, 4, 2) end
    t = t:contiguous()
    local d = M.tensor(t:size():totable())
    A.upload(d.buf, t, d.n * 4)
    return d
end
function DT:float()
    local t = torch.FloatTensor(torch.LongStorage(self.sizes))
    A.download(t, self.buf, self.n * 4)
    if #self.sizes == 4 then t = t:permute(1, 4, 2, 3):contiguous() end
    return t
end
local function is_dt(x) return getmetatable(x) == DT end

-- ------------------------------------------------------------------------------------------------ pending values
-- {kind = 'affine', x = host tensor, add, mul}                      AddConstant / MulConstant in front of conv1 (cnn.lua:9-10)
-- {kind = 'conv', conv = module, x = DeviceTensor | affine, relu = bool, pool = 0|1|2}
-- {kind = 'bn', bn = module, x = DeviceTensor (conv output)}
local function resolve(v)
    if is_dt(v) then return v end
    if torch.isTensor(v) then return M.from_host(v) end
    assert(type(v) == 'table' and v.kind, 'aocr_nn: unexpected value between modules')
    if v.value then return v.value end
    if v.kind == 'conv' then v.value = v.conv:_run(v)
    elseif v.kind == 'bn' then error('nn.SpatialBatchNormalization without the ReLU behind it is not a module of the path (cnn.lua:23-24)')
    else error('aocr_nn: AddConstant / MulConstant only in front of the first convolution (cnn.lua:9-12)') end
    return v.value
end
M.resolve = resolve

-- parameters of a module on the device, refreshed when the host copy changed (`module._dirty = true`, set by M.sync / :reset)
local function dev_param(mod, name, layout)
    mod._dev = mod._dev or {}
    local e = mod._dev[name]
    if e == nil or mod._dirty then
        local t = mod[name]
        if layout == 'conv' then t = t:view(mod.nOutputPlane, mod.nInputPlane, mod.kH, mod.kW):permute(1, 3, 4, 2) end   -- [Cout][Cin][kH][kW] -> [Cout][kH][kW][Cin]
        t = t:float():contiguous()
        e = e or M.tensor({t:nElement()})
        A.upload(e.buf, t, e.n * 4)
        mod._dev[name] = e
    end
    return e
end
local function dev_grad(mod, name)            -- device-side gradWeight / gradBias, zeroed by zeroGradParameters
    mod._devg = mod._devg or {}
    if mod._devg[name] == nil then mod._devg[name] = M.tensor({mod[name]:nElement()}) end
    return mod._devg[name]
end
-- device gradients -> the module's host gradWeight / gradBias (what optim.sgd_list and getParameters() see)
function M.pull_gradients(mod)
    for name, g in pairs(mod._devg or {}) do
        local t = torch.FloatTensor(g.n); A.download(t, g.buf, g.n * 4)
        if name == 'gradWeight' and mod.kH then t = t:view(mod.nOutputPlane, mod.kH, mod.kW, mod.nInputPlane):permute(1, 4, 2, 3):contiguous() end
        mod[name]:copy(t:viewAs(mod[name]))
    end
end
local bn_scratch
local function scratch() bn_scratch = bn_scratch or A.device_bytes(8 * 2 ^ 20); return bn_scratch.ptr end      -- AOCR_BN_SCRATCH_BYTES

-- ------------------------------------------------------------------------------------------------ cnn.lua:9-45
local function pool_mode(m)                   -- cudnn.SpatialMaxPooling(kW,kH,dW,dH): (2,2,2,2) -> 1, (1,2,1,2) -> 2 (halves the HEIGHT, cnn.lua:29)
    if m.kW == 2 and m.kH == 2 and m.dW == 2 and m.dH == 2 then return 1 end
    if m.kW == 1 and m.kH == 2 and m.dW == 1 and m.dH == 2 then return 2 end
    error('aocr_nn: pooling window not on the path')
end

local function install_cnn()
    local AC, MC = nn.AddConstant, nn.MulConstant
    function AC:updateOutput(input) self.output = {kind = 'affine', x = input, add = self.constant_scalar, mul = 1}; return self.output end
    function MC:updateOutput(input)
        assert(type(input) == 'table' and input.kind == 'affine', 'aocr_nn: MulConstant only behind AddConstant (cnn.lua:9-10)')
        self.output = {kind = 'affine', x = input.x, add = input.add, mul = self.constant_scalar}; return self.output
    end
    function AC:updateGradInput(input, gradOutput) self.gradInput = gradOutput; return gradOutput end      -- d(image) is never needed (model.lua:692)
    MC.updateGradInput = AC.updateGradInput

    local SC = cudnn.SpatialConvolution
    function SC:updateOutput(input)
        self.output = {kind = 'conv', conv = self, x = input, relu = false, pool = 0}
        return self.output
    end
    -- the one fused forward call of a conv (+ ReLU + pooling) record
    function SC:_run(p)
        local w, b = dev_param(self, 'weight', 'conv'), dev_param(self, 'bias'); self._dirty = false
        if self.nInputPlane == 1 then                                          -- cnn.lua:9-15: normalise + conv1 + ReLU + 2x2 pooling in one kernel
            local a = p.x
            assert(type(a) == 'table' and a.kind == 'affine' and a.add == -128 and math.abs(a.mul - 1 / 128) < 1e-9 and p.relu and p.pool == 1,
                   'aocr_nn: the 1-channel layer is (x - 128) / 128 -> conv 3x3 -> ReLU -> maxpool 2x2 (cnn.lua:9-15)')
            local img = a.x:float():contiguous()                               -- (B,1,32,W) values 0..255
            local B, H, W = img:size(1), img:size(3), img:size(4)
            self._x = M.tensor({B, H, W}); A.upload(self._x.buf, img, self._x.n * 4)
            local y = M.tensor({B, H / 2, W / 2, self.nOutputPlane})
            A.check(L.aocr_conv1_forward(nil, self._x:ptr(), w:ptr(), b:ptr(), y:ptr(), B, H, W), 'aocr_conv1_forward')
            self._pooled = y
            return y
        end
        local x = resolve(p.x)
        local B, H, W, Cin = x.sizes[1], x.sizes[2], x.sizes[3], x.sizes[4]
        local Ho, Wo = H + 2 * self.padH - self.kH + 1, W + 2 * self.padW - self.kW + 1
        local Hp, Wp = Ho, Wo
        if p.pool == 1 then Hp, Wp = math.floor(Ho / 2), math.floor(Wo / 2) elseif p.pool == 2 then Hp = math.floor(Ho / 2) end
        local y = M.tensor({B, Hp, Wp, self.nOutputPlane})
        self._x, self._pool, self._relu, self._geo = x, p.pool, p.relu, {B, H, W, Cin, Ho, Wo}
        self._idx = p.pool > 0 and A.device_bytes(y.n) or nil                   -- uint8 arg-max position inside the window
        A.check(L.aocr_conv2d_forward(nil, M.compute, x:ptr(), w:ptr(), b:ptr(), y:ptr(), self._idx and self._idx:as('uint8_t*') or nil,
                                      B, H, W, Cin, self.nOutputPlane, self.kH, self.padH, p.relu and 1 or 0, p.pool), 'aocr_conv2d_forward')
        self._pooled = y
        return y
    end
    -- gradOutput: DeviceTensor w.r.t. this layer's (pooled, rectified) output, or a record {kind='dconv', g=...} from the ReLU / pooling
    -- modules behind it.  dy (pre-pool, pre-ReLU) is made once and shared by updateGradInput and accGradParameters.
    function SC:_dy(gradOutput)
        if self._dy_for == gradOutput then return self._dyv end
        local g = is_dt(gradOutput) and gradOutput or gradOutput.g
        local dy = g
        if self.nInputPlane ~= 1 then
            local B, H, W, Cin, Ho, Wo = unpack(self._geo)
            if self._pool > 0 then
                dy = M.tensor({B, Ho, Wo, self.nOutputPlane})
                A.check(L.aocr_unpool_relu_backward(nil, g:ptr(), self._pooled:ptr(), self._idx:as('uint8_t*'), dy:ptr(), B, Ho, Wo, self.nOutputPlane, self._pool),
                        'aocr_unpool_relu_backward')
            elseif self._relu then
                dy = M.tensor(g.sizes)
                A.check(L.aocr_pointwise(nil, 2, g:ptr(), self._pooled:ptr(), dy:ptr(), g.n), 'aocr_pointwise')
            end
        end
        self._dy_for, self._dyv = gradOutput, dy
        return dy
    end
    function SC:updateGradInput(input, gradOutput)
        if self.nInputPlane == 1 then self.gradInput = nil; return nil end     -- d(image) is not computed (conv1_bwd recomputes the window for the filter gradient only)
        local dy = self:_dy(gradOutput)
        local B, H, W, Cin = unpack(self._geo)
        local dx = M.tensor({B, H, W, Cin})
        A.check(L.aocr_conv2d_backward_data(nil, M.compute, dy:ptr(), dev_param(self, 'weight', 'conv'):ptr(), dx:ptr(), B, H, W, Cin, self.nOutputPlane,
                                            self.kH, self.padH), 'aocr_conv2d_backward_data')
        self.gradInput = dx
        return dx
    end
    function SC:accGradParameters(input, gradOutput, scale)
        assert(scale == nil or scale == 1, 'aocr_nn: gradient scale other than 1 is not on the path')
        local dw, db = dev_grad(self, 'gradWeight'), dev_grad(self, 'gradBias')
        local dy = self:_dy(gradOutput)
        if self.nInputPlane == 1 then
            local B, H, W = self._x.sizes[1], self._x.sizes[2], self._x.sizes[3]
            A.check(L.aocr_conv1_backward(nil, self._x:ptr(), dev_param(self, 'weight', 'conv'):ptr(), dev_param(self, 'bias'):ptr(), dy:ptr(), dw:ptr(), db:ptr(), B, H, W),
                    'aocr_conv1_backward')
        else
            local B, H, W, Cin = unpack(self._geo)
            A.check(L.aocr_conv2d_backward_filter(nil, M.compute, self._x:ptr(), dy:ptr(), dw:ptr(), db:ptr(), B, H, W, Cin, self.nOutputPlane, self.kH, self.padH),
                    'aocr_conv2d_backward_filter')
        end
    end
    function SC:zeroGradParameters() for _, g in pairs(self._devg or {}) do g:zero() end; self.gradWeight:zero(); self.gradBias:zero() end

    local RL = cudnn.ReLU
    function RL:updateOutput(input)
        if type(input) == 'table' and input.kind == 'conv' and not input.value then input.relu = true; self.output = input
        elseif type(input) == 'table' and input.kind == 'bn' then self.output = input.bn:_run(input)                       -- BatchNorm + ReLU: one kernel
        else
            local x = resolve(input); local y = M.tensor(x.sizes)
            A.check(L.aocr_pointwise(nil, 3, x:ptr(), nil, y:ptr(), x.n), 'aocr_pointwise'); self.output = y; self._plain = true
        end
        return self.output
    end
    function RL:updateGradInput(input, gradOutput)                              -- folded into the producer's backward (conv: _dy, BatchNorm: its own call)
        if self._plain then
            local g = M.tensor(gradOutput.sizes)
            A.check(L.aocr_pointwise(nil, 2, gradOutput:ptr(), self.output:ptr(), g:ptr(), g.n), 'aocr_pointwise'); self.gradInput = g
        else self.gradInput = gradOutput end
        return self.gradInput
    end
    local MP = cudnn.SpatialMaxPooling
    function MP:updateOutput(input)
        assert(type(input) == 'table' and input.kind == 'conv' and input.relu and not input.value, 'aocr_nn: max-pooling follows conv + ReLU on the path (cnn.lua:12-36)')
        input.pool = pool_mode(self); self.output = input
        return self.output
    end
    function MP:updateGradInput(input, gradOutput) self.gradInput = gradOutput; return gradOutput end                       -- un-pooled by the convolution's _dy

    local BN = nn.SpatialBatchNormalization
    function BN:updateOutput(input)
        self.output = {kind = 'bn', bn = self, x = resolve(input)}
        return self.output
    end
    function BN:_run(p, tb_rows)
        local x = p.x; local C = x.sizes[#x.sizes]; local rows = x.n / C
        local y = M.tensor(x.sizes)
        self._dev = self._dev or {}
        self._dev.rm = self._dev.rm or M.from_host(self.running_mean); self._dev.rv = self._dev.rv or M.from_host(self.running_var)
        self._save = self._save or M.tensor({2 * C})
        A.check(L.aocr_batchnorm_relu_forward(nil, x:ptr(), y:ptr(), dev_param(self, 'weight'):ptr(), dev_param(self, 'bias'):ptr(), self._dev.rm:ptr(), self._dev.rv:ptr(),
                                              self._save:ptr(), scratch(), rows, C, self.train and 1 or 0, self.train and 1 or 0, tb_rows or 0), 'aocr_batchnorm_relu_forward')
        self._dirty = false; self._x, self._y, self._tb = x, y, tb_rows or 0
        p.value = y
        return y
    end
    function BN:updateGradInput(input, gradOutput)                              -- gradient at the ReLU output -> gradient at the BatchNorm input, + gradWeight / gradBias
        local C = self._x.sizes[#self._x.sizes]
        local dx = M.tensor(self._x.sizes)
        A.check(L.aocr_batchnorm_relu_backward(nil, self._x:ptr(), self._y:ptr(), gradOutput:ptr(), dev_param(self, 'weight'):ptr(), self._save:ptr(), dx:ptr(),
                                               dev_grad(self, 'gradWeight'):ptr(), dev_grad(self, 'gradBias'):ptr(), scratch(), self._x.n / C, C, self._tb), 'aocr_batchnorm_relu_backward')
        self.gradInput = dx
        return dx
    end
    function BN:accGradParameters() end                                         -- accumulated by the same kernel as the input gradient
    BN.zeroGradParameters = SC.zeroGradParameters
    -- running statistics back to the host module (what model:save serializes)
    function BN:pull_running() if self._dev and self._dev.rm then self.running_mean:copy(self._dev.rm:float()); self.running_var:copy(self._dev.rv:float()) end end

    -- nn.View(512, -1) + nn.Transpose({2, 3}) (cnn.lua:44-45): (B,512,1,T') -> (B,T',512).  Channels-last (B,1,T',512) IS that layout.
    function nn.View:updateOutput(input) local x = resolve(input); self.output = x:view({x.sizes[1], x.n / x.sizes[1] / x.sizes[#x.sizes], x.sizes[#x.sizes]}); return self.output end
    function nn.View:updateGradInput(input, gradOutput) self.gradInput = gradOutput; return gradOutput end
    function nn.Transpose:updateOutput(input) self.output = resolve(input); return self.output end
    function nn.Transpose:updateGradInput(input, gradOutput) self.gradInput = gradOutput; return gradOutput end
end

-- ------------------------------------------------------------------------------------------------ Linear / LinearNoBias / LookupTable
local function install_linear()
    local LN = nn.Linear
    function LN:updateOutput(input)                                             -- (rows, in) -> (rows, out); LinearNoBias (model_utils.lua:57-116) has bias == nil
        local x = resolve(input); local rows = x.n / self.weight:size(2)
        local y = M.tensor({rows, self.weight:size(1)})
        A.check(L.aocr_gemm(nil, M.compute, x:ptr(), self.weight:size(2), 1, dev_param(self, 'weight'):ptr(), self.weight:size(2), 1, y:ptr(), self.weight:size(1),
                            rows, self.weight:size(1), self.weight:size(2), self.bias and dev_param(self, 'bias'):ptr() or nil, self._act or 0), 'aocr_gemm')
        self._dirty = false; self._x, self.output = x, y
        return y
    end
    function LN:updateGradInput(input, gradOutput)                              -- gradInput = gradOutput W
        local rows = gradOutput.n / self.weight:size(1)
        local dx = M.tensor({rows, self.weight:size(2)})
        A.check(L.aocr_gemm(nil, M.compute, gradOutput:ptr(), self.weight:size(1), 1, dev_param(self, 'weight'):ptr(), self.weight:size(2), 0, dx:ptr(), self.weight:size(2),
                            rows, self.weight:size(2), self.weight:size(1), nil, 0), 'aocr_gemm')
        self.gradInput = dx
        return dx
    end
    function LN:accGradParameters(input, gradOutput, scale)                     -- gradWeight += gradOutput^T input; gradBias += column sums (a product with ones)
        local rows = gradOutput.n / self.weight:size(1)
        A.check(L.aocr_gemm(nil, M.compute, gradOutput:ptr(), self.weight:size(1), 0, self._x:ptr(), self.weight:size(2), 0, dev_grad(self, 'gradWeight'):ptr(), self.weight:size(2),
                            self.weight:size(1), self.weight:size(2), rows, nil, 1), 'aocr_gemm')
        if self.bias then
            M._ones = (M._ones and M._ones.n >= rows) and M._ones or M.from_host(torch.FloatTensor(math.max(rows, 4096)):fill(1))
            A.check(L.aocr_gemm(nil, 0, gradOutput:ptr(), self.weight:size(1), 0, M._ones:ptr(), 1, 0, dev_grad(self, 'gradBias'):ptr(), 1, self.weight:size(1), 1, rows, nil, 1), 'aocr_gemm')
        end
    end
    function LN:zeroGradParameters() for _, g in pairs(self._devg or {}) do g:zero() end; self.gradWeight:zero(); if self.gradBias then self.gradBias:zero() end end

    local LT = nn.LookupTable
    function LT:updateOutput(input)                                             -- input: IntTensor / LongTensor of 1-based ids
        local ids = input:int():contiguous(); local n = ids:nElement()
        self._ids = A.device_bytes(n * 4); A.upload(self._ids, ids, n * 4); self._n = n
        local y = M.tensor({n, self.weight:size(2)})
        A.check(L.aocr_lookup_forward(nil, dev_param(self, 'weight'):ptr(), self._ids:as('int32_t*'), y:ptr(), n, self.weight:size(2)), 'aocr_lookup_forward')
        self._dirty = false; self.output = y
        return y
    end
    function LT:updateGradInput() self.gradInput = nil; return nil end
    function LT:accGradParameters(input, gradOutput)
        A.check(L.aocr_lookup_backward(nil, gradOutput:ptr(), self._ids:as('int32_t*'), dev_grad(self, 'gradWeight'):ptr(), self._n, self.weight:size(2), self.weight:size(1)),
                'aocr_lookup_backward')
    end
end

-- ------------------------------------------------------------------------------------------------ LSTM cell, attention, criterion
-- One layer of createLSTM at one time step (LSTM.lua:79-105): i2h / h2h are the nn.Linear pair of the layer (parameter containers),
-- x (B,in), prev_c / prev_h (B,H) DeviceTensors.  Returns next_c, next_h and the cache the backward call needs.
function M.lstm_cell_forward(i2h, h2h, x, prev_c, prev_h)
    local B, H, inp = prev_c.sizes[1], prev_c.sizes[2], i2h.weight:size(2)
    local c, h, gates = M.tensor({B, H}), M.tensor({B, H}), M.tensor({B, 4 * H})
    if inp % 16 == 0 then
        A.check(L.aocr_lstm_cell_forward(nil, M.compute, x:ptr(), inp, prev_h:ptr(), prev_c:ptr(), dev_param(i2h, 'weight'):ptr(), dev_param(i2h, 'bias'):ptr(),
                                         dev_param(h2h, 'weight'):ptr(), dev_param(h2h, 'bias'):ptr(), c:ptr(), h:ptr(), gates:ptr(), B, H), 'aocr_lstm_cell_forward')
    else                                                                        -- the decoder's first layer: E + Hd input columns (LSTM.lua:59-64)
        i2h._bsum = i2h._bsum or M.tensor({4 * H})
        A.check(L.aocr_pointwise(nil, 0, dev_param(i2h, 'bias'):ptr(), dev_param(h2h, 'bias'):ptr(), i2h._bsum:ptr(), 4 * H), 'aocr_pointwise')
        local zx = M.tensor({B, 4 * H})
        A.check(L.aocr_gemm(nil, M.compute, x:ptr(), inp, 1, dev_param(i2h, 'weight'):ptr(), inp, 1, zx:ptr(), 4 * H, B, 4 * H, inp, i2h._bsum:ptr(), 0), 'aocr_gemm')
        A.check(L.aocr_lstm_cell_forward_zx(nil, M.compute, zx:ptr(), 4 * H, prev_h:ptr(), prev_c:ptr(), dev_param(h2h, 'weight'):ptr(), c:ptr(), h:ptr(), gates:ptr(), B, H),
                'aocr_lstm_cell_forward_zx')
    end
    i2h._dirty, h2h._dirty = false, false
    return c, h, {x = x, prev_c = prev_c, prev_h = prev_h, c = c, gates = gates}
end
-- d(next_c), d(next_h) -> d x, d prev_c, d prev_h; accumulates the four parameter gradients of the layer (model.lua:654-661,668-690)
function M.lstm_cell_backward(i2h, h2h, cache, dc, dh)
    local B, H, inp = cache.c.sizes[1], cache.c.sizes[2], i2h.weight:size(2)
    local dz, dcp = M.tensor({B, 4 * H}), M.tensor({B, H})
    A.check(L.aocr_lstm_cell_backward(nil, dc:ptr(), dh:ptr(), cache.gates:ptr(), cache.prev_c:ptr(), cache.c:ptr(), dz:ptr(), dcp:ptr(), B, H), 'aocr_lstm_cell_backward')
    local dx, dhp = M.tensor({B, inp}), M.tensor({B, H})
    A.check(L.aocr_gemm(nil, M.compute, dz:ptr(), 4 * H, 1, dev_param(i2h, 'weight'):ptr(), inp, 0, dx:ptr(), inp, B, inp, 4 * H, nil, 0), 'aocr_gemm')
    A.check(L.aocr_gemm(nil, M.compute, dz:ptr(), 4 * H, 1, dev_param(h2h, 'weight'):ptr(), H, 0, dhp:ptr(), H, B, H, 4 * H, nil, 0), 'aocr_gemm')
    A.check(L.aocr_gemm(nil, M.compute, dz:ptr(), 4 * H, 0, cache.x:ptr(), inp, 0, dev_grad(i2h, 'gradWeight'):ptr(), inp, 4 * H, inp, B, nil, 1), 'aocr_gemm')
    A.check(L.aocr_gemm(nil, M.compute, dz:ptr(), 4 * H, 0, cache.prev_h:ptr(), H, 0, dev_grad(h2h, 'gradWeight'):ptr(), H, 4 * H, H, B, nil, 1), 'aocr_gemm')
    M._ones = (M._ones and M._ones.n >= B) and M._ones or M.from_host(torch.FloatTensor(math.max(B, 4096)):fill(1))
    for _, lin in ipairs({i2h, h2h}) do                                          -- both biases receive the column sums of dz (LSTM.lua:86-88)
        A.check(L.aocr_gemm(nil, 0, dz:ptr(), 4 * H, 0, M._ones:ptr(), 1, 0, dev_grad(lin, 'gradBias'):ptr(), 1, 4 * H, 1, B, nil, 1), 'aocr_gemm')
    end
    return dx, dcp, dhp
end
-- create_decoder_attn (LSTM.lua:124-162): wa = LinearNoBias(H,H), wc = LinearNoBias(2H,H); h_top (B,H), context (B,T,H) DeviceTensors
function M.attention_forward(wa, wc, h_top, context)
    local B, T, H = context.sizes[1], context.sizes[2], context.sizes[3]
    local q, a, cat, out = M.tensor({B, H}), M.tensor({B, T}), M.tensor({B, 2 * H}), M.tensor({B, H})
    A.check(L.aocr_gemm(nil, M.compute, h_top:ptr(), H, 1, dev_param(wa, 'weight'):ptr(), H, 1, q:ptr(), H, B, H, H, nil, 0), 'aocr_gemm')                      -- :131
    A.check(L.aocr_attention_forward(nil, context:ptr(), q:ptr(), a:ptr(), cat:ptr(), 2 * H, B, T, H), 'aocr_attention_forward')                             -- :135-150, c -> JoinTable slot 1
    for b = 0, B - 1 do A.hip_ok(A.hip.hipMemcpy(cat:ptr(b * 2 * H + H), h_top:ptr(b * H), H * 4, A.D2D), 'hipMemcpy D2D') end                                -- :153
    A.check(L.aocr_gemm(nil, M.compute, cat:ptr(), 2 * H, 1, dev_param(wc, 'weight'):ptr(), 2 * H, 1, out:ptr(), H, B, H, 2 * H, nil, 4), 'aocr_gemm')          -- :155-157 Linear + Tanh
    wa._dirty, wc._dirty = false, false
    return out, {q = q, a = a, cat = cat, out = out, h_top = h_top, context = context}
end
-- d(out) -> d h_top, and d(context) ACCUMULATED into dcontext (model.lua:652-653); accumulates gradWeight of W_a, W_c
function M.attention_backward(wa, wc, cache, dout, dcontext)
    local B, T, H = cache.context.sizes[1], cache.context.sizes[2], cache.context.sizes[3]
    local dpre, dcat, ds, dq, dh = M.tensor({B, H}), M.tensor({B, 2 * H}), M.tensor({B, T}), M.tensor({B, H}), M.tensor({B, H})
    A.check(L.aocr_pointwise(nil, 1, dout:ptr(), cache.out:ptr(), dpre:ptr(), B * H), 'aocr_pointwise')                                                       -- Tanh backward
    A.check(L.aocr_gemm(nil, M.compute, dpre:ptr(), H, 0, cache.cat:ptr(), 2 * H, 0, dev_grad(wc, 'gradWeight'):ptr(), 2 * H, H, 2 * H, B, nil, 1), 'aocr_gemm')
    A.check(L.aocr_gemm(nil, M.compute, dpre:ptr(), H, 1, dev_param(wc, 'weight'):ptr(), 2 * H, 0, dcat:ptr(), 2 * H, B, 2 * H, H, nil, 0), 'aocr_gemm')
    A.check(L.aocr_attention_backward(nil, cache.context:ptr(), cache.q:ptr(), cache.a:ptr(), dcat:ptr(), 2 * H, ds:ptr(), dq:ptr(), B, T, H), 'aocr_attention_backward')
    A.check(L.aocr_gemm(nil, M.compute, dq:ptr(), H, 0, cache.h_top:ptr(), H, 0, dev_grad(wa, 'gradWeight'):ptr(), H, H, H, B, nil, 1), 'aocr_gemm')
    A.check(L.aocr_gemm(nil, M.compute, dq:ptr(), H, 1, dev_param(wa, 'weight'):ptr(), H, 0, dh:ptr(), H, B, H, H, nil, 0), 'aocr_gemm')                        -- d h_top through q
    for b = 0, B - 1 do                                                                                                                                      -- + the JoinTable half
        A.check(L.aocr_pointwise(nil, 0, dh:ptr(b * H), dcat:ptr(b * 2 * H + H), dh:ptr(b * H), H), 'aocr_pointwise')
    end
    -- d(context)[b][t][:] += a[b][t] * dc[b][:] + ds[b][t] * q[b][:]   (two rank-1 updates per row, as products with K = 1)
    for b = 0, B - 1 do
        A.check(L.aocr_gemm(nil, 0, cache.a:ptr(b * T), 1, 1, dcat:ptr(b * 2 * H), 1, 1, dcontext:ptr(b * T * H), H, T, H, 1, nil, 1), 'aocr_gemm')
        A.check(L.aocr_gemm(nil, 0, ds:ptr(b * T), 1, 1, cache.q:ptr(b * H), 1, 1, dcontext:ptr(b * T * H), H, T, H, 1, nil, 1), 'aocr_gemm')
    end
    return dh
end
-- nn.LogSoftMax + ClassNLLCriterion(weights; PAD weight 0; sizeAverage false) (output_projector.lua:6, criterion.lua:3-9, model.lua:644-648)
-- logits (rows, V) DeviceTensor, targets IntTensor (rows) 1-based; returns the loss (a Lua number: the step's host sync) and d(logits)
function M.criterion(logits, targets, grad_scale)
    local rows, V = logits.sizes[1], logits.sizes[2]
    local ids = A.device_bytes(rows * 4); A.upload(ids, targets:int():contiguous(), rows * 4)
    local nll, dlog = M.tensor({rows}), M.tensor({rows, V})
    A.check(L.aocr_logsoftmax_nll(nil, logits:ptr(), V, ids:as('int32_t*'), nil, dlog:ptr(), nll:ptr(), rows, V, grad_scale or 1), 'aocr_logsoftmax_nll')
    return nll:float():sum(), dlog
end

function M.install()
    assert(cudnn and cudnn.SpatialConvolution, "require 'cudnn' (lua/cudnn.lua) first")
    install_cnn(); install_linear()
    return M
end
-- mark every module of a net "host copy changed" (after optim.sgd_list wrote the flat parameter vector)
function M.sync(net) net:apply(function(m) m._dirty = true end) end
return M
