--[[ model.lua -- drop-in replacement of the reference's src/model/model.lua (class `Model`, lines 18-731) for MI355X.

     Put this directory first on LUA_PATH (`LUA_PATH="<repo>/lua/?.lua;;" th src/train.lua ...`): the unchanged src/train.lua then
     resolves `require 'model'`, 'cudnn', 'cutorch', 'cunn', 'hdf5' here (train.lua:4-12,244-245) and drives the same surface --
         Model(), model:create(opt), model:load(path, opt), model:step(batch, forward_only, beam_size, trie) -> loss*batch_size,
         {num_nonzeros, accuracy}, model:save(path), model:vis(dir), model:shutdown(), model.global_step,
         model.optim_state.learningRate            (train.lua:79-215, 256-293)
     -- while every FLOP of feval + optim.sgd_list (model.lua:284-706, optim_sgd.lua:38-95) runs in libaocr's HIP kernels through
     the LuaJIT FFI (lua/aocr_ffi.lua).  What stays in Lua is what the reference keeps in Lua: flags, logging, checkpoint cadence.

     The five nets of the reference (createCNNModel / createLSTM / createOutputUnit, built by the reference's own cnn.lua, LSTM.lua,
     output_projector.lua on the CPU) are kept as PARAMETER CONTAINERS only: fresh parameters are Torch7's own module
     initialisation, and model:save / model:load write and read exactly the table the reference serializes
     ({{cnn, enc_fw, enc_bw, decoder, projector}, config, global_step, optim_state}, model.lua:720-725 / :45-80), so checkpoints
     travel both ways.  They are never run.

     Unexecuted in the build container (no Lua / Torch7 there); tests/abi_harness.cc performs this file's call sequence against
     the C ABI with no host framework in the process. ]]
-- the packages train.lua expects its model file to have loaded (model.lua:2-16), then the reference's own builders of the five nets
for _, pkg in ipairs({'nn', 'nngraph', 'hdf5', 'cudnn', 'optim', 'paths'}) do require(pkg) end
package.path = package.path .. ';src/?.lua;src/utils/?.lua;src/model/?.lua;src/optim/?.lua'
for _, pkg in ipairs({'cnn', 'LSTM', 'output_projector', 'criterion', 'model_utils', 'memory'}) do require(pkg) end

local ffi = require 'ffi'
local A = require 'aocr_ffi'
local Dict = require 'dictionary'

local model = torch.class('Model')

function model:__init()
    -- the global `log` of the reference (model.lua:19-25): the logger train.lua created, else print
    log = (logging ~= nil) and function(msg) logging:info(msg) end or print
end

-- ------------------------------------------------------------------------------------------------ parameter containers
-- i2h / h2h of every layer of an LSTM gModule, bottom layer first: the nn.CAddTable fed by two nn.Linear is
-- CAddTable()({i2h, h2h}) of LSTM.lua:86-88 (first input i2h, second h2h); forwardnodes are in topological order, and layer L+1
-- depends on layer L [upstream nngraph: node.data.module, node.data.mapindex[i] = data of the i-th input node].
local function lstm_linears(gmod)
    local layers = {}
    for _, node in ipairs(gmod.forwardnodes) do
        local m = node.data.module
        if m and torch.type(m) == 'nn.CAddTable' and node.data.mapindex and #node.data.mapindex == 2 then
            local a, b = node.data.mapindex[1].module, node.data.mapindex[2].module
            if a and b and torch.type(a) == 'nn.Linear' and torch.type(b) == 'nn.Linear' then
                table.insert(layers, {i2h = a, h2h = b})
            end
        end
    end
    return layers
end
local function find_modules(gmod, typename)
    local out = {}
    for _, node in ipairs(gmod.forwardnodes) do
        local m = node.data.module
        if m and torch.type(m) == typename then table.insert(out, m) end
    end
    return out
end
local function attention_linears(decoder, H)
    local attn
    for _, node in ipairs(decoder.forwardnodes) do
        local m = node.data.module
        if m and torch.type(m) == 'nn.gModule' then attn = m end          -- create_decoder_attn, LSTM.lua:113-115
    end
    assert(attn, 'decoder has no attention graph')
    local wa, wc
    for _, m in ipairs(find_modules(attn, 'nn.LinearNoBias')) do
        if m.weight:size(2) == H then wa = m else wc = m end               -- (H x H) = W_a, LSTM.lua:131; (H x 2H) = W_c, :155
    end
    return wa, wc
end

-- name -> tensor of the containers, in the names of aocr_param_entry ("cnn.conv3.w", "enc_fw.l1.i2h.b", "dec.attn.wa", ...)
function model:_named_tensors(field)      -- field: 'weight'/'bias' or 'gradWeight'/'gradBias'
    local w, b = field, (field == 'weight') and 'bias' or 'gradBias'
    local t = {}
    local ci = 0
    for _, m in ipairs(self.cnn_model.modules) do
        local ty = torch.type(m)
        if ty == 'cudnn.SpatialConvolution' or ty == 'nn.SpatialConvolution' or ty == 'nn.SpatialConvolutionMM' then
            ci = ci + 1
            t['cnn.conv' .. ci .. '.w'] = m[w]; t['cnn.conv' .. ci .. '.b'] = m[b]
        elseif ty == 'nn.SpatialBatchNormalization' then
            t['cnn.bn' .. ci .. '.w'] = m[w]; t['cnn.bn' .. ci .. '.b'] = m[b]
            t['cnn.bn' .. ci .. '.rm'] = m.running_mean; t['cnn.bn' .. ci .. '.rv'] = m.running_var
        end
    end
    local function lstm(prefix, gmod)
        for l, pair in ipairs(lstm_linears(gmod)) do
            t[prefix .. '.l' .. l .. '.i2h.w'] = pair.i2h[w]; t[prefix .. '.l' .. l .. '.i2h.b'] = pair.i2h[b]
            t[prefix .. '.l' .. l .. '.h2h.w'] = pair.h2h[w]; t[prefix .. '.l' .. l .. '.h2h.b'] = pair.h2h[b]
        end
    end
    lstm('enc_fw', self.encoder_fw); lstm('enc_bw', self.encoder_bw); lstm('dec', self.decoder)
    t['dec.lookup'] = find_modules(self.decoder, 'nn.LookupTable')[1][w]
    local wa, wc = attention_linears(self.decoder, self.decoder_num_hidden)
    t['dec.attn.wa'] = wa[w]; t['dec.attn.wc'] = wc[w]
    local lin = self.output_projector.modules[1]                          -- output_projector.lua:5
    t['proj.w'] = lin[w]; t['proj.b'] = lin[b]
    return t
end

-- containers -> flat host vector in the library's layout (conv taps channels-last [Cout][kH][kW][Cin]) -> HBM
function model:_nets_to_device()
    local named = self:_named_tensors('weight')
    local host = torch.FloatTensor(self.num_params):zero()
    for _, e in ipairs(self.table) do
        local src = named[e.name]
        assert(src, 'parameter ' .. e.name .. ' not found in the nets')
        src = src:float()
        if #e.shape == 4 then
            src = src:view(e.shape[1], e.shape[4], e.shape[2], e.shape[3]):permute(1, 3, 4, 2):contiguous()   -- [Cout][Cin][kH][kW] -> channels-last
        end
        assert(src:nElement() == e.numel, 'size mismatch for ' .. e.name)
        host:narrow(1, e.offset + 1, e.numel):copy(src:contiguous():view(-1))
    end
    A.upload(self.params_dev, host, self.num_params * 4)
    local bn = torch.FloatTensor(tonumber(A.lib.aocr_bn_state_count()))
    local o = 0
    for _, i in ipairs({3, 5, 7}) do
        local rm, rv = named['cnn.bn' .. i .. '.rm']:float(), named['cnn.bn' .. i .. '.rv']:float()
        bn:narrow(1, o + 1, rm:nElement()):copy(rm); o = o + rm:nElement()
        bn:narrow(1, o + 1, rv:nElement()):copy(rv); o = o + rv:nElement()
    end
    A.upload(self.bn_dev, bn, bn:nElement() * 4)
end
-- HBM -> containers (before model:save)
function model:_device_to_nets()
    local named = self:_named_tensors('weight')
    local host = torch.FloatTensor(self.num_params)
    A.download(host, self.params_dev, self.num_params * 4)
    for _, e in ipairs(self.table) do
        local v = host:narrow(1, e.offset + 1, e.numel)
        local dst = named[e.name]
        if #e.shape == 4 then
            v = v:view(e.shape[1], e.shape[2], e.shape[3], e.shape[4]):permute(1, 4, 2, 3):contiguous()       -- channels-last -> [Cout][Cin][kH][kW]
        end
        dst:copy(v:view(dst:size()):typeAs(dst))
    end
    local bn = torch.FloatTensor(tonumber(A.lib.aocr_bn_state_count()))
    A.download(bn, self.bn_dev, bn:nElement() * 4)
    local o = 0
    for _, i in ipairs({3, 5, 7}) do
        local rm, rv = named['cnn.bn' .. i .. '.rm'], named['cnn.bn' .. i .. '.rv']
        rm:copy(bn:narrow(1, o + 1, rm:nElement()):typeAs(rm)); o = o + rm:nElement()
        rv:copy(bn:narrow(1, o + 1, rv:nElement()):typeAs(rv)); o = o + rv:nElement()
    end
end

-- ------------------------------------------------------------------------------------------------ model.lua:45-142
-- The reference spells the configuration out field by field three times (load, create, the log block + the `config` copy of
-- _build, model.lua:63-80, 86-98, 115-142).  Here it is ONE table: which fields the saved parameters fix, which a caller may
-- override at load time (model.lua:72-75), and the line each one logs.  Field names, the `config` table a checkpoint carries and the
-- log strings (typo of model.lua:115 included: downstream log parsers see the same text) are the reference's.
local STRUCTURAL = {'dropout', 'encoder_num_hidden', 'encoder_num_layers', 'decoder_num_layers', 'target_vocab_size',
                    'target_embedding_size', 'input_feed'}
local RUNTIME = {'max_encoder_l', 'max_decoder_l', 'batch_size'}
local LOGGED = {      -- {field, format}; order of model.lua:115-127
    {'cnn_feature_size', 'cnn_featuer_size: %d'}, {'dropout', 'dropout: %f'}, {'encoder_num_hidden', 'encoder_num_hidden: %d'},
    {'encoder_num_layers', 'encoder_num_layers: %d'}, {'decoder_num_hidden', 'decoder_num_hidden: %d'},
    {'decoder_num_layers', 'decoder_num_layers: %d'}, {'target_vocab_size', 'target_vocab_size: %d'},
    {'target_embedding_size', 'target_embedding_size: %d'}, {'max_encoder_l', 'max_encoder_l: %d'},
    {'max_decoder_l', 'max_decoder_l: %d'}, {'input_feed', 'input_feed: %s'}, {'batch_size', 'batch_size: %d'},
    {'prealloc', 'prealloc: %s'},
}
local SAVED = {'dropout', 'encoder_num_hidden', 'encoder_num_layers', 'decoder_num_hidden', 'decoder_num_layers', 'target_vocab_size',
               'target_embedding_size', 'max_encoder_l', 'max_decoder_l', 'input_feed', 'batch_size', 'prealloc'}   -- model.lua:130-142

-- structural fields from `fixed`; run-time fields from `caller` first, then from `recorded` (nil for a fresh model)
local function adopt_config(self, fixed, caller, recorded)
    for _, k in ipairs(STRUCTURAL) do self[k] = fixed[k] end
    for _, k in ipairs(RUNTIME) do
        local v = caller[k]
        if v == nil and recorded then v = recorded[k] end
        self[k] = v
    end
    self.cnn_feature_size = 512                                   -- conv7's maps, cnn.lua:39
    self.decoder_num_hidden = 2 * self.encoder_num_hidden         -- model.lua:66,88: the decoder state is [h_fw ; h_bw]
    self.prealloc = caller.prealloc
    self.seed = caller.seed or 910820
    preallocateMemory(caller.prealloc)
end

function model:load(model_path, config)
    config = config or {}
    assert(paths.filep(model_path), string.format('Model %s does not exist!', model_path))
    local nets, saved_config, step, optim_state = unpack(torch.load(model_path), 1, 4)     -- the tuple of model.lua:720-725
    adopt_config(self, saved_config, config, saved_config)
    self.layers = {}
    for i = 1, 5 do self.layers[i] = nets[i]:double() end                                  -- parameter containers only, never run
    self.global_step = step
    self.optim_state = optim_state or {}
    if self.optim_state.learningRate == nil then self.optim_state.learningRate = config.learning_rate end   -- train.lua:87
    self:_build()
end

function model:create(config)
    adopt_config(self, config, config, nil)
    -- the reference's own constructors: fresh parameters are Torch7's module initialisation (nn.Linear / SpatialConvolution reset(),
    -- LookupTable N(0,1), BatchNorm weight U(0,1))
    local F, He, Hd = self.cnn_feature_size, self.encoder_num_hidden, self.decoder_num_hidden
    local function encoder(name)                                   -- model.lua:101-102
        return createLSTM(F, He, self.encoder_num_layers, self.dropout, false, false, false, nil, self.batch_size, self.max_encoder_l, name)
    end
    self.layers = {
        createCNNModel(), encoder('encoder-fw'), encoder('encoder-bw'),
        createLSTM(self.target_embedding_size, Hd, self.decoder_num_layers, self.dropout, true, self.input_feed, true,
                   self.target_vocab_size, self.batch_size, self.max_encoder_l, 'decoder'),              -- model.lua:103
        createOutputUnit(Hd, self.target_vocab_size),
    }
    self.global_step = 0
    self.optim_state = {learningRate = config.learning_rate}
    self:_build()
end

-- ------------------------------------------------------------------------------------------------ model.lua:115-223
function model:_build()
    self.cnn_model, self.encoder_fw, self.encoder_bw, self.decoder, self.output_projector = unpack(self.layers, 1, 5)
    self.config = {}
    for _, e in ipairs(LOGGED) do log(string.format(e[2], (e[2]:sub(-1) == 's') and tostring(self[e[1]]) or self[e[1]])) end
    for _, k in ipairs(SAVED) do self.config[k] = self[k] end
    self.optim_state = self.optim_state or {}

    -- the widest crop the workspace is sized for: T = W/4 - 1 <= max_encoder_l (the reference clones max_encoder_l cells, model.lua:172-173)
    self.max_img_w = 4 * (self.max_encoder_l + 1)
    self.ccfg = A.make_config({batch_size = self.batch_size, max_img_w = self.max_img_w, encoder_num_hidden = self.encoder_num_hidden,
        encoder_num_layers = self.encoder_num_layers, decoder_num_layers = self.decoder_num_layers,
        target_vocab_size = self.target_vocab_size, target_embedding_size = self.target_embedding_size, input_feed = self.input_feed,
        max_decoder_l = self.max_decoder_l, max_beam = math.min(self.target_vocab_size, (opt and opt.beam_size) or 5),
        compute = os.getenv('AOCR_COMPUTE') or 'bf16'})
    self.table = A.param_table(self.ccfg)
    local counts = ffi.new('int64_t[5]')
    A.check(A.lib.aocr_param_counts(self.ccfg, counts), 'aocr_param_counts')
    self.num_params = 0
    for g = 0, 4 do self.num_params = self.num_params + tonumber(counts[g]) end
    log(string.format('Number of parameters: %d', self.num_params))

    -- device memory: the library allocates nothing (include/aocr.h): parameters, gradients, running statistics and one workspace arena
    local ws = tonumber(A.lib.aocr_workspace_bytes(self.ccfg))
    assert(ws > 0, ffi.string(A.lib.aocr_last_error()))
    self.params_dev = A.device_bytes(self.num_params * 4)
    self.grads_dev = A.device_bytes(self.num_params * 4); self.grads_dev:zero()
    self.bn_dev = A.device_bytes(tonumber(A.lib.aocr_bn_state_count()) * 4)
    self.ws_dev = A.device_bytes(ws)
    self.scal_dev = A.device_bytes(64 * 4)                                     -- loss + per-group norms
    local B, Lt, Wm = self.batch_size, self.max_decoder_l, self.max_img_w
    self.images_dev = A.device_bytes(B * 32 * Wm * 4)
    self.targets_dev = A.device_bytes(B * Lt * 4); self.targets_eval_dev = A.device_bytes(B * Lt * 4)
    self.labels_dev = A.device_bytes(B * Lt * 4); self.scores_dev = A.device_bytes(B * 4); self.gold_dev = A.device_bytes(B * 4)
    self.dist_dev = A.device_bytes(B * 4); self.tlen_dev = A.device_bytes(B * 4); self.tge_pad_dev = A.device_bytes(B * Lt * 4)
    local h = ffi.new('aocr_model*[1]')
    A.check(A.lib.aocr_model_create(self.ccfg, self.params_dev:as('float*'), self.grads_dev:as('float*'), self.bn_dev:as('float*'),
                                    self.ws_dev.ptr, ws, nil, h), 'aocr_model_create')
    self.handle = h[0]
    self:_nets_to_device()
    -- data parallelism (one process per GPU, launched with AOCR_RANK / AOCR_WORLD_SIZE / AOCR_COMM_ID_FILE): the one exchange step of
    -- the path, RCCL all-reduce of the flat gradient vector between feval and the per-group clip (optim_sgd.lua:38 -> :40)
    self.world = tonumber(os.getenv('AOCR_WORLD_SIZE') or '1')
    if self.world > 1 then
        local rank = tonumber(os.getenv('AOCR_RANK'))
        self.rank = rank
        local path = assert(os.getenv('AOCR_COMM_ID_FILE'), 'AOCR_COMM_ID_FILE not set')
        local id = ffi.new('char[128]')
        if rank == 0 then
            A.check(A.lib.aocr_comm_unique_id(id), 'aocr_comm_unique_id')
            local f = assert(io.open(path .. '.tmp', 'wb')); f:write(ffi.string(id, 128)); f:close(); os.rename(path .. '.tmp', path)
        else
            local f
            repeat f = io.open(path, 'rb') until f
            ffi.copy(id, f:read(128), 128); f:close()
        end
        A.check(A.lib.aocr_comm_init_rank(self.handle, id, self.world, rank, 1), 'aocr_comm_init_rank')
    end
    self.init_beam = false
    self.visualize = false
    self.trie_cache = nil
end

-- ------------------------------------------------------------------------------------------------ model.lua:226-706
function model:step(batch, forward_only, beam_size, trie)
    local input_batch = batch[1]:float():contiguous()                         -- (B,1,32,W) values 0..255 (data_gen.lua:120)
    local target_batch = batch[2]:int():contiguous()
    local target_eval_batch = batch[3]:int():contiguous()
    local num_nonzeros = batch[4]
    local img_paths
    if self.visualize then img_paths = batch[5] end
    local batch_size = input_batch:size(1)
    local W = input_batch:size(4)
    local target_l = target_batch:size(2)
    assert(target_l <= self.max_decoder_l, string.format('max_decoder_l (%d) < target_l (%d)!', self.max_decoder_l, target_l))
    A.upload(self.images_dev, input_batch, batch_size * 32 * W * 4)
    A.upload(self.targets_dev, target_batch, batch_size * target_l * 4)
    A.upload(self.targets_eval_dev, target_eval_batch, batch_size * target_l * 4)
    local L = A.lib
    if not forward_only then
        -- feval with training-mode BatchNorm (model.lua:276-278), then optim.sgd_list: clip each of the 5 groups to 5, x -= lr * g
        if self.dropout and self.dropout > 0 then      -- LSTM.lua:68-69,116-118: masks = f(seed, global_step, site, element), include/aocr.h
            -- (rank-dependent seed under data parallelism: every rank drops different units of its slice, as aocr/model.py does)
            A.check(L.aocr_set_dropout(self.handle, self.dropout, (cutorch._seed or 910820) + 7919 * (self.rank or 0), self.global_step), 'aocr_set_dropout')
        end
        A.check(L.aocr_train_forward_backward(self.handle, self.images_dev:as('float*'), self.targets_dev:as('int32_t*'),
                                              self.targets_eval_dev:as('int32_t*'), batch_size, W, target_l,
                                              1.0 / (batch_size * self.world), self.scal_dev:as('float*')), 'aocr_train_forward_backward')
        if self.world > 1 then A.check(L.aocr_allreduce_grads(self.handle, self.scal_dev:as('float*')), 'aocr_allreduce_grads') end
        A.check(L.aocr_sgd_step(self.handle, self.optim_state.learningRate, 5.0, self.scal_dev:as('float*') + 2), 'aocr_sgd_step')
        local loss = A.read_scalar(self.scal_dev, 0)                          -- = loss * batch_size of model.lua:701
        -- health of the whole-sequence kernels (include/aocr.h: aocr_cluster_status).  A timed-out wait invalidates the step; the
        -- library then skips the update itself (aocr_sgd_step's device-side predicate), so the weights are intact: report and redo.
        -- With AOCR_WORLD_SIZE > 1 the code is the same decision on every rank (the flag is summed with the gradient exchange; a rank whose own
        -- kernels were healthy reads 0x7e), so all ranks repeat together and their collectives pair up; the library keeps the BatchNorm running
        -- statistics as they are during the repeat (include/aocr.h: aocr_cluster_status).
        local code = ffi.new('int32_t[1]')
        A.check(L.aocr_cluster_status(self.handle, code), 'aocr_cluster_status')
        if code[0] ~= 0 then
            self.redo = (self.redo or 0) + 1
            assert(self.redo <= 3, string.format('cluster kernel timed out (code %d) three times in a row: is another process using this GPU?', code[0]))
            log(string.format('Warning: a cluster kernel timed out waiting for its group (code %d); the step is repeated', code[0]))
            return self:step(batch, forward_only, beam_size, trie)
        end
        self.redo = 0
        return loss, {num_nonzeros, 0.0}
    end
    -- forward only: beam search over max_decoder_l steps + gold pass (model.lua:321-627)
    beam_size = beam_size or 1
    beam_size = math.min(beam_size, self.target_vocab_size)
    local Lt = self.max_decoder_l
    local tdesc = nil
    if trie ~= nil then
        if self.trie_cache == nil or self.trie_cache.source ~= trie then self.trie_cache = Dict.flatten(trie, A) end   -- once per dictionary
        tdesc = self.trie_cache.desc
    end
    A.check(L.aocr_decode_dict(self.handle, self.images_dev:as('float*'), self.targets_dev:as('int32_t*'), self.targets_eval_dev:as('int32_t*'),
                               batch_size, W, target_l, beam_size, tdesc, self.labels_dev:as('int32_t*'), self.scores_dev:as('float*'),
                               self.gold_dev:as('float*'), self.scal_dev:as('float*')), 'aocr_decode_dict')
    -- evalWordErrRate (utils.lua:136-175) on the device: a word is right iff its edit distance to the target (both cut at EOS) is 0
    local tge_pad = torch.IntTensor(batch_size, Lt):fill(1)
    tge_pad[{{1, batch_size}, {1, target_l}}]:copy(target_eval_batch)
    A.upload(self.tge_pad_dev, tge_pad, batch_size * Lt * 4)
    A.check(L.aocr_edit_distance(nil, self.labels_dev:as('int32_t*'), self.tge_pad_dev:as('int32_t*'), batch_size, Lt,
                                 self.dist_dev:as('int32_t*'), self.tlen_dev:as('int32_t*')), 'aocr_edit_distance')
    local dist = torch.IntTensor(batch_size); A.download(dist, self.dist_dev, batch_size * 4)
    local accuracy = batch_size - dist:ne(0):sum()
    if self.visualize then
        local labels = torch.IntTensor(batch_size, Lt); A.download(labels, self.labels_dev, batch_size * Lt * 4)
        local scores = torch.FloatTensor(batch_size); A.download(scores, self.scores_dev, batch_size * 4)
        local gold = torch.FloatTensor(batch_size); A.download(gold, self.gold_dev, batch_size * 4)
        local function cut(row)                                               -- up to the first EOS, as evalWordErrRate does
            local t = {}
            for i = 1, row:size(1) do if row[i] == 3 then break end; table.insert(t, row[i]) end
            return numlist2str(t)
        end
        for i = 1, #img_paths do                                              -- model.lua:628-633
            self.visualize_file:write(string.format('%s\t%s\t%s\t%f\t%f\n', img_paths[i], cut(tge_pad[i]), cut(labels[i]), scores[i], gold[i]))
        end
        self.visualize_file:flush()
    end
    local loss = A.read_scalar(self.scal_dev, 0)
    return loss, {num_nonzeros, accuracy}
end

-- ------------------------------------------------------------------------------------------------ model.lua:708-731
function model:vis(output_dir)
    local path = paths.concat(output_dir, 'results.txt')
    local file = io.open(path, 'w')
    self.visualize_path, self.visualize_file, self.visualize = path, file, file ~= nil
    if not file then log(string.format('Error: visualize file %s cannot be created', path)) end
end

function model:save(model_path)
    self:_device_to_nets()
    for _, net in ipairs(self.layers) do net:clearState() end
    torch.save(model_path, {self.layers, self.config, self.global_step, self.optim_state})   -- {5 nets, config, step, optim state}: what model:load reads
end

local DEVICE_BUFFERS = {'params_dev', 'grads_dev', 'bn_dev', 'ws_dev', 'scal_dev', 'images_dev', 'targets_dev', 'targets_eval_dev', 'labels_dev',
                        'scores_dev', 'gold_dev', 'dist_dev', 'tlen_dev', 'tge_pad_dev'}
function model:shutdown()
    if self.visualize_file then self.visualize_file:close(); self.visualize_file = nil end
    if self.handle ~= nil then A.lib.aocr_model_destroy(self.handle); self.handle = nil end
    for _, k in ipairs(DEVICE_BUFFERS) do
        if self[k] then self[k]:free(); self[k] = nil end
    end
end
