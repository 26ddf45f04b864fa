--[[ aocr_ffi.lua -- LuaJIT FFI binding of libaocr.so (include/aocr.h) and of the handful of HIP runtime calls a host needs
     (device memory, copies, synchronisation).  Everything the Lua side of the boundary uses goes through this file; there is
     no cutorch, no CUDA-compat header and no Torch7 GPU package underneath.

     Unexecuted in the build container (no Lua / LuaJIT / Torch7 there: SURVEY.md 8(c)); the same call sequence is exercised
     without any host framework by tests/abi_harness.cc, which is what pins the ABI's behaviour.

     Search order of the shared objects: $AOCR_LIB (full path) or libaocr.so on the loader path; libamdhip64.so of ROCm. ]]
local ffi = require 'ffi'

ffi.cdef[[
/* ---- HIP runtime subset (hip/hip_runtime_api.h) */
typedef int hipError_t;
typedef struct ihipStream_t* hipStream_t;
hipError_t hipSetDevice(int device);
hipError_t hipGetDeviceCount(int* count);
hipError_t hipMalloc(void** ptr, size_t size);
hipError_t hipFree(void* ptr);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, int kind);   /* 1 = host->device, 2 = device->host, 3 = device->device */
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipDeviceSynchronize(void);
hipError_t hipStreamCreate(hipStream_t* stream);
hipError_t hipStreamDestroy(hipStream_t stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
const char* hipGetErrorString(hipError_t e);

/* ---- include/aocr.h */
typedef struct aocr_model aocr_model;
typedef struct aocr_config {
  int32_t batch_size, img_h, max_img_w, enc_hidden, enc_layers, dec_layers, vocab, emb, input_feed, max_decoder_l, max_beam, compute;
} aocr_config;
typedef struct aocr_trie { const uint64_t* child_mask_dev; const int32_t* child_base_dev; const int32_t* child_dev; int32_t n_nodes, n_edges; } aocr_trie;
typedef struct aocr_image_desc { int64_t offset; int32_t height, width, channels, reserved; } aocr_image_desc;
const char* aocr_last_error(void);
int aocr_version(void);
int aocr_param_counts(const aocr_config* cfg, int64_t counts[5]);
int aocr_param_entry(const aocr_config* cfg, int32_t index, char name[64], int32_t* group, int64_t* offset, int32_t* ndim, int64_t shape[4]);
int64_t aocr_bn_state_count(void);
size_t aocr_workspace_bytes(const aocr_config* cfg);
int aocr_model_create(const aocr_config* cfg, float* params_dev, float* grads_dev, float* bn_state_dev, void* workspace_dev, size_t workspace_bytes, void* stream, aocr_model** out);
int aocr_model_destroy(aocr_model* m);
int aocr_model_set_stream(aocr_model* m, void* stream);
int aocr_set_dropout(aocr_model* m, double p, uint64_t seed, uint64_t train_step);
int aocr_cluster_status(aocr_model* m, int32_t* code);
int aocr_train_forward_backward(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, float grad_scale, float* loss_dev);
int aocr_sgd_step(aocr_model* m, float lr, float clip, float* norms_dev);
int aocr_decode_dict(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, int32_t beam, const aocr_trie* trie, int32_t* labels_dev, float* scores_dev, float* gold_scores_dev, float* loss_dev);
int aocr_edit_distance(void* stream, const int32_t* labels_dev, const int32_t* targets_dev, int32_t B, int32_t L, int32_t* dist_dev, int32_t* target_len_dev);
int aocr_preprocess_lines(void* stream, const uint8_t* src_dev, const aocr_image_desc* desc_dev, int32_t n_images, int32_t out_h, int32_t out_w, float* out_dev);
/* data parallelism inside the library: RCCL over xGMI (one process per GPU) */
int aocr_comm_unique_id(char id[128]);
int aocr_comm_init_rank(aocr_model* m, const char id[128], int32_t nranks, int32_t rank, int32_t sync_bn);
int aocr_allreduce_grads(aocr_model* m, float* loss_dev);
]]

local M = {}
M.hip = ffi.load(os.getenv('AOCR_HIP_LIB') or 'amdhip64')
M.lib = ffi.load(os.getenv('AOCR_LIB') or 'aocr')
M.H2D, M.D2H, M.D2D = 1, 2, 3

function M.hip_ok(e, what)
    if e ~= 0 then error(string.format('%s: %s', what or 'HIP', ffi.string(M.hip.hipGetErrorString(e)))) end
end
-- every ABI entry point returns 0 / non-zero + aocr_last_error(): non-zero becomes a Lua error, exactly like a failing nn call
function M.check(rc, what)
    if rc ~= 0 then error(string.format('%s: %s', what or 'aocr', ffi.string(M.lib.aocr_last_error()))) end
end

-- a device allocation owned by Lua (freed by the garbage collector or :free())
local Buffer = {}
Buffer.__index = Buffer
function M.device_bytes(nbytes)
    local p = ffi.new('void*[1]')
    M.hip_ok(M.hip.hipMalloc(p, math.max(nbytes, 16)), 'hipMalloc')
    local self = setmetatable({ptr = ffi.gc(p[0], M.hip.hipFree), nbytes = nbytes}, Buffer)
    return self
end
function Buffer:free() if self.ptr ~= nil then M.hip.hipFree(ffi.gc(self.ptr, nil)); self.ptr = nil end end
function Buffer:zero() M.hip_ok(M.hip.hipMemset(self.ptr, 0, self.nbytes), 'hipMemset') end
function Buffer:as(ctype) return ffi.cast(ctype, self.ptr) end

-- upload a contiguous torch tensor (FloatTensor / IntTensor / ByteTensor / LongTensor) or a cdata array
function M.upload(buf, src, nbytes, offset_bytes)
    local p = type(src) == 'cdata' and src or src:data()
    M.hip_ok(M.hip.hipMemcpy(ffi.cast('char*', buf.ptr) + (offset_bytes or 0), p, nbytes, M.H2D), 'hipMemcpy H2D')
end
function M.download(dst, buf, nbytes, offset_bytes)
    local p = type(dst) == 'cdata' and dst or dst:data()
    M.hip_ok(M.hip.hipMemcpy(p, ffi.cast('char*', buf.ptr) + (offset_bytes or 0), nbytes, M.D2H), 'hipMemcpy D2H')
end
-- one float back from the device (this is the host sync of a step, like criterion:forward returning a Lua number)
function M.read_scalar(buf, index)
    local v = ffi.new('float[1]')
    M.hip_ok(M.hip.hipMemcpy(v, ffi.cast('float*', buf.ptr) + (index or 0), 4, M.D2H), 'hipMemcpy D2H')
    return tonumber(v[0])
end
function M.make_config(t)
    local c = ffi.new('aocr_config')
    c.batch_size = t.batch_size; c.img_h = t.img_h or 32; c.max_img_w = t.max_img_w; c.enc_hidden = t.encoder_num_hidden
    c.enc_layers = t.encoder_num_layers; c.dec_layers = t.decoder_num_layers; c.vocab = t.target_vocab_size
    c.emb = t.target_embedding_size; c.input_feed = t.input_feed and 1 or 0; c.max_decoder_l = t.max_decoder_l
    c.max_beam = t.max_beam or 5; c.compute = (t.compute == 'f32') and 0 or 1
    return c
end
-- the parameter table of the library: { {name=, group=, offset=, shape={...}, numel=}, ... } in Torch7 getParameters() order
function M.param_table(cfg)
    local out, i = {}, 0
    local name, group, off = ffi.new('char[64]'), ffi.new('int32_t[1]'), ffi.new('int64_t[1]')
    local nd, shape = ffi.new('int32_t[1]'), ffi.new('int64_t[4]')
    while true do
        local rc = M.lib.aocr_param_entry(cfg, i, name, group, off, nd, shape)
        if rc == 1 then break end
        M.check(rc, 'aocr_param_entry')
        local e = {name = ffi.string(name), group = group[0] + 1, offset = tonumber(off[0]), shape = {}, numel = 1}
        for k = 0, nd[0] - 1 do e.shape[k + 1] = tonumber(shape[k]); e.numel = e.numel * tonumber(shape[k]) end
        table.insert(out, e)
        i = i + 1
    end
    return out
end
return M
