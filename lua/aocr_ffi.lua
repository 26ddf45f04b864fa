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

/* ---- include/aocr.h: every declaration of the header, comments stripped, array bounds resolved (tests/test_lua_cdef_cpu.py
   compares this block with the header declaration by declaration) */
typedef struct aocr_model aocr_model;
typedef struct aocr_config { int32_t batch_size; int32_t img_h; int32_t max_img_w; int32_t enc_hidden; int32_t enc_layers; int32_t dec_layers; int32_t vocab; int32_t emb; int32_t input_feed; int32_t max_decoder_l; int32_t max_beam; int32_t compute; } aocr_config;
const char* aocr_last_error(void);
int aocr_version(void);
int aocr_param_counts(const aocr_config* cfg, int64_t counts[AOCR_NUM_GROUPS]);
int aocr_param_entry(const aocr_config* cfg, int32_t index, char name[64], int32_t* group, int64_t* offset, int32_t* ndim, int64_t shape[4]);
int64_t aocr_bn_state_count(void);
size_t aocr_workspace_bytes(const aocr_config* cfg);
int aocr_model_create(const aocr_config* cfg, float* params_dev, float* grads_dev, float* bn_state_dev, void* workspace_dev, size_t workspace_bytes, void* stream, aocr_model** out);
int aocr_model_destroy(aocr_model* m);
int aocr_model_set_stream(aocr_model* m, void* stream);
int aocr_cluster_status(aocr_model* m, int32_t* code);
int aocr_set_dropout(aocr_model* m, double p, uint64_t seed, uint64_t train_step);
int aocr_train_forward_backward(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, float grad_scale, float* loss_dev);
int aocr_grad_buckets(const aocr_config* cfg, int64_t begin[4], int64_t end[4]);
int aocr_stream_wait_grads(aocr_model* m, int32_t bucket, void* stream);
typedef int (*aocr_allreduce_fn)(void* user, void* buf_dev, int64_t count, int32_t dtype, void* stream);
int aocr_comm_unique_id(char id[128]);
int aocr_comm_init_rank(aocr_model* m, const char id[128], int32_t nranks, int32_t rank, int32_t sync_bn);
int aocr_comm_set_callback(aocr_model* m, aocr_allreduce_fn fn, void* user, int32_t nranks, int32_t sync_bn);
int aocr_allreduce_grads(aocr_model* m, float* loss_dev);
int aocr_comm_destroy(aocr_model* m);
int aocr_comm_info(aocr_model* m, int32_t* nranks, int32_t* sync_bn, int32_t* provider);
int aocr_comm_exposed_ms(aocr_model* m, float* ms);
int aocr_sgd_step(aocr_model* m, float lr, float clip, float* norms_dev);
int aocr_adadelta_step(aocr_model* m, float rho, float eps, float weight_decay, float* state_dev);
int aocr_forward_logits(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, int32_t training, float* logits_dev, float* loss_dev);
int aocr_decode(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, int32_t beam, int32_t* labels_dev, float* scores_dev, float* gold_scores_dev, float* loss_dev);
typedef struct aocr_trie { const uint64_t* child_mask_dev; const int32_t* child_base_dev; const int32_t* child_dev; int32_t n_nodes, n_edges; } aocr_trie;
int aocr_decode_dict(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, int32_t beam, const aocr_trie* trie, int32_t* labels_dev, float* scores_dev, float* gold_scores_dev, float* loss_dev);
int aocr_get_tensor(aocr_model* m, const char* name, const void** ptr_dev, int32_t* ndim, int64_t shape[4]);
int aocr_profile_kernel(aocr_model* m, int32_t which, int32_t iters, float* ms_per_launch, double* flops_per_launch);
int aocr_profile_enable(aocr_model* m, int32_t on);
int aocr_profile_read(aocr_model* m, float ms[12], int32_t* marks);
int aocr_gemm(void* stream, int32_t compute, const float* A_dev, int64_t lda, int32_t a_kmajor, const float* B_dev, int64_t ldb, int32_t b_kmajor, float* C_dev, int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias_dev, int32_t accumulate);
int aocr_conv2d_forward(void* stream, int32_t compute, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, uint8_t* idx_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad, int32_t relu, int32_t pool);
int aocr_conv2d_backward_data(void* stream, int32_t compute, const float* dy_dev, const float* w_dev, float* dx_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad);
int aocr_conv2d_backward_filter(void* stream, int32_t compute, const float* x_dev, const float* dy_dev, float* dw_dev, float* dbias_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad);
int aocr_unpool_relu_backward(void* stream, const float* dpooled_dev, const float* pooled_dev, const uint8_t* idx_dev, float* dy_dev, int32_t B, int32_t Ho, int32_t Wo, int32_t C, int32_t pool);
int aocr_conv1_forward(void* stream, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int32_t B, int32_t H, int32_t W);
int aocr_conv1_backward(void* stream, const float* x_dev, const float* w_dev, const float* bias_dev, const float* dy_pooled_dev, float* dw_dev, float* dbias_dev, int32_t B, int32_t H, int32_t W);
int aocr_batchnorm_relu_forward(void* stream, const float* x_dev, float* y_dev, const float* weight_dev, const float* bias_dev, float* running_mean_dev, float* running_var_dev, float* save_dev, void* scratch_dev, int64_t rows, int32_t C, int32_t training, int32_t update_running, int32_t tb_rows);
int aocr_batchnorm_relu_backward(void* stream, const float* x_dev, const float* y_dev, const float* dA_dev, const float* weight_dev, const float* save_dev, float* dx_dev, float* dweight_dev, float* dbias_dev, void* scratch_dev, int64_t rows, int32_t C, int32_t tb_rows);
int aocr_lstm_cell_forward(void* stream, int32_t compute, const float* x_dev, int32_t in_size, const float* h_prev_dev, const float* c_prev_dev, const float* w_i2h_dev, const float* b_i2h_dev, const float* w_h2h_dev, const float* b_h2h_dev, float* c_dev, float* h_dev, float* gates_dev, int32_t B, int32_t H);
int aocr_lstm_cell_forward_zx(void* stream, int32_t compute, const float* zx_dev, int64_t ldzx, const float* h_prev_dev, const float* c_prev_dev, const float* w_h2h_dev, float* c_dev, float* h_dev, float* gates_dev, int32_t B, int32_t H);
int aocr_lstm_cell_backward(void* stream, const float* dc_dev, const float* dh_dev, const float* gates_dev, const float* c_prev_dev, const float* c_dev, float* dz_dev, float* dc_prev_dev, int32_t B, int32_t H);
int aocr_attention_forward(void* stream, const float* ctx_dev, const float* q_dev, float* a_dev, float* c_dev, int64_t ldc, int32_t B, int32_t T, int32_t Hd);
int aocr_attention_backward(void* stream, const float* ctx_dev, const float* q_dev, const float* a_dev, const float* dc_dev, int64_t lddc, float* ds_dev, float* dq_dev, int32_t B, int32_t T, int32_t Hd);
int aocr_pointwise(void* stream, int32_t op, const float* a_dev, const float* b_dev, float* y_dev, int64_t n);
int aocr_lookup_forward(void* stream, const float* weight_dev, const int32_t* ids_dev, float* out_dev, int32_t n, int32_t E);
int aocr_lookup_backward(void* stream, const float* grad_out_dev, const int32_t* ids_dev, float* grad_weight_dev, int32_t n, int32_t E, int32_t V);
int aocr_logsoftmax_nll(void* stream, const float* logits_dev, int64_t ld, const int32_t* targets_dev, float* logp_dev, float* dlogits_dev, float* nll_rows_dev, int64_t rows, int32_t V, float grad_scale);
int aocr_beam_select(void* stream, const float* logp_dev, const int32_t* prev_tok_dev, float* beam_scores_dev, int32_t* tokens_dev, int32_t* parents_dev, int32_t B, int32_t kin, int32_t kout, int32_t V);
int aocr_beam_select_dict(void* stream, const float* logp_dev, const int32_t* prev_tok_dev, float* beam_scores_dev, int32_t* tokens_dev, int32_t* parents_dev, int32_t B, int32_t kin, int32_t kout, int32_t V, const aocr_trie* trie, const int32_t* loc_in_dev, int32_t* loc_out_dev);
int aocr_edit_distance(void* stream, const int32_t* labels_dev, const int32_t* targets_dev, int32_t B, int32_t L, int32_t* dist_dev, int32_t* target_len_dev);
typedef struct aocr_image_desc { int64_t offset; int32_t height, width, channels, reserved; } aocr_image_desc;
int aocr_preprocess_lines(void* stream, const uint8_t* src_dev, const aocr_image_desc* desc_dev, int32_t n_images, int32_t out_h, int32_t out_w, float* out_dev);
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
