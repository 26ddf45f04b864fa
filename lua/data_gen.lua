--[[ data_gen.lua -- optional replacement of src/data/data_gen.lua (class DataGen, lines 15-154) that moves the per-image arithmetic --
     255 * rgb2y and image.scale to 32 x W (data_gen.lua:68-79) -- onto the GPU (aocr_preprocess_lines, bit-identical to torch/image's
     single-precision arithmetic: tests/test_data_gpu.py).  Same constructor, shuffle(), size() and nextBatch(batch_size) surface and the
     same batch table {images (B,1,32,W) float, targets, targets_eval, num_nonzeros, img_paths}.  JPEG / PNG decoding stays with the
     `image` package on the host (image.load, as in the reference).  The reference's own data_gen.lua also works unchanged with
     lua/model.lua; use this one when the loader is the bottleneck. ]]
require 'image'
require 'paths'
require 'utils'
local ffi = require 'ffi'
local A = require 'aocr_ffi'
local tds = require 'tds'

local DataGen = torch.class('DataGen')

-- list file: one "<image path> <label>" per line, looked for as given and then under data_base_dir (data_gen.lua:29-37)
local function open_list(base_dir, path)
    for _, candidate in ipairs({path, paths.concat(base_dir, path)}) do
        local f = io.open(candidate, 'r')
        if f then return f end
    end
    log(string.format('Error: Data file %s not found ', path))
    os.exit()
end

function DataGen:__init(data_base_dir, data_path, max_aspect_ratio, max_encoder_l_h, max_encoder_l_w, max_decoder_l)
    self.imgH, self.min_aspect_ratio = 32, 0.5
    self.data_base_dir, self.data_path = data_base_dir, data_path
    self.max_aspect_ratio = max_aspect_ratio or math.huge
    self.max_encoder_l_h, self.max_encoder_l_w = max_encoder_l_h or math.huge, max_encoder_l_w or math.huge
    self.max_decoder_l = max_decoder_l or math.huge
    self.lines = tds.Hash()                                            -- off the Lua heap, as in the reference (millions of lines)
    local n = 0
    for line in open_list(data_base_dir, data_path):lines() do
        n = n + 1
        local fields = split(line)
        self.lines[n] = tds.Vec({fields[1], fields[2]})
        if n % 1000000 == 0 then log(string.format('%d lines read', n)) end
    end
    self.cursor, self.buffer = 1, {}
end

function DataGen:shuffle() shuffle(self.lines) end
function DataGen:size() return #self.lines end

-- the images of one width bucket: decoded bytes back to back -> device -> (n,1,32,imgW) float 0..255, one launch
local function preprocess(items, imgW)
    local n, total = #items, 0
    for _, it in ipairs(items) do total = total + it.h * it.w * it.c end
    local src = ffi.new('uint8_t[?]', total)
    local desc = ffi.new('aocr_image_desc[?]', n)
    local o = 0
    for i, it in ipairs(items) do
        ffi.copy(src + o, it.bytes:data(), it.h * it.w * it.c)
        desc[i - 1].offset = o; desc[i - 1].height = it.h; desc[i - 1].width = it.w; desc[i - 1].channels = it.c; desc[i - 1].reserved = 0
        o = o + it.h * it.w * it.c
    end
    local src_dev, desc_dev, out_dev = A.device_bytes(total), A.device_bytes(n * ffi.sizeof('aocr_image_desc')), A.device_bytes(n * 32 * imgW * 4)
    A.upload(src_dev, src, total); A.upload(desc_dev, desc, n * ffi.sizeof('aocr_image_desc'))
    A.check(A.lib.aocr_preprocess_lines(nil, src_dev:as('uint8_t*'), desc_dev:as('aocr_image_desc*'), n, 32, imgW, out_dev:as('float*')), 'aocr_preprocess_lines')
    local images = torch.FloatTensor(n, 1, 32, imgW)
    A.download(images, out_dev, n * 32 * imgW * 4)
    src_dev:free(); desc_dev:free(); out_dev:free()
    return images
end

local function emit(bucket, imgW)
    local n = #bucket
    local max_target_length = -math.huge
    for i = 1, n do max_target_length = math.max(max_target_length, #bucket[i].label_list) end
    local targets = torch.IntTensor(n, max_target_length - 1):fill(1)            -- data_gen.lua:107-117
    local targets_eval = torch.IntTensor(n, max_target_length - 1):fill(1)
    local num_nonzeros, img_paths = 0, {}
    for i = 1, n do
        local ll = bucket[i].label_list
        num_nonzeros = num_nonzeros + #ll - 1
        for j = 1, #ll - 1 do targets[i][j] = ll[j]; targets_eval[i][j] = ll[j + 1] end
        img_paths[i] = bucket[i].path
    end
    return {preprocess(bucket, imgW), targets, targets_eval, num_nonzeros, img_paths}
end

function DataGen:nextBatch(batch_size)
    while true do
        if self.cursor > #self.lines then break end
        local img_path = self.lines[self.cursor][1]
        local status, img = pcall(image.load, paths.concat(self.data_base_dir, img_path), nil, 'byte')   -- decoded bytes CHW, data_gen.lua:67
        if status then
            local c, h, w = img:size(1), img:size(2), img:size(3)
            local label_list = str2numlist(self.lines[self.cursor][2])
            self.cursor = self.cursor + 1
            local aspect_ratio = math.max(math.min(w / h, self.max_aspect_ratio), self.min_aspect_ratio)
            local imgW = math.ceil(aspect_ratio * self.imgH)
            imgW = 100                                                          -- data_gen.lua:78 (every crop is forced to 32 x 100)
            local hwc = img:permute(2, 3, 1):contiguous()                       -- interleaved HWC, the layout aocr_preprocess_lines reads
            if self.buffer[imgW] == nil then self.buffer[imgW] = {} end
            table.insert(self.buffer[imgW], {bytes = hwc, h = h, w = w, c = c, label_list = label_list, path = img_path})
            if #self.buffer[imgW] == batch_size then
                local out = emit(self.buffer[imgW], imgW)
                self.buffer[imgW] = nil
                return out
            end
        else
            self.cursor = self.cursor + 1
        end
    end
    -- end of the list: flush the buckets one by one (data_gen.lua:123-153)
    for imgW, bucket in pairs(self.buffer) do
        if #bucket > 0 then
            local out = emit(bucket, imgW)
            self.buffer[imgW] = nil
            return out
        end
    end
    self.cursor = 1
    return nil
end
