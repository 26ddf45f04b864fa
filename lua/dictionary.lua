--[[ dictionary.lua -- the trie loadDictionary (src/utils/utils.lua:177-218) builds out of nested tds.Hash tables, flattened once into
     the three device arrays of `aocr_trie` (include/aocr.h): child_mask[n] (uint64, bit v-1 <=> node n has a child for vocab id v),
     child_base[n] (int32) and child[] (int32, ascending v per node).  Node 0 is trie[2], the start symbol's node (model.lua:380-387).
     The walk keeps tds.Hash identity, so the aliases -allow_digit_prefix creates (node[3] = trie[2], utils.lua:193-198) become
     edges back to node 0 instead of an endless expansion.  The reference walks these tables on the host for every image, beam and
     candidate of every decode step (model.lua:405-445, 460-513); here the constraint is tested inside the selection kernel. ]]
local ffi = require 'ffi'
local M = {}

local function key_of(h)                    -- identity of a tds.Hash (a cdata pointer to the C hash [upstream tds])
    return tostring(ffi.cast('intptr_t', ffi.cast('void*', h)))
end

function M.flatten(trie, A)
    local root = trie[2]
    local ids, nodes = {[key_of(root)] = 0}, {root}
    local masks, bases, childs = {}, {}, {}
    local i = 1
    while i <= #nodes do                      -- breadth first; ids are assigned on first sight
        local node = nodes[i]
        local vs = {}
        for v, _ in pairs(node) do table.insert(vs, tonumber(v)) end
        table.sort(vs)
        local lo, hi = 0, 0                   -- 64-bit mask as two 32-bit halves (LuaJIT doubles hold 53 bits)
        bases[i] = #childs
        for _, v in ipairs(vs) do
            assert(v >= 1 and v <= 64, 'vocab id outside 1..64')
            if v <= 32 then lo = lo + 2 ^ (v - 1) else hi = hi + 2 ^ (v - 33) end
            local child = node[v]
            local k = key_of(child)
            if ids[k] == nil then ids[k] = #nodes; table.insert(nodes, child) end
            table.insert(childs, ids[k])
        end
        masks[i] = {lo, hi}
        i = i + 1
    end
    local n, e = #nodes, #childs
    local mask = ffi.new('uint32_t[?]', 2 * n); local base = ffi.new('int32_t[?]', n); local child = ffi.new('int32_t[?]', math.max(e, 1))
    for j = 1, n do mask[2 * (j - 1)] = masks[j][1]; mask[2 * (j - 1) + 1] = masks[j][2]; base[j - 1] = bases[j] end   -- little endian: low half first
    for j = 1, e do child[j - 1] = childs[j] end
    local out = {source = trie, n_nodes = n, n_edges = e}
    out.mask_dev = A.device_bytes(8 * n); A.upload(out.mask_dev, mask, 8 * n)
    out.base_dev = A.device_bytes(4 * n); A.upload(out.base_dev, base, 4 * n)
    out.child_dev = A.device_bytes(4 * math.max(e, 1)); A.upload(out.child_dev, child, 4 * math.max(e, 1))
    out.desc = ffi.new('aocr_trie')
    out.desc.child_mask_dev = out.mask_dev:as('const uint64_t*'); out.desc.child_base_dev = out.base_dev:as('const int32_t*')
    out.desc.child_dev = out.child_dev:as('const int32_t*'); out.desc.n_nodes = n; out.desc.n_edges = e
    return out
end
return M
