---
-- DeviceChainBlock: a maximal linear run of device-capable blocks collapsed into one block / one process /
-- one lrhip_chain_t, so intermediate vectors never leave HBM (one H2D at the head, one D2H at the tail)
-- instead of crossing a UNIX socket per edge (radio/core/pipe.lua:53-69).
--
-- CompositeBlock:_prepare_to_run (radio/core/composite.lua:426) calls collapse() on the flattened
-- connection list right after _crawl_connections (:343): every run b1 -> b2 -> ... -> bk in which each
-- block has exactly one input, one output, a single downstream reader and an `lrhip_stage` constructor is
-- replaced by DeviceChainBlock(b1..bk).
--
-- @block DeviceChainBlock

local ffi = require('ffi')

local block = require('radio.core.block')
local lrhip = require('radio.core.lrhip')

local DeviceChainBlock = block.factory("DeviceChainBlock")

function DeviceChainBlock:instantiate(blocks)
    self.blocks = assert(blocks, "Missing argument #1 (blocks)")
    local first, last = blocks[1], blocks[#blocks]
    self:add_type_signature({block.Input("in", first:get_input_type())}, {block.Output("out", last:get_output_type())})
end

function DeviceChainBlock:get_rate()
    return self.blocks[#self.blocks]:get_rate()
end

function DeviceChainBlock:initialize()
    -- host-side initialisation of the members (tap design etc.); device objects are created post-fork
    for _, b in ipairs(self.blocks) do b:initialize() end
    self.out = self:get_output_type().vector()
    self.chain = nil
end

local function create_chain(self)
    lrhip.ensure()
    local stages = ffi.new("lrhip_stage_t *[?]", #self.blocks)
    for i, b in ipairs(self.blocks) do
        stages[i-1] = b:create_stage()    -- each device block exposes its lazy constructor as create_stage()
    end
    self.stages = stages
    self.chain = ffi.gc(lrhip.check_object(lrhip.lib.lrhip_chain_create(stages, #self.blocks), "Creating lrhip chain object"),
                        lrhip.lib.lrhip_chain_destroy)
end

function DeviceChainBlock:process(x)
    if self.chain == nil then create_chain(self) end
    local lib = lrhip.lib
    local cap = tonumber(lib.lrhip_chain_max_output(self.chain, x.length))
    self.out:resize(cap)
    local n = tonumber(lib.lrhip_chain_execute(self.chain, x.data, x.length, self.out.data, cap))
    if n < 0 then error("lrhip_chain_execute: " .. ffi.string(lib.lrhip_strerror())) end
    return self.out:resize(n)
end

return DeviceChainBlock
