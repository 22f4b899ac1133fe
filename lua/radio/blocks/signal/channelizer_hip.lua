---
-- PolyphaseChannelizerBlock: a critically sampled K-channel analysis filterbank (BASELINE.json configs[4]).  NOT a block of the reference - there is no
-- channelizer under radio/blocks/ - so this file is a whole block, not a patch: tools/apply_lua_binding.py copies it to
-- radio/blocks/signal/channelizer_hip.lua and a script reaches it as
--
--     local PolyphaseChannelizerBlock = require('radio.blocks.signal.channelizer_hip').PolyphaseChannelizerBlock
--     top:connect(source, PolyphaseChannelizerBlock(64, taps), sink)
--
-- It is defined by reference blocks: K parallel chains FrequencyTranslatorBlock(-c * rate / K) -> FIRFilterBlock(taps) -> DownsamplerBlock(K),
-- c = 0 .. K-1, evaluated as ONE dense GEMM on the f32 matrix cores (lrhip_channelizer_create).  Output: frames of K ComplexFloat32 values, channel c at
-- position c of its frame, one frame per K input samples - so the port carries rate samples per second in total and each channel runs at rate / K.
-- K in {32, 64}; #taps a multiple of 32.  Without the library the constructor raises (there is no host implementation to fall back to).
--
-- @block PolyphaseChannelizerBlock
-- @tparam int num_channels Number of channels K
-- @tparam array|vector taps Real-valued prototype lowpass taps (e.g. radio.utilities.filter_utils.firwin_lowpass(16 * K, 1 / K))

local ffi = require('ffi')

local block = require('radio.core.block')
local types = require('radio.types')
local lrhip = require('radio.core.lrhip')

local PolyphaseChannelizerBlock = block.factory("PolyphaseChannelizerBlock")

function PolyphaseChannelizerBlock:instantiate(num_channels, taps)
    assert(lrhip.available, "PolyphaseChannelizerBlock needs liblrhip.so")
    self.num_channels = assert(num_channels, "Missing argument #1 (num_channels)")
    assert(taps, "Missing argument #2 (taps)")
    if type(taps) == "table" and taps.data_type == nil then
        self.taps = types.Float32.vector_from_array(taps)
    else
        assert(taps.data_type == types.Float32, "Unsupported taps type")
        self.taps = taps
    end
    self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.ComplexFloat32)})
end

function PolyphaseChannelizerBlock:initialize()
    self.out = types.ComplexFloat32.vector()
end

local M = {PolyphaseChannelizerBlock = PolyphaseChannelizerBlock}

function M.patch(Block)
    lrhip.device_block(Block, function (self)
        return lrhip.lib.lrhip_channelizer_create(ffi.cast("const float *", self.taps.data), self.taps.length, self.num_channels)
    end)
    function Block:process(x)
        return lrhip.execute(self:create_stage(), x, self.out, self)
    end
end

M.patch(PolyphaseChannelizerBlock)

return M
