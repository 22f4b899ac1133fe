-- One long IQ recording, G GPUs, from LuaRadio: time partitions of the WBFM-mono receiver of examples/rtlsdr_wbfm_mono.lua (INTEGRATION.md 3a, DESIGN.md 6).
-- For a LuaRadio checkout with the lrhip binding applied (tools/apply_lua_binding.py); run as
--     luaradio iqfile_wbfm_partitions.lua capture.u8 audio 8        -> audio.0.f32 ... audio.7.f32, whose concatenation is the single-process run's audio
--
-- Every partition is an ordinary flow graph IQFileSource -> TunerBlock -> WBFMMonoDemodulator... -> RealFileSink; collapse() turns each into ONE device chain that
-- reads the recording itself (2 bytes per sample over PCIe) and writes its own file, and places partition g on device g % (number of GPUs).  The only thing the
-- script adds is WHERE each chain starts and stops: DeviceChainBlock.on_initialized runs in the parent - after the sources have opened the file, before the block
-- processes are forked - and calls chain:partition(first, last).  That asks a short-lived helper process for the chain's replay start (the parent itself never
-- touches the device: its forked children could not use it then), positions the absorbed source and arms the chain; nothing is exchanged between the partitions.
-- (LuaJIT is not in the build image: this file itself is not executed by the test suite; the same calls - hook, shard_align(), partition(), the source's record
-- window, a chain that reads its window and nothing else, three partitions adding up to the single run on the GPU - are, from the script embedded in
-- tests/test_lua_blocks.py: PARTITIONED.)
local radio = require('radio')
local DeviceChainBlock = require('radio.composites.devicechain')

local path, prefix, G = arg[1], arg[2] or "audio", tonumber(arg[3] or "2")
assert(path, "usage: luaradio iqfile_wbfm_partitions.lua <capture.u8> [<output prefix> [<partitions>]]")
local fs, tune_offset = 1102500, -250e3
local f = assert(io.open(path, "rb"))
local nsamples = math.floor(f:seek("end") / 2)                  -- 'u8' records: two bytes per complex sample
f:close()

local part_of, tops = {}, {}
for g = 0, G - 1 do
    local src = radio.IQFileSource(path, 'u8', fs)
    local tuner = radio.TunerBlock(tune_offset, 200e3, 5)
    local fm = radio.FrequencyDiscriminatorBlock(1.25)
    local af = radio.LowpassFilterBlock(128, 15e3)
    local de = radio.FMDeemphasisFilterBlock(75e-6)
    local ds = radio.DownsamplerBlock(5)
    local sink = radio.RealFileSink(string.format("%s.%d.f32", prefix, g), 'f32le')
    local top = radio.CompositeBlock()
    top:connect(src, tuner, fm, af, de, ds, sink)
    part_of[src] = g
    tops[#tops + 1] = top
end

DeviceChainBlock.on_initialized = function (chain)
    local g = chain.source and part_of[chain.source]
    if g == nil then return end
    -- boundaries on the chain's own grid (128 000 samples for this receiver: its tile grid and 25 = the total decimation), so that the cuts fall where the
    -- uninterrupted run's tiles fall; the last partition takes the rest
    local align = chain:shard_align()
    local per = math.floor(nsamples / G / align) * align
    local first, last = g * per, (g == G - 1) and nsamples or (g + 1) * per
    chain.device = g                                              -- placement index, wrapped over the devices of the box by lrhip.ensure()
    chain:partition(first, last)
end

for _, top in ipairs(tops) do top:start() end
for _, top in ipairs(tops) do top:wait() end
