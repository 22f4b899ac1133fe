-- Stand-ins for the reference's block files as far as their DEVICE VARIANTS look at them - TEST INFRASTRUCTURE for tests/test_lua_blocks.py.
-- Each class has the reference's constructor arguments, the fields its instantiate() sets and its type signatures, re-stated from the reference's
-- documentation and the field names the *_hip.lua files read (radio/blocks/signal/*.lua, radio/blocks/sources/iqfile.lua, radio/blocks/sinks/iqfile.lua,
-- radio/utilities/spectrum_utils.lua: cited per class); the host arithmetic of the reference (its process() loops) is NOT restated - a stand-in's
-- process() raises, so a test can only pass through the device variant.  Then the REAL patch line of tools/apply_lua_binding.py is applied to each:
-- require('radio.core.lrhip').patch('<name>', <Block>).
local ffi = require('ffi')
local block = require('radio.core.block')
local types = require('radio.types')
local format_utils = require('radio.utilities.format_utils')
local lrhip = require('radio.core.lrhip')

local R = {}

local function host_loop()
    error("the reference's host loop: not under test")
end

-- radio/blocks/sources/iqfile.lua:36-81, realfile.lua (file by name only)
local function file_source(name, data_type)
    local Source = block.factory(name)
    function Source:instantiate(file, format, rate, repeat_on_eof)
        self.filename = assert(file, "Missing argument #1 (file)")
        assert(format, "Missing argument #2 (format)")
        self.format = assert(format_utils.formats[format], "Unsupported format (\"" .. format .. "\")")
        self.rate = assert(rate, "Missing argument #3 (rate)")
        self.repeat_on_eof = repeat_on_eof or false
        self.chunk_size = 8192
        self:add_type_signature({}, {block.Output("out", data_type)})
    end
    function Source:get_rate() return self.rate end
    function Source:initialize()
        self.file = ffi.C.fopen(self.filename, "rb")
        if self.file == nil then error("fopen(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
        self.files[self.file] = true
        self.out = data_type.vector()
    end
    Source.process = host_loop
    function Source:cleanup()
        if ffi.C.fclose(self.file) ~= 0 then error("fclose()") end
    end
    return Source
end
R.IQFileSource = file_source("IQFileSource", types.ComplexFloat32)
R.RealFileSource = file_source("RealFileSource", types.Float32)

-- radio/blocks/sinks/iqfile.lua:36-92, realfile.lua
local function file_sink(name, data_type)
    local Sink = block.factory(name)
    function Sink:instantiate(file, format)
        self.filename = assert(file, "Missing argument #1 (file)")
        assert(format, "Missing argument #2 (format)")
        self.format = assert(format_utils.formats[format], "Unsupported format (\"" .. format .. "\")")
        self:add_type_signature({block.Input("in", data_type)}, {})
    end
    function Sink:initialize()
        self.file = ffi.C.fopen(self.filename, "wb")
        if self.file == nil then error("fopen(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
        self.files[self.file] = true
    end
    Sink.process = host_loop
    function Sink:cleanup()
        if ffi.C.fclose(self.file) ~= 0 then error("fclose()") end
    end
    return Sink
end
R.IQFileSink = file_sink("IQFileSink", types.ComplexFloat32)
R.RealFileSink = file_sink("RealFileSink", types.Float32)

-- a plain host sink / source pair for the graph's outer ends
R.HostSink = block.factory("HostSink")
function R.HostSink:instantiate(data_type)
    self:add_type_signature({block.Input("in", data_type or types.ComplexFloat32)}, {})
end
R.HostSource = block.factory("HostSource")
function R.HostSource:instantiate(rate, data_type)
    self.rate = rate
    self:add_type_signature({}, {block.Output("out", data_type or types.ComplexFloat32)})
end
function R.HostSource:get_rate() return self.rate end

-- radio/blocks/signal/frequencytranslator.lua:26-30
R.FrequencyTranslatorBlock = block.factory("FrequencyTranslatorBlock")
function R.FrequencyTranslatorBlock:instantiate(offset)
    self.offset = assert(offset, "Missing argument #1 (offset)")
    self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.ComplexFloat32)})
end
R.FrequencyTranslatorBlock.process = host_loop

-- radio/blocks/signal/firfilter.lua:43-74 (taps as a Float32 / ComplexFloat32 vector)
R.FIRFilterBlock = block.factory("FIRFilterBlock")
function R.FIRFilterBlock:instantiate(taps, use_fft)
    self.taps = assert(taps, "Missing argument #1 (taps)")
    self.use_fft = use_fft
    if self.taps.data_type == types.ComplexFloat32 then
        self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.ComplexFloat32)},
                                R.FIRFilterBlock.process_complex_input_complex_taps)
    else
        self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.ComplexFloat32)},
                                R.FIRFilterBlock.process_complex_input_real_taps)
        self:add_type_signature({block.Input("in", types.Float32)}, {block.Output("out", types.Float32)}, R.FIRFilterBlock.process_real_input_real_taps)
    end
end
for _, name in ipairs({"process_complex_input_complex_taps", "process_complex_input_real_taps", "process_real_input_real_taps"}) do
    R.FIRFilterBlock[name] = host_loop
end

-- radio/blocks/signal/downsampler.lua:29-38
R.DownsamplerBlock = block.factory("DownsamplerBlock")
function R.DownsamplerBlock:instantiate(factor)
    self.factor = assert(factor, "Missing argument #1 (factor)")
    self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.ComplexFloat32)})
    self:add_type_signature({block.Input("in", types.Float32)}, {block.Output("out", types.Float32)})
end
function R.DownsamplerBlock:get_rate() return block.Block.get_rate(self) / self.factor end
R.DownsamplerBlock.process = host_loop

-- radio/blocks/signal/frequencydiscriminator.lua:25-31
R.FrequencyDiscriminatorBlock = block.factory("FrequencyDiscriminatorBlock")
function R.FrequencyDiscriminatorBlock:instantiate(modulation_index)
    assert(modulation_index, "Missing argument #1 (modulation_index)")
    self.gain = 2 * math.pi * modulation_index
    self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.Float32)})
end
R.FrequencyDiscriminatorBlock.process = host_loop

-- radio/blocks/signal/iirfilter.lua:39-61 (b_taps / a_taps as Float32 vectors)
R.IIRFilterBlock = block.factory("IIRFilterBlock")
function R.IIRFilterBlock:instantiate(b_taps, a_taps)
    self.b_taps = assert(b_taps, "Missing argument #1 (b_taps)")
    self.a_taps = assert(a_taps, "Missing argument #2 (a_taps)")
    self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {block.Output("out", types.ComplexFloat32)}, R.IIRFilterBlock.process_complex)
    self:add_type_signature({block.Input("in", types.Float32)}, {block.Output("out", types.Float32)}, R.IIRFilterBlock.process_real)
end
R.IIRFilterBlock.process_complex = host_loop
R.IIRFilterBlock.process_real = host_loop

-- radio/blocks/signal/multiplyconjugate.lua:26-32, add.lua
local function two_input(name, signatures)
    local B = block.factory(name)
    function B:instantiate()
        for _, t in ipairs(signatures) do
            self:add_type_signature({block.Input("in1", t), block.Input("in2", t)}, {block.Output("out", t)})
        end
    end
    function B:initialize() self.out = self:get_output_type().vector() end
    B.process = host_loop
    return B
end
R.MultiplyConjugateBlock = two_input("MultiplyConjugateBlock", {types.ComplexFloat32})
R.AddBlock = two_input("AddBlock", {types.ComplexFloat32, types.Float32})

-- radio/blocks/sinks/gnuplotspectrum.lua:42-56, :73-137 as far as the Welch variant reads it: initialize_gnuplot() sets the counters, the average vector
-- and self.psd; write_gnuplot() collects what would go down the pipe to gnuplot
R.GnuplotSpectrumSink = block.factory("GnuplotSpectrumSink")
function R.GnuplotSpectrumSink:instantiate(num_samples, title, options)
    self.num_samples = num_samples or 1024
    self.title = title or ""
    self.options = options or {}
    self.update_time = self.options.update_time or 0.10
    self.overlap = self.options.overlap or 0.00
    self.reference_level = self.options.reference_level or 0.00
    self:add_type_signature({block.Input("in", types.Float32)}, {})
    self:add_type_signature({block.Input("in", types.ComplexFloat32)}, {})
end
function R.GnuplotSpectrumSink:write_gnuplot(s)
    self.written = self.written or {}
    self.written[#self.written + 1] = s
end
function R.GnuplotSpectrumSink:initialize_gnuplot()
    local sample_rate = self:get_rate()
    self.gnuplot_f = true
    self.plot_str = "plot\n"
    self.state_index = 0
    self.sample_count = 0
    self.num_overlap = math.floor(self.overlap * self.num_samples)
    self.num_plot_update = math.floor(self.update_time * sample_rate)
    self.state_psd_average = types.Float32.vector(self.num_samples)
    self.state_psd_average_count = 0
    self.psd = R.spectrum_utils.PSD(self:get_input_type().vector(self.num_samples), types.Float32.vector(self.num_samples),
                                    self.options.window or "hamming", sample_rate, true)
end
R.GnuplotSpectrumSink.process = host_loop

-- radio/utilities/spectrum_utils.lua:25-57, :259-291, :522-561: the three classes' constructors (fields only); the window comes from the test
-- (window_utils is host code of the reference), compute() of a stand-in raises
local function class_factory()
    local cls = {}
    cls.__index = cls
    setmetatable(cls, {__call = function (c, ...) return c.new(...) end})
    return cls
end
local DFT, IDFT, PSD = class_factory(), class_factory(), class_factory()
function DFT.new(input_samples, output_samples)
    local self = setmetatable({}, DFT)
    self.input_samples, self.output_samples = input_samples, output_samples
    self.num_samples = input_samples.length
    self.data_type = input_samples.data_type
    if self.data_type == types.ComplexFloat32 then self.compute = self.compute_complex else self.compute = self.compute_real end
    self:initialize()
    return self
end
function IDFT.new(input_samples, output_samples)
    local self = setmetatable({}, IDFT)
    self.input_samples, self.output_samples = input_samples, output_samples
    self.num_samples = input_samples.length
    self.data_type = output_samples.data_type
    if self.data_type == types.ComplexFloat32 then self.compute = self.compute_complex else self.compute = self.compute_real end
    self:initialize()
    return self
end
for _, cls in ipairs({DFT, IDFT}) do
    function cls:initialize() self.reference_initialized = true end
    cls.compute_complex = host_loop
    cls.compute_real = host_loop
end
R.window_of = nil           -- set by the test: function (num_samples, window_type) -> Float32 vector (the periodic window of spectrum_utils.lua:547)
function PSD.new(input_samples, output_samples, window_type, sample_rate, logarithmic)
    local self = setmetatable({}, PSD)
    self.input_samples, self.output_samples = input_samples, output_samples
    self.window_type = window_type or "hamming"
    self.sample_rate = sample_rate or 2
    self.logarithmic = (logarithmic == nil) and true or logarithmic
    self.num_samples = input_samples.length
    self.data_type = input_samples.data_type
    self.window = R.window_of(self.num_samples, self.window_type)
    self.window_energy = 0
    for i = 0, self.num_samples - 1 do
        self.window_energy = self.window_energy + self.window.data[i].value * self.window.data[i].value
    end
    self.windowed_samples = input_samples.data_type.vector(self.num_samples)
    self.dft_samples = types.ComplexFloat32.vector(self.num_samples)
    self.dft = DFT(self.windowed_samples, self.dft_samples)
    return self
end
PSD.compute = host_loop
R.spectrum_utils = {DFT = DFT, IDFT = IDFT, PSD = PSD}

-- the lines tools/apply_lua_binding.py inserts
lrhip.patch('iqfilesource', R.IQFileSource)
lrhip.patch('realfilesource', R.RealFileSource)
lrhip.patch('iqfilesink', R.IQFileSink)
lrhip.patch('realfilesink', R.RealFileSink)
lrhip.patch('frequencytranslator', R.FrequencyTranslatorBlock)
lrhip.patch('firfilter', R.FIRFilterBlock)
lrhip.patch('downsampler', R.DownsamplerBlock)
lrhip.patch('frequencydiscriminator', R.FrequencyDiscriminatorBlock)
lrhip.patch('iirfilter', R.IIRFilterBlock)
lrhip.patch('multiplyconjugate', R.MultiplyConjugateBlock)
lrhip.patch('add', R.AddBlock)
lrhip.patch('gnuplotspectrum', R.GnuplotSpectrumSink)
lrhip.patch_spectrum(DFT, IDFT, PSD)

-- connect(a, b, ...) / connect(a, "out", b, "in2"): the flattened {[InputPort] = OutputPort} table of CompositeBlock:_crawl_connections
function R.graph()
    local g = {connections = {}}
    local function port(ports, name)
        for _, p in ipairs(ports) do
            if p.name == name then return p end
        end
        error("no port " .. name)
    end
    function g.connect(...)
        local a = {...}
        if #a == 4 and type(a[2]) == "string" then
            g.connections[port(a[3].inputs, a[4])] = port(a[1].outputs, a[2])
        else
            for i = 2, #a do g.connections[a[i].inputs[1]] = a[i-1].outputs[1] end
        end
    end
    return g
end

-- CompositeBlock:_prepare_to_run with the hooks of tools/apply_lua_binding.py, on a flattened connection table and the list of the composite's own
-- blocks (radio/core/composite.lua:426-470): collapse, connect pipes, initialize the blocks, then the device blocks
function R.prepare(connections, blocks)
    local pipe = require('radio.core.pipe')
    local graphs, chains, device_chains
    connections, graphs = require('radio.composites.devicegraph').collapse(connections)
    connections, chains = require('radio.composites.devicechain').collapse(connections)
    connections, device_chains = require('radio.composites.devicefanout').collapse(connections, chains)
    for _, g in ipairs(graphs) do device_chains[#device_chains + 1] = g end
    for input, output in pairs(connections) do
        local p = pipe.Pipe(output, input)
        output.pipes[#output.pipes + 1] = p
        input.pipe = p
    end
    for _, b in ipairs(blocks) do b:initialize() end
    for _, c in ipairs(device_chains) do c:initialize() end
    return connections, device_chains
end

return R
