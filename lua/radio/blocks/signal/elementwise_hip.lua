---
-- Device variants of FrequencyTranslatorBlock (radio/blocks/signal/frequencytranslator.lua:32-53),
-- DownsamplerBlock (downsampler.lua:40-56), FrequencyDiscriminatorBlock (frequencydiscriminator.lua:33-64),
-- IIRFilterBlock (iirfilter.lua:79-109) and the element-wise / one-tap blocks next to the path.  Each patch is applied by
-- ONE line directly above the reference file's final `return <Block>` - require('radio.core.lrhip').patch('<file>', <Block>) -
-- i.e. after every top-level statement of that file (several define initialize()/process() outside any feature ladder, e.g.
-- downsampler.lua:40-56, and firfilter.lua:400-402 assigns methods after its ladder); instantiate() and the type signatures are
-- unchanged.  A patch sets every function a type signature of the block can bind (add_type_signature's process_func argument, or
-- `process` by default - radio/core/block.lua:283-288); tests/test_lua_glue.py checks that against the reference files.
-- Every block gets create_stage() (lrhip.device_block): the stage is created lazily on the first process() because
-- initialize() runs pre-fork (radio/core/composite.lua:443 vs :569), and DeviceChainBlock collects the same stages.

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')
local types = require('radio.types')

local M = {}

local function is_complex(self)
    return (self:get_input_type() == types.ComplexFloat32) and 1 or 0
end

-- one-input process(): execute the block's stage on the chunk
local function process(self, x)
    return lrhip.execute(self:create_stage(), x, self.out, self)
end

-- a numeric or ComplexFloat32 / Float32 constant -> re, im, is_complex (addconstant.lua:33-44, multiplyconstant.lua:33-44)
local function split_constant(c)
    if ffi.istype(types.ComplexFloat32, c) then return c.real, c.imag, 1 end
    if type(c) == "number" then return c, 0, 0 end
    return c.value, 0, 0
end

function M.patch_frequencytranslator(FrequencyTranslatorBlock)
    function FrequencyTranslatorBlock:initialize()
        self.omega = 2*math.pi*(self.offset/self:get_rate())
        self.out = types.ComplexFloat32.vector()
    end
    lrhip.device_block(FrequencyTranslatorBlock, function (self) return lrhip.lib.lrhip_rotator_create(self.omega) end)
    FrequencyTranslatorBlock.process = process
end

function M.patch_downsampler(DownsamplerBlock)
    function DownsamplerBlock:initialize()
        self.out = self:get_input_type().vector()
    end
    lrhip.device_block(DownsamplerBlock, function (self)
        return lrhip.lib.lrhip_downsampler_create(self.factor, ffi.sizeof(self:get_input_type()))
    end)
    DownsamplerBlock.process = process
end

function M.patch_frequencydiscriminator(FrequencyDiscriminatorBlock)
    function FrequencyDiscriminatorBlock:initialize()
        self.out = types.Float32.vector()
    end
    lrhip.device_block(FrequencyDiscriminatorBlock, function (self) return lrhip.lib.lrhip_fmdiscrim_create(self.gain) end)
    FrequencyDiscriminatorBlock.process = process
end

function M.patch_iirfilter(IIRFilterBlock)
    function IIRFilterBlock:initialize()
        self.out = self:get_input_type().vector()
    end
    lrhip.device_block(IIRFilterBlock, function (self)
        return lrhip.lib.lrhip_iir_create(ffi.cast("const float *", self.b_taps.data), self.b_taps.length,
                                          ffi.cast("const float *", self.a_taps.data), self.a_taps.length, is_complex(self))
    end)
    IIRFilterBlock.process_complex = process
    IIRFilterBlock.process_real = process
end

-- One-input element-wise blocks (complexmagnitude.lua, complexphase.lua, complextoreal.lua, complextoimag.lua,
-- complexconjugate.lua, realtocomplex.lua, absolutevalue.lua): M.patch_unary(ComplexMagnitudeBlock, "complexmagnitude")
function M.patch_unary(Block, op)
    lrhip.device_block(Block, function (self) return lrhip.lib.lrhip_unary_create(op, 0, 0, 0, is_complex(self)) end)
    Block.process = process
end

-- AddConstantBlock (addconstant.lua:26-75): the constant's type decides the arithmetic, as in the reference
function M.patch_addconstant(AddConstantBlock)
    lrhip.device_block(AddConstantBlock, function (self)
        local re, im, cc = split_constant(self.constant)
        return lrhip.lib.lrhip_unary_create("addconstant", re, im, cc, is_complex(self))
    end)
    AddConstantBlock.process = process                       -- addconstant.lua:45,47: the default process() of two signatures
    AddConstantBlock.process_complex_by_complex = process
    AddConstantBlock.process_complex_by_real = process
    AddConstantBlock.process_real_by_real = process
end

-- DelayBlock (delay.lua:43-72), ComplexFloat32 / Float32 signatures
function M.patch_delay(DelayBlock)
    local reference_process = DelayBlock.process
    lrhip.device_block(DelayBlock, function (self)
        return lrhip.lib.lrhip_delay_create(self.num_samples, ffi.sizeof(self:get_input_type()))
    end)
    -- the Bit / Byte signatures (delay.lua:32-33) are not sample streams of the device path: they keep the reference's loop
    function DelayBlock:device_capable()
        local data_type = self:get_input_type()
        return data_type == types.ComplexFloat32 or data_type == types.Float32
    end
    function DelayBlock:process(x)
        if not self:device_capable() then return reference_process(self, x) end
        return process(self, x)
    end
end

-- HilbertTransformBlock (hilberttransform.lua:100-160); self.hilbert_taps as computed by instantiate() (not reversed)
function M.patch_hilberttransform(HilbertTransformBlock)
    function HilbertTransformBlock:initialize()
        self.out = types.ComplexFloat32.vector()
    end
    lrhip.device_block(HilbertTransformBlock, function (self)
        return lrhip.lib.lrhip_hilbert_create(ffi.cast("const float *", self.hilbert_taps.data), self.hilbert_taps.length)
    end)
    HilbertTransformBlock.process = process
end

-- Two-input element-wise blocks (multiply.lua, multiplyconjugate.lua, add.lua, subtract.lua, floattocomplex.lua):
-- M.patch_binary(MultiplyBlock, "multiply").  Two inputs: not a member of linear device chains (DeviceChainBlock.collapse skips them).
function M.patch_binary(Block, op)
    lrhip.device_block(Block, function (self) return lrhip.lib.lrhip_binary_create(op, is_complex(self)) end)
    local function process2(self, x, y)
        return lrhip.execute2(self:create_stage(), x, y, self.out, self)
    end
    Block.process = process2
    Block.process_complex = process2
    Block.process_real = process2
end

-- MultiplyConstantBlock (multiplyconstant.lua:26-75), same constant-type rules as AddConstantBlock
function M.patch_multiplyconstant(MultiplyConstantBlock)
    lrhip.device_block(MultiplyConstantBlock, function (self)
        local re, im, cc = split_constant(self.constant)
        return lrhip.lib.lrhip_multiply_constant_create(re, im, cc, is_complex(self))
    end)
    MultiplyConstantBlock.process = process                  -- multiplyconstant.lua:46,48: the default process() of two signatures
    MultiplyConstantBlock.process_complex_by_complex = process
    MultiplyConstantBlock.process_complex_by_real = process
    MultiplyConstantBlock.process_real_by_real = process
end

-- UpsamplerBlock (upsampler.lua:45-53)
function M.patch_upsampler(UpsamplerBlock)
    lrhip.device_block(UpsamplerBlock, function (self)
        return lrhip.lib.lrhip_upsampler_create(self.factor, ffi.sizeof(self:get_input_type()))
    end)
    UpsamplerBlock.process = process
end

-- FrequencyModulatorBlock (frequencymodulator.lua:24-90)
function M.patch_frequencymodulator(FrequencyModulatorBlock)
    function FrequencyModulatorBlock:initialize()
        self.out = types.ComplexFloat32.vector()
    end
    lrhip.device_block(FrequencyModulatorBlock, function (self) return lrhip.lib.lrhip_fmmod_create(self.modulation_index) end)
    FrequencyModulatorBlock.process = process
end

-- AGCBlock (agc.lua:45-96): the two recurrences run as prefix scans on the device; alphas as computed in initialize()
function M.patch_agc(AGCBlock)
    lrhip.device_block(AGCBlock, function (self)
        return lrhip.lib.lrhip_agc_create(self.power_alpha, self.gain_alpha, self.target, self.threshold, is_complex(self))
    end)
    AGCBlock.process_real = process
    AGCBlock.process_complex = process
end

-- PowerSquelchBlock (powersquelch.lua:24-80): alpha and the linearised threshold as computed by the reference's initialize()
function M.patch_powersquelch(PowerSquelchBlock)
    lrhip.device_block(PowerSquelchBlock, function (self)
        return lrhip.lib.lrhip_powersquelch_create(self.alpha, self.threshold, is_complex(self))
    end)
    PowerSquelchBlock.process_real = process
    PowerSquelchBlock.process_complex = process
end

return M
