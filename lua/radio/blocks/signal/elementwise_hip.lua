---
-- Device variants of FrequencyTranslatorBlock (radio/blocks/signal/frequencytranslator.lua:32-53),
-- DownsamplerBlock (downsampler.lua:40-56), FrequencyDiscriminatorBlock (frequencydiscriminator.lua:33-64)
-- and IIRFilterBlock (iirfilter.lua:79-109). Each is the `if platform.features.hip then` branch of the
-- corresponding file; instantiate() and type signatures are unchanged.  Stages are created lazily on the
-- first process() because initialize() runs pre-fork (radio/core/composite.lua:443 vs :569).

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')
local types = require('radio.types')

local M = {}

local function lazy(self, create)
    if self.stage == nil then
        lrhip.ensure()
        self.stage = ffi.gc(lrhip.check_object(create(), "Creating lrhip " .. self.name .. " object"),
                            lrhip.lib.lrhip_stage_destroy)
    end
    return self.stage
end

function M.patch_frequencytranslator(FrequencyTranslatorBlock)
    function FrequencyTranslatorBlock:initialize()
        self.omega = 2*math.pi*(self.offset/self:get_rate())
        self.out = types.ComplexFloat32.vector()
    end
    function FrequencyTranslatorBlock:process(x)
        local stage = lazy(self, function () return lrhip.lib.lrhip_rotator_create(self.omega) end)
        return lrhip.execute(stage, x, self.out)
    end
end

function M.patch_downsampler(DownsamplerBlock)
    function DownsamplerBlock:initialize()
        self.out = self:get_input_type().vector()
    end
    function DownsamplerBlock:process(x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_downsampler_create(self.factor, ffi.sizeof(self:get_input_type()))
        end)
        return lrhip.execute(stage, x, self.out)
    end
end

function M.patch_frequencydiscriminator(FrequencyDiscriminatorBlock)
    function FrequencyDiscriminatorBlock:initialize()
        self.out = types.Float32.vector()
    end
    function FrequencyDiscriminatorBlock:process(x)
        local stage = lazy(self, function () return lrhip.lib.lrhip_fmdiscrim_create(self.gain) end)
        return lrhip.execute(stage, x, self.out)
    end
end

function M.patch_iirfilter(IIRFilterBlock)
    function IIRFilterBlock:initialize()
        self.out = self:get_input_type().vector()
    end
    local function process(self, x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_iir_create(ffi.cast("const float *", self.b_taps.data), self.b_taps.length,
                                              ffi.cast("const float *", self.a_taps.data), self.a_taps.length,
                                              (self:get_input_type() == types.ComplexFloat32) and 1 or 0)
        end)
        return lrhip.execute(stage, x, self.out)
    end
    IIRFilterBlock.process_complex = process
    IIRFilterBlock.process_real = process
end

-- One-input element-wise blocks (complexmagnitude.lua, complexphase.lua, complextoreal.lua, complextoimag.lua,
-- complexconjugate.lua, realtocomplex.lua, absolutevalue.lua): M.patch_unary(ComplexMagnitudeBlock, "complexmagnitude")
function M.patch_unary(Block, op)
    function Block:process(x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_unary_create(op, 0, 0, 0, (self:get_input_type() == types.ComplexFloat32) and 1 or 0)
        end)
        return lrhip.execute(stage, x, self.out)
    end
end

-- AddConstantBlock (addconstant.lua:26-75): the constant's type decides the arithmetic, as in the reference
function M.patch_addconstant(AddConstantBlock)
    local function process(self, x)
        local stage = lazy(self, function ()
            local c = self.constant
            local cc = ffi.istype(types.ComplexFloat32, c)
            return lrhip.lib.lrhip_unary_create("addconstant", cc and c.real or (type(c) == "number" and c or c.value),
                                                cc and c.imag or 0, cc and 1 or 0,
                                                (self:get_input_type() == types.ComplexFloat32) and 1 or 0)
        end)
        return lrhip.execute(stage, x, self.out)
    end
    AddConstantBlock.process_complex_by_complex = process
    AddConstantBlock.process_complex_by_real = process
    AddConstantBlock.process_real_by_real = process
end

-- DelayBlock (delay.lua:43-72), ComplexFloat32 / Float32 signatures
function M.patch_delay(DelayBlock)
    function DelayBlock:process(x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_delay_create(self.num_samples, ffi.sizeof(self:get_input_type()))
        end)
        return lrhip.execute(stage, x, self.out)
    end
end

-- HilbertTransformBlock (hilberttransform.lua:100-160); self.hilbert_taps as computed by instantiate() (not reversed)
function M.patch_hilberttransform(HilbertTransformBlock)
    function HilbertTransformBlock:initialize()
        self.out = types.ComplexFloat32.vector()
    end
    function HilbertTransformBlock:process(x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_hilbert_create(ffi.cast("const float *", self.hilbert_taps.data), self.hilbert_taps.length)
        end)
        return lrhip.execute(stage, x, self.out)
    end
end

-- Two-input element-wise blocks (multiply.lua, multiplyconjugate.lua, add.lua, subtract.lua, floattocomplex.lua):
-- M.patch_binary(MultiplyBlock, "multiply")
function M.patch_binary(Block, op)
    local function process(self, x, y)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_binary_create(op, (self:get_input_type() == types.ComplexFloat32) and 1 or 0)
        end)
        return lrhip.execute2(stage, x, y, self.out)
    end
    Block.process = process
    Block.process_complex = process
    Block.process_real = process
end

-- MultiplyConstantBlock (multiplyconstant.lua:26-75), same constant-type rules as AddConstantBlock
function M.patch_multiplyconstant(MultiplyConstantBlock)
    local function process(self, x)
        local stage = lazy(self, function ()
            local c = self.constant
            local cc = ffi.istype(types.ComplexFloat32, c)
            return lrhip.lib.lrhip_multiply_constant_create(cc and c.real or (type(c) == "number" and c or c.value), cc and c.imag or 0,
                                                            cc and 1 or 0, (self:get_input_type() == types.ComplexFloat32) and 1 or 0)
        end)
        return lrhip.execute(stage, x, self.out)
    end
    MultiplyConstantBlock.process_complex_by_complex = process
    MultiplyConstantBlock.process_complex_by_real = process
    MultiplyConstantBlock.process_real_by_real = process
end

-- UpsamplerBlock (upsampler.lua:45-53)
function M.patch_upsampler(UpsamplerBlock)
    function UpsamplerBlock:process(x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_upsampler_create(self.factor, ffi.sizeof(self:get_input_type()))
        end)
        return lrhip.execute(stage, x, self.out)
    end
end

-- FrequencyModulatorBlock (frequencymodulator.lua:24-90)
function M.patch_frequencymodulator(FrequencyModulatorBlock)
    function FrequencyModulatorBlock:initialize()
        self.out = types.ComplexFloat32.vector()
    end
    function FrequencyModulatorBlock:process(x)
        local stage = lazy(self, function () return lrhip.lib.lrhip_fmmod_create(self.modulation_index) end)
        return lrhip.execute(stage, x, self.out)
    end
end

-- AGCBlock (agc.lua:45-96): the two recurrences run as prefix scans on the device; alphas as computed in initialize()
function M.patch_agc(AGCBlock)
    local function process(self, x)
        local stage = lazy(self, function ()
            return lrhip.lib.lrhip_agc_create(self.power_alpha, self.gain_alpha, self.target, self.threshold,
                                              (self:get_input_type() == types.ComplexFloat32) and 1 or 0)
        end)
        return lrhip.execute(stage, x, self.out)
    end
    AGCBlock.process_real = process
    AGCBlock.process_complex = process
end

return M
