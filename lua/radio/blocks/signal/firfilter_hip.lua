---
-- Device variant of FIRFilterBlock's process path. In a LuaRadio checkout this is one more branch of the
-- `if platform.features.volk ... elseif platform.features.liquid ... else` ladder of
-- radio/blocks/signal/firfilter.lua:88,165,228, placed first:
--
--     if platform.features.hip then  <the three functions below>  elseif platform.features.volk then ...
--
-- instantiate() and the type signatures (firfilter.lua:43-74) are unchanged.

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')
local types = require('radio.types')

return function (FIRFilterBlock)
    function FIRFilterBlock:initialize()
        -- host-side only: initialize() runs before fork() (radio/core/composite.lua:443)
        self.out = self:get_input_type().vector()
        self.stage = nil
    end

    local function create_stage(self)
        lrhip.ensure()
        local input_complex = (self:get_input_type() == types.ComplexFloat32) and 1 or 0
        local taps_complex = (self.taps.data_type == types.ComplexFloat32) and 1 or 0
        local stage = lrhip.lib.lrhip_fir_create(ffi.cast("const float *", self.taps.data), self.taps.length,
                                                 taps_complex, input_complex, self.decimation or 1, self.use_fft and 1 or 0)
        self.stage = ffi.gc(lrhip.check_object(stage, "Creating lrhip fir object"), lrhip.lib.lrhip_stage_destroy)
    end

    local function process(self, x)
        if self.stage == nil then create_stage(self) end
        return lrhip.execute(self.stage, x, self.out)
    end

    FIRFilterBlock.process_complex_input_complex_taps = process
    FIRFilterBlock.process_complex_input_real_taps = process
    FIRFilterBlock.process_real_input_real_taps = process
    FIRFilterBlock.process_fft_complex_input_complex_taps = process
    FIRFilterBlock.process_fft_complex_input_real_taps = process
    FIRFilterBlock.process_fft_real_input_real_taps = process
end
