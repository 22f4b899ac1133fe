---
-- Device variant of FIRFilterBlock's process path. In a LuaRadio checkout this is one more branch of the
-- `if platform.features.volk ... elseif platform.features.liquid ... else` ladder of
-- radio/blocks/signal/firfilter.lua:88,165,228, placed first:
--
--     if platform.features.hip then  <the functions below>  elseif platform.features.volk then ...
--
-- The type signatures (firfilter.lua:59-74) are unchanged; instantiate() is wrapped only to remember whether the
-- caller chose use_fft at all (nil = let the library pick the fast arithmetic, lrhip.fir_mode()).

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')
local types = require('radio.types')

return function (FIRFilterBlock)
    local reference_instantiate = FIRFilterBlock.instantiate

    function FIRFilterBlock:instantiate(taps, use_fft)
        self.use_fft_argument = use_fft            -- nil / true / false / "fast" / "auto", before firfilter.lua:56-58 turns it into a boolean
        reference_instantiate(self, taps, use_fft)
    end

    function FIRFilterBlock:initialize()
        -- host-side only: initialize() runs before fork() (radio/core/composite.lua:443)
        self.out = self:get_output_type().vector()
        self.stage = nil
    end

    lrhip.device_block(FIRFilterBlock, function (self)
        local input_complex = (self:get_input_type() == types.ComplexFloat32) and 1 or 0
        local taps_complex = (self.taps.data_type == types.ComplexFloat32) and 1 or 0
        return lrhip.lib.lrhip_fir_create(ffi.cast("const float *", self.taps.data), self.taps.length,
                                          taps_complex, input_complex, 1, lrhip.fir_mode(self.use_fft_argument))
    end)

    local function process(self, x)
        return lrhip.execute(self:create_stage(), x, self.out)
    end

    FIRFilterBlock.process_complex_input_complex_taps = process
    FIRFilterBlock.process_complex_input_real_taps = process
    FIRFilterBlock.process_real_input_real_taps = process
    FIRFilterBlock.process_fft_complex_input_complex_taps = process
    FIRFilterBlock.process_fft_complex_input_real_taps = process
    FIRFilterBlock.process_fft_real_input_real_taps = process
end
