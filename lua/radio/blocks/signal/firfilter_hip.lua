---
-- Device variant of FIRFilterBlock's process path.  Applied by ONE line directly above the final `return FIRFilterBlock` of
-- radio/blocks/signal/firfilter.lua (i.e. after :492):
--
--     require('radio.core.lrhip').patch('firfilter', FIRFilterBlock)
--
-- NOT as a first branch of the dot-product ladder at :88: the file's second ladder (:313-492) assigns
-- FIRFilterBlock.process_fft_* = FIRFilterBlock.process_fft at :400-402 / :488-490, after the first one, and instantiate()
-- (:56-66) binds exactly those names whenever use_fft is truthy - the default with FFTW installed.  At the end of the file
-- nothing runs after the patch, and every name a type signature can bind (six process_* functions) is the device function.
-- The type signatures (firfilter.lua:59-74) are unchanged; instantiate() is wrapped only to remember whether the
-- caller chose use_fft at all (nil = let the library pick the fast arithmetic, lrhip.fir_mode()); initialize() never builds the
-- FFTW plans of initialize_fft() (:320-359) nor the VOLK state of initialize_dotprod().
-- A stand-alone FIRFilterBlock(taps, true) therefore runs mode 1: the library's overlap-save kernel WITH the reference's block
-- framing (only whole blocks of N - M + 1 samples are emitted, firfilter.lua:361-398); nil runs mode 3 (overlap-save arithmetic,
-- one output per input, from 48 taps up).

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
        return lrhip.execute(self:create_stage(), x, self.out, self)
    end

    FIRFilterBlock.process_complex_input_complex_taps = process
    FIRFilterBlock.process_complex_input_real_taps = process
    FIRFilterBlock.process_real_input_real_taps = process
    FIRFilterBlock.process_fft_complex_input_complex_taps = process
    FIRFilterBlock.process_fft_complex_input_real_taps = process
    FIRFilterBlock.process_fft_real_input_real_taps = process
end
