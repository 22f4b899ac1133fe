---
-- Device variants of the DFT, IDFT and PSD classes of radio/utilities/spectrum_utils.lua.  Applied by ONE line directly above that file's final
-- `return {DFT = DFT, IDFT = IDFT, PSD = PSD, fftshift = fftshift}` (i.e. after the FFTW / liquid / VOLK / pure-Lua ladders of :68-246, :298-506 and
-- :563-642 have assigned their initialize() / compute*() functions):
--
--     require('radio.core.lrhip').patch_spectrum(DFT, IDFT, PSD)
--
-- DFT.new / IDFT.new / PSD.new (spectrum_utils.lua:27-57, :261-291, :524-561) are unchanged: argument checks, the periodic window and its energy are the
-- reference's host code.  What changes is initialize() - no FFTW plan, no table of exponentials - and compute(): one lrhip_stage_execute() from the
-- object's input vector to its output vector (lrhip_dft_create: forward / inverse with the 1/N of :335-338, real input with the Hermitian half filled in
-- as :109-112 does, real output as :499-503; lrhip_psd_create: window -> DFT -> |X|^2 / (fs sum w^2) -> 10 log10 in one launch, :585-607).
-- The library's transforms are powers of two from 8 to 4 096 points; any other (even) length keeps the reference's implementation, object by object.
-- The device object is created on the first compute() of the PROCESS that computes (blocks build these objects in initialize(), before fork();
-- radio/core/composite.lua:443 vs :569) and re-created if the object is carried across a fork().

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')
local types = require('radio.types')

local function supported(n)
    if n < 8 or n > 4096 then return false end
    while n > 1 do
        if n % 2 ~= 0 then return false end
        n = n / 2
    end
    return true
end

-- the object's stage, in this process
local function stage_of(self, create, what)
    local pid = ffi.C.getpid()
    if self.stage == nil or self.stage_pid ~= pid then
        lrhip.ensure()
        self.stage = ffi.gc(lrhip.check_object(create(self), "Creating lrhip " .. what .. " object"), lrhip.lib.lrhip_stage_destroy)
        self.stage_pid = pid
    end
    return self.stage
end

local function execute(stage, input_samples, output_samples, num_samples)
    local n = tonumber(lrhip.lib.lrhip_stage_execute(stage, input_samples.data, num_samples, output_samples.data, num_samples))
    if n < 0 then
        error("lrhip_stage_execute: " .. ffi.string(lrhip.lib.lrhip_strerror()))
    end
end

local function patch_transform(Class, inverse, what)
    local reference_initialize = Class.initialize
    local reference_compute_complex, reference_compute_real = Class.compute_complex, Class.compute_real

    function Class:initialize()
        if not supported(self.num_samples) then
            return reference_initialize(self)
        end
        self.hip = true
    end

    local function create(self)
        -- data_type is the REAL-capable side: the input of a DFT (spectrum_utils.lua:45), the output of an IDFT (:279)
        return lrhip.lib.lrhip_dft_create(self.num_samples, inverse, (self.data_type == types.Float32) and 1 or 0)
    end

    local function compute(self)
        execute(stage_of(self, create, what), self.input_samples, self.output_samples, self.num_samples)
    end

    -- DFT.new binds self.compute = self.compute_complex / self.compute_real BEFORE it calls initialize() (spectrum_utils.lua:47-54): both names
    -- dispatch per object
    function Class:compute_complex()
        if self.hip then return compute(self) end
        return reference_compute_complex(self)
    end

    function Class:compute_real()
        if self.hip then return compute(self) end
        return reference_compute_real(self)
    end
end

return function (DFT, IDFT, PSD)
    patch_transform(DFT, 0, "DFT")
    patch_transform(IDFT, 1, "IDFT")

    local reference_compute = PSD.compute

    local function create(self)
        return lrhip.lib.lrhip_psd_create(self.num_samples, ffi.cast("const float *", self.window.data), self.sample_rate * self.window_energy,
                                          self.logarithmic and 1 or 0, (self.data_type == types.ComplexFloat32) and 1 or 0, 0)
    end

    function PSD:compute()
        if not supported(self.num_samples) then
            return reference_compute(self)
        end
        execute(stage_of(self, create, "PSD"), self.input_samples, self.output_samples, self.num_samples)
    end
end
