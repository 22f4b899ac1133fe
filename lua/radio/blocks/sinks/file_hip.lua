---
-- Device variants of IQFileSink (radio/blocks/sinks/iqfile.lua), RealFileSink (radio/blocks/sinks/realfile.lua) and GnuplotSpectrumSink
-- (radio/blocks/sinks/gnuplotspectrum.lua).  Each is applied by ONE line directly above the reference file's final `return <Block>`:
--
--     require('radio.core.lrhip').patch('iqfilesink', IQFileSink)
--     require('radio.core.lrhip').patch('realfilesink', RealFileSink)
--     require('radio.core.lrhip').patch('gnuplotspectrum', GnuplotSpectrumSink)
--
-- File sinks: a sink fed by host blocks keeps the reference's process() (the per-sample `x * scale + offset` loop, byte swap and fwrite of
-- iqfile.lua:68-85).  What is ADDED is what DeviceChainBlock.collapse() needs to absorb the sink as the TAIL of a device chain:
--
--   create_stage()      lrhip_format_pack_create(format, complex): samples -> raw records on the device (C conversion = truncation for the integer
--                       formats, byte swap for the big-endian ones), so the chain's D2H carries 2 bytes per complex sample for 'u8' instead of 8;
--   write_raw(src, n)   the reference's fwrite + error check (iqfile.lua:80-84) of n finished records.
--
-- GnuplotSpectrumSink: process() keeps the reference's frame / plot cadence (gnuplotspectrum.lua:140-186) but the frames never reach the interpreter -
-- window, DFT, |X|^2 / (fs sum w^2), 10 log10, fftshift and the running sum of Welch's method all happen on the device (lrhip_welch_create, one call per
-- process() vector instead of one PSD per 1 024 samples); the host reads the num_samples-point average back only when a plot is due.

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')
local types = require('radio.types')

local M = {}

local function patch_file_sink(Sink, complex_in)
    local reference_instantiate = Sink.instantiate

    function Sink:instantiate(file, format)
        self.format_name = format
        reference_instantiate(self, file, format)
    end

    lrhip.device_block(Sink, function (self)
        return lrhip.lib.lrhip_format_pack_create(self.format_name, complex_in)
    end)

    function Sink:raw_record_size()
        return ffi.sizeof((complex_in == 1) and self.format.complex_ctype or self.format.real_ctype)
    end

    function Sink:write_raw(src, num_samples)
        if num_samples == 0 then return end
        local written = tonumber(ffi.C.fwrite(src, self:raw_record_size(), num_samples, self.file))
        if written ~= num_samples then
            error("fwrite(): " .. ffi.string(ffi.C.strerror(ffi.errno())))
        end
    end
end

function M.patch_iqfilesink(IQFileSink) patch_file_sink(IQFileSink, 1) end
function M.patch_realfilesink(RealFileSink) patch_file_sink(RealFileSink, 0) end

function M.patch_gnuplotspectrum(GnuplotSpectrumSink)
    -- the Welch stage of this sink: created in the block's own process, on the first vector (initialize_gnuplot() has then computed the overlap, the plot
    -- interval and - through spectrum_utils.PSD - the periodic window and its energy, gnuplotspectrum.lua:121-137, spectrum_utils.lua:547-553)
    lrhip.device_block(GnuplotSpectrumSink, function (self)
        local psd = self.psd
        return lrhip.lib.lrhip_welch_create(self.num_samples, ffi.cast("const float *", psd.window.data), psd.sample_rate * psd.window_energy, 1,
                                            (self:get_input_type() == types.ComplexFloat32) and 1 or 0, self.num_overlap)
    end)

    local function feed(self, x, first, count)
        if count == 0 then return end
        local n = tonumber(lrhip.lib.lrhip_stage_execute(self:create_stage(), x.data + first, count, nil, 0))
        if n < 0 then error("lrhip_stage_execute: " .. ffi.string(lrhip.lib.lrhip_strerror())) end
    end

    function GnuplotSpectrumSink:process(x)
        if not self.gnuplot_f then
            self:initialize_gnuplot()
            self.frames_pending = 0
        end

        -- The reference's loop (gnuplotspectrum.lua:148-185) on its COUNTERS only: where frames complete and where a plot falls.  The samples between
        -- two plots go to the device in one call.
        local fed, sample_index = 0, 0
        while sample_index < x.length do
            local num = math.min(self.num_samples - self.state_index, x.length - sample_index)
            self.state_index = self.state_index + num
            self.sample_count = self.sample_count + num
            sample_index = sample_index + num

            if self.state_index == self.num_samples then
                self.frames_pending = self.frames_pending + 1
                self.state_index = self.num_overlap
            end

            if self.sample_count >= self.num_plot_update and self.frames_pending > 0 then
                feed(self, x, fed, sample_index - fed)
                fed = sample_index

                -- the average of the frames since the last plot (sum / count, fftshifted), minus the reference level (gnuplotspectrum.lua:172-174)
                local frames = tonumber(lrhip.lib.lrhip_welch_read(self:create_stage(), ffi.cast("float *", self.state_psd_average.data), 1))
                if frames < 0 then error("lrhip_welch_read: " .. ffi.string(lrhip.lib.lrhip_strerror())) end
                assert(frames == self.frames_pending, "Welch frame count mismatch")
                if self.reference_level ~= 0 then
                    for i = 0, self.state_psd_average.length-1 do
                        self.state_psd_average.data[i].value = self.state_psd_average.data[i].value - self.reference_level
                    end
                end

                self:write_gnuplot(self.plot_str)
                self:write_gnuplot(ffi.string(self.state_psd_average.data, self.state_psd_average.length*ffi.sizeof(self.state_psd_average.data[0])))

                self.frames_pending = 0
                self.sample_count = 0
            end
        end
        feed(self, x, fed, x.length - fed)
    end
end

return M
