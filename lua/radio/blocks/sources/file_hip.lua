---
-- Device variants of IQFileSource (radio/blocks/sources/iqfile.lua) and RealFileSource (radio/blocks/sources/realfile.lua).  Applied by ONE line
-- directly above the final `return IQFileSource` / `return RealFileSource` of the reference files:
--
--     require('radio.core.lrhip').patch('iqfilesource', IQFileSource)
--     require('radio.core.lrhip').patch('realfilesource', RealFileSource)
--
-- What changes: nothing for a source that feeds host blocks - instantiate(), initialize(), process() (fread of 8 192 records, byte swap, the
-- per-sample `(value - offset) / scale` loop of iqfile.lua:82-116) and cleanup() are the reference's.  What is ADDED is what DeviceChainBlock.collapse()
-- (radio/composites/devicechain.lua) needs to absorb the source as the HEAD of a device chain:
--
--   create_stage()      lrhip_format_convert_create(format, complex): the conversion of the raw records - byte swap, offset and scale of
--                       radio/utilities/format_utils.lua:82-97 - as the chain's first stage, where it folds into the first filter's launch
--                       (u8 / s8 / s16le records -> the receiver / tuner kernels read the records themselves);
--   raw_record_size()   bytes per raw record (2 for 'u8' IQ, 8 for 'f32le' IQ, 4 for 'f32le' real, ...);
--   read_raw(dst, max)  the reference's fread + EOF / repeat / ferror handling (iqfile.lua:82-96) WITHOUT the conversion loop, straight into `dst` -
--                       which DeviceChainBlock points at the pinned input slot of the chain's ring (lrhip_chain_ring_input), so an RTL-SDR style
--                       capture crosses the host once, as 2 bytes per complex sample, and is never touched by the interpreter;
--   submit_raw(chain, max)  the same for a regular file with the read done INSIDE the library, on its copy threads (lrhip_chain_submit_fd): what
--                       DeviceChainBlock uses first; read_raw() is the path for FIFOs and devices.
--
-- The format NAME is what the library takes; the reference keeps only the table entry (iqfile.lua:48), so instantiate() is wrapped to remember it.

local ffi = require('ffi')

local lrhip = require('radio.core.lrhip')

ffi.cdef[[
long ftell(FILE *stream);
]]

local M = {}

local function patch_source(Source, complex_out)
    local reference_instantiate = Source.instantiate

    function Source:instantiate(file, format, rate, repeat_on_eof)
        self.format_name = format
        reference_instantiate(self, file, format, rate, repeat_on_eof)
    end

    lrhip.device_block(Source, function (self)
        return lrhip.lib.lrhip_format_convert_create(self.format_name, complex_out)
    end)

    function Source:raw_record_size()
        return ffi.sizeof((complex_out == 1) and self.format.complex_ctype or self.format.real_ctype)
    end

    -- A TIME PARTITION of the recording (INTEGRATION.md 3a, DeviceChainBlock:partition): deliver records [first, first + count) only (count nil: to the end
    -- of the file).  Runs in the flow graph's parent, after initialize() has opened the file; the position travels with the FILE * into the chain's process.
    function Source:set_raw_window(first, count)
        if ffi.C.fseek(self.file, first * self:raw_record_size(), ffi.C.SEEK_SET) ~= 0 then
            error("fseek(): " .. ffi.string(ffi.C.strerror(ffi.errno())))
        end
        self.raw_fd, self.raw_offset = nil, nil          -- submit_raw() takes its offset from the stream again
        self.raw_left = count
    end

    -- the records a window still allows (max_records itself without one); nil = the window is used up: end of the partition
    local function window(self, max_records)
        if self.raw_left == nil then return max_records end
        if self.raw_left <= 0 then return nil end
        return math.min(max_records, self.raw_left)
    end

    -- up to `max_records` raw records into `dst`; returns the number read (0 right after a rewind, as the reference returns an empty vector from
    -- that call, iqfile.lua:86-90), or nil at the end of the file
    function Source:read_raw(dst, max_records)
        max_records = window(self, max_records)
        if max_records == nil then return nil end
        local num_samples = tonumber(ffi.C.fread(dst, self:raw_record_size(), max_records, self.file))
        if self.raw_left ~= nil then
            self.raw_left = self.raw_left - num_samples
            if num_samples < max_records then self.raw_left = 0 end           -- a window that reaches past the end of the file ends with the file (no repeat)
            if num_samples == 0 then return nil end
            return num_samples
        end
        if num_samples < max_records then
            if num_samples == 0 and ffi.C.feof(self.file) ~= 0 then
                if self.repeat_on_eof then
                    ffi.C.rewind(self.file)
                else
                    return nil
                end
            else
                if ffi.C.ferror(self.file) ~= 0 then
                    error("fread(): " .. ffi.string(ffi.C.strerror(ffi.errno())))
                end
            end
        end
        return num_samples
    end

    -- The same step for a REGULAR file, without fread(): the library reads the records from the page cache into the pinned input of the chain's next ring
    -- slot itself - positional reads on its copy threads, several times what one read(2) stream delivers - and submits the slot
    -- (lrhip_chain_submit_fd).  Returns the number of records submitted (0 right after a rewind), nil at the end of the file, false when the descriptor is
    -- not a regular file (a FIFO, a character device: DeviceChainBlock then uses read_raw() for the rest of the run).
    -- The offset starts at the STREAM's position when the chain took over - ftell(), which counts what stdio has buffered, not the descriptor's lseek()
    -- position (a source built on a caller's FILE * / fd that was read from before: the kernel offset is ahead of the stream by the unread part of the
    -- buffer) - and a repeating source goes back to byte 0, as the reference's rewind() and read_raw() do (iqfile.lua:86-90).
    function Source:submit_raw(chain, max_records)
        if self.raw_fd == nil then
            self.raw_fd = ffi.C.fileno(self.file)
            self.raw_offset = tonumber(ffi.C.ftell(self.file))
            if self.raw_offset < 0 then self.raw_offset = tonumber(ffi.C.lseek(self.raw_fd, 0, 1)) end          -- SEEK_CUR
            if self.raw_offset < 0 then return false end
        end
        max_records = window(self, max_records)
        if max_records == nil then return nil end
        local n = tonumber(lrhip.lib.lrhip_chain_submit_fd(chain, self.raw_fd, self.raw_offset, max_records))
        if n == -4 then return false end
        if n < 0 then error("lrhip_chain_submit_fd: " .. ffi.string(lrhip.lib.lrhip_strerror())) end
        if self.raw_left ~= nil then
            if n == 0 then self.raw_left = 0; return nil end                 -- (no repeat inside a window)
            self.raw_left = self.raw_left - n
            self.raw_offset = self.raw_offset + n * self:raw_record_size()
            return n
        end
        if n == 0 then
            if not self.repeat_on_eof then return nil end
            self.raw_offset = 0
            return 0
        end
        self.raw_offset = self.raw_offset + n * self:raw_record_size()
        return n
    end
end

function M.patch_iqfilesource(IQFileSource) patch_source(IQFileSource, 1) end
function M.patch_realfilesource(RealFileSource) patch_source(RealFileSource, 0) end

return M
