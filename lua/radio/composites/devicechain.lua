---
-- DeviceChainBlock: a maximal linear run of device-capable blocks collapsed into one block / one process /
-- one lrhip_chain_t, so intermediate vectors never leave HBM (one H2D at the head, one D2H at the tail)
-- instead of crossing a UNIX socket per edge (radio/core/pipe.lua:53-69), and adjacent stages fuse inside the library
-- (rotator -> FIR -> downsampler -> discriminator = one launch).
--
-- CompositeBlock:_prepare_to_run (radio/core/composite.lua:426) calls DeviceChainBlock.collapse() on the flattened
-- connection table right after _crawl_connections (:434), before _connect_pipes (:437):
--
--     local all_connections = self:_crawl_connections()
--     local device_chains = {}
--     if platform.features.hip then
--         all_connections, device_chains = require('radio.composites.devicechain').collapse(all_connections)
--     end
--     ...                                                  -- _connect_pipes, _validate_rates, _initialize: unchanged
--     for _, chain in ipairs(device_chains) do chain:initialize() end      -- after self:_initialize() (:443)
--
-- Everything downstream (pipes, rate validation, the global evaluation order, control sockets, fork) sees an ordinary block.
-- The member blocks stay in their composite's own evaluation order, so CompositeBlock:_initialize() still runs their
-- host-side initialize() (tap design, omega); they simply no longer appear in the connection table, so they get no
-- socket and no process.
--
-- @block DeviceChainBlock

local ffi = require('ffi')

local block = require('radio.core.block')
local pipe = require('radio.core.pipe')
local lrhip = require('radio.core.lrhip')

local DeviceChainBlock = block.factory("DeviceChainBlock")

-- batches the ring accumulates process() vectors into (samples), and its depth: IQFileSource hands over 8 192 samples per
-- call (radio/blocks/sources/iqfile.lua:52), a pipe at most 131 072 (radio/core/pipe.lua:495-533)
DeviceChainBlock.batch_samples = 1048576
-- A chain that reads a file itself batches by BYTES: a batch of 2^20 'u8' records is 2 MB, and per-batch fixed costs (a dozen HIP calls) then bound the
-- path - measured through C (tools/host_path_driver.cpp): 12.2 GS/s at 2^20-record batches, 15.7 GS/s at 2^22 for u8; 4.8 GS/s either way for f32le.
-- The batch of a source-headed chain is max(batch_samples, source_batch_bytes / record size).
DeviceChainBlock.source_batch_bytes = 8388608
DeviceChainBlock.ring_depth = 3
-- LIVE flow graphs (an SDR source, a network source): set max_latency (seconds, wall clock) and a batch is also launched once its
-- oldest sample has waited that long, so an RTL-SDR at 1.1 MS/s sees ~22 000-sample batches every 20 ms instead of waiting a second
-- for 2^20 samples, and an audio-rate chain does not sit on minutes of samples.  The bound is kept even when the source STALLS:
-- DeviceChainBlock:run() waits for input only as long as the pending batch may still wait (lrhip_chain_poll_due) and hands the
-- batch on when that wait times out (lrhip_chain_poll) - the reference streams every chunk through as it arrives
-- (radio/core/block.lua:575-602).  0 (the default) = batches run only when full: file and benchmark sources deliver faster than real
-- time, and where a batch is cut decides the Float32 rounding of overlap-save filters and of the single-launch receiver
-- (include/lrhip.h, "chains"), so clock-cut batches would make a file's output bits differ from run to run.
DeviceChainBlock.max_latency = 0
-- ... and since a script that just writes top:connect(RtlSdrSource(...), TunerBlock(...), ...) sets no knob: a chain whose stream comes from a REAL-TIME
-- source - an SDR, a sound card, a network socket, or anything behind a ThrottleBlock - gets live_latency as its bound by default (collapse() looks upstream;
-- the names are the reference's block names, radio/blocks/sources/*.lua).  File, signal and benchmark sources keep 0.  An explicit max_latency wins.
DeviceChainBlock.live_latency = 0.02
DeviceChainBlock.live_sources = {RtlSdrSource = true, AirspySource = true, AirspyHFSource = true, BladeRFSource = true, HackRFSource = true, HydraSDRSource = true,
                                 SDRplaySource = true, SoapySDRSource = true, UHDSource = true, NetworkClientSource = true, NetworkServerSource = true,
                                 PortAudioSource = true, PulseAudioSource = true, ThrottleBlock = true}
-- synchronous = true: no batching at all - every process() vector goes through the chain at once (lrhip_chain_execute: H2D, kernels, D2H, wait) and its
-- output is returned from the same call, as the reference's blocks do (radio/core/block.lua:585).  The lowest latency and the fewest samples per launch;
-- both ends are zero-copy: the vector is DMA'd from the pipe's read buffer and the result into the block's output vector (lrhip.pin_inputs, lrhip.pin).
DeviceChainBlock.synchronous = false
-- Placement: the index of the device this chain's process binds to (lrhip.ensure wraps it over the devices of the box).  nil = the
-- library's default device / LUARADIO_HIP_DEVICE.  collapse() numbers the chains that read ONE fanned-out output port 0, 1, 2, ...
-- (radio/core/block.lua:119-166, radio/core/pipe.lua:617-627: one OutputPort, several readers, each reader its own process), so
-- `source -> 8 x Tuner` runs one branch per GPU on an 8-GPU node; set it by hand on the object collapse() returns to override.
DeviceChainBlock.device = nil
-- the chain's numerical contract (lrhip_chain_create_ex flags): false = fused kernels with the stated roundings of
-- include/lrhip.h (window-relative rotator staging, the polyphase audio tail, consecutive overlap-save filters merged into one
-- filter, the single-launch FM receiver); true = lrhip.CHAIN_EXACT, what the member blocks compute one by one; or a flag number.
-- Set on the class before top:run(), or per chain on the object collapse() returns.
DeviceChainBlock.exact = false

-- on_initialized(chain): called at the end of every chain's initialize() - in the flow graph's PARENT, after the members' own initialize() (files are open)
-- and before fork().  The one place a script can reach the chains collapse() built before they run: it positions a time partition there
-- (chain:partition(first, last), examples/iqfile_wbfm_partitions.lua), sets chain.exact / chain.device / chain.max_latency per chain, ...
DeviceChainBlock.on_initialized = nil

-- a file source / sink with the raw-record hooks of radio/blocks/sources/file_hip.lua / radio/blocks/sinks/file_hip.lua
local function is_raw_source(b)
    return type(b.read_raw) == "function" and type(b.create_stage) == "function" and #b.inputs == 0 and #b.outputs == 1
end
local function is_raw_sink(b)
    return type(b.write_raw) == "function" and type(b.create_stage) == "function" and #b.inputs == 1 and #b.outputs == 0
end

-- The members are device blocks in a row.  The first may be a file source (the chain then has NO input port: it reads the raw records itself, straight
-- into the pinned input slot of its ring) and the last a file sink (NO output port: the chain's output are raw records, written by the sink's fwrite).
function DeviceChainBlock:instantiate(blocks)
    self.blocks = assert(blocks, "Missing argument #1 (blocks)")
    local first, last = blocks[1], blocks[#blocks]
    self.source = is_raw_source(first) and first or nil
    self.sink = is_raw_sink(last) and last or nil
    self:add_type_signature(self.source and {} or {block.Input("in", first:get_input_type())},
                            self.sink and {} or {block.Output("out", last:get_output_type())})
end

function DeviceChainBlock:get_rate()
    local last = self.blocks[#self.blocks]
    if self.sink then last = self.blocks[#self.blocks - 1] end
    return last:get_rate()
end

function DeviceChainBlock:initialize()
    -- the members' host-side initialize() (tap design, fopen() of a file source / sink) has been run by their composite; device objects are created
    -- post-fork, on the first process()
    if not self.sink then self.out = self:get_output_type().vector() end
    self.chain = nil
    self.finished = false
    -- the descriptors the members opened stay open in this block's process (radio/core/composite.lua:594-611 closes everything that is not in
    -- block.files or one of its pipes)
    for _, b in ipairs(self.blocks) do
        for file, _ in pairs(b.files or {}) do self.files[file] = true end
    end
    local hook = self.on_initialized
    if hook then hook(self) end
end

-- start_at() / seek() record what was asked; this applies it to self.chain and returns the sample the source has to deliver from
local function position(self)
    local lib = lrhip.lib
    if self.pending_start ~= nil then
        local seek_sample = ffi.new("unsigned long long[1]")
        if lib.lrhip_chain_start_at(self.chain, self.pending_start, seek_sample) ~= 0 then
            error("lrhip_chain_start_at: " .. ffi.string(lib.lrhip_strerror()))
        end
        return tonumber(seek_sample[0])
    end
    if lib.lrhip_chain_seek(self.chain, self.pending_seek) ~= 0 then
        error("lrhip_chain_seek: " .. ffi.string(lib.lrhip_strerror()))
    end
    return self.pending_seek
end

local function create_chain(self)
    lrhip.ensure(self.device)       -- binds this process to its device BEFORE the members create their stages
    local lib = lrhip.lib
    local stages = ffi.new("lrhip_stage_t *[?]", #self.blocks)
    for i, b in ipairs(self.blocks) do
        stages[i-1] = b:create_stage()
    end
    self.stages = stages        -- keep the array (and through the members, the stages) alive as long as the chain
    local flags = (self.exact == true) and lrhip.CHAIN_EXACT or (tonumber(self.exact) or 0)
    self.chain = ffi.gc(lrhip.check_object(lib.lrhip_chain_create_ex(stages, #self.blocks, flags), "Creating lrhip chain object"),
                        lib.lrhip_chain_destroy)
    self.batch = self.batch_samples
    if self.source then self.batch = math.max(self.batch, math.floor(self.source_batch_bytes / self.source:raw_record_size())) end
    if lib.lrhip_chain_set_ring(self.chain, self.ring_depth, self.batch) ~= 0 then
        error("lrhip_chain_set_ring: " .. ffi.string(lib.lrhip_strerror()))
    end
    if lib.lrhip_chain_set_latency(self.chain, self.source and 0 or self.max_latency) ~= 0 then
        error("lrhip_chain_set_latency: " .. ffi.string(lib.lrhip_strerror()))
    end
    if self.sink then self.raw_size = lib.lrhip_stage_output_size(self.sink:create_stage()) end
    -- a position asked for before this process existed (start_at() / seek() in the flow graph's parent): applied to THIS process's chain
    if self.pending_start ~= nil or self.pending_seek ~= nil then self.position_sample = position(self) end
end

-- Output of a call: `cap` samples of room, fill(ptr, cap) -> n.  Without a sink member the samples land in self.out (returned, as process() output);
-- with one they are raw records in a pinned buffer of the library's, handed to the sink's fwrite (nothing returned: the chain has no output port).
local function deliver(self, cap, fill, what)
    local lib = lrhip.lib
    if self.sink then
        if self.raw == nil or self.raw_cap < cap then
            if self.raw ~= nil then lib.lrhip_host_free(self.raw) end
            self.raw = lrhip.check_object(lib.lrhip_host_alloc(math.max(cap, 1) * self.raw_size), "lrhip_host_alloc")
            self.raw_cap = cap
        end
        local n = tonumber(fill(self.raw, cap))
        if n < 0 then error(what .. ": " .. ffi.string(lib.lrhip_strerror())) end
        self.sink:write_raw(self.raw, n)
        return nil
    end
    self.out:resize(cap)
    if self.synchronous then lrhip.pin(self.out, self.out.data, self.out._capacity * ffi.sizeof(self.out.data_type)) end
    local n = tonumber(fill(self.out.data, cap))
    if n < 0 then error(what .. ": " .. ffi.string(lib.lrhip_strerror())) end
    return self.out:resize(n)
end

-- A chain headed by a file source: one call = keep the ring full, hand on the oldest finished batch.  fread() goes straight into the pinned input of the
-- next ring slot (lrhip_chain_ring_input: no staging copy, batch_samples records per read instead of the reference's 8 192, iqfile.lua:52), submit()
-- starts H2D -> kernels -> D2H of that slot and returns; when the ring is full (or the file has ended) the oldest batch is collected.  After the last
-- batch process() returns nothing: block-generated EOF (radio/core/block.lua:588).
local function process_source(self)
    local lib = lrhip.lib
    local chain = self.chain
    while true do
        local slot = nil
        if not self.source_eof then slot = lib.lrhip_chain_ring_input(chain) end
        if slot == nil then                          -- ring full, or the file has ended (a NULL pointer compares equal to nil)
            if lib.lrhip_chain_in_flight(chain) == 0 then
                self.finished = true
                return nil, true
            end
            local cap = tonumber(lib.lrhip_chain_max_output(chain, self.batch)) + 64
            return deliver(self, cap, function (ptr, room) return lib.lrhip_chain_collect(chain, ptr, room) end, "lrhip_chain_collect"), false
        end
        local n = false
        if self.source.submit_raw and not self.source_streamed then
            n = self.source:submit_raw(chain, self.batch)          -- regular file: read and submitted inside the library
            if n == false then self.source_streamed = true end            -- a FIFO / device: fread() into the slot from here on
        end
        if n == false then
            n = self.source:read_raw(slot, self.batch)
            if n ~= nil and n > 0 and tonumber(lib.lrhip_chain_submit(chain, slot, n)) < 0 then
                error("lrhip_chain_submit: " .. ffi.string(lib.lrhip_strerror()))
            end
        end
        if n == nil then self.source_eof = true end
    end
end

-- process(): append the chunk to the current batch; returns the samples of the batches that have finished (an empty
-- vector while the first ones are in flight - the reference's own FFT FIR delays its output the same way, firfilter.lua:361-398)
function DeviceChainBlock:process(x)
    if self.chain == nil then create_chain(self) end
    local lib = lrhip.lib
    local chain = self.chain
    if self.source then
        local out, eof = process_source(self)
        if eof or self.sink then return end         -- nothing returned: end of the file, or a chain without an output port
        return out
    end
    if self.synchronous then
        lrhip.pin_inputs(self)
        local cap = tonumber(lib.lrhip_chain_max_output(chain, x.length))
        return deliver(self, cap, function (ptr, room) return lib.lrhip_chain_execute(chain, x.data, x.length, ptr, room) end, "lrhip_chain_execute")
    end
    local cap = tonumber(lib.lrhip_chain_push_bound(chain, x.length))
    return deliver(self, cap, function (ptr, room) return lib.lrhip_chain_push(chain, x.data, x.length, ptr, room) end, "lrhip_chain_push")
end

-- The host's wait for input timed out (run() below): hand on what the latency bound releases.
function DeviceChainBlock:poll()
    local lib = lrhip.lib
    local chain = self.chain
    local cap = tonumber(lib.lrhip_chain_push_bound(chain, 0))
    return deliver(self, cap, function (ptr, room) return lib.lrhip_chain_poll(chain, ptr, room) end, "lrhip_chain_poll")
end

-- Block:run (radio/core/block.lua:556-608) with ONE change: in front of the blocking pipe_mux:read() the input descriptors are
-- polled for at most lrhip_chain_poll_due() seconds, and a timeout calls poll() instead of process().  Chains without a latency
-- bound (max_latency = 0) run the reference's loop itself.  Shared with DeviceFanoutBlock (devicefanout.lua), whose poll() launches
-- its partial slab.
local block_run = DeviceChainBlock.run
local function timed_run(self, due_seconds)
    local input_pipes, output_pipes = {}, {}
    for i = 1, #self.inputs do input_pipes[i] = self.inputs[i].pipe end
    for i = 1, #self.outputs do
        output_pipes[i] = {}
        for j = 1, #self.outputs[i].pipes do output_pipes[i][j] = self.outputs[i].pipes[j] end
    end
    local pipe_mux = pipe.PipeMux(input_pipes, output_pipes, self.control_socket)

    local function emit(data_out)           -- block.lua:587-602: false = stop
        if #data_out ~= #self.outputs then return false end
        local eof, eof_pipe, shutdown = pipe_mux:write(data_out)
        if shutdown then return false end
        if eof then
            io.stderr:write(string.format("[%s] Downstream block %s terminated unexpectedly.\n", self.name, eof_pipe.input.owner.name))
            return false
        end
        return true
    end

    while true do
        local timed_out = false
        local due = due_seconds(self)
        if due >= 0 and input_pipes[1]:_read_buffer_count() == 0 then
            -- pollfds[0] = control socket, [1] = the input pipe (radio/core/pipe.lua:417-431)
            local ret = ffi.C.poll(pipe_mux.input_pollfds, 2, math.ceil(due * 1000))
            if ret < 0 then error("poll(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
            timed_out = (ret == 0)
        end
        if timed_out then
            local tail = self:poll()
            if tail ~= nil and tail.length > 0 and not emit({tail}) then break end
        else
            local data_in, eof, shutdown = pipe_mux:read()
            if eof or shutdown then break end
            if not emit({self:process(unpack(data_in))}) then break end
        end
    end

    self:cleanup()
end
DeviceChainBlock.timed_run = timed_run

-- file -> device -> file (source AND sink absorbed): a block without ports.  Block:run would wait on the control socket forever (PipeMux:_read_control,
-- radio/core/pipe.lua:475-493), so this shape runs its own loop: process() until the file has ended, a look at the control socket in between.
local function portless_run(self)
    local pipe_mux = pipe.PipeMux({}, {}, self.control_socket)
    while not self.finished do
        self:process()
        if self.control_socket then
            local ret = ffi.C.poll(pipe_mux.input_pollfds, 1, 0)
            if ret < 0 then error("poll(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
            if ret > 0 then break end           -- shutdown requested (pipe.lua:417-431: pollfds[0] is the control socket)
        end
    end
    self:cleanup()
end

function DeviceChainBlock:run()
    if self.source and self.sink then return portless_run(self) end
    if self.source or not (self.max_latency > 0) then return block_run(self) end
    return timed_run(self, function (b)
        if b.chain == nil then return -1 end
        return tonumber(lrhip.lib.lrhip_chain_poll_due(b.chain))
    end)
end

-- top:run(false) (radio/core/composite.lua:663-693) drives every block with run_once(): true = new samples, false = none, nil = EOF.
local block_run_once = DeviceChainBlock.run_once
function DeviceChainBlock:run_once()
    if self.source and self.sink then
        self:process()
        if self.finished then return nil end
        return true
    end
    return block_run_once(self)
end

-- cleanup() runs when the input reached EOF (radio/core/block.lua:606): run the partly filled batch and hand the tail
-- to the readers of the output port, as process() output would have been (block.lua:585-593)
function DeviceChainBlock:cleanup()
    if self.chain ~= nil and not self.cleaned_up then
        self.cleaned_up = true
        local lib = lrhip.lib
        local chain = self.chain
        if self.source then
            -- a shutdown in the middle of the file: what is in flight is still handed on
            while lib.lrhip_chain_in_flight(chain) > 0 do
                local cap = tonumber(lib.lrhip_chain_max_output(chain, self.batch)) + 64
                local tail = deliver(self, cap, function (ptr, room) return lib.lrhip_chain_collect(chain, ptr, room) end, "lrhip_chain_collect")
                if tail ~= nil and tail.length > 0 then
                    for _, p in ipairs(self.outputs[1].pipes) do p:write(tail) end
                end
            end
        else
            local cap = tonumber(lib.lrhip_chain_push_bound(chain, 0))
            local tail = deliver(self, cap, function (ptr, room) return lib.lrhip_chain_flush(chain, ptr, room) end, "lrhip_chain_flush")
            if tail ~= nil and tail.length > 0 then
                for _, p in ipairs(self.outputs[1].pipes) do p:write(tail) end
            end
        end
        if self.raw ~= nil then
            lib.lrhip_host_free(self.raw)
            self.raw = nil
        end
    end
    -- the absorbed file blocks are no longer in the evaluation order: their cleanup() (fclose, iqfile.lua:118-124) is this block's to call
    if self.source then self.source:cleanup() end
    if self.sink then self.sink:cleanup() end
end

-- Time partitions (INTEGRATION.md 3a; include/lrhip.h "time-axis sharding"): a DeviceChainBlock that starts in the middle of a recording.
-- start_at(first_sample) positions the chain on an aligned sample at or before (first_sample - halo) and returns that sample: the source
-- has to deliver the stream from there on.  The library drops what the replayed samples in front of first_sample produce
-- (lrhip_chain_start_at arms the chain; process() / cleanup() need no special case), so the first sample this block emits is the one
-- the single-process run emits for input sample first_sample.  Chains holding a stage with unbounded memory (AGC, ...) raise an
-- error: they cannot be sharded.
function DeviceChainBlock:start_at(first_sample)
    self.pending_start, self.pending_seek = first_sample, nil
    if self.chain ~= nil then
        self.position_sample = position(self)
        return self.position_sample
    end
    -- one helper process answers all three questions (a helper binds to the device and builds the chain: seconds, not microseconds)
    local seek, h, a = lrhip.in_helper(function ()
        create_chain(self)
        return self.position_sample, tonumber(lrhip.lib.lrhip_chain_halo(self.chain)), tonumber(lrhip.lib.lrhip_chain_shard_align(self.chain))
    end)
    if self.chain == nil then self.partition_halo, self.partition_align = h, a end
    return seek
end

-- The partition helpers next to start_at(): how many input samples a partition replays in front of its first own sample (-1 with an error message for
-- chains with unbounded memory), and the grid partition boundaries should lie on to reproduce the uninterrupted run's tiles (include/lrhip.h).
--
-- WHERE these run (ADVICE r05): a partitioned flow graph asks them in its PARENT, before top:run() forks the block processes - and a parent that has
-- created device objects leaves its children a device they cannot use (lrhip.in_helper).  So while this block has no chain of its own yet the answer
-- comes from a fork()ed helper process (ONE helper answers halo and alignment together, and start_at() brings both along; the answers are kept), the request
-- (start_at / seek) is recorded in the object, and create_chain() applies it to the chain the block's own process builds on its first process().  Once the
-- chain exists (the block's process, or top:run(false)) the calls go to it directly.
local function partition_info(self)
    if self.chain ~= nil then
        return tonumber(lrhip.lib.lrhip_chain_halo(self.chain)), tonumber(lrhip.lib.lrhip_chain_shard_align(self.chain))
    end
    if self.partition_halo == nil then
        local h, a = lrhip.in_helper(function ()
            create_chain(self)
            return tonumber(lrhip.lib.lrhip_chain_halo(self.chain)), tonumber(lrhip.lib.lrhip_chain_shard_align(self.chain))
        end)
        if self.chain ~= nil then return h, a end           -- (in_helper ran in this process: it owns a device, the chain exists now)
        self.partition_halo, self.partition_align = h, a
    end
    return self.partition_halo, self.partition_align
end

function DeviceChainBlock:halo()
    local h = partition_info(self)
    if h < 0 then
        -- chains with unbounded memory (AGC, FM modulator ...): the library's message - from this process's chain, or from one more helper that asks again
        local function complain()
            lrhip.lib.lrhip_chain_halo(self.chain)
            error("lrhip_chain_halo: " .. ffi.string(lrhip.lib.lrhip_strerror()), 0)
        end
        if self.chain ~= nil then complain() end
        lrhip.in_helper(function ()
            create_chain(self)
            complain()
        end)
    end
    return h
end

function DeviceChainBlock:shard_align()
    local _, a = partition_info(self)
    return a
end

-- partition(first_sample, end_sample): this chain - headed by a file source it has absorbed - processes samples [first_sample, end_sample) of the recording
-- and emits exactly what the uninterrupted run emits for them: start_at() arms the chain and says where the replay has to begin, the source's window is set
-- to [that sample, end_sample).  Boundaries on multiples of shard_align() (and of the chain's total decimation) reproduce the uninterrupted run bit for bit
-- where the chain promises that (include/lrhip.h).  Call it from DeviceChainBlock.on_initialized.  Returns the sample the replay starts at.
function DeviceChainBlock:partition(first_sample, end_sample)
    assert(self.source and self.source.set_raw_window, "partition(): the chain has to be headed by a file source (IQFileSource / RealFileSource)")
    local seek = self:start_at(first_sample)
    self.source:set_raw_window(seek, end_sample and (end_sample - seek) or nil)
    return seek
end

-- seek(n0): forget every carried sample, the next vector is sample n0 of the stream (no replay: the caller feeds the halo itself and drops its output).
-- Before the block's process exists the position is only recorded (nothing to ask the device).
function DeviceChainBlock:seek(n0)
    self.pending_start, self.pending_seek = nil, n0
    if self.chain ~= nil then self.position_sample = position(self) end
end

-- reset(): back to the initial state (a flow graph run a second time in the same process, top:run(false) twice)
function DeviceChainBlock:reset()
    if self.chain ~= nil and lrhip.lib.lrhip_chain_reset(self.chain) ~= 0 then
        error("lrhip_chain_reset: " .. ffi.string(lrhip.lib.lrhip_strerror()))
    end
    self.finished, self.source_eof, self.cleaned_up = false, false, false
end

-- kernels launched by the last batch (diagnostic: a fused receiver is ONE launch per batch)
function DeviceChainBlock:last_launches()
    if self.chain == nil then return 0 end
    return tonumber(lrhip.lib.lrhip_chain_last_launches(self.chain))
end

-- a block the library can run as a chain stage: a device variant (create_stage), one input, one output
local function chainable(b)
    if type(b.create_stage) ~= "function" or #b.inputs ~= 1 or #b.outputs ~= 1 then return false end
    if type(b.device_capable) == "function" and not b:device_capable() then return false end      -- e.g. DelayBlock on a Bit stream
    return true
end
DeviceChainBlock.chainable = chainable
DeviceChainBlock.is_raw_source = is_raw_source
DeviceChainBlock.is_raw_sink = is_raw_sink

---
-- Replace every maximal linear run of two or more chainable blocks in the flattened connection table
-- {[InputPort] = OutputPort} (radio/core/composite.lua:343-384) by one DeviceChainBlock; returns the new table and the
-- list of chain blocks created (to be initialized by the caller after the composite's own blocks).
-- A run continues from block a to block b when a's output port has exactly one reader (b) and both are chainable.  A run is extended by the file source
-- that feeds its first block (when that block is the source's only reader) and by the file sink that reads its last block (when it is the only reader):
-- the raw records then cross the host untouched (radio/blocks/sources/file_hip.lua, radio/blocks/sinks/file_hip.lua).
function DeviceChainBlock.collapse(connections)
    -- readers of every output port
    local readers = {}
    for input, output in pairs(connections) do
        readers[output] = readers[output] or {}
        table.insert(readers[output], input)
    end
    local function sole_reader(b)       -- the block reading b's output port, if it is the only one
        if #b.outputs ~= 1 then return nil end
        local r = readers[b.outputs[1]]
        if r and #r == 1 then return r[1].owner end
        return nil
    end
    local function sole_writer(b)       -- the block feeding b's input port
        if #b.inputs ~= 1 then return nil end
        local output = connections[b.inputs[1]]
        return output and output.owner or nil
    end
    local function links(a, b)          -- a -> b is an interior edge of a run
        return a and b and chainable(a) and chainable(b) and sole_reader(a) == b
    end

    -- run heads: chainable blocks whose upstream edge is not a link
    local seen, runs = {}, {}
    for input, _ in pairs(connections) do
        local b = input.owner
        if not seen[b] and chainable(b) and not links(sole_writer(b), b) then
            seen[b] = true
            local run = {b}
            while links(run[#run], sole_reader(run[#run])) do
                run[#run + 1] = sole_reader(run[#run])
                seen[run[#run]] = true
            end
            -- the file source in front, the file sink behind
            local writer, reader = sole_writer(run[1]), sole_reader(run[#run])
            if writer and is_raw_source(writer) and sole_reader(writer) == run[1] then
                table.insert(run, 1, writer)
                seen[writer] = true
            end
            if reader and is_raw_sink(reader) then
                run[#run + 1] = reader
                seen[reader] = true
            end
            if #run >= 2 then runs[#runs + 1] = run end
        end
    end
    -- file source -> file sink with nothing in between (a format conversion)
    for input, output in pairs(connections) do
        local b, w = input.owner, output.owner
        if not seen[b] and not seen[w] and is_raw_sink(b) and is_raw_source(w) and sole_reader(w) == b then
            seen[b], seen[w] = true, true
            runs[#runs + 1] = {w, b}
        end
    end

    local result, chains = {}, {}
    local chain_of = {}                 -- first member of a run -> its DeviceChainBlock
    local port_of = {}                  -- output port of a run's last member -> the chain's output port
    for input, output in pairs(connections) do result[input] = output end
    for _, run in ipairs(runs) do
        local first, last = run[1], run[#run]
        local chain = DeviceChainBlock(run)
        chain:differentiate(chain.source and {} or {first:get_input_type()})
        -- upstream: the chain's input reads what the first member read
        if not chain.source then
            result[chain.inputs[1]] = connections[first.inputs[1]]
            result[first.inputs[1]] = nil
        end
        -- downstream: every reader of the last member now reads the chain (re-pointed below, once every chain exists: a reader may itself be the
        -- first member of another chain, whose input entry is rewritten too)
        if not chain.sink then port_of[last.outputs[1]] = chain.outputs[1] end
        -- interior edges leave the table (no socket, no process); the members keep rate-only pipes so that their
        -- get_rate() still walks upstream (radio/core/block.lua:383-390, radio/core/pipe.lua:36-38)
        for i = 2, #run do
            result[run[i].inputs[1]] = nil
            run[i].inputs[1].pipe = pipe.Pipe(run[i-1].outputs[1], run[i].inputs[1])
        end
        if not chain.source then
            first.inputs[1].pipe = {get_rate = function () return chain.inputs[1].pipe:get_rate() end}
        end
        -- a live stream upstream (walk the first inputs up to the source): bounded latency unless the script set one
        if not chain.source and rawget(chain, "max_latency") == nil then
            local up, hops = connections[first.inputs[1]], 0
            while up ~= nil and hops < 64 do
                local b = up.owner
                if DeviceChainBlock.live_sources[b.name] then
                    chain.max_latency = DeviceChainBlock.live_latency
                    break
                end
                up = (#b.inputs >= 1) and connections[b.inputs[1]] or nil
                hops = hops + 1
            end
        end
        chains[#chains + 1] = chain
        chain_of[first] = chain
    end
    for input, output in pairs(result) do
        if port_of[output] then result[input] = port_of[output] end
    end

    -- placement: the chains that read ONE fanned-out output port are numbered 0, 1, 2, ... (in table order, which is as arbitrary as the
    -- reference's own evaluation order of parallel branches: build_dependency_graph walks the same pairs(), composite.lua:386-424)
    for output, inputs in pairs(readers) do
        local branches = {}
        for _, input in ipairs(inputs) do
            local chain = chain_of[input.owner]
            if chain then branches[#branches + 1] = chain end
        end
        if #branches >= 2 then
            for k, chain in ipairs(branches) do
                chain.device = k - 1
                chain.fanout_port = output
            end
        end
    end
    return result, chains
end

return DeviceChainBlock
