---
-- DeviceGraphBlock: a connected subgraph of device blocks that contains a JOIN - a two-input block (MultiplyBlock, MultiplyConjugateBlock, AddBlock,
-- SubtractBlock, FloatToComplexBlock; radio/blocks/signal/multiplyconjugate.lua:41-47) - collapsed into ONE block / one process with several inputs and
-- one output.  Every edge inside it is a device vector: what the reference moves through a UNIX socket per edge (radio/core/pipe.lua:53-69,
-- radio/core/block.lua:119-166) and what a stand-alone device block moves over PCIe per edge (lrhip.execute2: two uploads and one download per call)
-- stays in HBM.  The reference's own end-to-end test graph (tests/top_spec.lua:13-54)
--
--     IQFileSource x 2 -> MultiplyConjugate -> LowpassFilter -> FrequencyDiscriminator -> Decimator -> RawFileSink
--
-- becomes: two sources -> ONE DeviceGraphBlock (two inputs, one output) -> sink; a port read by several blocks of the subgraph (source -> {filter, join})
-- is uploaded once and read in place by all of them.  Linear runs inside the subgraph are built as lrhip_chain_t, so they fuse exactly as in a
-- DeviceChainBlock (filter -> discriminator -> filter -> downsampler = the launches of that chain).  It is the Lua twin of luaradio_amd/graph.py
-- (DeviceGraph), driven with the same entry points: lrhip_stage_execute_device / lrhip_stage_execute2_device / lrhip_chain_execute_device.
--
-- CompositeBlock:_prepare_to_run calls DeviceGraph.collapse() FIRST (tools/apply_lua_binding.py), then DeviceChainBlock.collapse() and
-- DeviceFanout.collapse() on what is left: a component WITHOUT a join stays a set of linear chains, which may spread over the GPUs of the box.
--
--     all_connections, device_chains = require('radio.composites.devicegraph').collapse(all_connections)
--
-- Inputs are accumulated into batches like a DeviceChainBlock's (batch_samples; the vectors PipeMux:_read_multiple hands to process() have one common
-- length, radio/core/pipe.lua:535-583): process() returns empty vectors until a batch has run; cleanup() runs the partial batch and hands its output to
-- the readers.  A join whose inputs momentarily differ in length (an overlap-save filter with the reference's block framing on one side) consumes the
-- shorter count and keeps the excess of the longer input on the device for the next batch, like a pipe would.
--
-- Round 5: when EVERY graph input is a file source (IQFileSource / RealFileSource with the raw-record hooks of radio/blocks/sources/file_hip.lua) that only
-- members of the subgraph read - tests/top_spec.lua is `IQFileSource x 2 -> ...` - the sources are absorbed like a DeviceChainBlock's: the block has no
-- input port, reads batch_samples raw records per source and batch itself (fread() into pinned memory, no interpreter per sample, no socket), uploads the
-- RECORDS (2 bytes per sample for 'u8' instead of 8) and converts them on the device (lrhip_format_convert_create) in front of the first members.  A source
-- that comes up short keeps the others' surplus for the next batch; the first source at its end ends the block, as PipeMux:_read_multiple would.
-- DeviceGraph.absorb_sources = false (or LUARADIO_HIP_NO_GRAPH_SOURCES=1) keeps the input ports.  Likewise the file SINK that is the only reader of the
-- port leaving the subgraph (tests/top_spec.lua ends in a RawFileSink): the output is packed into the sink's records on the device and written by its
-- fwrite (write_raw, radio/blocks/sinks/file_hip.lua) - no output port; with sources AND sink absorbed the reference's end-to-end graph is ONE block without
-- ports, which runs its own loop (Block:run would wait on pipes it does not have).
--
-- Restrictions: exactly one output port leaves the subgraph (a block has ONE rate, radio/core/pipe.lua:36-38: Pipe:get_rate asks the owner of the output
-- port); members have one or two inputs and one output.  Anything else keeps the chains and the stand-alone blocks it had.
-- DeviceGraph.enabled = false (or LUARADIO_HIP_NO_GRAPH=1) switches the rewrite off.
--
-- @module radio.composites.devicegraph

local ffi = require('ffi')

local block = require('radio.core.block')
local pipe = require('radio.core.pipe')
local lrhip = require('radio.core.lrhip')
local DeviceChainBlock = require('radio.composites.devicechain')

local M = {enabled = not os.getenv("LUARADIO_HIP_NO_GRAPH"), absorb_sources = not os.getenv("LUARADIO_HIP_NO_GRAPH_SOURCES")}

local DeviceGraphBlock = block.factory("DeviceGraphBlock")

DeviceGraphBlock.batch_samples = 1048576
DeviceGraphBlock.device = nil

-- members: the blocks in topological order.  wiring[b][j] = {input = i} (graph input i feeds input j of member b) or {member = a} (member a's output).
-- output_member: the member whose output port leaves the subgraph.  sources (optional): the file source behind every graph input - the block then has no
-- input ports and reads the files itself.
function DeviceGraphBlock:instantiate(members, wiring, input_types, output_member, sources, sink)
    self.blocks = assert(members, "Missing argument #1 (members)")
    self.wiring = assert(wiring, "Missing argument #2 (wiring)")
    self.output_member = assert(output_member, "Missing argument #4 (output member)")
    self.input_types = assert(input_types, "Missing argument #3 (input types)")
    self.sources = sources
    self.sink = sink
    local inputs = {}
    if not sources then
        for i, data_type in ipairs(input_types) do inputs[i] = block.Input("in" .. i, data_type) end
    end
    self:add_type_signature(inputs, sink and {} or {block.Output("out", output_member:get_output_type())})
end

function DeviceGraphBlock:get_rate()
    return self.output_member:get_rate()
end

function DeviceGraphBlock:initialize()
    self.out = self.output_member:get_output_type().vector()
    self.started = false
    self.finished = false
    -- the descriptors the absorbed sources opened stay open in this block's process (radio/core/composite.lua:594-611 closes everything that is not in
    -- block.files or one of its pipes)
    for _, src in ipairs(self.sources or {}) do
        for file, _ in pairs(src.files or {}) do self.files[file] = true end
    end
    if self.sink then
        for file, _ in pairs(self.sink.files or {}) do self.files[file] = true end
    end
end

local function check(rc, what)
    if rc ~= 0 then error(what .. ": " .. ffi.string(lrhip.lib.lrhip_strerror())) end
end

-- a growable device vector
local function reserve(buf, bytes)
    if buf.ptr == nil or buf.cap < bytes then
        local lib = lrhip.lib
        -- the old buffer may still be read by kernels in flight on the library stream
        if buf.ptr ~= nil then
            check(lib.lrhip_synchronize(), "lrhip_synchronize")
            lib.lrhip_free(buf.ptr)
        end
        buf.cap = math.max(bytes, 256)
        buf.ptr = lrhip.check_object(lib.lrhip_malloc(buf.cap), "lrhip_malloc")
    end
    return buf.ptr
end

local function start(self)
    local lib = lrhip.lib
    lrhip.ensure(self.device)
    self.batch = self.batch_samples          -- fixed from here on: the staging buffers are sized for it
    -- readers per member (inside the subgraph)
    local readers = {}
    for _, b in ipairs(self.blocks) do
        for _, w in ipairs(self.wiring[b]) do
            if w.member then readers[w.member] = (readers[w.member] or 0) + 1 end
        end
    end
    -- linear runs: a -> b is interior when b has one input, a one input too, and b is a's only reader (the port that leaves the subgraph has one more)
    local function single(b) return #self.wiring[b] == 1 end
    local function only_reader_of(a) return (readers[a] or 0) + ((a == self.output_member) and 1 or 0) == 1 end
    local next_of = {}
    for _, b in ipairs(self.blocks) do
        local w = self.wiring[b][1]
        if single(b) and w.member and single(w.member) and only_reader_of(w.member) then next_of[w.member] = b end
    end
    local in_run = {}
    self.steps = {}                         -- what one batch executes, in order
    for _, b in ipairs(self.blocks) do
        if not in_run[b] then
            local run = {b}
            while single(run[#run]) and next_of[run[#run]] do
                run[#run + 1] = next_of[run[#run]]
                in_run[run[#run]] = true
            end
            local step = {head = b, tail = run[#run], out = {}, count = 0}
            if #run >= 2 then
                local stages = ffi.new("lrhip_stage_t *[?]", #run)
                for i, m in ipairs(run) do stages[i-1] = m:create_stage() end
                step.stages = stages
                local exact = DeviceChainBlock.exact
                local flags = (exact == true) and lrhip.CHAIN_EXACT or (tonumber(exact) or 0)
                step.chain = ffi.gc(lrhip.check_object(lib.lrhip_chain_create_ex(stages, #run, flags), "Creating lrhip chain object"), lib.lrhip_chain_destroy)
            else
                step.stage = b:create_stage()
            end
            step.out_size = lib.lrhip_stage_output_size(step.tail:create_stage())
            if #self.wiring[b] == 2 then
                step.pending = {{buf = {}, count = 0}, {buf = {}, count = 0}}
                step.joined = {{}, {}}
                step.in_size = {}
            end
            self.steps[#self.steps + 1] = step
        end
    end
    self.step_of = {}
    for _, step in ipairs(self.steps) do self.step_of[step.tail] = step end
    -- graph inputs: pinned staging for the batch, a device vector each
    self.input = {}
    for i = 1, #self.input_types do
        local size = ffi.sizeof(self.input_types[i])
        if self.sources then
            -- the source's records: pinned staging for a batch of them, their device copy, the conversion stage (format_utils.lua:82-97 on the device)
            local src = self.sources[i]
            local raw_size = src:raw_record_size()
            self.input[i] = {size = size, raw_size = raw_size, have = 0, eof = false, dev = {}, raw_dev = {}, fmt = src:create_stage(),
                             raw_staging = lrhip.check_object(lib.lrhip_host_alloc(self.batch * raw_size), "lrhip_host_alloc")}
        else
            self.input[i] = {size = size, staging = lrhip.check_object(lib.lrhip_host_alloc(self.batch * size), "lrhip_host_alloc"), dev = {}}
        end
    end
    self.fill = 0
    self.out_size = ffi.sizeof(self.output_member:get_output_type())
    if self.sink then
        self.pack = self.sink:create_stage()                 -- lrhip_format_pack_create: samples -> the file's records (format_utils.lua:99-120), on the device
        self.raw_size = lib.lrhip_stage_output_size(self.pack)
        self.raw_dev, self.raw, self.raw_cap = {}, nil, 0
    end
    self.started = true
end

-- where input j of a step's head comes from: device pointer, sample count, bytes per sample
local function source_of(self, w)
    if w.input then
        local inp = self.input[w.input]
        return inp.dev.ptr, self.batch_count, inp.size
    end
    local step = self.step_of[w.member]
    return step.out.ptr, step.count, step.out_size
end

-- [pending | new] of both inputs of a join, cut to the common count; the excess of the longer side stays pending (device to device)
local function join_inputs(self, step)
    local lib = lrhip.lib
    local ptrs, avail, new = {}, {}, {}
    for j = 1, 2 do
        local ptr, count, size = source_of(self, self.wiring[step.head][j])
        new[j] = {ptr = ptr, count = count, size = size}
        avail[j] = step.pending[j].count + count
    end
    local take = math.min(avail[1], avail[2])
    for j = 1, 2 do
        local pend, size = step.pending[j], new[j].size
        if pend.count == 0 and avail[j] == take then
            ptrs[j] = new[j].ptr                -- the common case: the producer's vector, read in place
        else
            local joined = step.joined[j]
            local base = ffi.cast("char *", reserve(joined, math.max(avail[j], 1) * size))
            if pend.count > 0 then check(lib.lrhip_memcpy_d2d(base, pend.buf.ptr, pend.count * size), "lrhip_memcpy_d2d") end
            if new[j].count > 0 then check(lib.lrhip_memcpy_d2d(base + pend.count * size, new[j].ptr, new[j].count * size), "lrhip_memcpy_d2d") end
            local left = avail[j] - take
            if left > 0 then
                -- (the copy into `joined` above reads the old pending vector: same stream, so the order holds)
                check(lib.lrhip_memcpy_d2d(reserve(pend.buf, left * size), base + take * size, left * size), "lrhip_memcpy_d2d")
            end
            pend.count = left
            ptrs[j] = base
        end
    end
    return ptrs, take
end

-- one batch: upload the accumulated inputs, run every step, download the output
local function run_batch(self)
    local lib = lrhip.lib
    local n = self.fill
    self.fill = 0
    self.batch_count = n
    for _, inp in ipairs(self.input) do
        reserve(inp.dev, math.max(n, 1) * inp.size)
        if n > 0 and inp.fmt then
            reserve(inp.raw_dev, n * inp.raw_size)
            check(lib.lrhip_memcpy_h2d(inp.raw_dev.ptr, inp.raw_staging, n * inp.raw_size), "lrhip_memcpy_h2d")
            if tonumber(lib.lrhip_stage_execute_device(inp.fmt, inp.raw_dev.ptr, n, inp.dev.ptr, n)) ~= n then
                error("DeviceGraphBlock (file records): " .. ffi.string(lib.lrhip_strerror()))
            end
        elseif n > 0 then
            check(lib.lrhip_memcpy_h2d(inp.dev.ptr, inp.staging, n * inp.size), "lrhip_memcpy_h2d")
        end
    end
    for _, step in ipairs(self.steps) do
        local got
        if step.pending then
            local ptrs, count = join_inputs(self, step)
            reserve(step.out, math.max(count, 1) * step.out_size)
            got = (count > 0) and tonumber(lib.lrhip_stage_execute2_device(step.stage, ptrs[1], ptrs[2], count, step.out.ptr, count)) or 0
        else
            local ptr, count = source_of(self, self.wiring[step.head][1])
            if step.chain then
                local cap = tonumber(lib.lrhip_chain_max_output(step.chain, count)) + 16
                reserve(step.out, cap * step.out_size)
                got = (count > 0) and tonumber(lib.lrhip_chain_execute_device(step.chain, ptr, count, step.out.ptr, cap)) or 0
            else
                local cap = tonumber(lib.lrhip_stage_max_output(step.stage, count)) + 16
                reserve(step.out, cap * step.out_size)
                got = (count > 0) and tonumber(lib.lrhip_stage_execute_device(step.stage, ptr, count, step.out.ptr, cap)) or 0
            end
        end
        if got < 0 then error("DeviceGraphBlock " .. step.head.name .. ": " .. ffi.string(lib.lrhip_strerror())) end
        step.count = got
    end
    local last = self.step_of[self.output_member]
    if self.sink then
        -- pack on the device, download the records, hand them to the sink's fwrite
        if last.count > 0 then
            reserve(self.raw_dev, last.count * self.raw_size)
            if self.raw == nil or self.raw_cap < last.count then
                if self.raw ~= nil then lib.lrhip_host_free(self.raw) end
                self.raw = lrhip.check_object(lib.lrhip_host_alloc(last.count * self.raw_size), "lrhip_host_alloc")
                self.raw_cap = last.count
            end
            if tonumber(lib.lrhip_stage_execute_device(self.pack, last.out.ptr, last.count, self.raw_dev.ptr, last.count)) ~= last.count then
                error("DeviceGraphBlock (file records out): " .. ffi.string(lib.lrhip_strerror()))
            end
            check(lib.lrhip_memcpy_d2h(self.raw, self.raw_dev.ptr, last.count * self.raw_size), "lrhip_memcpy_d2h")
            self.sink:write_raw(self.raw, last.count)
        else
            check(lib.lrhip_synchronize(), "lrhip_synchronize")
        end
        return self.out:resize(0)
    end
    self.out:resize(last.count)
    if last.count > 0 then
        check(lib.lrhip_memcpy_d2h(self.out.data, last.out.ptr, last.count * self.out_size), "lrhip_memcpy_d2h")
    else
        check(lib.lrhip_synchronize(), "lrhip_synchronize")
    end
    return self.out
end

-- A graph fed by its own file sources: one call = one batch.  Every source is asked for what its staging still lacks; the batch is the common count.  Returns
-- the batch's output, an empty vector when nothing could run yet (a rewind: iqfile.lua:86-90 returns an empty vector from that call too), nil at the end
local function process_sources(self)
    if self.finished then return nil end
    for i, src in ipairs(self.sources) do
        local inp = self.input[i]
        if not inp.eof and inp.have < self.batch then
            local got = src:read_raw(ffi.cast("char *", inp.raw_staging) + inp.have * inp.raw_size, self.batch - inp.have)
            if got == nil then inp.eof = true else inp.have = inp.have + got end
        end
    end
    local n = self.batch
    for _, inp in ipairs(self.input) do n = math.min(n, inp.have) end
    if n == 0 then
        for _, inp in ipairs(self.input) do
            if inp.eof and inp.have == 0 then self.finished = true end
        end
        if self.finished then return nil end
        return self.out:resize(0)
    end
    self.fill = n
    local out = run_batch(self)
    for _, inp in ipairs(self.input) do
        local left = inp.have - n
        if left > 0 then
            -- (rare: the sources came up with different counts) the surplus moves to the front through a scratch copy - ffi.copy is memcpy
            local scratch = ffi.new("char[?]", left * inp.raw_size)
            ffi.copy(scratch, ffi.cast("char *", inp.raw_staging) + n * inp.raw_size, left * inp.raw_size)
            ffi.copy(inp.raw_staging, scratch, left * inp.raw_size)
        end
        inp.have = left
    end
    return out
end

-- process(x1, ..., xk): the vectors of one read share their length
function DeviceGraphBlock:process(...)
    if not self.started then start(self) end
    if self.sources then
        local out = process_sources(self)
        if self.sink then return end                   -- a block without ports: run() / run_once() below look at self.finished
        return out
    end
    local vectors = {...}
    local n, done = vectors[1].length, 0
    local out = nil
    while done < n do
        local take = math.min(n - done, self.batch - self.fill)
        for i, x in ipairs(vectors) do
            local inp = self.input[i]
            ffi.copy(ffi.cast("char *", inp.staging) + self.fill * inp.size, ffi.cast("const char *", x.data) + done * inp.size, take * inp.size)
        end
        self.fill = self.fill + take
        done = done + take
        if self.fill == self.batch then
            if out ~= nil and not self.sink then
                -- a second full batch inside one call (a vector longer than a batch): the first batch's output goes to the readers now
                for _, p in ipairs(self.outputs[1].pipes) do p:write(out) end
            end
            out = run_batch(self)
        end
    end
    if self.sink then return end                       -- no output port: nothing returned (radio/core/block.lua:585-593)
    if out == nil then out = self.out:resize(0) end
    return out
end

-- flush(): run what has been accumulated now (a live graph that wants its samples through; cleanup() at EOF) - returns the output vector
function DeviceGraphBlock:flush()
    if not self.started or self.fill == 0 or self.sources then return self.out:resize(0) end
    return run_batch(self)
end

-- EOF upstream (radio/core/block.lua:606): the partial batch, handed to the readers of the output port as process() output would have been
function DeviceGraphBlock:cleanup()
    if not self.started then return end
    local tail = self:flush()
    if tail.length > 0 and not self.sink then
        for _, p in ipairs(self.outputs[1].pipes) do p:write(tail) end
    end
    local lib = lrhip.lib
    if self.sink then
        if self.raw ~= nil then lib.lrhip_host_free(self.raw) end
        if self.raw_dev.ptr ~= nil then lib.lrhip_free(self.raw_dev.ptr) end
        self.raw = nil
        self.sink:cleanup()                                  -- fclose(): the composite no longer runs it as a block
    end
    for _, inp in ipairs(self.input) do
        lib.lrhip_host_free(inp.staging or inp.raw_staging)
        if inp.dev.ptr ~= nil then lib.lrhip_free(inp.dev.ptr) end
        if inp.raw_dev and inp.raw_dev.ptr ~= nil then lib.lrhip_free(inp.raw_dev.ptr) end
    end
    if self.sources then
        for _, src in ipairs(self.sources) do src:cleanup() end          -- fclose(): the composite no longer runs them as blocks
    end
    for _, step in ipairs(self.steps) do
        if step.out.ptr ~= nil then lib.lrhip_free(step.out.ptr) end
        if step.pending then
            for j = 1, 2 do
                if step.pending[j].buf.ptr ~= nil then lib.lrhip_free(step.pending[j].buf.ptr) end
                if step.joined[j].ptr ~= nil then lib.lrhip_free(step.joined[j].ptr) end
            end
        end
    end
    self.started = false
end

-- files -> device -> file (sources AND sink absorbed): a block without ports runs its own loop, as a DeviceChainBlock of that shape does (devicechain.lua)
local block_run, block_run_once = DeviceGraphBlock.run, DeviceGraphBlock.run_once
function DeviceGraphBlock:run()
    if not (self.sources and self.sink) then return block_run(self) end
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
function DeviceGraphBlock:run_once()
    if not (self.sources and self.sink) then return block_run_once(self) end
    self:process()
    if self.finished then return nil end
    return true
end

----------------------------------------------------------------------------------------------------------------------------------
-- collapse(): rewrite the flattened connection table {[InputPort] = OutputPort} (radio/core/composite.lua:343-384)
----------------------------------------------------------------------------------------------------------------------------------

-- a member candidate: a device variant with one or two inputs and one output
local function member_candidate(b)
    if type(b.create_stage) ~= "function" or #b.outputs ~= 1 or #b.inputs < 1 or #b.inputs > 2 then return false end
    if type(b.read_raw) == "function" or type(b.write_raw) == "function" then return false end
    if type(b.device_capable) == "function" and not b:device_capable() then return false end
    return true
end

---
-- Returns the new connection table and the list of DeviceGraphBlocks created (the caller initializes them after the composite's own blocks).
function M.collapse(connections)
    if not M.enabled then return connections, {} end
    -- blocks of the table, edges between candidates
    local neighbours, candidates = {}, {}
    for input, output in pairs(connections) do
        for _, b in ipairs({input.owner, output.owner}) do
            if candidates[b] == nil then candidates[b] = member_candidate(b) end
        end
        local a, b = output.owner, input.owner
        if candidates[a] and candidates[b] then
            neighbours[a] = neighbours[a] or {}
            neighbours[b] = neighbours[b] or {}
            table.insert(neighbours[a], b)
            table.insert(neighbours[b], a)
        end
    end
    -- connected components (in a stable order: by the block names, then as found)
    local seen, components = {}, {}
    for b, ok in pairs(candidates) do
        if ok and not seen[b] and neighbours[b] then
            local comp, queue = {}, {b}
            seen[b] = true
            while #queue > 0 do
                local cur = table.remove(queue)
                comp[#comp + 1] = cur
                for _, nb in ipairs(neighbours[cur] or {}) do
                    if not seen[nb] then
                        seen[nb] = true
                        queue[#queue + 1] = nb
                    end
                end
            end
            components[#components + 1] = comp
        end
    end

    local result, graphs = {}, {}
    for input, output in pairs(connections) do result[input] = output end
    for _, comp in ipairs(components) do
        local inside, has_join = {}, false
        for _, b in ipairs(comp) do
            inside[b] = true
            if #b.inputs == 2 then has_join = true end
        end
        -- the ports that leave the component
        local leaving = {}
        for input, output in pairs(connections) do
            if inside[output.owner] and not inside[input.owner] then leaving[output] = true end
        end
        local nleaving, out_port = 0, nil
        for output, _ in pairs(leaving) do
            nleaving = nleaving + 1
            out_port = output
        end
        if has_join and #comp >= 2 and nleaving == 1 then
            -- topological order of the members (Kahn), graph inputs = the distinct outside ports that feed members
            local indegree, order = {}, {}
            for _, b in ipairs(comp) do
                indegree[b] = 0
                for _, inp in ipairs(b.inputs) do
                    if inside[connections[inp].owner] then indegree[b] = indegree[b] + 1 end
                end
            end
            local ready = {}
            for _, b in ipairs(comp) do
                if indegree[b] == 0 then ready[#ready + 1] = b end
            end
            while #ready > 0 do
                local cur = table.remove(ready, 1)
                order[#order + 1] = cur
                for input, output in pairs(connections) do
                    if output.owner == cur and inside[input.owner] then
                        local r = input.owner
                        indegree[r] = indegree[r] - 1
                        if indegree[r] == 0 then ready[#ready + 1] = r end
                    end
                end
            end
            if #order == #comp then
                local wiring, outside_ports, index_of, input_types = {}, {}, {}, {}
                for _, b in ipairs(order) do
                    wiring[b] = {}
                    for j, inp in ipairs(b.inputs) do
                        local output = connections[inp]
                        if inside[output.owner] then
                            wiring[b][j] = {member = output.owner}
                        else
                            if index_of[output] == nil then
                                outside_ports[#outside_ports + 1] = output
                                index_of[output] = #outside_ports
                                input_types[#outside_ports] = output.data_type
                            end
                            wiring[b][j] = {input = index_of[output]}
                        end
                    end
                end
                -- every graph input a file source with the raw-record hooks, read by members of this subgraph only: absorbed
                local sources = M.absorb_sources and {} or nil
                for i, output in ipairs(outside_ports) do
                    if sources and not DeviceChainBlock.is_raw_source(output.owner) then sources = nil end
                    if sources then
                        for input, out in pairs(connections) do
                            if out == output and not inside[input.owner] then sources = nil end
                        end
                    end
                    if sources then sources[i] = output.owner end
                end
                -- ... and the file sink that is the only reader of the port that leaves
                local sink, nreaders = nil, 0
                for input, out in pairs(connections) do
                    if out == out_port and not inside[input.owner] then
                        nreaders = nreaders + 1
                        sink = input.owner
                    end
                end
                if not (M.absorb_sources and nreaders == 1 and DeviceChainBlock.is_raw_sink(sink)) then sink = nil end
                local graph = DeviceGraphBlock(order, wiring, input_types, out_port.owner, sources, sink)
                graph:differentiate(sources and {} or input_types)
                -- upstream: graph input i reads outside port i; the members' own entries leave the table
                for _, b in ipairs(order) do
                    for j, inp in ipairs(b.inputs) do
                        result[inp] = nil
                        local w = wiring[b][j]
                        if w.member then
                            inp.pipe = pipe.Pipe(w.member.outputs[1], inp)         -- rate-only (radio/core/pipe.lua:36-38)
                        else
                            local i = w.input
                            if sources then
                                local src = sources[i]
                                inp.pipe = {get_rate = function () return src:get_rate() end}
                            else
                                inp.pipe = {get_rate = function () return graph.inputs[i].pipe:get_rate() end}
                            end
                        end
                    end
                end
                if not sources then
                    for i, output in ipairs(outside_ports) do result[graph.inputs[i]] = output end
                end
                -- downstream: whoever read the leaving port reads the graph (an absorbed sink's edge leaves the table; it learns its rate from the member)
                for input, output in pairs(connections) do
                    if output == out_port and not inside[input.owner] then
                        if sink then
                            result[input] = nil
                            input.pipe = pipe.Pipe(out_port, input)
                        else
                            result[input] = graph.outputs[1]
                        end
                    end
                end
                graphs[#graphs + 1] = graph
            end
        end
    end
    return result, graphs
end

M.DeviceGraphBlock = DeviceGraphBlock

return M
