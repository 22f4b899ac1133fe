---
-- Device fan-out: one output port read by several device chains (radio/core/block.lua:119-166, radio/core/pipe.lua:617-627: the
-- reference writes the same vector to every connected pipe, and every reader is its own process) with the copies made GPU to GPU.
--
--     source -> [head chain] -+-> branch chain 0 -> ...          DeviceFanoutBlock   = source's reader: uploads ONCE (and runs the
--                             +-> branch chain 1 -> ...                                head chain, if the fanned-out port belonged to one),
--                             +-> ...                                                  then pushes every slab to every branch's device
--                                                                DeviceBranchBlock k = one per reader, on device k % lrhip_device_count()
--
-- Without this, `IQFileSource -> 8 x Tuner` costs eight socket writes of the same samples plus eight PCIe uploads; with it the samples
-- cross the host once and reach the branch GPUs over xGMI (hipMemcpyPeerAsync), one link per receiving GPU (~153 GB/s = 19 GS/s of
-- ComplexFloat32).  The branch processes get no data pipe at all: head and branch talk over a private UNIX socket pair that carries
-- 64-byte IPC handles once and 16-byte tokens per slab (include/lrhip.h "fan-out across processes"; tests/test_ipc_gpu.py replays this
-- exact call sequence with one head and three branch processes; luaradio_amd/fanout.py ProcessFanOut is its Python twin).
--
-- Protocol per branch b (slab k, buffer i = k % 2):
--   branch, once : lrhip_malloc x 2, lrhip_ipc_export x 2, lrhip_ipc_event_create x 4  ->  hello{device, capacity, mem[2], filled[2], consumed[2]}
--   head,   once : lrhip_ipc_open x 2, lrhip_ipc_event_open x 4
--   head,   per k: (k >= 2: read ack{k-2}; lrhip_ipc_event_wait(consumed[i], copy stream)); lrhip_peer_copy; lrhip_ipc_event_record(filled[i],
--                  copy stream); write token{k, n}
--   branch, per k: read token; lrhip_ipc_event_wait(filled[i]) on the GPU; lrhip_chain_execute_device; lrhip_ipc_event_record(consumed[i]);
--                  write ack{k}; D2H of the branch output; return it from process() (Block:run writes it to the branch's own readers)
--   EOF          : token{k, -1}; the branch's process() returns nothing = block-generated EOF (radio/core/block.lua:588)
--
-- CompositeBlock:_prepare_to_run calls DeviceFanout.collapse() right after DeviceChainBlock.collapse() (tools/apply_lua_binding.py):
--
--     all_connections, device_chains = require('radio.composites.devicechain').collapse(all_connections)
--     all_connections, device_chains = require('radio.composites.devicefanout').collapse(all_connections, device_chains)
--
-- Multi-process mode only (the branches block on their sockets): top:run(false) keeps the pipes - set DeviceFanout.enabled = false
-- before running single-process, or LUARADIO_HIP_NO_FANOUT=1.
--
-- @module radio.composites.devicefanout

local ffi = require('ffi')

local block = require('radio.core.block')
local platform = require('radio.core.platform')
local lrhip = require('radio.core.lrhip')
local DeviceChainBlock = require('radio.composites.devicechain')

ffi.cdef[[
typedef struct {
    int32_t device;
    int32_t reserved;
    uint64_t capacity;              /* samples a slab holds */
    uint8_t mem[2][64];
    uint8_t filled[2][64];
    uint8_t consumed[2][64];
} lrhip_fanout_hello_t;
typedef struct {
    int64_t k;
    int64_t n;                      /* samples in slab k; < 0: end of stream */
} lrhip_fanout_token_t;
]]

local M = {enabled = not os.getenv("LUARADIO_HIP_NO_FANOUT")}

-- slab size (samples of the fanned-out port) and the latency bound of the head's accumulation, as DeviceChainBlock's
M.slab_samples = 1048576
M.max_latency = 0

-- exact-length socket I/O (tokens are tiny; a short read only happens at EOF)
local function sock_read(fd, buf, size)
    local p, got = ffi.cast("char *", buf), 0
    while got < size do
        local r = tonumber(ffi.C.read(fd, p + got, size - got))
        -- ECONNRESET (104 on Linux): the peer went away with tokens of ours unread in its socket buffer - the same news as EOF, and the caller words it
        if r < 0 and ffi.errno() == 104 then return false end
        if r < 0 then error("read(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
        if r == 0 then return false end
        got = got + r
    end
    return true
end

-- `who`: the peer's name for the message when it has gone away (EPIPE: block processes ignore SIGPIPE, radio/core/composite.lua:577)
local function sock_write(fd, buf, size, who)
    local p, put = ffi.cast("const char *", buf), 0
    while put < size do
        local r = tonumber(ffi.C.write(fd, p + put, size - put))
        if r < 0 then
            if who then error(who .. " terminated unexpectedly (write(): " .. ffi.string(ffi.C.strerror(ffi.errno())) .. ")") end
            error("write(): " .. ffi.string(ffi.C.strerror(ffi.errno())))
        end
        put = put + r
    end
end

local function check(rc, what)
    if rc ~= 0 then error(what .. ": " .. ffi.string(lrhip.lib.lrhip_strerror())) end
end

local function build_chain(self)
    -- nil for an empty member list (a head that only uploads)
    if #self.blocks == 0 then return nil end
    local lib = lrhip.lib
    local stages = ffi.new("lrhip_stage_t *[?]", #self.blocks)
    for i, b in ipairs(self.blocks) do stages[i-1] = b:create_stage() end
    self.stages = stages
    local exact = DeviceChainBlock.exact
    local flags = (exact == true) and lrhip.CHAIN_EXACT or (tonumber(exact) or 0)
    return ffi.gc(lrhip.check_object(lib.lrhip_chain_create_ex(stages, #self.blocks, flags), "Creating lrhip chain object"), lib.lrhip_chain_destroy)
end

----------------------------------------------------------------------------------------------------------------------------------
-- DeviceBranchBlock: a reader of the fanned-out port.  No input pipe: a source as far as the flow graph is concerned.
----------------------------------------------------------------------------------------------------------------------------------
local DeviceBranchBlock = block.factory("DeviceBranchBlock")

function DeviceBranchBlock:instantiate(blocks, index, head)
    self.blocks = assert(blocks, "Missing argument #1 (blocks)")
    self.index = assert(index, "Missing argument #2 (index)")
    self.head = assert(head, "Missing argument #3 (head)")
    self.device = index                 -- placement index, wrapped over the devices of the box by lrhip.ensure()
    self:add_type_signature({}, {block.Output("out", blocks[#blocks]:get_output_type())})
end

function DeviceBranchBlock:get_rate()
    return self.blocks[#self.blocks]:get_rate()
end

function DeviceBranchBlock:initialize()
    self.out = self:get_output_type().vector()
    self.chain = nil
end

local function branch_start(self)
    local lib = lrhip.lib
    local device = lrhip.ensure(self.device)
    self.chain = build_chain(self)
    self.in_size = lib.lrhip_stage_input_size(self.blocks[1]:create_stage())
    self.out_size = lib.lrhip_stage_output_size(self.blocks[#self.blocks]:create_stage())
    self.capacity = self.head.slab_capacity
    self.out_cap = tonumber(lib.lrhip_chain_max_output(self.chain, self.capacity)) + 64
    self.d_out = lrhip.check_object(lib.lrhip_malloc(self.out_cap * self.out_size), "lrhip_malloc")
    local hello = ffi.new("lrhip_fanout_hello_t")
    hello.device = device
    hello.capacity = self.capacity
    self.slab, self.filled, self.consumed = {}, {}, {}
    for i = 0, 1 do
        self.slab[i] = lrhip.check_object(lib.lrhip_malloc(self.capacity * self.in_size), "lrhip_malloc")
        check(lib.lrhip_ipc_export(self.slab[i], hello.mem[i]), "lrhip_ipc_export")
        self.filled[i] = lrhip.check_object(lib.lrhip_ipc_event_create(hello.filled[i]), "lrhip_ipc_event_create")
        self.consumed[i] = lrhip.check_object(lib.lrhip_ipc_event_create(hello.consumed[i]), "lrhip_ipc_event_create")
    end
    sock_write(self.sock, hello, ffi.sizeof(hello))
    self.token = ffi.new("lrhip_fanout_token_t")
end

-- process() of a source block takes no arguments (radio/core/pipe.lua:471-473 _read_none); it blocks until the head announces a slab
function DeviceBranchBlock:process()
    if self.chain == nil then branch_start(self) end
    local lib = lrhip.lib
    local token = self.token
    if not sock_read(self.sock, token, ffi.sizeof(token)) or token.n < 0 then
        return                          -- nothing returned: block-generated EOF (radio/core/block.lua:588)
    end
    local k, n = tonumber(token.k), tonumber(token.n)
    local i = k % 2
    check(lib.lrhip_ipc_event_wait(self.filled[i], 0), "lrhip_ipc_event_wait")            -- the library stream waits ON THE GPU for the head's copy
    local m = tonumber(lib.lrhip_chain_execute_device(self.chain, self.slab[i], n, self.d_out, self.out_cap))
    if m < 0 then error("lrhip_chain_execute_device: " .. ffi.string(lib.lrhip_strerror())) end
    check(lib.lrhip_ipc_event_record(self.consumed[i], 0), "lrhip_ipc_event_record")      -- the slab may be overwritten once the chain has read it
    sock_write(self.sock, token, ffi.sizeof(token))                                        -- ack{k}
    self.out:resize(m)
    if m > 0 then check(lib.lrhip_memcpy_d2h(self.out.data, self.d_out, m * self.out_size), "lrhip_memcpy_d2h") end
    return self.out
end

function DeviceBranchBlock:cleanup()
    if self.sock then ffi.C.close(self.sock) end
    if self.chain ~= nil then
        local lib = lrhip.lib
        check(lib.lrhip_synchronize(), "lrhip_synchronize")
        for i = 0, 1 do
            lib.lrhip_ipc_event_destroy(self.filled[i])
            lib.lrhip_ipc_event_destroy(self.consumed[i])
            lib.lrhip_free(self.slab[i])
        end
        lib.lrhip_free(self.d_out)
        self.chain = nil
    end
end

----------------------------------------------------------------------------------------------------------------------------------
-- DeviceFanoutBlock: the one reader of the source.  No output pipe: a sink as far as the flow graph is concerned.
----------------------------------------------------------------------------------------------------------------------------------
local DeviceFanoutBlock = block.factory("DeviceFanoutBlock")

-- blocks: the head chain's members (may be empty: upload and fan out).  When the fanned-out port belongs to a FILE SOURCE with the raw-record hooks
-- (radio/blocks/sources/file_hip.lua) the source is absorbed as the first member: the head then has no input port at all - it reads the raw records into its
-- pinned staging buffer itself (2 bytes per complex sample cross the host and the link for a 'u8' capture), converts them on the device and fans the
-- ComplexFloat32 slab out: IQFileSource -> 8 x Tuner is one head and eight branches, no sample ever in the interpreter.
function DeviceFanoutBlock:instantiate(blocks, input_type, output_type)
    self.blocks = blocks or {}
    self.branches = {}
    self.output_type = output_type
    self.slab_capacity = M.slab_samples
    self.max_latency = M.max_latency
    self.device = 0
    self.source = (#self.blocks > 0 and DeviceChainBlock.is_raw_source(self.blocks[1])) and self.blocks[1] or nil
    self:add_type_signature(self.source and {} or {block.Input("in", input_type)}, {})
end

-- rate of the fanned-out port (what the branches' first members see upstream)
function DeviceFanoutBlock:get_output_rate()
    if #self.blocks > 0 then return self.blocks[#self.blocks]:get_rate() end
    return self.inputs[1].pipe:get_rate()
end

-- runs in the PARENT, before fork(): the socket pairs have to exist on both sides; block.files keeps them open across the child's
-- close-everything-else loop (radio/core/composite.lua:594-611)
function DeviceFanoutBlock:initialize()
    self.socks = {}
    for k, branch in ipairs(self.branches) do
        local fds = ffi.new("int[2]")
        if ffi.C.socketpair(ffi.C.AF_UNIX, ffi.C.SOCK_STREAM, 0, fds) ~= 0 then
            error("socketpair(): " .. ffi.string(ffi.C.strerror(ffi.errno())))
        end
        self.socks[k] = fds[0]
        self.files[fds[0]] = true
        branch.sock = fds[1]
        branch.files[fds[1]] = true
    end
    self.chain = nil
    self.finished = false
    if self.source then
        for file, _ in pairs(self.source.files or {}) do self.files[file] = true end
    end
end

-- runs in the PARENT after every block has been forked (the hook tools/apply_lua_binding.py puts next to the reference's own "close all pipe inputs and
-- outputs in the top-level process", radio/core/composite.lua:638-642): with the parent's copies gone, a branch that dies closes the LAST descriptor of
-- its end, the head's read() returns 0 ("terminated unexpectedly" below) and the graph tears down instead of hanging; a head that dies ends every
-- branch's read() the same way.
function DeviceFanoutBlock:close_parent_fds()
    for k, fd in ipairs(self.socks or {}) do
        ffi.C.close(fd)
        ffi.C.close(self.branches[k].sock)
    end
end

local function head_start(self)
    local lib = lrhip.lib
    self.my_device = lrhip.ensure(self.device)
    self.chain = build_chain(self)
    local out_type = self.output_type
    self.in_size = self.source and self.source:raw_record_size() or ffi.sizeof(self:get_input_type())
    self.slab_size = ffi.sizeof(out_type)
    -- input samples per slab: what the head chain turns into at most slab_capacity outputs
    self.batch = self.slab_capacity
    if self.chain ~= nil then
        while self.batch > 1 and tonumber(lib.lrhip_chain_max_output(self.chain, self.batch)) > self.slab_capacity do
            self.batch = math.floor(self.batch / 2)
        end
    end
    self.staging = lrhip.check_object(lib.lrhip_host_alloc(self.batch * self.in_size), "lrhip_host_alloc")      -- pinned
    self.d_in = (self.chain ~= nil) and lrhip.check_object(lib.lrhip_malloc(self.batch * self.in_size), "lrhip_malloc") or nil
    self.my_slab, self.ready, self.sent = {}, {}, {}
    local scratch = ffi.new("uint8_t[64]")
    for i = 0, 1 do
        self.my_slab[i] = lrhip.check_object(lib.lrhip_malloc(self.slab_capacity * self.slab_size), "lrhip_malloc")
        self.ready[i] = lrhip.check_object(lib.lrhip_ipc_event_create(scratch), "lrhip_ipc_event_create")    -- head chain done -> copy stream may read
        self.sent[i] = lrhip.check_object(lib.lrhip_ipc_event_create(scratch), "lrhip_ipc_event_create")     -- every branch's copy of this slab issued and done
    end
    -- the branches' hellos (each branch sends its own as soon as its process has a device context)
    self.peer = {}
    local hello = ffi.new("lrhip_fanout_hello_t")
    for b, fd in ipairs(self.socks) do
        if not sock_read(fd, hello, ffi.sizeof(hello)) then error("fan-out branch " .. b .. " closed its socket before the handshake") end
        assert(tonumber(hello.capacity) >= self.slab_capacity, "fan-out branch slab smaller than the head's")
        local peer = {device = hello.device, slab = {}, filled = {}, consumed = {}}
        for i = 0, 1 do
            peer.slab[i] = lrhip.check_object(lib.lrhip_ipc_open(hello.mem[i]), "lrhip_ipc_open")
            peer.filled[i] = lrhip.check_object(lib.lrhip_ipc_event_open(hello.filled[i]), "lrhip_ipc_event_open")
            peer.consumed[i] = lrhip.check_object(lib.lrhip_ipc_event_open(hello.consumed[i]), "lrhip_ipc_event_open")
        end
        self.peer[b] = peer
    end
    self.token = ffi.new("lrhip_fanout_token_t")
    self.fill, self.k = 0, 0
    self.fill_t0 = 0
    self.started = true
end

-- slab k: upload, head chain, one peer copy per branch
local function head_launch(self)
    local lib = lrhip.lib
    local n, k = self.fill, self.k
    local i = k % 2
    self.fill = 0
    if n == 0 then return end
    -- my_slab[i] was last read by the copies of slab k - 2
    if k >= 2 then check(lib.lrhip_ipc_event_synchronize(self.sent[i]), "lrhip_ipc_event_synchronize") end
    local m = n
    if self.chain ~= nil then
        check(lib.lrhip_memcpy_h2d(self.d_in, self.staging, n * self.in_size), "lrhip_memcpy_h2d")
        m = tonumber(lib.lrhip_chain_execute_device(self.chain, self.d_in, n, self.my_slab[i], self.slab_capacity))
        if m < 0 then error("lrhip_chain_execute_device: " .. ffi.string(lib.lrhip_strerror())) end
    else
        check(lib.lrhip_memcpy_h2d(self.my_slab[i], self.staging, n * self.in_size), "lrhip_memcpy_h2d")
    end
    check(lib.lrhip_ipc_event_record(self.ready[i], 0), "lrhip_ipc_event_record")
    check(lib.lrhip_ipc_event_wait(self.ready[i], 1), "lrhip_ipc_event_wait")             -- copy stream: after the head chain's kernels
    local token = self.token
    for b, peer in ipairs(self.peer) do
        if k >= 2 then
            -- the branch has recorded consumed[i] for slab k - 2 before it sent the ack; only then is waiting on the event meaningful
            if not sock_read(self.socks[b], token, ffi.sizeof(token)) then error("fan-out branch " .. b .. " terminated unexpectedly") end
            check(lib.lrhip_ipc_event_wait(peer.consumed[i], 1), "lrhip_ipc_event_wait")
        end
        if m > 0 then
            check(lib.lrhip_peer_copy(peer.slab[i], peer.device, self.my_slab[i], self.my_device, m * self.slab_size), "lrhip_peer_copy")
        end
        check(lib.lrhip_ipc_event_record(peer.filled[i], 1), "lrhip_ipc_event_record")
    end
    check(lib.lrhip_ipc_event_record(self.sent[i], 1), "lrhip_ipc_event_record")
    token.k, token.n = k, m
    for b, fd in ipairs(self.socks) do sock_write(fd, token, ffi.sizeof(token), "fan-out branch " .. b) end
    self.k = k + 1
end

local function now()
    return platform.time_us() / 1e6           -- radio/core/platform.lua:288-294
end

-- a sink's process() returns nothing (#data_out == #self.outputs == 0)
function DeviceFanoutBlock:process(x)
    if not self.started then head_start(self) end
    if self.source then
        -- one slab per call, read by the source's own fread() straight into the pinned staging buffer
        local n = self.source:read_raw(self.staging, self.batch)
        if n == nil then
            self.finished = true
        elseif n > 0 then
            self.fill = n
            head_launch(self)
        end
        return
    end
    local src, left = ffi.cast("const char *", x.data), x.length
    while left > 0 do
        local take = math.min(left, self.batch - self.fill)
        if self.fill == 0 then self.fill_t0 = now() end
        ffi.copy(ffi.cast("char *", self.staging) + self.fill * self.in_size, src, take * self.in_size)
        self.fill = self.fill + take
        src = src + take * self.in_size
        left = left - take
        if self.fill == self.batch then head_launch(self) end
    end
    if self.max_latency > 0 and self.fill > 0 and now() - self.fill_t0 >= self.max_latency then head_launch(self) end
end

-- the wait for input timed out (DeviceChainBlock.timed_run): the partial slab has waited long enough
function DeviceFanoutBlock:poll()
    if self.started and self.fill > 0 then head_launch(self) end
    return nil
end

local block_run = DeviceFanoutBlock.run     -- Block:run, copied into the class by block.factory (radio/core/class.lua:18-40)
function DeviceFanoutBlock:run()
    if self.source then
        -- no ports at all (the branches hang on their sockets): Block:run would wait on the control socket forever (PipeMux:_read_control,
        -- radio/core/pipe.lua:475-493) - process() until the file has ended, a look at the control socket in between, as DeviceChainBlock does
        local pipe = require('radio.core.pipe')
        local pipe_mux = pipe.PipeMux({}, {}, self.control_socket)
        while not self.finished do
            self:process()
            if self.control_socket then
                local ret = ffi.C.poll(pipe_mux.input_pollfds, 1, 0)
                if ret < 0 then error("poll(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
                if ret > 0 then break end
            end
        end
        return self:cleanup()
    end
    if not (self.max_latency > 0) then return block_run(self) end
    return DeviceChainBlock.timed_run(self, function (b)
        if not b.started or b.fill == 0 then return -1 end
        return math.max(0, b.fill_t0 + b.max_latency - now())
    end)
end

-- EOF upstream: the partial slab, then the end-of-stream token; the last two acks are drained so no branch writes into a closed socket
function DeviceFanoutBlock:cleanup()
    if not self.started then
        if #self.branches == 0 then return end
        head_start(self)                -- a stream that ended before its first sample still has to release the branches
    end
    local lib = lrhip.lib
    head_launch(self)
    local token = self.token
    for back = math.min(2, self.k), 1, -1 do
        for b, fd in ipairs(self.socks) do sock_read(fd, token, ffi.sizeof(token)) end
    end
    check(lib.lrhip_copy_stream_synchronize(), "lrhip_copy_stream_synchronize")
    token.k, token.n = self.k, -1
    for _, fd in ipairs(self.socks) do
        sock_write(fd, token, ffi.sizeof(token))
        ffi.C.close(fd)
    end
    -- the mappings of the branches' slabs and events, and this block's own
    for _, peer in ipairs(self.peer) do
        for i = 0, 1 do
            lib.lrhip_ipc_event_destroy(peer.filled[i])
            lib.lrhip_ipc_event_destroy(peer.consumed[i])
            check(lib.lrhip_ipc_close(peer.slab[i]), "lrhip_ipc_close")
        end
    end
    for i = 0, 1 do
        lib.lrhip_ipc_event_destroy(self.ready[i])
        lib.lrhip_ipc_event_destroy(self.sent[i])
        lib.lrhip_free(self.my_slab[i])
    end
    lib.lrhip_host_free(self.staging)
    if self.d_in ~= nil then lib.lrhip_free(self.d_in) end
    self.started = false
    if self.source then self.source:cleanup() end          -- fclose (iqfile.lua:118-124): the absorbed source is no longer in the evaluation order
end

----------------------------------------------------------------------------------------------------------------------------------
-- collapse(): rewrite the connection table after DeviceChainBlock.collapse()
----------------------------------------------------------------------------------------------------------------------------------
local function chainable(b)
    if type(b.create_stage) ~= "function" or #b.inputs ~= 1 or #b.outputs ~= 1 then return false end
    if type(b.device_capable) == "function" and not b:device_capable() then return false end
    return true
end

-- the device blocks a reader stands for: a DeviceChainBlock's members, or a single chainable block
local function members_of(b, is_chain)
    if is_chain[b] then
        if b.source or b.sink then return nil end       -- a chain that reads or writes a file itself has no port on that side
        return b.blocks
    end
    if chainable(b) then return {b} end
    return nil
end

---
-- For every output port with two or more readers that are ALL device chains (or single device blocks): one DeviceFanoutBlock in front
-- (absorbing the writer when the writer is itself a device chain) and one DeviceBranchBlock per reader.  Returns the new connection
-- table and the new list of device blocks whose initialize() the caller runs after the composite's own (head blocks create the socket
-- pairs there, pre-fork).
function M.collapse(connections, chains)
    if not M.enabled then return connections, chains end
    local is_chain = {}
    for _, c in ipairs(chains) do is_chain[c] = true end
    local readers = {}
    for input, output in pairs(connections) do
        readers[output] = readers[output] or {}
        table.insert(readers[output], input)
    end

    -- An output port qualifies when it has two or more readers and ALL of them are device chains (or single device blocks) ...
    local function all_device(output)
        local inputs = readers[output]
        if inputs == nil or #inputs < 2 then return false end
        for _, input in ipairs(inputs) do
            if members_of(input.owner, is_chain) == nil then return false end
        end
        return true
    end
    -- ... and its writer is not itself a reader of a port that is rewritten: in source -> {A -> {C, D}, B} the port of the source is rewritten (A and B
    -- become branches), so A's own port keeps its pipes - a block plays ONE role, and C, D stay ordinary chains reading branch A's output.
    -- Decided from the sources downwards, so a third level (C -> {E, F}) qualifies again.
    local accepted = {}
    local function accept(output)
        if accepted[output] == nil then
            accepted[output] = false                  -- (a flow graph has no cycles; this only guards the recursion)
            if all_device(output) then
                local writer = output.owner
                local upstream = (#writer.inputs == 1) and connections[writer.inputs[1]] or nil
                accepted[output] = not (upstream ~= nil and accept(upstream))
            end
        end
        return accepted[output]
    end

    local result, dropped, created = {}, {}, {}
    for input, output in pairs(connections) do result[input] = output end
    for output, inputs in pairs(readers) do
        if accept(output) then
            local writer = output.owner
            local head
            if is_chain[writer] and writer.source then
                -- the writer is a device chain that reads a file itself: it becomes the head chain, source included
                head = DeviceFanoutBlock(writer.blocks, nil, output.data_type)
                head:differentiate({})
                dropped[writer] = true
            elseif is_chain[writer] then
                -- the writer is a device chain: it becomes the head chain and keeps its output on its device
                head = DeviceFanoutBlock(writer.blocks, writer:get_input_type(), output.data_type)
                head:differentiate({writer:get_input_type()})
                result[head.inputs[1]] = connections[writer.inputs[1]]
                result[writer.inputs[1]] = nil
                writer.blocks[1].inputs[1].pipe = {get_rate = function () return head.inputs[1].pipe:get_rate() end}
                dropped[writer] = true
            elseif DeviceChainBlock.is_raw_source(writer) then
                -- the writer is a file source: the head reads the raw records itself and converts them on its device
                head = DeviceFanoutBlock({writer}, nil, output.data_type)
                head:differentiate({})
            else
                head = DeviceFanoutBlock({}, output.data_type, output.data_type)
                head:differentiate({output.data_type})
                result[head.inputs[1]] = output
            end
            for k, input in ipairs(inputs) do
                local reader = input.owner
                local members = members_of(reader, is_chain)
                local branch = DeviceBranchBlock(members, k - 1, head)
                branch:differentiate({})
                head.branches[k] = branch
                result[input] = nil
                -- whoever read the reader now reads the branch
                for downstream, out in pairs(connections) do
                    if out == reader.outputs[1] then result[downstream] = branch.outputs[1] end
                end
                -- the first member's upstream rate is the fanned-out port's
                members[1].inputs[1].pipe = {get_rate = function () return head:get_output_rate() end}
                if is_chain[reader] then dropped[reader] = true end
                created[#created + 1] = branch
            end
            created[#created + 1] = head
        end
    end

    local out_chains = {}
    for _, c in ipairs(chains) do
        if not dropped[c] then out_chains[#out_chains + 1] = c end
    end
    for _, c in ipairs(created) do out_chains[#out_chains + 1] = c end
    return result, out_chains
end

M.DeviceFanoutBlock = DeviceFanoutBlock
M.DeviceBranchBlock = DeviceBranchBlock

return M
