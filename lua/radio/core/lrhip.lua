---
-- LuaJIT FFI binding of liblrhip.so (include/lrhip.h), the MI355X DSP block engine.
--
-- Drop-in for a LuaRadio checkout: copy this file to radio/core/lrhip.lua. It registers the library the
-- same way radio/core/platform.lua:277-299 registers VOLK / liquid-dsp / FFTW3f: `platform.libs.hip` and
-- `platform.features.hip`, with the `LUARADIO_DISABLE_HIP` escape hatch mirroring platform.lua:328-330.
--
-- NOTE: LuaJIT is not installed in the build image.  The Lua files are held to the C ABI and to each other by tests/test_lua_glue.py (a tokenizer
-- model: every lib.lrhip_* used is declared in the cdef below with the prototype of include/lrhip.h, every method called on a block is defined by a
-- device variant or by the reference's Block class, every local function used is defined, the binding survives the reference's module load order) and -
-- since round 4 - they are EXECUTED by tests/test_lua_exec.py under a small Lua 5.1 interpreter (tests/helpers/minilua.py) with an ffi look-alike on
-- ctypes and stand-ins for radio.core.block / pipe / platform: collapse(), the fan-out rewrite, process / cleanup / poll / run and the head / branch
-- socket protocol run as written, against a recording fake of the library on a CPU box and against the real liblrhip.so on the GPU box.
--
-- @module radio.core.lrhip

local ffi = require('ffi')

local platform = require('radio.core.platform')

-- Every entry point declared here is called by a file under lua/radio/ and executed by a test (tests/test_lua_glue.py holds the list to the header:
-- same prototypes; NOT declared are the ones only a measurement harness or a PyTorch host needs - lrhip_set_stream, lrhip_timer_*, lrhip_ipc_event_query - and lrhip_chain_create, which is lrhip_chain_create_ex with flags 0).
ffi.cdef[[
typedef struct lrhip_stage lrhip_stage_t;
typedef struct lrhip_chain lrhip_chain_t;
typedef struct lrhip_ipc_event lrhip_ipc_event_t;

int lrhip_init(int device);
const char *lrhip_strerror(void);
int lrhip_device_count(void);
int lrhip_device(void);
int lrhip_synchronize(void);
const char *lrhip_version(void);

lrhip_stage_t *lrhip_fir_create(const float *taps, unsigned ntaps, int taps_complex, int input_complex, unsigned decim, int use_fft);
lrhip_stage_t *lrhip_rotator_create(double omega);
lrhip_stage_t *lrhip_downsampler_create(unsigned factor, int elem_size);
lrhip_stage_t *lrhip_fmdiscrim_create(double gain);
lrhip_stage_t *lrhip_iir_create(const float *b, unsigned nb, const float *a, unsigned na, int input_complex);
lrhip_stage_t *lrhip_psd_create(unsigned n, const float *window, double scale, int logarithmic, int input_complex, int fftshift);
lrhip_stage_t *lrhip_dft_create(unsigned n, int inverse, int real_side);
lrhip_stage_t *lrhip_format_convert_create(const char *format, int complex_out);
lrhip_stage_t *lrhip_format_pack_create(const char *format, int complex_in);
lrhip_stage_t *lrhip_binary_create(const char *op, int input_complex);
lrhip_stage_t *lrhip_multiply_constant_create(float re, float im, int constant_complex, int input_complex);
lrhip_stage_t *lrhip_upsampler_create(unsigned factor, int elem_size);
lrhip_stage_t *lrhip_channelizer_create(const float *taps, unsigned ntaps, unsigned nchannels);
lrhip_stage_t *lrhip_welch_create(unsigned n, const float *window, double scale, int logarithmic, int input_complex, unsigned overlap);
long lrhip_welch_read(lrhip_stage_t *q, float *avg_host, int reset);
lrhip_stage_t *lrhip_fmmod_create(double modulation_index);
lrhip_stage_t *lrhip_powersquelch_create(double alpha, double threshold, int input_complex);
lrhip_stage_t *lrhip_agc_create(double power_alpha, double gain_alpha, double target, double threshold, int input_complex);
lrhip_stage_t *lrhip_unary_create(const char *op, float re, float im, int constant_complex, int input_complex);
lrhip_stage_t *lrhip_delay_create(unsigned num_samples, int elem_size);
lrhip_stage_t *lrhip_hilbert_create(const float *taps, unsigned ntaps);
void lrhip_stage_destroy(lrhip_stage_t *q);
int lrhip_stage_reset(lrhip_stage_t *q);
int lrhip_stage_input_size(const lrhip_stage_t *q);
int lrhip_stage_output_size(const lrhip_stage_t *q);
unsigned long lrhip_stage_max_output(const lrhip_stage_t *q, unsigned long n_in);
long lrhip_stage_execute(lrhip_stage_t *q, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity);
long lrhip_stage_execute_device(lrhip_stage_t *q, const void *in_dev, unsigned long n_in, void *out_dev, unsigned long out_capacity);
long lrhip_stage_execute2(lrhip_stage_t *q, const void *in1_host, const void *in2_host, unsigned long n_in, void *out_host, unsigned long out_capacity);
long lrhip_stage_execute2_device(lrhip_stage_t *q, const void *in1_dev, const void *in2_dev, unsigned long n_in, void *out_dev, unsigned long out_capacity);

lrhip_chain_t *lrhip_chain_create_ex(lrhip_stage_t **stages, unsigned nstages, unsigned flags);
void lrhip_chain_destroy(lrhip_chain_t *c);
int lrhip_chain_reset(lrhip_chain_t *c);
unsigned long lrhip_chain_max_output(const lrhip_chain_t *c, unsigned long n_in);
long lrhip_chain_execute(lrhip_chain_t *c, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity);
long lrhip_chain_execute_device(lrhip_chain_t *c, const void *in_dev, unsigned long n_in, void *out_dev, unsigned long out_capacity);
int lrhip_chain_last_launches(const lrhip_chain_t *c);
int lrhip_chain_set_ring(lrhip_chain_t *c, unsigned depth, unsigned long max_chunk);
long lrhip_chain_submit(lrhip_chain_t *c, const void *in_host, unsigned long n_in);
void *lrhip_chain_ring_input(lrhip_chain_t *c);
long lrhip_chain_submit_fd(lrhip_chain_t *c, int fd, unsigned long long offset, unsigned long max_in);
long lrhip_chain_collect(lrhip_chain_t *c, void *out_host, unsigned long out_capacity);
int lrhip_chain_in_flight(const lrhip_chain_t *c);
long lrhip_chain_push(lrhip_chain_t *c, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity);
long lrhip_chain_flush(lrhip_chain_t *c, void *out_host, unsigned long out_capacity);
unsigned long lrhip_chain_push_bound(const lrhip_chain_t *c, unsigned long n_in);
int lrhip_chain_set_latency(lrhip_chain_t *c, double max_seconds);
long lrhip_chain_poll(lrhip_chain_t *c, void *out_host, unsigned long out_capacity);
double lrhip_chain_poll_due(const lrhip_chain_t *c);

void *lrhip_malloc(unsigned long bytes);
void lrhip_free(void *dev_ptr);
int lrhip_memcpy_h2d(void *dev_dst, const void *host_src, unsigned long bytes);
int lrhip_memcpy_d2h(void *host_dst, const void *dev_src, unsigned long bytes);
int lrhip_memcpy_d2d(void *dev_dst, const void *dev_src, unsigned long bytes);
void *lrhip_host_alloc(unsigned long bytes);
void lrhip_host_free(void *host_ptr);
int lrhip_host_register(void *host_ptr, unsigned long bytes);
int lrhip_host_unregister(void *host_ptr);

int lrhip_stage_seek(lrhip_stage_t *q, unsigned long long n0);
int lrhip_chain_seek(lrhip_chain_t *c, unsigned long long n0);
long lrhip_chain_halo(const lrhip_chain_t *c);
unsigned long lrhip_chain_shard_align(const lrhip_chain_t *c);
int lrhip_chain_start_at(lrhip_chain_t *c, unsigned long long first_sample, unsigned long long *seek_sample);
int lrhip_ipc_export(const void *dev_ptr, void *handle_out);
void *lrhip_ipc_open(const void *handle);
int lrhip_ipc_close(void *dev_ptr);
lrhip_ipc_event_t *lrhip_ipc_event_create(void *handle_out);
lrhip_ipc_event_t *lrhip_ipc_event_open(const void *handle);
void lrhip_ipc_event_destroy(lrhip_ipc_event_t *e);
int lrhip_ipc_event_record(lrhip_ipc_event_t *e, int on_copy_stream);
int lrhip_ipc_event_wait(lrhip_ipc_event_t *e, int on_copy_stream);
int lrhip_ipc_event_synchronize(lrhip_ipc_event_t *e);
int lrhip_peer_copy(void *dst, int dst_device, const void *src, int src_device, unsigned long bytes);
int lrhip_copy_stream_synchronize(void);
]]

local M = {available = false}

-- lrhip_chain_create_ex flags (include/lrhip.h): the numerical contract of a collapsed run of blocks
M.CHAIN_EXACT_ROTATOR = 1
M.CHAIN_NO_POLYPHASE_TAIL = 2
M.CHAIN_NO_FUSION = 4
M.CHAIN_NO_SINGLE_LAUNCH = 8
M.CHAIN_EXACT = 11

if not os.getenv("LUARADIO_DISABLE_HIP") then
    local ok, lib = pcall(ffi.load, "lrhip")
    if not ok then ok, lib = pcall(ffi.load, "liblrhip.so") end
    if ok then
        M.lib = lib
        M.available = true
        M.version = ffi.string(lib.lrhip_version())
        platform.libs.hip = lib
        platform.features.hip = true
    end
end

---
-- Raise a Lua error carrying the library's message, the way the reference raises
-- `error("Creating liquid firfilt object.")` (radio/blocks/signal/firfilter.lua:199-201).
function M.check_object(obj, what)
    if obj == nil then
        error(what .. ": " .. ffi.string(M.lib.lrhip_strerror()))
    end
    return obj
end

---
-- The device context must be created in the block's own process: CompositeBlock calls initialize() in the
-- parent before fork() (radio/core/composite.lua:443 vs :569), so device blocks create their stage lazily on
-- the first process() call.  ensure() is what they call first.
--
-- `device` is a PLACEMENT INDEX, not necessarily a device number: it wraps over the devices of the box
-- (`index % lrhip_device_count()`), so a flow graph that fans out into 8 branches runs one branch per GPU on an
-- 8-GPU node and all of them on the one GPU of a workstation, unchanged.  nil = the library's default device
-- (or LUARADIO_HIP_DEVICE).  A process is bound to ONE device by its first ensure(); later calls with another
-- index keep that device (top:run(false) puts every block into one process) and return it.
local initialized_pid, initialized_device = nil, nil
function M.ensure(device)
    local pid = ffi.C.getpid()
    if initialized_pid ~= pid then
        local want = device or tonumber(os.getenv("LUARADIO_HIP_DEVICE"))
        if want ~= nil then
            local count = M.lib.lrhip_device_count()
            if count < 1 then
                error("lrhip_device_count: " .. ffi.string(M.lib.lrhip_strerror()))
            end
            want = want % count
        end
        if M.lib.lrhip_init(want or -1) ~= 0 then
            error("lrhip_init: " .. ffi.string(M.lib.lrhip_strerror()))
        end
        initialized_pid = pid
        initialized_device = M.lib.lrhip_device()
    end
    return initialized_device
end

---
-- Answer a question that needs device objects (a chain's halo, its partition grid, where a partition's source must start) from a process that must
-- NOT touch the device: the flow graph's parent.  CompositeBlock forks one process per block after initialize() (radio/core/composite.lua:443 vs :569),
-- and a child forked after its parent created a device context cannot use the device (the library refuses with a message: the HIP runtime's threads and
-- queues exist only in the parent).  So the question is asked in a short-lived fork()ed helper: it binds to the device, builds the objects, writes up to
-- three numbers to a pipe and _exit()s - no ffi.gc finalisers, no atexit handlers; the parent stays clean and the block processes forked later create
-- their own objects as always.  A process that already owns a device context (top:run(false): every block in one process; or a block's own process) just
-- calls fn().  fn returns up to three numbers; an error in the helper is re-raised here with its message.
ffi.cdef[[
int pipe(int pipefd[2]);
void _exit(int status);
]]
function M.in_helper(fn)
    if M.lib.lrhip_device() >= 0 then return fn() end
    local fds = ffi.new("int[2]")
    if ffi.C.pipe(fds) ~= 0 then error("pipe(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
    local pid = ffi.C.fork()
    if pid < 0 then error("fork(): " .. ffi.string(ffi.C.strerror(ffi.errno()))) end
    local head = ffi.new("double[4]")
    if pid == 0 then
        ffi.C.close(fds[0])
        local ok, a, b, c = pcall(fn)
        head[0] = ok and 1 or 0
        if ok then head[1], head[2], head[3] = tonumber(a) or 0, tonumber(b) or 0, tonumber(c) or 0 end
        ffi.C.write(fds[1], head, 32)
        if not ok then
            local msg = tostring(a)
            ffi.C.write(fds[1], msg, #msg)
        end
        ffi.C._exit(ok and 0 or 1)
    end
    ffi.C.close(fds[1])
    local buf, got = ffi.new("uint8_t[?]", 32 + 1024), 0
    while got < 32 + 1024 do
        local r = tonumber(ffi.C.read(fds[0], buf + got, 32 + 1024 - got))
        if r <= 0 then break end
        got = got + r
    end
    ffi.C.close(fds[0])
    local status = ffi.new("int[1]")
    ffi.C.waitpid(pid, status, 0)
    if got < 32 then error("lrhip helper process ended without an answer") end
    ffi.copy(head, buf, 32)
    if head[0] ~= 1 then error(ffi.string(buf + 32, got - 32), 0) end
    return head[1], head[2], head[3]
end

---
-- Make `Block` a device block: `create(self)` returns a fresh lrhip_stage_t* built from the block's host-side
-- parameters.  The stage is created lazily, in the process that runs the block (see ensure()), and exposed as
-- Block:create_stage() - which is also what DeviceChainBlock collects to build one lrhip_chain_t for a run of blocks.
function M.device_block(Block, create)
    function Block:create_stage()
        if self.stage == nil then
            M.ensure()
            self.stage = ffi.gc(M.check_object(create(self), "Creating lrhip " .. self.name .. " object"),
                                M.lib.lrhip_stage_destroy)
        end
        return self.stage
    end
    -- back to the just-created state (zero history, phase 0, index 0): a flow graph that is run a second time in the same process
    function Block:reset_stage()
        if self.stage ~= nil and M.lib.lrhip_stage_reset(self.stage) ~= 0 then
            error("lrhip_stage_reset: " .. ffi.string(M.lib.lrhip_strerror()))
        end
    end
    -- a stand-alone block that starts in the middle of a recording (include/lrhip.h "time-axis sharding"): forget every carried sample and set the
    -- absolute counters (rotator phase, downsampler index) as if n0 input samples had been consumed
    function Block:seek_stage(n0)
        if M.lib.lrhip_stage_seek(self:create_stage(), n0) ~= 0 then
            error("lrhip_stage_seek: " .. ffi.string(M.lib.lrhip_strerror()))
        end
    end
end

---
-- The ONE line a block file of the checkout gains, directly above its final `return <Block>`:
--
--     require('radio.core.lrhip').patch('firfilter', FIRFilterBlock)
--
-- At that point every top-level statement of the reference file has run - including ladders that (re)assign methods
-- late, e.g. radio/blocks/signal/firfilter.lua:400-402 / :488-490, which set process_fft_* AFTER the dot-product
-- ladder of :88-307 - so nothing can overwrite what the patch installs, and derived classes copy the patched methods
-- when block.factory(name, parent) runs (radio/core/class.lua:18-40 copies the parent's functions at factory time;
-- the derived file require()s the parent file first).  Without the library (or with LUARADIO_DISABLE_HIP) it is a no-op.
-- `name` is the reference file's base name; tools/apply_lua_binding.py inserts the lines, tests/test_lua_glue.py
-- models the load order.
local unary_ops = {complexmagnitude = true, complexphase = true, complextoreal = true, complextoimag = true,
                   complexconjugate = true, realtocomplex = true, absolutevalue = true}
local binary_ops = {multiply = true, multiplyconjugate = true, add = true, subtract = true, floattocomplex = true}
local file_sources = {iqfilesource = true, realfilesource = true}
local file_sinks = {iqfilesink = true, realfilesink = true, gnuplotspectrum = true}
function M.patch(name, Block)
    if not M.available then return Block end
    if name == "firfilter" then
        require('radio.blocks.signal.firfilter_hip')(Block)
        return Block
    end
    -- radio/blocks/sources/{iqfile,realfile}.lua, radio/blocks/sinks/{iqfile,realfile,gnuplotspectrum}.lua: two directories hold an iqfile.lua, so the
    -- patch names carry the role (tools/apply_lua_binding.py knows which file gets which name)
    if file_sources[name] then
        require('radio.blocks.sources.file_hip')["patch_" .. name](Block)
        return Block
    end
    if file_sinks[name] then
        require('radio.blocks.sinks.file_hip')["patch_" .. name](Block)
        return Block
    end
    local elementwise = require('radio.blocks.signal.elementwise_hip')
    if unary_ops[name] then
        elementwise.patch_unary(Block, name)
    elseif binary_ops[name] then
        elementwise.patch_binary(Block, name)
    else
        local patch = elementwise["patch_" .. name]
        if patch == nil then error("radio.core.lrhip: no device variant for " .. name) end
        patch(Block)
    end
    return Block
end

---
-- radio/utilities/spectrum_utils.lua returns a table of classes, not a block: the ONE line it gains, directly above its final `return {DFT = DFT, ...}`, is
--
--     require('radio.core.lrhip').patch_spectrum(DFT, IDFT, PSD)
--
-- (radio/utilities/spectrum_utils_hip.lua: initialize() / compute() of the three classes on lrhip_dft_create / lrhip_psd_create).
function M.patch_spectrum(DFT, IDFT, PSD)
    if not M.available then return end
    require('radio.utilities.spectrum_utils_hip')(DFT, IDFT, PSD)
end

---
-- FIRFilterBlock's use_fft argument -> lrhip_fir_create's mode (the same table as luaradio_amd/block.py fir_mode).
-- nil (the caller did not choose; the reference then picks FFT when FFTW is present, firfilter.lua:57) = 3, automatic:
-- overlap-save arithmetic with one output per input from 48 taps up, direct form below - and 0 under tests.jigs, as there.  true = 1, the reference's overlap-save INCLUDING its block-emission
-- framing (firfilter.lua:361-398).  false = 0, direct form (bit-identical to the fmaf chain in tap order).
-- "fast" = 2 and "auto" = 3 select the sample-exact overlap-save arithmetic explicitly.
function M.fir_mode(use_fft)
    if use_fft == nil then
        -- firfilter.lua:57: under the reference's unit-test jig an unspecified use_fft means the direct form
        if package.loaded['tests.jigs'] then return 0 end
        return 3
    end
    if use_fft == "auto" then return 3 end
    if use_fft == "fast" then return 2 end
    return use_fft and 1 or 0
end

---
-- Zero-copy bookkeeping for the vectors a block owns: `registered[owner]` = the buffer pinned for it.  A Vector that outgrew its buffer got a new one
-- (radio/core/vector.lua:108-136): the old range is unregistered first, before the collector frees it.  `M.zero_copy = false` keeps the staging copies.
M.zero_copy = true
local registered = setmetatable({}, {__mode = "k"})
function M.pin(owner, data, size)
    if not M.zero_copy or size == 0 then return end
    local p = ffi.cast("void *", data)
    local r = registered[owner]
    if r ~= nil and r.ptr == p and r.size >= size then return end
    if r ~= nil then M.lib.lrhip_host_unregister(r.ptr) end
    if M.lib.lrhip_host_register(p, size) == 0 then
        registered[owner] = {ptr = p, size = size}
    else
        registered[owner] = nil        -- not fatal: the library stages the copy as before
    end
end

---
-- The input side of zero-copy: a block's input vector is a cast into its pipe's read buffer (radio/core/vector.lua:48-65), which is page-aligned, 1 MiB
-- and lives as long as the pipe (radio/core/pipe.lua:72-76) - registered once per process, every vector read from the pipe is DMA'd from where read(2)
-- put it.  Blocks fed by something else than a reference Pipe (the test jig's vectors) have no such buffer and are staged as before.
function M.pin_inputs(b)
    if not M.zero_copy then return end
    for _, input in ipairs(b.inputs) do
        local p = input.pipe
        if p ~= nil and p._rbuf ~= nil then M.pin(p, p._rbuf, p._rbuf_capacity) end
    end
end

---
-- One process() call: resize the output vector to the bound, execute, trim (firfilter.lua:130 pattern).
-- The output vector's buffer (page-aligned, radio/core/vector.lua:19-37) is registered with the library once per (re)allocation, so the result is
-- written into it by DMA; the input is a cast into the pipe's read buffer, which DeviceChainBlock / the block's run loop may register as a whole.
function M.execute(stage, x, out, owner)
    local lib = M.lib
    if owner ~= nil then M.pin_inputs(owner) end
    local cap = tonumber(lib.lrhip_stage_max_output(stage, x.length))
    out:resize(cap)
    M.pin(out, out.data, out._capacity * ffi.sizeof(out.data_type))      -- capacity in elements (radio/core/vector.lua:27,132)
    local n = tonumber(lib.lrhip_stage_execute(stage, x.data, x.length, out.data, cap))
    if n < 0 then
        error("lrhip_stage_execute: " .. ffi.string(lib.lrhip_strerror()))
    end
    return out:resize(n)
end

---
-- Two-input variant (multiply.lua:43-57 pattern: both inputs have the same length).
function M.execute2(stage, x, y, out, owner)
    local lib = M.lib
    if owner ~= nil then M.pin_inputs(owner) end
    out:resize(x.length)
    M.pin(out, out.data, out._capacity * ffi.sizeof(out.data_type))
    local n = tonumber(lib.lrhip_stage_execute2(stage, x.data, y.data, x.length, out.data, x.length))
    if n < 0 then
        error("lrhip_stage_execute2: " .. ffi.string(lib.lrhip_strerror()))
    end
    return out:resize(n)
end

return M
