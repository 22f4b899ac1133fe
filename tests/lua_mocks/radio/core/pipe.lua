-- Stand-in for radio/core/pipe.lua: Pipe (rate lookup, write) and a SCRIPTED PipeMux (radio/core/pipe.lua:36-38, :252-262, :417-533) - the test
-- hands the input pipe a list of events {kind = "data" | "stall" | "eof", vec = ...}; ffi.C.poll (tests/helpers/lua_mocks.py) consults the same list.
-- TEST INFRASTRUCTURE for tests/test_lua_exec.py.
local pipe = {}

local Pipe = {}
Pipe.__index = Pipe
function pipe.Pipe(output, input)
    return setmetatable({output = output, input = input, written = {}, script = {}, cursor = 1}, Pipe)
end
function Pipe:get_rate()
    assert(self.output, "Sample rate unavailable for anonymous pipes")
    return self.output.owner:get_rate()
end
function Pipe:initialize() end
function Pipe:write(vec)
    -- keep a copy: the block reuses its output vector
    self.written[#self.written + 1] = __copy_vector(vec)
    return true
end
function Pipe:_read_buffer_count() return 0 end
function Pipe:next_event() return self.script[self.cursor] end

local PipeMux = {}
PipeMux.__index = PipeMux
function pipe.PipeMux(input_pipes, output_pipes, control_socket)
    local self = setmetatable({input_pipes = input_pipes, output_pipes = output_pipes, input_pollfds = {}}, PipeMux)
    __current_mux = self
    return self
end
function PipeMux:read()
    local p = self.input_pipes[1]
    if p == nil then return {}, false, false end
    local ev = p.script[p.cursor]
    if ev == nil or ev.kind == "eof" then return {}, true, false end
    assert(ev.kind == "data", "PipeMux:read() reached a stall event: the run loop should have polled first")
    p.cursor = p.cursor + 1
    return {ev.vec}, false, false
end
function PipeMux:write(data_out)
    for i, pipes in ipairs(self.output_pipes) do
        for _, p in ipairs(pipes) do p:write(data_out[i]) end
    end
    return false, nil, false
end

return pipe
