-- Stand-in for radio/core/block.lua: the Block API surface lua/radio/** relies on, re-stated from the reference's documentation of it
-- (docs/0.reference-manual.md "Block", docs/3.creating-blocks.md; radio/core/block.lua:119-166 ports, :238-352 differentiate, :556-608 run,
-- :610-660 factory).  TEST INFRASTRUCTURE for tests/test_lua_exec.py; not product, not a copy.
local pipe = require('radio.core.pipe')

local block = {}

function block.Input(name, data_type) return {name = name, data_type = data_type, is_input = true} end
function block.Output(name, data_type) return {name = name, data_type = data_type, is_input = false} end

local InputPort, OutputPort = {}, {}
InputPort.__index = InputPort
OutputPort.__index = OutputPort
function InputPort.new(owner, name) return setmetatable({owner = owner, name = name, data_type = nil, pipe = nil}, InputPort) end
function OutputPort.new(owner, name) return setmetatable({owner = owner, name = name, data_type = nil, pipes = {}}, OutputPort) end
function InputPort:filenos() return {} end
function OutputPort:filenos() return {} end
function InputPort:close() end
block.InputPort, block.OutputPort = InputPort, OutputPort

local Block = {}

function Block:add_type_signature(inputs, outputs, process_func, initialize_func)
    if self.inputs == nil then
        self.inputs, self.outputs = {}, {}
        for i, d in ipairs(inputs) do self.inputs[i] = InputPort.new(self, d.name) end
        for i, d in ipairs(outputs) do self.outputs[i] = OutputPort.new(self, d.name) end
    else
        assert(#inputs == #self.inputs and #outputs == #self.outputs, "Invalid type signature, input or output count mismatch.")
    end
    self.signatures[#self.signatures + 1] = {inputs = inputs, outputs = outputs, process_func = process_func or self.process,
                                             initialize_func = initialize_func or self.initialize}
end

function Block:differentiate(input_types)
    for _, sig in ipairs(self.signatures) do
        local ok = (#sig.inputs == #input_types)
        if ok then
            for i, d in ipairs(sig.inputs) do
                if d.data_type ~= input_types[i] then ok = false end
            end
        end
        if ok then
            self.signature = sig
            for i, d in ipairs(sig.inputs) do self.inputs[i].data_type = d.data_type end
            for i, d in ipairs(sig.outputs) do self.outputs[i].data_type = d.data_type end
            self.process, self.initialize = sig.process_func, sig.initialize_func
            return
        end
    end
    error("No compatible type signatures found for block " .. self.name)
end

function Block:get_input_type(index) return self.signature.inputs[index or 1].data_type end
function Block:get_output_type(index) return self.signature.outputs[index or 1].data_type end
function Block:get_rate() return self.inputs[1].pipe:get_rate() end
function Block:initialize() end
function Block:cleanup() end
function Block:__tostring() return self.name end

-- radio/core/block.lua:556-608
function Block:run()
    local input_pipes, output_pipes = {}, {}
    for i = 1, #self.inputs do input_pipes[i] = self.inputs[i].pipe end
    for i = 1, #self.outputs do
        output_pipes[i] = {}
        for j = 1, #self.outputs[i].pipes do output_pipes[i][j] = self.outputs[i].pipes[j] end
    end
    local pipe_mux = pipe.PipeMux(input_pipes, output_pipes, self.control_socket)
    while true do
        local data_in, eof, shutdown = pipe_mux:read()
        if eof or shutdown then break end
        local data_out = {self:process(unpack(data_in))}
        if #data_out ~= #self.outputs then break end
        local weof, eof_pipe, wshutdown = pipe_mux:write(data_out)
        if wshutdown or weof then break end
    end
    self:cleanup()
end

function block.factory(name, parent)
    local class = {}
    for k, v in pairs(parent or Block) do class[k] = v end      -- the parent's functions are COPIED at factory time (radio/core/class.lua:18-40)
    class.__index = class
    class.name = name
    setmetatable(class, {__call = function (cls, ...)
        local self = setmetatable({}, cls)
        self.inputs, self.outputs = nil, nil
        self.files = {}
        self.signatures, self.signature = {}, nil
        self:instantiate(...)
        return self
    end})
    return class
end

block.Block = Block
return block
