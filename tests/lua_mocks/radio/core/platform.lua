-- stand-in for radio/core/platform.lua as far as the glue touches it (tests/helpers/lua_mocks.py): TEST INFRASTRUCTURE
local platform = {libs = {}, features = {}, os = "Linux"}
platform.time_us = function () return __now_us() end
return platform
