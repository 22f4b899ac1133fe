"""The PATCHED REFERENCE FILES THEMSELVES, executed (CPU box only: /root/reference does not exist on the GPU box, and nothing here needs a GPU).

tests/test_lua_blocks.py runs the device variants against stand-ins of the reference's block files.  Here the real files are loaded: every file
tools/apply_lua_binding.py touches is taken from /root/reference WITH its patch line applied (in memory, nothing is written), the reference's own pure-Lua
modules (radio/core/class.lua, radio/core/util.lua, radio/utilities/filter_utils.lua, window_utils.lua, the derived block files lowpassfilter.lua,
singlepolelowpassfilter.lua, fmdeemphasisfilter.lua, ...) are loaded from /root/reference as they are, and only the modules that need LuaJIT's FFI runtime or
the operating system stay stand-ins (radio.core.block / pipe / platform / vector, radio.types, format_utils: tests/helpers/lua_mocks.py).  Under
tests/helpers/minilua.py this shows that

  * the one-line patches survive the reference's real module bodies and load order (firfilter.lua's two ladders, block.factory(name, parent) copying the
    patched parent into LowpassFilterBlock / FMDeemphasisFilterBlock, spectrum_utils.lua's class tables);
  * the fields the device variants read are the ones the reference's instantiate() / initialize() really set (offset, factor, gain, taps, b_taps / a_taps,
    format, file, num_samples / overlap / update_time ...), with the reference's own tap design (filter_utils.firwin_lowpass in Lua) producing the taps;
  * the whole receiver of examples/rtlsdr_wbfm_mono.lua fed from IQFileSource collapses into one chain and makes the same library calls as the stand-in test;
  * the Python restatement of the host-side design code (luaradio_amd/filter_utils.py, used by every GPU parity test) equals the reference's Lua, bit for bit.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "radio")), reason="the reference checkout is not on this box")

from tests.helpers import lua_mocks as LM          # noqa: E402
from tests.helpers import minilua as ml            # noqa: E402
from tests.test_lua_blocks import fvec, lua_list, written_of, wbfm_u8_capture          # noqa: E402

# reference modules that are pure Lua and are executed AS THEY ARE
REAL_MODULES_OK = ("radio.core.class", "radio.core.util", "radio.utilities.", "radio.blocks.signal.", "radio.blocks.sources.iqfile", "radio.blocks.sources.realfile",
                   "radio.blocks.sinks.iqfile", "radio.blocks.sinks.realfile", "radio.blocks.sinks.gnuplotspectrum")


class VectorClass:
    """radio.core.vector.Vector as far as the block files use it: a constructor and an isinstanceof() target"""
    lua_type = "table"

    def __call__(self, ctype, n=0):
        size = getattr(ctype, "size", None) or ctype.dtype.itemsize
        v = LM.Vector(LM.DataType("Byte", np.uint8), int(ml.tonum(n or 0)) * size)
        v.ctype = ctype
        return v

    def lua_index(self, key):
        return None


def reference_interpreter(env=None):
    import apply_lua_binding as AB
    files = AB.patched_sources(REFERENCE)
    I, proxy, ffi = LM.make_interpreter(None, env)
    I.globals.set("__copy_vector", lambda v: LM.Vector(v.data_type, 0, v.array().copy()))
    vector_class = VectorClass()
    types_f32 = I.require("radio.types").get("Float32")
    types_cf = I.require("radio.types").get("ComplexFloat32")
    marker = LM.LuaTable()
    marker.set(vector_class, True)
    orig_vec_index = LM.Vector.lua_index

    def vec_index(self, key):                      # class.isinstanceof(v, vector.Vector) looks at v._types (radio/core/class.lua:49-51)
        if key == "_types":
            return marker
        return orig_vec_index(self, key)
    LM.Vector.lua_index = vec_index

    def from_array(dt):
        def f(arr):
            vals = [arr.get(k) for k in range(1, arr.length() + 1)]
            if dt.name == "ComplexFloat32":
                vals = [complex(v.get(1), v.get(2)) for v in vals]
            return LM.Vector(dt, 0, np.asarray(vals, dt.dtype))
        return f
    orig_dt_index = LM.DataType.lua_index

    def dt_index(self, key):
        if key == "vector_from_array":
            return from_array(self)
        return orig_dt_index(self, key)
    LM.DataType.lua_index = dt_index
    I.register("radio.core.vector", LM.T(Vector=vector_class))
    I.register("os", I.globals.get("os"))
    I.register("bit", LM.T(bor=lambda *a: float(int(a[0]) | int(a[1]))))
    loaded_real = []
    fallback = I.require

    def require(name):
        v = I.loaded.get(name)
        if v is not None:
            return v
        rel = name.replace(".", "/") + ".lua"
        text = files[rel][1] if rel in files and files[rel][0] is not None else None      # a reference file WITH its patch line
        if text is None and name.startswith(REAL_MODULES_OK) and os.path.exists(os.path.join(REFERENCE, rel)):
            text = open(os.path.join(REFERENCE, rel)).read()                               # a reference file as it is
        if text is not None:
            loaded_real.append(rel)
            r = I.run(text, rel, [name])
            if I.loaded.get(name) is None:
                I.loaded.set(name, r[0] if r and r[0] is not None else True)
            return I.loaded.get(name)
        return fallback(name)
    I.require = require
    I.globals.set("require", require)

    def restore():
        LM.Vector.lua_index = orig_vec_index
        LM.DataType.lua_index = orig_dt_index
    return I, proxy, ffi, loaded_real, restore


PREPARE = r'''
-- CompositeBlock:_prepare_to_run with the hooks of tools/apply_lua_binding.py on a flattened connection table (radio/core/composite.lua:426-470)
local function prepare(connections, blocks)
    local pipe = require('radio.core.pipe')
    local graphs, chains, device_chains
    connections, graphs = require('radio.composites.devicegraph').collapse(connections)
    connections, chains = require('radio.composites.devicechain').collapse(connections)
    connections, device_chains = require('radio.composites.devicefanout').collapse(connections, chains)
    for _, g in ipairs(graphs) do device_chains[#device_chains + 1] = g end
    for input, output in pairs(connections) do
        local p = pipe.Pipe(output, input)
        output.pipes[#output.pipes + 1] = p
        input.pipe = p
    end
    for _, b in ipairs(blocks) do b:initialize() end
    for _, c in ipairs(device_chains) do c:initialize() end
    return connections, device_chains
end
'''

WBFM_REAL = PREPARE + r'''
local types = require('radio.types')
local block = require('radio.core.block')
local IQFileSource = require('radio.blocks.sources.iqfile')
local FrequencyTranslatorBlock = require('radio.blocks.signal.frequencytranslator')
local LowpassFilterBlock = require('radio.blocks.signal.lowpassfilter')
local DownsamplerBlock = require('radio.blocks.signal.downsampler')
local FrequencyDiscriminatorBlock = require('radio.blocks.signal.frequencydiscriminator')
local FMDeemphasisFilterBlock = require('radio.blocks.signal.fmdeemphasisfilter')
local RealFileSink = require('radio.blocks.sinks.realfile')
local path, sink_path = ...
-- examples/rtlsdr_wbfm_mono.lua:12-17 with the RTL-SDR replaced by a recording; TunerBlock(-250e3, 200e3, 5) flattened as radio/composites/tuner.lua:40-43 builds it
local blocks = {IQFileSource(path, 'u8', 1102500), FrequencyTranslatorBlock(-250e3), LowpassFilterBlock(128, 200e3 / 2), DownsamplerBlock(5),
                FrequencyDiscriminatorBlock(1.25), LowpassFilterBlock(128, 15e3), FMDeemphasisFilterBlock(75e-6), DownsamplerBlock(5)}
local sink
if sink_path then
    sink = RealFileSink(sink_path, 'f32le')
else
    sink = block.factory("HostSink")
    function sink:instantiate() self:add_type_signature({block.Input("in", types.Float32)}, {}) end
    sink = sink()
end
blocks[#blocks + 1] = sink
local connections = {}
blocks[1]:differentiate({})
local t = blocks[1]:get_output_type()
for i = 2, #blocks do
    blocks[i]:differentiate({t})
    connections[blocks[i].inputs[1]] = blocks[i-1].outputs[1]
    if i < #blocks then t = blocks[i]:get_output_type() end
end
local conns, device_blocks = prepare(connections, blocks)
return conns, device_blocks, blocks
'''


def test_the_patched_reference_files_build_the_receiver_chain_from_a_u8_file(tmp_path):
    import luaradio_amd as lr
    n = 2 * 65536 + 999
    path = tmp_path / "x.u8"
    path.write_bytes(wbfm_u8_capture(n))
    I, proxy, ffi, loaded_real, restore = reference_interpreter()
    try:
        conns, devs, blocks = I.run(WBFM_REAL, "wbfm_real", [str(path), None])
        devs = lua_list(devs)
        assert len(devs) == 1 and ml.index(devs[0], "name") == "DeviceChainBlock" and len(conns.hash) == 1
        chain = devs[0]
        members = lua_list(ml.index(chain, "blocks"))
        assert [ml.index(b, "name") for b in members] == ["IQFileSource", "FrequencyTranslatorBlock", "LowpassFilterBlock", "DownsamplerBlock",
                                                           "FrequencyDiscriminatorBlock", "LowpassFilterBlock", "FMDeemphasisFilterBlock", "DownsamplerBlock"]
        # the files that ran are the reference's, with the patch line: block files, the FIR / IIR parents of the derived blocks, the tap design
        for rel in ("radio/blocks/sources/iqfile.lua", "radio/blocks/signal/firfilter.lua", "radio/blocks/signal/lowpassfilter.lua", "radio/blocks/signal/iirfilter.lua",
                    "radio/blocks/signal/singlepolelowpassfilter.lua", "radio/blocks/signal/fmdeemphasisfilter.lua", "radio/utilities/filter_utils.lua",
                    "radio/utilities/window_utils.lua", "radio/core/class.lua"):
            assert rel in loaded_real, rel
        chain.set("batch_samples", 65536.0)
        chain.set("source_batch_bytes", 0.0)
        ml.call(ml.index(chain, "run"), [chain])
        t = proxy.trace
        i = t.index("lrhip_chain_create_ex")
        assert t[i - 8:i] == ["lrhip_format_convert_create", "lrhip_rotator_create", "lrhip_fir_create", "lrhip_downsampler_create", "lrhip_fmdiscrim_create",
                              "lrhip_fir_create", "lrhip_iir_create", "lrhip_downsampler_create"]
        calls = proxy.fake.calls
        assert [a for nm, a in calls if nm == "lrhip_format_convert_create"] == [[b"u8", 1]]
        # what the reference's own initialize() code computed and the device variants handed over: omega, decimation, gain, the IIR taps
        omega = [a for nm, a in calls if nm == "lrhip_rotator_create"][0][0]
        assert omega == 2 * np.pi * (-250e3 / 1102500)
        assert [a[0] for nm, a in calls if nm == "lrhip_downsampler_create"] == [5, 5]
        assert [a[0] for nm, a in calls if nm == "lrhip_fmdiscrim_create"] == [2 * np.pi * 1.25]
        firs = [a for nm, a in calls if nm == "lrhip_fir_create"]
        assert [a[1:] for a in firs] == [[128, 0, 1, 1, 3], [128, 0, 0, 1, 3]]          # real taps; complex then real stream; use_fft left nil -> automatic
        # ... and the taps are the reference's own design (filter_utils.lua executed here), equal to the Python restatement the GPU tests use - bit for bit
        rf_taps = ml.index(members[2], "taps").array()
        af_taps = ml.index(members[5], "taps").array()
        assert np.array_equal(rf_taps, np.asarray(lr.filter_utils.firwin_lowpass(128, 100e3 / (1102500 / 2)), np.float32))
        assert np.array_equal(af_taps, np.asarray(lr.filter_utils.firwin_lowpass(128, 15e3 / (220500 / 2)), np.float32))
        py = lr.FMDeemphasisFilterBlock(75e-6)
        py.rate = 220500.0
        from tests.test_lua_blocks import deemphasis_taps
        b, a = deemphasis_taps(75e-6, 220500.0)
        assert np.array_equal(ml.index(members[6], "b_taps").array(), b) and np.array_equal(ml.index(members[6], "a_taps").array(), a)
        # the run: the library read the file (three batches), the audio went to the sink's pipe, the reference's cleanup() closed the file
        assert t.count("lrhip_chain_submit_fd") == 4 and [v.length for v in written_of(chain)] == [65536, 65536, 999]
        assert ml.index(members[0], "file").closed
    finally:
        restore()


def test_the_patched_reference_sink_and_spectrum_files(tmp_path):
    """file -> file with the reference's RealFileSink as the chain's tail, and the reference's spectrum_utils.lua classes on the library"""
    n = 40000
    path, out = tmp_path / "x.u8", tmp_path / "audio.f32"
    path.write_bytes(wbfm_u8_capture(n))
    I, proxy, ffi, loaded_real, restore = reference_interpreter()
    try:
        conns, devs, blocks = I.run(WBFM_REAL, "wbfm_real", [str(path), str(out)])
        chain = lua_list(devs)[0]
        assert len(conns.hash) == 0 and ml.index(chain, "blocks").length() == 9 and ml.index(chain, "sink") is blocks.get(9)
        ml.call(ml.index(chain, "run"), [chain])
        assert [a for nm, a in proxy.fake.calls if nm == "lrhip_format_pack_create"] == [[b"f32le", 0]]
        assert out.stat().st_size == 4 * n and ml.index(blocks.get(9), "file").closed
        # radio/utilities/spectrum_utils.lua (patched): DFT / IDFT / PSD objects built by the reference's constructors compute on the library
        # (a length the library has no transform for keeps the reference's own branch: tests/test_lua_blocks.py)
        src = r"""
        local types = require('radio.types')
        local S = require('radio.utilities.spectrum_utils')
        local x, X, p = types.ComplexFloat32.vector(128), types.ComplexFloat32.vector(128), types.Float32.vector(128)
        local d, i, psd = S.DFT(x, X), S.IDFT(X, x), S.PSD(x, p, 'hamming', 48000, true)
        d:compute(); i:compute(); psd:compute()
        return psd
        """
        psd = I.run(src, "spectrum", [])[0]
        assert "radio/utilities/spectrum_utils.lua" in loaded_real
        assert [a for nm, a in proxy.fake.calls if nm == "lrhip_dft_create"] == [[128, 0, 0], [128, 1, 0]]
        pc = [a for nm, a in proxy.fake.calls if nm == "lrhip_psd_create"][0]
        w = np.asarray([0.54 - 0.46 * np.cos(2 * np.pi * k / 128) for k in range(128)], np.float32)          # the periodic window of spectrum_utils.lua:547
        assert pc[0] == 128 and pc[3:] == [1, 1, 0] and abs(pc[2] - 48000 * float(np.sum(w.astype(np.float64) ** 2))) < 1e-3
        assert np.array_equal(ml.index(psd, "window").array(), w)
    finally:
        restore()
