"""Block base class - the slice of radio/core/block.lua the hot path touches.

Mirrors: type-signature differentiation (block.lua:238-352), get_input_type / get_output_type / get_rate
(:354-390) and the instantiate / initialize / process / cleanup hooks (:459-485).  A device block owns one
`lrhip_stage_t` (include/lrhip.h); process(x) is one lrhip_stage_execute() call, exactly what the Lua
process() body does through LuaJIT FFI (lua/radio/blocks/signal/*.lua).
"""
import ctypes as C

import numpy as np

from . import _lib, types


class Input:
    def __init__(self, name, data_type):
        self.name, self.data_type = name, data_type


class Output:
    def __init__(self, name, data_type):
        self.name, self.data_type = name, data_type


class Block:
    name = "Block"

    def __init__(self, *args, **kwargs):
        self.type_signatures = []
        self.signature = None
        self.rate = None          # set by the graph (upstream propagation) or by the caller
        self._stage = None
        self.instantiate(*args, **kwargs)

    # ---- hooks (block.lua:459-485)
    def instantiate(self, *args):
        pass

    def initialize(self):
        pass

    def cleanup(self):
        pass

    def process(self, x):
        raise NotImplementedError("process() not implemented")

    # ---- type signatures (block.lua:238-352)
    def add_type_signature(self, inputs, outputs, tag=None):
        self.type_signatures.append((inputs, outputs, tag))

    def differentiate(self, input_types):
        for inputs, outputs, tag in self.type_signatures:
            if len(inputs) == len(input_types) and all(i.data_type is t for i, t in zip(inputs, input_types)):
                self.signature = (inputs, outputs, tag)
                return
        raise TypeError("No compatible type signatures found for block %s with input types [%s]."
                        % (self.name, ", ".join(map(str, input_types))))

    def get_input_type(self, index=1):
        return self.signature[0][index - 1].data_type

    def get_output_type(self, index=1):
        return self.signature[1][index - 1].data_type

    def get_rate(self):
        if self.rate is None:
            raise RuntimeError("Block %s has no sample rate: connect it in a graph or set block.rate" % self.name)
        return self.rate

    # ---- device plumbing
    def _set_stage(self, ptr, what):
        self._destroy_stage()
        self._stage = _lib.check_ptr(ptr, what)

    def _destroy_stage(self):
        if self._stage:
            _lib.load().lrhip_stage_destroy(self._stage)
            self._stage = None

    def __del__(self):
        try:
            self._destroy_stage()
        except Exception:
            pass

    def stage_handle(self):
        if not self._stage:
            raise RuntimeError("Block %s is not initialized" % self.name)
        return self._stage

    def _execute(self, x, out_dtype):
        """one process() call through the C ABI with host vectors"""
        L = _lib.load()
        x = np.ascontiguousarray(x)
        if x.dtype != self.get_input_type().dtype:
            raise TypeError("Block %s expects %s input, got %s" % (self.name, self.get_input_type(), x.dtype))
        cap = L.lrhip_stage_max_output(self._stage, len(x))
        out = np.empty(cap, dtype=out_dtype)
        n = L.lrhip_stage_execute(self._stage, x.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p), cap)
        _lib.check(n, "%s:process" % self.name)
        return out[:n]

    def process_device(self, in_ptr, n_in, out_ptr, out_capacity):
        """device-resident process(): raw device addresses (e.g. torch tensor .data_ptr()), asynchronous on the
        library stream.  Returns the number of output samples."""
        n = _lib.load().lrhip_stage_execute_device(self.stage_handle(), in_ptr, n_in, out_ptr, out_capacity)
        return _lib.check(n, "%s:process_device" % self.name)

    def max_output(self, n_in):
        return _lib.load().lrhip_stage_max_output(self.stage_handle(), n_in)

    def seek(self, n0):
        """time-axis sharding (include/lrhip.h): continue as if n0 input samples of the stream had been consumed - zero history,
        absolute rotator / decimation phase"""
        _lib.check(_lib.load().lrhip_stage_seek(self.stage_handle(), int(n0)), "%s:seek" % type(self).__name__)

    def reset(self):
        _lib.check(_lib.load().lrhip_stage_reset(self.stage_handle()), "%s:reset" % self.name)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def as_taps(taps):
    """FIRFilterBlock:instantiate tap handling (firfilter.lua:43-54): Float32 vector, ComplexFloat32 vector or
    a plain number array (-> Float32)."""
    if taps is None:
        raise AssertionError("Missing argument #1 (taps)")
    if isinstance(taps, np.ndarray) and taps.dtype == np.complex64:
        return np.ascontiguousarray(taps)
    if isinstance(taps, np.ndarray) and taps.dtype == np.float32:
        return np.ascontiguousarray(taps)
    if isinstance(taps, (list, tuple)) or (isinstance(taps, np.ndarray) and taps.dtype.kind == "f"):
        return types.Float32.vector_from_array(taps)
    raise TypeError("Unsupported taps type")


# firfilter.lua:57: `use_fft == nil` means "FFT form when available ... and not package.loaded['tests.jigs']" - the reference's unit-test
# jig expects one output per input from filters it did not explicitly ask FFT framing of.  tests/conftest.py (this repository's jig)
# sets the flag the same way; the Lua glue reads package.loaded['tests.jigs'] itself.
TESTS_JIGS_LOADED = False


def fir_mode(use_fft):
    """FIRFilterBlock's use_fft argument -> lrhip_fir_create's mode: the ONE table of every front end (lrhip.fir_mode in
    lua/radio/core/lrhip.lua is the same, tests/test_host_cpu.py holds the two together).  None (the caller did not choose; the
    reference then picks its FFT form when FFTW is present - except under its unit-test jig, firfilter.lua:57) = 3 automatic, 0 under
    the jig; "auto" = 3; "fast" = 2 overlap-save
    arithmetic, one output per input; True = 1 the reference's overlap-save INCLUDING its block-emission framing
    (firfilter.lua:361-398); False = 0 direct form (bit-identical to the fmaf chain in tap order)."""
    if use_fft is None:
        return 0 if TESTS_JIGS_LOADED else 3
    if use_fft == "auto":
        return 3
    if use_fft == "fast":
        return 2
    if use_fft in (0, 1, 2, 3) and not isinstance(use_fft, bool):
        return int(use_fft)             # already a mode number (blk.use_fft pokes of the tools)
    return 1 if use_fft else 0
