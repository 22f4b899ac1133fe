"""DFT / IDFT / PSD / fftshift - mirrors the classes of radio/utilities/spectrum_utils.lua on the device.

Same constructor shapes as the reference (input vector, output vector) and a compute() method; the vectors are
numpy arrays bound at construction, like the reference's FFTW plans bind their buffers (:89-91)."""
import ctypes as C

import numpy as np

from . import _lib, types, window_utils


class _Transform:
    def __init__(self):
        self._stage = None

    def __del__(self):
        try:
            if self._stage:
                _lib.load().lrhip_stage_destroy(self._stage)
                self._stage = None
        except Exception:
            pass

    def compute(self):
        L = _lib.load()
        n = L.lrhip_stage_execute(self._stage, self.input_samples.ctypes.data_as(C.c_void_p), len(self.input_samples),
                                  self.output_samples.ctypes.data_as(C.c_void_p), len(self.output_samples))
        _lib.check(n, type(self).__name__ + ":compute")


def _check(input_samples, output_samples, in_types, out_types, what):
    # spectrum_utils.lua:28-37
    if types.type_of(input_samples) not in in_types:
        raise TypeError("Unsupported input samples data type.")
    if types.type_of(output_samples) not in out_types:
        raise TypeError("Unsupported output samples data type.")
    if len(input_samples) != len(output_samples):
        raise ValueError("Input samples and output samples length mismatch.")
    if len(input_samples) % 2:
        raise ValueError("%s length must be even." % what)


class DFT(_Transform):
    """spectrum_utils.lua:25-57"""

    def __init__(self, input_samples, output_samples):
        super().__init__()
        _check(input_samples, output_samples, (types.ComplexFloat32, types.Float32), (types.ComplexFloat32,), "DFT")
        self.input_samples, self.output_samples = input_samples, output_samples
        self._stage = _lib.check_ptr(_lib.load().lrhip_dft_create(len(input_samples), 0, int(input_samples.dtype == np.float32)),
                                     "Creating lrhip dft object")


class IDFT(_Transform):
    """spectrum_utils.lua:259-291"""

    def __init__(self, input_samples, output_samples):
        super().__init__()
        _check(input_samples, output_samples, (types.ComplexFloat32,), (types.ComplexFloat32, types.Float32), "DFT")
        self.input_samples, self.output_samples = input_samples, output_samples
        self._stage = _lib.check_ptr(_lib.load().lrhip_dft_create(len(input_samples), 1, int(output_samples.dtype == np.float32)),
                                     "Creating lrhip idft object")


class PSD(_Transform):
    """spectrum_utils.lua:522-561: PSD(input, output[, window_type='hamming'[, sample_rate=2[, logarithmic=true]]])"""

    def __init__(self, input_samples, output_samples, window_type=None, sample_rate=None, logarithmic=None, frames=1, fftshift=False):
        super().__init__()
        if types.type_of(input_samples) not in (types.ComplexFloat32, types.Float32):
            raise TypeError("Unsupported input samples data type.")
        if types.type_of(output_samples) is not types.Float32:
            raise TypeError("Unsupported output samples data type.")
        if len(input_samples) != len(output_samples):
            raise ValueError("Input samples and output samples length mismatch.")
        if len(input_samples) % frames or (len(input_samples) // frames) % 2:
            raise ValueError("PSD length must be even.")
        self.input_samples, self.output_samples = input_samples, output_samples
        self.window_type = window_type or "hamming"
        self.sample_rate = sample_rate or 2
        self.logarithmic = True if logarithmic is None else logarithmic
        self.num_samples = len(input_samples) // frames
        # :547 periodic window, stored as Float32; :550-553 energy summed from the Float32 values
        self.window = types.Float32.vector_from_array(window_utils.window(self.num_samples, self.window_type, True))
        self.window_energy = 0.0
        for v in self.window:
            self.window_energy = self.window_energy + float(v) * float(v)
        scale = self.sample_rate * self.window_energy           # :597
        self._stage = _lib.check_ptr(
            _lib.load().lrhip_psd_create(self.num_samples, self.window.ctypes.data_as(C.POINTER(C.c_float)), scale,
                                         int(self.logarithmic), int(input_samples.dtype == np.complex64), int(fftshift)),
            "Creating lrhip psd object")


def fftshift(samples):
    """spectrum_utils.lua:654-667 - in-place swap of the two halves (host vectors; on the device the shift is
    fused into PSD via fftshift=True)."""
    off = len(samples) // 2
    tmp = samples[:off].copy()
    samples[:off] = samples[off:2 * off]
    samples[off:2 * off] = tmp



class WelchSpectrum(_Transform):
    """The arithmetic of GnuplotSpectrumSink / GnuplotWaterfallSink (radio/blocks/sinks/gnuplotspectrum.lua:104-134,
    140-193): frames of num_samples every num_samples - floor(overlap*num_samples) samples, log PSD of each frame,
    fftshift, running average; the gnuplot text protocol itself stays on the host and is out of scope.

    WelchSpectrum(data_type[, num_samples=1024[, window='hamming'[, sample_rate=2[, overlap=0.0[, reference_level=0.0]]]]])
    process(x) consumes a chunk; average() returns (mean PSD - reference_level) and resets, or None if no frame is complete."""

    def __init__(self, data_type, num_samples=1024, window=None, sample_rate=None, overlap=0.0, reference_level=0.0, logarithmic=True):
        super().__init__()
        if data_type not in (types.ComplexFloat32, types.Float32):
            raise TypeError("Unsupported input samples data type.")
        if num_samples % 2:
            raise ValueError("PSD length must be even.")
        self.data_type, self.num_samples = data_type, int(num_samples)
        self.num_overlap = int(np.floor(overlap * num_samples))                  # gnuplotspectrum.lua:121
        self.reference_level = reference_level
        self.sample_rate = sample_rate or 2
        self.window = types.Float32.vector_from_array(window_utils.window(self.num_samples, window or "hamming", True))
        energy = 0.0
        for v in self.window:
            energy = energy + float(v) * float(v)
        self._stage = _lib.check_ptr(
            _lib.load().lrhip_welch_create(self.num_samples, self.window.ctypes.data_as(C.POINTER(C.c_float)), self.sample_rate * energy,
                                           int(logarithmic), int(data_type is types.ComplexFloat32), self.num_overlap),
            "Creating lrhip welch object")

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=self.data_type.dtype)
        n = _lib.load().lrhip_stage_execute(self._stage, x.ctypes.data_as(C.c_void_p), len(x), None, 0)
        _lib.check(n, "WelchSpectrum:process")

    def process_device(self, in_ptr, n):
        _lib.check(_lib.load().lrhip_stage_execute_device(self._stage, in_ptr, n, None, 0), "WelchSpectrum:process_device")

    def average(self, reset=True):
        out = np.empty(self.num_samples, np.float32)
        frames = _lib.check(_lib.load().lrhip_welch_read(self._stage, out.ctypes.data_as(C.POINTER(C.c_float)), int(reset)), "WelchSpectrum:average")
        if not frames:
            return None
        self.frames = int(frames)
        return out - np.float32(self.reference_level)
