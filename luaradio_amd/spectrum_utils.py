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


def value_to_pixel(value):
    """radio/blocks/sinks/gnuplotwaterfall.lua:151-182 on a vector: normalised magnitude in [0, 1] -> RGB, five linear segments
    black -> blue -> green -> yellow -> red -> white (Lua doubles, math.floor)"""
    v = np.asarray(value, np.float64)
    norm = lambda x, lo, hi: (np.maximum(np.minimum(x, hi), lo) - lo) / (hi - lo)
    rgb = np.zeros(v.shape + (3,), np.uint8)
    seg = np.minimum((v * 5).astype(np.int64), 4)
    seg = np.where(v < 1 / 5, 0, np.where(v < 2 / 5, 1, np.where(v < 3 / 5, 2, np.where(v < 4 / 5, 3, 4))))
    c = [np.floor(255 * norm(v, k / 5, (k + 1) / 5)).astype(np.int64) for k in range(5)]
    z, f = np.zeros_like(c[0]), np.full_like(c[0], 255)
    table = [(z, z, c[0]), (z, c[1], 255 - c[1]), (c[2], f, z), (f, 255 - c[3], z), (f, c[4], c[4])]
    for k, (r, g, b) in enumerate(table):
        m = seg == k
        rgb[m, 0], rgb[m, 1], rgb[m, 2] = r[m], g[m], b[m]
    return rgb


class WaterfallSpectrum:
    """The arithmetic of GnuplotWaterfallSink (radio/blocks/sinks/gnuplotwaterfall.lua:184-236): Welch frames as in WelchSpectrum (on the
    device), every `num_psd_averages` frames one new pixel row - averaged log PSD, clamped to [min_magnitude, max_magnitude], mapped through
    value_to_pixel - pushed into a `rows` x num_samples RGB image that scrolls up.  The gnuplot pipe itself stays on the host, out of scope.

    WaterfallSpectrum(data_type[, num_samples=1024[, window[, sample_rate[, overlap=0.0[, num_psd_averages=1[, min_magnitude=-150[,
                      max_magnitude=0[, rows=64]]]]]]]])
    process(x) -> number of rows added; .pixels is the image (uint8 [rows][columns][3])."""

    def __init__(self, data_type, num_samples=1024, window=None, sample_rate=None, overlap=0.0, num_psd_averages=1, min_magnitude=-150.0,
                 max_magnitude=0.0, rows=64):
        if not overlap < 1:
            raise AssertionError("Overlap should be a fraction in [0.00, 1.00)")        # gnuplotwaterfall.lua:60
        self.welch = WelchSpectrum(data_type, num_samples, window, sample_rate, overlap, 0.0)
        self.num_samples, self.columns, self.rows = self.welch.num_samples, self.welch.num_samples, int(rows)
        self.num_psd_averages = int(num_psd_averages)
        self.min_magnitude, self.max_magnitude = float(min_magnitude), float(max_magnitude)
        self.hop = self.num_samples - self.welch.num_overlap
        self.fill, self.frames = 0, 0                   # host mirror of the device's frame counters (no read-back needed to know them)
        self.pixels = np.zeros((self.rows, self.columns, 3), np.uint8)

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=self.welch.data_type.dtype)
        i, added = 0, 0
        while i < len(x):
            # samples that complete exactly the frames still missing for the next row
            m = (self.num_samples - self.fill) + (self.num_psd_averages - self.frames - 1) * self.hop
            take = min(m, len(x) - i)
            self.welch.process(x[i:i + take])
            i += take
            total = self.fill + take
            if total >= self.num_samples:
                f = (total - self.num_samples) // self.hop + 1
                self.fill = total - f * self.hop
                self.frames += f
            else:
                self.fill = total
            if self.frames == self.num_psd_averages:
                avg = self.welch.average()              # mean of the frames' log PSDs (gnuplotwaterfall.lua:213-216), fftshifted
                value = (np.maximum(np.minimum(avg.astype(np.float64), self.max_magnitude), self.min_magnitude) - self.min_magnitude) / (self.max_magnitude - self.min_magnitude)
                self.pixels[:-1] = self.pixels[1:]      # :219 shift pixels one row up
                self.pixels[-1] = value_to_pixel(value)
                self.frames = 0
                added += 1
        return added
