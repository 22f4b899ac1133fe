"""ctypes/numpy binding of oracle/liblroracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under luaradio_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import math
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblroracle.so")

MODE_LUA, MODE_FMA, MODE_F64, MODE_SIMD = 0, 1, 2, 3


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("lr_oracle.c", "lr_cpu_baseline.c", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p
        L.lro_window.argtypes = [C.c_int, C.c_char_p, C.c_int, dp]
        L.lro_firwin_lowpass.argtypes = [C.c_int, C.c_double, C.c_char_p, dp]
        L.lro_firwin_highpass.argtypes = [C.c_int, C.c_double, C.c_char_p, dp]
        L.lro_firwin_bandpass.argtypes = [C.c_int, C.c_double, C.c_double, C.c_char_p, dp]
        L.lro_firwin_bandstop.argtypes = [C.c_int, C.c_double, C.c_double, C.c_char_p, dp]
        L.lro_fir_create.restype = vp
        L.lro_fir_create.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.lro_fir_process.restype = C.c_long
        L.lro_fir_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_fir_destroy.argtypes = [vp]
        L.lro_fir_process_simd.restype = C.c_long
        L.lro_fir_process_simd.argtypes = [vp, fp, C.c_long, fp, C.c_int]
        L.lrb_fir_dot.restype = C.c_long
        L.lrb_fir_dot.argtypes = [fp, C.c_int, C.c_int, fp, C.c_long, fp, C.c_int]
        L.lrb_fir_overlap_save.restype = C.c_long
        L.lrb_fir_overlap_save.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_long, fp, C.c_int]
        L.lro_firfft_create.restype = vp
        L.lro_firfft_create.argtypes = [fp, C.c_int, C.c_int, C.c_int]
        L.lro_firfft_process.restype = C.c_long
        L.lro_firfft_process.argtypes = [vp, fp, C.c_long, fp, C.c_long]
        L.lro_firfft_block_length.argtypes = [vp]
        L.lro_firfft_destroy.argtypes = [vp]
        L.lro_rotator_create.restype = vp
        L.lro_rotator_create.argtypes = [C.c_double, C.c_int]
        L.lro_rotator_process.restype = C.c_long
        L.lro_rotator_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_rotator_destroy.argtypes = [vp]
        L.lro_downsampler_create.restype = vp
        L.lro_downsampler_create.argtypes = [C.c_long, C.c_int]
        L.lro_downsampler_process.restype = C.c_long
        L.lro_downsampler_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_downsampler_destroy.argtypes = [vp]
        L.lro_fmdiscrim_create.restype = vp
        L.lro_fmdiscrim_create.argtypes = [C.c_double]
        L.lro_fmdiscrim_process.restype = C.c_long
        L.lro_fmdiscrim_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_fmdiscrim_destroy.argtypes = [vp]
        L.lro_agc_create.restype = vp
        L.lro_agc_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.lro_agc_process.restype = C.c_long
        L.lro_agc_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_agc_destroy.argtypes = [vp]
        L.lro_fmmod_create.restype = vp
        L.lro_fmmod_create.argtypes = [C.c_double]
        L.lro_fmmod_process.restype = C.c_long
        L.lro_fmmod_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_fmmod_destroy.argtypes = [vp]
        L.lro_iir_create.restype = vp
        L.lro_iir_create.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, C.c_int]
        L.lro_iir_process.restype = C.c_long
        L.lro_iir_process.argtypes = [vp, fp, C.c_long, fp]
        L.lro_iir_destroy.argtypes = [vp]
        L.lro_singlepole_lowpass_taps.argtypes = [C.c_double, C.c_double, fp, fp]
        L.lro_dft.argtypes = [fp, C.c_int, C.c_int, fp]
        L.lro_idft.argtypes = [fp, C.c_int, C.c_int, fp]
        L.lro_psd.argtypes = [fp, C.c_int, C.c_int, C.c_char_p, C.c_double, C.c_int, fp]
        L.lro_fftshift.argtypes = [fp, C.c_int, C.c_int]
        L.lro_multiply_conjugate.argtypes = [fp, fp, C.c_long, fp]
        L.lro_format_convert.restype = C.c_long
        L.lro_format_convert.argtypes = [C.c_char_p, C.c_char_p, C.c_long, fp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _as_f32(x):
    """complex64 / float32 array -> (flat float32 view, is_complex)."""
    x = np.ascontiguousarray(x)
    if np.iscomplexobj(x):
        x = x.astype(np.complex64, copy=False)
        return x.view(np.float32), True
    return x.astype(np.float32, copy=False), False


def _out(n, is_complex):
    return np.empty(n, dtype=np.complex64 if is_complex else np.float32)


# ---------------------------------------------------------------- tap design (double)
def window(M, window_type, periodic=False):
    w = np.empty(M, dtype=np.float64)
    if lib().lro_window(M, window_type.encode(), int(periodic), w.ctypes.data_as(C.POINTER(C.c_double))):
        raise ValueError("Unsupported window %r" % window_type)
    return w


def _firwin(fn, M, *args):
    h = np.empty(M, dtype=np.float64)
    rc = fn(M, *args, h.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise ValueError("firwin failed rc=%d" % rc)
    return h


def firwin_lowpass(M, cutoff, window_type="hamming"):
    return _firwin(lib().lro_firwin_lowpass, M, cutoff, window_type.encode())


def firwin_highpass(M, cutoff, window_type="hamming"):
    return _firwin(lib().lro_firwin_highpass, M, cutoff, window_type.encode())


def firwin_bandpass(M, cutoffs, window_type="hamming"):
    return _firwin(lib().lro_firwin_bandpass, M, cutoffs[0], cutoffs[1], window_type.encode())


def firwin_bandstop(M, cutoffs, window_type="hamming"):
    return _firwin(lib().lro_firwin_bandstop, M, cutoffs[0], cutoffs[1], window_type.encode())


# ---------------------------------------------------------------- stateful blocks
class _Stage:
    _destroy = None

    def __del__(self):
        if getattr(self, "q", None) and self._destroy:
            getattr(lib(), self._destroy)(self.q)
            self.q = None


class FIR(_Stage):
    """FIRFilterBlock dot-product form (firfilter.lua:230-305)."""
    _destroy = "lro_fir_destroy"

    def __init__(self, taps, input_complex, mode=MODE_LUA):
        t, tc = _as_f32(np.asarray(taps))
        self.input_complex = bool(input_complex)
        self.q = lib().lro_fir_create(_fp(t), t.size // (2 if tc else 1), int(tc), int(input_complex), mode)
        if not self.q:
            raise ValueError("bad FIR config")

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc == self.input_complex
        y = _out(len(x), xc)
        lib().lro_fir_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)))
        return y


def _fir_process_simd(self, x, nthreads=1, out=None):
    """VOLK-style SIMD dot products (the timed CPU baseline), optionally OpenMP-threaded."""
    xf, xc = _as_f32(x)
    assert xc == self.input_complex
    y = out if out is not None else _out(len(x), xc)
    n = lib().lro_fir_process_simd(self.q, _fp(xf), len(x), _fp(y.view(np.float32)), int(nthreads))
    assert n == len(x)
    return y


FIR.process_simd = _fir_process_simd


def baseline_fir_dot(taps, x, nthreads=1, out=None):
    """Timed CPU form (1): one SIMD dot product per output (firfilter.lua:129-145), zero history.  lr_cpu_baseline.c"""
    t, tc = _as_f32(np.asarray(taps))
    assert not tc
    xf, xc = _as_f32(x)
    y = out if out is not None else _out(len(x), xc)
    n = lib().lrb_fir_dot(_fp(t), t.size, int(xc), _fp(xf), len(x), _fp(y.view(np.float32)), int(nthreads))
    assert n == len(x)
    return y


def baseline_fir_overlap_save(taps, x, nthreads=1, out=None):
    """Timed CPU form (2): FFT overlap-save, the reference's default (firfilter.lua:57, :320-398), Float32 Stockham FFT,
    zero history, every output emitted.  lr_cpu_baseline.c"""
    t, tc = _as_f32(np.asarray(taps))
    xf, xc = _as_f32(x)
    y = out if out is not None else _out(len(x), xc)
    n = lib().lrb_fir_overlap_save(_fp(t), t.size // (2 if tc else 1), int(tc), int(xc), _fp(xf), len(x), _fp(y.view(np.float32)), int(nthreads))
    assert n == len(x)
    return y


class FIRFFT(_Stage):
    """FIRFilterBlock overlap-save form (firfilter.lua:406-486): chunked emission."""
    _destroy = "lro_firfft_destroy"

    def __init__(self, taps, input_complex):
        t, tc = _as_f32(np.asarray(taps))
        self.input_complex = bool(input_complex)
        self.q = lib().lro_firfft_create(_fp(t), t.size // (2 if tc else 1), int(tc), int(input_complex))
        self.block_length = lib().lro_firfft_block_length(self.q)

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc == self.input_complex
        cap = len(x) + self.block_length
        y = _out(cap, xc)
        n = lib().lro_firfft_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)), cap)
        assert n >= 0
        return y[:n].copy()


class Rotator(_Stage):
    """FrequencyTranslatorBlock (frequencytranslator.lua:93-110); mode F64 = closed form."""
    _destroy = "lro_rotator_destroy"

    def __init__(self, omega, mode=MODE_LUA):
        self.q = lib().lro_rotator_create(float(omega), mode)

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc
        y = _out(len(x), True)
        lib().lro_rotator_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)))
        return y


class Downsampler(_Stage):
    """DownsamplerBlock (downsampler.lua:40-56)."""
    _destroy = "lro_downsampler_destroy"

    def __init__(self, factor, input_complex):
        self.input_complex = bool(input_complex)
        self.q = lib().lro_downsampler_create(int(factor), 2 if input_complex else 1)
        if not self.q:
            raise ValueError("bad factor")

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc == self.input_complex
        y = _out(len(x) + 1, xc)
        n = lib().lro_downsampler_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)))
        return y[:n].copy()


class FMDiscriminator(_Stage):
    """FrequencyDiscriminatorBlock (frequencydiscriminator.lua:68-88)."""
    _destroy = "lro_fmdiscrim_destroy"

    def __init__(self, modulation_index):
        self.q = lib().lro_fmdiscrim_create(float(modulation_index))

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc
        y = _out(len(x), False)
        lib().lro_fmdiscrim_process(self.q, _fp(xf), len(x), _fp(y))
        return y


class AGC(_Stage):
    """AGCBlock (agc.lua:25-96): AGC(mode, target_dB, threshold_dB, rate[, gain_tau, power_tau])."""
    _destroy = "lro_agc_destroy"

    def __init__(self, mode, target, threshold, rate, input_complex, gain_tau=None, power_tau=1.0):
        gain_tau = {"fast": 0.1, "slow": 3.0}.get(mode, gain_tau)
        self.input_complex = bool(input_complex)
        self.q = lib().lro_agc_create(1 / (1 + power_tau * rate), 1 / (1 + gain_tau * rate), 10 ** (target / 10), 10 ** (threshold / 10),
                                      int(input_complex))

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc == self.input_complex
        y = _out(len(x), xc)
        lib().lro_agc_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)))
        return y


class FMModulator(_Stage):
    """FrequencyModulatorBlock (frequencymodulator.lua:71-90, pure-Lua branch)."""
    _destroy = "lro_fmmod_destroy"

    def __init__(self, modulation_index):
        self.q = lib().lro_fmmod_create(float(modulation_index))

    def process(self, x):
        xf, xc = _as_f32(x)
        assert not xc
        y = _out(len(x), True)
        lib().lro_fmmod_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)))
        return y


class IIR(_Stage):
    """IIRFilterBlock (iirfilter.lua:113-181)."""
    _destroy = "lro_iir_destroy"

    def __init__(self, b, a, input_complex, mode=MODE_LUA):
        b = np.ascontiguousarray(b, dtype=np.float32)
        a = np.ascontiguousarray(a, dtype=np.float32)
        self.input_complex = bool(input_complex)
        self.q = lib().lro_iir_create(_fp(b), len(b), _fp(a), len(a), int(input_complex), mode)

    def process(self, x):
        xf, xc = _as_f32(x)
        assert xc == self.input_complex
        y = _out(len(x), xc)
        lib().lro_iir_process(self.q, _fp(xf), len(x), _fp(y.view(np.float32)))
        return y


def singlepole_lowpass_taps(cutoff, rate):
    """SinglepoleLowpassFilterBlock:initialize (singlepolelowpassfilter.lua:55-67) -> (b[2], a[2]) f32."""
    b = np.empty(2, np.float32)
    a = np.empty(2, np.float32)
    lib().lro_singlepole_lowpass_taps(float(cutoff), float(rate), _fp(b), _fp(a))
    return b, a


def fm_deemphasis_taps(tau, rate):
    """FMDeemphasisFilterBlock (fmdeemphasisfilter.lua:24-27): cutoff = 1/(2 pi tau)."""
    return singlepole_lowpass_taps(1 / (2 * np.pi * tau), rate)


# ---------------------------------------------------------------- spectrum
def dft(x):
    xf, xc = _as_f32(x)
    y = np.empty(len(x), np.complex64)
    if lib().lro_dft(_fp(xf), len(x), int(xc), _fp(y.view(np.float32))):
        raise ValueError("DFT length must be even.")
    return y


def idft(X, output_complex=True):
    Xf, _ = _as_f32(np.asarray(X, dtype=np.complex64))
    y = _out(len(X), output_complex)
    if lib().lro_idft(_fp(Xf), len(X), int(output_complex), _fp(y.view(np.float32))):
        raise ValueError("DFT length must be even.")
    return y


def psd(x, window_type="hamming", sample_rate=2.0, logarithmic=True):
    xf, xc = _as_f32(x)
    y = np.empty(len(x), np.float32)
    rc = lib().lro_psd(_fp(xf), len(x), int(xc), window_type.encode(), float(sample_rate), int(logarithmic), _fp(y))
    if rc:
        raise ValueError("psd failed rc=%d" % rc)
    return y


def fftshift(x):
    xf, xc = _as_f32(np.array(x, copy=True))
    lib().lro_fftshift(_fp(xf), len(x), int(xc))
    return xf.view(np.complex64) if xc else xf


class WelchSpectrum:
    """GnuplotSpectrumSink's averaging loop, radio/blocks/sinks/gnuplotspectrum.lua:140-193, restated with the oracle's PSD:
    fill a state buffer of num_samples, PSD (logarithmic) -> fftshift -> accumulate (Float32), keep num_overlap samples.
    No golden vectors exist for the sink in the reference (it writes gnuplot text); this restatement is pinned only
    through the PSD / fftshift vectors of tests/utilities/spectrum_utils_vectors.gen.lua."""

    def __init__(self, is_complex, num_samples=1024, window_type="hamming", sample_rate=2.0, overlap=0.0, reference_level=0.0):
        self.n, self.window_type, self.sample_rate = num_samples, window_type, sample_rate
        self.num_overlap = int(np.floor(overlap * num_samples))          # :121
        self.reference_level = reference_level
        self.state = np.zeros(num_samples, np.complex64 if is_complex else np.float32)
        self.state_index = 0
        self.acc = np.zeros(num_samples, np.float32)
        self.count = 0

    def process(self, x):
        i = 0
        while i < len(x):                                                 # :149-172
            num = min(self.n - self.state_index, len(x) - i)
            self.state[self.state_index:self.state_index + num] = x[i:i + num]
            self.state_index += num
            i += num
            if self.state_index == self.n:
                p = fftshift(psd(self.state, self.window_type, self.sample_rate, True))
                self.acc = (self.acc + p).astype(np.float32)              # Float32 accumulation, frame order
                self.count += 1
                self.state[:self.num_overlap] = self.state[self.n - self.num_overlap:].copy()
                self.state_index = self.num_overlap

    def average(self):
        if not self.count:
            return None
        out = (self.acc / np.float32(self.count) - np.float32(self.reference_level)).astype(np.float32)   # :179-181
        self.acc[:] = 0
        self.count = 0
        return out


class Waterfall:
    """GnuplotWaterfallSink's process loop, radio/blocks/sinks/gnuplotwaterfall.lua:184-236 (rows of value_to_pixel(normalize(mean log PSD)),
    :147-182), restated with the oracle's PSD.  The reference holds no vector for the sink (it writes to a gnuplot pipe): unpinned beyond the
    PSD / fftshift vectors."""

    def __init__(self, is_complex, num_samples=1024, window_type="hamming", sample_rate=2.0, overlap=0.0, num_psd_averages=1, min_magnitude=-150.0,
                 max_magnitude=0.0, rows=64):
        self.w = WelchSpectrum(is_complex, num_samples, window_type, sample_rate, overlap, 0.0)
        self.navg, self.lo, self.hi = num_psd_averages, float(min_magnitude), float(max_magnitude)
        self.pixels = np.zeros((rows, num_samples, 3), np.uint8)
        self.rows_added = 0

    @staticmethod
    def _normalize(value, lo, hi):
        return (max(min(value, hi), lo) - lo) / (hi - lo)

    @classmethod
    def value_to_pixel(cls, value):
        n = cls._normalize
        if value < 1 / 5:
            c = math.floor(255 * n(value, 0, 1 / 5)); return (0, 0, c)
        if value < 2 / 5:
            c = math.floor(255 * n(value, 1 / 5, 2 / 5)); return (0, c, 255 - c)
        if value < 3 / 5:
            c = math.floor(255 * n(value, 2 / 5, 3 / 5)); return (c, 255, 0)
        if value < 4 / 5:
            c = math.floor(255 * n(value, 3 / 5, 4 / 5)); return (255, 255 - c, 0)
        c = math.floor(255 * n(value, 4 / 5, 5 / 5)); return (255, c, c)

    def process(self, x):
        # one sample at a time keeps the reference's loop order (a frame completes, then the row check, :191-235) without restating its
        # index arithmetic; sizes in the tests are small
        for k in range(len(x)):
            self.w.process(x[k:k + 1])
            if self.w.count == self.navg:
                avg = self.w.average()
                self.pixels[:-1] = self.pixels[1:]
                self.pixels[-1] = [self.value_to_pixel(self._normalize(float(v), self.lo, self.hi)) for v in avg]
                self.rows_added += 1


def multiply_conjugate(a, b):
    af, _ = _as_f32(np.asarray(a, np.complex64))
    bf, _ = _as_f32(np.asarray(b, np.complex64))
    y = np.empty(len(a), np.complex64)
    lib().lro_multiply_conjugate(_fp(af), _fp(bf), len(a), _fp(y.view(np.float32)))
    return y


FORMAT_BYTES = {"u8": 1, "s8": 1, "u16le": 2, "u16be": 2, "s16le": 2, "s16be": 2, "u32le": 4, "u32be": 4,
                "s32le": 4, "s32be": 4, "f32le": 4, "f32be": 4, "f64le": 8, "f64be": 8}


def format_convert(fmt, raw, complex_out):
    """IQFileSource / RealFileSource conversion of raw file bytes (iqfile.lua:99-113, realfile.lua:99-110)."""
    nscalars = len(raw) // FORMAT_BYTES[fmt]
    out = np.empty(nscalars, np.float32)
    n = lib().lro_format_convert(fmt.encode(), bytes(raw), nscalars, _fp(out))
    if n < 0:
        raise ValueError('Unsupported format ("%s")' % fmt)
    return out.view(np.complex64) if complex_out else out


# ---------------------------------------------------------------- composites (composition of the pinned blocks)
class Chain:
    def __init__(self, stages):
        self.stages = stages

    def process(self, x):
        for s in self.stages:
            x = s.process(x)
        return x


def lowpass(num_taps, cutoff, rate, input_complex, nyquist=None, window_type="hamming", mode=MODE_LUA):
    """LowpassFilterBlock (lowpassfilter.lua:32-50)."""
    nyq = nyquist if nyquist is not None else rate / 2
    taps = firwin_lowpass(num_taps, cutoff / nyq, window_type).astype(np.float32)
    return FIR(taps, input_complex, mode)


def decimator(factor, rate, input_complex, num_taps=128, window_type="hamming", mode=MODE_LUA):
    """DecimatorBlock (composites/decimator.lua:28-42): Lowpass(num_taps, 1/M, nyquist=1.0) -> Downsampler(M)."""
    return Chain([lowpass(num_taps, 1.0 / factor, rate, input_complex, 1.0, window_type, mode),
                  Downsampler(factor, input_complex)])


def tuner(offset, bandwidth, decimation, rate, num_taps=128, window_type="hamming", mode=MODE_LUA, rot_mode=MODE_LUA):
    """TunerBlock (composites/tuner.lua:32-48)."""
    return Chain([Rotator(2 * np.pi * (offset / rate), rot_mode),
                  lowpass(num_taps, bandwidth / 2, rate, True, None, window_type, mode),
                  Downsampler(decimation, True)])


def wbfm_mono_chain(rate=1102500.0, tune_offset=-250e3, mode=MODE_LUA, rot_mode=MODE_LUA):
    """examples/rtlsdr_wbfm_mono.lua:12-17,28: Tuner -> Discriminator -> Lowpass -> Deemphasis -> Downsampler."""
    r1 = rate / 5
    b, a = fm_deemphasis_taps(75e-6, r1)
    return Chain(tuner(tune_offset, 200e3, 5, rate, mode=mode, rot_mode=rot_mode).stages + [
        FMDiscriminator(1.25),
        lowpass(128, 15e3, r1, False, mode=mode),
        IIR(b, a, False, mode),
        Downsampler(5, False)])
