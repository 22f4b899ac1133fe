"""Device variants of the signal blocks on the hot path (same names, arguments and semantics as the
reference's radio/blocks/signal/*.lua).  All arithmetic runs in liblrhip.so on the GPU; this file is the
host glue the reference keeps in Lua (tap design, argument checks, rate bookkeeping).
"""
import math

import numpy as np

from . import _lib, filter_utils, types
from .block import Block, Input, Output, _fptr, as_taps, fir_mode


class FIRFilterBlock(Block):
    """radio/blocks/signal/firfilter.lua.  FIRFilterBlock(taps[, use_fft]).

    use_fft (block.fir_mode, the same table as the Lua glue's lrhip.fir_mode): True is the reference's overlap-save
    (firfilter.lua:320-398: only whole L = N-M+1 blocks are emitted, tail retained); "fast" runs the same overlap-save
    arithmetic (fused FFT kernel) but emits one output per input; None (the default, as `nil` in the reference, which then picks
    its FFT form when FFTW is present, firfilter.lua:57) and "auto" pick "fast" from 48 taps up and the direct form below; False
    is the direct form on the f32 matrix cores, bit-identical to the fmaf chain in the reference's tap order."""
    name = "FIRFilterBlock"

    def instantiate(self, taps, use_fft=None):
        self.taps = as_taps(taps)
        self.use_fft = fir_mode(use_fft)
        self.decimation = 1
        if self.taps.dtype == np.complex64:      # firfilter.lua:68-74
            self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        else:
            self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
            self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])

    def initialize(self):
        L = _lib.load()
        tc = self.taps.dtype == np.complex64
        flat = self.taps.view(np.float32) if tc else self.taps
        self._set_stage(L.lrhip_fir_create(_fptr(flat), len(self.taps), int(tc),
                                           int(self.get_input_type() is types.ComplexFloat32),
                                           self.decimation, int(self.use_fft)),
                        "Creating lrhip fir object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class LowpassFilterBlock(FIRFilterBlock):
    """radio/blocks/signal/lowpassfilter.lua:32-50. LowpassFilterBlock(num_taps, cutoff[, nyquist[, window]])."""
    name = "LowpassFilterBlock"

    def instantiate(self, num_taps, cutoff, nyquist=None, window=None):
        assert num_taps, "Missing argument #1 (num_taps)"
        assert cutoff is not None, "Missing argument #2 (cutoff)"
        self.cutoff = cutoff
        self.window = window or "hamming"
        self.nyquist = nyquist
        FIRFilterBlock.instantiate(self, types.Float32.vector(num_taps))

    def _design(self, nyquist):
        return filter_utils.firwin_lowpass(len(self.taps), self.cutoff / nyquist, self.window)

    def initialize(self):
        nyquist = self.nyquist or (self.get_rate() / 2)       # lowpassfilter.lua:43
        self.taps = types.Float32.vector_from_array(self._design(nyquist))
        FIRFilterBlock.initialize(self)


class HighpassFilterBlock(LowpassFilterBlock):
    """radio/blocks/signal/highpassfilter.lua"""
    name = "HighpassFilterBlock"

    def _design(self, nyquist):
        return filter_utils.firwin_highpass(len(self.taps), self.cutoff / nyquist, self.window)


class BandpassFilterBlock(LowpassFilterBlock):
    """radio/blocks/signal/bandpassfilter.lua: cutoff = {low, high}"""
    name = "BandpassFilterBlock"

    def _design(self, nyquist):
        return filter_utils.firwin_bandpass(len(self.taps), [self.cutoff[0] / nyquist, self.cutoff[1] / nyquist], self.window)


class BandstopFilterBlock(LowpassFilterBlock):
    """radio/blocks/signal/bandstopfilter.lua: cutoff = {low, high}"""
    name = "BandstopFilterBlock"

    def _design(self, nyquist):
        return filter_utils.firwin_bandstop(len(self.taps), [self.cutoff[0] / nyquist, self.cutoff[1] / nyquist], self.window)


class FrequencyTranslatorBlock(Block):
    """radio/blocks/signal/frequencytranslator.lua:26-30, :93-110. FrequencyTranslatorBlock(offset)."""
    name = "FrequencyTranslatorBlock"

    def instantiate(self, offset):
        assert offset is not None, "Missing argument #1 (offset)"
        self.offset = offset
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])

    def initialize(self):
        self.omega = 2 * math.pi * (self.offset / self.get_rate())     # frequencytranslator.lua:95
        self._set_stage(_lib.load().lrhip_rotator_create(self.omega), "Creating lrhip rotator object")

    def process(self, x):
        return self._execute(x, np.complex64)


class DownsamplerBlock(Block):
    """radio/blocks/signal/downsampler.lua:29-56. DownsamplerBlock(factor)."""
    name = "DownsamplerBlock"

    def instantiate(self, factor):
        assert factor, "Missing argument #1 (factor)"
        self.factor = int(factor)
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])

    def get_rate(self):
        return Block.get_rate(self) / self.factor          # downsampler.lua:36-38

    def initialize(self):
        self._set_stage(_lib.load().lrhip_downsampler_create(self.factor, self.get_input_type().size),
                        "Creating lrhip downsampler object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class FrequencyDiscriminatorBlock(Block):
    """radio/blocks/signal/frequencydiscriminator.lua:25-38. FrequencyDiscriminatorBlock(modulation_index)."""
    name = "FrequencyDiscriminatorBlock"

    def instantiate(self, modulation_index):
        assert modulation_index, "Missing argument #1 (modulation_index)"
        self.gain = 2 * math.pi * modulation_index          # frequencydiscriminator.lua:28
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.Float32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_fmdiscrim_create(self.gain), "Creating lrhip fmdiscrim object")

    def process(self, x):
        return self._execute(x, np.float32)


class IIRFilterBlock(Block):
    """radio/blocks/signal/iirfilter.lua:39-61. IIRFilterBlock(b_taps, a_taps)."""
    name = "IIRFilterBlock"

    def instantiate(self, b_taps, a_taps):
        assert b_taps is not None, "Missing argument #1 (b_taps)"
        assert a_taps is not None, "Missing argument #2 (a_taps)"
        self.b_taps = types.Float32.vector_from_array(b_taps)
        self.a_taps = types.Float32.vector_from_array(a_taps)
        assert len(self.a_taps) >= 1, "Feedback taps must be at least length 1"
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_iir_create(_fptr(self.b_taps), len(self.b_taps), _fptr(self.a_taps), len(self.a_taps),
                                                     int(self.get_input_type() is types.ComplexFloat32)),
                        "Creating lrhip iir object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class SinglepoleLowpassFilterBlock(IIRFilterBlock):
    """radio/blocks/signal/singlepolelowpassfilter.lua:27-67. SinglepoleLowpassFilterBlock(cutoff)."""
    name = "SinglepoleLowpassFilterBlock"

    def instantiate(self, cutoff):
        assert cutoff, "Missing argument #1 (cutoff)"
        self.cutoff = cutoff
        IIRFilterBlock.instantiate(self, types.Float32.vector(2), types.Float32.vector(2))

    def initialize(self):
        rate = self.get_rate()
        tau = 1 / (2 * math.pi * self.cutoff)                       # :57
        tau = 1 / (2 * rate * math.tan(1 / (2 * rate * tau)))       # :58 pre-warp
        self.b_taps = types.Float32.vector_from_array([1 / (1 + 2 * tau * rate), 1 / (1 + 2 * tau * rate)])
        self.a_taps = types.Float32.vector_from_array([1, (1 - 2 * tau * rate) / (1 + 2 * tau * rate)])
        IIRFilterBlock.initialize(self)


class FMDeemphasisFilterBlock(SinglepoleLowpassFilterBlock):
    """radio/blocks/signal/fmdeemphasisfilter.lua:24-27. FMDeemphasisFilterBlock(tau)."""
    name = "FMDeemphasisFilterBlock"

    def instantiate(self, tau):
        assert tau, "Missing argument #1 (tau)"
        SinglepoleLowpassFilterBlock.instantiate(self, 1 / (2 * math.pi * tau))


class _BinaryBlock(Block):
    """Two-input element-wise blocks (MultiplyBlock, MultiplyConjugateBlock, AddBlock, SubtractBlock)."""
    _op = "multiply"
    _complex_only = False

    def instantiate(self):
        self.add_type_signature([Input("in1", types.ComplexFloat32), Input("in2", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        if not self._complex_only:
            self.add_type_signature([Input("in1", types.Float32), Input("in2", types.Float32)], [Output("out", types.Float32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_binary_create(self._op.encode(), int(self.get_input_type() is types.ComplexFloat32)),
                        "Creating lrhip %s object" % self._op)

    def process(self, x, y):
        import ctypes as C
        L = _lib.load()
        x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
        dt = self.get_input_type().dtype
        if x.dtype != dt or y.dtype != dt or len(x) != len(y):
            raise TypeError("Block %s expects two %s vectors of equal length" % (self.name, self.get_input_type()))
        out = np.empty(len(x), dtype=self.get_output_type().dtype)
        n = L.lrhip_stage_execute2(self._stage, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), len(x),
                                   out.ctypes.data_as(C.c_void_p), len(out))
        _lib.check(n, "%s:process" % self.name)
        return out[:n]


class MultiplyBlock(_BinaryBlock):
    """radio/blocks/signal/multiply.lua"""
    name, _op = "MultiplyBlock", "multiply"


class MultiplyConjugateBlock(_BinaryBlock):
    """radio/blocks/signal/multiplyconjugate.lua"""
    name, _op, _complex_only = "MultiplyConjugateBlock", "multiplyconjugate", True


class AddBlock(_BinaryBlock):
    """radio/blocks/signal/add.lua"""
    name, _op = "AddBlock", "add"


class SubtractBlock(_BinaryBlock):
    """radio/blocks/signal/subtract.lua"""
    name, _op = "SubtractBlock", "subtract"


class ComplexBandpassFilterBlock(FIRFilterBlock):
    """radio/blocks/signal/complexbandpassfilter.lua. ComplexBandpassFilterBlock(num_taps, cutoffs[, nyquist[, window]])."""
    name = "ComplexBandpassFilterBlock"
    _design = staticmethod(filter_utils.firwin_complex_bandpass)

    def instantiate(self, num_taps, cutoffs, nyquist=None, window=None):
        assert num_taps, "Missing argument #1 (num_taps)"
        assert cutoffs is not None, "Missing argument #2 (cutoffs)"
        self.cutoffs, self.window, self.nyquist = cutoffs, window or "hamming", nyquist
        FIRFilterBlock.instantiate(self, types.ComplexFloat32.vector(num_taps))

    def initialize(self):
        nyquist = self.nyquist or (self.get_rate() / 2)
        taps = self._design(len(self.taps), [self.cutoffs[0] / nyquist, self.cutoffs[1] / nyquist], self.window)
        self.taps = types.ComplexFloat32.vector_from_array(taps)
        FIRFilterBlock.initialize(self)


class ComplexBandstopFilterBlock(ComplexBandpassFilterBlock):
    """radio/blocks/signal/complexbandstopfilter.lua"""
    name = "ComplexBandstopFilterBlock"
    _design = staticmethod(filter_utils.firwin_complex_bandstop)


class RootRaisedCosineFilterBlock(FIRFilterBlock):
    """radio/blocks/signal/rootraisedcosinefilter.lua. RootRaisedCosineFilterBlock(num_taps, beta, symbol_rate)."""
    name = "RootRaisedCosineFilterBlock"

    def instantiate(self, num_taps, beta, symbol_rate):
        assert num_taps, "Missing argument #1 (num_taps)"
        assert beta is not None, "Missing argument #2 (beta)"
        assert symbol_rate, "Missing argument #3 (symbol_rate)"
        self.beta, self.symbol_rate = beta, symbol_rate
        FIRFilterBlock.instantiate(self, types.Float32.vector(num_taps))

    def initialize(self):
        taps = filter_utils.fir_root_raised_cosine(len(self.taps), self.get_rate(), self.beta, 1 / self.symbol_rate)
        self.taps = types.Float32.vector_from_array(taps)
        FIRFilterBlock.initialize(self)


class MultiplyConstantBlock(Block):
    """radio/blocks/signal/multiplyconstant.lua. MultiplyConstantBlock(constant): number / Float32 or a complex constant."""
    name = "MultiplyConstantBlock"

    def instantiate(self, constant):
        assert constant is not None, "Missing argument #1 (constant)"
        if isinstance(constant, (complex, np.complexfloating)):
            self.constant = np.complex64(constant)
            self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        elif isinstance(constant, (int, float, np.floating, np.integer)):
            self.constant = np.float32(constant)
            self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])
            self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        else:
            raise TypeError("Unsupported constant type")

    def initialize(self):
        cc = isinstance(self.constant, np.complexfloating)
        self._set_stage(_lib.load().lrhip_multiply_constant_create(float(np.real(self.constant)), float(np.imag(self.constant)), int(cc),
                                                                   int(self.get_input_type() is types.ComplexFloat32)),
                        "Creating lrhip multiplyconstant object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class UpsamplerBlock(Block):
    """radio/blocks/signal/upsampler.lua. UpsamplerBlock(factor)."""
    name = "UpsamplerBlock"

    def instantiate(self, factor):
        assert factor, "Missing argument #1 (factor)"
        self.factor = int(factor)
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])

    def get_rate(self):
        return Block.get_rate(self) * self.factor          # upsampler.lua:41-43

    def initialize(self):
        self._set_stage(_lib.load().lrhip_upsampler_create(self.factor, self.get_input_type().size), "Creating lrhip upsampler object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class PolyphaseChannelizerBlock(Block):
    """Critically sampled K-channel analysis filterbank (BASELINE.json configs[4]).  Not a block of the reference: it is
    K parallel chains FrequencyTranslatorBlock(-c*rate/K) -> FIRFilterBlock(taps) -> DownsamplerBlock(K), evaluated as
    one dense GEMM on the f32 matrix cores.  PolyphaseChannelizerBlock(num_channels[, taps]); default prototype =
    firwin_lowpass(16*K, 1/K).  Output: frames of K ComplexFloat32 (channel c at position c), one per K inputs."""
    name = "PolyphaseChannelizerBlock"

    def instantiate(self, num_channels, taps=None):
        assert num_channels, "Missing argument #1 (num_channels)"
        self.num_channels = int(num_channels)
        if taps is None:
            taps = filter_utils.firwin_lowpass(16 * self.num_channels, 1.0 / self.num_channels)
        self.taps = types.Float32.vector_from_array(taps)
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])

    def get_rate(self):
        return Block.get_rate(self)      # K values per K input samples; each channel runs at rate/K

    def initialize(self):
        self._set_stage(_lib.load().lrhip_channelizer_create(_fptr(self.taps), len(self.taps), self.num_channels),
                        "Creating lrhip channelizer object")

    def process(self, x):
        """returns an array of shape (frames, K)"""
        return self._execute(x, np.complex64).reshape(-1, self.num_channels)


class _UnaryBlock(Block):
    """One-input element-wise blocks with fixed types."""
    _op, _in, _out = "", types.ComplexFloat32, types.Float32

    def instantiate(self):
        self.add_type_signature([Input("in", self._in)], [Output("out", self._out)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_unary_create(self._op.encode(), 0.0, 0.0, 0, int(self._in is types.ComplexFloat32)),
                        "Creating lrhip %s object" % self._op)

    def process(self, x):
        return self._execute(x, self._out.dtype)


class ComplexMagnitudeBlock(_UnaryBlock):
    """radio/blocks/signal/complexmagnitude.lua"""
    name, _op = "ComplexMagnitudeBlock", "complexmagnitude"


class ComplexPhaseBlock(_UnaryBlock):
    """radio/blocks/signal/complexphase.lua"""
    name, _op = "ComplexPhaseBlock", "complexphase"


class ComplexToRealBlock(_UnaryBlock):
    """radio/blocks/signal/complextoreal.lua"""
    name, _op = "ComplexToRealBlock", "complextoreal"


class ComplexToImagBlock(_UnaryBlock):
    """radio/blocks/signal/complextoimag.lua"""
    name, _op = "ComplexToImagBlock", "complextoimag"


class ComplexConjugateBlock(_UnaryBlock):
    """radio/blocks/signal/complexconjugate.lua"""
    name, _op, _out = "ComplexConjugateBlock", "complexconjugate", types.ComplexFloat32


class RealToComplexBlock(_UnaryBlock):
    """radio/blocks/signal/realtocomplex.lua"""
    name, _op, _in, _out = "RealToComplexBlock", "realtocomplex", types.Float32, types.ComplexFloat32


class AbsoluteValueBlock(_UnaryBlock):
    """radio/blocks/signal/absolutevalue.lua"""
    name, _op, _in, _out = "AbsoluteValueBlock", "absolutevalue", types.Float32, types.Float32


class AddConstantBlock(MultiplyConstantBlock):
    """radio/blocks/signal/addconstant.lua. AddConstantBlock(constant)."""
    name = "AddConstantBlock"

    def initialize(self):
        cc = isinstance(self.constant, np.complexfloating)
        self._set_stage(_lib.load().lrhip_unary_create(b"addconstant", float(np.real(self.constant)), float(np.imag(self.constant)), int(cc),
                                                       int(self.get_input_type() is types.ComplexFloat32)),
                        "Creating lrhip addconstant object")


class DelayBlock(Block):
    """radio/blocks/signal/delay.lua. DelayBlock(num_samples) (ComplexFloat32 / Float32 signatures)."""
    name = "DelayBlock"

    def instantiate(self, num_samples):
        assert num_samples is not None, "Missing argument #1 (num_samples)"
        assert num_samples > 0, "Number of samples must be greater than 0"
        self.num_samples = int(num_samples)
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_delay_create(self.num_samples, self.get_input_type().size), "Creating lrhip delay object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class HilbertTransformBlock(Block):
    """radio/blocks/signal/hilberttransform.lua. HilbertTransformBlock(num_taps[, window]): Float32 -> ComplexFloat32."""
    name = "HilbertTransformBlock"

    def instantiate(self, num_taps, window=None):
        assert num_taps, "Missing argument #1 (num_taps)"
        assert (num_taps % 2) == 1, "Number of taps must be odd"
        self.hilbert_taps = types.Float32.vector_from_array(filter_utils.fir_hilbert_transform(num_taps, window or "hamming"))
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.ComplexFloat32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_hilbert_create(_fptr(self.hilbert_taps), len(self.hilbert_taps)), "Creating lrhip hilbert object")

    def process(self, x):
        return self._execute(x, np.complex64)


class SinglepoleHighpassFilterBlock(IIRFilterBlock):
    """radio/blocks/signal/singlepolehighpassfilter.lua:27-48. SinglepoleHighpassFilterBlock(cutoff)."""
    name = "SinglepoleHighpassFilterBlock"

    def instantiate(self, cutoff):
        assert cutoff is not None, "Missing argument #1 (cutoff)"
        self.cutoff = cutoff
        super().instantiate(types.Float32.vector(2), types.Float32.vector(2))

    def initialize(self):
        rate = self.get_rate()
        tau = 1 / (2 * math.pi * self.cutoff)                       # :36-37 warped time constant
        tau = 1 / (2 * rate * math.tan(1 / (2 * rate * tau)))
        self.b_taps[0] = (2 * tau * rate) / (1 + 2 * tau * rate)
        self.b_taps[1] = -(2 * tau * rate) / (1 + 2 * tau * rate)
        self.a_taps[0] = 1
        self.a_taps[1] = (1 - 2 * tau * rate) / (1 + 2 * tau * rate)
        super().initialize()


class FMPreemphasisFilterBlock(SinglepoleHighpassFilterBlock):
    """radio/blocks/signal/fmpreemphasisfilter.lua:30-33. FMPreemphasisFilterBlock(tau)."""
    name = "FMPreemphasisFilterBlock"

    def instantiate(self, tau):
        assert tau is not None, "Missing argument #1 (tau)"
        super().instantiate(1 / (2 * math.pi * tau))


class FloatToComplexBlock(_BinaryBlock):
    """radio/blocks/signal/floattocomplex.lua: (real, imag) Float32 inputs -> ComplexFloat32."""
    name, _op = "FloatToComplexBlock", "floattocomplex"

    def instantiate(self):
        self.add_type_signature([Input("real", types.Float32), Input("imag", types.Float32)], [Output("out", types.ComplexFloat32)])


class ComplexToFloatBlock(Block):
    """radio/blocks/signal/complextofloat.lua: ComplexFloat32 -> (real, imag) Float32 outputs (two device passes)."""
    name = "ComplexToFloatBlock"

    def instantiate(self):
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("real", types.Float32), Output("imag", types.Float32)])

    def initialize(self):
        self._real, self._imag = ComplexToRealBlock(), ComplexToImagBlock()
        self._sub_blocks = [self._real, self._imag]       # DeviceGraph: one device pass per output
        for b in (self._real, self._imag):
            b.differentiate([types.ComplexFloat32])
            b.initialize()

    def process(self, x):
        return self._real.process(x), self._imag.process(x)


class FrequencyModulatorBlock(Block):
    """radio/blocks/signal/frequencymodulator.lua. FrequencyModulatorBlock(modulation_index): Float32 -> ComplexFloat32."""
    name = "FrequencyModulatorBlock"

    def instantiate(self, modulation_index):
        assert modulation_index is not None, "Missing argument #1 (modulation_index)"
        self.modulation_index = modulation_index
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.ComplexFloat32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_fmmod_create(float(self.modulation_index)), "Creating lrhip fmmod object")

    def process(self, x):
        return self._execute(x, np.complex64)


class PulseMatchedFilterBlock(FIRFilterBlock):
    """radio/blocks/signal/pulsematchedfilter.lua:27-47. PulseMatchedFilterBlock(baudrate[, invert=false])."""
    name = "PulseMatchedFilterBlock"
    _pattern = (1,)

    def instantiate(self, baudrate, invert=False):
        assert baudrate is not None, "Missing argument #1 (baudrate)"
        self.baudrate, self.invert = baudrate, bool(invert)
        FIRFilterBlock.instantiate(self, types.Float32.vector(32))

    def initialize(self):
        symbol_period = self.get_rate() / self.baudrate
        count = int(math.floor(symbol_period))            # Lua: for i = 1, symbol_period
        sign = -1.0 if self.invert else 1.0
        taps = []
        for half in self._pattern:
            taps.extend([sign * half] * count)
        self.taps = types.Float32.vector_from_array(taps)
        FIRFilterBlock.initialize(self)


class ManchesterMatchedFilterBlock(PulseMatchedFilterBlock):
    """radio/blocks/signal/manchestermatchedfilter.lua:27-50: a -1 half symbol followed by a +1 half symbol."""
    name = "ManchesterMatchedFilterBlock"
    _pattern = (-1, 1)


class AGCBlock(Block):
    """radio/blocks/signal/agc.lua:25-96. AGCBlock(mode[, target=-35[, threshold=-75[, {gain_tau=, power_tau=}]]])."""
    name = "AGCBlock"

    def instantiate(self, mode, target=None, threshold=None, options=None):
        assert mode, 'Missing argument #1 (mode), can be "fast", "slow", or "custom"'
        assert mode in ("fast", "slow", "custom"), 'Invalid mode "%s"' % mode
        options = options or {}
        self.mode = mode
        self.target = -35 if target is None else target
        self.threshold = -75 if threshold is None else threshold
        self.gain_tau = {"fast": 0.1, "slow": 3.0}.get(mode, options.get("gain_tau"))
        self.power_tau = options.get("power_tau") or 1.0
        assert self.gain_tau, 'Missing gain_tau parameter for "custom" mode'
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])

    def initialize(self):
        rate = self.get_rate()
        self._set_stage(_lib.load().lrhip_agc_create(1 / (1 + self.power_tau * rate), 1 / (1 + self.gain_tau * rate), 10 ** (self.target / 10),
                                                     10 ** (self.threshold / 10), int(self.get_input_type() is types.ComplexFloat32)),
                        "Creating lrhip agc object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)


class PowerSquelchBlock(Block):
    """radio/blocks/signal/powersquelch.lua:26-80. PowerSquelchBlock(threshold_dBFS[, tau=0.001])."""
    name = "PowerSquelchBlock"

    def instantiate(self, threshold, tau=None):
        assert threshold is not None, "Missing argument #1 (threshold)"
        self.threshold = threshold
        self.tau = 0.001        # powersquelch.lua:28: `self.tau = tau or 0.001` reads an undefined global, so it is always 0.001
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])

    def initialize(self):
        self._set_stage(_lib.load().lrhip_powersquelch_create(1 / (1 + self.tau * self.get_rate()), 10 ** (self.threshold / 10),
                                                              int(self.get_input_type() is types.ComplexFloat32)),
                        "Creating lrhip powersquelch object")

    def process(self, x):
        return self._execute(x, self.get_output_type().dtype)
