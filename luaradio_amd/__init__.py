"""luaradio_amd - MI355X-native DSP block engine behind LuaRadio's block API (hot path only).

Python mirror of the reference's Lua host side (radio.block / radio.types / blocks / composites) over the
C ABI of liblrhip.so (include/lrhip.h).  The Lua glue a LuaRadio checkout would use is under lua/.
"""
from . import _lib, filter_utils, spectrum_utils, types, window_utils  # noqa: F401
from ._lib import LrhipError, adopt_torch_stream, init  # noqa: F401
from .block import Block, Input, Output  # noqa: F401
from .blocks import (BandpassFilterBlock, BandstopFilterBlock, DownsamplerBlock, FIRFilterBlock,  # noqa: F401
                     FMDeemphasisFilterBlock, FrequencyDiscriminatorBlock, FrequencyTranslatorBlock,
                     HighpassFilterBlock, IIRFilterBlock, LowpassFilterBlock, SinglepoleLowpassFilterBlock,
                     MultiplyBlock, MultiplyConjugateBlock, AddBlock, SubtractBlock, ComplexBandpassFilterBlock,
                     ComplexBandstopFilterBlock, RootRaisedCosineFilterBlock, MultiplyConstantBlock, UpsamplerBlock, PolyphaseChannelizerBlock, ComplexMagnitudeBlock,
                     ComplexPhaseBlock, ComplexToRealBlock, ComplexToImagBlock, ComplexConjugateBlock, RealToComplexBlock,
                     AbsoluteValueBlock, AddConstantBlock, DelayBlock, HilbertTransformBlock, SinglepoleHighpassFilterBlock,
                     FMPreemphasisFilterBlock, FloatToComplexBlock, ComplexToFloatBlock, FrequencyModulatorBlock,
                     PulseMatchedFilterBlock, ManchesterMatchedFilterBlock, AGCBlock, PowerSquelchBlock)
from .sources import IQFileSource, RealFileSource, IQFileSink, RealFileSink  # noqa: F401
from .meters import BenchmarkSink, RawFileSource, ZeroSource  # noqa: F401
from . import ipc, meters, procfanout, timeshard  # noqa: F401
from .graph import DeviceGraph  # noqa: F401
from .composites import (Chain, CompositeBlock, DecimatorBlock, InterpolatorBlock, RationalResamplerBlock, TunerBlock, WBFMMonoDemodulator,  # noqa: F401
                         NBFMDemodulator, AMEnvelopeDemodulator, SSBDemodulator, SSBModulator, wbfm_mono_receiver, am_envelope_receiver,
                         ssb_receiver, nbfm_receiver)

version = "0.1.0"
