"""Composite blocks and the linear-run collapse (host side).

The reference's CompositeBlock (radio/core/composite.lua) forks one process per block and moves every
vector over a UNIX socket per edge (radio/core/pipe.lua:53-69).  Here a maximal linear run of
device-capable blocks is collapsed into ONE `lrhip_chain_t` (include/lrhip.h): one H2D at the head, one
D2H at the tail, device-resident edges in between, and adjacent rotator / FIR / downsampler blocks fused
into a single decimating kernel.  Mirrors connect() (:111), rate propagation (:394), differentiate (:314)
and initialize (:416-424) for linear graphs, which is all the hot path's composites need.
"""
import ctypes as C

import numpy as np

from . import _lib, types
from .block import Block, Input, Output, fir_mode
from . import blocks as B


class Chain:
    """A linear run of initialized device blocks executed as one lrhip_chain_t."""

    def __init__(self, blocks, exact=False):
        """exact: the chain's numerical contract, as DeviceChainBlock.exact in lua/radio/composites/devicechain.lua - False = fused
        kernels with the stated roundings of include/lrhip.h, True = _lib.CHAIN_EXACT (what the blocks compute one by one), or a
        number of lrhip_chain_create_ex flags (_lib.CHAIN_*)."""
        self.blocks = list(blocks)
        self.flags = _lib.CHAIN_EXACT if exact is True else int(exact or 0)
        self._ring_out = None
        L = _lib.load()
        arr = (C.c_void_p * len(self.blocks))(*[b.stage_handle() for b in self.blocks])
        self._chain = _lib.check_ptr(L.lrhip_chain_create_ex(arr, len(self.blocks), self.flags), "Creating lrhip chain object")
        # a chain may start with a file source's format stage: its input is then RAW file records (uint8, record_size bytes each)
        self.in_record = getattr(self.blocks[0], "record_size", None)
        self.in_type = None if self.in_record else self.blocks[0].get_input_type()
        self.out_type = self.blocks[-1].get_output_type()

    def __del__(self):
        try:
            if self._chain:
                _lib.load().lrhip_chain_destroy(self._chain)
                self._chain = None
        except Exception:
            pass

    def get_output_type(self):
        """the last block's output type (what a block answers: a Chain can stand where a block stands, e.g. as a fan-out branch)"""
        return self.out_type

    def max_output(self, n_in):
        return _lib.load().lrhip_chain_max_output(self._chain, n_in)

    def reset(self):
        _lib.check(_lib.load().lrhip_chain_reset(self._chain), "chain:reset")

    def seek(self, n0):
        """continue as if n0 input samples of the stream had been consumed (zero histories, absolute phases): lrhip_chain_seek"""
        _lib.check(_lib.load().lrhip_chain_seek(self._chain, int(n0)), "chain:seek")

    def halo(self):
        """input samples a time partition replays in front of its first own sample (lrhip_chain_halo); raises if the chain
        holds a stage with unbounded memory"""
        return _lib.check(_lib.load().lrhip_chain_halo(self._chain), "chain:halo")

    def shard_align(self):
        """partition boundaries on multiples of this many input samples reproduce the single-stream run bit for bit"""
        return int(_lib.load().lrhip_chain_shard_align(self._chain))

    def start_at(self, first_sample):
        """lrhip_chain_start_at: seek to the aligned sample s <= first_sample - halo (returned: feed the stream from s on); the library
        drops the output of the replayed samples in front of first_sample (DeviceChainBlock:start_at in the Lua glue)"""
        s = C.c_ulonglong(0)
        _lib.check(_lib.load().lrhip_chain_start_at(self._chain, int(first_sample), C.byref(s)), "chain:start_at")
        return int(s.value)

    def set_latency(self, max_seconds):
        """live sources: a pushed sample waits at most this long (wall clock) for its batch to be launched (lrhip_chain_set_latency)"""
        _lib.check(_lib.load().lrhip_chain_set_latency(self._chain, float(max_seconds)), "chain:set_latency")

    def _count(self, x):
        """input vector -> (contiguous array, number of input samples)"""
        x = np.ascontiguousarray(x)
        if self.in_record:
            if x.dtype != np.uint8 or len(x) % self.in_record:
                raise TypeError("chain expects raw file records (uint8, %d bytes each)" % self.in_record)
            return x, len(x) // self.in_record
        if x.dtype != self.in_type.dtype:
            raise TypeError("chain expects %s input, got %s" % (self.in_type, x.dtype))
        return x, len(x)

    def process(self, x):
        L = _lib.load()
        x, count = self._count(x)
        cap = L.lrhip_chain_max_output(self._chain, count)
        out = np.empty(cap, dtype=self.out_type.dtype)
        n = L.lrhip_chain_execute(self._chain, x.ctypes.data_as(C.c_void_p), count, out.ctypes.data_as(C.c_void_p), cap)
        _lib.check(n, "chain:process")
        return out[:n]

    def process_device(self, in_ptr, n_in, out_ptr, out_capacity):
        n = _lib.load().lrhip_chain_execute_device(self._chain, in_ptr, n_in, out_ptr, out_capacity)
        return _lib.check(n, "chain:process_device")

    @property
    def last_launches(self):
        return _lib.load().lrhip_chain_last_launches(self._chain)

    # ---- pipelined host path (ring of pinned + device slots; H2D / kernels / D2H of neighbouring chunks overlap)
    def set_ring(self, depth, max_chunk):
        L = _lib.load()
        _lib.check(L.lrhip_chain_set_ring(self._chain, depth, max_chunk), "chain:set_ring")
        self._ring_chunk, self._ring_depth = int(max_chunk), int(depth)
        # sized by the library's own bound (interpolating chains and block-framed FIRs emit more than they take in)
        self._ring_out = np.empty(L.lrhip_chain_push_bound(self._chain, max_chunk), dtype=self.out_type.dtype)

    def submit(self, x):
        x, count = self._count(x)
        return _lib.check(_lib.load().lrhip_chain_submit(self._chain, x.ctypes.data_as(C.c_void_p), count), "chain:submit")

    def submit_fd(self, fd, offset, max_samples=None):
        """File-fed chain (lrhip_chain_submit_fd): up to max_samples raw records of the regular file `fd`, from byte offset `offset`, are read by the library
        itself - positional reads on its copy threads, straight into the pinned input of the next ring slot - and the slot is submitted.  Returns the number
        of samples submitted; 0 at the end of the file."""
        n = _lib.load().lrhip_chain_submit_fd(self._chain, int(fd), int(offset), int(max_samples or self._ring_chunk))
        return _lib.check(n, "chain:submit_fd")

    def ring_input(self):
        """numpy view (uint8 for raw-record chains, else the input dtype) of the pinned buffer the next submit() will use:
        fill it in place (file.readinto(view), socket.recv_into(view)) and submit(view[:n]) - no staging copy.  None when the
        ring is full."""
        p = _lib.load().lrhip_chain_ring_input(self._chain)
        if not p:
            return None
        dt = np.dtype(np.uint8) if self.in_record else self.in_type.dtype
        nbytes = self._ring_chunk * (self.in_record or dt.itemsize)
        return np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=dt)

    def collect(self):
        n = _lib.load().lrhip_chain_collect(self._chain, self._ring_out.ctypes.data_as(C.c_void_p), len(self._ring_out))
        _lib.check(n, "chain:collect")
        return self._ring_out[:n].copy()

    @property
    def in_flight(self):
        return _lib.load().lrhip_chain_in_flight(self._chain)

    def stream(self, chunks):
        """Run an iterable of input vectors through the ring (one slot per vector), yielding the outputs in order."""
        for x in chunks:
            if self.in_flight == self._ring_depth:
                yield self.collect()
            self.submit(x)
        while self.in_flight:
            yield self.collect()

    # ---- chunk coalescing: small process() vectors accumulate in the ring slot until it holds max_chunk samples
    def push(self, x):
        """Append one input vector; returns the outputs of the batches that have finished (possibly empty)."""
        L = _lib.load()
        x, count = self._count(x)
        need = L.lrhip_chain_push_bound(self._chain, count)
        if self._ring_out is None or len(self._ring_out) < need:      # no ring: need == 0 and the library reports "chain has no ring"
            self._ring_out = np.empty(need, dtype=self.out_type.dtype)
        n = L.lrhip_chain_push(self._chain, x.ctypes.data_as(C.c_void_p), count, self._ring_out.ctypes.data_as(C.c_void_p), len(self._ring_out))
        _lib.check(n, "chain:push")
        return self._ring_out[:n].copy()

    def poll(self):
        """The host's wait for input timed out (lrhip_chain_poll): a partial batch whose oldest sample has waited out the latency bound is
        launched and returned, finished batches are handed out; empty when nothing is due."""
        L = _lib.load()
        need = max(1, L.lrhip_chain_push_bound(self._chain, 0))
        if self._ring_out is None or len(self._ring_out) < need:
            self._ring_out = np.empty(need, dtype=self.out_type.dtype)
        n = L.lrhip_chain_poll(self._chain, self._ring_out.ctypes.data_as(C.c_void_p), len(self._ring_out))
        _lib.check(n, "chain:poll")
        return self._ring_out[:n].copy()

    def poll_due(self):
        """seconds the host may wait for input before calling poll() (lrhip_chain_poll_due): -1 = forever (nothing pending / no latency bound)"""
        return float(_lib.load().lrhip_chain_poll_due(self._chain))

    def flush(self):
        """Launch the partly filled batch, wait, return everything still pending (EOF / cleanup)."""
        L = _lib.load()
        if self._ring_out is None:
            self._ring_out = np.empty(max(1, L.lrhip_chain_push_bound(self._chain, 0)), dtype=self.out_type.dtype)
        n = L.lrhip_chain_flush(self._chain, self._ring_out.ctypes.data_as(C.c_void_p), len(self._ring_out))
        _lib.check(n, "chain:flush")
        return self._ring_out[:n].copy()


class CompositeBlock(Block):
    """Linear composite: connect(b1, b2, ...) then differentiate / initialize / process like a block.

    Rates propagate downstream through get_rate() overrides exactly as in the reference
    (radio/core/block.lua:383-390; DownsamplerBlock divides, downsampler.lua:36-38)."""
    name = "CompositeBlock"

    def instantiate(self):
        self._blocks = []
        self._chain = None

    def connect(self, *blocks):
        for b in blocks:
            if isinstance(b, CompositeBlock) and b is not self:
                self._blocks.extend(b._blocks)
            elif b is not self:
                self._blocks.append(b)
        return self

    def differentiate(self, input_types):
        # block.lua:238-352: a composite that declares signatures accepts only those input types
        if self.type_signatures and not any(len(ins) == len(input_types) and all(i.data_type is t for i, t in zip(ins, input_types))
                                            for ins, _, _ in self.type_signatures):
            raise TypeError("No compatible type signatures found for block %s with input types [%s]."
                            % (self.name, ", ".join(map(str, input_types))))
        t = list(input_types)
        for b in self._blocks:
            b.differentiate(t)
            t = [b.get_output_type()]
        self.signature = ([Input("in", input_types[0])], [Output("out", t[0])], None)

    def get_rate(self):
        return self._blocks[-1].get_rate() if self._blocks else Block.get_rate(self)

    def _propagate_rates(self):
        rate = Block.get_rate(self)
        for b in self._blocks:
            b.rate = rate
            rate = b.get_rate()

    def initialize(self):
        self._propagate_rates()
        for b in self._blocks:
            b.initialize()
        self._chain = Chain(self._blocks, getattr(self, "exact", False))     # composite.exact = True: lrhip_chain_create_ex(LRHIP_CHAIN_EXACT)

    def process(self, x):
        return self._chain.process(x)

    def process_device(self, in_ptr, n_in, out_ptr, out_capacity):
        return self._chain.process_device(in_ptr, n_in, out_ptr, out_capacity)

    def max_output(self, n_in):
        return self._chain.max_output(n_in)

    def reset(self):
        self._chain.reset()

    def seek(self, n0):
        self._chain.seek(n0)

    def halo(self):
        return self._chain.halo()

    def shard_align(self):
        return self._chain.shard_align()

    @property
    def chain(self):
        return self._chain


def _fft_option(options):
    """options["use_fft"] of the decimating composites, through the ONE mapping every front end shares (block.fir_mode, the same table
    as lrhip.fir_mode in lua/radio/core/lrhip.lua): None / "auto" = 3 (the library picks: direct form on the matrix cores for a
    decimating filter, which is the faster one on MI355X and bit-exact), False = 0 direct form, "fast" = 2 overlap-save arithmetic
    (fused with the downsampler: the polyphase FFT kernel, kernels_firdecfft.h; direct form where no such kernel exists),
    True = 1 the reference's block-emission framing (firfilter.lua:361-398; the filter then runs unfused in front of the downsampler)."""
    return fir_mode(options.get("use_fft"))


class DecimatorBlock(CompositeBlock):
    """radio/composites/decimator.lua:28-42. DecimatorBlock(decimation[, {num_taps=, window=}])."""
    name = "DecimatorBlock"

    def instantiate(self, decimation, options=None):
        CompositeBlock.instantiate(self)
        assert decimation, "Missing argument #1 (decimation)"
        options = options or {}
        filt = B.LowpassFilterBlock(options.get("num_taps") or 128, 1 / decimation, 1.0, options.get("window"))
        filt.use_fft = _fft_option(options)
        downsampler = B.DownsamplerBlock(decimation)
        self.connect(filt, downsampler)
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])


class InterpolatorBlock(CompositeBlock):
    """radio/composites/interpolator.lua:24-42. InterpolatorBlock(interpolation[, {num_taps=, window=}])."""
    name = "InterpolatorBlock"

    def instantiate(self, interpolation, options=None):
        CompositeBlock.instantiate(self)
        assert interpolation, "Missing argument #1 (interpolation)"
        options = options or {}
        self.connect(B.MultiplyConstantBlock(interpolation), B.UpsamplerBlock(interpolation),
                     B.LowpassFilterBlock(options.get("num_taps") or 128, 1 / interpolation, 1.0, options.get("window")))
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])


class RationalResamplerBlock(CompositeBlock):
    """radio/composites/rationalresampler.lua:25-49. RationalResamplerBlock(interpolation, decimation[, options])."""
    name = "RationalResamplerBlock"

    def instantiate(self, interpolation, decimation, options=None):
        CompositeBlock.instantiate(self)
        assert interpolation, "Missing argument #1 (interpolation)"
        assert decimation, "Missing argument #2 (decimation)"
        options = options or {}
        cutoff = 1 / interpolation if (1 / interpolation < 1 / decimation) else 1 / decimation
        self.connect(B.MultiplyConstantBlock(interpolation), B.UpsamplerBlock(interpolation),
                     B.LowpassFilterBlock(options.get("num_taps") or 128, cutoff, 1.0, options.get("window")),
                     B.DownsamplerBlock(decimation))
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.Float32)])


class TunerBlock(CompositeBlock):
    """radio/composites/tuner.lua:32-48. TunerBlock(offset, bandwidth, decimation[, options])."""
    name = "TunerBlock"

    def instantiate(self, offset, bandwidth, decimation, options=None):
        CompositeBlock.instantiate(self)
        assert offset is not None, "Missing argument #1 (offset)"
        assert bandwidth, "Missing argument #2 (bandwidth)"
        assert decimation, "Missing argument #3 (decimation)"
        options = options or {}
        translator = B.FrequencyTranslatorBlock(offset)
        filt = B.LowpassFilterBlock(options.get("num_taps") or 128, bandwidth / 2, None, options.get("window"))
        filt.use_fft = _fft_option(options)
        downsampler = B.DownsamplerBlock(decimation)
        self.connect(translator, filt, downsampler)
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.ComplexFloat32)])


class WBFMMonoDemodulator(CompositeBlock):
    """radio/composites/wbfmmonodemodulator.lua:22-36. WBFMMonoDemodulator([tau])."""
    name = "WBFMMonoDemodulator"

    def instantiate(self, tau=None):
        CompositeBlock.instantiate(self)
        tau = tau or 75e-6
        bandwidth = 15e3
        self.connect(B.FrequencyDiscriminatorBlock(1.25), B.LowpassFilterBlock(128, bandwidth), B.FMDeemphasisFilterBlock(tau))
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.Float32)])


class NBFMDemodulator(CompositeBlock):
    """radio/composites/nbfmdemodulator.lua:25-41. NBFMDemodulator([deviation=5e3[, bandwidth=4e3]])."""
    name = "NBFMDemodulator"

    def instantiate(self, deviation=None, bandwidth=None):
        CompositeBlock.instantiate(self)
        deviation = deviation or 5e3
        bandwidth = bandwidth or 4e3
        self.connect(B.LowpassFilterBlock(128, 2 * (deviation + bandwidth) / 2), B.FrequencyDiscriminatorBlock(deviation / bandwidth),
                     B.LowpassFilterBlock(128, bandwidth))
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.Float32)])


class AMEnvelopeDemodulator(CompositeBlock):
    """radio/composites/amenvelopedemodulator.lua:24-38. AMEnvelopeDemodulator([bandwidth=5e3])."""
    name = "AMEnvelopeDemodulator"

    def instantiate(self, bandwidth=None):
        CompositeBlock.instantiate(self)
        bandwidth = bandwidth or 5e3
        self.connect(B.ComplexMagnitudeBlock(), B.SinglepoleHighpassFilterBlock(100), B.LowpassFilterBlock(128, bandwidth))
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.Float32)])


class SSBDemodulator(CompositeBlock):
    """radio/composites/ssbdemodulator.lua:25-43. SSBDemodulator(sideband[, bandwidth=3e3])."""
    name = "SSBDemodulator"

    def instantiate(self, sideband, bandwidth=None):
        CompositeBlock.instantiate(self)
        assert sideband, "Missing argument #1 (sideband)"
        assert sideband in ("lsb", "usb"), "Sideband should be 'lsb' or 'usb'"
        bandwidth = bandwidth or 3e3
        self.connect(B.ComplexBandpassFilterBlock(129, [0, -bandwidth] if sideband == "lsb" else [0, bandwidth]),
                     B.ComplexToRealBlock(), B.LowpassFilterBlock(128, bandwidth))
        self.add_type_signature([Input("in", types.ComplexFloat32)], [Output("out", types.Float32)])


class SSBModulator(CompositeBlock):
    """radio/composites/ssbmodulator.lua:25-50. SSBModulator(sideband[, bandwidth=3e3]): Float32 audio -> ComplexFloat32."""
    name = "SSBModulator"

    def instantiate(self, sideband, bandwidth=None):
        CompositeBlock.instantiate(self)
        assert sideband, "Missing argument #1 (sideband)"
        assert sideband in ("lsb", "usb"), "Sideband should be 'lsb' or 'usb'"
        bandwidth = bandwidth or 3e3
        af_filter = B.LowpassFilterBlock(128, bandwidth)
        hilbert = B.HilbertTransformBlock(129)
        sb_filter = B.ComplexBandpassFilterBlock(129, [-bandwidth, 0] if sideband == "lsb" else [0, bandwidth])
        if sideband == "lsb":
            self.connect(af_filter, hilbert, B.ComplexConjugateBlock(), sb_filter)
        else:
            self.connect(af_filter, hilbert, sb_filter)
        self.add_type_signature([Input("in", types.Float32)], [Output("out", types.ComplexFloat32)])


def wbfm_mono_receiver(rate=1102500.0, tune_offset=-250e3, use_fft=False):
    """The compute blocks of examples/rtlsdr_wbfm_mono.lua:12-17,28 as one composite:
    Tuner(-250e3, 200e3, 5) -> FrequencyDiscriminator(1.25) -> Lowpass(128, 15e3) -> FMDeemphasis(75e-6) -> Downsampler(5).
    use_fft selects the tuner's arithmetic: False = direct form on the f32 matrix cores (bit-exact, and the faster one on MI355X:
    0.218 ms against 0.246 ms per 2^26 samples for the whole chain, same box, tools/ab_chain.py), "fast" = polyphase FFT overlap-save."""
    top = CompositeBlock()
    af_filter = B.LowpassFilterBlock(128, 15e3)
    # automatic: inside a chain the audio filter, the de-emphasis recurrence and the final downsampler become ONE launch on the
    # register-window kernel (direct form, bit-exact fmaf chains); alone it would take the overlap-save arithmetic
    af_filter.use_fft = 3
    top.connect(TunerBlock(tune_offset, 200e3, 5, {"use_fft": use_fft}), B.FrequencyDiscriminatorBlock(1.25), af_filter,
                B.FMDeemphasisFilterBlock(75e-6), B.DownsamplerBlock(5))
    top.rate = rate
    top.differentiate([types.ComplexFloat32])
    top.initialize()
    return top


def _receiver(blocks, rate, in_type=types.ComplexFloat32):
    top = CompositeBlock()
    top.connect(*blocks)
    top.rate = rate
    top.differentiate([in_type])
    top.initialize()
    return top


def am_envelope_receiver(rate=1102500.0, tune_offset=-100e3, bandwidth=5e3):
    """The compute blocks of examples/rtlsdr_am_envelope.lua:11-20,28 as one device chain:
    Tuner(offset, 2*bw, 50) -> ComplexMagnitude -> SinglepoleHighpass(100) -> Lowpass(128, bw) -> AGC('slow')."""
    return _receiver([TunerBlock(tune_offset, 2 * bandwidth, 50), B.ComplexMagnitudeBlock(), B.SinglepoleHighpassFilterBlock(100),
                      B.LowpassFilterBlock(128, bandwidth), B.AGCBlock("slow")], rate)


def ssb_receiver(sideband="usb", rate=1102500.0, tune_offset=-100e3, bandwidth=3e3):
    """The compute blocks of examples/rtlsdr_ssb.lua:13-24,34 as one device chain:
    Tuner(offset, 2*bw, 50) -> ComplexBandpass(129, {0, +-bw}) -> ComplexToReal -> Lowpass(128, bw) -> AGC('fast')."""
    assert sideband in ("lsb", "usb"), "Sideband should be 'lsb' or 'usb'."
    return _receiver([TunerBlock(tune_offset, 2 * bandwidth, 50),
                      B.ComplexBandpassFilterBlock(129, [0, -bandwidth] if sideband == "lsb" else [0, bandwidth]), B.ComplexToRealBlock(),
                      B.LowpassFilterBlock(128, bandwidth), B.AGCBlock("fast")], rate)


def nbfm_receiver(rate=1102500.0, tune_offset=-100e3, deviation=5e3, bandwidth=4e3):
    """The compute blocks of examples/rtlsdr_nbfm.lua as one device chain:
    Tuner(offset, 2*(deviation + bw), 50) -> FrequencyDiscriminator(deviation/bw) -> Lowpass(128, bw)."""
    return _receiver([TunerBlock(tune_offset, 2 * (deviation + bandwidth), 50), B.FrequencyDiscriminatorBlock(deviation / bandwidth),
                      B.LowpassFilterBlock(128, bandwidth)], rate)
