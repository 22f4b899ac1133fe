"""The FM receiver as ONE launch (luaradio_amd/csrc/kernels_rx.h; examples/rtlsdr_wbfm_mono.lua:12-17 after DeviceChainBlock.collapse).

No chain-level vector exists in the reference (unpinned), so the bar is the composition of the per-block-pinned oracle
restatements (RMS <= 1e-5, north_star) plus the properties the single launch must keep: it carries exactly the state of the two-launch
form (the two may alternate chunk by chunk), any chunking gives the same audio to Float32 rounding, a run boundary between workgroups
leaves no seam, and the sample count never depends on how the stream was cut."""
import numpy as np
import pytest

import luaradio_amd as lr
from luaradio_amd import _lib
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FS = 1102500.0


def fm(n, seed=3, noise=0.01):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / FS * np.cumsum(m)
    return (np.exp(1j * ph) + noise * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)


def receiver(flags=0):
    rx = lr.wbfm_mono_receiver(FS, -250e3)
    if flags:
        rx._chain = lr.Chain(rx._blocks, flags)
    return rx


def chunked(rx, x, cuts):
    parts, a = [], 0
    for b in list(cuts) + [len(x)]:
        parts.append(rx.process(x[a:b]))
        a = b
    return np.concatenate(parts)


def test_single_launch_vs_oracle_and_two_launch_form():
    n = 2500000                                          # 245 tiles: 31 runs of 8 tiles - run boundaries, partial batches, a partial last tile
    x = fm(n)
    one, two = receiver(), receiver(_lib.CHAIN_NO_SINGLE_LAUNCH)
    a, b = one.process(x), two.process(x)
    assert one.chain.last_launches == 1 and two.chain.last_launches == 2
    assert len(a) == len(b) == (n + 24) // 25
    assert float(np.max(np.abs(a - b))) < 2e-7           # same arithmetic up to the recurrence's scan order and the audio filter's block placement
    want = O.wbfm_mono_chain(FS, -250e3, mode=O.MODE_LUA, rot_mode=O.MODE_F64).process(x[:1000000])
    k = len(want) - 8
    err = a[:k].astype(np.float64) - want[:k]
    assert float(np.sqrt(np.mean(err ** 2))) <= 1e-5 and float(np.max(np.abs(err))) < 1e-6
    # no seam where one workgroup's run ends and the next begins: the error against the two-launch form is flat over the whole vector
    d = np.abs(a.astype(np.float64) - b)
    assert float(d[len(d) // 2:].max()) < 2e-7 and float(d.max()) < 2e-7


def _oracle_receiver(fs, tau):
    """oracle.wbfm_mono_chain with the time constant as a parameter (examples/rtlsdr_wbfm_mono.lua:12-17 with another FMDeemphasisFilter argument)"""
    r1 = fs / 5
    b, a = O.fm_deemphasis_taps(tau, r1)
    return O.Chain(O.tuner(-250e3, 200e3, 5, fs, mode=O.MODE_LUA, rot_mode=O.MODE_F64).stages +
                   [O.FMDiscriminator(1.25), O.lowpass(128, 15e3, r1, False, mode=O.MODE_LUA), O.IIR(b, a, False, O.MODE_LUA), O.Downsampler(5, False)])


@pytest.mark.parametrize("fs,tau,single", [(1102500.0, 75e-6, True), (1102500.0, 50e-6, True), (2400000.0, 75e-6, False), (2048000.0, 75e-6, False),
                                           (1102500.0, 750e-6, False)])
def test_slow_deemphasis_poles_keep_the_two_launch_form(fs, tau, single):
    """the single launch warms the low-rate recurrence of every workgroup run over 75 zero-state audio outputs, which only
    covers poles with q^75 <= 2^-25 (stock: 75 us at 220.5 kHz, q = 0.739, q^75 = 1.4e-10).  The same receiver at 2.4 / 2.048 MS/s (q = 0.87 / 0.85)
    or with a long time constant must NOT take it: run boundaries would carry an error of q^75 of the state, above the 1e-6 contract.  Such chains
    run the two-launch form, whose warm-up scales with the pole, and agree with the oracle chain everywhere - also far from the stream's start."""
    from luaradio_amd import blocks as B, composites as C
    n = 3000000
    rng = np.random.default_rng(21)
    t = np.arange(n) / fs
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    x = (np.exp(1j * (2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * np.cumsum(m))) + 0.01 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)

    def blocks():
        af = B.LowpassFilterBlock(128, 15e3)
        af.use_fft = 3
        return [C.TunerBlock(-250e3, 200e3, 5), B.FrequencyDiscriminatorBlock(1.25), af, B.FMDeemphasisFilterBlock(tau), B.DownsamplerBlock(5)]

    rx = C._receiver(blocks(), fs)
    got = rx.process(x)
    # two launches where the polyphase tail still fuses (q^320 w < 1e-12, stage_fir.h fuse_iir1); the 750 us pole is too slow even for that and
    # runs filter, recurrence and downsampler as launches of their own
    assert rx.chain.last_launches == 1 if single else rx.chain.last_launches >= 2
    two = C._receiver(blocks(), fs)
    two._chain = lr.Chain(two._blocks, _lib.CHAIN_NO_SINGLE_LAUNCH)
    ref = two.process(x)
    assert len(got) == len(ref) == (n + 24) // 25
    if single:
        assert float(np.max(np.abs(got - ref))) < 2e-7
    else:
        assert np.array_equal(got, ref)                 # it IS the two-launch form
    # against the oracle chain over the last 20 % of the stream (dozens of run boundaries in front of it)
    lo = (n * 4 // 5) // 25 * 25
    warm = 400000 // 25 * 25
    want = _oracle_receiver(fs, tau).process(x[lo - warm:])[warm // 25:]
    k = min(len(want), len(got) - lo // 25) - 8
    err = got[lo // 25:lo // 25 + k].astype(np.float64) - want[:k]
    assert float(np.max(np.abs(err))) < 1e-6, float(np.max(np.abs(err)))


@pytest.mark.parametrize("cuts", [[1], [5], [24, 25, 26], [8192, 8193, 500000], [12800 * 7, 12800 * 7 + 3, 12800 * 7 + 9, 1500000],
                                  list(range(100000, 2000000, 333337)), list(range(8192, 400000, 8192))])
def test_chunking_changes_nothing_but_float32_rounding(cuts):
    x = fm(2000000, seed=5)
    whole = receiver().process(x)
    got = chunked(receiver(), x, cuts)
    assert len(got) == len(whole)
    assert float(np.max(np.abs(got - whole))) < 2e-7


def test_forms_alternate_chunk_by_chunk_on_the_same_state():
    """the single launch reads and writes the two stages' own state buffers, so a chain may take either form from one chunk to the next
    (what happens when a chunk is too short to emit an audio sample, or its device pointer is not 8-byte aligned)"""
    x = fm(1500000, seed=7)
    whole = receiver().process(x)
    rx = receiver()
    # 3-sample and 20-sample chunks emit no audio sample on their own -> two-launch form; the long ones take the single launch
    cuts = [400000, 400003, 400023, 900000, 900001, 900002, 900020]
    forms, parts, a = [], [], 0
    for b in cuts + [len(x)]:
        parts.append(rx.process(x[a:b]))
        forms.append(rx.chain.last_launches)
        a = b
    got = np.concatenate(parts)
    assert 1 in forms and max(forms) >= 2
    assert len(got) == len(whole) and float(np.max(np.abs(got - whole))) < 2e-7


def test_weak_and_silent_input_stays_finite():
    """zeros and denormal-level input: angles of zero products follow the reference's sign rule; nothing in the window may turn into NaN
    through the Toeplitz product's zero taps"""
    n = 600000
    x = np.zeros(n, np.complex64)
    x[200000:400000] = fm(200000, seed=9, noise=0.0) * np.float32(1e-30)
    got = receiver().process(x)
    want = O.wbfm_mono_chain(FS, -250e3, mode=O.MODE_LUA, rot_mode=O.MODE_F64).process(x)
    assert len(got) == len(want) and np.all(np.isfinite(got))
    # silent stretches: exactly the reference's zeros (a -0 in a filter output would turn one angle per tile into pi)
    lead = 200000 // 25 - 8
    assert float(np.max(np.abs(got[:lead]))) == 0.0 and float(np.max(np.abs(want[:lead]))) == 0.0
    # the weak stretch: products of 1e-30-level outputs underflow, the angles are ill-conditioned - finite and bounded is all that can be asked
    assert float(np.max(np.abs(got))) < 1.0
    # the tail after the weak stretch decays back to silence like the oracle's
    assert float(np.max(np.abs(got[-2000:] - want[-2000:]))) < 1e-6


def test_time_partitions_of_the_single_launch_receiver():
    """lrhip_chain_start_at on the single-launch form: the partition's audio equals the single stream's to the warm-up's 1e-9 (run boundaries
    fall differently in a partition, so this form is not bit-identical there; LRHIP_CHAIN_NO_SINGLE_LAUNCH is)"""
    n = 1 << 21
    x = fm(n, seed=11)
    whole = receiver().process(x)
    for first in (128000 * 3, 128000 * 9 + 12825):
        rx = receiver()
        s = rx.chain.start_at(first)
        got = rx.process(x[s:])
        want = whole[(first + 24) // 25:]
        assert len(got) == len(want) and float(np.max(np.abs(got - want))) < 5e-5


# ---- the other round-3 kernels at size (properties that need no oracle run over millions of samples) --------------------------------------------
@pytest.mark.parametrize("M", [65, 129, 33])
def test_hilbert_single_launch_properties_at_size(M):
    """HilbertTransformBlock in one launch (65 / 129 taps: kernels_firwin.h hilbert_win_kernel, the zero taps skipped; other counts: kernels_fir.h HILB
    epilogue), 2^22 samples in ragged chunks: the real part IS the input delayed by (M - 1) / 2 samples (bit for bit: it is a copy out of the staged
    window), the imaginary part has the bits of the same taps run as a plain direct-form FIRFilterBlock - skipping a zero tap adds +-0 to a finite
    sum (hilberttransform.lua:107-124 computes both in one loop)"""
    from luaradio_amd import types
    n = 1 << 22
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    hb = lr.HilbertTransformBlock(M)
    hb.rate = 2.0
    hb.differentiate([types.Float32])
    hb.initialize()
    cuts = [1, 2, 33, 64, 65, 8191, 8192, 1000003, 3000001]
    parts, a = [], 0
    for b in cuts + [n]:
        parts.append(hb.process(x[a:b]))
        a = b
    got = np.concatenate(parts)
    assert got.dtype == np.complex64 and len(got) == n
    half = (M - 1) // 2
    assert np.array_equal(got.real[half:], x[:n - half]) and not got.real[:half].any()
    fir = lr.FIRFilterBlock(np.asarray(hb.hilbert_taps if hasattr(hb, "hilbert_taps") else hb.taps, np.float32), False)
    fir.rate = 2.0
    fir.differentiate([types.Float32])
    fir.initialize()
    assert np.array_equal(got.imag, fir.process(x))


@pytest.mark.parametrize("L,D", [(3, 2), (2, 3), (4, 3), (3, 4), (5, 4), (4, 5)])
def test_rational_resampler_kept_phases_kernel_at_size(L, D):
    """fir_rational_kernel on 2^21 samples in two unequal chunks (second one starts at an absolute input position that is not a multiple of D):
    the bits of the unfused device blocks MultiplyConstant -> Upsampler -> Lowpass -> Downsampler, and linearity in the input"""
    from luaradio_amd import types
    n = (1 << 21) + 1237
    rng = np.random.default_rng(50 + 10 * L + D)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)

    def resampler():
        r = lr.RationalResamplerBlock(L, D)
        r.rate = 2.0
        r.differentiate([types.ComplexFloat32])
        r.initialize()
        return r

    rs = resampler()
    cut = 700001
    got = np.concatenate([rs.process(x[:cut]), rs.process(x[cut:])])
    assert rs.chain.last_launches == 1
    want, rate = x, 2.0
    for b in (lr.MultiplyConstantBlock(float(L)), lr.UpsamplerBlock(L), lr.LowpassFilterBlock(128, min(1 / L, 1 / D), 1.0), lr.DownsamplerBlock(D)):
        b.rate = rate
        b.differentiate([types.ComplexFloat32])
        b.initialize()
        want, rate = b.process(want), b.get_rate()
    assert len(got) == len(want) == (n * L + D - 1) // D
    assert np.array_equal(got, want)
    # linearity (exact for a power-of-two scale)
    assert np.array_equal(resampler().process(x * np.float32(0.5)), (want * np.float32(0.5)).astype(np.complex64))


@pytest.mark.parametrize("L", [5, 2, 3, 4])
def test_interpolator_register_window_kernel_at_size(L):
    """fir_interp_kernel on 2^21 samples in two unequal chunks: the first launch does not fill the chip (persistent workgroups, register prefetch), the
    second does (round 4: one tile per workgroup in address order, whole tiles stored with every LDS read ahead of the first store, the chunk's ragged
    last tile on the rolled loop) - both give the bits of the unfused device blocks MultiplyConstant -> Upsampler -> Lowpass (interpolator.lua:31-34);
    and the same through LRHIP_RESAMPLE_ROUNDS-independent properties: linearity for a power-of-two scale"""
    from luaradio_amd import types
    n = (1 << 21) + 1237
    rng = np.random.default_rng(70 + L)
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)

    def interpolator():
        r = lr.InterpolatorBlock(L)
        r.rate = 2.0
        r.differentiate([types.ComplexFloat32])
        r.initialize()
        return r

    it = interpolator()
    cut = 700001
    got = np.concatenate([it.process(x[:cut]), it.process(x[cut:])])
    assert it.chain.last_launches == 1
    want, rate = x, 2.0
    for b in (lr.MultiplyConstantBlock(float(L)), lr.UpsamplerBlock(L), lr.LowpassFilterBlock(128, 1 / L, 1.0)):
        b.rate = rate
        b.differentiate([types.ComplexFloat32])
        b.initialize()
        want, rate = b.process(want), b.get_rate()
    assert len(got) == len(want) == n * L
    assert np.array_equal(got, want)
    assert np.array_equal(interpolator().process(x * np.float32(0.5)), (want * np.float32(0.5)).astype(np.complex64))


# ---- element-wise blocks on 16-byte accesses (round 3): the vector kernels against the one-sample kernels they replaced -----------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 1023, 1024, 1025, 300003])
def test_elementwise_vector_kernels_equal_the_scalar_ones(n):
    """Every *_vec_kernel of kernels_elem.h (16-byte accesses, tail on a spare thread of the same launch) gives the bits of the one-sample kernel: the
    library picks the vector form when the device pointers are 16-byte aligned, so the same block is run on an aligned and on a 4-byte-offset copy of
    the same data.  The golden-vector tests pin the values; this pins the two code paths to each other for ragged lengths."""
    import torch
    from luaradio_amd import _lib, types
    L = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n)
    pool = torch.rand(4 * n + 64, device="cuda", generator=g) * 2 - 1

    def views(nfloats, off):
        buf = torch.empty(nfloats + 16, device="cuda")
        v = buf[off:off + nfloats]
        v.copy_(pool[:nfloats])
        return buf, v

    def mk(cls, args, t):
        b = cls(*args)
        b.rate = 2.0
        b.differentiate([t] if not isinstance(t, list) else t)
        b.initialize()
        return b

    c, f = types.ComplexFloat32, types.Float32
    cases = [(lr.ComplexMagnitudeBlock, [], c, 2, 1, 1), (lr.ComplexPhaseBlock, [], c, 2, 1, 1), (lr.ComplexToRealBlock, [], c, 2, 1, 1),
             (lr.ComplexToImagBlock, [], c, 2, 1, 1), (lr.ComplexConjugateBlock, [], c, 2, 2, 1), (lr.RealToComplexBlock, [], f, 1, 2, 1),
             (lr.AbsoluteValueBlock, [], f, 1, 1, 1), (lr.AddConstantBlock, [0.25], f, 1, 1, 1), (lr.AddConstantBlock, [complex(0.25, -1.5)], c, 2, 2, 1),
             (lr.UpsamplerBlock, [3], c, 2, 2, 3), (lr.UpsamplerBlock, [5], f, 1, 1, 5), (lr.DelayBlock, [4], c, 2, 2, 1), (lr.DelayBlock, [3], f, 1, 1, 1),
             (lr.MultiplyConstantBlock, [complex(0.5, 2.0)], c, 2, 2, 1)]
    for cls, args, t, fin, fout, up in cases:
        outs = []
        for off in (4, 1):            # 16-byte aligned, 4-byte offset
            blk = mk(cls, args, t)
            _, xv = views(n * fin, off)
            ybuf = torch.zeros(n * up * fout + 16, device="cuda")
            yv = ybuf[off:off + n * up * fout]
            torch.cuda.synchronize()       # torch filled the buffers on ITS stream; the library launches on its own
            got = blk.process_device(xv.data_ptr(), n, yv.data_ptr(), n * up)
            assert got == n * up
            # a second chunk through the same block (Delay carries state; the others must not care)
            got = blk.process_device(xv.data_ptr(), n, yv.data_ptr(), n * up)
            torch.cuda.synchronize()
            outs.append(yv.cpu().numpy().copy())
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (cls.__name__, args, n)
    # two-input FloatToComplex
    outs = []
    for off in (4, 1):
        blk = mk(lr.FloatToComplexBlock, [], [f, f])
        _, av = views(n, off)
        bbuf = torch.empty(n + 16, device="cuda")
        bv = bbuf[off:off + n]
        bv.copy_(pool[n:2 * n])
        ybuf = torch.zeros(2 * n + 16, device="cuda")
        yv = ybuf[off:off + 2 * n]
        torch.cuda.synchronize()
        assert _lib.check(L.lrhip_stage_execute2_device(blk.stage_handle(), av.data_ptr(), bv.data_ptr(), n, yv.data_ptr(), n), "f2c") == n
        torch.cuda.synchronize()
        outs.append(yv.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0][0::2], pool[:n].cpu().numpy()) and np.array_equal(outs[0][1::2], pool[n:2 * n].cpu().numpy())
    # IQ file records: vector kernel (aligned) against the one-scalar kernel (1-byte offset), every format class
    raw = (torch.rand(16 * n + 64, device="cuda", generator=g) * 256).to(torch.uint8)
    for fmt, nbytes in (("u8", 1), ("s8", 1), ("u16le", 2), ("s16be", 2), ("u32le", 4), ("s32be", 4), ("f32le", 4), ("f32be", 4), ("f64le", 8)):
        outs = []
        for off in (16, 1):
            q = L.lrhip_format_convert_create(fmt.encode(), 1)
            assert q
            rbuf = torch.empty(2 * n * nbytes + 64, dtype=torch.uint8, device="cuda")
            rv = rbuf[off:off + 2 * n * nbytes]
            rv.copy_(raw[:2 * n * nbytes])
            if fmt.startswith("f"):          # keep the float formats finite: small integers in every 4 / 8-byte word
                rv.copy_((raw[:2 * n * nbytes] & 0x3f))
            ybuf = torch.zeros(2 * n + 16, device="cuda")
            yv = ybuf[4:4 + 2 * n]
            torch.cuda.synchronize()       # torch filled the buffers on ITS stream; the library launches on its own
            assert _lib.check(L.lrhip_stage_execute_device(q, rv.data_ptr(), n, yv.data_ptr(), n), fmt) == n
            torch.cuda.synchronize()
            outs.append(yv.cpu().numpy().copy())
            L.lrhip_stage_destroy(q)
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (fmt, n)


# ---- the receiver on raw u8 IQ records (round 3: IQFileSource's format stage folded into the single launch) ---------------------------------------
def test_s8_s16_record_conversion_formula_is_the_reference_expression():
    """the same for the other two folded formats: s8 (offset 0, scale 127.5) and s16 (offset 0, scale 32767.5), every raw value"""
    for vals, scale in ((np.arange(-128, 128), 127.5), (np.arange(-32768, 32768), 32767.5)):
        x = vals.astype(np.float64)
        want = (x / scale).astype(np.float32)
        rh = np.float32(1.0 / scale)
        rl = np.float32(1.0 / scale - np.float64(rh))
        t = np.float32(x * np.float64(rl))
        # fma(x, RH, t) with one rounding: exact rational arithmetic on the three Float32 values
        from fractions import Fraction
        got = np.array([np.float32(Fraction(float(xv)) * Fraction(float(rh)) + Fraction(float(tv))) for xv, tv in zip(x, t)], np.float32)
        assert np.array_equal(got, want), scale


def test_u8_record_conversion_formula_is_the_reference_expression():
    """kernels_rx.h rx_u8_sample: x = raw - 127.5 (exact), fma(x, RH, x * RL) with RH + RL = 1 / 127.5 - emulated here in double with one rounding per
    Float32 operation - gives Float32((raw - 127.5) / 127.5 evaluated in double) (format_utils.lua:82, iqfile.lua:99-113) for every byte value"""
    b = np.arange(256, dtype=np.float64)
    want = ((b - 127.5) / 127.5).astype(np.float32)
    rh = np.float32(1.0 / 127.5)
    rl = np.float32(1.0 / 127.5 - np.float64(rh))
    x = b - 127.5
    t = np.float32(x * np.float64(rl))                               # x * RL, rounded to Float32
    got = np.float32(x * np.float64(rh) + t.astype(np.float64))      # fma: one rounding (the sum is exact in double: 9 + 24 bits)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("fmt", ["u8", "s8", "s16le", "s16be"])
def test_receiver_reads_u8_records_in_the_single_launch(fmt):
    """[IQFileSource(u8 / s8 / s16le) format stage, Translator, Lowpass, Downsampler, Discriminator, Lowpass, Deemphasis, Downsampler] is ONE launch on the
    2- or 4-byte records, and its audio has the bits of the same receiver fed the converted ComplexFloat32 samples (the conversion is the same Float32
    values, the arithmetic behind it the same kernel), for ragged chunks; a big-endian format keeps its conversion launch (two launches, same bits)"""
    import importlib.util
    import os
    import torch
    from luaradio_amd import _lib, types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("iqfile_wbfm_mono", os.path.join(root, "examples", "iqfile_wbfm_mono.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    raw = np.frombuffer(ex.synth_capture(1102500.0, -250e3, 1.2), np.uint8)
    n = len(raw) // 2
    rb = {"u8": 2, "s8": 2, "s16le": 4, "s16be": 4}[fmt]                  # bytes per record
    if fmt == "s8":
        raw = (raw.astype(np.int16) - 128).astype(np.int8).view(np.uint8)
    elif fmt.startswith("s16"):
        rng16 = np.random.default_rng(3)
        v = ((raw.astype(np.int32) - 128) * 256 + rng16.integers(0, 256, len(raw))).astype(np.int16)      # every low byte occurs
        raw = v.astype("<i2" if fmt == "s16le" else ">i2").view(np.uint8)
    _src, chain, _rate = ex.build_chain(bytes(16), fmt, 1102500.0, -250e3)
    host = lr.IQFileSource(raw.tobytes(), fmt, 1102500.0)
    host.initialize()
    xc = host.read_all()
    assert len(xc) == n
    ref = lr.wbfm_mono_receiver(1102500.0, -250e3)
    d_raw = torch.from_numpy(raw.copy()).cuda()
    d_x = torch.from_numpy(xc.view(np.float32).copy()).cuda()
    cap = chain.max_output(n) + 64
    out8, outc = torch.zeros(cap, device="cuda"), torch.zeros(cap, device="cuda")
    cuts = [0, 25, 1000000, 1000001, 1000002, 1100000, n]
    got8, gotc = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        m8 = chain.process_device(d_raw.data_ptr() + rb * a, b - a, out8.data_ptr(), cap)
        if b - a > 1000:
            assert chain.last_launches == (2 if fmt == "s16be" else 1)
        mc = ref.process_device(d_x.data_ptr() + 8 * a, b - a, outc.data_ptr(), cap)
        assert m8 == mc
        torch.cuda.synchronize()
        got8.append(out8[:m8].cpu().numpy().copy())
        gotc.append(outc[:mc].cpu().numpy().copy())
    got8, gotc = np.concatenate(got8), np.concatenate(gotc)
    assert len(got8) == (n + 24) // 25
    if fmt == "s16be":
        # (not folded: the receiver then reads the chain's own 16-byte aligned edge buffer where the stand-alone receiver reads the caller's pointer, and the
        # window-relative rotator staging rounds with the alignment slack of the window, DESIGN.md 4.3a: same values to Float32 rounding)
        assert float(np.max(np.abs(got8 - gotc))) < 1e-6
    else:
        assert np.array_equal(got8.view(np.uint32), gotc.view(np.uint32))


def test_u8_receiver_time_partitions_agree_with_the_single_stream():
    """SURVEY.md 8e on the record stream: the chain that starts with IQFileSource's u8 format stage is cut into 2 and 4 time partitions (seek to the
    aligned sample in front of a - halo, replay, keep [a, b)); like the ComplexFloat32 single-launch receiver the partitions agree with the single
    stream to the 1e-7 of the run warm-up (tests/test_timeshard.py), the counts exactly"""
    import importlib.util
    import os
    from luaradio_amd import timeshard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("iqfile_wbfm_mono", os.path.join(root, "examples", "iqfile_wbfm_mono.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    raw = np.frombuffer(ex.synth_capture(1102500.0, -250e3, 1.9), np.uint8)
    n = len(raw) // 2
    _src, chain, _rate = ex.build_chain(bytes(16), "u8", 1102500.0, -250e3)
    whole = chain.process(raw)
    assert len(whole) == (n + 24) // 25
    h, align = chain.halo(), chain.shard_align()
    assert 0 < h < 200000 and align == 128000
    for parts in (2, 4):
        got = []
        for a, b in timeshard.bounds(n, parts, align):
            s = timeshard.replay_start(a, h, align)
            chain.seek(s)
            if a > s:
                chain.process(raw[2 * s:2 * a])
            got.append(chain.process(raw[2 * a:2 * b]))
        got = np.concatenate(got)
        assert len(got) == len(whole)
        assert float(np.max(np.abs(got - whole))) < 1e-7, parts


def test_fanout_branch_on_raw_records():
    """configs[3] fed from an IQ file: the slab that is broadcast holds the u8 records (a quarter of the bytes per sample on every xGMI link) and each
    branch is a chain that starts with the format stage; DeviceBranch counts records, the branch output equals the chain run on its own"""
    import importlib.util
    import os
    import torch
    from luaradio_amd import fanout
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("iqfile_wbfm_mono", os.path.join(root, "examples", "iqfile_wbfm_mono.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    raw = np.frombuffer(ex.synth_capture(1102500.0, -250e3, 0.5), np.uint8)
    n = len(raw) // 2
    _src, chain, _rate = ex.build_chain(bytes(16), "u8", 1102500.0, -250e3)
    want = chain.process(raw)
    chain.reset()
    fo = fanout.FanOut(None, 0, 1, 1, {0: fanout.DeviceBranch(chain, n)}, src=0)
    slab = torch.from_numpy(raw.copy()).cuda()
    got = fo.push(slab)[0]
    torch.cuda.synchronize()
    # (round 4: a host-pointer call of this size travels as pipelined pieces, i.e. as another chunking of the stream - the single-launch receiver agrees
    # across chunkings to 2e-7, include/lrhip.h "chains" 4; same count, same samples)
    g = got.cpu().numpy()
    assert len(g) == len(want) and float(np.max(np.abs(g - want))) < 2e-7


def test_fir_fft4k_kernel_at_size_against_the_f64_oracle():
    """kernels_firfft4k.h at 2^24 ComplexFloat32 samples, 1 276 real taps (the suite's five 256-tap filters as one), three ragged chunks: eight slabs of 4096
    outputs spread over the vector - including the chunk seams, where the window comes from the carried history - against the f64 oracle at the
    reference's 1e-6 (relative to the output scale), and the launch count (one per chunk)"""
    import torch
    from luaradio_amd import types
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    n, M = 1 << 24, 1276
    taps = (rng.uniform(0, 1, M) / M).astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
    blk = lr.FIRFilterBlock(taps, "fast")
    blk.rate = 2.0
    blk.differentiate([types.ComplexFloat32])
    blk.initialize()
    y = torch.zeros(2 * n + 16, device="cuda")
    cuts = [0, 5000001, 5000002 + 2816 * 1000, n]
    got = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        m = blk.process_device(x.data_ptr() + 8 * a, b - a, y.data_ptr() + 8 * a, b - a)
        assert m == b - a
        got += m
    torch.cuda.synchronize()
    assert got == n
    ln = 4096
    for o in (0, 1275, 5000001 - 2000, 5000001, 5000002 + 2816 * 1000 - 100, n // 2 + 333, n - ln - 1, n - ln):
        lo = max(0, o - (M - 1))
        xs = x[2 * lo:2 * (o + ln)].cpu().numpy().view(np.complex64)
        ref = O.FIR(taps, True, O.MODE_F64).process(xs)[o - lo:]
        out = y[2 * o:2 * (o + ln)].cpu().numpy().view(np.complex64)
        assert len(ref) == ln
        assert float(np.max(np.abs(out - ref))) <= 1e-6 * max(1.0, float(np.max(np.abs(ref)))), o


@pytest.mark.parametrize("decim", [5, 50, 9])
@pytest.mark.parametrize("rotate", [True, False])
@pytest.mark.parametrize("fmt", ["u8", "s8", "s16le", "u16le"])
def test_tuner_reads_raw_records_in_its_launch(fmt, rotate, decim):
    """[IQFileSource(u8 / s8 / s16le) format stage, FrequencyTranslator, Lowpass(128), Downsampler(5 / 9 / 50)] - a fan-out branch, or the tuner of the AM / SSB /
    NBFM receivers, fed from an IQ file - is ONE launch of the persistent Toeplitz kernel (decimation 5) or the LDS-staged decimator (9, 50) on the records, bit-equal to the same Tuner on the converted ComplexFloat32 samples (block-of-8 rotator
    staging: the stand-alone translator's phasors whatever the alignment), ragged chunks incl. one that emits nothing; another format (u16le) keeps its
    conversion launch and the same bits"""
    import torch
    from luaradio_amd import types
    rng = np.random.default_rng(11)
    n = 700003
    rb = {"u8": 2, "s8": 2, "s16le": 4, "u16le": 4}[fmt]
    raw = rng.integers(0, 256, n * rb, dtype=np.uint8)
    fs = 1102500.0

    def tuner(head):
        blocks = head + ([lr.FrequencyTranslatorBlock(-350e3)] if rotate else []) + [lr.LowpassFilterBlock(128, 500e3 / decim), lr.DownsamplerBlock(decim)]
        r, t = fs, types.ComplexFloat32
        for b in blocks[len(head):]:
            b.rate = r
            b.differentiate([t])
            b.initialize()
            r, t = b.get_rate(), b.get_output_type()
        return lr.Chain(blocks)

    src = lr.IQFileSource(bytes(16), fmt, fs)
    src.initialize()
    chain = tuner([src])
    host = lr.IQFileSource(raw.tobytes(), fmt, fs)
    host.initialize()
    xc = host.read_all()
    ref = tuner([])
    d_raw = torch.from_numpy(raw.copy()).cuda()
    d_x = torch.from_numpy(xc.view(np.float32).copy()).cuda()
    cap = chain.max_output(n) + 64
    o1, o2 = torch.zeros(2 * cap, device="cuda"), torch.zeros(2 * cap, device="cuda")
    cuts = [0, 3, 4, 100000, 100001, 400002, n]
    g1, g2 = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        m1 = chain.process_device(d_raw.data_ptr() + rb * a, b - a, o1.data_ptr(), cap)
        if b - a > 1000:
            assert chain.last_launches == (2 if fmt == "u16le" else 1)
        m2 = ref.process_device(d_x.data_ptr() + 8 * a, b - a, o2.data_ptr(), cap)
        assert m1 == m2
        torch.cuda.synchronize()
        g1.append(o1[:2 * m1].cpu().numpy().copy())
        g2.append(o2[:2 * m2].cpu().numpy().copy())
    g1, g2 = np.concatenate(g1), np.concatenate(g2)
    assert len(g1) == 2 * ((n + decim - 1) // decim)
    assert np.array_equal(g1.view(np.uint32), g2.view(np.uint32))


@pytest.mark.parametrize("op", ["ComplexMagnitudeBlock", "ComplexPhaseBlock", "ComplexToRealBlock", "ComplexToImagBlock"])
@pytest.mark.parametrize("records", [False, True])
def test_complex_to_real_block_runs_in_the_decimators_store(op, records):
    """examples/rtlsdr_am_envelope.lua:11-14: Tuner(offset, bw, 50) -> ComplexMagnitude.  Behind an LDS-staged decimator the complex -> real element-wise block
    (complexmagnitude.lua / complexphase.lua / complextoreal.lua / complextoimag.lua) runs on the filter's accumulators: ONE launch (also on u8 records), and the
    bits of the same Tuner chain followed by the stand-alone block, ragged chunks incl. one that emits nothing."""
    import torch
    from luaradio_amd import types
    rng = np.random.default_rng(5)
    n = 600007
    fs = 1102500.0
    raw = rng.integers(0, 256, n * 2, dtype=np.uint8)

    def build(head, with_op):
        blocks = head + [lr.FrequencyTranslatorBlock(-100e3), lr.LowpassFilterBlock(128, 5e3), lr.DownsamplerBlock(50)] + ([getattr(lr, op)()] if with_op else [])
        r, t = fs, types.ComplexFloat32
        for b in blocks[len(head):]:
            b.rate = r
            b.differentiate([t])
            b.initialize()
            r, t = b.get_rate(), b.get_output_type()
        return lr.Chain(blocks)

    def head():
        if not records:
            return []
        src = lr.IQFileSource(bytes(16), "u8", fs)
        src.initialize()
        return [src]

    host = lr.IQFileSource(raw.tobytes(), "u8", fs)
    host.initialize()
    xc = host.read_all()
    fused, tuner = build(head(), True), build([], False)
    alone = getattr(lr, op)()
    alone.rate = fs / 50
    alone.differentiate([types.ComplexFloat32])
    alone.initialize()
    d_in = torch.from_numpy(raw.copy()).cuda() if records else torch.from_numpy(xc.view(np.float32).copy()).cuda()
    d_x = torch.from_numpy(xc.view(np.float32).copy()).cuda()
    isz = 2 if records else 8
    cap = fused.max_output(n) + 64
    o1, o2 = torch.zeros(cap, device="cuda"), torch.zeros(2 * cap, device="cuda")
    cuts = [0, 7, 8, 100000, 100049, 400002, n]
    g1, g2 = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        m1 = fused.process_device(d_in.data_ptr() + isz * a, b - a, o1.data_ptr(), cap)
        if b - a > 1000:
            assert fused.last_launches == 1
        m2 = tuner.process_device(d_x.data_ptr() + 8 * a, b - a, o2.data_ptr(), cap)
        assert m1 == m2
        torch.cuda.synchronize()
        g1.append(o1[:m1].cpu().numpy().copy())
        g2.append(o2[:2 * m2].cpu().numpy().copy())
    g1, g2 = np.concatenate(g1), np.concatenate(g2).view(np.complex64)
    want = alone.process(g2)
    assert len(g1) == len(want) == (n + 49) // 50
    assert np.array_equal(g1.view(np.uint32), want.view(np.uint32))
