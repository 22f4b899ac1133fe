"""Round-3 boundary features of the C ABI, on the GPU (all through liblrhip.so):

  * lrhip_chain_create_ex flags - the numerical contract per chain (EXACT = what the blocks compute one by one, bit for bit);
  * lrhip_chain_start_at - a time partition's replayed halo is dropped inside the library, whatever entry point feeds the chain;
  * lrhip_chain_set_latency - live sources get their partial batches without an EOF flush;
  * the deferred wave-boundary fix-up when the consumer's chunk emits nothing (stage_fir.h);
  * "fast" arithmetic falls back to the direct form stand-alone exactly as inside a chain;
  * push() keeps its samples when a launch fails.
"""
import ctypes as C
import time

import os

import numpy as np
import pytest

import luaradio_amd as lr
from luaradio_amd import _lib, types
from luaradio_amd.composites import _fft_option
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FS = 1102500.0


def fm_signal(n, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / FS * np.cumsum(m)
    return (np.exp(1j * ph) + 0.01 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)


def make(cls, args, dtype, rate):
    blk = cls(*args)
    blk.rate = rate
    blk.differentiate([dtype])
    blk.initialize()
    return blk


def receiver_blocks():
    """examples/rtlsdr_wbfm_mono.lua:12-17 flattened: 7 blocks"""
    c, f = types.ComplexFloat32, types.Float32
    return [make(lr.FrequencyTranslatorBlock, [-250e3], c, FS), make(lr.LowpassFilterBlock, [128, 100e3], c, FS),
            make(lr.DownsamplerBlock, [5], c, FS), make(lr.FrequencyDiscriminatorBlock, [1.25], c, FS / 5),
            make(lr.LowpassFilterBlock, [128, 15e3], f, FS / 5), make(lr.FMDeemphasisFilterBlock, [75e-6], f, FS / 5),
            make(lr.DownsamplerBlock, [5], f, FS / 5)]


def run_chunked(proc, x, cuts):
    parts, a = [], 0
    for b in list(cuts) + [len(x)]:
        parts.append(proc(x[a:b]))
        a = b
    return np.concatenate(parts)


def test_exact_flag_gives_the_blocks_own_bits_for_any_chunking():
    x = fm_signal(400000)
    cuts = [1, 7, 8192, 8193, 100000, 100001, 333333]
    # the blocks one by one, as the reference runs them, on the same chunks (the recurrence's scan rounds with its tile grid, which starts
    # with the chunk - iirfilter parity is 1e-6 - so like is compared with like; everything in front of it does not depend on chunking)
    blocks = receiver_blocks()

    def one_by_one(v):
        for b in blocks:
            v = b.process(v)
        return v

    want = run_chunked(one_by_one, x, cuts)
    exact = lr.Chain(receiver_blocks(), exact=True)
    assert exact.flags == _lib.CHAIN_EXACT == 11
    got = run_chunked(exact.process, x, cuts)
    assert len(got) == len(want) and np.array_equal(got, want)
    # ... and up to the de-emphasis the bits do not depend on the chunking either
    head = lr.Chain(receiver_blocks()[:5], exact=True)
    whole = x
    for b in receiver_blocks()[:5]:
        whole = b.process(whole)
    assert np.array_equal(run_chunked(head.process, x, cuts), whole)
    # the default contract: same values to the stated roundings, fewer launches
    fast = lr.Chain(receiver_blocks())
    got2 = fast.process(x)
    assert len(got2) == len(want)
    assert float(np.sqrt(np.mean((got2.astype(np.float64) - want) ** 2))) < 1e-6
    exact.process(x)
    assert fast.last_launches < exact.last_launches


@pytest.mark.parametrize("decim,ntaps", [(50, 128), (25, 128), (80, 128), (50, 200)])
@pytest.mark.parametrize("with_disc", [False, True])
def test_exact_chain_on_the_lds_staged_decimator_shapes(decim, ntaps, with_disc):
    """the second LDS-staged decimator form (kernels_firdecim.h) splits an output's taps over the two half-waves in its rotator form
    (tiles of <= 128 outputs: decimation 50 / 25 / 80 with 128 taps) - one add joins two partial chains, which is not the bits of the blocks run
    one by one.  An exact chain (include/lrhip.h: LRHIP_CHAIN_EXACT, 'bit for bit for direct-form filters') must keep one fmaf chain per output:
    TunerBlock(.., decim) [+ FrequencyDiscriminatorBlock] of rtlsdr_nbfm.lua / rtlsdr_ax25.lua flattened, on ragged chunks."""
    c = types.ComplexFloat32
    x = fm_signal(300000, seed=11)
    cuts = [3, 4099, 4100, 77777, 200001]

    def blocks():
        lp = lr.LowpassFilterBlock(ntaps, 10e3)
        lp.use_fft = lr.block.fir_mode(False)                  # the direct form: the contract's 'bit for bit' case
        lp.rate = FS
        lp.differentiate([c])
        lp.initialize()
        bs = [make(lr.FrequencyTranslatorBlock, [-250e3], c, FS), lp, make(lr.DownsamplerBlock, [decim], c, FS)]
        if with_disc:
            bs.append(make(lr.FrequencyDiscriminatorBlock, [1.25], c, FS / decim))
        return bs

    ref = blocks()

    def one_by_one(v):
        for b in ref:
            v = b.process(v)
        return v

    want = run_chunked(one_by_one, x, cuts)
    exact = lr.Chain(blocks(), exact=True)
    got = run_chunked(exact.process, x, cuts)
    assert len(got) == len(want) == len(x) // decim
    assert np.array_equal(got, want)
    assert exact.last_launches < len(ref)                     # still fused: the flag changes the arithmetic's order, not the launch count
    # the default contract on the same shape: same values to Float32 rounding of the filter outputs (the angles behind them are as well
    # conditioned as the filtered signal is strong - not compared here)
    if not with_disc:
        fast = lr.Chain(blocks())
        got2 = run_chunked(fast.process, x, cuts)
        assert float(np.max(np.abs(got2.astype(np.complex128) - want))) < 2e-6


def test_no_fusion_flag_and_unknown_bits():
    x = fm_signal(100000)
    blocks = receiver_blocks()
    plain = lr.Chain(blocks, exact=_lib.CHAIN_NO_FUSION)
    got = plain.process(x)
    assert plain.last_launches >= 7
    want = x
    for b in receiver_blocks():
        want = b.process(want)
    assert np.array_equal(got, want)
    L = _lib.load()
    arr = (C.c_void_p * 2)(blocks[0].stage_handle(), blocks[1].stage_handle())
    assert not L.lrhip_chain_create_ex(arr, 2, 1 << 9)
    assert b"unknown flag" in L.lrhip_strerror()


@pytest.mark.parametrize("entry", ["execute", "push"])
def test_start_at_drops_the_replayed_halo_inside_the_library(entry):
    n = 1 << 20
    x = fm_signal(n, seed=9)
    # the two-launch form: partitions on the tile grid are bit-identical (the single launch agrees to its recurrence warm-up, tests/test_gpu_rx.py)
    whole = lr.Chain(receiver_blocks(), _lib.CHAIN_NO_SINGLE_LAUNCH).process(x)
    chain = lr.Chain(receiver_blocks(), _lib.CHAIN_NO_SINGLE_LAUNCH)
    align = chain.shard_align()
    first = 4 * align                                   # a partition boundary on the tile grid: bit-identical
    s = chain.start_at(first)
    assert s <= first - chain.halo() and s % align == 0 and first - s < chain.halo() + align
    want = whole[first // 25:]
    if entry == "execute":
        got = chain.process(x[s:])                      # one call: the library splits it at `first`, which is on the tile grid
        assert len(got) == len(want) and np.array_equal(got, want)
        # cuts inside and across the replay, off the grid: same sample COUNT, values to Float32 rounding of the tuner outputs
        assert chain.start_at(first) == s
        got = run_chunked(chain.process, x[s:], [1000, first - s - 3, first - s + 5])
        assert len(got) == len(want) and float(np.max(np.abs(got - want))) < 5e-5
    else:
        chain.set_ring(3, align)                        # batches on the tile grid
        parts = [chain.push(x[k:k + 8192]) for k in range(s, n, 8192)]
        parts.append(chain.flush())
        got = np.concatenate(parts)
        assert len(got) == len(want) and np.array_equal(got, want)
    # an unaligned first sample: same values to Float32 rounding (documented), never duplicated or missing samples
    chain2 = lr.Chain(receiver_blocks(), _lib.CHAIN_NO_SINGLE_LAUNCH)
    first2 = 3 * align + 12350                          # multiple of 25: the output grid of the single stream
    s2 = chain2.start_at(first2)
    got2 = chain2.process(x[s2:])
    want2 = whole[first2 // 25:]
    assert len(got2) == len(want2) and float(np.max(np.abs(got2 - want2))) < 5e-5
    # chains with unbounded memory refuse
    agc = lr.Chain([make(lr.AGCBlock, ["fast"], types.ComplexFloat32, FS)])
    with pytest.raises(_lib.LrhipError):
        agc.start_at(1000)


def test_latency_bound_releases_partial_batches_without_a_flush():
    x = fm_signal(1 << 18, seed=11)
    want = lr.Chain(receiver_blocks()).process(x)
    chain = lr.Chain(receiver_blocks())
    chain.set_ring(3, 1 << 20)                          # DeviceChainBlock's batch: a second of RF at 1.1 MS/s
    chain.set_latency(0.01)
    parts, emitted_calls = [], 0
    for k in range(0, len(x), 8192):
        if k in (8 * 8192, 20 * 8192):
            time.sleep(0.03)                            # a live source: the batch is far from full, the clock is not
        out = chain.push(x[k:k + 8192])
        emitted_calls += len(out) > 0
        parts.append(out)
    before_flush = sum(len(p) for p in parts)
    parts.append(chain.flush())
    got = np.concatenate(parts)
    assert emitted_calls >= 2 and before_flush > 0      # without the bound nothing leaves a 2^20 batch before EOF
    assert len(got) == len(want)
    assert float(np.max(np.abs(got - want))) < 5e-5      # batches cut off the tile grid: Float32 rounding of the tuner outputs
    quiet = lr.Chain(receiver_blocks())
    quiet.set_ring(3, 1 << 20)
    assert sum(len(quiet.push(x[k:k + 8192])) for k in range(0, len(x), 8192)) == 0
    with pytest.raises(_lib.LrhipError):
        quiet.set_latency(-1.0)


def test_poll_releases_the_partial_batch_of_a_source_that_stalls():
    """VERDICT r03 missing 7: the latency bound used to be checked only inside push(), so a live source that goes quiet left its partial batch
    unlaunched for ever, where the reference streams every chunk through as it arrives (radio/core/block.lua:575-602).  lrhip_chain_poll_due()
    tells the host how long it may wait for input; lrhip_chain_poll() after that wait launches the batch and returns its samples."""
    x = fm_signal(100000, seed=17)
    want = lr.Chain(receiver_blocks()).process(x)
    chain = lr.Chain(receiver_blocks())
    chain.set_ring(3, 1 << 20)
    assert chain.poll_due() == -1.0 and len(chain.poll()) == 0           # nothing pending, no bound: wait for input forever
    chain.set_latency(0.05)
    assert chain.poll_due() == -1.0                                     # a bound, but nothing pending
    t0 = time.monotonic()
    assert len(chain.push(x[:1000])) == 0                               # 1000 samples of a 2^20 batch, well inside the 50 ms
    due = chain.poll_due()
    assert 0.0 < due <= 0.05
    assert len(chain.poll()) == 0 or time.monotonic() - t0 >= 0.05      # not due yet: poll() returns at once with nothing
    time.sleep(due + 0.005)                                             # ... the source stalls; the host's poll(2) on its input times out
    assert chain.poll_due() == 0.0
    first = chain.poll()
    assert len(first) == 1000 // 25 and chain.poll_due() == -1.0        # the 1000 samples' audio, without another push and without EOF
    assert len(chain.poll()) == 0
    rest = chain.push(x[1000:])                                         # the source comes back
    time.sleep(0.06)
    got = np.concatenate([first, rest, chain.poll(), chain.flush()])
    assert len(got) == len(want) and float(np.max(np.abs(got - want))) < 5e-5
    # without a latency bound poll() never cuts a batch (file / benchmark sources: batches run only when full)
    quiet = lr.Chain(receiver_blocks())
    quiet.set_ring(3, 1 << 20)
    quiet.push(x[:1000])
    time.sleep(0.02)
    assert quiet.poll_due() == -1.0 and len(quiet.poll()) == 0 and len(quiet.flush()) == 40


def test_deferred_fixup_when_the_tail_emits_nothing():
    """the tuner + discriminator leaves its wave-first outputs to the audio tail's staging; chunks so small that the tail launches no
    kernel (fewer than 5 tuner outputs) must still patch them before they enter the tail's history"""
    x = fm_signal(60000, seed=13)
    want = lr.Chain(receiver_blocks()).process(x)
    chain = lr.Chain(receiver_blocks())
    cuts = [5120 * 4, 5120 * 4 + 3, 5120 * 4 + 9, 5120 * 4 + 10, 5120 * 4 + 17, 40000, 40004, 40011]
    got = run_chunked(chain.process, x, cuts)
    assert len(got) == len(want)
    assert float(np.max(np.abs(got - want))) < 5e-5
    ora = O.wbfm_mono_chain(FS, -250e3, mode=O.MODE_LUA, rot_mode=O.MODE_F64).process(x)
    assert float(np.sqrt(np.mean((got.astype(np.float64) - ora) ** 2))) <= 1e-5


def test_fast_arithmetic_falls_back_to_the_direct_form_standalone_as_in_a_chain():
    rng = np.random.default_rng(17)
    x = (rng.uniform(-1, 1, 50000) + 1j * rng.uniform(-1, 1, 50000)).astype(np.complex64)
    # 40 taps at decimation 7: no polyphase-FFT instantiation
    fast = make(lr.DecimatorBlock, [7, {"num_taps": 40, "use_fft": "fast"}], types.ComplexFloat32, 2.0)
    direct = make(lr.DecimatorBlock, [7, {"num_taps": 40, "use_fft": False}], types.ComplexFloat32, 2.0)
    assert np.array_equal(fast.process(x), direct.process(x))
    taps = np.asarray(lr.filter_utils.firwin_lowpass(40, 1 / 7), np.float32)
    L = _lib.load()
    st = L.lrhip_fir_create(taps.ctypes.data_as(C.POINTER(C.c_float)), 40, 0, 1, 7, 2)
    assert st, L.lrhip_strerror()
    L.lrhip_stage_destroy(st)


def test_fir_mode_is_one_table_for_every_front_end():
    from luaradio_amd import block
    assert [block.fir_mode(v) for v in ("auto", "fast", True, False)] == [3, 2, 1, 0]
    assert block.fir_mode(None) == 0                    # under the jig (tests/conftest.py), as firfilter.lua:57
    saved, block.TESTS_JIGS_LOADED = block.TESTS_JIGS_LOADED, False
    try:
        assert block.fir_mode(None) == 3
        blk = lr.LowpassFilterBlock(128, 0.2, 1.0)
        assert blk.use_fft == 3
    finally:
        block.TESTS_JIGS_LOADED = saved


def test_push_keeps_its_samples_when_the_launch_fails():
    x = fm_signal(40000, seed=19)
    chain = lr.Chain(receiver_blocks())
    with pytest.raises(_lib.LrhipError, match="no ring"):
        chain.push(x[:100])
    with pytest.raises(_lib.LrhipError, match="no ring"):
        chain.submit(x[:100])
    assert len(chain.flush()) == 0                      # nothing is ever pending without a ring
    chain.set_ring(2, 16384)
    parts = [chain.push(x[k:k + 5000]) for k in range(0, len(x), 5000)]
    parts.append(chain.flush())
    got = np.concatenate(parts)
    want = lr.Chain(receiver_blocks()).process(x)
    assert len(got) == len(want) and float(np.max(np.abs(got - want))) < 5e-5


def test_cascaded_overlap_save_filters_run_as_one_filter():
    """benchmarks/luaradio_benchmark.lua:17-37, the suite's first entry: five 256-tap FIRFilterBlocks back to back.  A chain of filters is a filter
    (1 276 taps); filters that asked for the overlap-save arithmetic are merged into ONE launch of the 4096-point kernel, and the merged
    filter is held to the f64 oracle CHAIN (five stages, each rounded to Float32 like the reference's vectors between blocks)."""
    rng = np.random.default_rng(21)
    c = types.ComplexFloat32
    taps = [(rng.uniform(0, 1, 256) / 128).astype(np.float32) for _ in range(5)]
    n = 300001
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    want = x
    for t in taps:
        want = O.FIR(t, True, O.MODE_F64).process(want)
    scale = float(np.max(np.abs(want)))
    cuts = [1, 4095, 4096, 100000, 100001]
    merged = lr.Chain([make(lr.FIRFilterBlock, [t, "fast"], c, FS) for t in taps])
    got = run_chunked(merged.process, x, cuts)
    assert merged.last_launches == 1
    assert len(got) == n and float(np.max(np.abs(got - want))) <= 2e-6 * max(scale, 1.0)
    # LRHIP_CHAIN_EXACT keeps every block's own arithmetic: five filters, five launches, the same values to the same bar
    apart = lr.Chain([make(lr.FIRFilterBlock, [t, "fast"], c, FS) for t in taps], exact=True)
    got5 = run_chunked(apart.process, x, cuts)
    assert apart.last_launches == 5
    assert float(np.max(np.abs(got5 - want))) <= 5e-6 * max(scale, 1.0)
    assert float(np.max(np.abs(got5 - got))) <= 5e-6 * max(scale, 1.0)
    # direct-form filters (bit-exact by contract) are never merged
    direct = lr.Chain([make(lr.FIRFilterBlock, [t[:16], False], c, FS) for t in taps[:2]])
    direct.process(x[:5000])
    assert direct.last_launches == 2
    # six filters = 1 531 taps do not fit one block: five are merged, the sixth stays
    six = lr.Chain([make(lr.FIRFilterBlock, [t, "fast"], c, FS) for t in taps + taps[:1]])
    six.process(x[:100000])
    assert six.last_launches == 2


def test_tuner_without_decimation_is_a_rotator_and_a_filter():
    """radio/composites/tuner.lua:32-48 with decimation 1: FrequencyTranslator -> LowpassFilter -> Downsampler(1).  The identity downsampler gets no
    launch and the rotator is NOT folded into a filter that does not decimate (the rotating Toeplitz kernel at D = 1 is slower than the pair): two
    launches, and with the direct form they produce the bits of the blocks run one by one."""
    c = types.ComplexFloat32
    n = 200003
    x = fm_signal(n)
    cuts = [1, 777, 65536, 100000]

    def blocks(mode):
        f = lr.LowpassFilterBlock(128, 100e3)
        f.use_fft = _fft_option({"use_fft": mode})      # as TunerBlock.instantiate does
        f.rate = FS
        f.differentiate([c])
        f.initialize()
        return [make(lr.FrequencyTranslatorBlock, [-250e3], c, FS), f, make(lr.DownsamplerBlock, [1], c, FS)]

    one_by_one = x
    for b in blocks(False)[:2]:
        one_by_one = b.process(one_by_one)
    chain = lr.Chain(blocks(False))
    got = run_chunked(chain.process, x, cuts)
    assert chain.last_launches == 2
    assert len(got) == n and np.array_equal(got.view(np.uint32), one_by_one.view(np.uint32))
    fast = lr.Chain(blocks("fast"))
    gotf = run_chunked(fast.process, x, cuts)
    assert fast.last_launches == 2
    assert len(gotf) == n and float(np.max(np.abs(gotf - one_by_one))) <= 2e-6
    # a chain that is nothing but Downsampler(1) still copies
    ident = lr.Chain([make(lr.DownsamplerBlock, [1], c, FS)])
    assert np.array_equal(ident.process(x[:1000]), x[:1000])


def _aligned(count, dtype):
    raw = np.empty(count * np.dtype(dtype).itemsize + 4096, np.uint8)
    off = (-raw.ctypes.data) % 4096
    return raw[off:off + count * np.dtype(dtype).itemsize].view(dtype)


def test_host_path_travels_in_pieces_and_takes_registered_vectors_without_a_copy():
    """VERDICT r03 missing 6: radio/core/vector.lua:19-37 hands out page-aligned, long-lived buffers; lrhip_host_register pins them where they lie and the
    synchronous host-pointer entry points DMA from / to them.  Independently, a call of 2^19 samples and more travels as up to eight pipelined pieces
    (H2D, kernels, D2H on three streams).  Neither changes a bit of what a direct-form block computes: one big call == the same stream in small calls
    == registered vectors, for a rate-preserving filter and for a decimating chain whose pieces end off the decimation grid."""
    import ctypes as C
    L = _lib.load()
    rng = np.random.default_rng(31)
    n = (1 << 21) + 777
    x = _aligned(n, np.complex64)
    x[:] = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    taps = (rng.uniform(-1, 1, 128) / 128).astype(np.float32)

    def fir():
        return make(lr.FIRFilterBlock, [taps, False], types.ComplexFloat32, FS)

    small = np.concatenate([b for blk in [fir()] for b in (blk.process(x[a:a + 100000]) for a in range(0, n, 100000))])
    whole = fir().process(x)                                  # 8 pieces
    assert np.array_equal(whole, small)
    y = _aligned(n, np.complex64)
    _lib.check(L.lrhip_host_register(x.ctypes.data_as(C.c_void_p), x.nbytes), "register")
    _lib.check(L.lrhip_host_register(y.ctypes.data_as(C.c_void_p), y.nbytes), "register")
    try:
        assert L.lrhip_host_register(x.ctypes.data_as(C.c_void_p), x.nbytes) == 0           # same range again: fine
        assert L.lrhip_host_register(x.ctypes.data_as(C.c_void_p), x.nbytes // 2) < 0       # another size: refused
        blk = fir()
        got = L.lrhip_stage_execute(blk.stage_handle(), x.ctypes.data_as(C.c_void_p), n, y.ctypes.data_as(C.c_void_p), n)
        assert got == n and np.array_equal(y, small)
        # a sub-vector of a registered range is registered; an unregistered output next to a registered input is staged
        blk = fir()
        part = blk.process(x[5:5 + (1 << 20)])
        assert np.array_equal(part, fir().process(x[5:5 + (1 << 20)].copy()))
        # decimating chain: pieces end off the decimation grid, the count and the samples are those of the small calls
        dec = lr.Chain([make(lr.LowpassFilterBlock, [128, 0.1, 1.0], types.ComplexFloat32, FS), make(lr.DownsamplerBlock, [5], types.ComplexFloat32, FS)])
        d_whole = dec.process(x)
        dec2 = lr.Chain([make(lr.LowpassFilterBlock, [128, 0.1, 1.0], types.ComplexFloat32, FS), make(lr.DownsamplerBlock, [5], types.ComplexFloat32, FS)])
        d_small = np.concatenate([dec2.process(x[a:a + 99991]) for a in range(0, n, 99991)])
        assert len(d_whole) == len(d_small) == (n + 4) // 5 and np.array_equal(d_whole, d_small)
    finally:
        assert L.lrhip_host_unregister(x.ctypes.data_as(C.c_void_p)) == 0
        assert L.lrhip_host_unregister(y.ctypes.data_as(C.c_void_p)) == 0
    assert L.lrhip_host_unregister(x.ctypes.data_as(C.c_void_p)) < 0                        # not registered any more


def test_registered_vectors_are_read_and_written_in_place():
    """Round 5, host_execute's direct mode (luaradio_amd/csrc/chain.h): with both vectors registered, a stage or chain whose first kernel reads its input once
    and whose last kernel writes its output once is handed the caller's HOST memory itself - the overlap-save kernel then runs a short persistent grid.  Same
    kernels on the same values: the bits of the device-resident run, for a filter at three sizes (one block, part of a grid, many rounds of the grid) and for
    the single-launch receiver."""
    import ctypes as C
    import torch
    if os.environ.get("LRHIP_HOST_DIRECT") == "0":
        pytest.skip("LRHIP_HOST_DIRECT=0 (A/B knob): the staged piece pipeline, whose overlap-save pieces are chunks of their own")
    L = _lib.load()
    rng = np.random.default_rng(77)
    n = (1 << 22) + 12345
    x = _aligned(n, np.complex64)
    x[:] = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    y = _aligned(n, np.complex64)
    audio = _aligned(n // 25 + 64, np.float32)
    for v in (x, y, audio):
        _lib.check(L.lrhip_host_register(v.ctypes.data_as(C.c_void_p), v.nbytes), "register")
    try:
        xd = torch.from_numpy(x.view(np.float32).copy()).cuda()
        for m in (900, 200001, n):
            blk = make(lr.LowpassFilterBlock, [128, 15e3], types.ComplexFloat32, FS)
            got = L.lrhip_stage_execute(blk.stage_handle(), x.ctypes.data_as(C.c_void_p), m, y.ctypes.data_as(C.c_void_p), m)
            assert got == m, _lib.last_error()
            ref = make(lr.LowpassFilterBlock, [128, 15e3], types.ComplexFloat32, FS)
            yd = torch.empty(2 * m, device="cuda")
            assert ref.process_device(xd.data_ptr(), m, yd.data_ptr(), m) == m
            torch.cuda.synchronize()
            assert np.array_equal(y[:m].view(np.float32), yd.cpu().numpy()), m
        # vectors that start inside the registered ranges, off the 16-byte grid (a pipe's read position, radio/core/pipe.lua:131-160)
        for off_in, off_out, m in ((5, 3, 100001), (1, 0, 4097), (2, 7, 1)):
            blk = make(lr.LowpassFilterBlock, [128, 15e3], types.ComplexFloat32, FS)
            got = L.lrhip_stage_execute(blk.stage_handle(), C.c_void_p(x.ctypes.data + 8 * off_in), m, C.c_void_p(y.ctypes.data + 8 * off_out), m)
            assert got == m, _lib.last_error()
            ref = make(lr.LowpassFilterBlock, [128, 15e3], types.ComplexFloat32, FS)
            yd = torch.empty(2 * m, device="cuda")
            assert ref.process_device(xd.data_ptr() + 8 * off_in, m, yd.data_ptr(), m) == m
            torch.cuda.synchronize()
            assert np.array_equal(y[off_out:off_out + m].view(np.float32), yd.cpu().numpy()), (off_in, off_out, m)
        # the reference's block-emission framing (use_fft = true, firfilter.lua:361-398) copies its input into a pending buffer first - it is not a
        # read-once form and stays on the staged path; registered vectors give the values of the device-resident run all the same
        def framed():
            b = lr.LowpassFilterBlock(128, 15e3)
            b.use_fft = lr.block.fir_mode(True)
            b.rate = FS
            b.differentiate([types.ComplexFloat32])
            b.initialize()
            return b
        blk, ref, m = framed(), framed(), 300001
        got = L.lrhip_stage_execute(blk.stage_handle(), x.ctypes.data_as(C.c_void_p), m, y.ctypes.data_as(C.c_void_p), m)
        assert got > 0 and got <= m, _lib.last_error()
        yd = torch.empty(2 * m, device="cuda")
        assert ref.process_device(xd.data_ptr(), m, yd.data_ptr(), m) == got
        torch.cuda.synchronize()
        assert np.array_equal(y[:got].view(np.float32), yd.cpu().numpy()[:2 * got])
        # round 6: 4 098 taps and more run as TWO launches, the second adding to y - not a write-once form: registered vectors take the staged path (one piece
        # at this size: the same launches as the device-resident run, hence its bits)
        rngt = np.random.default_rng(5)
        long_taps = rngt.uniform(-1, 1, 5000).astype(np.float32)
        long_taps /= np.sum(np.abs(long_taps))
        blk, ref, m = make(lr.FIRFilterBlock, [long_taps, "fast"], types.ComplexFloat32, FS), make(lr.FIRFilterBlock, [long_taps, "fast"], types.ComplexFloat32, FS), 300001
        got = L.lrhip_stage_execute(blk.stage_handle(), x.ctypes.data_as(C.c_void_p), m, y.ctypes.data_as(C.c_void_p), m)
        assert got == m, _lib.last_error()
        yd = torch.empty(2 * m, device="cuda")
        assert ref.process_device(xd.data_ptr(), m, yd.data_ptr(), m) == m
        torch.cuda.synchronize()
        assert np.array_equal(y[:m].view(np.float32), yd.cpu().numpy())
        # an in-place call (output vector == input vector, or overlapping slices of one registered buffer) takes the staged path, where it is
        # safe - in direct mode tiles of the persistent grid would store over samples other tiles have not loaded yet
        for shift in (0, 100, -100):
            m = 1 << 21
            z = _aligned(m + 4096, np.complex64)
            z[:] = x[:len(z)]
            _lib.check(L.lrhip_host_register(z.ctypes.data_as(C.c_void_p), z.nbytes), "register")
            try:
                blk = make(lr.LowpassFilterBlock, [128, 15e3], types.ComplexFloat32, FS)
                src, dst = 2048, 2048 + shift
                got = L.lrhip_stage_execute(blk.stage_handle(), C.c_void_p(z.ctypes.data + 8 * src), m, C.c_void_p(z.ctypes.data + 8 * dst), m)
                assert got == m, _lib.last_error()
                ref = make(lr.LowpassFilterBlock, [128, 15e3], types.ComplexFloat32, FS)
                yd = torch.empty(2 * m, device="cuda")
                assert ref.process_device(xd.data_ptr() + 8 * src, m, yd.data_ptr(), m) == m
                torch.cuda.synchronize()
                assert np.array_equal(z[dst:dst + m].view(np.float32), yd.cpu().numpy()), shift
            finally:
                assert L.lrhip_host_unregister(z.ctypes.data_as(C.c_void_p)) == 0
        rx = lr.Chain(receiver_blocks())
        got = L.lrhip_chain_execute(rx._chain, x.ctypes.data_as(C.c_void_p), n, audio.ctypes.data_as(C.c_void_p), len(audio))
        assert got == (n + 24) // 25, _lib.last_error()
        want = lr.Chain(receiver_blocks())
        ad = torch.empty(len(audio), device="cuda")
        assert want.process_device(xd.data_ptr(), n, ad.data_ptr(), len(audio)) == got
        torch.cuda.synchronize()
        assert np.array_equal(audio[:got], ad.cpu().numpy()[:got])
    finally:
        for v in (x, y, audio):
            assert L.lrhip_host_unregister(v.ctypes.data_as(C.c_void_p)) == 0


def test_single_pass_element_wise_stages_take_registered_vectors_in_place():
    """Round 6: single-pass stages join host_execute's direct mode where it was measured to pay (FrequencyTranslator, Downsampler, the complex -> real
    unary blocks; MultiplyConstant, conjugate, Upsampler stay on the staged pipeline - tools/ab_direct_elem.py): with both vectors registered the kernel
    loads and stores the caller's host memory across the link - no staging, no pieces.  Either way the same kernels run on the same values: the bits of
    the device-resident run, also off the 16-byte grid (the one-sample kernels) and across calls (the rotator's phase, the downsampler's index carry on)."""
    import torch
    if os.environ.get("LRHIP_HOST_DIRECT") == "0":
        pytest.skip("LRHIP_HOST_DIRECT=0 (A/B knob)")
    L = _lib.load()
    rng = np.random.default_rng(78)
    n = (1 << 21) + 777
    x = _aligned(n, np.complex64)
    x[:] = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    y = _aligned(3 * n + 64, np.complex64)
    for v in (x, y):
        _lib.check(L.lrhip_host_register(v.ctypes.data_as(C.c_void_p), v.nbytes), "register")
    c = types.ComplexFloat32
    try:
        xd = torch.from_numpy(x.view(np.float32).copy()).cuda()
        cases = [(lr.FrequencyTranslatorBlock, [-250e3]), (lr.DownsamplerBlock, [5]), (lr.MultiplyConstantBlock, [0.5]), (lr.ComplexMagnitudeBlock, []),
                 (lr.ComplexConjugateBlock, []), (lr.UpsamplerBlock, [3])]
        for cls, args in cases:
            for off in (0, 3):
                blk, ref = make(cls, args, c, FS), make(cls, args, c, FS)
                q = blk.stage_handle()
                osz = L.lrhip_stage_output_size(q)
                done_out = 0
                yd = torch.zeros(2 * (3 * n + 64), device="cuda")
                for a, m in ((off, 100001), (off + 100001, 4097), (off + 104098, n - 104098 - 3)):
                    cap = L.lrhip_stage_max_output(q, m)
                    got = L.lrhip_stage_execute(q, C.c_void_p(x.ctypes.data + 8 * a), m, C.c_void_p(y.ctypes.data + osz * done_out), cap)
                    assert got >= 0, _lib.last_error()
                    want = ref.process_device(xd.data_ptr() + 8 * a, m, yd.data_ptr() + osz * done_out, cap)
                    assert got == want, (cls.__name__, off, got, want)
                    done_out += got
                torch.cuda.synchronize()
                host = y.view(np.uint8)[:osz * done_out]
                dev = yd.cpu().numpy().view(np.uint8)[:osz * done_out]
                assert np.array_equal(host, dev), (cls.__name__, off)
    finally:
        for v in (x, y):
            assert L.lrhip_host_unregister(v.ctypes.data_as(C.c_void_p)) == 0


def test_poll_due_stays_bounded_while_a_launched_batch_is_in_flight():
    """batch == chunk - push() launches the full batch, its non-waiting collect finds it unfinished, then the source stalls.  Nothing is
    accumulating, but the batch's output still has to reach the host: poll_due() keeps the wait for input bounded until the ring has drained, and poll()
    hands the finished batch out without another push and without EOF"""
    x = fm_signal(50000, seed=19)
    want = lr.Chain(receiver_blocks()).process(x)
    chain = lr.Chain(receiver_blocks())
    chain.set_ring(3, 50000)                                            # a batch is exactly one vector
    chain.set_latency(0.02)
    first = chain.push(x)                                               # launched at once; usually not finished when push() looks
    if len(first) == 0:
        due = chain.poll_due()
        assert 0.0 <= due <= 0.02, due                                  # was -1 (wait for input forever): the audio sat in the ring until more input came
        got, t0 = first, time.monotonic()
        while len(got) == 0 and time.monotonic() - t0 < 5.0:
            time.sleep(max(chain.poll_due(), 0.0) if chain.poll_due() >= 0 else 0.001)
            got = chain.poll()
    else:
        got = first
    assert len(got) == len(want) and float(np.max(np.abs(got - want))) < 5e-5
    assert chain.poll_due() == -1.0                                     # drained: wait for input again


def test_two_host_threads_on_two_objects_share_the_piece_pipeline_safely():
    """the piece-wise host path has ONE set of copy streams and events per process.  Two host threads (ctypes releases the GIL around
    the call) running big vectors through two different stages must not record and wait on each other's events: the thread that does not get the pipeline
    takes the single-piece path.  Results are those of the sequential runs, bit for bit, over several rounds"""
    import threading
    rng = np.random.default_rng(41)
    n = (1 << 21) + 12345
    xs = [(rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) for _ in range(2)]
    taps = [(rng.uniform(-1, 1, 128) / 128).astype(np.float32) for _ in range(2)]
    want = [make(lr.FIRFilterBlock, [taps[k], False], types.ComplexFloat32, FS).process(xs[k]) for k in range(2)]
    for _ in range(4):
        blks = [make(lr.FIRFilterBlock, [taps[k], False], types.ComplexFloat32, FS) for k in range(2)]
        got = [None, None]
        start = threading.Barrier(2)

        def run(k):
            start.wait()
            got[k] = blks[k].process(xs[k])
        th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        assert all(g is not None for g in got)
        for k in range(2):
            assert np.array_equal(got[k], want[k]), k


def shaped_receiver_blocks(rate, decim, audio_taps=128):
    """the receiver of examples/rtlsdr_wbfm_mono.lua at another input rate / tuner decimation / audio tap count; every use_fft left to the library"""
    c, f = types.ComplexFloat32, types.Float32
    r1 = rate / decim
    return [make(lr.FrequencyTranslatorBlock, [-250e3], c, rate), make(lr.LowpassFilterBlock, [128, 100e3], c, rate),
            make(lr.DownsamplerBlock, [decim], c, rate), make(lr.FrequencyDiscriminatorBlock, [1.25], c, r1),
            make(lr.LowpassFilterBlock, [audio_taps, 15e3], f, r1), make(lr.FMDeemphasisFilterBlock, [75e-6], f, r1),
            make(lr.DownsamplerBlock, [5], f, r1)]


def shaped_fm_signal(rate, n, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / rate
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / rate * np.cumsum(m)
    return (np.exp(1j * ph) + 0.01 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)


@pytest.mark.parametrize("decim,audio_taps,max_launches", [(4, 128, 3), (8, 128, 3), (10, 128, 3), (5, 96, 5), (10, 160, 5)])
def test_fm_receivers_off_the_stock_shape(decim, audio_taps, max_launches):
    """the receiver at other tuner decimations (input rates 0.882 / 1.764 / 2.205 MS/s) and audio tap counts.  Round 5 gives the Toeplitz tuner
    its discriminator epilogue at decimation 4, 8 and 10 too (tuner + discriminator = one launch instead of four).  Parity as for the stock shape: RMS <= 1e-5
    against the oracle's chain (measured ~2e-8), for one chunk and for ragged chunks; the exact chain gives the bits of the blocks run one by one; and a
    recording that starts ON the real axis - where the first angle is decided by the signs of zeros - gets the reference's first sample"""
    rate = 220500.0 * decim
    n = 300000
    x = shaped_fm_signal(rate, n)
    bb, aa = O.fm_deemphasis_taps(75e-6, rate / decim)
    ora = O.Chain(O.tuner(-250e3, 200e3, decim, rate, mode=O.MODE_LUA, rot_mode=O.MODE_F64).stages +
                  [O.FMDiscriminator(1.25), O.lowpass(audio_taps, 15e3, rate / decim, False, mode=O.MODE_LUA), O.IIR(bb, aa, False, O.MODE_LUA), O.Downsampler(5, False)])
    want = ora.process(x)
    chain = lr.Chain(shaped_receiver_blocks(rate, decim, audio_taps))
    got = chain.process(x)
    assert chain.last_launches <= max_launches, chain.last_launches
    assert len(got) == len(want)
    assert float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))) <= 1e-5
    assert float(np.max(np.abs(got.astype(np.float64) - want))) < 1e-5
    ragged = lr.Chain(shaped_receiver_blocks(rate, decim, audio_taps))
    got_r = run_chunked(ragged.process, x, [1, 7, 8192, 8193, 100000, 100001, 222222])
    assert len(got_r) == len(want) and float(np.max(np.abs(got_r.astype(np.float64) - want))) < 1e-5
    # the exact chain: what the blocks compute one by one, bit for bit, on the same chunks
    cuts = [8192, 100001]
    blocks = shaped_receiver_blocks(rate, decim, audio_taps)

    def one_by_one(v):
        for b in blocks:
            v = b.process(v)
        return v
    exact = lr.Chain(shaped_receiver_blocks(rate, decim, audio_taps), exact=True)
    assert np.array_equal(run_chunked(exact.process, x, cuts), run_chunked(one_by_one, x, cuts))
    # tuner + discriminator alone on a carrier 10 kHz off the tuned frequency that starts at phase 0 (x[0] = 1 + 0j): sample 0 is arg(y[0] conj(0)) - zeros
    # whose signs the reference's arithmetic decides (0 here for every decimation, also where the filter's first tap is negative: decimation 8)
    t = np.arange(4096) / rate
    clean = np.exp(1j * 2 * np.pi * 260e3 * t).astype(np.complex64)
    td = lr.Chain(shaped_receiver_blocks(rate, decim, audio_taps)[:4])
    ang = td.process(clean)
    want_ang = O.Chain(O.tuner(-250e3, 200e3, decim, rate, mode=O.MODE_LUA, rot_mode=O.MODE_F64).stages + [O.FMDiscriminator(1.25)]).process(clean)
    assert float(ang[0]) == float(want_ang[0]) == 0.0
    # (the first outputs are 1e-4 of full scale - the filter has just started - so their angles carry the 1e-7 rounding of the filter outputs magnified)
    assert float(np.max(np.abs(ang[16:] - want_ang[16:]))) < 1e-5
