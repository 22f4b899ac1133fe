"""(run by tests/test_gpu_firwin.py in a child pytest with LRHIP_FIR_WIN_REAL=1 LRHIP_FIR_IIR_WIN=1: the knobs are read once per process)
GPU parity of the register-window FIR kernels (luaradio_amd/csrc/kernels_firwin.h) through the C ABI.

Float32-stream FIRFilterBlock: bit-exact against the oracle's fmaf-chain mode (firfilter.lua:288-305 order) for every tap count with
an instantiation, ragged chunkings, tile edges and unaligned device pointers.  Fused FIR -> first-order IIR -> Downsampler (the audio
tail of examples/rtlsdr_wbfm_mono.lua:14-16): <= 1e-6 against the oracle blocks (iirfilter.lua:113-181, downsampler.lua:45-56) and
against the same blocks run one by one on the device; identical values for any chunking down to one sample per call."""
import numpy as np
import pytest

import luaradio_amd as lr
from luaradio_amd import types
from oracle import oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu


def make(cls, args, x, rate=2.0):
    blk = cls(*args)
    blk.rate = rate
    blk.differentiate([types.type_of(x)])
    blk.initialize()
    return blk


def chunked(blk, x, cuts):
    parts, a = [], 0
    for b in list(cuts) + [len(x)]:
        parts.append(blk.process(x[a:b]))
        a = b
    return np.concatenate(parts)


@pytest.mark.parametrize("ntaps", [32, 64, 128])
def test_real_fir_window_kernel_bit_exact(ntaps):
    rng = np.random.default_rng(ntaps)
    n = 3 * 4096 + 1234
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
    want = O.FIR(taps, False, O.MODE_FMA).process(x)
    blk = make(lr.FIRFilterBlock, [taps], x)
    assert np.array_equal(blk.process(x), want)
    for cuts in ([1], [1, 2, 3, 4095, 4096, 4097, 8192], [5000], [4096, 8192, 12288], list(range(1, 40))):
        blk.reset()
        assert np.array_equal(chunked(blk, x, cuts), want), cuts
    # one sample per call (tests/jigs.lua:213-250) on a prefix
    blk.reset()
    got = np.concatenate([blk.process(x[i:i + 1]) for i in range(300)])
    assert np.array_equal(got, want[:300])

def test_real_fir_window_kernel_unaligned_device_pointers():
    import torch
    rng = np.random.default_rng(5)
    n = 2 * 4096 + 77
    x = rng.uniform(-1, 1, n + 8).astype(np.float32)
    taps = O.firwin_lowpass(128, 0.2).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros(n + 16, dtype=torch.float32, device="cuda")
    for off_in, off_out in ((0, 0), (1, 0), (2, 3), (3, 1)):
        blk = make(lr.FIRFilterBlock, [taps], x)
        got_n = blk.process_device(xd.data_ptr() + 4 * off_in, n, yd.data_ptr() + 4 * off_out, n)
        torch.cuda.synchronize()
        assert got_n == n
        want = O.FIR(taps, False, O.MODE_FMA).process(x[off_in:off_in + n])
        assert np.array_equal(yd[off_out:off_out + n].cpu().numpy(), want), (off_in, off_out)

def test_real_fir_nonfinite_sample_reaches_exactly_its_window():
    """firfilter.lua:288-305: an Inf / NaN input sample contributes to the M outputs whose window holds it and to no other"""
    rng = np.random.default_rng(6)
    n, M = 9000, 128
    x = rng.uniform(-1, 1, n).astype(np.float32)
    x[5000] = np.inf
    x[7001] = np.nan
    taps = O.firwin_lowpass(M, 0.3).astype(np.float32)
    blk = make(lr.FIRFilterBlock, [taps], x)
    got = blk.process(x)
    want = O.FIR(taps, False, O.MODE_FMA).process(x)
    bad = ~np.isfinite(want)
    assert np.array_equal(~np.isfinite(got), bad)
    assert np.array_equal(got[~bad], want[~bad])
    assert bad.sum() <= 2 * M


def _tail_blocks(ntaps, factor, rate=220500.0, tau=75e-6):
    taps = O.firwin_lowpass(ntaps, 0.2).astype(np.float32)
    x0 = np.zeros(1, np.float32)
    fir = make(lr.FIRFilterBlock, [taps, "auto"], x0, rate=rate)
    iir = make(lr.FMDeemphasisFilterBlock, [tau], x0, rate=rate)
    blocks = [fir, iir]
    if factor > 1:
        blocks.append(make(lr.DownsamplerBlock, [factor], x0, rate=rate))
    return taps, blocks


def test_fir_iir_fused_in_kernel_recurrence_path_many_tiles(monkeypatch):
    """the one-launch form (recurrence on the accumulators, downsampler in the store) at a decimation the polyphase form does not take"""
    rng = np.random.default_rng(18)
    n = 1 << 21
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps, blocks = _tail_blocks(128, 17)
    chain = lr.Chain(blocks)
    got = chain.process(x)
    assert chain.last_launches == 1
    b, a = O.fm_deemphasis_taps(75e-6, 220500.0)
    want = O.IIR(b, a, False, O.MODE_F64).process(O.FIR(taps, False, O.MODE_FMA).process(x))[::17]
    assert len(got) == len(want) and G.max_abs_err(got, want) < 1e-6


@pytest.mark.parametrize("ntaps,factor", [(128, 1), (32, 4099), (64, 17)])
def test_fir_iir_in_kernel_recurrence_vs_oracle_and_chunkings(ntaps, factor):
    rng = np.random.default_rng(100 + ntaps + factor)
    n = 7 * 4096 + 333
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps, blocks = _tail_blocks(ntaps, factor)
    chain = lr.Chain(blocks)
    whole = chain.process(x)
    assert chain.last_launches == 1
    b, a = O.fm_deemphasis_taps(75e-6, 220500.0)
    for mode in (O.MODE_LUA, O.MODE_F64):
        want = O.IIR(b, a, False, mode).process(O.FIR(taps, False, O.MODE_FMA).process(x))[::factor]
        assert len(whole) == len(want) and G.max_abs_err(whole, want) < 1e-6, mode
    for cuts in ([1], [1, 2, 4095, 4096, 4097, 12288, 12289], [20000], list(range(1, 30)) + [4096 * 3]):
        chain.reset()
        got = chunked(chain, x, cuts)
        assert len(got) == len(whole) and G.max_abs_err(got, whole) < 5e-7, cuts
