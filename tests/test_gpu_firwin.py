"""GPU parity of the register-window FIR kernels (luaradio_amd/csrc/kernels_firwin.h) through the C ABI.

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


def _tail_blocks(ntaps, factor, rate=220500.0, tau=75e-6):
    taps = O.firwin_lowpass(ntaps, 0.2).astype(np.float32)
    x0 = np.zeros(1, np.float32)
    fir = make(lr.FIRFilterBlock, [taps, "auto"], x0, rate=rate)
    iir = make(lr.FMDeemphasisFilterBlock, [tau], x0, rate=rate)
    blocks = [fir, iir]
    if factor > 1:
        blocks.append(make(lr.DownsamplerBlock, [factor], x0, rate=rate))
    return taps, blocks


@pytest.mark.parametrize("ntaps,factor", [(128, 5), (128, 1), (64, 3), (32, 4099), (128, 16), (64, 17)])
def test_fir_iir_downsampler_fused_vs_oracle_and_unfused(ntaps, factor):
    rng = np.random.default_rng(100 + ntaps + factor)
    n = 7 * 4096 + 333
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps, blocks = _tail_blocks(ntaps, factor)
    chain = lr.Chain(blocks)
    whole = chain.process(x)
    # factor 2..16: polyphase form - decimating filter g = h * b * (1, p, .., p^(D-1)) with the low-rate recurrence on its accumulators (one
    # launch for 128 taps at factor 5: register-window kernel) or behind it; other factors: the blocks' own kernels, iir + downsampler fused
    assert chain.last_launches <= (1 if (ntaps, factor) == (128, 5) else 4)
    b, a = O.fm_deemphasis_taps(75e-6, 220500.0)
    for mode in (O.MODE_LUA, O.MODE_F64):
        want = O.IIR(b, a, False, mode).process(O.FIR(taps, False, O.MODE_FMA).process(x))[::factor]
        assert len(whole) == len(want)
        assert G.max_abs_err(whole, want) < 1e-6, mode
    # the same blocks one by one on the device (direct-form filter)
    taps2, ref = _tail_blocks(ntaps, factor)
    ref[0] = make(lr.FIRFilterBlock, [taps2], np.zeros(1, np.float32), rate=220500.0)
    y = x
    for blk in ref:
        y = blk.process(y)
    assert len(y) == len(whole) and G.max_abs_err(whole, y) < 1e-6
    # chunking must not matter (warm-up tiles, carried filter history, recurrence state, downsampler index)
    for cuts in ([1], [1, 2, 4095, 4096, 4097, 12288, 12289], [20000], list(range(1, 30)) + [4096 * 3]):
        chain.reset()
        got = chunked(chain, x, cuts)
        assert len(got) == len(whole)
        assert G.max_abs_err(got, whole) < 5e-7, cuts


def test_fir_iir_fused_many_tiles_many_workgroups():
    """enough tiles that workgroups start in the middle of the stream from a warm-up tile: compare with the oracle on slabs"""
    rng = np.random.default_rng(8)
    n = 1 << 23
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps, blocks = _tail_blocks(128, 5)
    chain = lr.Chain(blocks)
    got = chain.process(x)
    assert len(got) == (n + 4) // 5
    b, a = O.fm_deemphasis_taps(75e-6, 220500.0)
    warm = 20000
    for s0 in (0, 1 << 20, 3 * (1 << 20) + 4095, n - 300000):
        s0 = s0 // 5 * 5
        lo = max(0, s0 - warm)
        seg = x[lo:s0 + 250000]
        want = O.IIR(b, a, False, O.MODE_F64).process(O.FIR(taps, False, O.MODE_FMA).process(seg))[(s0 - lo)::5]
        g = got[s0 // 5:s0 // 5 + len(want)]
        assert G.max_abs_err(g, want[:len(g)]) < 1e-6, s0


@pytest.mark.parametrize("two_launch", [True, False])
def test_wbfm_receiver_launch_forms(two_launch):
    """round 3: ONE launch (kernels_rx.h) by default; LRHIP_CHAIN_NO_SINGLE_LAUNCH keeps the round-2 form - tuner + discriminator (Toeplitz MFMA)
    and the audio tail (pair-mode window kernel, which also applies the tuner's wave-boundary fix-up while it stages its window).  Ragged chunks
    exercise the carried state and edge tiles of both, against the oracle chain"""
    from luaradio_amd import _lib
    fs = 1102500.0
    rx = lr.wbfm_mono_receiver(fs, -250e3)
    if two_launch:
        rx._chain = lr.Chain(rx._blocks, _lib.CHAIN_NO_SINGLE_LAUNCH)
    rng = np.random.default_rng(9)
    n = 1 << 19
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    got = chunked(rx, x, [1, 2, 1281, 70001, 70002, 300000])
    assert rx.chain.last_launches == (2 if two_launch else 1)
    want = O.wbfm_mono_chain(fs, -250e3, mode=O.MODE_LUA, rot_mode=O.MODE_F64).process(x)
    assert len(got) == len(want)
    err = got.astype(np.float64) - want.astype(np.float64)
    assert float(np.sqrt(np.mean(err ** 2))) <= 1e-5 and float(np.max(np.abs(err))) < 1e-4


def test_complex_window_kernel_chains_bit_equal_in_a_child_process():
    """the ComplexFloat32 register-window kernel is opt-in (LRHIP_FIR_WIN_CPLX=1, read once per process): tests/helpers/winc_check.py"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, LRHIP_FIR_WIN_CPLX="1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "winc_check.py")
    r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "winc ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_optin_float32_window_kernels_in_a_child_pytest():
    """fir_win_real_kernel (Float32 stream, D = 1, with and without the in-kernel recurrence) is opt-in - the Toeplitz-MFMA / overlap-save
    kernels are faster on MI355X for these shapes (tools/ab_firwin.py) - so its parity cases run in a child pytest with the knobs set"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, LRHIP_FIR_WIN_REAL="1", LRHIP_FIR_IIR_WIN="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "helpers", "firwin_real_cases.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("cplx", [True, False])
def test_toeplitz_kernel_nonfinite_sample_poisons_whole_output_blocks(cplx):
    """Documented deviation of the Toeplitz-MFMA direct form: a structural zero of the banded matrix times Inf is NaN, so a
    non-finite input sample makes every output of the 16-output blocks whose staged window holds it non-finite - a superset of the M outputs
    the reference's dot products (firfilter.lua:266-305) would poison, confined to a few dozen samples around them; everything else is the
    bit-exact fmaf chain."""
    rng = np.random.default_rng(6 + cplx)
    n, M, pos = 20000, 128, 9001
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) if cplx else rng.uniform(-1, 1, n).astype(np.float32)
    x[pos] = np.inf
    taps = O.firwin_lowpass(M, 0.3).astype(np.float32)
    blk = make(lr.FIRFilterBlock, [taps], x)
    got = blk.process(x)
    want = O.FIR(taps, cplx, O.MODE_FMA).process(x)
    bad_ref, bad_got = ~np.isfinite(want), ~np.isfinite(got)
    assert bad_ref.sum() == M and np.all(bad_got[bad_ref])                      # the reference's M poisoned outputs are poisoned here too
    extra = np.flatnonzero(bad_got & ~bad_ref)
    assert extra.size <= 96 and (extra.size == 0 or (extra.min() >= pos - 48 and extra.max() <= pos + M + 48))
    assert np.array_equal(got[~bad_got], want[~bad_got])


def test_slow_recurrence_is_not_rewritten_in_polyphase_form():
    """The polyphase identity moves the pole to p^D; rounded to one Float32 that shifts the DC gain by 2^-24 q / (1 - q) - for a slow filter
    (50 Hz single-pole lowpass at 220.5 kHz: p = 0.9986) two orders above the recurrence's 1e-6 bar.  Such a chain keeps the blocks' own
    kernels (the fused window form carries the pole as a Float32 pair but needs the recurrence to decay within its warm-up)."""
    rng = np.random.default_rng(31)
    n = 5 * 4096 + 77
    x = (rng.uniform(-1, 1, n) + 0.5).astype(np.float32)            # with a DC component: the gain error would show
    taps = O.firwin_lowpass(128, 0.2).astype(np.float32)
    x0 = np.zeros(1, np.float32)
    blocks = [make(lr.FIRFilterBlock, [taps, "auto"], x0, rate=220500.0), make(lr.SinglepoleLowpassFilterBlock, [50.0], x0, rate=220500.0),
              make(lr.DownsamplerBlock, [5], x0, rate=220500.0)]
    chain = lr.Chain(blocks)
    got = chain.process(x)
    assert chain.last_launches >= 2
    b, a = O.singlepole_lowpass_taps(50.0, 220500.0)
    want = O.IIR(b, a, False, O.MODE_F64).process(O.FIR(taps, False, O.MODE_F64).process(x))[::5]
    assert len(got) == len(want) and G.max_abs_err(got, want) < 2e-6


@pytest.mark.parametrize("cplx,ntaps", [(True, 16), (True, 32), (True, 64), (False, 16), (False, 32), ("ctaps", 16), ("ctaps", 32)])
def test_short_filter_streaming_kernels_tile_edges_and_unaligned_pointers(cplx, ntaps):
    """the one-shot kernels for short filters (ComplexFloat32: register-window kernel, a workgroup per 1280-output tile, real or ComplexFloat32
    taps; Float32: four outputs per thread): sizes around the tile / vector boundaries, device pointers off the 16-byte grid, one sample per call -
    bit-exact fmaf chains"""
    import torch
    ctaps = cplx == "ctaps"
    cplx = bool(cplx)
    rng = np.random.default_rng(ntaps + cplx + 7 * ctaps)
    n = 5 * 1280 + 3
    mk = lambda m: ((rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)).astype(np.complex64) if cplx else rng.uniform(-1, 1, m).astype(np.float32))
    x = mk(n + 8)
    taps = (mk(ntaps) / ntaps).astype(np.complex64) if ctaps else (rng.uniform(-1, 1, ntaps) / ntaps).astype(np.float32)
    for m in (1, 3, 4, 5, 1279, 1280, 1281, n):
        blk = make(lr.FIRFilterBlock, [taps], x)
        assert np.array_equal(blk.process(x[:m]), O.FIR(taps, cplx, O.MODE_FMA).process(x[:m])), m
    blk = make(lr.FIRFilterBlock, [taps], x)
    got = np.concatenate([blk.process(x[i:i + 1]) for i in range(200)])
    assert np.array_equal(got, O.FIR(taps, cplx, O.MODE_FMA).process(x[:200]))
    es = 8 if cplx else 4
    xd = torch.from_numpy(x.view(np.float32)).cuda()
    yd = torch.zeros((n + 16) * (2 if cplx else 1), dtype=torch.float32, device="cuda")
    for off_in, off_out in ((1, 0), (0, 1), (3, 2)):
        blk = make(lr.FIRFilterBlock, [taps], x)
        assert blk.process_device(xd.data_ptr() + es * off_in, n, yd.data_ptr() + es * off_out, n) == n
        torch.cuda.synchronize()
        out = yd.cpu().numpy()
        out = out.view(np.complex64) if cplx else out
        assert np.array_equal(out[off_out:off_out + n], O.FIR(taps, cplx, O.MODE_FMA).process(x[off_in:off_in + n])), (off_in, off_out)
