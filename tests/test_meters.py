"""ZeroSource / RawFileSource / BenchmarkSink and the benchmark suite's trial protocol (luaradio_amd/meters.py), and the
Welch / waterfall averaging of the spectrum sinks pinned to the reference's PSD vectors.
CPU: RawFileSource framing, BenchmarkSink report / JSON, trial statistics.  GPU: ZeroSource's resident vector, Welch and waterfall."""
import io
import json

import numpy as np
import pytest

from luaradio_amd import meters, types
from tests import golden_util as G


def test_rawfile_source_elements_partial_tail_and_eof():
    """rawfile.lua:77-108 (and tests/blocks/sources/rawfile_spec.lua): whole elements only, a partial element waits for more bytes"""
    x = (np.arange(100000) + 1j * np.arange(100000)).astype(np.complex64)
    raw = x.tobytes() + b"\x01\x02\x03"                      # 3 stray bytes: never a whole element
    src = meters.RawFileSource(raw, types.ComplexFloat32, 1.0)
    src.initialize()
    assert src.get_rate() == 1.0 and src.get_output_type() is types.ComplexFloat32
    parts = []
    while True:
        v = src.process()
        if v is None:
            break
        parts.append(v)
    assert len(parts[0]) == 262144 // 8                      # buffer capacity in elements
    assert np.array_equal(np.concatenate(parts), x)
    r = meters.RawFileSource(np.float32([1, 2, 3]).tobytes(), types.Float32, 5.0, True)     # repeat_on_eof
    r.initialize()
    got = np.concatenate([r.process() for _ in range(4)])
    assert np.array_equal(got[:6], np.float32([1, 2, 3, 1, 2, 3]))
    with pytest.raises(AssertionError):
        meters.RawFileSource(None, types.Float32, 1.0)


def test_benchmark_sink_report_and_json():
    clock = iter([0.0, 1.0, 4.0, 4.5, 9.0])                  # initialize, then one reading per process()
    out = io.StringIO()
    snk = meters.BenchmarkSink(out, False, "fir", clock=lambda: next(clock))
    snk.differentiate([types.ComplexFloat32])
    snk.initialize()
    snk.process(np.zeros(1000, np.complex64))               # t = 1: below the 3 s report period
    assert out.getvalue() == ""
    snk.process(7000)                                        # t = 4: 8000 samples in 4 s
    assert out.getvalue() == "[fir] 2.00 KS/s (16.00 KB/s)\n"            # benchmark.lua:104
    snk.process(10)
    assert out.getvalue().count("\n") == 1
    t = iter([10.0, 12.0])
    j = io.StringIO()
    s2 = meters.BenchmarkSink(j, True, clock=lambda: next(t))
    s2.differentiate([types.Float32])
    s2.initialize()
    s2.process(np.zeros(5000, np.float32))
    s2.cleanup()
    assert json.loads(j.getvalue()) == {"samples_per_second": 2500.0, "bytes_per_second": 10000.0}   # benchmark.lua:127-131
    assert meters._normalize(2.5e9) == (2.5, "G") and meters._normalize(999.0) == (999.0, "")


def test_trial_protocol_mean_and_population_sigma():
    """luaradio_benchmark.lua:690-738: five trials, mean and sqrt(sum((x - mean)^2) / N)"""
    rates = iter([100.0, 110.0, 90.0, 105.0, 95.0])

    def make_top(results):
        rate = next(rates)
        t = iter([0.0, 1.0])
        snk = meters.BenchmarkSink(results, True, clock=lambda: next(t))
        snk.differentiate([types.Float32])
        snk.initialize()
        state = {"done": False}

        def step():
            if not state["done"]:
                snk.process(int(rate))
                state["done"] = True
        return step, snk

    r = meters.run_trials(make_top, 5, 0.01)
    assert r["trials"] == 5 and abs(r["samples_per_second"] - 100.0) < 1e-9
    assert abs(r["samples_per_second_stdev"] - np.std([100.0, 110.0, 90.0, 105.0, 95.0])) < 1e-9
    assert abs(r["bytes_per_second"] - 400.0) < 1e-9


@pytest.mark.gpu
def test_zero_source_resident_vector_feeds_a_block():
    import ctypes as C
    import luaradio_amd as lr
    src = lr.ZeroSource(types.ComplexFloat32, 1e6)
    src.initialize()
    assert src.get_rate() == 1e6 and len(src.process()) == 8192 and not src.process().any()
    ptr, n = src.process_device()
    fir = lr.LowpassFilterBlock(128, 0.1)
    fir.rate = 1e6
    fir.differentiate([types.ComplexFloat32])
    fir.initialize()
    L = lr._lib.load()
    out = lr._lib.check_ptr(L.lrhip_malloc(8 * n), "malloc")
    assert fir.process_device(ptr, n, out, n) == n
    host = np.ones(n, np.complex64)
    lr._lib.check(L.lrhip_memcpy_d2h(host.ctypes.data_as(C.c_void_p), out, 8 * n), "d2h")
    assert not host.any()
    L.lrhip_free(out)
    src.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["complex", "real"])
def test_welch_average_pinned_to_the_reference_psd_vectors(kind):
    """Every frame of the stream IS the reference's test vector (tests/utilities/spectrum_utils_vectors.gen.lua), so the Welch mean of k frames
    (gnuplotspectrum.lua:140-193: framing, PSD, fftshift, accumulate, normalise, reference level) must equal the reference's own PSD of that
    vector, fftshifted: linear PSD at the reference's 1e-5, log PSD at its 3 dB and at 2e-3 dB of 10 log10 of the linear average."""
    import luaradio_amd as lr
    v = G.load("spectrum_utils_vectors")["values"]
    x = v["%s_test_vector" % kind]
    dt = types.ComplexFloat32 if kind == "complex" else types.Float32
    n = len(x)
    stream = np.concatenate([x] * 5 + [x[:37]])              # five whole frames and a partial one
    lin = lr.spectrum_utils.WelchSpectrum(dt, n, "hamming", 44100, 0.0, 0.0, logarithmic=False)
    for a, b in ((0, 1), (1, 200), (200, 300), (300, len(stream))):
        lin.process(stream[a:b])
    got = lin.average()
    assert lin.frames == 5
    want = np.fft.fftshift(v["%s_test_vector_hamming_psd" % kind])
    assert G.max_abs_err(got, want) < 1e-5
    log = lr.spectrum_utils.WelchSpectrum(dt, n, "hamming", 44100, 0.0, 10.0)
    log.process(stream)
    got_log = log.average()
    assert G.max_abs_err(got_log + 10.0, np.fft.fftshift(v["%s_test_vector_hamming_psd_log" % kind])) < 3           # the reference's epsilon
    # (the reference's vectors carry four significant digits; the linear average just checked against them gives the tighter log comparison)
    assert G.max_abs_err(got_log + 10.0, 10 * np.log10(got.astype(np.float64))) < 2e-3
    assert log.average() is None


@pytest.mark.gpu
@pytest.mark.parametrize("is_complex,overlap,navg", [(True, 0.0, 1), (False, 0.5, 3), (True, 0.25, 2)])
def test_waterfall_rows_vs_oracle(is_complex, overlap, navg):
    """GnuplotWaterfallSink's rows (gnuplotwaterfall.lua:184-236) on the device against the restatement: the colour map is continuous, so a
    2e-3 dB difference of the averaged PSD moves a channel by at most one step"""
    import luaradio_amd as lr
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    n, N = 6000, 128
    t = np.arange(n)
    x = (rng.standard_normal(n) * 0.05 + np.cos(2 * np.pi * 0.11 * t)).astype(np.float32)
    if is_complex:
        x = (x + 1j * (rng.standard_normal(n) * 0.05 + np.sin(2 * np.pi * 0.11 * t))).astype(np.complex64)
    dev = lr.spectrum_utils.WaterfallSpectrum(types.ComplexFloat32 if is_complex else types.Float32, N, "hamming", 1e6, overlap, navg, -120.0, -20.0, rows=16)
    ora = O.Waterfall(is_complex, N, "hamming", 1e6, overlap, navg, -120.0, -20.0, rows=16)
    added, pos = 0, 0
    for size in (1, 100, 129, 1000, 7, n):
        added += dev.process(x[pos:pos + size])
        pos += size
    ora.process(x)
    assert added == ora.rows_added > 16
    d = np.abs(dev.pixels.astype(np.int32) - ora.pixels.astype(np.int32))
    assert d.max() <= 1
    assert len(np.unique(dev.pixels.reshape(-1, 3), axis=0)) > 20            # the image is not flat
    # the colour map itself, at the segment boundaries of gnuplotwaterfall.lua:151-182
    pts = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1 / 5 - 1e-12, 3 / 5 + 1e-12])
    assert np.array_equal(lr.spectrum_utils.value_to_pixel(pts), np.array([O.Waterfall.value_to_pixel(p) for p in pts], np.uint8))
