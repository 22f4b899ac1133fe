"""Pins the CPU oracle (oracle/lr_oracle.c) against the reference's own golden vectors, at the
reference's own epsilons, in both modes of tests/jigs.lua (whole vector / one sample per call)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import golden_util as G

RATE = 2.0   # tests/jigs.lua:69 monkey-patches get_rate() to 2.0


def _check(make, vec, eps):
    x, want = vec["inputs"][0], vec["outputs"][0]
    whole, samplewise = G.run_whole_and_samplewise(make, x)
    assert G.max_abs_err(whole[:len(want)], want) < eps, vec["desc"]
    assert len(whole) == len(want)
    assert G.max_abs_err(samplewise, want) < eps, vec["desc"]


@pytest.mark.parametrize("mode", [O.MODE_LUA, O.MODE_FMA, O.MODE_F64])
def test_firfilter_dotprod(mode):
    doc = G.load("firfilter_spec")
    n = 0
    for vec in doc["vectors"]:
        taps, use_fft = vec["args"]
        if use_fft:
            continue
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: O.FIR(taps, cplx, mode), vec, doc["epsilon"])
        n += 1
    assert n == 12


def test_firfilter_fft_overlap_save():
    doc = G.load("firfilter_spec")
    n = 0
    for vec in doc["vectors"]:
        taps, use_fft = vec["args"]
        if not use_fft:
            continue
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: O.FIRFFT(taps, cplx), vec, doc["epsilon"])
        n += 1
    assert n == 12


@pytest.mark.parametrize("mode", [O.MODE_LUA, O.MODE_FMA, O.MODE_F64])
def test_lowpassfilter(mode):
    doc = G.load("lowpassfilter_spec")
    for vec in doc["vectors"]:
        a = vec["args"]
        num_taps, cutoff = a[0], a[1]
        nyq = a[2] if len(a) > 2 else None
        win = a[3] if len(a) > 3 else "hamming"
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: O.lowpass(num_taps, cutoff, RATE, cplx, nyq, win, mode), vec, doc["epsilon"])


def _firwin_block(design, vec, cplx, mode=O.MODE_LUA):
    a = vec["args"]
    nyq = a[2] if len(a) > 2 and a[2] is not None else RATE / 2
    win = a[3] if len(a) > 3 else "hamming"
    c = a[1]
    c = [v / nyq for v in c] if isinstance(c, list) else c / nyq
    return O.FIR(design(a[0], c, win).astype(np.float32), cplx, mode)


@pytest.mark.parametrize("name,design", [("highpassfilter_spec", O.firwin_highpass),
                                         ("bandpassfilter_spec", O.firwin_bandpass),
                                         ("bandstopfilter_spec", O.firwin_bandstop)])
def test_other_firwin_blocks(name, design):
    doc = G.load(name)
    for vec in doc["vectors"]:
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: _firwin_block(design, vec, cplx), vec, doc["epsilon"])


@pytest.mark.parametrize("mode", [O.MODE_LUA, O.MODE_F64])
def test_frequencytranslator(mode):
    doc = G.load("frequencytranslator_spec")
    assert doc["epsilon"] == 1e-5
    for vec in doc["vectors"]:
        omega = 2 * np.pi * (vec["args"][0] / RATE)
        _check(lambda: O.Rotator(omega, mode), vec, doc["epsilon"])


def test_frequencydiscriminator():
    doc = G.load("frequencydiscriminator_spec")
    for vec in doc["vectors"]:
        _check(lambda: O.FMDiscriminator(vec["args"][0]), vec, doc["epsilon"])


def test_agc():
    doc = G.load("agc_spec")
    assert len(doc["vectors"]) == 6
    for vec in doc["vectors"]:
        x = vec["inputs"][0]
        _check(lambda: O.AGC(vec["args"][0], vec["args"][1], vec["args"][2], 2.0, np.iscomplexobj(x)), vec, doc["epsilon"])


def test_powersquelch():
    """powersquelch.lua:45-80 restated with numpy (the power estimator of the AGC + a gate); tau is always 0.001 there"""
    doc = G.load("powersquelch_spec")
    for vec in doc["vectors"]:
        x, want = vec["inputs"][0], vec["outputs"][0]
        alpha, thr, p = 1 / (1 + 0.001 * 2.0), 10 ** (vec["args"][0] / 10), 0.0
        e = np.abs(x.astype(np.complex128)) ** 2
        got = np.empty_like(x)
        for i in range(len(x)):
            p = (1 - alpha) * p + alpha * e[i]
            got[i] = x[i] if p >= thr else 0
        assert G.max_abs_err(got, want) < doc["epsilon"], vec["desc"]


def test_frequencymodulator():
    doc = G.load("frequencymodulator_spec")
    assert doc["epsilon"] == 5e-5          # "radio.platform.features.liquid and 5e-3 or 5e-5": the pure-Lua bound
    for vec in doc["vectors"]:
        _check(lambda: O.FMModulator(vec["args"][0]), vec, 5e-5)


@pytest.mark.parametrize("name,pattern", [("pulsematchedfilter_spec", (1,)), ("manchestermatchedfilter_spec", (-1, 1))])
def test_matched_filters(name, pattern):
    """pulsematchedfilter.lua:39-47 / manchestermatchedfilter.lua:39-50: FIRFilterBlock with +-1 taps over the symbol period"""
    doc = G.load(name)
    for vec in doc["vectors"]:
        baudrate, invert = vec["args"]
        count = int(np.floor(2.0 / baudrate))           # jig rate 2.0
        sign = -1.0 if invert else 1.0
        taps = np.concatenate([np.full(count, sign * h, np.float32) for h in pattern])
        _check(lambda: O.FIR(taps, False), vec, doc["epsilon"])


def test_downsampler_bit_exact():
    doc = G.load("downsampler_spec")
    assert len(doc["vectors"]) == 20
    for vec in doc["vectors"]:
        x, want = vec["inputs"][0], vec["outputs"][0]
        cplx = np.iscomplexobj(x)
        whole, samplewise = G.run_whole_and_samplewise(lambda: O.Downsampler(vec["args"][0], cplx), x)
        assert np.array_equal(whole, want) and np.array_equal(samplewise, want), vec["desc"]


@pytest.mark.parametrize("mode", [O.MODE_LUA, O.MODE_F64])
def test_iirfilter(mode):
    doc = G.load("iirfilter_spec")
    for vec in doc["vectors"]:
        b, a = vec["args"]
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: O.IIR(b, a, cplx, mode), vec, doc["epsilon"])


@pytest.mark.parametrize("mode", [O.MODE_LUA, O.MODE_F64])
def test_singlepole_and_deemphasis(mode):
    doc = G.load("singlepolelowpassfilter_spec")
    for vec in doc["vectors"]:
        b, a = O.singlepole_lowpass_taps(vec["args"][0], RATE)
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: O.IIR(b, a, cplx, mode), vec, doc["epsilon"])
    doc = G.load("fmdeemphasisfilter_spec")
    for vec in doc["vectors"]:
        b, a = O.fm_deemphasis_taps(vec["args"][0], RATE)
        cplx = np.iscomplexobj(vec["inputs"][0])
        _check(lambda: O.IIR(b, a, cplx, mode), vec, doc["epsilon"])


def test_decimator():
    doc = G.load("decimator_spec")
    for vec in doc["vectors"]:
        cplx = np.iscomplexobj(vec["inputs"][0])
        # composite jig runs the graph once over the whole file (tests/jigs.lua:89-147); also check sample-wise
        _check(lambda: O.decimator(vec["args"][0], RATE, cplx), vec, doc["epsilon"])


@pytest.mark.parametrize("rot_mode", [O.MODE_LUA, O.MODE_F64])
def test_tuner(rot_mode):
    doc = G.load("tuner_spec")
    assert doc["epsilon"] == 1e-5
    for vec in doc["vectors"]:
        off, bw, dec = vec["args"]
        _check(lambda: O.tuner(off, bw, dec, RATE, rot_mode=rot_mode), vec, doc["epsilon"])


def test_window_utils():
    vals = G.load("window_utils_vectors")["values"]
    for name, want in vals.items():
        kind = name[len("window_"):]
        periodic = kind.endswith("_periodic")
        kind = kind[:-len("_periodic")] if periodic else kind
        got = O.window(128, kind, periodic).astype(np.float32)
        assert G.max_abs_err(got, want) < 1e-6, name


def test_filter_utils():
    vals = G.load("filter_utils_vectors")["values"]   # args from tests/utilities/filter_utils_spec.lua:8-27
    assert G.max_abs_err(O.firwin_lowpass(128, 0.5).astype(np.float32), vals["firwin_lowpass"]) < 1e-6
    assert G.max_abs_err(O.firwin_highpass(129, 0.5).astype(np.float32), vals["firwin_highpass"]) < 1e-6
    assert G.max_abs_err(O.firwin_bandpass(129, [0.4, 0.6]).astype(np.float32), vals["firwin_bandpass"]) < 1e-6
    assert G.max_abs_err(O.firwin_bandstop(129, [0.4, 0.6]).astype(np.float32), vals["firwin_bandstop"]) < 1e-6


def test_spectrum_utils():
    v = G.load("spectrum_utils_vectors")["values"]    # tests/utilities/spectrum_utils_spec.lua:58-91
    cx, rx = v["complex_test_vector"], v["real_test_vector"]
    assert G.max_abs_err(O.dft(cx), v["complex_test_vector_dft"]) < 1e-5
    assert G.max_abs_err(O.dft(rx), v["real_test_vector_dft"]) < 1e-5
    assert G.max_abs_err(O.idft(v["complex_test_vector_dft"], True), cx) < 1e-5
    assert G.max_abs_err(O.idft(v["real_test_vector_dft"], False), rx) < 1e-5
    for x, nm in ((cx, "complex"), (rx, "real")):
        for win in ("rectangular", "hamming"):
            assert G.max_abs_err(O.psd(x, win, 44100, False), v["%s_test_vector_%s_psd" % (nm, win)]) < 1e-5
            assert G.max_abs_err(O.psd(x, win, 44100, True), v["%s_test_vector_%s_psd_log" % (nm, win)]) < 3
    assert np.array_equal(O.fftshift(cx), v["complex_test_vector_fftshift"])
    assert np.array_equal(O.fftshift(rx), v["real_test_vector_fftshift"])


def test_top_chain():
    """tests/top_spec.lua:13-54: IQ x2 -> MultiplyConjugate -> Lowpass(16,100e3) -> Discriminator(5) ->
    Decimator(25,{num_taps=16}) at 1 MHz, against tests/top_vectors.gen.lua (epsilon 1e-6)."""
    v = G.load("top_vectors")["values"]
    s1 = np.frombuffer(v["SRC1_TEST_VECTOR"], np.complex64)
    s2 = np.frombuffer(v["SRC2_TEST_VECTOR"], np.complex64)
    want = np.frombuffer(v["SNK_TEST_VECTOR"], np.float32)
    x = O.multiply_conjugate(s1, s2)
    x = O.lowpass(16, 100e3, 1e6, True).process(x)
    x = O.FMDiscriminator(5.0).process(x)
    x = O.decimator(25, 1e6, False, num_taps=16).process(x)
    assert len(x) == len(want)
    assert G.max_abs_err(x, want) < 1e-6


def test_fir_modes_agree_long():
    rng = np.random.default_rng(7)
    x = (rng.uniform(-1, 1, 5000) + 1j * rng.uniform(-1, 1, 5000)).astype(np.complex64)
    taps = O.firwin_lowpass(128, 0.136).astype(np.float32)
    ref = O.FIR(taps, True, O.MODE_F64).process(x)
    for mode in (O.MODE_LUA, O.MODE_FMA):
        assert G.max_abs_err(O.FIR(taps, True, mode).process(x), ref) < 1e-6
    # chunking must not change a single bit (history carry)
    f = O.FIR(taps, True, O.MODE_FMA)
    whole = O.FIR(taps, True, O.MODE_FMA).process(x)
    parts = np.concatenate([f.process(x[a:b]) for a, b in ((0, 1), (1, 130), (130, 131), (131, 4000), (4000, 5000))])
    assert np.array_equal(whole, parts)


def test_simd_baseline_kernel_matches_golden_and_other_modes():
    """the timed CPU baseline (VOLK-style partial sums, optional OpenMP) is held to the same vectors"""
    doc = G.load("lowpassfilter_spec")
    for vec in doc["vectors"][:3] + doc["vectors"][6:9]:
        x, want = vec["inputs"][0], vec["outputs"][0]
        cplx = np.iscomplexobj(x)
        f = O.lowpass(vec["args"][0], vec["args"][1], RATE, cplx)
        assert G.max_abs_err(f.process_simd(x, 1), want) < doc["epsilon"]
    rng = np.random.default_rng(3)
    x = (rng.uniform(-1, 1, 40000) + 1j * rng.uniform(-1, 1, 40000)).astype(np.complex64)
    taps = O.firwin_lowpass(128, 0.136).astype(np.float32)
    ref = O.FIR(taps, True, O.MODE_F64).process(x)
    a, b = O.FIR(taps, True), O.FIR(taps, True)
    y1 = np.concatenate([a.process_simd(x[:777], 1), a.process_simd(x[777:], 1)])
    y4 = b.process_simd(x, 4)
    assert G.max_abs_err(y1, ref) < 1e-6 and np.array_equal(y1, y4)


def test_file_source_formats():
    """IQFileSource / RealFileSource conversion (14 formats each) against the reference's vectors"""
    for name, cplx in (("iqfile_spec", True), ("realfile_spec", False)):
        doc = G.load(name)
        assert len(doc["vectors"]) == 14
        for vec in doc["vectors"]:
            raw, fmt = vec["args"][0], vec["args"][1]
            got = O.format_convert(fmt, raw, cplx)
            want = vec["outputs"][0]
            assert len(got) == len(want) and G.max_abs_err(got, want) < doc["epsilon"], vec["desc"]


def test_timed_cpu_baseline_forms_match_golden_and_f64():
    """bench.py's cpu_baseline legs (oracle/lr_cpu_baseline.c: SIMD dot product per output, and the reference's default FFT
    overlap-save form on a Float32 Stockham FFT) are held to the 128-tap golden vectors and to the f64 restatement"""
    doc = G.load("lowpassfilter_spec")
    for vec in doc["vectors"][:3] + doc["vectors"][6:9]:        # the default-nyquist, default-window vectors (as the test above)
        x, want = vec["inputs"][0], vec["outputs"][0]
        taps = O.firwin_lowpass(vec["args"][0], vec["args"][1] / (RATE / 2)).astype(np.float32)
        for fn in (O.baseline_fir_dot, O.baseline_fir_overlap_save):
            assert G.max_abs_err(fn(taps, x), want) < doc["epsilon"], vec["desc"]
    rng = np.random.default_rng(11)
    for M, cplx in ((128, True), (128, False), (33, True), (200, False), (513, True)):
        n = 30000
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) if cplx else rng.uniform(-1, 1, n).astype(np.float32)
        taps = O.firwin_lowpass(M, 0.21).astype(np.float32)
        ref = O.FIR(taps, cplx, O.MODE_F64).process(x)
        assert G.max_abs_err(O.baseline_fir_dot(taps, x), ref) < 1e-6
        assert G.max_abs_err(O.baseline_fir_overlap_save(taps, x), ref) < 1e-6
        assert np.array_equal(O.baseline_fir_overlap_save(taps, x, 3), O.baseline_fir_overlap_save(taps, x))
    ct = (O.firwin_lowpass(64, 0.3) * (1 + 0.5j)).astype(np.complex64)
    x = (rng.uniform(-1, 1, 9000) + 1j * rng.uniform(-1, 1, 9000)).astype(np.complex64)
    assert G.max_abs_err(O.baseline_fir_overlap_save(ct, x), O.FIR(ct, True, O.MODE_F64).process(x)) < 1e-6
