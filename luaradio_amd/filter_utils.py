"""FIR tap design by the window method - mirrors radio/utilities/filter_utils.lua.

Host side, plain double arithmetic in the reference's operation order; the caller casts to Float32 once
(radio/blocks/signal/lowpassfilter.lua:46-47).
"""
import math

from . import window_utils


def _fir_lowpass(num_taps, cutoff):
    # filter_utils.lua:21-33
    h = []
    for n in range(num_taps):
        c = n - (num_taps - 1) / 2
        h.append(cutoff if c == 0 else math.sin(math.pi * cutoff * c) / (math.pi * c))
    return h


def _fir_highpass(num_taps, cutoff):
    # filter_utils.lua:43-57
    assert (num_taps % 2) == 1, "Number of taps must be odd."
    h = []
    for n in range(num_taps):
        c = n - (num_taps - 1) / 2
        h.append(1 - cutoff if c == 0 else -math.sin(math.pi * cutoff * c) / (math.pi * c))
    return h


def _fir_bandpass(num_taps, cutoffs):
    # filter_utils.lua:67-82
    assert (num_taps % 2) == 1, "Number of taps must be odd."
    assert len(cutoffs) == 2, "Cutoffs should be a length two array."
    h = []
    for n in range(num_taps):
        c = n - (num_taps - 1) / 2
        if c == 0:
            h.append(cutoffs[1] - cutoffs[0])
        else:
            h.append(math.sin(math.pi * cutoffs[1] * c) / (math.pi * c) - math.sin(math.pi * cutoffs[0] * c) / (math.pi * c))
    return h


def _fir_bandstop(num_taps, cutoffs):
    # filter_utils.lua:92-107
    assert (num_taps % 2) == 1, "Number of taps must be odd."
    assert len(cutoffs) == 2, "Cutoffs should be a length two array."
    h = []
    for n in range(num_taps):
        c = n - (num_taps - 1) / 2
        if c == 0:
            h.append(1 - (cutoffs[1] - cutoffs[0]))
        else:
            h.append(math.sin(math.pi * cutoffs[0] * c) / (math.pi * c) - math.sin(math.pi * cutoffs[1] * c) / (math.pi * c))
    return h


def _firwin(h, window_type, scale_freq):
    # filter_utils.lua:121-141
    window_type = window_type or "hamming"
    w = window_utils.window(len(h), window_type)
    h = [h[n] * w[n] for n in range(len(h))]
    scale = 0.0
    for n in range(len(h)):
        scale = scale + h[n] * math.cos(math.pi * (n - (len(h) - 1) / 2) * scale_freq)
    return [v / scale for v in h]


def firwin_lowpass(num_taps, cutoff, window_type=None):
    """filter_utils.lua:152-157"""
    return _firwin(_fir_lowpass(num_taps, cutoff), window_type, 0.0)


def firwin_highpass(num_taps, cutoff, window_type=None):
    """filter_utils.lua:168-173"""
    return _firwin(_fir_highpass(num_taps, cutoff), window_type, 1.0)


def firwin_bandpass(num_taps, cutoffs, window_type=None):
    """filter_utils.lua:184-189"""
    return _firwin(_fir_bandpass(num_taps, cutoffs), window_type, (cutoffs[0] + cutoffs[1]) / 2)


def firwin_bandstop(num_taps, cutoffs, window_type=None):
    """filter_utils.lua:200-205"""
    return _firwin(_fir_bandstop(num_taps, cutoffs), window_type, 0.0)


def _complex_firwin(h, center_freq, window_type, scale_freq):
    # filter_utils.lua:218-247
    window_type = window_type or "hamming"
    h = [[h[n] * math.cos(math.pi * center_freq * n), h[n] * math.sin(math.pi * center_freq * n)] for n in range(len(h))]
    w = window_utils.window(len(h), window_type)
    for n in range(len(h)):
        h[n][0] = h[n][0] * w[n]
        h[n][1] = h[n][1] * w[n]
    scale = [0.0, 0.0]
    for n in range(len(h)):
        e = [math.cos(math.pi * (n - (len(h) - 1) / 2) * scale_freq), math.sin(-1 * math.pi * (n - (len(h) - 1) / 2) * scale_freq)]
        scale[0] = scale[0] + (h[n][0] * e[0] - h[n][1] * e[1])
        scale[1] = scale[1] + (h[n][1] * e[0] + h[n][0] * e[1])
    denom = scale[0] * scale[0] + scale[1] * scale[1]
    return [[(v[0] * scale[0] + v[1] * scale[1]) / denom, (v[1] * scale[0] - v[0] * scale[1]) / denom] for v in h]


def firwin_complex_bandpass(num_taps, cutoffs, window_type=None):
    """filter_utils.lua:258-263"""
    h = _fir_lowpass(num_taps, (max(cutoffs) - min(cutoffs)) / 2)
    return _complex_firwin(h, (cutoffs[0] + cutoffs[1]) / 2, window_type, (cutoffs[0] + cutoffs[1]) / 2)


def firwin_complex_bandstop(num_taps, cutoffs, window_type=None):
    """filter_utils.lua:274-281"""
    h = _fir_highpass(num_taps, (max(cutoffs) - min(cutoffs)) / 2)
    scale_freq = 1.0 if (cutoffs[0] < 0.0 and 0.0 < cutoffs[1]) else 0.0
    return _complex_firwin(h, (cutoffs[0] + cutoffs[1]) / 2, window_type, scale_freq)


def fir_root_raised_cosine(num_taps, sample_rate, beta, symbol_period):
    """filter_utils.lua:294-329"""
    if (num_taps % 2) == 0:
        raise ValueError("Number of taps must be odd.")
    h = []
    for n in range(num_taps):
        t = (n - (num_taps - 1) / 2) / sample_rate
        if t == 0:
            h.append((1 / (math.sqrt(symbol_period))) * (1 - beta + 4 * beta / math.pi))
        elif abs(t - (-symbol_period / (4 * beta))) < 1e-5 or abs(t - symbol_period / (4 * beta)) < 1e-5:
            h.append((beta / math.sqrt(2 * symbol_period)) * ((1 + 2 / math.pi) * math.sin(math.pi / (4 * beta)) + (1 - 2 / math.pi) * math.cos(math.pi / (4 * beta))))
        else:
            num = math.cos((1 + beta) * math.pi * t / symbol_period) + math.sin((1 - beta) * math.pi * t / symbol_period) / (4 * beta * t / symbol_period)
            denom = (1 - (4 * beta * t / symbol_period) * (4 * beta * t / symbol_period))
            h.append(((4 * beta) / (math.pi * math.sqrt(symbol_period))) * num / denom)
    scale = 0.0
    for v in h:
        scale = scale + v
    return [v / scale for v in h]


def fir_hilbert_transform(num_taps, window_type=None):
    """filter_utils.lua:340-366"""
    window_type = window_type or "hamming"
    if (num_taps % 2) == 0:
        raise ValueError("Number of taps must be odd.")
    h = []
    for n in range(num_taps):
        n_shifted = n - (num_taps - 1) / 2
        h.append(0 if (n_shifted % 2) == 0 else 2 / (n_shifted * math.pi))
    w = window_utils.window(num_taps, window_type)
    return [h[n] * w[n] for n in range(num_taps)]
