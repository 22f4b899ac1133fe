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
