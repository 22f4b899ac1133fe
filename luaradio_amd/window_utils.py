"""Window generation - mirrors radio/utilities/window_utils.lua (host side, double precision)."""
import math

_window_functions = {
    # radio/utilities/window_utils.lua:11-27
    "rectangular": lambda n, M: 1.0,
    "hamming": lambda n, M: 0.54 - 0.46 * math.cos((2 * math.pi * n) / (M - 1)),
    "hanning": lambda n, M: 0.5 - 0.5 * math.cos((2 * math.pi * n) / (M - 1)),
    "bartlett": lambda n, M: (2 / (M - 1)) * ((M - 1) / 2 - abs(n - (M - 1) / 2)),
    "blackman": lambda n, M: 0.42 - 0.5 * math.cos((2 * math.pi * n) / (M - 1)) + 0.08 * math.cos((4 * math.pi * n) / (M - 1)),
}


def window(M, window_type, periodic=False):
    """radio/utilities/window_utils.lua:39-50"""
    if window_type not in _window_functions:
        raise ValueError('Unsupported window "%s".' % str(window_type))
    f = _window_functions[window_type]
    Mw = (M + 1) if periodic else M
    return [f(n, Mw) for n in range(M)]
