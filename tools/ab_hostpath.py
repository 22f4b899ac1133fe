#!/usr/bin/env python3
"""A/B of the stand-alone host path (lrhip_stage_execute, LowpassFilter cf32 -> cf32): vector size x registered / staged.  Knobs via the environment:
LRHIP_HOST_PIECE_MIN (samples per pipelined piece), LRHIP_HOST_NO_PIECES, LRHIP_COPY_THREADS."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import luaradio_amd as lr                     # noqa: E402
from luaradio_amd import _lib, types          # noqa: E402


def aligned(count, dtype):
    raw = np.empty(count * np.dtype(dtype).itemsize + 4096, np.uint8)
    off = (-raw.ctypes.data) % 4096
    return raw[off:off + count * np.dtype(dtype).itemsize].view(dtype)


def main():
    lr.init(0)
    L = _lib.load()
    n = 1 << 25
    rng = np.random.default_rng(1)
    x = aligned(n, np.complex64)
    x[:] = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    y = aligned(n, np.complex64)
    for reg in (0, 1):
        if reg:
            _lib.check(L.lrhip_host_register(x.ctypes.data_as(C.c_void_p), x.nbytes), "reg")
            _lib.check(L.lrhip_host_register(y.ctypes.data_as(C.c_void_p), y.nbytes), "reg")
        for log2v in (17, 20, 22, 24):
            vec = 1 << log2v
            blk = lr.LowpassFilterBlock(128, 15e3)
            blk.use_fft = 2
            blk.rate = 220500.0
            blk.differentiate([types.ComplexFloat32])
            blk.initialize()
            q = blk.stage_handle()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for a in range(0, n, vec):
                    L.lrhip_stage_execute(q, C.c_void_p(x.ctypes.data + 8 * a), vec, C.c_void_p(y.ctypes.data + 8 * a), vec)
                best = min(best, time.perf_counter() - t0)
            print("registered=%d vector=2^%d: %.2f GS/s (%.1f GB/s each way)" % (reg, log2v, n / best / 1e9, 8 * n / best / 1e9), flush=True)


if __name__ == "__main__":
    main()
