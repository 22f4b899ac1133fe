#!/usr/bin/env python3
"""A/B of host_execute's DIRECT mode on the single-pass element-wise stages (round 6): stand-alone blocks on registered host vectors,
lrhip_stage_execute per vector.  Run twice: LRHIP_HOST_DIRECT=1 (default) / LRHIP_HOST_DIRECT=0 (the staged piece pipeline)."""
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
    _lib.check(L.lrhip_host_register(x.ctypes.data_as(C.c_void_p), x.nbytes), "reg")
    _lib.check(L.lrhip_host_register(y.ctypes.data_as(C.c_void_p), y.nbytes), "reg")
    c = types.ComplexFloat32
    blocks = [("MultiplyConstant(0.5)", lr.MultiplyConstantBlock, [0.5], 1), ("FrequencyTranslator(-250e3)", lr.FrequencyTranslatorBlock, [-250e3], 1),
              ("Downsampler(5)", lr.DownsamplerBlock, [5], 5), ("ComplexMagnitude", lr.ComplexMagnitudeBlock, [], 1), ("ComplexConjugate", lr.ComplexConjugateBlock, [], 1)]
    mode = "direct" if os.environ.get("LRHIP_HOST_DIRECT", "1") != "0" else "staged"
    for name, cls, args, dec in blocks:
        for log2v in (20, 22, 24):
            vec = 1 << log2v
            blk = cls(*args)
            blk.rate = 1102500.0
            blk.differentiate([c])
            blk.initialize()
            q = blk.stage_handle()
            osz = L.lrhip_stage_output_size(q)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for a in range(0, n, vec):
                    got = L.lrhip_stage_execute(q, C.c_void_p(x.ctypes.data + 8 * a), vec, C.c_void_p(y.ctypes.data + osz * (a // dec)), vec // dec + 1)
                    assert got >= 0, _lib.last_error()
                best = min(best, time.perf_counter() - t0)
            print("%s %s vector=2^%d: %.2f GS/s in (%.1f GB/s in, %.1f GB/s out)" % (mode, name, log2v, n / best / 1e9, 8 * n / best / 1e9, osz * (n // dec) / best / 1e9), flush=True)


if __name__ == "__main__":
    main()
