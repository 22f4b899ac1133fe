#!/usr/bin/env python3
"""short FIR filters (16 / 32 / 64 real taps) on 2^26 ComplexFloat32 and Float32 samples: direct form as dispatched (register-window kernel,
one-shot grid) - run again with LRHIP_NO_FIR_WIN_SHORT=1 for the Toeplitz-MFMA kernel - and the overlap-save kernel where it exists"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import luaradio_amd as lr
from luaradio_amd import types
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
n = 1 << 26
g = torch.Generator(device="cuda").manual_seed(1)
xc = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
y = torch.empty(2 * n + 64, device="cuda")
tag = "mfma" if os.environ.get("LRHIP_NO_FIR_WIN_SHORT") else "window"
rng = np.random.default_rng(3)
for cplx in (True, False):
    for m in (16, 32, 64):
        taps = (rng.uniform(0, 1, m) / m).astype(np.float32)
        for mode in (None, "fast"):
            if mode == "fast" and m < 32:
                continue
            b = lr.FIRFilterBlock(taps, mode) if mode else lr.FIRFilterBlock(taps)
            b.rate = 1.0; b.differentiate([types.ComplexFloat32 if cplx else types.Float32]); b.initialize()
            for _ in range(3): b.process_device(xc.data_ptr(), n, y.data_ptr(), n)
            torch.cuda.synchronize()
            t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
            for _ in range(10): b.process_device(xc.data_ptr(), n, y.data_ptr(), n)
            L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 10; L.lrhip_timer_destroy(t)
            bps = (16 if cplx else 8) * n / ms / 1e6
            print("%-6s %s %3d taps %-13s %.4f ms  %6.0f GB/s (%.0f %%)" % (tag, "cf32" if cplx else "f32 ", m, "overlap-save" if mode else "direct", ms, bps, bps / 80), flush=True)
