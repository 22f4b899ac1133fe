#!/usr/bin/env python3
"""FIRFilterBlock with 16 / 32 ComplexFloat32 taps on 2^26 ComplexFloat32 samples: ms per pass (run with and without LRHIP_NO_FIR_WIN_SHORT=1)"""
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
x = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
y = torch.empty(2 * n + 64, device="cuda")
rng = np.random.default_rng(5)
for m in (16, 32):
    taps = ((rng.uniform(0, 1, m) + 1j * rng.uniform(0, 1, m)) / m).astype(np.complex64)
    b = lr.FIRFilterBlock(taps)
    b.rate = 1.0
    b.differentiate([types.ComplexFloat32])
    b.initialize()
    for _ in range(100): b.process_device(x.data_ptr(), n, y.data_ptr(), n)
    torch.cuda.synchronize()
    t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
    for _ in range(30): b.process_device(x.data_ptr(), n, y.data_ptr(), n)
    L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 30; L.lrhip_timer_destroy(t)
    print(os.environ.get("TAG", ""), m, "complex taps: %.4f ms" % ms, flush=True)
