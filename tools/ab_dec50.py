#!/usr/bin/env python3
"""what the rotator costs in the LDS-staged decimator: Tuner(-100k, 10k, 50) against Decimator(50) (same filter, no rotator), 2^26 samples"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import luaradio_amd as lr
from luaradio_amd import types
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
n = 1 << 26
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
y = torch.empty(2 * (n // 50) + 64, device="cuda")
for rnd in range(2):
    for name, mk in (("Tuner(-100k, 10k, 50)", lambda: lr.TunerBlock(-100e3, 10e3, 50, {"use_fft": False})), ("Decimator(50)", lambda: lr.DecimatorBlock(50, {"use_fft": False}))):
        b = mk(); b.rate = 1102500.0; b.differentiate([types.ComplexFloat32]); b.initialize()
        cap = b.max_output(n)
        for _ in range(3): b.process_device(x.data_ptr(), n, y.data_ptr(), cap)
        torch.cuda.synchronize()
        t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
        for _ in range(10): b.process_device(x.data_ptr(), n, y.data_ptr(), cap)
        L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 10; L.lrhip_timer_destroy(t)
        print("%-24s %.4f ms  launches %d" % (name, ms, b.chain.last_launches), flush=True)
