#!/usr/bin/env python3
"""FrequencyTranslator -> LowpassFilter(128) (a Tuner without decimation) on 2^26 samples: the fused rotator + FIR launch, direct form and overlap-save"""
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
y = torch.empty(2 * n + 64, device="cuda")
for rnd in range(2):
    for mode in (False, "fast"):
        b = lr.TunerBlock(-250e3, 200e3, 1, {"use_fft": mode})
        b.rate = 1102500.0
        b.differentiate([types.ComplexFloat32]); b.initialize()
        cap = b.max_output(n)
        for _ in range(3): b.process_device(x.data_ptr(), n, y.data_ptr(), cap)
        torch.cuda.synchronize()
        t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
        for _ in range(10): b.process_device(x.data_ptr(), n, y.data_ptr(), cap)
        L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 10; L.lrhip_timer_destroy(t)
        print("Tuner(-250k, 200k, 1) use_fft=%-5s %.4f ms  launches %d" % (mode, ms, b.chain.last_launches), flush=True)
