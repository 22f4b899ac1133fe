#!/usr/bin/env python3
"""the audio tail as a chain of its own: LowpassFilter(128, 15 kHz) -> FMDeemphasisFilter(75 us) -> Downsampler(5) on 2^26 Float32 samples (fir_win_real_kernel<128, true>)"""
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
fs = 220500.0
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(n, device="cuda", generator=g) * 2 - 1
y = torch.empty(n // 5 + 64, device="cuda")
def mk(cls, args, rate):
    b = cls(*args); b.rate = rate; b.differentiate([types.Float32]); b.initialize(); return b
ch = lr.Chain([mk(lr.LowpassFilterBlock, [128, 15e3], fs), mk(lr.FMDeemphasisFilterBlock, [75e-6], fs), mk(lr.DownsamplerBlock, [5], fs)])
cap = ch.max_output(n)
for rnd in range(3):
    for _ in range(3): ch.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
    for _ in range(10): ch.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 10; L.lrhip_timer_destroy(t)
    print("audio tail chain %.4f ms  launches %d" % (ms, ch.last_launches), flush=True)
