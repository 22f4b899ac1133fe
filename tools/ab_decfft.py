#!/usr/bin/env python3
"""same-box A/B of an environment knob on the fused tuner(+discriminator), polyphase FFT form, 2^26 samples
usage: ab_decfft.py KNOB v1 v2 ...   (LRHIP_DECFFT_ROUNDS, LRHIP_DECFFT_DBG ablation bits, ...)"""
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
y = torch.empty(2 * n, device="cuda")
knob, vals = sys.argv[1], sys.argv[2:]
def mk(disc):
    rate = 1102500.0
    t = lr.TunerBlock(-250e3, 200e3, 5, {"use_fft": "fast"})
    if not disc:
        t.rate = rate; t.differentiate([types.ComplexFloat32]); t.initialize(); return t
    top = lr.CompositeBlock(); top.connect(t, lr.FrequencyDiscriminatorBlock(1.25)); top.rate = rate
    top.differentiate([types.ComplexFloat32]); top.initialize(); return top
for disc in (True, False):
    blk = mk(disc)
    cap = blk.max_output(n)
    for rnd in range(2):
        for v in vals:
            os.environ[knob] = v
            for _ in range(3): blk.process_device(x.data_ptr(), n, y.data_ptr(), cap)
            torch.cuda.synchronize()
            t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
            for _ in range(20): blk.process_device(x.data_ptr(), n, y.data_ptr(), cap)
            L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 20; L.lrhip_timer_destroy(t)
            print("disc=%d %s=%s round %d: %.4f ms  %.1f GS/s  in %.1f GB/s" % (disc, knob, v, rnd, ms, n / ms / 1e6, 8 * n / ms / 1e6), flush=True)
    if knob == "LRHIP_DECFFT_DBG":
        break
