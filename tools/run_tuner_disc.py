#!/usr/bin/env python3
"""profiling target: the fused tuner + discriminator (polyphase FFT form) and the whole WBFM chain, 2^26 samples, a few launches"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
rx = lr.wbfm_mono_receiver(1102500.0, -250e3, use_fft=(mode if mode != "direct" else False))
cap = rx.max_output(n)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    rx.process_device(x.data_ptr(), n, y.data_ptr(), cap)
torch.cuda.synchronize()
