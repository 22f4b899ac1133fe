#!/usr/bin/env python3
"""same-box A/B of the WBFM-mono chain with the direct-form (Toeplitz MFMA) tuner against the polyphase-FFT tuner, on bench.py's
FM test signal and on uniform random IQ, 2^26 samples, alternating"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import luaradio_amd as lr
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
dev = torch.device("cuda")
fs, n = 1102500.0, 1 << 26
t = torch.arange(n, dtype=torch.float64, device=dev) / fs
m = 0.5 * torch.sin(2 * np.pi * 1e3 * t) + 0.5 * torch.sin(2 * np.pi * 5e3 * t)
ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * torch.cumsum(m, 0)
g = torch.Generator(device=dev).manual_seed(7)
xfm = torch.stack([torch.cos(ph).float(), torch.sin(ph).float()], 1).reshape(-1)
xfm += 0.01 * (torch.rand(2 * n, dtype=torch.float32, device=dev, generator=g) * 2 - 1)
del t, m, ph
xr = torch.rand(2 * n, dtype=torch.float32, device=dev, generator=g) * 2 - 1
y = torch.empty(n, dtype=torch.float32, device=dev)
rx = {"direct": lr.wbfm_mono_receiver(fs, -250e3, use_fft=False), "fft": lr.wbfm_mono_receiver(fs, -250e3, use_fft="fast")}
for rnd in range(3):
    for data, x in (("fm", xfm), ("random", xr)):
        for name, r in rx.items():
            cap = r.max_output(n)
            for _ in range(3): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
            torch.cuda.synchronize()
            tm = L.lrhip_timer_create(); L.lrhip_timer_start(tm)
            for _ in range(20): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
            L.lrhip_timer_stop(tm); ms = L.lrhip_timer_elapsed_ms(tm) / 20; L.lrhip_timer_destroy(tm)
            print("round %d %-6s %-6s %.4f ms  launches %d" % (rnd, data, name, ms, r.chain.last_launches), flush=True)
