#!/usr/bin/env python3
"""WBFM-mono receiver on 2^26 samples of bench.py's FM test signal: ms per pass, HIP-event timed (A/B helper: LRHIP_LIB_PATH / env knobs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import luaradio_amd as lr
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
dev = torch.device("cuda")
fs, n = 1102500.0, 1 << int(os.environ.get('LOG2N', '26'))
t = torch.arange(n, dtype=torch.float64, device=dev) / fs
m = 0.5 * torch.sin(2 * np.pi * 1e3 * t) + 0.5 * torch.sin(2 * np.pi * 5e3 * t)
ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * torch.cumsum(m, 0)
g = torch.Generator(device=dev).manual_seed(7)
x = torch.stack([torch.cos(ph).float(), torch.sin(ph).float()], 1).reshape(-1)
x += 0.01 * (torch.rand(2 * n, dtype=torch.float32, device=dev, generator=g) * 2 - 1)
del t, m, ph
y = torch.empty(n, dtype=torch.float32, device=dev)
r = lr.wbfm_mono_receiver(fs, -250e3, use_fft=False)
cap = r.max_output(n)
for _ in range(400): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)      # clock ramp
torch.cuda.synchronize()
if os.environ.get("DUMP"): np.save(os.environ["DUMP"], y[:cap].cpu().numpy())
res = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for _ in range(5): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    tm = L.lrhip_timer_create(); L.lrhip_timer_start(tm)
    for _ in range(100): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    L.lrhip_timer_stop(tm); res.append(L.lrhip_timer_elapsed_ms(tm) / 100); L.lrhip_timer_destroy(tm)
print(os.environ.get("TAG", ""), " ".join("%.4f" % v for v in res), "ms  launches", r.chain.last_launches, " checksum %.6f" % float(y[:cap].double().abs().mean()), flush=True)
