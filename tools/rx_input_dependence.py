#!/usr/bin/env python3
"""the single-launch WBFM receiver takes 154.7 us on the FM test signal and 170.5 us on U(-1, 1) noise - same kernel, same sizes.
This runs the receiver on one input (INPUT = fm | noise | fm_small | noise_small | const | zeros) for ITERS passes, HIP-event timed, so that the
same command can be wrapped in rocprofv3 --pmc passes (tools/rx_input_dependence.sh) and sampled by rocm-smi."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import luaradio_amd as lr

lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
dev = torch.device("cuda")
fs, n = 1102500.0, 1 << 26
kind = os.environ.get("INPUT", "fm")
iters = int(os.environ.get("ITERS", "200"))
g = torch.Generator(device=dev).manual_seed(7)
if kind.startswith("fm"):
    t = torch.arange(n, dtype=torch.float64, device=dev) / fs
    m = 0.5 * torch.sin(2 * np.pi * 1e3 * t) + 0.5 * torch.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * torch.cumsum(m, 0)
    x = torch.stack([torch.cos(ph).float(), torch.sin(ph).float()], 1).reshape(-1)
    x += 0.01 * (torch.rand(2 * n, dtype=torch.float32, device=dev, generator=g) * 2 - 1)
    del t, m, ph
elif kind.startswith("noise"):
    x = torch.rand(2 * n, dtype=torch.float32, device=dev, generator=g) * 2 - 1
elif kind == "const":
    x = torch.full((2 * n,), 0.5, dtype=torch.float32, device=dev)
else:
    x = torch.zeros(2 * n, dtype=torch.float32, device=dev)
if kind.endswith("_small"):
    x *= 0.01
y = torch.empty(n, dtype=torch.float32, device=dev)
r = lr.wbfm_mono_receiver(fs, -250e3)
cap = r.max_output(n)
for _ in range(int(os.environ.get("RAMP", "300"))):
    r.process_device(x.data_ptr(), n, y.data_ptr(), cap)      # clock ramp
torch.cuda.synchronize()
tm = L.lrhip_timer_create()
L.lrhip_timer_start(tm)
for _ in range(iters):
    r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
L.lrhip_timer_stop(tm)
torch.cuda.synchronize()
print("INPUT=%s: %.4f ms per pass over %d passes (launches %d)" % (kind, L.lrhip_timer_elapsed_ms(tm) / iters, iters, r.chain.last_launches), flush=True)
