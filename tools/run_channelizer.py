#!/usr/bin/env python3
"""BASELINE.json configs[4] alone (for the profiler): run_channelizer.py [steps] - 64-channel filterbank, 1024-tap prototype, 2^24 samples per step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import luaradio_amd as lr
from luaradio_amd import types
lr.init(0)
lr.adopt_torch_stream()
n, K = 1 << 24, 64
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(2 * n, dtype=torch.float32, device="cuda", generator=g) * 2 - 1
ch = lr.PolyphaseChannelizerBlock(K)
ch.rate = 1102500.0
ch.differentiate([types.ComplexFloat32])
ch.initialize()
cap = ch.max_output(n)
y = torch.empty(2 * cap + 64, dtype=torch.float32, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ch.process_device(x.data_ptr(), n, y.data_ptr(), cap)
torch.cuda.synchronize()
print("ok")
