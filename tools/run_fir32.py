#!/usr/bin/env python3
"""a few launches of the Float32-stream 128-tap FIR (register-window kernel) on 2^26 samples, for rocprofv3 (tools/prof_counters.sh)"""
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
x = torch.rand(n, dtype=torch.float32, device="cuda") * 2 - 1
y = torch.empty(n + 64, dtype=torch.float32, device="cuda")
fir = lr.LowpassFilterBlock(128, 15e3); fir.rate = 220500.0; fir.differentiate([types.Float32]); fir.initialize()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    fir.process_device(x.data_ptr(), n, y.data_ptr(), n)
torch.cuda.synchronize()
