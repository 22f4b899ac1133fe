#!/usr/bin/env python3
"""a few steps of the WBFM receiver's audio tail (Lowpass(128) -> FMDeemphasis -> Downsampler(5) on 2^26/5 Float32 samples) and of the whole
receiver on 2^26 samples, for rocprofv3 kernel traces"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import luaradio_amd as lr
from luaradio_amd import types
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
n = (1 << 26) // 5
x = torch.rand(n, dtype=torch.float32, device="cuda") * 2 - 1
y = torch.empty(n + 64, dtype=torch.float32, device="cuda")
def mk(cls, args):
    b = cls(*args); b.rate = 220500.0; b.differentiate([types.Float32]); b.initialize(); return b
fir = lr.LowpassFilterBlock(128, 15e3); fir.use_fft = 3; fir.rate = 220500.0; fir.differentiate([types.Float32]); fir.initialize()
ch = lr.Chain([fir, mk(lr.FMDeemphasisFilterBlock, [75e-6]), mk(lr.DownsamplerBlock, [5])])
cap = ch.max_output(n)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(reps):
    ch.process_device(x.data_ptr(), n, y.data_ptr(), cap)
torch.cuda.synchronize()
if len(sys.argv) > 2:
    n2 = 1 << 26
    xr = torch.rand(2 * n2, dtype=torch.float32, device="cuda") * 2 - 1
    rx = lr.wbfm_mono_receiver(1102500.0, -250e3)
    cap = rx.max_output(n2)
    for _ in range(reps):
        rx.process_device(xr.data_ptr(), n2, y.data_ptr(), cap)
    torch.cuda.synchronize()
