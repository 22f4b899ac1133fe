#!/usr/bin/env python3
"""timing of the register-window kernels (kernels_firwin.h) on one MI355X:
  tail   : LowpassFilter(128) -> FMDeemphasis -> Downsampler(5) on the 220.5 kHz Float32 stream of a 2^26-sample WBFM step
  fir32  : FIRFilterBlock 128 real taps on 2^26 Float32 samples, direct form
  wbfm   : the whole receiver on 2^26 RF samples
Run it twice for an A/B: plain, and with LRHIP_NO_FIR_IIR_FUSION=1 LRHIP_NO_FIR_WIN=1 (the knobs are read once per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import luaradio_amd as lr
from luaradio_amd import types
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(7)
tag = "nofuse" if os.environ.get("LRHIP_NO_FIR_IIR_FUSION") else "fused"


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tm = L.lrhip_timer_create(); L.lrhip_timer_start(tm)
    for _ in range(reps): fn()
    L.lrhip_timer_stop(tm); ms = L.lrhip_timer_elapsed_ms(tm) / reps; L.lrhip_timer_destroy(tm)
    return ms


def mk(cls, args, t, rate):
    b = cls(*args); b.rate = rate; b.differentiate([t]); b.initialize(); return b


n = (1 << 26) // 5
x = torch.rand(n, dtype=torch.float32, device=dev, generator=g) * 2 - 1
y = torch.empty(n + 64, dtype=torch.float32, device=dev)
ch = lr.Chain([mk(lr.LowpassFilterBlock, [128, 15e3], types.Float32, 220500.0), mk(lr.FMDeemphasisFilterBlock, [75e-6], types.Float32, 220500.0),
               mk(lr.DownsamplerBlock, [5], types.Float32, 220500.0)])
ch.blocks[0].use_fft = 3
fir_auto = lr.LowpassFilterBlock(128, 15e3); fir_auto.use_fft = 3; fir_auto.rate = 220500.0; fir_auto.differentiate([types.Float32]); fir_auto.initialize()
ch = lr.Chain([fir_auto, ch.blocks[1], ch.blocks[2]])
cap = ch.max_output(n)
for rnd in range(3):
    ms = timeit(lambda: ch.process_device(x.data_ptr(), n, y.data_ptr(), cap))
    print("%s tail  %.4f ms  launches %d  (%.0f MS/s in)" % (tag, ms, ch.last_launches, n / ms / 1e3), flush=True)
n2 = 1 << 26
x2 = torch.rand(n2, dtype=torch.float32, device=dev, generator=g) * 2 - 1
y2 = torch.empty(n2 + 64, dtype=torch.float32, device=dev)
fir = mk(lr.LowpassFilterBlock, [128, 15e3], types.Float32, 220500.0)
for rnd in range(3):
    ms = timeit(lambda: fir.process_device(x2.data_ptr(), n2, y2.data_ptr(), n2), 10)
    print("%s fir32 %.4f ms  (%.0f MS/s, %.0f GB/s)" % (tag, ms, n2 / ms / 1e3, 8.0 * n2 / ms / 1e6), flush=True)
del x2, y2
xr = torch.rand(2 * n2, dtype=torch.float32, device=dev, generator=g) * 2 - 1
rx = lr.wbfm_mono_receiver(1102500.0, -250e3)
cap = rx.max_output(n2)
for rnd in range(3):
    ms = timeit(lambda: rx.process_device(xr.data_ptr(), n2, y.data_ptr(), cap))
    print("%s wbfm  %.4f ms  launches %d" % (tag, ms, rx.chain.last_launches), flush=True)
