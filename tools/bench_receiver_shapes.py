#!/usr/bin/env python3
"""The WBFM-mono receiver at its stock shape (examples/rtlsdr_wbfm_mono.lua: 1 102 500 S/s, Tuner decimation 5, 128 audio taps - the ONE launch of
kernels_rx.h) and OFF that shape (another tuner decimation / input rate, another audio tap count: the two-launch form) on 2^26 RF samples
resident in HBM: launches per chunk, ms, GS/s, fraction of the 8 TB/s roofline (8 B per RF sample in + 4 B per audio sample out), and the RMS error against
the oracle's chain on the first 2^18 samples.  One JSON object per line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import luaradio_amd as lr
from luaradio_amd import blocks as B, composites as C, types

lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n


def receiver(rate, decim, audio_taps, tuner_taps=None):
    af = B.LowpassFilterBlock(audio_taps, 15e3)
    af.use_fft = 3
    opts = {}                     # use_fft left to the library (nil in Lua): what a LuaRadio script that just writes TunerBlock(...) gets
    if tuner_taps:
        opts["num_taps"] = tuner_taps
    return C._receiver([C.TunerBlock(-250e3, 200e3, decim, opts), B.FrequencyDiscriminatorBlock(1.25), af, B.FMDeemphasisFilterBlock(75e-6), B.DownsamplerBlock(5)], rate)


def fm(rate, count, seed=3):
    t = np.arange(count) / rate
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    return np.exp(1j * (2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / rate * np.cumsum(m))).astype(np.complex64)


g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
y = torch.empty(n, device="cuda")
shapes = [("stock: 1.1025 MS/s, Tuner /5, 128 audio taps", 1102500.0, 5, 128), ("2.205 MS/s, Tuner /10", 2205000.0, 10, 128), ("0.882 MS/s, Tuner /4", 882000.0, 4, 128),
          ("1.764 MS/s, Tuner /8", 1764000.0, 8, 128), ("1.1025 MS/s, Tuner /5, 96 audio taps", 1102500.0, 5, 96), ("1.1025 MS/s, Tuner /5, 160 audio taps", 1102500.0, 5, 160)]
base = None
for name, rate, decim, taps in shapes:
    r = receiver(rate, decim, taps)
    cap = r.max_output(n)
    for _ in range(3):
        r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    t = L.lrhip_timer_create()
    L.lrhip_timer_start(t)
    for _ in range(10):
        got = r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    L.lrhip_timer_stop(t)
    ms = L.lrhip_timer_elapsed_ms(t) / 10
    L.lrhip_timer_destroy(t)
    launches = r.chain.last_launches
    # parity of THIS shape against the oracle's chain (BASELINE: RMS <= 1e-5), on an FM signal
    from oracle import oracle as O
    m = 1 << 18
    iq = fm(rate, m)
    chk = receiver(rate, decim, taps)
    audio = chk.process(iq)
    r1 = rate / decim
    bb, aa = O.fm_deemphasis_taps(75e-6, r1)
    ora = O.Chain(O.tuner(-250e3, 200e3, decim, rate, mode=O.MODE_LUA, rot_mode=O.MODE_F64).stages +
                  [O.FMDiscriminator(1.25), O.lowpass(taps, 15e3, r1, False, mode=O.MODE_LUA), O.IIR(bb, aa, False, O.MODE_LUA), O.Downsampler(5, False)])
    want = ora.process(iq)
    rms = float(np.sqrt(np.mean((audio.astype(np.float64) - want) ** 2))) if len(audio) == len(want) else None
    alg = 8.0 * n + 4.0 * got
    row = {"receiver": name, "log2_samples": log2n, "launches": launches, "ms": round(ms, 4), "GS/s": round(n / ms / 1e6, 1), "hbm_frac": round(alg / (ms * 1e-3) / 8e12, 3),
           "rms_err_vs_oracle": rms, "audio_samples": int(got)}
    if base is None:
        base = ms
    row["x_stock"] = round(ms / base, 2)
    print(json.dumps(row), flush=True)
