"""Run by tests/test_gpu_firwin.py in a child process with LRHIP_FIR_WIN_CPLX=1 (the knob is read once per process): the
ComplexFloat32 register-window kernel (kernels_firwin2.h) behind Decimator / Tuner / Tuner + discriminator chains must give the
bits of the same blocks run one by one (fmaf chains in the reference's tap order, firfilter.lua:266-283; block-of-8 rotator phasors;
the discriminator's arithmetic), for ragged chunkings including one-sample chunks and odd sample offsets."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import luaradio_amd as lr
import tests  # noqa: F401  (the jig: use_fft = None means the direct form, as under tests.jigs in the reference)
from luaradio_amd import types
from oracle import oracle as O

assert os.environ.get("LRHIP_FIR_WIN_CPLX")
rate = 1102500.0
rng = np.random.default_rng(77)
n = 70000
x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def init(blocks):
    r, t = rate, types.ComplexFloat32
    for b in blocks:
        b.rate = r
        b.differentiate([t])
        b.initialize()
        r, t = b.get_rate(), b.get_output_type()
    return blocks


shapes = {
    "tuner+disc": lambda: [lr.FrequencyTranslatorBlock(-250e3), lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5), lr.FrequencyDiscriminatorBlock(1.25)],
    "decimator+disc": lambda: [lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5), lr.FrequencyDiscriminatorBlock(0.7)],
    "decimator": lambda: [lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5)],
    "tuner": lambda: [lr.FrequencyTranslatorBlock(123456.0), lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5)],
}
for name, build in shapes.items():
    ref = init(build())
    want = x
    for b in ref:
        want = b.process(want)
    for cuts in ([], [1], [1, 2, 3, 6361, 6362, 12720, 40001], [7, 20000, 20001, 50003], sorted(set(int(c) for c in rng.integers(0, n, 12)))):
        chain = lr.Chain(init(build()))
        parts, a = [], 0
        for b in list(cuts) + [n]:
            parts.append(chain.process(x[a:b]))
            a = b
        got = np.concatenate(parts)
        assert chain.last_launches == 1, (name, chain.last_launches)
        assert len(got) == len(want), (name, cuts)
        assert np.array_equal(got, want), (name, cuts, float(np.max(np.abs(got - want))))
# against the oracle (fmaf-chain filter behind the closed-form rotator)
ora = O.Chain([O.Rotator(2 * np.pi * 123456.0 / rate, O.MODE_F64), O.lowpass(128, 100e3, rate, True, mode=O.MODE_FMA)]).process(x)[::5]
got = lr.Chain(init(shapes["tuner"]())).process(x)
assert float(np.max(np.abs(got - ora))) < 2e-6
dec = lr.Chain(init(shapes["decimator"]())).process(x)
assert np.array_equal(dec, O.lowpass(128, 100e3, rate, True, mode=O.MODE_FMA).process(x)[::5])
print("winc ok")
