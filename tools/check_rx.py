#!/usr/bin/env python3
"""Single-launch FM receiver (kernels_rx.h) against the two-launch form and the oracle chain: whole vector, ragged chunkings, tiny chunks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import luaradio_amd as lr
from luaradio_amd import _lib
from oracle import oracle as O

fs = 1102500.0


def fm(n, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * np.cumsum(m)
    return (np.exp(1j * ph) + 0.01 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)


def rx(flags=0):
    r = lr.wbfm_mono_receiver(fs, -250e3)
    if flags:
        r.exact = flags
        r._chain = lr.Chain(r._blocks, flags)
    return r


def chunked(r, x, cuts):
    parts, a = [], 0
    for b in list(cuts) + [len(x)]:
        parts.append(r.process(x[a:b]))
        a = b
    return np.concatenate(parts)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
x = fm(n)
one, two = rx(), rx(_lib.CHAIN_NO_SINGLE_LAUNCH)
a = one.process(x)
print("launches: single", one.chain.last_launches, "two", end=" ")
b = two.process(x)
print(two.chain.last_launches)
assert len(a) == len(b), (len(a), len(b))
d = np.abs(a.astype(np.float64) - b)
print("single vs two-launch: max %.3e rms %.3e  (first diff at %s)" % (d.max(), np.sqrt(np.mean(d ** 2)), np.argmax(d > 1e-6) if (d > 1e-6).any() else None))
want = O.wbfm_mono_chain(fs, -250e3, mode=O.MODE_LUA, rot_mode=O.MODE_F64).process(x[:1200000])
k = len(want) - 10
e = a[:k].astype(np.float64) - want[:k]
print("single vs oracle (first 1.2M): max %.3e rms %.3e" % (np.abs(e).max(), np.sqrt(np.mean(e ** 2))))
for cuts in ([1], [5], [24, 25, 26], [8192, 8193, 500000], [12800 * 7, 12800 * 7 + 3, 12800 * 7 + 9, 2000000], list(range(100000, n, 333337))):
    r = rx()
    c = chunked(r, x, cuts)
    assert len(c) == len(a), (cuts, len(c), len(a))
    dd = np.abs(c.astype(np.float64) - a)
    print("chunked %-40s max diff vs whole %.3e" % (str(cuts)[:40], dd.max()))
print("ok")
