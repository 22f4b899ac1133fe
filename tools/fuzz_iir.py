"""Fuzz (round 6, GPU box: python tools/fuzz_iir.py [seeds]): IIRFilterBlock with random stable filters - feedback orders 1 .. 8 (poles of radius 0.3 .. 0.97, real
and in conjugate pairs), 1 .. 8 feed-forward taps, ComplexFloat32 / Float32 streams, random lengths (1 .. 1.5 M samples) and ragged chunk cuts - against the
sequential double-precision recurrence (scipy.signal.lfilter on the same Float32 coefficients), relative to the output's scale.  The device evaluates the recurrence
as a scan over affine maps (kernels_iir.h), so what is checked is the scan's algebra and its tile / chunk seams, relative to the output's scale: 2e-5, or 3 x what scipy's Float32 lfilter loses on the same filter (ill-conditioned draws)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.signal import lfilter
import luaradio_amd as lr
from luaradio_amd import types


def main():
    lr.init(0)
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad, worst = 0, 0.0
    for seed in range(nseeds):
        rng = np.random.default_rng(52000 + seed)
        order = int(rng.integers(1, 9))
        poles = []
        while len(poles) < order:
            r = float(rng.uniform(0.3, 0.97))
            if order - len(poles) >= 2 and rng.integers(0, 2):
                th = float(rng.uniform(0.05, 3.0))
                poles += [r * np.exp(1j * th), r * np.exp(-1j * th)]
            else:
                poles.append(r * (1 if rng.integers(0, 2) else -1))
        a = np.real(np.poly(poles)).astype(np.float32)
        b = rng.uniform(-1, 1, int(rng.integers(1, 9))).astype(np.float32)
        cplx = bool(rng.integers(0, 2))
        n = int(rng.integers(1, 1500000)) if seed % 3 else int(rng.integers(1, 5000))
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) if cplx else rng.uniform(-1, 1, n).astype(np.float32)
        cuts = sorted(int(v) for v in rng.integers(1, max(n, 2), int(rng.integers(0, 5)))) if n > 1 else []
        blk = lr.IIRFilterBlock(b, a)
        blk.rate = 1e6
        blk.differentiate([types.ComplexFloat32 if cplx else types.Float32])
        blk.initialize()
        parts, s = [], 0
        for e in cuts + [n]:
            parts.append(blk.process(x[s:e]))
            s = e
        got = np.concatenate(parts)
        want = lfilter(b.astype(np.float64), a.astype(np.float64), x.astype(np.complex128 if cplx else np.float64))
        scale = max(1.0, float(np.max(np.abs(want))))
        err = float(np.max(np.abs(got - want))) / scale if len(got) == len(want) else 1e9
        # what Float32 arithmetic itself gives on this filter (scipy's Float32 lfilter: another structure, the same conditioning): filters with clustered poles near
        # the unit circle lose 1e-4 of their scale in ANY Float32 recurrence, the reference's own included
        f32 = lfilter(b, a, x.astype(np.complex64 if cplx else np.float32))
        yard = float(np.max(np.abs(f32 - want))) / scale
        worst = max(worst, err / max(yard, 1e-7))
        if not (err < max(2e-5, 3 * yard)):
            bad += 1
            print("seed %d order=%d nb=%d cplx=%s n=%d cuts=%s rel err %.3g (Float32 lfilter: %.3g; scale %.3g)" % (seed, order, len(b), cplx, n, cuts, err, yard, scale), flush=True)
    print("fuzz iir: %d cases, %d bad, worst error / Float32-lfilter error %.3g" % (nseeds, bad, worst))


if __name__ == "__main__":
    main()
