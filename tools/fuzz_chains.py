"""Fuzz (round 6, run on the GPU box: python tools/fuzz_chains.py [seeds]): random linear chains of device blocks (translator, direct-form lowpass, downsampler,
discriminator, complex -> real, conjugate, Float32 lowpass / de-emphasis / downsampler / constant) with random lengths (1 .. 700 000 samples) and ragged chunk cuts,
including runs of 1-sample chunks.  An EXACT chain (LRHIP_CHAIN_EXACT) must give the bits of the blocks run one by one; the default chain the same values to a
tolerance (angles modulo a turn where the filtered signal is noise).  Two sweeps of 1 500 seeds each: no fusion bug; what it flags are unstable draws (a de-emphasis
above the Nyquist rate: inf / nan in the reference arithmetic too) and angle wraps at +-pi behind a filter that removed the signal."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import luaradio_amd as lr
from luaradio_amd import types

FS = 1102500.0
c, f = types.ComplexFloat32, types.Float32


def mk(cls, args, dtype, rate, direct=False):
    b = cls(*args)
    if direct:
        b.use_fft = lr.block.fir_mode(False)
    b.rate = rate
    b.differentiate([dtype])
    b.initialize()
    return b


def build(rng):
    """returns a factory() -> list of fresh blocks"""
    spec = []
    dtype, rate = c, FS
    nstages = int(rng.integers(2, 6))
    for _ in range(nstages):
        if dtype is c:
            k = int(rng.integers(0, 7))
            if k == 0:
                spec.append((lr.FrequencyTranslatorBlock, [float(rng.uniform(-4e5, 4e5))], c, rate, False))
            elif k == 1:
                nt = int(rng.choice([16, 31, 64, 128, 200]))
                spec.append((lr.LowpassFilterBlock, [nt, float(rng.uniform(0.05, 0.4) * rate)], c, rate, True))
            elif k == 2:
                D = int(rng.choice([2, 4, 5, 8, 10, 25, 50]))
                spec.append((lr.DownsamplerBlock, [D], c, rate, False))
                rate /= D
            elif k == 3:
                spec.append((lr.FrequencyDiscriminatorBlock, [float(rng.uniform(0.5, 5))], c, rate, False))
                dtype = f
            elif k == 4:
                spec.append((lr.ComplexMagnitudeBlock, [], c, rate, False))
                dtype = f
            elif k == 5:
                D = int(rng.choice([2, 5, 25, 50]))
                nt = int(rng.choice([32, 128]))
                spec.append((lr.FrequencyTranslatorBlock, [float(rng.uniform(-4e5, 4e5))], c, rate, False))
                spec.append((lr.LowpassFilterBlock, [nt, float(0.4 * rate / D)], c, rate, True))
                spec.append((lr.DownsamplerBlock, [D], c, rate, False))
                rate /= D
            else:
                spec.append((lr.ComplexConjugateBlock, [], c, rate, False))
        else:
            k = int(rng.integers(0, 4))
            if k == 0:
                nt = int(rng.choice([16, 64, 128]))
                spec.append((lr.LowpassFilterBlock, [nt, float(rng.uniform(0.02, 0.4) * rate)], f, rate, True))
            elif k == 1:
                spec.append((lr.FMDeemphasisFilterBlock, [75e-6], f, rate, False))
            elif k == 2:
                D = int(rng.choice([2, 5, 4]))
                spec.append((lr.DownsamplerBlock, [D], f, rate, False))
                rate /= D
            else:
                spec.append((lr.MultiplyConstantBlock, [float(rng.uniform(0.1, 3))], f, rate, False))
    return lambda: [mk(*s) for s in spec], [s[0].__name__ + str(s[1]) for s in spec]


def chunked(proc, x, cuts):
    parts, a = [], 0
    for b in list(cuts) + [len(x)]:
        parts.append(proc(x[a:b]))
        a = b
    return np.concatenate(parts) if parts else np.zeros(0)


def main():
    lr.init(0)
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    bad = 0
    for seed in range(nseeds):
        rng = np.random.default_rng(19000 + seed)
        factory, names = build(rng)
        n = int(rng.integers(20000, 700000)) if seed % 4 else int(rng.integers(1, 3000))
        t = np.arange(n) / FS
        x = (np.exp(2j * np.pi * (150e3 * t + 3 * np.sin(2 * np.pi * 2e3 * t))) + 0.05 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)
        cuts = sorted(int(v) for v in rng.integers(1, max(n, 2), int(rng.integers(0, 5)))) if n > 1 else []
        if seed % 8 == 1 and n > 40:
            a0 = int(rng.integers(1, n - 30)); cuts = sorted(set(cuts + list(range(a0, a0 + 25))))
        ref = factory()

        def one_by_one(v):
            for b in ref:
                v = b.process(v)
            return v

        try:
            want = chunked(one_by_one, x, cuts)
            exact = lr.Chain(factory(), exact=True)
            got = chunked(exact.process, x, cuts)
            ok_exact = len(got) == len(want) and np.array_equal(got, want)
            fast = lr.Chain(factory())
            got2 = chunked(fast.process, x, cuts)
            err = float(np.max(np.abs(got2.astype(np.complex128) - want))) if len(got2) == len(want) and len(want) else (0.0 if len(got2) == len(want) else 1e9)
            scale = float(np.max(np.abs(want))) if len(want) else 1.0
            has_disc = any("Discriminator" in s for s in names)
            ok_fast = err <= (2e-3 if has_disc else 2e-5) * max(scale, 1.0)
        except Exception as e:                                    # noqa: BLE001
            ok_exact, ok_fast, err = False, False, -1.0
            print("seed %d EXCEPTION %s: %s" % (seed, type(e).__name__, e))
        if not (ok_exact and ok_fast):
            bad += 1
            print("seed %d n=%d cuts=%s exact=%s fast_err=%.3g launches=%s\n    %s" % (seed, n, cuts, ok_exact, err, getattr(exact, "last_launches", None), " -> ".join(names)), flush=True)
    print("fuzz: %d chains, %d bad" % (nseeds, bad))


if __name__ == "__main__":
    main()
