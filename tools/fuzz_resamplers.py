"""Fuzz (round 6, GPU box: python tools/fuzz_resamplers.py [seeds]): Interpolator(1..8), RationalResampler(1..8, 1..8), Decimator(2..60) with 16 .. 200 taps on
ComplexFloat32 / Float32 streams, random lengths and chunk cuts - the fused composite against its member blocks run one by one: bit for bit with direct-form filters,
to 2e-6 x L with the automatic (overlap-save) filter.  800 seeds: 0 bad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import luaradio_amd as lr
from luaradio_amd import types, _lib


def fresh(kind, args, dtype, direct):
    cls = {"interp": lr.InterpolatorBlock, "rational": lr.RationalResamplerBlock, "decim": lr.DecimatorBlock}[kind]
    b = cls(*args)
    for m in b._blocks:
        if hasattr(m, "use_fft") and direct:
            m.use_fft = lr.block.fir_mode(False)
    b.rate = 1e6
    b.differentiate([dtype])
    return b


def chunked(proc, x, cuts):
    parts, a = [], 0
    for b in list(cuts) + [len(x)]:
        parts.append(proc(x[a:b]))
        a = b
    return np.concatenate(parts)


def main():
    lr.init(0)
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = 0
    worst = 0.0
    for seed in range(nseeds):
        rng = np.random.default_rng(31000 + seed)
        kind = ["interp", "rational", "decim"][int(rng.integers(0, 3))]
        nt = int(rng.choice([16, 33, 64, 128, 200]))
        opts = {"num_taps": nt}
        if kind == "interp":
            args = [int(rng.integers(1, 9)), opts]
        elif kind == "rational":
            args = [int(rng.integers(1, 9)), int(rng.integers(1, 9)), opts]
        else:
            args = [int(rng.choice([2, 3, 4, 5, 7, 8, 9, 10, 16, 25, 50, 60])), opts]
        cplx = bool(rng.integers(0, 2))
        dtype = types.ComplexFloat32 if cplx else types.Float32
        direct = bool(rng.integers(0, 2))
        n = int(rng.integers(1, 300000)) if seed % 3 else int(rng.integers(1, 2000))
        x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) if cplx else rng.uniform(-1, 1, n).astype(np.float32)
        cuts = sorted(int(v) for v in rng.integers(1, max(n, 2), int(rng.integers(0, 5)))) if n > 1 else []
        try:
            fused = fresh(kind, args, dtype, direct)
            fused.initialize()
            got = chunked(fused.process, x, cuts)
            ref = fresh(kind, args, dtype, direct)
            ref._propagate_rates()
            for m in ref._blocks:
                m.initialize()

            def one_by_one(v):
                for m in ref._blocks:
                    v = m.process(v)
                return v
            want = chunked(one_by_one, x, cuts)
            same_len = len(got) == len(want)
            err = float(np.max(np.abs(got.astype(np.complex128) - want))) if same_len and len(want) else 0.0
            bits = same_len and np.array_equal(got, want)
            ok = same_len and (bits if direct else err < 2e-6 * max(1.0, args[0] if kind != "decim" else 1.0))
            worst = max(worst, err)
        except Exception as e:                                   # noqa: BLE001
            ok, bits, err, same_len = False, False, -1, False
            print("seed %d EXC %s: %s" % (seed, type(e).__name__, e))
        if not ok:
            bad += 1
            print("seed %d %s%s cplx=%s direct=%s n=%d cuts=%s same_len=%s bits=%s err=%.3g" % (seed, kind, args, cplx, direct, n, cuts, same_len, bits, err), flush=True)
    print("fuzz resamplers: %d cases, %d bad, worst err %.3g" % (nseeds, bad, worst))


if __name__ == "__main__":
    main()
