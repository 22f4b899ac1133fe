#!/usr/bin/env python3
"""The hot-path entries of the reference's own benchmark suite (benchmarks/luaradio_benchmark.lua; names and
parameters as there), run device-resident on one MI355X, printed next to the reference's published i5-4570T numbers
(website/_includes/benchmarks/benchmarks.i5.luaradio.json).

The reference measures ZeroSource -> block -> BenchmarkSink across three processes (output samples per second);
here the block runs on vectors already in HBM, HIP-event timed - the published column is context, not a like-for-like
comparison (different hardware, and the reference's figure includes its socket transport).  Random data instead of
zeros; random taps as in the reference, except the IIR entry (a stable filter: random feedback taps overflow on
non-zero input)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-samples", type=int, default=26)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--trials", type=int, default=0, help="> 0: the suite's own protocol instead of HIP-event timing - ZeroSource -> block -> "
                    "BenchmarkSink(JSON), that many trials of --trial-seconds each, mean and sigma of OUTPUT samples per second "
                    "(luaradio_benchmark.lua:690-738); the zero vector is resident in HBM, 2^log2-samples per process()")
    ap.add_argument("--trial-seconds", type=float, default=0.5)
    args = ap.parse_args()
    import numpy as np
    import torch
    import luaradio_amd as lr
    from luaradio_amd import types

    lr.init(0)
    L = lr._lib.load()
    lr.adopt_torch_stream()
    n = 1 << args.log2_samples
    g = torch.Generator(device="cuda").manual_seed(1)
    xc = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
    xr = torch.rand(n, device="cuda", generator=g) * 2 - 1
    out = torch.empty(2 * n + 64, device="cuda")
    rng = np.random.default_rng(5)

    def timeit(fn):
        fn()
        t = L.lrhip_timer_create()
        L.lrhip_timer_start(t)
        for _ in range(args.reps):
            fn()
        L.lrhip_timer_stop(t)
        torch.cuda.synchronize()
        ms = L.lrhip_timer_elapsed_ms(t) / args.reps
        L.lrhip_timer_destroy(t)
        return ms

    def mk(cls, a, cplx, rate=1.0):
        b = cls(*a)
        b.rate = rate
        b.differentiate([types.ComplexFloat32 if cplx else types.Float32])
        b.initialize()
        return b

    rows = []

    def run(name, published, blk, cplx, third=False):
        if args.trials > 0:
            from luaradio_amd import meters
            src = lr.ZeroSource(types.ComplexFloat32 if cplx else types.Float32, 1.0, n)
            src.initialize()
            ptr, cnt = src.process_device()
            cap = blk.max_output(cnt)

            def make_top(results):
                snk = lr.BenchmarkSink(results, True)
                snk.differentiate([types.ComplexFloat32 if cplx else types.Float32])
                snk.initialize()
                return (lambda: snk.process(int(blk.process_device(ptr, cnt, out.data_ptr(), cap)))), snk

            r = meters.run_trials(make_top, args.trials, args.trial_seconds, sync=torch.cuda.synchronize)
            src.cleanup()
            rows.append({"benchmark": name, "MS/s (output)": round(r["samples_per_second"] / 1e6, 1), "sigma": round(r["samples_per_second_stdev"] / 1e6, 1),
                         "trials": args.trials, "reference_i5_MS/s": published,
                         "ratio": round(r["samples_per_second"] / 1e6 / published, 1) if published else None})
            return
        x = xc if cplx else xr
        nin = n // 3 if third else n           # (an upsampler's output has to fit the output vector)
        cap = blk.max_output(nin)
        ms = timeit(lambda: blk.process_device(x.data_ptr(), nin, out.data_ptr(), cap))
        got = nin / ms / 1e3                   # MS/s of input; the reference counts output samples
        rows.append({"benchmark": name, "MS/s": round(got, 1), "reference_i5_MS/s": published,
                     "ratio": round(got / published, 1) if published else None, "ms": round(ms, 4)})

    t16 = rng.uniform(0, 1, 16).astype(np.float32)
    t128 = rng.uniform(0, 1, 128).astype(np.float32) / 64
    t256 = rng.uniform(0, 1, 256).astype(np.float32) / 128
    c16 = (rng.uniform(0, 1, 16) + 1j * rng.uniform(0, 1, 16)).astype(np.complex64)
    c128 = ((rng.uniform(0, 1, 128) + 1j * rng.uniform(0, 1, 128)) / 64).astype(np.complex64)
    five = lr.CompositeBlock()
    five.connect(*[lr.FIRFilterBlock(t256, "fast") for _ in range(5)])
    five.rate = 1.0
    five.differentiate([types.ComplexFloat32])
    five.initialize()
    run("Five Back to Back FIR Filters (FFT, 256 Real taps, Complex input)", 42.62, five, True)
    run("FIR Filter (Dotprod, 16 Real taps, Complex input)", 67.45, mk(lr.FIRFilterBlock, [t16], True), True)
    run("FIR Filter (Dotprod, 16 Real taps, Real input)", 84.72, mk(lr.FIRFilterBlock, [t16], False), False)
    run("FIR Filter (Dotprod, 16 Complex taps, Complex input)", 58.92, mk(lr.FIRFilterBlock, [c16], True), True)
    run("FIR Filter (FFT, 128 Real taps, Complex input)", 133.85, mk(lr.FIRFilterBlock, [t128, "fast"], True), True)
    run("FIR Filter (FFT, 128 Real taps, Real input)", 141.49, mk(lr.FIRFilterBlock, [t128, "fast"], False), False)
    run("FIR Filter (FFT, 128 Complex taps, Complex input)", 132.66, mk(lr.FIRFilterBlock, [c128, "fast"], True), True)
    b_taps, a_taps = [0.2, 0.3, 0.3, 0.2], [1.0, -1.2, 0.5]
    run("IIR Filter (5 ff 3 fb Real taps, Complex input)", 52.22, mk(lr.IIRFilterBlock, [b_taps, a_taps], True), True)
    run("IIR Filter (5 ff 3 fb Real taps, Real input)", 98.94, mk(lr.IIRFilterBlock, [b_taps, a_taps], False), False)
    run("FM Deemphasis Filter", 139.93, mk(lr.FMDeemphasisFilterBlock, [75e-6], False, 220500.0), False)
    run("Downsampler (M = 5), Complex input", 144.11, mk(lr.DownsamplerBlock, [5], True), True)
    run("Downsampler (M = 5), Real input", 253.07, mk(lr.DownsamplerBlock, [5], False), False)
    run("Frequency Translator", 396.69, mk(lr.FrequencyTranslatorBlock, [0.2], True), True)
    run("Upsampler (L = 3), Complex input", 702.62, mk(lr.UpsamplerBlock, [3], True), True, third=True)
    run("Upsampler (L = 3), Real input", 1259.63, mk(lr.UpsamplerBlock, [3], False), False, third=True)
    run("Hilbert Transform (65 taps)", 67.66, mk(lr.HilbertTransformBlock, [65], False), False)
    run("Hilbert Transform (129 taps)", 47.47, mk(lr.HilbertTransformBlock, [129], False), False)
    run("Frequency Discriminator", 111.61, mk(lr.FrequencyDiscriminatorBlock, [1.25], True), True)

    # the element-wise entries (luaradio_benchmark.lua:422-625): the reference feeds both inputs of a two-input block from one source
    def mk2(cls, cplx):
        b = cls()
        b.rate = 1.0
        t = types.ComplexFloat32 if cplx else types.Float32
        b.differentiate([t, t])
        b.initialize()
        return b

    def run2(name, published, blk, cplx):
        x = xc if cplx else xr
        ms = timeit(lambda: lr._lib.check(L.lrhip_stage_execute2_device(blk.stage_handle(), x.data_ptr(), x.data_ptr(), n, out.data_ptr(), n), name))
        rows.append({"benchmark": name, "MS/s": round(n / ms / 1e3, 1), "reference_i5_MS/s": published, "ratio": round(n / ms / 1e3 / published, 1), "ms": round(ms, 4)})

    if args.trials == 0:
        run2("Add (Complex)", 226.38, mk2(lr.AddBlock, True), True)
        run2("Subtract (Complex)", 224.0, mk2(lr.SubtractBlock, True), True)
        run2("Multiply (Complex)", 280.61, mk2(lr.MultiplyBlock, True), True)
        run2("Multiply (Real)", 608.55, mk2(lr.MultiplyBlock, False), False)
        run2("Multiply Conjugate", 291.61, mk2(lr.MultiplyConjugateBlock, True), True)
        run2("Float to Complex", 397.66, mk2(lr.FloatToComplexBlock, False), False)
    run("Multiply Constant (Real constant, Complex input)", 308.63, mk(lr.MultiplyConstantBlock, [5.0], True), True)
    run("Multiply Constant (Complex constant, Complex input)", 254.46, mk(lr.MultiplyConstantBlock, [complex(3.0, 2.0)], True), True)
    run("Multiply Constant (Real constant, Real input)", 570.66, mk(lr.MultiplyConstantBlock, [5.0], False), False)
    run("Absolute Value", 647.47, mk(lr.AbsoluteValueBlock, [], False), False)
    run("Complex Conjugate", 383.44, mk(lr.ComplexConjugateBlock, [], True), True)
    run("Complex Magnitude", 297.39, mk(lr.ComplexMagnitudeBlock, [], True), True)
    run("Complex Phase", 130.04, mk(lr.ComplexPhaseBlock, [], True), True)
    run("Delay (N = 3000, Complex input)", 473.35, mk(lr.DelayBlock, [3000], True), True)
    run("Complex to Real", 554.76, mk(lr.ComplexToRealBlock, [], True), True)
    run("Complex to Imaginary", 555.63, mk(lr.ComplexToImagBlock, [], True), True)
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
