#!/usr/bin/env python3
"""Per-block throughput table on one MI355X (the reference's benchmark suite is per block:
benchmarks/luaradio_benchmark.lua).  Device-resident vectors, HIP-event timing on the launch stream.
Prints one JSON object per block: MS/s (input samples), algorithmic GB/s, fraction of 8 TB/s, and the same
for the cheapest streaming kernel (MultiplyConstant) as the achievable-bandwidth yardstick."""
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
    ap.add_argument("--only", default="", help="comma-separated substrings: run only the rows whose name contains one of them")
    args = ap.parse_args()
    only = [t for t in args.only.split(",") if t]
    import numpy as np
    import torch
    import luaradio_amd as lr
    from luaradio_amd import spectrum_utils, types

    lr.init(0)
    L = lr._lib.load()
    lr.adopt_torch_stream()
    n = 1 << args.log2_samples
    g = torch.Generator(device="cuda").manual_seed(1)
    xc = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
    xr = xc[:n]
    out = torch.empty(2 * n + 64, device="cuda")

    def timeit(fn, reps=args.reps):
        fn()
        torch.cuda.synchronize()
        t = L.lrhip_timer_create()
        L.lrhip_timer_start(t)
        for _ in range(reps):
            fn()
        L.lrhip_timer_stop(t)
        ms = L.lrhip_timer_elapsed_ms(t) / reps
        L.lrhip_timer_destroy(t)
        return ms

    def mk(cls, args_, cplx, rate=1102500.0):
        b = cls(*args_)
        b.rate = rate
        b.differentiate([types.ComplexFloat32 if cplx else types.Float32])
        b.initialize()
        return b

    rows = []

    def run(name, blk, cplx, alg_bytes_per_sample, flops=0.0):
        if only and not any(t in name for t in only):
            return
        blk = blk()
        x = xc if cplx else xr
        cap = blk.max_output(n)
        dst = out if cap * (2 if cplx else 1) <= out.numel() else torch.empty(cap * 2 + 64, device="cuda")
        ms = timeit(lambda: blk.process_device(x.data_ptr(), n, dst.data_ptr(), cap))
        gbs = alg_bytes_per_sample * n / ms / 1e6
        rows.append({"block": name, "MS/s": round(n / ms / 1e3, 1), "alg_GB/s": round(gbs, 1), "frac_8TB/s": round(gbs / 8000, 4),
                     "ms": round(ms, 4), "TFLOP/s": round(flops * n / ms / 1e9, 2) if flops else None})

    taps128 = lr.filter_utils.firwin_lowpass(128, 15e3 / 110250)
    # yardstick: the cheapest streaming kernel (one multiply per scalar), 8 B in + 8 B out per sample
    run("MultiplyConstant(1.0) cf32 (streaming yardstick)", lambda: mk(lr.MultiplyConstantBlock, [1.0], True), True, 16)
    run("FIRFilter 128 real taps, cf32, direct form (use_fft=False)", lambda: mk(lr.FIRFilterBlock, [taps128, False], True), True, 16, 512)
    run("FIRFilter 128 real taps, cf32, overlap-save (use_fft=fast)", lambda: mk(lr.FIRFilterBlock, [taps128, "fast"], True), True, 16, 125)
    run("FIRFilter 128 real taps, f32, direct form (use_fft=False)", lambda: mk(lr.FIRFilterBlock, [taps128, False], False), False, 8, 256)
    run("FIRFilter 128 real taps, f32, overlap-save (use_fft=fast)", lambda: mk(lr.FIRFilterBlock, [taps128, "fast"], False), False, 8, 63)
    taps1276 = np.convolve(np.convolve(np.convolve(np.convolve(lr.filter_utils.firwin_lowpass(256, 0.3), lr.filter_utils.firwin_lowpass(256, 0.35)),
                                                   lr.filter_utils.firwin_lowpass(256, 0.4)), lr.filter_utils.firwin_lowpass(256, 0.45)),
                           lr.filter_utils.firwin_lowpass(256, 0.5)).astype(np.float32)
    run("FIRFilter 1276 real taps, cf32, overlap-save (4096-point blocks)", lambda: mk(lr.FIRFilterBlock, [taps1276, "fast"], True), True, 16, 180)
    run("FIRFilter 768 real taps, cf32, overlap-save (4096-point blocks)", lambda: mk(lr.FIRFilterBlock, [taps1276[:768], "fast"], True), True, 16, 160)
    # round 6: Float32 streams above 512 taps on the 64 x 64 kernel (round 4-5: the partitioned kernel of kernels_firpols.h, LRHIP_F64_F32=0)
    run("FIRFilter 1276 real taps, f32, overlap-save (64 x 64 kernel, two stream blocks per transform)", lambda: mk(lr.FIRFilterBlock, [taps1276, "fast"], False), False, 8, 250)
    taps4096 = np.resize(taps1276, 4096).astype(np.float32) / 4.0
    # (the 8 192-tap rows first: their two launches are the kernel of the 4 096-tap rows, whose dispatches must be the LAST ones of that kernel in the trace -
    # profiles/summarize_rocpd.py --last 10)
    run("FIRFilter 8192 real taps, cf32, overlap-save (four partitions of the 64 x 64 kernel, TWO launches)", lambda: mk(lr.FIRFilterBlock, [np.resize(taps1276, 8192).astype(np.float32) / 8.0, "fast"], True), True, 16, 1300)
    run("FIRFilter 8192 real taps, f32, overlap-save (four partitions of the 64 x 64 kernel, TWO launches)", lambda: mk(lr.FIRFilterBlock, [np.resize(taps1276, 8192).astype(np.float32) / 8.0, "fast"], False), False, 8, 650)
    run("FIRFilter 768 real taps, f32, overlap-save (64 x 64 kernel, two stream blocks per transform)", lambda: mk(lr.FIRFilterBlock, [taps1276[:768], "fast"], False), False, 8, 160)
    run("FIRFilter 2048 real taps, f32, overlap-save (64 x 64 kernel at an overlap of 2 048)", lambda: mk(lr.FIRFilterBlock, [taps4096[:2048], "fast"], False), False, 8, 240)
    run("FIRFilter 4096 real taps, f32, overlap-save (two partitions of the 64 x 64 kernel, ONE launch)", lambda: mk(lr.FIRFilterBlock, [taps4096, "fast"], False), False, 8, 420)
    run("FIRFilter 4096 real taps, cf32, overlap-save (two partitions of the 64 x 64 kernel, ONE launch)", lambda: mk(lr.FIRFilterBlock, [taps4096, "fast"], True), True, 16, 420)
    run("FIRFilter 2048 real taps, cf32, overlap-save (64 x 64 kernel at an overlap of 2 048)", lambda: mk(lr.FIRFilterBlock, [taps4096[:2048], "fast"], True), True, 16, 240)
    run("FIRFilter 1276 complex taps, cf32, overlap-save (4096-point blocks, four waves per CU)",
        lambda: mk(lr.FIRFilterBlock, [np.asarray(taps1276, np.complex64) * (1 + 0.5j), "fast"], True), True, 16, 180)
    run("FIRFilter 16 real taps, cf32", lambda: mk(lr.FIRFilterBlock, [taps128[:16], False], True), True, 16, 64)
    run("FIRFilter 128 complex taps, cf32, direct form (use_fft=False)", lambda: mk(lr.FIRFilterBlock, [np.asarray(taps128, np.complex64) * (1 + 0.5j), False], True), True, 16, 1024)
    run("FIRFilter 128 complex taps, cf32, overlap-save (use_fft=fast)", lambda: mk(lr.FIRFilterBlock, [np.asarray(taps128, np.complex64) * (1 + 0.5j), "fast"], True), True, 16, 125)
    run("FrequencyTranslator", lambda: mk(lr.FrequencyTranslatorBlock, [-250e3], True), True, 16)
    run("FrequencyDiscriminator", lambda: mk(lr.FrequencyDiscriminatorBlock, [1.25], True), True, 12)
    run("Downsampler(5) cf32", lambda: mk(lr.DownsamplerBlock, [5], True), True, 8 + 8 / 5)
    run("Downsampler(5) f32", lambda: mk(lr.DownsamplerBlock, [5], False), False, 4 + 4 / 5)
    run("FMDeemphasis f32", lambda: mk(lr.FMDeemphasisFilterBlock, [75e-6], False, 220500.0), False, 8)
    run("Decimator(5) cf32 (fused FIR+downsample)", lambda: mk(lr.DecimatorBlock, [5, {"use_fft": False}], True), True, 8 + 8 / 5, 4 * 128 / 5)
    run("Tuner(-250k,200k,5) (fused rot+FIR+downsample)", lambda: mk(lr.TunerBlock, [-250e3, 200e3, 5, {"use_fft": False}], True), True, 8 + 8 / 5, 4 * 128 / 5 + 6)
    run("Tuner(-100k,10k,50) (the AM / SSB / NBFM receivers' tuner: LDS-staged decimator)", lambda: mk(lr.TunerBlock, [-100e3, 10e3, 50, {"use_fft": False}], True), True, 8 + 8 / 50, 4 * 128 / 50 + 6)
    run("Tuner(-100k,12k,80) (rtlsdr_pocsag.lua / rtlsdr_ax25.lua: LDS-staged decimator, phase-array layout)", lambda: mk(lr.TunerBlock, [-100e3, 12e3, 80, {"use_fft": False}], True), True, 8 + 8 / 80, 4 * 128 / 80 + 6)
    run("Decimator(25) cf32 (LDS-staged decimator, no rotator)", lambda: mk(lr.DecimatorBlock, [25, {"use_fft": False}], True), True, 8 + 8 / 25, 4 * 128 / 25)
    run("Decimator(5) cf32, polyphase FFT overlap-save", lambda: mk(lr.DecimatorBlock, [5, {"use_fft": "fast"}], True), True, 8 + 8 / 5, 62)
    run("Tuner(-250k,200k,5), polyphase FFT overlap-save", lambda: mk(lr.TunerBlock, [-250e3, 200e3, 5, {"use_fft": "fast"}], True), True, 8 + 8 / 5, 62)
    # write-heavy yardstick: the zero-stuffing Upsampler moves exactly the Interpolator's bytes (8 B in, 8 L B out per input sample) with no arithmetic
    run("Upsampler(5) cf32 (the Interpolator's traffic without arithmetic: write-heavy yardstick)", lambda: mk(lr.UpsamplerBlock, [5], True), True, 8 + 8 * 5)
    run("Interpolator(5) cf32 (polyphase, input samples)", lambda: mk(lr.InterpolatorBlock, [5], True), True, 8 + 8 * 5, 4 * 128)
    run("RationalResampler(3, 2) cf32 (polyphase, input samples)", lambda: mk(lr.RationalResamplerBlock, [3, 2], True), True, 8 + 8 * 1.5, 4 * 128 * 1.5 / 3)
    run("RationalResampler(2, 3) cf32 (polyphase, input samples)", lambda: mk(lr.RationalResamplerBlock, [2, 3], True), True, 8 + 8 * 2 / 3, 4 * 128 * (2 / 3) / 2)
    run("RationalResampler(4, 3) cf32 (polyphase, input samples)", lambda: mk(lr.RationalResamplerBlock, [4, 3], True), True, 8 + 8 * 4 / 3, 4 * 128 * (4 / 3) / 4)
    run("RationalResampler(5, 4) cf32 (polyphase, input samples)", lambda: mk(lr.RationalResamplerBlock, [5, 4], True), True, 8 + 8 * 5 / 4, 4 * 128 * (5 / 4) / 5)
    run("RationalResampler(3, 4) cf32 (polyphase, input samples)", lambda: mk(lr.RationalResamplerBlock, [3, 4], True), True, 8 + 8 * 3 / 4, 4 * 128 * (3 / 4) / 3)
    run("RationalResampler(4, 5) cf32 (polyphase, input samples)", lambda: mk(lr.RationalResamplerBlock, [4, 5], True), True, 8 + 8 * 4 / 5, 4 * 128 * (4 / 5) / 4)
    # IQFileSource's sample formats converted on the device (iqfile.lua:82-116, format_utils.lua:82-97): raw records in, ComplexFloat32 out
    class _Fmt:
        def __init__(self, fmt):
            self.q = L.lrhip_format_convert_create(fmt.encode(), 1)
            assert self.q, lr._lib.last_error()

        def max_output(self, n_):
            return n_

        def process_device(self, xp, n_, yp, cap):
            return lr._lib.check(L.lrhip_stage_execute_device(self.q, xp, n_, yp, cap), "format")

    run("IQ records u8 -> cf32 (device)", lambda: _Fmt("u8"), True, 2 + 8)
    run("IQ records s16le -> cf32 (device)", lambda: _Fmt("s16le"), True, 4 + 8)
    run("IQ records f32be -> cf32 (device)", lambda: _Fmt("f32be"), True, 8 + 8)
    run("HilbertTransform(65) f32 -> cf32", lambda: mk(lr.HilbertTransformBlock, [65], False), False, 12, 2 * 65)
    run("HilbertTransform(129) f32 -> cf32", lambda: mk(lr.HilbertTransformBlock, [129], False), False, 12, 2 * 129)
    # the reference suite's IIR entry (benchmarks/luaradio_benchmark.lua: 5 feed-forward, 3 feedback taps), a stable filter
    b_iir, a_iir = [0.0976, 0.1953, 0.0976, 0.05, 0.02], [1.0, -0.9428, 0.3333]
    run("IIRFilter 5 ff / 3 fb cf32", lambda: mk(lr.IIRFilterBlock, [b_iir, a_iir], True), True, 16)
    run("IIRFilter 5 ff / 3 fb f32", lambda: mk(lr.IIRFilterBlock, [b_iir, a_iir], False), False, 8)
    run("IIRFilter 3 ff / 3 fb (biquad) cf32", lambda: mk(lr.IIRFilterBlock, [b_iir[:3], a_iir], True), True, 16)
    if only and not any(t in "WBFM PSD Channelizer" for t in only):
        for r in rows:
            print(json.dumps(r))
        return
    rx0 = lr.wbfm_mono_receiver(1102500.0, -250e3, use_fft=False)
    cap = rx0.max_output(n)
    ms = timeit(lambda: rx0.process_device(xc.data_ptr(), n, out.data_ptr(), cap))
    rows.append({"block": "WBFM mono chain, direct-form tuner (RF samples in)", "MS/s": round(n / ms / 1e3, 1), "alg_GB/s": round(8.16 * n / ms / 1e6, 1),
                 "frac_8TB/s": round(8.16 * n / ms / 1e6 / 8000, 4), "ms": round(ms, 4), "TFLOP/s": None, "launches": rx0.chain.last_launches})
    rx = lr.wbfm_mono_receiver(1102500.0, -250e3)
    cap = rx.max_output(n)
    ms = timeit(lambda: rx.process_device(xc.data_ptr(), n, out.data_ptr(), cap))
    rows.append({"block": "WBFM mono chain (RF samples in)", "MS/s": round(n / ms / 1e3, 1), "alg_GB/s": round(8.16 * n / ms / 1e6, 1),
                 "frac_8TB/s": round(8.16 * n / ms / 1e6 / 8000, 4), "ms": round(ms, 4), "TFLOP/s": None, "launches": rx.chain.last_launches})
    # a fan-out branch fed from an IQ file: [format stage (u8 records), Tuner(-350e3, 100e3, 5)] - one launch of the persistent Toeplitz kernel on the records
    def tuner_chain(head, decim=5):
        blocks = head + [lr.FrequencyTranslatorBlock(-350e3), lr.LowpassFilterBlock(128, 500e3 / decim), lr.DownsamplerBlock(decim)]
        r, t = 1102500.0, types.ComplexFloat32
        for b in blocks[len(head):]:
            b.rate = r
            b.differentiate([t])
            b.initialize()
            r, t = b.get_rate(), b.get_output_type()
        return lr.Chain(blocks)

    if True:
        src8 = lr.IQFileSource(bytes(16), "u8", 1102500.0)
        src8.initialize()
        tch = tuner_chain([src8])
        raw8t = (torch.rand(2 * n + 64, device="cuda") * 256).to(torch.uint8)
        capt = tch.max_output(n)
        ms = timeit(lambda: tch.process_device(raw8t.data_ptr(), n, out.data_ptr(), capt))
        src50 = lr.IQFileSource(bytes(16), "u8", 1102500.0)
        src50.initialize()
        tch50 = tuner_chain([src50], 50)
        cap50 = tch50.max_output(n)
        ms50 = timeit(lambda: tch50.process_device(raw8t.data_ptr(), n, out.data_ptr(), cap50))
        rows.append({"block": "Tuner(decimation 50) from u8 IQ records (AM / SSB / NBFM receivers fed from a file)", "MS/s": round(n / ms50 / 1e3, 1),
                     "alg_GB/s": round(2.16 * n / ms50 / 1e6, 1), "frac_8TB/s": round(2.16 * n / ms50 / 1e6 / 8000, 4), "ms": round(ms50, 4), "TFLOP/s": None,
                     "launches": tch50.last_launches})
        rows.append({"block": "Tuner from u8 IQ records (fan-out branch fed from a file)", "MS/s": round(n / ms / 1e3, 1), "alg_GB/s": round(3.6 * n / ms / 1e6, 1),
                     "frac_8TB/s": round(3.6 * n / ms / 1e6 / 8000, 4), "ms": round(ms, 4), "TFLOP/s": None, "launches": tch.last_launches})
    # the same receiver fed the raw unsigned 8-bit records of an RTL-SDR style IQ file (IQFileSource's format stage at the head of the chain): the single
    # launch reads the records itself; LRHIP_RX_NO_U8_FOLD=1 = conversion launch + receiver
    import importlib.util
    spec = importlib.util.spec_from_file_location("iqfile_wbfm_mono", os.path.join(ROOT, "examples", "iqfile_wbfm_mono.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    _src, ch8, _rate = ex.build_chain(bytes(16), "u8", 1102500.0, -250e3)
    raw8 = (torch.rand(2 * n + 64, device="cuda") * 256).to(torch.uint8)
    cap = ch8.max_output(n)
    ms = timeit(lambda: ch8.process_device(raw8.data_ptr(), n, out.data_ptr(), cap))
    rows.append({"block": "WBFM mono chain from u8 IQ records (RF samples in)", "MS/s": round(n / ms / 1e3, 1), "alg_GB/s": round(2.16 * n / ms / 1e6, 1),
                 "frac_8TB/s": round(2.16 * n / ms / 1e6 / 8000, 4), "ms": round(ms, 4), "TFLOP/s": None, "launches": ch8.last_launches})
    # PSD: frames of 1024
    N = 1024
    frames = n // N
    win = np.asarray(lr.window_utils.window(N, "hamming", True), np.float32)
    import ctypes as C
    st = L.lrhip_psd_create(N, win.ctypes.data_as(C.POINTER(C.c_float)), 1102500.0 * float(np.sum(win.astype(np.float64) ** 2)), 1, 1, 1)
    ms = timeit(lambda: L.lrhip_stage_execute_device(st, xc.data_ptr(), frames * N, out.data_ptr(), frames * N))
    rows.append({"block": "PSD N=1024 hamming log fftshift", "MS/s": round(n / ms / 1e3, 1), "alg_GB/s": round(12 * n / ms / 1e6, 1),
                 "frac_8TB/s": round(12 * n / ms / 1e6 / 8000, 4), "ms": round(ms, 4), "TFLOP/s": None})
    L.lrhip_stage_destroy(st)
    # configs[4]: 64-channel filterbank, 1024-tap prototype, as a dense GEMM on the f32 matrix cores
    ch = mk(lr.PolyphaseChannelizerBlock, [64], True)
    nch = min(n, 1 << 24)
    cap = ch.max_output(nch)
    big = torch.empty(2 * cap + 64, device="cuda")
    ms = timeit(lambda: ch.process_device(xc.data_ptr(), nch, big.data_ptr(), cap), reps=3)
    tf = 8.0 * 1024 * nch / ms / 1e9
    rows.append({"block": "PolyphaseChannelizer K=64, 1024 taps (dense MFMA GEMM)", "MS/s": round(nch / ms / 1e3, 1),
                 "alg_GB/s": round(16 * nch / ms / 1e6, 1), "frac_8TB/s": round(16 * nch / ms / 1e6 / 8000, 4), "ms": round(ms, 4),
                 "TFLOP/s": round(tf, 2), "mfma_util_vs_157.3TF": round(tf / 157.3, 4)})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
