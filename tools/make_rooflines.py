#!/usr/bin/env python3
"""profiles/rooflines.json from the kernel-trace summaries of tools/profile_round.sh: one row per line of DESIGN.md section 7 with
{kernel, algorithmic bytes per launch, average microseconds over the timed dispatches, fraction of the 8 TB/s roof} so that the
judge's recomputation is one division.  usage: make_rooflines.py gpurun_out/prof_<tag> profiles/rooflines.json [tag]"""
import json
import re
import sys

HBM = 8000.0e9
FP32 = 157.3e12


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(lrhip::\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+%\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows[m.group(1)] = {"calls": int(m.group(2)), "avg_us": float(m.group(3)), "median_us": float(m.group(4)), "min_us": float(m.group(5)),
                                "avg_timed_us": float(m.group(7)), "min_timed_us": float(m.group(8))}
    return rows


def parse_pmc(path):
    """summary_pmc_blocks.txt (tools/pmc_blocks.sh): {kernel: {"FETCH_SIZE": KB, "WRITE_SIZE": KB}} per dispatch"""
    out = {}
    try:
        for line in open(path):
            m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(lrhip::\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
            if m:
                out.setdefault(m.group(2), {})[m.group(1)] = float(m.group(4))
    except OSError:
        pass
    return out


def find(rows, *subs):
    hits = [(k, v) for k, v in rows.items() if all(s in k for s in subs)]
    if not hits:
        return None, None
    return max(hits, key=lambda kv: kv[1]["calls"])


def main():
    prof, out = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r06"
    bench = parse(prof + "/summary_kernel_trace.txt")
    blocks = parse(prof + "/summary_blocks_kernel_trace.txt")
    pmc = parse_pmc(prof + "/summary_pmc_blocks.txt")
    n28, n26, n24 = 1 << 28, 1 << 26, 1 << 24
    # (row, trace, kernel substrings, algorithmic bytes per sample, samples per launch, flops per sample or 0)
    table = [
        ("FIR 128 real taps cf32, overlap-save (headline, configs[1])", bench, ("fir_fft_kernel<2, 0>",), 16, n28, 125),
        ("FIR 128 real taps cf32, direct form (bit-exact)", bench, ("fir_mfma_persistent_kernel<2, 1, 8",), 16, n28, 512),
        ("streaming yardstick: MultiplyConstant cf32 on the headline buffers", bench, ("multiply_constant_vec4",), 16, n28, 0),
        ("WBFM mono receiver, ONE launch (configs[2])", bench, ("rx_fused_kernel<0>",), 8.16, n26, 167),
        ("PolyphaseChannelizer K=64, 1024 taps (configs[4])", bench, ("channelizer_kernel",), 16, n24, 8192),
        ("MultiplyConstant cf32 (2^26)", blocks, ("multiply_constant_vec4",), 16, n26, 0),
        ("FIR 128 real taps cf32, overlap-save (2^26)", blocks, ("fir_fft_kernel<2, 0>",), 16, n26, 125),
        ("FIR 128 real taps f32, overlap-save (2^26)", blocks, ("fir_fft_kernel<1, 0>",), 8, n26, 63),
        ("FIR 128 real taps cf32, direct (2^26)", blocks, ("fir_mfma_persistent_kernel<2, 1, 8",), 16, n26, 512),
        ("FIR 128 real taps f32, direct (2^26)", blocks, ("fir_mfma_persistent_kernel<1, 1, 8",), 8, n26, 256),
        ("FIR 16 real taps cf32 (window kernel)", blocks, ("fir_win_cplx_kernel<1, 5, 16, 0>",), 16, n26, 64),
        ("FrequencyTranslator", blocks, ("rotator_kernel",), 16, n26, 0),
        ("FrequencyDiscriminator", blocks, ("fmdiscrim",), 12, n26, 0),
        ("Downsampler(5) cf32", blocks, ("downsample_kernel<HIP_vector_type<float, 2u>",), 9.6, n26, 0),
        ("Downsampler(5) f32", blocks, ("downsample_kernel<float>",), 4.8, n26, 0),
        ("FMDeemphasis f32 (first-order IIR, one launch)", blocks, ("iir_stream_kernel<1, 1, 2>",), 8, n26, 0),
        ("IIR 5 ff / 3 fb cf32", blocks, ("iir_stream_kernel<2, 2, 8>",), 16, n26, 0),
        ("IIR 5 ff / 3 fb f32", blocks, ("iir_stream_kernel<1, 2, 8>",), 8, n26, 0),
        ("IIR biquad cf32", blocks, ("iir_stream_kernel<2, 2, 4>",), 16, n26, 0),
        ("Decimator(5) cf32, fused", blocks, ("fir_mfma_persistent_kernel<2, 5, 2, false, 51, 0",), 9.6, n26, 102.4),
        ("Tuner(-250k, 200k, 5), fused rotator + FIR + downsampler", blocks, ("fir_mfma_persistent_kernel<2, 5, 2, true, 51, 0, false, 4, 0",), 9.6, n26, 108.4),
        ("Tuner(-100k, 10k, 50): LDS-staged decimator with rotator (AM / SSB / NBFM receivers)", blocks, ("fir_decim_lds2_kernel<true, 0, false>",), 8.16, n26, 16.2),
        ("Tuner(-100k, 12k, 80): the same with the phase-array layout (rtlsdr_pocsag.lua / rtlsdr_ax25.lua)", blocks, ("fir_decim_lds2_kernel<true, 0, true>",), 8.1, n26, 12.4),
        ("Decimator(25) cf32: LDS-staged decimator", blocks, ("fir_decim_lds2_kernel<false, 0, false>",), 8.32, n26, 20.5),
        ("Decimator / Tuner, polyphase FFT overlap-save", blocks, ("fir_decfft_kernel<5, 0>",), 9.6, n26, 62),
        ("Upsampler(5) cf32: the Interpolator's bytes without arithmetic (input samples)", blocks, ("upsample_vec_kernel<HIP_vector_type<float, 2u>",), 48, n26, 0),
        ("Interpolator(5) cf32 (input samples)", blocks, ("fir_interp_kernel<5, 26>",), 48, n26, 512),
        ("RationalResampler(3, 2) cf32 (input samples)", blocks, ("fir_rational_kernel<3, 2",), 20, n26, 256),
        ("RationalResampler(2, 3) cf32", blocks, ("fir_rational_kernel<2, 3",), 8 + 16 / 3, n26, 0),
        ("RationalResampler(4, 3) cf32", blocks, ("fir_rational_kernel<4, 3",), 8 + 32 / 3, n26, 0),
        ("RationalResampler(3, 4) cf32", blocks, ("fir_rational_kernel<3, 4",), 14, n26, 0),
        ("RationalResampler(5, 4) cf32", blocks, ("fir_rational_kernel<5, 4",), 18, n26, 0),
        ("RationalResampler(4, 5) cf32", blocks, ("fir_rational_kernel<4, 5",), 14.4, n26, 0),
        ("HilbertTransform(65) f32 -> cf32 (window kernel, zero taps skipped)", blocks, ("hilbert_win_kernel<65>",), 12, n26, 64),
        ("HilbertTransform(129) f32 -> cf32", blocks, ("hilbert_win_kernel<129>",), 12, n26, 128),
        ("IQ records u8 -> cf32", blocks, ("format_convert_vec_kernel<unsigned char, unsigned char",), 10, n26, 0),
        ("IQ records s16le -> cf32", blocks, ("format_convert_vec_kernel<unsigned short, short",), 12, n26, 0),
        ("IQ records f32be -> cf32", blocks, ("format_convert_vec_kernel<unsigned int, float, true",), 16, n26, 0),
        ("FIR 1276 real taps cf32, overlap-save, one 4096-point launch (a wave per block, 64 x 64)", blocks, ("fir_fft64_kernel<1280, 8, 1, 2>",), 16, n26, 180),
        ("FIR 768 real taps cf32, overlap-save, one 4096-point launch (a wave per block, 64 x 64)", blocks, ("fir_fft64_kernel<768, 8, 1, 2>",), 16, n26, 160),
        ("FIR 1276 complex taps cf32, overlap-save, 4096-point launch (a wave per block, four waves per CU, full H)", blocks, ("fir_fft64_kernel<1280, 4, 1, 2>",), 16, n26, 180),
        ("FIR 1276 real taps f32, overlap-save: 64 x 64 kernel, two stream blocks per transform (round 6; round 5: partitioned, 0.322)", blocks, ("fir_fft64_kernel<1280, 8, 1, 1>",), 8, n26, 250),
        ("FIR 768 real taps f32, overlap-save: 64 x 64 kernel, two stream blocks per transform", blocks, ("fir_fft64_kernel<768, 8, 1, 1>",), 8, n26, 160),
        ("FIR 2048 real taps f32, overlap-save: 64 x 64 kernel at an overlap of 2 048", blocks, ("fir_fft64_kernel<2048, 8, 1, 1>",), 8, n26, 240),
        ("FIR 4096 real taps f32, overlap-save: two partitions of the 64 x 64 kernel, ONE launch", blocks, ("fir_fft64_kernel<2048, 4, 2, 1>",), 8, n26, 420),
        ("FIR 4096 real taps cf32, overlap-save: two partitions of the 64 x 64 kernel, ONE launch", blocks, ("fir_fft64_kernel<2048, 4, 2, 2>",), 16, n26, 420),
        ("FIR 2048 real taps cf32, overlap-save: 64 x 64 kernel at an overlap of 2 048", blocks, ("fir_fft64_kernel<2048, 8, 1, 2>",), 16, n26, 240),
        ("WBFM mono receiver from u8 IQ records, ONE launch (bench_blocks: noise input)", blocks, ("rx_fused_kernel<1>",), 2.16, n26, 167),
        ("Tuner from u8 IQ records (fan-out branch fed from a file), ONE launch", blocks, ("fir_mfma_persistent_kernel<2, 5, 2, true, 51, 0, false, 4, 1",), 3.6, n26, 108.4),
        ("Tuner(decimation 50) from u8 IQ records, ONE launch (AM / SSB / NBFM receivers fed from a file)", blocks, ("fir_decim_lds2_kernel<true, 1, false>",), 2.16, n26, 16.2),
        ("PSD N=1024 hamming log fftshift", blocks, ("spectrum1024_kernel",), 12, n26, 58),
        ("WBFM mono receiver (bench_blocks: U(-1,1) noise input)", blocks, ("rx_fused_kernel<0>",), 8.16, n26, 167),
    ]
    # un-profiled bench lines of the SAME lease (tools/profile_round.sh runs bench.py before and after the traces): what the driver's clock would have seen
    driver = {}
    for tagname in ("before", "after"):
        for wl in ("fir", "wbfm"):
            try:
                line = [l for l in open("%s/bench_%s_%s.json" % (prof, wl, tagname)) if l.startswith("{")][0]
                d = json.loads(line)
                driver.setdefault(wl, {})[tagname] = {"ms_per_step": d["ms_per_step"], "launch_ms_hip_events": d["roofline"]["launch_ms_hip_events"], "frac": d["roofline"]["frac"]}
            except (OSError, IndexError, KeyError, ValueError):
                pass
    rows = []
    for name, src, subs, bps, n, flops in table:
        k, v = find(src, *subs)
        if not k:
            continue
        us = v["avg_timed_us"]
        alg = bps * n
        row = {"row": name, "kernel": k, "samples_per_launch": n, "algorithmic_bytes_per_sample": round(bps, 4), "algorithmic_bytes_per_launch": round(alg),
               "avg_us": us, "min_us": v["min_timed_us"], "dispatches": v["calls"], "GB/s": round(alg / us / 1e3, 1), "frac_of_8TB/s": round(alg / (us * 1e-6) / HBM, 4)}
        if src is bench and "(headline" in name and "fir" in driver:
            row["driver_like"] = driver["fir"]
        if src is bench and "configs[2]" in name and "wbfm" in driver:
            row["driver_like"] = driver["wbfm"]
        if flops:
            row["TFLOP/s"] = round(flops * n / (us * 1e-6) / 1e12, 2)
            row["frac_of_157.3TF"] = round(flops * n / (us * 1e-6) / FP32, 4)
        if src is blocks and pmc:
            # HBM traffic of the same kernel in the counter passes of the blocks table: FETCH_SIZE x 2 (gfx950 reports half of wide coalesced reads,
            # profiles/hbm_traffic.json) + WRITE_SIZE, KB
            hits = [c for kk, c in pmc.items() if all(t in kk for t in subs) and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
            if hits:
                tb = (2 * hits[0]["FETCH_SIZE"] + hits[0]["WRITE_SIZE"]) * 1024
                row["hbm_traffic_bytes"] = round(tb)
                row["traffic_over_algorithmic"] = round(tb / alg, 3)
        rows.append(row)
    doc = {"_comment": "per-kernel rooflines of round %s: avg_us = average kernel duration over the TIMED dispatches (the last K of a kernel in its rocprofv3 kernel trace: "
                       "bench.py's 20 timed steps / bench_blocks.py's 10 repetitions; summarize_rocpd.py --last), one MI355X; frac = algorithmic bytes (SURVEY.md 8d x samples) / "
                       "avg_us / 8 TB/s.  Sources: gpurun_out/prof_%s/summary_*.txt (copied to profiles/%s_*kernel_trace.txt); the raw rocpd databases of the traces stay under "
                       "gpurun_out/prof_%s/kt*/" % (tag, tag, tag, tag),
           "driver_like": "rows of the two bench workloads carry `driver_like`: bench.py's own line (ms_per_step = wall clock, launch_ms_hip_events, roofline.frac) run UN-profiled "
                          "on the same lease directly before and after the traces - kernel_us (avg_us) and the driver-style step are then from one box (VERDICT r03 next 9)",
           "rows": rows}
    json.dump(doc, open(out, "w"), indent=1)
    for r in rows:
        print("%-70s %9.1f us  %7.1f GB/s  %.3f" % (r["row"], r["avg_us"], r["GB/s"], r["frac_of_8TB/s"]))


if __name__ == "__main__":
    main()
