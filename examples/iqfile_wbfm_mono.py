#!/usr/bin/env python3
"""Wideband FM broadcast (mono) receiver from an IQ recording - the reference's examples/rtlsdr_wbfm_mono.lua with the
RTL-SDR source replaced by IQFileSource (as its docs suggest for offline use), everything between the file read and the
WAV write on the MI355X:

    IQFileSource(file, 'u8', 1102500) -> Tuner(-250e3, 200e3, 5) -> FrequencyDiscriminator(1.25) -> Lowpass(128, 15e3)
        -> FMDeemphasis(75e-6) -> Downsampler(5) -> WAV (44.1 kHz, s16)

The raw u8 records cross PCIe as they are (2 bytes per complex sample) and are converted on the device; chunks go through
the pinned ring (H2D / kernels / D2H of neighbouring chunks overlap).

    python examples/iqfile_wbfm_mono.py capture.u8 out.wav [--format u8] [--rate 1102500] [--offset -250e3]
    python examples/iqfile_wbfm_mono.py --selftest        # synthesises a capture with two tones, demodulates, checks the spectrum
"""
import argparse
import os
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luaradio_amd as lr                     # noqa: E402
from luaradio_amd import types                # noqa: E402


def build_chain(path_or_bytes, fmt, rate, offset):
    src = lr.IQFileSource(path_or_bytes, fmt, rate)
    src.initialize()
    blocks = [src, lr.FrequencyTranslatorBlock(offset), lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5),
              lr.FrequencyDiscriminatorBlock(1.25), lr.LowpassFilterBlock(128, 15e3), lr.FMDeemphasisFilterBlock(75e-6), lr.DownsamplerBlock(5)]
    blocks[5].use_fft = 3                     # audio filter: automatic - in this chain it merges with the de-emphasis and the downsampler
    r, t = src.get_rate(), src.get_output_type()
    for b in blocks[1:]:                      # what CompositeBlock does before run(): types and rates downstream (composite.lua:443-470)
        b.rate = r
        b.differentiate([t])
        b.initialize()
        r, t = b.get_rate(), b.get_output_type()
    return src, lr.Chain(blocks), r


def demodulate(src, chain, chunk_records=1 << 20):
    """file -> pinned ring slot (readinto, no staging copy) -> device -> audio.  A recording on disk (a regular file) is read by the library itself
    (Chain.submit_fd: positional reads on its copy threads); an in-memory capture goes through readinto()."""
    if getattr(chain, "_ring_chunk", 0) != chunk_records:
        chain.set_ring(3, chunk_records)       # pinned host + device slots, allocated once
    parts = []
    if isinstance(src.file, str):
        fd, offset = src._fh.fileno(), 0
        while True:
            if chain.in_flight == 3:           # ring full: take the oldest chunk out first
                parts.append(chain.collect())
            got = chain.submit_fd(fd, offset, chunk_records)
            if got == 0:
                break
            offset += got * src.record_size
        while chain.in_flight:
            parts.append(chain.collect())
        return np.concatenate(parts)
    while True:
        view = chain.ring_input()
        if view is None:                       # ring full: take the oldest chunk out first
            parts.append(chain.collect())
            continue
        got = src._fh.readinto(view)
        records = (got or 0) // src.record_size
        if records == 0:
            break
        chain.submit(view[:records * src.record_size])
    while chain.in_flight:
        parts.append(chain.collect())
    return np.concatenate(parts)


def write_wav(path, audio, rate):
    pcm = np.clip(audio * 32767.0, -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(round(rate)))
        w.writeframes(pcm.tobytes())


def synth_capture(rate, offset, seconds=0.5):
    n = int(rate * seconds)
    t = np.arange(n) / rate
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * (-offset) * t + 2 * np.pi * 75e3 / rate * np.cumsum(m)
    iq = 0.8 * np.exp(1j * ph)
    u8 = np.empty(2 * n, np.uint8)
    u8[0::2] = np.clip(np.round(iq.real * 127.5 + 127.5), 0, 255)
    u8[1::2] = np.clip(np.round(iq.imag * 127.5 + 127.5), 0, 255)
    return u8.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("iqfile", nargs="?")
    ap.add_argument("wavfile", nargs="?")
    ap.add_argument("--format", default="u8")
    ap.add_argument("--rate", type=float, default=1102500.0)
    ap.add_argument("--offset", type=float, default=-250e3)
    ap.add_argument("--selftest", action="store_true")
    a = ap.parse_args()
    if a.selftest:
        src, chain, out_rate = build_chain(synth_capture(a.rate, a.offset), "u8", a.rate, a.offset)
        audio = demodulate(src, chain, 1 << 17)
        seg = audio[4000:]
        spec = np.abs(np.fft.rfft(seg * np.hanning(len(seg))))
        freqs = np.fft.rfftfreq(len(seg), 1 / out_rate)
        peak = freqs[int(np.argmax(spec))]
        band = (freqs > 4900) & (freqs < 5100)
        ok = abs(peak - 1e3) < 30 and spec[band].max() > 50 * np.median(spec)
        print("selftest: %d audio samples at %.0f Hz, strongest tone %.0f Hz, 5 kHz tone %s -> %s"
              % (len(audio), out_rate, peak, "present" if spec[band].max() > 50 * np.median(spec) else "missing", "OK" if ok else "FAILED"))
        return 0 if ok else 1
    if not a.iqfile or not a.wavfile:
        ap.error("need an IQ file and a WAV file (or --selftest)")
    src, chain, out_rate = build_chain(a.iqfile, a.format, a.rate, a.offset)
    audio = demodulate(src, chain)
    write_wav(a.wavfile, audio, out_rate)
    print("%d audio samples at %.0f Hz -> %s (%d kernel launches per chunk)" % (len(audio), out_rate, a.wavfile, chain.last_launches))
    return 0


if __name__ == "__main__":
    sys.exit(main())
