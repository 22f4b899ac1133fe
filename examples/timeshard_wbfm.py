#!/usr/bin/env python3
"""One long IQ recording, G GPUs: time partitions of the WBFM-mono receiver (examples/rtlsdr_wbfm_mono.lua's chain) - DESIGN.md section 6,
luaradio_amd/timeshard.py, include/lrhip.h "time-axis sharding".

Every partition seeks its chain to an aligned sample a little before its first own sample, replays that halo with the output thrown away and
then produces exactly the audio samples the single-process run would have produced for its range - no exchange between the partitions.

    python examples/timeshard_wbfm.py recording.cf32 audio.f32 [--parts 8]          # the partitions run one after the other on this GPU
    torchrun --nproc-per-node 8 examples/timeshard_wbfm.py recording.cf32 audio.f32 # one partition per GPU (rank r writes audio.f32.part<r>)
    python examples/timeshard_wbfm.py --selftest                                     # synthetic FM, 2 / 4 / 8 partitions == one stream, bit for bit (boundaries on the receiver's shard_align)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luaradio_amd as lr                     # noqa: E402
from luaradio_amd import timeshard            # noqa: E402

FS, OFFSET = 1102500.0, -250e3


def synth(n, seed=3):
    """an FM carrier at +250 kHz modulated by two tones, plus a little noise"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    ph = 2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / FS * np.cumsum(m)
    return (np.exp(1j * ph) + 0.01 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)


def partition_audio(x, a, b, rx=None):
    rx = rx or lr.wbfm_mono_receiver(FS, OFFSET)
    return timeshard.run_partition(rx, x, a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("recording", nargs="?")
    ap.add_argument("audio", nargs="?")
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--selftest", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    lr.init(int(os.environ.get("LOCAL_RANK", "0")))
    rx = lr.wbfm_mono_receiver(FS, OFFSET)
    align = rx.shard_align()
    if args.selftest:
        n = (1 << 21) + 4321
        x = synth(n)
        whole = lr.wbfm_mono_receiver(FS, OFFSET).process(x)
        for parts in (2, 4, 8):
            got = np.concatenate([partition_audio(x, a, b, rx) for a, b in timeshard.bounds(n, parts, align)])
            assert len(got) == len(whole) and float(np.max(np.abs(got - whole))) < 1e-7, parts      # single-launch receiver: to its 1e-10 warm-up (bit for bit with Chain(..., CHAIN_NO_SINGLE_LAUNCH))
        print("selftest ok: halo %d samples, boundaries on multiples of %d, 2 / 4 / 8 partitions equal one stream to 1e-7 (%d audio samples)"
              % (rx.halo(), align, len(whole)))
        return
    if not args.recording or not args.audio:
        ap.error("recording and audio file names, or --selftest")
    x = np.memmap(args.recording, dtype=np.complex64, mode="r")           # ComplexFloat32 records as RawFileSource reads them
    n = len(x)
    if world > 1:                                                          # one partition per process / GPU; no collective on the data path
        a, b = timeshard.rank_partition(n, world, rank, align)
        partition_audio(x, a, b, rx).tofile("%s.part%d" % (args.audio, rank))
    else:
        with open(args.audio, "wb") as f:
            for a, b in timeshard.bounds(n, args.parts, align):
                partition_audio(x, a, b, rx).tofile(f)


if __name__ == "__main__":
    main()
