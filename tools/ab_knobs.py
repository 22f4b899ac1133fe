#!/usr/bin/env python3
"""Same-box A/B of run-time knobs (environment variables read per launch by liblrhip.so) on the headline FIR.
usage: ab_knobs.py KNOB v1 v2 ... [--log2-samples 28] [--mode fft|direct]   -> one line per value, alternating twice."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knob")
    ap.add_argument("values", nargs="+")
    ap.add_argument("--log2-samples", type=int, default=28)
    ap.add_argument("--mode", default="fft")
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import luaradio_amd as lr
    from luaradio_amd import types

    lr.init(0)
    L = lr._lib.load()
    lr.adopt_torch_stream()
    n = 1 << args.log2_samples
    S = 1 if args.real else 2
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(S * n, device="cuda", generator=g) * 2 - 1
    y = torch.empty(S * n, device="cuda")
    blk = lr.LowpassFilterBlock(128, 15e3)
    blk.use_fft = 2 if args.mode == "fft" else 0
    blk.rate = 220500.0
    blk.differentiate([types.Float32 if args.real else types.ComplexFloat32])
    blk.initialize()

    def run():
        blk.process_device(x.data_ptr(), n, y.data_ptr(), n)

    for rnd in range(2):
        for v in args.values:
            os.environ[args.knob] = v
            run(); run()
            torch.cuda.synchronize()
            t = L.lrhip_timer_create()
            L.lrhip_timer_start(t)
            for _ in range(args.reps):
                run()
            L.lrhip_timer_stop(t)
            ms = L.lrhip_timer_elapsed_ms(t) / args.reps
            L.lrhip_timer_destroy(t)
            print("%s=%-6s round %d: %.4f ms  %.1f GS/s  %.1f GB/s" % (args.knob, v, rnd, ms, n / ms / 1e6, 4 * S * 2 * n / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
