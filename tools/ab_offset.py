#!/usr/bin/env python3
"""Does the distance between the input and the output vector matter to the headline kernel (HBM channel / bank mapping)?  One 5 GiB allocation, x at its start,
y at 2 GiB + offset; same-box alternation.  usage: ab_offset.py [--log2-samples 28]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-samples", type=int, default=28)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import luaradio_amd as lr
    from luaradio_amd import types

    lr.init(0)
    L = lr._lib.load()
    lr.adopt_torch_stream()
    n = 1 << args.log2_samples
    big = torch.empty(2 * n * 2 + (1 << 28), device="cuda")            # floats: x (2n) | y (2n) | 1 GiB of slack
    g = torch.Generator(device="cuda").manual_seed(1)
    big[:2 * n].uniform_(-1, 1, generator=g)
    blk = lr.LowpassFilterBlock(128, 15e3)
    blk.use_fft = 2
    blk.rate = 220500.0
    blk.differentiate([types.ComplexFloat32])
    blk.initialize()
    offs = [0, 4096, 65536, (1 << 20) + 4096, (1 << 21) + 8192, 17 << 20, (64 << 20) + (1 << 12), (256 << 20) + (3 << 12), 1000003 * 16]
    for rnd in range(2):
        for off in offs:
            xp = big.data_ptr()
            yp = big.data_ptr() + 8 * n + off

            def run():
                blk.process_device(xp, n, yp, n)
            run(); run()
            torch.cuda.synchronize()
            t = L.lrhip_timer_create()
            L.lrhip_timer_start(t)
            for _ in range(args.reps):
                run()
            L.lrhip_timer_stop(t)
            ms = L.lrhip_timer_elapsed_ms(t) / args.reps
            L.lrhip_timer_destroy(t)
            print("y = x + 2^%d B + %10d B  round %d: %.4f ms  %.1f GS/s" % (args.log2_samples + 3, off, rnd, ms, n / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
