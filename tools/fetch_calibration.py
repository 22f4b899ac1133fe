"""FETCH_SIZE calibration (MI355X_MICROARCH.md: "other access widths ... are uncalibrated: calibrate on a known byte count in your own access pattern").

One streaming block with exactly known traffic - MultiplyConstant on a ComplexFloat32 stream: 8 B read + 8 B written per sample - run with the input pointer
at a chosen byte offset from a 128-byte line: 0 and 32 take the 16-byte vector kernel (aligned / straddling lines), 8 takes the 8-byte-per-lane kernel
(the load shape of fir_decfft_kernel, whose windows start 32 B into a line).  Run under `rocprofv3 --pmc FETCH_SIZE`; tools/fetch_calibration.sh does the
three passes and prints counter x 2 KB / bytes read.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import luaradio_amd as lr
from luaradio_amd import types


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--offset-bytes", type=int, default=0)
    ap.add_argument("--log2-samples", type=int, default=26)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    n = 1 << a.log2_samples
    off = a.offset_bytes // 4
    xbuf = torch.rand(2 * n + 64, device="cuda") - 0.5
    ybuf = torch.empty(2 * n + 64, device="cuda")
    blk = lr.MultiplyConstantBlock(complex(0.5, 2.0))
    blk.rate = 2.0
    blk.differentiate([types.ComplexFloat32])
    blk.initialize()
    torch.cuda.synchronize()
    for _ in range(a.reps):
        got = blk.process_device(xbuf[off:].data_ptr(), n, ybuf[off:].data_ptr(), n)
        assert got == n
    torch.cuda.synchronize()
    print("offset", a.offset_bytes, "bytes read per launch", 8 * n)


if __name__ == "__main__":
    main()
