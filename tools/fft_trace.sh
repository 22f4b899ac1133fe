#!/bin/bash
# Where a block of the headline overlap-save kernel (fir_fft_kernel<2,0>, luaradio_amd/csrc/kernels_firfft.h) spends its clocks: a variant library built with
# -DLRHIP_FFT_TRACE stamps clock64() at the phase boundaries (lane 0 of every wave of the first eight workgroups, 30 blocks each) and prints the averages of the
# twelfth launch to stderr.  The stamps cost ~10 % (0.937 against 0.81-0.85 ms): read the proportions, not the sum.  profiles/r05_fft_phase_trace.txt is its output.
#   tools/fft_trace.sh            (on the GPU box; builds luaradio_amd/ab/liblrhip_ffttrace.so if it is missing: ~2 min)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/luaradio_amd/ab/liblrhip_ffttrace.so
if [ ! -f "$LIB" ]; then
    mkdir -p "$ROOT/luaradio_amd/ab"
    (cd "$ROOT/luaradio_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -pthread -DLRHIP_ABLATION -DLRHIP_FFT_TRACE \
        -I ../../include -shared -o "$LIB" lrhip.hip) || exit 1
fi
LRHIP_LIB_PATH=$LIB python "$ROOT/bench.py" --workload fir --no-cpu-baseline 2>&1 | grep "fft trace"
