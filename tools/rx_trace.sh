#!/bin/bash
# Where a tile of the single-launch WBFM receiver (rx_fused_kernel, luaradio_amd/csrc/kernels_rx.h) spends its clocks: a variant library built with
# -DLRHIP_RX_TRACE stamps clock64() at the phase boundaries (lane 0 of every wave of the first eight workgroups, 64 tiles each) and prints the averages of
# the twelfth launch to stderr.  The stamps themselves cost ~30 % (0.206 against 0.156 ms): read the proportions, not the sum.
#   tools/rx_trace.sh            (on the GPU box; builds luaradio_amd/ab/liblrhip_rxtrace.so if it is missing: ~2 min)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/luaradio_amd/ab/liblrhip_rxtrace.so
if [ ! -f "$LIB" ]; then
    mkdir -p "$ROOT/luaradio_amd/ab"
    (cd "$ROOT/luaradio_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -pthread -DLRHIP_ABLATION -DLRHIP_RX_TRACE \
        -I ../../include -shared -o "$LIB" lrhip.hip) || exit 1
fi
LRHIP_LIB_PATH=$LIB python "$ROOT/bench.py" --workload wbfm --no-cpu-baseline 2>&1 | grep "rx trace"
