#!/bin/bash
# same-box A/B of the single-launch FM receiver (kernels_rx.h): build variants (LRHIP_LIB_PATH), the two-launch form
ROOT=$(pwd)
for r in 1 2; do
  TAG=full python tools/time_wbfm.py 2
  for v in "$@"; do TAG=$v LRHIP_LIB_PATH=$ROOT/luaradio_amd/ab/liblrhip_$v.so python tools/time_wbfm.py 2; done
done
TAG=two LRHIP_NO_SINGLE_LAUNCH=1 python tools/time_wbfm.py 2
