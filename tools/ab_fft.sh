#!/bin/bash
# same-box A/B of the headline overlap-save kernel builds (luaradio_amd/ab/liblrhip_<variant>.so, LRHIP_LIB_PATH):
#   base   4 waves per workgroup x 3 workgroups per CU, one block per wave            (the shipped kernel)
#   w16    one 1024-thread workgroup = 16 waves per CU
#   w16p   ... + register prefetch of the next block
#   nb2    TWO blocks in flight per wave (stage-interleaved, own exchange buffers), 8 waves per CU
#   nb2p   ... + register prefetch
# usage: tools/ab_fft.sh [log2-samples ...]
ROOT=$(pwd)
for lg in ${@:-28}; do
  for rnd in 1 2; do
    for v in base w16 w16p nb2 nb2p; do
      lib=$ROOT/luaradio_amd/ab/liblrhip_$v.so; [ $v = base ] && lib=$ROOT/luaradio_amd/liblrhip.so
      LRHIP_LIB_PATH=$lib python tools/ab_knobs.py LRHIP_DUMMY $v --log2-samples $lg --reps 20 2>/dev/null | grep "round 1" | sed "s/^/2^$lg /"
    done
  done
done
