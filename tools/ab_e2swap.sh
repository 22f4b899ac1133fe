#!/bin/bash
# same-box A/B of the headline overlap-save kernel with its inner transposes in registers (LRHIP_FFT_E2_SWAP=1 build, luaradio_amd/ab/liblrhip_e2swap.so)
# against the shipped build: three alternations per size.   usage: tools/ab_e2swap.sh [log2-samples ...]
ROOT=$(pwd)
for lg in ${@:-28 26}; do
  for rnd in 1 2 3; do
    for v in base ${AB_VARIANT:-e2swap}; do
      lib=$ROOT/luaradio_amd/ab/liblrhip_$v.so; [ $v = base ] && lib=$ROOT/luaradio_amd/liblrhip.so
      LRHIP_LIB_PATH=$lib python tools/ab_knobs.py LRHIP_DUMMY $v --log2-samples $lg --reps 20 2>/dev/null | grep "round 1" | sed "s/^/2^$lg alt $rnd /"
    done
  done
done
