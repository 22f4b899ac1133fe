#!/bin/bash
# same-box A/B of two library builds on rows of tools/bench_blocks.py, three alternations.
#   usage: tools/ab_blocks.sh <variant .so under luaradio_amd/ab/, without lib prefix> "<row substrings, comma separated>" [log2-samples]
ROOT=$(pwd)
v=$1; rows=$2; lg=${3:-26}
for rnd in 1 2 3; do
  for w in base $v; do
    lib=$ROOT/luaradio_amd/ab/liblrhip_$w.so; [ $w = base ] && lib=$ROOT/luaradio_amd/liblrhip.so
    LRHIP_LIB_PATH=$lib python tools/bench_blocks.py --log2-samples $lg --only "$rows" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print('alt $rnd %-7s %-90s %.4f ms  frac %.3f' % ('$w', d.get('block', d.get('name', '?'))[:90], d.get('ms', 0), d.get('frac_8TB/s', 0)))
"
  done
done
