#!/bin/bash
# same-box A/B of a run-time knob (environment variable) on rows of tools/bench_blocks.py, three alternations
#   usage: tools/ab_env_blocks.sh VAR "v1 v2" "<row substring>" [log2-samples]        (value "-" = variable unset)
ROOT=$(pwd)
var=$1; vals=$2; rows=$3; lg=${4:-26}
for rnd in 1 2 3; do
  for v in $vals; do
    ( if [ "$v" != "-" ]; then export $var=$v; fi
      python tools/bench_blocks.py --log2-samples $lg --only "$rows" 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print('alt $rnd $var=%-3s %-80s %.4f ms  frac %.3f' % ('$v', d.get('block', '?')[:80], d.get('ms', 0), d.get('frac_8TB/s', 0)))
" )
  done
done
