#!/bin/bash
# kernel durations of the WBFM receiver under the current environment: tools/kt_wbfm.sh <tag>
TAG=${1:-k}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o k -- python $ROOT/tools/time_wbfm.py 1 > $OUT/log.txt 2>&1; tail -3 $OUT/log.txt
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/.." 2>&1 | grep -E "lrhip::" | head -4
find "$OUT" -name "*.db" -delete
