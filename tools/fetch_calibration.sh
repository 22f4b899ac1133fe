#!/bin/bash
# three --pmc FETCH_SIZE passes of tools/fetch_calibration.py (input pointer 0 / 32 / 8 bytes off a 128-byte line); prints FETCH_SIZE x 2 KB over the bytes read
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for off in 0 32 8; do
    rm -rf /tmp/fcal_$off
    timeout 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/fcal_$off/fetch -o p -- python $ROOT/tools/fetch_calibration.py --offset-bytes $off > /tmp/fcal_$off.log 2>&1 || { echo "pass $off failed"; tail -5 /tmp/fcal_$off.log; }
    (cd $ROOT && python profiles/summarize_rocpd.py /tmp/fcal_$off lrhip 2>&1 | grep FETCH_SIZE | awk -v off=$off '{ printf "offset %2d B: %s  FETCH_SIZE %.1f KB  x2KB / (8 B x 2^26) = %.4f\n", off, $2, $(NF-2), $(NF-2) * 2048 / (8 * 67108864) }')
done
