#!/bin/bash
# HBM traffic counters of every kernel of tools/bench_blocks.py: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), no trace option
#   tools/pmc_blocks.sh <tag>  -> gpurun_out/prof_<tag>/summary_pmc_blocks.txt
TAG=${1:-r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; rm -rf "$OUT/pmc_blocks"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_blocks/fetch" -o p -- python $ROOT/tools/bench_blocks.py --reps 3 > /dev/null 2>&1 || echo "FETCH_SIZE pass failed"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_blocks/write" -o p -- python $ROOT/tools/bench_blocks.py --reps 3 > /dev/null 2>&1 || echo "WRITE_SIZE pass failed"
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/pmc_blocks" lrhip > "$OUT/summary_pmc_blocks.txt" 2>&1
find "$OUT/pmc_blocks" -name "*.db" -delete
grep -c lrhip "$OUT/summary_pmc_blocks.txt"
