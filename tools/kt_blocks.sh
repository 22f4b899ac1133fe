#!/bin/bash
# rocprofv3 kernel traces of tools/bench_blocks.py and tools/bench_reference_suite.py -> gpurun_out/kt_blocks/summary.txt
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt_blocks
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/blocks" -o b -- python $ROOT/tools/bench_blocks.py > "$OUT/blocks.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/suite" -o s -- python $ROOT/tools/bench_reference_suite.py > "$OUT/suite.log" 2>&1
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -c lrhip "$OUT/summary.txt"
