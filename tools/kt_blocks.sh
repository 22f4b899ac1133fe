#!/bin/bash
# rocprofv3 kernel trace of tools/bench_blocks.py only -> gpurun_out/prof_<tag>/summary_blocks_kernel_trace.txt (tools/profile_round.sh does the same as its last step)
TAG=${1:-r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; rm -rf "$OUT/kt_blocks"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_blocks/blocks" -o b -- python $ROOT/tools/bench_blocks.py --reps 10 > "$OUT/blocks.log" 2>&1
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/kt_blocks" --last 10 > "$OUT/summary_blocks_kernel_trace.txt" 2>&1
grep block "$OUT/blocks.log" > "$OUT/blocks_table.jsonl"
grep -c lrhip "$OUT/summary_blocks_kernel_trace.txt"
