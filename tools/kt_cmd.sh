#!/bin/bash
# rocprofv3 kernel trace of one command -> gpurun_out/kt_<tag>/summary.txt:  tools/kt_cmd.sh <tag> <command...>
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kt_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- bash -c 'cd "$0" && "$@"' "$ROOT" "$@" > "$OUT/cmd.log" 2>&1
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -size +8M -delete
grep lrhip "$OUT/summary.txt" | head -40
