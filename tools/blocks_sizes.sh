#!/bin/bash
# the per-block table (tools/bench_blocks.py) at small launch sizes next to each other: which rows carry a fixed cost that a 2^20-sample LuaRadio batch would pay
#   tools/blocks_sizes.sh [log2a log2b ...]   -> one line per row: ms at each size
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SIZES=${@:-20 22 24}
mkdir -p "$ROOT/gpurun_out"
for lg in $SIZES; do python "$ROOT/tools/bench_blocks.py" --log2-samples $lg --reps 30 2>/dev/null | grep '^{"block"' > "$ROOT/gpurun_out/blocks_2p$lg.jsonl"; done
python - "$ROOT" $SIZES <<'PY'
import json, sys
root, sizes = sys.argv[1], sys.argv[2:]
tabs = [{json.loads(l)["block"]: json.loads(l) for l in open("%s/gpurun_out/blocks_2p%s.jsonl" % (root, s))} for s in sizes]
print("%-100s" % "block (ms per launch)" + "".join("   2^%-4s" % s for s in sizes))
for k in tabs[0]:
    print("%-100s" % k[:100] + "".join(" %8.4f" % t[k]["ms"] if k in t else "        -" for t in tabs))
PY
