#!/bin/bash
# Counter profile of one command (run on the GPU box through gpurun, from the repo root):
#   tools/prof_counters.sh <tag> <kernel-substring> -- <command...>
# -> gpurun_out/prof_<tag>/summary.txt : kernel durations and per-dispatch counter averages for the matching kernels.
# rocprofv3 runs from /tmp with TMPDIR=/tmp; --pmc passes never carry a trace option.
set -u
TAG=$1; FILT=$2; shift 3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="$*"
cd /tmp
run() { timeout 300 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
run --kernel-trace --stats -d "$OUT/kt/a" -o a -- bash -c "cd $ROOT && $CMD"
run --pmc FETCH_SIZE -d "$OUT/pmc/fetch" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc WRITE_SIZE -d "$OUT/pmc/write" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc/sq1" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d "$OUT/pmc/sq2" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$OUT/pmc/sq3" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d "$OUT/pmc/sq4" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc/sq5" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc/grbm" -o p -- bash -c "cd $ROOT && $CMD"
run --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ -d "$OUT/pmc/tcp" -o p -- bash -c "cd $ROOT && $CMD"
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/kt" > "$OUT/summary.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc" "$FILT" >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
