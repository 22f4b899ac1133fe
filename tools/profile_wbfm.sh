#!/bin/bash
# short profile of the WBFM chain only (kernel trace + the SQ counter groups): tools/profile_wbfm.sh <tag> -> gpurun_out/prof_<tag>/summary_*.txt
set -u
TAG=${1:-w}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-verify --headline-only --workload wbfm"
run() { timeout 240 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
run --kernel-trace --stats -d "$OUT/kt/wbfm" -o wbfm -- $B --steps 20 --warmup 3
W="$B --steps 3 --warmup 1"
run --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_wbfm/sq1" -o p -- $W
run --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d "$OUT/pmc_wbfm/sq2" -o p -- $W
run --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OUT/pmc_wbfm/sq3" -o p -- $W
run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_wbfm/grbm" -o p -- $W
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/kt" > "$OUT/summary_kernel_trace.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_wbfm" lrhip > "$OUT/summary_pmc_wbfm.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -n 12 "$OUT/summary_kernel_trace.txt"; tail -n 30 "$OUT/summary_pmc_wbfm.txt"
