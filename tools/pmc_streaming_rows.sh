#!/bin/bash
# the streaming rows below 0.60 of the roof, one counter table each (which unit is busy) - the LDS-staged decimator (Tuner(..., 50)), the fused
# Toeplitz tuner (decimation 5), the polyphase-FFT decimator and the Float32-stream overlap-save kernel.  Counter passes only (no trace option), small groups.
#   tools/pmc_streaming_rows.sh <tag>  -> gpurun_out/prof_<tag>/summary_pmc_streaming_rows.txt
TAG=${1:-r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; rm -rf "$OUT/pmc_rows"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/bench_blocks.py --reps 3 --only Tuner(-100k,Tuner(-250k,polyphase,f32"
run() { timeout 300 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
run --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_rows/sq1" -o p -- $CMD
run --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d "$OUT/pmc_rows/sq2" -o p -- $CMD
run --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d "$OUT/pmc_rows/sq3" -o p -- $CMD
run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_rows/grbm" -o p -- $CMD
run --pmc FETCH_SIZE -d "$OUT/pmc_rows/fetch" -o p -- $CMD
run --pmc WRITE_SIZE -d "$OUT/pmc_rows/write" -o p -- $CMD
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/pmc_rows" lrhip > "$OUT/summary_pmc_streaming_rows.txt" 2>&1
find "$OUT/pmc_rows" -name "*.db" -delete
grep -c lrhip "$OUT/summary_pmc_streaming_rows.txt"
