#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{kt_*,pmc_*}/... + gpurun_out/prof_<tag>/summary_*.txt
# rocprofv3 is run from /tmp with TMPDIR=/tmp; --pmc passes never carry a trace option.
set -u
TAG=${1:-r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline"
run() { timeout 300 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
# kernel traces (durations): headline FFT form, bit-exact direct form, WBFM chain
run --kernel-trace --stats -d "$OUT/kt/fft" -o fft -- $B --steps 10 --warmup 2
run --kernel-trace --stats -d "$OUT/kt/direct" -o direct -- $B --steps 10 --warmup 2 --fir-mode direct
run --kernel-trace --stats -d "$OUT/kt/wbfm" -o wbfm -- $B --steps 10 --warmup 2 --workload wbfm
# counters, one small group per pass (headline form)
P="$B --steps 3 --warmup 1"
run --pmc FETCH_SIZE -d "$OUT/pmc/fetch" -o p -- $P
run --pmc WRITE_SIZE -d "$OUT/pmc/write" -o p -- $P
run --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc/sq1" -o p -- $P
run --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d "$OUT/pmc/sq2" -o p -- $P
run --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$OUT/pmc/sq3" -o p -- $P
run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc/grbm" -o p -- $P
run --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ -d "$OUT/pmc/tcp" -o p -- $P
# bit-exact direct form and the WBFM chain: matrix-pipe busy cycles against the clock
D="$B --steps 3 --warmup 1 --fir-mode direct"
W="$B --steps 3 --warmup 1 --workload wbfm"
run --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_direct/sq" -o p -- $D
run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_direct/grbm" -o p -- $D
run --pmc FETCH_SIZE -d "$OUT/pmc_direct/fetch" -o p -- $D
run --pmc WRITE_SIZE -d "$OUT/pmc_direct/write" -o p -- $D
run --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_wbfm/sq" -o p -- $W
run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_wbfm/grbm" -o p -- $W
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/pmc_direct" fir_mfma > "$OUT/summary_pmc_direct.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_wbfm" lrhip > "$OUT/summary_pmc_wbfm.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/kt" > "$OUT/summary_kernel_trace.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc" fir_ > "$OUT/summary_pmc_fir.txt" 2>&1
tail -n 60 "$OUT/summary_kernel_trace.txt"
cat "$OUT/summary_pmc_fir.txt" "$OUT/summary_pmc_direct.txt" "$OUT/summary_pmc_wbfm.txt"
