#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{kt,pmc*}/... + gpurun_out/prof_<tag>/summary_*.txt
# rocprofv3 is run from /tmp with TMPDIR=/tmp; --pmc passes never carry a trace option.  The kernel traces use the bench's own default
# steps / warm-up (20 / 3) so that the per-kernel average can be held against the bench line's ms_per_step.
set -u
TAG=${1:-r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-verify --headline-only"
run() { timeout 240 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
# kernel traces (durations): headline FFT form, bit-exact direct form, WBFM chain
run --kernel-trace --stats -d "$OUT/kt/fft" -o fft -- $B --steps 20 --warmup 3
run --kernel-trace --stats -d "$OUT/kt/direct" -o direct -- $B --steps 20 --warmup 3 --fir-mode direct
run --kernel-trace --stats -d "$OUT/kt/wbfm" -o wbfm -- $B --steps 20 --warmup 3 --workload wbfm
# counters, one small group per pass
P="$B --steps 3 --warmup 1"
D="$P --fir-mode direct"
W="$P --workload wbfm"
for spec in "fir:$P" "direct:$D" "wbfm:$W"; do
    name=${spec%%:*}; cmd=${spec#*:}
    run --pmc FETCH_SIZE -d "$OUT/pmc_$name/fetch" -o p -- $cmd
    run --pmc WRITE_SIZE -d "$OUT/pmc_$name/write" -o p -- $cmd
    run --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_$name/sq1" -o p -- $cmd
    run --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d "$OUT/pmc_$name/sq2" -o p -- $cmd
    run --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OUT/pmc_$name/sq3" -o p -- $cmd
    run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_$name/grbm" -o p -- $cmd
done
run --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ -d "$OUT/pmc_fir/tcp" -o p -- $P
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/kt" > "$OUT/summary_kernel_trace.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_fir" fir_ > "$OUT/summary_pmc_fir.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_direct" fir_ > "$OUT/summary_pmc_direct.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_wbfm" lrhip > "$OUT/summary_pmc_wbfm.txt" 2>&1
# the rocpd databases are large (gpurun copies at most 64 MiB back): keep the summaries only
find "$OUT" -name "*.db" -delete
tail -n 45 "$OUT/summary_kernel_trace.txt"
