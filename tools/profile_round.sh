#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{kt,pmc*}/... + gpurun_out/prof_<tag>/summary_*.txt
# rocprofv3 is run from /tmp with TMPDIR=/tmp; --pmc passes never carry a trace option.  The kernel traces use the bench's own default
# steps / warm-up (20 / 3); summarize_rocpd.py --last 20 reports each kernel's last 20 dispatches = the timed region, which is what the
# bench line's ms_per_step has to be held against.  The (small) kernel-trace databases are kept next to the summaries.
set -u
TAG=${1:-r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-verify --headline-only"
run() { timeout 240 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
# the un-profiled bench lines of THIS lease, before and after the traces (profile and driver numbers from one box)
lines() { ( cd $ROOT && timeout 300 python bench.py --no-cpu-baseline --headline-only > "$OUT/bench_fir_$1.json" 2> /dev/null; timeout 300 python bench.py --no-cpu-baseline --workload wbfm > "$OUT/bench_wbfm_$1.json" 2> /dev/null ); }
lines before
# kernel traces (durations): headline FFT form, bit-exact direct form, WBFM chain, channelizer
run --kernel-trace --stats -d "$OUT/kt/fft" -o fft -- $B --steps 20 --warmup 3
run --kernel-trace --stats -d "$OUT/kt/direct" -o direct -- $B --steps 20 --warmup 3 --fir-mode direct
run --kernel-trace --stats -d "$OUT/kt/wbfm" -o wbfm -- $B --steps 20 --warmup 3 --workload wbfm
run --kernel-trace --stats -d "$OUT/kt/chan" -o chan -- python $ROOT/tools/run_channelizer.py 20
# counters, one small group per pass
P="$B --steps 3 --warmup 1"
D="$P --fir-mode direct"
W="$P --workload wbfm"
C="python $ROOT/tools/run_channelizer.py 3"
for spec in "fir:$P" "direct:$D" "wbfm:$W" "chan:$C"; do
    name=${spec%%:*}; cmd=${spec#*:}
    run --pmc FETCH_SIZE -d "$OUT/pmc_$name/fetch" -o p -- $cmd
    run --pmc WRITE_SIZE -d "$OUT/pmc_$name/write" -o p -- $cmd
    run --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_$name/sq1" -o p -- $cmd
    run --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d "$OUT/pmc_$name/sq2" -o p -- $cmd
    run --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OUT/pmc_$name/sq3" -o p -- $cmd
    run --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_$name/grbm" -o p -- $cmd
done
run --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ -d "$OUT/pmc_fir/tcp" -o p -- $P
# per-block table under the tracer (tools/bench_blocks.py, 2^26 samples)
run --kernel-trace --stats -d "$OUT/kt_blocks/blocks" -o b -- python $ROOT/tools/bench_blocks.py --reps 10
lines after
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/kt" --last 20 > "$OUT/summary_kernel_trace.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/kt_blocks" --last 10 > "$OUT/summary_blocks_kernel_trace.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_fir" fir_ > "$OUT/summary_pmc_fir.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_direct" fir_ > "$OUT/summary_pmc_direct.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_wbfm" lrhip > "$OUT/summary_pmc_wbfm.txt" 2>&1
python profiles/summarize_rocpd.py "$OUT/pmc_chan" channelizer > "$OUT/summary_pmc_chan.txt" 2>&1
# keep the kernel-trace databases (small); the counter databases are large (gpurun copies at most 64 MiB back)
find "$OUT" -path "*pmc_*" -name "*.db" -delete
find "$OUT" -name "*.db" -size +14M -delete
du -sh "$OUT"
tail -n 30 "$OUT/summary_kernel_trace.txt"
