#!/bin/bash
# tools/rx_input_dependence.sh -> gpurun_out/rx_inputs/summary.txt: per input, the un-profiled time, clock / power samples during the loop, and the
# counters that separate "more instructions" from "fewer cycles per second" (GRBM_GUI_ACTIVE = busy cycles, SQ_INSTS_* = wave instructions issued)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/rx_inputs
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for inp in fm noise fm_small noise_small const zeros; do
  echo "=== $inp" >> $OUT/summary.txt
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/smi_$inp.txt 2>&1 &
  SMI=$!
  INPUT=$inp ITERS=3000 timeout 120 python tools/rx_input_dependence.py >> $OUT/summary.txt 2>&1
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  tail -4 $OUT/smi_$inp.txt >> $OUT/summary.txt
  cd /tmp
  INPUT=$inp ITERS=20 RAMP=30 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc_$inp/grbm" -o p -- python $ROOT/tools/rx_input_dependence.py > /dev/null 2>&1
  INPUT=$inp ITERS=20 RAMP=30 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_$inp/insts" -o p -- python $ROOT/tools/rx_input_dependence.py > /dev/null 2>&1
  INPUT=$inp ITERS=20 RAMP=30 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d "$OUT/pmc_$inp/cyc" -o p -- python $ROOT/tools/rx_input_dependence.py > /dev/null 2>&1
  INPUT=$inp ITERS=20 RAMP=30 timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/pmc_$inp/kt" -o k -- python $ROOT/tools/rx_input_dependence.py > /dev/null 2>&1
  cd "$ROOT"
  python profiles/summarize_rocpd.py "$OUT/pmc_$inp" rx_fused --last 20 >> $OUT/summary.txt 2>&1
  rm -rf "$OUT/pmc_$inp"
done
cat $OUT/summary.txt
