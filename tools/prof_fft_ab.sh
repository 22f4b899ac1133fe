#!/bin/bash
# counters for the headline-kernel A/B (tools/ab_fft.sh variants): per variant a kernel trace + four small PMC passes
#   -> gpurun_out/prof_fftab/<variant>/summary.txt
ROOT=$(pwd)
export TMPDIR=/tmp
for v in ${@:-base w16p nb2}; do
  lib=$ROOT/luaradio_amd/ab/liblrhip_$v.so; [ $v = base ] && lib=$ROOT/luaradio_amd/liblrhip.so
  OUT=$ROOT/gpurun_out/prof_fftab/$v
  rm -rf "$OUT"; mkdir -p "$OUT"
  CMD="python $ROOT/bench.py --no-cpu-baseline --no-verify --headline-only --steps 20 --warmup 3"
  cd /tmp
  run() { LRHIP_LIB_PATH=$lib timeout 300 rocprofv3 "$@" > /dev/null 2>&1 || echo "rocprofv3 $* failed rc=$?"; }
  run --kernel-trace --stats -d "$OUT/kt/a" -o a -- $CMD
  run --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc/sq1" -o p -- $CMD
  run --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/pmc/sq2" -o p -- $CMD
  run --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ -d "$OUT/pmc/tcp" -o p -- $CMD
  run --pmc SQ_WAVES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$OUT/pmc/sq3" -o p -- $CMD
  cd "$ROOT"
  python profiles/summarize_rocpd.py "$OUT/kt" > "$OUT/summary.txt" 2>&1
  python profiles/summarize_rocpd.py "$OUT/pmc" fir_fft >> "$OUT/summary.txt" 2>&1
  find "$OUT" -name "*.db" -delete
  echo "== $v"; grep -E "fir_fft_kernel|SQ_|TCP_|GRBM" "$OUT/summary.txt" | cut -c1-170
done
