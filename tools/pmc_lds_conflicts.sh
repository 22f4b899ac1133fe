#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the headline kernel for a library build:  tools/pmc_lds_conflicts.sh <lib.so> <tag>
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_lds_$2
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
LRHIP_LIB_PATH=$1 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$OUT/pmc" -o p -- python $ROOT/bench.py --no-cpu-baseline --no-verify --headline-only --steps 3 --warmup 1 --log2-samples 26 > /dev/null 2>&1
cd "$ROOT"
python profiles/summarize_rocpd.py "$OUT/pmc" fir_fft | grep fir_fft | cut -c1-160
find "$OUT" -name "*.db" -delete
