#!/usr/bin/env python3
"""Derived shares per kernel from the counter summary of tools/pmc_streaming_rows.sh (profiles/summarize_rocpd.py output):
    python tools/derive_row_counters.py gpurun_out/prof_<tag>/summary_pmc_streaming_rows.txt [kernel substring ...]
valu_busy = SQ_INSTS_VALU x 4 issue cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)  (gfx950 has no VALU busy-cycle counter: instructions x wave64 issue)
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM / 8);  lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM / 8);  conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;  traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950 FETCH_SIZE correction, profiles/hbm_traffic.json)"""
import collections
import re
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(\w+)\s+(lrhip::.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m:
        rows[m.group(2)][m.group(1)] = float(m.group(4))
want = sys.argv[2:] or ["fir_decim_lds2_kernel<true, 0, false>", "fir_decim_lds2_kernel<true, 0, true>", "fir_decim_lds2_kernel<false, 0, false>",
                        "fir_mfma_persistent_kernel<2, 5, 2, true, 51, 0, false, 4, 0", "fir_decfft_kernel<5, 0>", "fir_fft_kernel<1, 0>", "fir_fft_kernel<2, 0>",
                        "fir_fft64_kernel<2048, 4, 2>", "fir_fft64_kernel<2048, 8, 1>", "fir_fft64_kernel<1280, 8, 1>", "fir_pols_kernel<1, 3>", "fir_pols_kernel<2, 4>"]
for w in want:
    for k, c in rows.items():
        if w in k and "GRBM_GUI_ACTIVE" in c:
            g = c["GRBM_GUI_ACTIVE"] / 8.0
            f = lambda name: c.get(name, float("nan"))
            print("%s" % k[:110])
            print("    valu_busy %.2f  mfma_busy %.2f  lds_busy %.2f  conflicts %.2f of the LDS cycles  wait_inst_any %.2f  wait_inst_lds %.2f | insts: valu %.1fM lds %.1fM salu %.1fM vmem_rd %.1fM"
                  " | traffic %.1f MB | GRBM %.2fM cycles" % (f("SQ_INSTS_VALU") * 4 / (1024 * g), f("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g), f("SQ_LDS_IDX_ACTIVE") / (256 * g),
                                                              f("SQ_LDS_BANK_CONFLICT") / max(f("SQ_LDS_IDX_ACTIVE"), 1.0), f("SQ_WAIT_INST_ANY") / f("SQ_WAVE_CYCLES"),
                                                              f("SQ_WAIT_INST_LDS") / f("SQ_WAVE_CYCLES"), f("SQ_INSTS_VALU") / 1e6, f("SQ_INSTS_LDS") / 1e6, f("SQ_INSTS_SALU") / 1e6,
                                                              f("SQ_INSTS_VMEM_RD") / 1e6, (2 * f("FETCH_SIZE") + f("WRITE_SIZE")) * 1024 / 1e6, c["GRBM_GUI_ACTIVE"] / 1e6))
            break
