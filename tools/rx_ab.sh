#!/bin/bash
# same-box A/B of the single-launch FM receiver (kernels_rx.h): ablation bits (1 audio product, 2 discriminator, 4 tuner MFMA, 8 HBM reads), the two-launch form
TAG=full python tools/time_wbfm.py 2
for d in 8 15 7 4 12; do TAG=dbg$d LRHIP_RX_DBG=$d python tools/time_wbfm.py 2; done
for w in 4 2; do TAG=wgs${w}dbg15 LRHIP_RX_WGS_PER_CU=$w LRHIP_RX_DBG=15 python tools/time_wbfm.py 2; done
TAG=two LRHIP_NO_SINGLE_LAUNCH=1 python tools/time_wbfm.py 2
