#!/bin/bash
# same-box A/B of the single-launch FM receiver (kernels_rx.h): workgroups per CU, ablation bits, the two-launch form
for w in 3 2; do TAG=wgs$w LRHIP_RX_WGS_PER_CU=$w python tools/time_wbfm.py 2; done
for d in 1 2 4 3 7; do TAG=dbg$d LRHIP_RX_DBG=$d python tools/time_wbfm.py 2; done
TAG=two LRHIP_NO_SINGLE_LAUNCH=1 python tools/time_wbfm.py 2
