#!/usr/bin/env python3
"""the reference's example receivers as device chains on 2^26 RF samples (ComplexFloat32, resident in HBM): ms per chunk, launches, GS/s"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import luaradio_amd as lr
from luaradio_amd import composites as C
lr.init(0)
L = lr._lib.load()
lr.adopt_torch_stream()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(2 * n, device="cuda", generator=g) * 2 - 1
y = torch.empty(n, device="cuda")
for name, mk in (("wbfm_mono (rtlsdr_wbfm_mono.lua)", lambda: C.wbfm_mono_receiver()), ("am_envelope (rtlsdr_am_envelope.lua)", lambda: C.am_envelope_receiver()),
                 ("ssb usb (rtlsdr_ssb.lua)", lambda: C.ssb_receiver("usb")), ("nbfm (rtlsdr_nbfm.lua)", lambda: C.nbfm_receiver())):
    r = mk()
    cap = r.max_output(n)
    for _ in range(3): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    t = L.lrhip_timer_create(); L.lrhip_timer_start(t)
    for _ in range(10): r.process_device(x.data_ptr(), n, y.data_ptr(), cap)
    L.lrhip_timer_stop(t); ms = L.lrhip_timer_elapsed_ms(t) / 10; L.lrhip_timer_destroy(t)
    print(json.dumps({"receiver": name, "log2_samples": log2n, "ms": round(ms, 4), "launches": r.chain.last_launches, "GS/s": round(n / ms / 1e6, 1)}), flush=True)
