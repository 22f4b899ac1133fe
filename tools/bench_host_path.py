#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer path (what a LuaRadio process() sees): host numpy vectors in,
host vectors out, through lrhip_chain_execute (synchronous, one chunk at a time) and through the pinned ring
(lrhip_chain_submit / lrhip_chain_collect, depth 3).  Never reported as bench.py's `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import luaradio_amd as lr
    from luaradio_amd import types

    lr.init(0)
    rng = np.random.default_rng(1)
    total = 1 << 25
    x = (rng.uniform(-1, 1, total) + 1j * rng.uniform(-1, 1, total)).astype(np.complex64)

    def fir_chain():
        c = lr.CompositeBlock()
        b = lr.LowpassFilterBlock(128, 15e3)
        b.use_fft = 2
        c.connect(b)
        c.rate = 220500.0
        c.differentiate([types.ComplexFloat32])
        c.initialize()
        return c

    rows = []
    for name, make in (("LowpassFilter(128) cf32", fir_chain), ("WBFM mono chain", lambda: lr.wbfm_mono_receiver(1102500.0, -250e3))):
        for chunk in (8192, 131072, 1 << 20, 1 << 22):
            chunks = [x[a:a + chunk] for a in range(0, total, chunk)]
            blk = make()
            blk.process(chunks[0])
            t0 = time.perf_counter()
            for c in chunks:
                blk.process(c)
            t_sync = time.perf_counter() - t0
            blk = make()
            blk.chain.set_ring(3, chunk)
            blk.chain.submit(chunks[0]); blk.chain.collect()
            t0 = time.perf_counter()
            n = 0
            for out in blk.chain.stream(chunks, depth=3):
                n += len(out)
            t_ring = time.perf_counter() - t0
            rows.append({"chain": name, "chunk_samples": chunk, "sync_MS/s": round(total / t_sync / 1e6, 1),
                         "ring3_MS/s": round(total / t_ring / 1e6, 1), "ring3_GB/s_h2d": round(8 * total / t_ring / 1e9, 2)})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
