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
            for out in blk.chain.stream(chunks):
                n += len(out)
            t_ring = time.perf_counter() - t0
            # chunk coalescing: the same small vectors pushed into batches of 2^20 samples (lrhip_chain_push / _flush)
            t_push = None
            if chunk < (1 << 20):
                blk = make()
                blk.chain.set_ring(3, 1 << 20)
                blk.chain.push(chunks[0]); blk.chain.flush()
                t0 = time.perf_counter()
                n = 0
                for c in chunks:
                    n += len(blk.chain.push(c))
                n += len(blk.chain.flush())
                t_push = time.perf_counter() - t0
            rows.append({"chain": name, "chunk_samples": chunk, "sync_MS/s": round(total / t_sync / 1e6, 1),
                         "ring3_MS/s": round(total / t_ring / 1e6, 1), "ring3_GB/s_h2d": round(8 * total / t_ring / 1e9, 2),
                         "coalesced_2^20_MS/s": round(total / t_push / 1e6, 1) if t_push else None})
    # raw u8 I/Q records as the chain head (IQFileSource format stage on the device): 2 bytes per complex sample cross PCIe
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from iqfile_wbfm_mono import build_chain
    raw = rng.integers(0, 256, 2 * total, dtype=np.uint8)
    for chunk in (8192, 32768, 131072, 1 << 20):
        _src, ch, _rate = build_chain(bytes(16), "u8", 1102500.0, -250e3)
        ch.set_ring(3, 1 << 20)
        recs = [raw[2 * a:2 * (a + chunk)] for a in range(0, total, chunk)]
        ch.push(recs[0]); ch.flush()
        t0 = time.perf_counter()
        n = 0
        for c in recs:
            n += len(ch.push(c))
        n += len(ch.flush())
        dt = time.perf_counter() - t0
        rows.append({"chain": "u8 IQ records -> WBFM mono chain, coalesced into 2^20-sample batches", "chunk_samples": chunk,
                     "coalesced_2^20_MS/s": round(total / dt / 1e6, 1), "GB/s_h2d": round(2 * total / dt / 1e9, 2)})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
