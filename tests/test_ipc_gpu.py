"""Fan-out below the host language (include/lrhip.h lrhip_ipc_* / lrhip_peer_copy): a producer process pushes slabs into a consumer
process's device buffers (double-buffered, interprocess events both ways) and the consumer runs its Tuner branch on them - two processes
on ONE device here (gpurun exposes one GPU); with one device per process the same calls move the slabs over xGMI."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSLAB, N = 5, 1 << 16


def _slab(k):
    rng = np.random.default_rng(100 + k)
    return (rng.uniform(-1, 1, N) + 1j * rng.uniform(-1, 1, N)).astype(np.complex64)


def _consumer(conn):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import luaradio_amd as lr
    from luaradio_amd import _lib, ipc, types
    lr.init(0)
    L = _lib.load()
    bufs = [_lib.check_ptr(L.lrhip_malloc(8 * N), "malloc") for _ in range(2)]
    out_dev = _lib.check_ptr(L.lrhip_malloc(8 * N), "malloc")
    filled = [ipc.Event.create() for _ in range(2)]          # recorded by the producer after its copy
    consumed = [ipc.Event.create() for _ in range(2)]        # recorded here after the branch has read the slab
    conn.send({"mem": [ipc.export_memory(b) for b in bufs], "filled": [e.handle for e in filled], "consumed": [e.handle for e in consumed]})
    tun = lr.TunerBlock(-250e3, 100e3, 5)
    tun.rate = 1102500.0
    tun.differentiate([types.ComplexFloat32])
    tun.initialize()
    outs = []
    for k in range(NSLAB):
        assert conn.recv() == ("pushed", k)                  # control message only: the data went device to device
        filled[k & 1].wait()                                 # the library stream waits on the GPU for the producer's copy
        got = tun.process_device(bufs[k & 1], N, out_dev, N)
        consumed[k & 1].record()
        conn.send(("consumed", k))
        host = np.empty(got, np.complex64)
        _lib.check(L.lrhip_memcpy_d2h(host.ctypes.data_as(C.c_void_p), out_dev, 8 * got), "d2h")
        outs.append(host)
    conn.send(np.concatenate(outs))
    conn.recv()
    conn.close()


def _producer(conn):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import luaradio_amd as lr
    from luaradio_amd import _lib, ipc
    lr.init(0)
    L = _lib.load()
    hs = conn.recv()
    bufs = [ipc.open_memory(h) for h in hs["mem"]]
    filled = [ipc.Event.open(h) for h in hs["filled"]]
    consumed = [ipc.Event.open(h) for h in hs["consumed"]]
    src = _lib.check_ptr(L.lrhip_malloc(8 * N * NSLAB), "malloc")
    for k in range(NSLAB):
        s = _slab(k)
        _lib.check(L.lrhip_memcpy_h2d(src + 8 * N * k, s.ctypes.data_as(C.c_void_p), 8 * N), "h2d")
    _lib.check(L.lrhip_synchronize(), "sync")
    for k in range(NSLAB):
        if k >= 2:
            assert conn.recv() == ("consumed", k - 2)
            consumed[k & 1].wait(on_copy_stream=True)        # do not overwrite a slab the branch is still reading
        ipc.peer_copy(bufs[k & 1], 0, src + 8 * N * k, 0, 8 * N)
        filled[k & 1].record(on_copy_stream=True)
        conn.send(("pushed", k))
    ipc.copy_stream_synchronize()
    for k in range(max(0, NSLAB - 2), NSLAB):
        assert conn.recv() == ("consumed", k)
    conn.send("done")
    for b in bufs:
        ipc.close_memory(b)
    conn.close()


def test_two_processes_one_device_double_buffered_slabs():
    ctx = mp.get_context("spawn")
    # parent relays control messages between the two children (it stands in for the host's control socket) and never touches the device
    pc, cc = ctx.Pipe()
    pp, cp = ctx.Pipe()
    cons = ctx.Process(target=_consumer, args=(cc,))
    prod = ctx.Process(target=_producer, args=(cp,))
    cons.start()
    prod.start()
    try:
        pp.send(pc.recv())                                   # handles: consumer -> producer
        result = None
        done = 0
        while done < 2:
            for src, dst in ((pp, pc), (pc, pp)):
                if src.poll(0.01):
                    m = src.recv()
                    if isinstance(m, np.ndarray):
                        result = m
                        done += 1
                        pc.send("bye")
                    elif m == "done":
                        done += 1
                    else:
                        dst.send(m)
            assert cons.is_alive() or prod.is_alive() or done == 2
    finally:
        cons.join(60)
        prod.join(60)
    assert cons.exitcode == 0 and prod.exitcode == 0
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    want = O.tuner(-250e3, 100e3, 5, 1102500.0, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(np.concatenate([_slab(k) for k in range(NSLAB)]))
    assert result is not None and len(result) == len(want)
    assert float(np.max(np.abs(result - want))) < 2e-6


# ---------------------------------------------------------------------------------------------------------------------------------
# lua/radio/composites/devicefanout.lua replayed: ONE head process and THREE branch processes, each branch bound
# to device `b % lrhip_device_count()` as lrhip.ensure(index) does in the Lua glue.  luaradio_amd/procfanout.py is the call-for-call
# twin of that file (same wire structs, same order of lrhip_* calls); the parent only creates the socket pairs - what
# DeviceFanoutBlock:initialize() does before CompositeBlock forks - and collects the branch outputs.
# ---------------------------------------------------------------------------------------------------------------------------------
FO_N, FO_SLAB, FO_CHUNK = 200000, 1 << 15, 8192            # 200 000 samples in 8 192-sample process() vectors, slabs of 32 768: 7 slabs, the last partial
FO_OFFSETS = [-350e3, -250e3, 150e3]


def _fo_input():
    rng = np.random.default_rng(77)
    return (rng.uniform(-1, 1, FO_N) + 1j * rng.uniform(-1, 1, FO_N)).astype(np.complex64)


def _fo_branch(index, sock, result, with_head_chain):
    import faulthandler
    faulthandler.dump_traceback_later(45, exit=True)        # a stuck process says where and leaves (it must not outlive the test holding the GPU)
    sys.path.insert(0, ROOT)
    import luaradio_amd as lr
    from luaradio_amd import _lib, procfanout, types
    device = procfanout.placement(index)
    lr.init(device)
    assert _lib.load().lrhip_device() == device
    if with_head_chain:
        blk = lr.DecimatorBlock(5)                          # the head chain already translated: the branch filters and decimates
    else:
        blk = lr.TunerBlock(FO_OFFSETS[index], 100e3, 5)
    blk.rate = 1102500.0
    blk.differentiate([types.ComplexFloat32])
    blk.initialize()
    br = procfanout.Branch(blk, index, sock, FO_SLAB, 8)
    outs = []
    while True:
        out = br.process()
        if out is None:
            break
        outs.append(out)
    br.cleanup()
    result.send((index, device, np.concatenate(outs) if outs else np.empty(0, np.complex64)))
    result.close()


def _fo_head(socks, result, with_head_chain, latency):
    import faulthandler
    faulthandler.dump_traceback_later(45, exit=True)
    sys.path.insert(0, ROOT)
    import time
    import luaradio_amd as lr
    from luaradio_amd import procfanout, types
    lr.init(procfanout.placement(0))
    chain = None
    if with_head_chain:
        chain = lr.FrequencyTranslatorBlock(-250e3)
        chain.rate = 1102500.0
        chain.differentiate([types.ComplexFloat32])
        chain.initialize()
    head = procfanout.Head(chain, socks, np.complex64, np.complex64, slab_capacity=FO_SLAB, max_latency=latency)
    x = _fo_input()
    polled = 0
    for a in range(0, FO_N, FO_CHUNK):
        head.process(x[a:a + FO_CHUNK])
        if latency and a == 5 * FO_CHUNK:
            # the source stalls: the run loop's bounded wait times out and poll() hands the partial slab on (DeviceChainBlock.timed_run)
            due = head.poll_due()
            assert 0.0 <= due <= latency
            time.sleep(due + 0.002)
            before = head.k
            head.poll()
            polled += head.k - before
    head.cleanup()
    result.send(("head", head.k, head.peer_copies, polled))
    result.close()


@pytest.mark.parametrize("with_head_chain,latency", [(False, 0.0), (True, 0.0), (False, 0.02)])
def test_one_head_three_branch_processes(with_head_chain, latency):
    import socket
    ctx = mp.get_context("spawn")
    pairs = [socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM) for _ in FO_OFFSETS]
    # one result pipe per process (several writers on one Connection interleave their messages)
    res = [ctx.Pipe(duplex=False) for _ in range(len(FO_OFFSETS) + 1)]
    procs = [ctx.Process(target=_fo_branch, args=(b, pairs[b][1], res[b][1], with_head_chain)) for b in range(len(FO_OFFSETS))]
    procs.append(ctx.Process(target=_fo_head, args=([p[0] for p in pairs], res[-1][1], with_head_chain, latency)))
    for p in procs:
        p.start()
    for a, b in pairs:
        a.close()
        b.close()
    for _, w in res:
        w.close()
    got = {}
    try:
        for r, _ in res:
            assert r.poll(60), "a fan-out process did not report (alive: %s)" % [p.is_alive() for p in procs]
            m = r.recv()
            got[m[0]] = m[1:]
    finally:
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.kill()                                     # the exact processes this test started
                p.join(10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    import torch
    count = torch.cuda.device_count()
    slabs, copies, polled = got["head"]
    assert slabs >= 7 and copies == 3 * slabs                                  # every slab went to every branch, device to device
    assert (polled >= 1) == bool(latency)
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    x = _fo_input()
    for b, off in enumerate(FO_OFFSETS):
        device, y = got[b]
        assert device == b % count                                             # branch -> GPU placement (one per GPU on an 8-GPU node)
        if with_head_chain:
            want = O.Chain([O.Rotator(2 * np.pi * -250e3 / 1102500.0, O.MODE_F64)] + O.decimator(5, 1102500.0, True, mode=O.MODE_FMA).stages).process(x)
        else:
            want = O.tuner(off, 100e3, 5, 1102500.0, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(x)
        assert len(y) == len(want) == FO_N // 5
        assert float(np.max(np.abs(y - want))) < 2e-6, (b, float(np.max(np.abs(y - want))))
