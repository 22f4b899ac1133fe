"""The reference's PROCESS MODEL run against the real library - started by tests/test_fork_gpu.py as a FRESH interpreter, because the
pytest process has long initialised the device and fork() after that is exactly what must not be done.

What LuaRadio does (radio/core/composite.lua:534-642): the library is loaded in the parent (`ffi.load` at module load), every block's initialize() runs in
the parent, then ONE fork() per block; the child closes EVERY descriptor that is not one of its pipes / files / its control socket (the /proc/self/fd loop of
:594-611 - stdin, stdout and stderr included), runs the block and exits.  This script does the same with ctypes in place of the FFI:

    parent: ctypes.CDLL(liblrhip.so)  - no HIP call - pipes / socket pairs created - fork() per block - close its copies - read results - waitpid
    child : close all other descriptors - lrhip_init (first device call of the process) - run - write the result to its pipe - _exit

    fork_model.py <shape> <out.npz>      shape: stage | chain | fanout | init_then_fork | lua_partition
"""
import os
import socket
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FS = 1102500.0
N = 600000
OFFSETS = [-250e3, 120e3, 300e3]


def stream():
    rng = np.random.default_rng(21)
    return (rng.uniform(-1, 1, N) + 1j * rng.uniform(-1, 1, N)).astype(np.complex64)


def close_all_but(keep):
    """radio/core/composite.lua:594-611"""
    for entry in os.listdir("/proc/self/fd"):
        fd = int(entry)
        if fd not in keep:
            try:
                os.close(fd)
            except OSError:
                pass


def send(fd, payload):
    os.write(fd, struct.pack("<q", len(payload)))
    view, sent = memoryview(payload), 0
    while sent < len(payload):
        sent += os.write(fd, view[sent:sent + (1 << 20)])


def recv_all(fd):
    parts = []
    while True:
        b = os.read(fd, 1 << 20)
        if not b:
            break
        parts.append(b)
    raw = b"".join(parts)
    if len(raw) < 8:
        return None
    n = struct.unpack("<q", raw[:8])[0]
    return raw[8:8 + n]


def child_stage(wfd):
    """one stand-alone device block: LowpassFilterBlock(128, 100e3) on the stream, 8 192-sample vectors (lrhip_stage_execute, as lrhip.execute does)"""
    import luaradio_amd as lr
    from luaradio_amd import types
    lr.init(0)
    blk = lr.LowpassFilterBlock(128, 100e3)
    blk.use_fft = 0
    blk.rate = FS
    blk.differentiate([types.ComplexFloat32])
    blk.initialize()
    x = stream()
    send(wfd, np.concatenate([blk.process(x[a:a + 8192]) for a in range(0, N, 8192)]).tobytes())


def child_chain(wfd):
    """DeviceChainBlock's call sequence (lua/radio/composites/devicechain.lua): stages, lrhip_chain_create_ex, set_ring, push per vector, flush"""
    import ctypes as C
    import luaradio_amd as lr
    from luaradio_amd import _lib
    lr.init(0)
    L = _lib.load()
    taps = np.asarray(lr.filter_utils.firwin_lowpass(128, 50e3 / (FS / 2)), np.float32)      # Tuner(-250e3, 100e3, 5): bandwidth / 2 (tuner.lua:41)
    st = [L.lrhip_rotator_create(2 * np.pi * -250e3 / FS), L.lrhip_fir_create(taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps), 0, 1, 1, 0),
          L.lrhip_downsampler_create(5, 8)]
    arr = (C.c_void_p * 3)(*st)
    ch = _lib.check_ptr(L.lrhip_chain_create_ex(arr, 3, 0), "chain")
    _lib.check(L.lrhip_chain_set_ring(ch, 3, 1 << 17), "ring")
    x = stream()
    parts = []
    for a in range(0, N, 8192):
        v = x[a:a + 8192]
        cap = L.lrhip_chain_push_bound(ch, len(v))
        out = np.empty(cap, np.complex64)
        n = _lib.check(L.lrhip_chain_push(ch, v.ctypes.data_as(C.c_void_p), len(v), out.ctypes.data_as(C.c_void_p), cap), "push")
        parts.append(out[:n])
    cap = L.lrhip_chain_push_bound(ch, 0)
    out = np.empty(cap, np.complex64)
    n = _lib.check(L.lrhip_chain_flush(ch, out.ctypes.data_as(C.c_void_p), cap), "flush")
    parts.append(out[:n])
    send(wfd, np.concatenate(parts).tobytes())


def child_branch(wfd, index, sock_fd):
    import luaradio_amd as lr
    from luaradio_amd import procfanout, types
    lr.init(procfanout.placement(index))
    blk = lr.TunerBlock(OFFSETS[index], 100e3, 5)
    blk.rate = FS
    blk.differentiate([types.ComplexFloat32])
    blk.initialize()
    branch = procfanout.Branch(blk, index, socket.socket(fileno=sock_fd), 1 << 16, 8)
    parts = []
    while True:
        y = branch.process()
        if y is None:
            break
        parts.append(y)
    branch.cleanup()
    send(wfd, np.concatenate(parts).tobytes())


def child_head(wfd, sock_fds):
    import luaradio_amd as lr
    from luaradio_amd import procfanout
    lr.init(0)
    head = procfanout.Head(None, [socket.socket(fileno=fd) for fd in sock_fds], np.complex64, np.complex64, slab_capacity=1 << 16)
    x = stream()
    for a in range(0, N, 8192):
        head.process(x[a:a + 8192])
    head.cleanup()
    send(wfd, struct.pack("<qq", head.k, head.peer_copies))


def child_init_then_fork(wfd):
    from luaradio_amd import _lib
    L = _lib.load()
    rc = L.lrhip_init(0)
    count = L.lrhip_device_count()
    msg = L.lrhip_strerror().decode()
    send(wfd, ("%d|%d|%s" % (rc, count, msg)).encode())


LUA_PARTITION = r"""
local R = require('reference_standins')
local types = require('radio.types')
local taps = ...
local g = R.graph()
local src, t, f, d, k = R.HostSource(1102500), R.FrequencyTranslatorBlock(-250e3), R.FIRFilterBlock(taps), R.DownsamplerBlock(5), R.HostSink()
src:differentiate({})
for _, b in ipairs({t, f, d, k}) do b:differentiate({types.ComplexFloat32}) end
g.connect(src, t, f, d, k)
local connections, device_blocks = R.prepare(g.connections, {src, t, f, d, k})
return device_blocks[1]
"""
PARTITION_FIRST = 300000


def lua_partition_parent(L):
    """the partition helpers of lua/radio/composites/devicechain.lua asked in the PARENT (the Lua glue itself, executed by tests/helpers/minilua.py
    on the real library): each answer comes from a fork()ed helper process (lrhip.in_helper), the parent never owns a device, and the block process forked
    afterwards builds its own chain armed with the recorded start_at()"""
    import luaradio_amd as lr
    from tests.helpers import lua_mocks as LM
    from tests.helpers import minilua as ml
    I, proxy, ffi = LM.make_interpreter(L)
    I.globals.set("__copy_vector", lambda v: LM.Vector(v.data_type, 0, v.array().copy()))
    taps = np.asarray(lr.filter_utils.firwin_lowpass(128, 50e3 / (FS / 2)), np.float32)
    chain = I.run(LUA_PARTITION, "partition", [LM.Vector(LM.DataType("Float32", np.float32), 0, taps.copy())])[0]
    halo = ml.call(ml.index(chain, "halo"), [chain])[0]
    align = ml.call(ml.index(chain, "shard_align"), [chain])[0]
    seek = ml.call(ml.index(chain, "start_at"), [chain, float(PARTITION_FIRST)])[0]
    parent_device = L.lrhip_device()                # -1: the helpers have come and gone (one for halo + alignment, one for start_at), this process still has no device context

    def child(wfd):
        x = stream()[int(seek):]
        cf = LM.DataType("ComplexFloat32", np.complex64)
        parts = []
        for a in range(0, len(x), 8192):
            y = ml.call(ml.index(chain, "process"), [chain, LM.Vector(cf, 0, x[a:a + 8192].copy())])[0]
            parts.append(y.array().copy())
        ml.call(ml.index(chain, "cleanup"), [chain])
        pipes = ml.index(ml.index(chain, "outputs").get(1), "pipes")
        written = ml.index(pipes.get(1), "written")
        for i in range(1, written.length() + 1):
            parts.append(written.get(i).array().copy())
        send(wfd, np.concatenate(parts).tobytes())

    results, codes = run_children([("block", child, ())])
    return results, codes, np.array([halo, align, seek, parent_device], np.float64)


def run_children(jobs, parent_closes=()):
    """jobs: [(name, function(wfd), descriptors the child keeps)]: one fork() per job, as CompositeBlock:start forks one process per block"""
    readers, pids = {}, {}
    for name, fn, keep in jobs:
        rfd, wfd = os.pipe()
        pid = os.fork()
        if pid == 0:
            status = 1
            try:
                close_all_but(set(keep) | {wfd})
                fn(wfd)
                status = 0
            except BaseException as e:              # noqa: BLE001 - the child reports and exits, it must never return into the parent's code
                try:
                    send(wfd, ("ERROR %s: %s" % (type(e).__name__, e)).encode())
                except OSError:
                    pass
            finally:
                os._exit(status)
        os.close(wfd)
        readers[name], pids[name] = rfd, pid
    for fd in parent_closes:                        # composite.lua:638-642 and DeviceFanoutBlock:close_parent_fds
        os.close(fd)
    results, codes = {}, {}
    for name, rfd in readers.items():
        results[name] = recv_all(rfd)
        os.close(rfd)
    for name, pid in pids.items():
        _, st = os.waitpid(pid, 0)
        codes[name] = os.WEXITSTATUS(st) if os.WIFEXITED(st) else -os.WTERMSIG(st)
    return results, codes


def main():
    shape, out_path = sys.argv[1], sys.argv[2]
    from luaradio_amd import _lib
    L = _lib.load()                                 # the parent loads the library (ffi.load at module load) - and makes NO device call
    out = {}
    if shape == "stage":
        results, codes = run_children([("stage", child_stage, ())])
        out["y"] = np.frombuffer(results["stage"] or b"", np.complex64)
    elif shape == "chain":
        results, codes = run_children([("chain", child_chain, ())])
        out["y"] = np.frombuffer(results["chain"] or b"", np.complex64)
    elif shape == "fanout":
        pairs = [socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM) for _ in OFFSETS]        # DeviceFanoutBlock:initialize(), pre-fork
        fds = [(a.detach(), b.detach()) for a, b in pairs]
        jobs = [("branch%d" % k, (lambda w, k=k: child_branch(w, k, fds[k][1])), (fds[k][1],)) for k in range(len(OFFSETS))]
        jobs.append(("head", lambda w: child_head(w, [f[0] for f in fds]), tuple(f[0] for f in fds)))
        results, codes = run_children(jobs, parent_closes=[fd for pair in fds for fd in pair])
        for k in range(len(OFFSETS)):
            r = results["branch%d" % k] or b""
            out["y%d" % k] = np.frombuffer(r, np.complex64) if not r.startswith(b"ERROR") else np.zeros(0, np.complex64)
        out["head"] = np.frombuffer(results["head"] or b"", np.int64) if results["head"] and not results["head"].startswith(b"ERROR") else np.zeros(0, np.int64)
    elif shape == "lua_partition":
        results, codes, answers = lua_partition_parent(L)
        r = results["block"] or b""
        out["y"] = np.frombuffer(r, np.complex64) if not r.startswith(b"ERROR") else np.zeros(0, np.complex64)
        out["answers"] = answers
    elif shape == "init_then_fork":
        assert L.lrhip_init(0) == 0                 # the mistake: a device call in the parent
        results, codes = run_children([("child", child_init_then_fork, ())])
    else:
        raise SystemExit("unknown shape " + shape)
    out["codes"] = np.array([codes[k] for k in sorted(codes)], np.int64)
    out["messages"] = np.array([(results[k] or b"")[:400].decode(errors="replace") if (results[k] or b"").startswith(b"ERROR") or shape == "init_then_fork" else ""
                                for k in sorted(results)])
    np.savez(out_path, **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
