"""Fan-out across PROCESSES, one per GPU: the Python twin of lua/radio/composites/devicefanout.lua, call for call.

The reference's fan-out is one OutputPort feeding several readers, every reader its own forked process
(radio/core/block.lua:119-166, radio/core/pipe.lua:617-627, radio/core/composite.lua:568-569).  `Head` is DeviceFanoutBlock (the
source's one reader: uploads once, runs the optional head chain, pushes every slab to every branch's device with lrhip_peer_copy);
`Branch` is DeviceBranchBlock (binds its process to device `index % lrhip_device_count()`, exports two slab buffers and four
interprocess events, runs its chain on each slab where it lands).  Head and branch talk over a UNIX socket pair: one `hello`
(IPC handles) and a 16-byte token per slab - the wire structs below are the Lua file's ffi.cdef, byte for byte.

luaradio_amd/fanout.py is the torch.distributed / RCCL form of the same pattern (one process group, collective broadcast); this is the
form a LuaJIT host can use: no process group, no collective, point-to-point copies over xGMI, which is what a broadcast from one root
decomposes into on a fully connected node anyway.
"""
import ctypes as C
import struct
import time

import numpy as np

from . import _lib, ipc

HELLO = struct.Struct("<iiQ64s64s64s64s64s64s")      # lrhip_fanout_hello_t: device, reserved, capacity, mem[2], filled[2], consumed[2]
TOKEN = struct.Struct("<qq")                         # lrhip_fanout_token_t: k, n (n < 0: end of stream)


def _recv_exact(sock, size):
    buf = bytearray()
    while len(buf) < size:
        part = sock.recv(size - len(buf))
        if not part:
            return None
        buf += part
    return bytes(buf)


def placement(index, count=None):
    """lrhip.ensure(index) of the Lua glue: placement indices wrap over the devices of the box"""
    if count is None:
        count = _lib.check(_lib.load().lrhip_device_count(), "device_count")
    return index % count


class Branch:
    """DeviceBranchBlock: `block` is an initialized luaradio_amd block / composite / Chain (process_device, max_output, get_output_type)."""

    def __init__(self, block, index, sock, capacity, in_size):
        self.block, self.index, self.sock, self.capacity, self.in_size = block, index, sock, int(capacity), int(in_size)
        self.started = False

    def start(self):                                 # branch_start()
        L = _lib.load()
        self.device = _lib.check(L.lrhip_device(), "device")          # the caller's lr.init(placement(index)) bound this process
        self.out_dtype = self.block.get_output_type().dtype
        self.out_size = self.out_dtype.itemsize
        self.out_cap = self.block.max_output(self.capacity) + 64
        self.d_out = _lib.check_ptr(L.lrhip_malloc(self.out_cap * self.out_size), "malloc")
        self.slab = [_lib.check_ptr(L.lrhip_malloc(self.capacity * self.in_size), "malloc") for _ in range(2)]
        mem = [ipc.export_memory(p) for p in self.slab]
        self.filled = [ipc.Event.create() for _ in range(2)]
        self.consumed = [ipc.Event.create() for _ in range(2)]
        self.sock.sendall(HELLO.pack(self.device, 0, self.capacity, mem[0], mem[1], self.filled[0].handle, self.filled[1].handle,
                                     self.consumed[0].handle, self.consumed[1].handle))
        self.started = True

    def process(self):
        """blocks until the head announces a slab; returns the branch output (numpy) or None at the end of the stream"""
        if not self.started:
            self.start()
        L = _lib.load()
        raw = _recv_exact(self.sock, TOKEN.size)
        if raw is None:
            return None
        k, n = TOKEN.unpack(raw)
        if n < 0:
            return None
        i = k % 2
        self.filled[i].wait()                        # the library stream waits ON THE GPU for the head's copy
        m = self.block.process_device(self.slab[i], n, self.d_out, self.out_cap)
        self.consumed[i].record()
        self.sock.sendall(TOKEN.pack(k, n))          # ack{k}
        out = np.empty(m, self.out_dtype)
        if m:
            _lib.check(L.lrhip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.d_out, m * self.out_size), "d2h")
        return out

    def cleanup(self):
        self.sock.close()


class Head:
    """DeviceFanoutBlock: `chain` is None (upload and fan out) or an initialized device block / Chain that runs in front of the fan-out."""

    def __init__(self, chain, socks, in_dtype, slab_dtype, slab_capacity=1 << 20, max_latency=0.0, device_index=0):
        self.chain, self.socks = chain, list(socks)
        self.in_dtype, self.slab_dtype = np.dtype(in_dtype), np.dtype(slab_dtype)
        self.slab_capacity, self.max_latency, self.device_index = int(slab_capacity), float(max_latency), device_index
        self.started = False
        self.peer_copies = 0

    def start(self):                                 # head_start()
        L = _lib.load()
        self.my_device = _lib.check(L.lrhip_device(), "device")
        self.in_size, self.slab_size = self.in_dtype.itemsize, self.slab_dtype.itemsize
        self.batch = self.slab_capacity
        if self.chain is not None:
            while self.batch > 1 and self.chain.max_output(self.batch) > self.slab_capacity:
                self.batch //= 2
        self.staging = _lib.check_ptr(L.lrhip_host_alloc(self.batch * self.in_size), "host_alloc")
        self.staging_view = np.frombuffer((C.c_uint8 * (self.batch * self.in_size)).from_address(self.staging), dtype=np.uint8)
        self.d_in = _lib.check_ptr(L.lrhip_malloc(self.batch * self.in_size), "malloc") if self.chain is not None else None
        self.my_slab = [_lib.check_ptr(L.lrhip_malloc(self.slab_capacity * self.slab_size), "malloc") for _ in range(2)]
        self.ready = [ipc.Event.create() for _ in range(2)]
        self.sent = [ipc.Event.create() for _ in range(2)]
        self.peer = []
        for b, sock in enumerate(self.socks):
            raw = _recv_exact(sock, HELLO.size)
            if raw is None:
                raise RuntimeError("fan-out branch %d closed its socket before the handshake" % b)
            device, _, capacity, m0, m1, f0, f1, c0, c1 = HELLO.unpack(raw)
            assert capacity >= self.slab_capacity, "fan-out branch slab smaller than the head's"
            self.peer.append({"device": device, "slab": [ipc.open_memory(m0), ipc.open_memory(m1)],
                              "filled": [ipc.Event.open(f0), ipc.Event.open(f1)], "consumed": [ipc.Event.open(c0), ipc.Event.open(c1)]})
        self.fill, self.k, self.fill_t0 = 0, 0, 0.0
        self.started = True

    def launch(self):                                # head_launch(): slab k = upload, head chain, one peer copy per branch
        L = _lib.load()
        n, k = self.fill, self.k
        i = k % 2
        self.fill = 0
        if n == 0:
            return
        if k >= 2:
            self.sent[i].synchronize()               # my_slab[i] was last read by the copies of slab k - 2
        m = n
        if self.chain is not None:
            _lib.check(L.lrhip_memcpy_h2d(self.d_in, self.staging, n * self.in_size), "h2d")
            m = self.chain.process_device(self.d_in, n, self.my_slab[i], self.slab_capacity)
        else:
            _lib.check(L.lrhip_memcpy_h2d(self.my_slab[i], self.staging, n * self.in_size), "h2d")
        self.ready[i].record()
        self.ready[i].wait(on_copy_stream=True)      # copy stream: after the head chain's kernels
        for b, peer in enumerate(self.peer):
            if k >= 2:
                if _recv_exact(self.socks[b], TOKEN.size) is None:
                    raise RuntimeError("fan-out branch %d terminated unexpectedly" % b)
                peer["consumed"][i].wait(on_copy_stream=True)
            if m > 0:
                ipc.peer_copy(peer["slab"][i], peer["device"], self.my_slab[i], self.my_device, m * self.slab_size)
                self.peer_copies += 1
            peer["filled"][i].record(on_copy_stream=True)
        self.sent[i].record(on_copy_stream=True)
        tok = TOKEN.pack(k, m)
        for sock in self.socks:
            sock.sendall(tok)
        self.k = k + 1

    def process(self, x):
        """a sink's process(): accumulate into the pinned staging buffer, launch whole slabs"""
        if not self.started:
            self.start()
        x = np.ascontiguousarray(x)
        if x.dtype != self.in_dtype:
            raise TypeError("fan-out head expects %s input, got %s" % (self.in_dtype, x.dtype))
        src, left, pos = x.view(np.uint8).reshape(-1), len(x), 0
        while left > 0:
            take = min(left, self.batch - self.fill)
            if self.fill == 0:
                self.fill_t0 = time.monotonic()
            a = self.fill * self.in_size
            self.staging_view[a:a + take * self.in_size] = src[pos * self.in_size:(pos + take) * self.in_size]
            self.fill += take
            pos += take
            left -= take
            if self.fill == self.batch:
                self.launch()
        if self.max_latency > 0 and self.fill > 0 and time.monotonic() - self.fill_t0 >= self.max_latency:
            self.launch()

    def poll_due(self):
        if not self.started or self.fill == 0 or not self.max_latency > 0:
            return -1.0
        return max(0.0, self.fill_t0 + self.max_latency - time.monotonic())

    def poll(self):
        if self.started and self.fill > 0:
            self.launch()

    def cleanup(self):
        """EOF upstream: the partial slab, the last acks, then the end-of-stream token"""
        if not self.started:
            if not self.socks:
                return
            self.start()
        self.launch()
        for _ in range(min(2, self.k)):
            for sock in self.socks:
                _recv_exact(sock, TOKEN.size)
        ipc.copy_stream_synchronize()
        tok = TOKEN.pack(self.k, -1)
        for sock in self.socks:
            sock.sendall(tok)
            sock.close()
