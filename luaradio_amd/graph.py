"""DeviceGraph - CompositeBlock:connect() / run() for a DAG of device blocks with every edge a device vector.

Mirrors the part of radio/core/composite.lua a flow graph of hot-path blocks needs: connect() in its linear and
explicit-port forms (composite.lua:140-330), type differentiation and rate propagation in topological order
(:426-470 _prepare_to_run), and the synchronous one-process variant of run (:626-700 run(false): every block's
process() is called once per input chunk, upstream first).  What the reference moves through socketpairs
(radio/core/pipe.lua) stays in HBM here: a block's output vector is a device buffer that its consumers read in
place; a linear run of blocks becomes one lrhip_chain_t (kernel fusion), a two-input block waits for the shorter of
its inputs and keeps the excess of the other for the next call, like a pipe would.

    g = DeviceGraph()
    src1, src2 = g.input("a", types.ComplexFloat32, rate=1e6), g.input("b", types.ComplexFloat32, rate=1e6)
    mul = MultiplyConjugateBlock()
    g.connect(src1, "out", mul, "in1"); g.connect(src2, "out", mul, "in2")
    g.connect(mul, LowpassFilterBlock(16, 100e3), FrequencyDiscriminatorBlock(5))
    g.initialize()
    out = g.process(a=xa, b=xb)          # dict: name of every unconnected output -> numpy vector
"""
import ctypes as C

import numpy as np

from . import _lib
from .block import Block, Output
from .composites import Chain, CompositeBlock


class _DevBuf:
    """growable device vector (lrhip_malloc)"""

    def __init__(self):
        self.ptr, self.cap = None, 0

    def reserve(self, nbytes):
        if nbytes > self.cap:
            L = _lib.load()
            if self.ptr:
                L.lrhip_free(self.ptr)
            self.cap = max(int(nbytes), 256)
            self.ptr = _lib.check_ptr(L.lrhip_malloc(self.cap), "lrhip_malloc(%d)" % self.cap)
        return self.ptr

    def free(self):
        if self.ptr:
            _lib.load().lrhip_free(self.ptr)
            self.ptr, self.cap = None, 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GraphInput(Block):
    """placeholder for a source block: one output named "out" fed by the caller"""
    name = "GraphInput"

    def instantiate(self, label, data_type, rate):
        self.label, self.data_type = label, data_type
        self.rate = rate
        self.signature = ([], [Output("out", data_type)], None)


class _Node:
    def __init__(self, block):
        self.block = block
        self.inputs = {}            # input index -> (src node, output index)
        self.consumers = []         # (dst node, input index, output index)
        self.out = []               # per output: [_DevBuf, count]
        self.pending = {}           # input index -> [_DevBuf, count]   (multi-input nodes only)
        self.runner = None          # Chain for merged linear runs
        self.merged_into = None


class DeviceGraph:
    def __init__(self):
        self._nodes = {}
        self._order = None
        self._edges_named = []
        self._expanded = set()

    # ---- construction -----------------------------------------------------------------------------------
    def input(self, label, data_type, rate):
        blk = GraphInput(label, data_type, rate)
        self._node(blk)
        return blk

    def _node(self, blk):
        if id(blk) not in self._nodes:
            self._nodes[id(blk)] = _Node(blk)
        return self._nodes[id(blk)]

    def _ends(self, blk):
        """(block that takes the input, block that gives the output): a linear CompositeBlock (Decimator, Tuner, ...) is
        expanded into its blocks, connected in a row, the first time it is seen"""
        if not isinstance(blk, CompositeBlock):
            self._node(blk)
            return blk, blk
        inner = blk._blocks
        if not inner:
            raise ValueError("empty composite %s" % blk.name)
        if id(blk) not in self._expanded:
            self._expanded.add(id(blk))
            for a, b in zip(inner[:-1], inner[1:]):
                self._edges_named.append((a, None, b, None))
            for b in inner:
                self._node(b)
        return inner[0], inner[-1]

    @staticmethod
    def _port_index(ports, name, what, blk):
        for i, p in enumerate(ports):
            if p.name == name:
                return i
        raise KeyError("Block %s has no %s port named %r" % (blk.name, what, name))

    def connect(self, *args):
        """connect(b1, b2, ...): first output -> first input of the next (composite.lua:176-190), or
        connect(src, "out_name", dst, "in_name")"""
        if len(args) == 4 and isinstance(args[1], str) and isinstance(args[3], str):
            src, dst = self._ends(args[0])[1], self._ends(args[2])[0]
            self._edges_named.append((src, args[1], dst, args[3]))
        else:
            for a, b in zip(args[:-1], args[1:]):
                self._edges_named.append((self._ends(a)[1], None, self._ends(b)[0], None))
        self._order = None
        return self

    # ---- preparation ------------------------------------------------------------------------------------
    def initialize(self):
        L = _lib.load()
        nodes = self._nodes
        for nd in nodes.values():          # initialize() may be called again after further connect() calls: start from a clean slate
            nd.inputs, nd.consumers, nd.out, nd.pending, nd.runner, nd.merged_into = {}, [], [], {}, None, None
        # resolve port names: needs every signature's port list; use the first signature's names (they do not depend on type)
        for src, oname, dst, iname in self._edges_named:
            s, d = nodes[id(src)], nodes[id(dst)]
            outs = (src.signature or src.type_signatures[0])[1]
            ins = (dst.signature or dst.type_signatures[0])[0]
            oi = 0 if oname is None else self._port_index(outs, oname, "output", src)
            ii = 0 if iname is None else self._port_index(ins, iname, "input", dst)
            if ii in d.inputs:
                raise ValueError("input %d of block %s is already connected" % (ii, dst.name))
            d.inputs[ii] = (s, oi)
            s.consumers.append((d, ii, oi))
        # topological order (Kahn)
        indeg = {k: len(n.inputs) for k, n in nodes.items()}
        ready = [n for k, n in nodes.items() if indeg[k] == 0]
        order = []
        while ready:
            n = ready.pop(0)
            order.append(n)
            for d, _, _ in n.consumers:
                indeg[id(d.block)] -= 1
                if indeg[id(d.block)] == 0:
                    ready.append(d)
        if len(order) != len(nodes):
            raise ValueError("the graph has a cycle (feedback blocks are out of scope)")
        # types, rates, initialize (composite.lua:443-470)
        for n in order:
            b = n.block
            if isinstance(b, GraphInput):
                continue
            nin = len((b.signature or b.type_signatures[0])[0])
            if sorted(n.inputs) != list(range(nin)):
                raise ValueError("block %s has unconnected inputs" % b.name)
            srcs = [n.inputs[i] for i in range(nin)]
            b.differentiate([s.block.get_output_type(oi + 1) for s, oi in srcs])
            rates = [s.block.get_rate() for s, _ in srcs]
            if any(abs(r - rates[0]) > 1e-9 * abs(rates[0]) for r in rates):
                raise ValueError("block %s: input sample rates differ (%s)" % (b.name, rates))
            b.rate = rates[0]
            b.initialize()
        # merge maximal linear runs of single-input/single-output device blocks into chains (kernel fusion)
        def linear(n):
            b = n.block
            return (not isinstance(b, GraphInput) and len(n.inputs) == 1 and len(b.signature[1]) == 1 and b._stage
                    and not getattr(b, "_sub_blocks", None))
        for n in order:
            if not linear(n) or n.merged_into is not None:
                continue
            src, _ = n.inputs[0]
            if linear(src) and len(src.consumers) == 1:
                continue                    # not the head of its run
            run = [n]
            while len(run[-1].consumers) == 1 and linear(run[-1].consumers[0][0]):
                run.append(run[-1].consumers[0][0])
            if len(run) > 1:
                n.runner = Chain([r.block for r in run])
                n.run_tail = run[-1]
                for r in run[1:]:
                    r.merged_into = n
        for n in order:
            nout = len(n.block.signature[1])
            n.out = [[_DevBuf(), 0] for _ in range(nout)]
        self._order = order
        self._L = L
        return self

    # ---- execution --------------------------------------------------------------------------------------
    @staticmethod
    def _out_of(n, oi):
        """device (ptr, count) of output oi of node n (a merged run publishes through its tail)"""
        return n.out[oi][0].ptr, n.out[oi][1]

    def _gather(self, n):
        """aligned inputs of a multi-input node: (ptrs, count); keeps the excess of the longer inputs pending"""
        L = self._L
        srcs = [n.inputs[i] for i in range(len(n.inputs))]
        sizes = [s.block.get_output_type(oi + 1).size for s, oi in srcs]
        new = [self._out_of(s, oi) for s, oi in srcs]
        pend = [n.pending.setdefault(i, [_DevBuf(), 0]) for i in range(len(srcs))]
        avail = [pend[i][1] + new[i][1] for i in range(len(srcs))]
        take = min(avail)
        ptrs = []
        for i in range(len(srcs)):
            buf, pc = pend[i]
            if pc == 0 and avail[i] == take:
                ptrs.append(new[i][0])                    # common case: read the producer's vector in place
                continue
            # [pending | new] into a fresh staging vector, the first `take` are consumed, the rest stays pending
            stage = _DevBuf()
            stage.reserve(max(avail[i], 1) * sizes[i])
            if pc:
                _lib.check(L.lrhip_memcpy_d2d(stage.ptr, buf.ptr, pc * sizes[i]), "d2d")
            if new[i][1]:
                _lib.check(L.lrhip_memcpy_d2d(stage.ptr + pc * sizes[i], new[i][0], new[i][1] * sizes[i]), "d2d")
            left = avail[i] - take
            keep = _DevBuf()
            if left:
                keep.reserve(left * sizes[i])
                _lib.check(L.lrhip_memcpy_d2d(keep.ptr, stage.ptr + take * sizes[i], left * sizes[i]), "d2d")
            n.pending[i] = [keep, left]
            n._stages = getattr(n, "_stages", [])
            n._stages.append(stage)                       # alive until the stream has run (freed at the end of process())
            n._stages.append(buf)
            ptrs.append(stage.ptr)
        return ptrs, take

    def process(self, **inputs):
        if self._order is None:
            raise RuntimeError("call initialize() first")
        L = self._L
        results = {}
        for n in self._order:
            b = n.block
            if n.merged_into is not None:
                continue
            if isinstance(b, GraphInput):
                x = np.asarray(inputs[b.label])
                if x.dtype != b.data_type.dtype:          # as Block._execute: no silent casts (complex128 -> complex64, complex -> real part)
                    raise TypeError("graph input '%s' expects %s, got %s" % (b.label, b.data_type, x.dtype))
                x = np.ascontiguousarray(x)
                n.out[0][0].reserve(max(x.nbytes, 1))
                if len(x):
                    _lib.check(L.lrhip_memcpy_h2d(n.out[0][0].ptr, x.ctypes.data_as(C.c_void_p), x.nbytes), "h2d")
                n.out[0][1] = len(x)
                continue
            if len(n.inputs) == 1:
                s, oi = n.inputs[0]
                ptrs, count = [self._out_of(s, oi)[0]], self._out_of(s, oi)[1]
            else:
                ptrs, count = self._gather(n)
            subs = getattr(b, "_sub_blocks", None)
            if subs:                                     # one input, several outputs (ComplexToFloatBlock)
                for k, sb in enumerate(subs):
                    cap = sb.max_output(count)
                    n.out[k][0].reserve(max(cap, 1) * sb.get_output_type().size)
                    n.out[k][1] = sb.process_device(ptrs[0], count, n.out[k][0].ptr, cap) if count else 0
                tail = n
            elif n.runner is not None:
                tail = n.run_tail
                cap = n.runner.max_output(count)
                tail.out[0][0].reserve(max(cap, 1) * tail.block.get_output_type().size)
                tail.out[0][1] = n.runner.process_device(ptrs[0], count, tail.out[0][0].ptr, cap) if count else 0
            elif len(ptrs) == 2:
                tail = n
                n.out[0][0].reserve(max(count, 1) * b.get_output_type().size)
                got = L.lrhip_stage_execute2_device(b.stage_handle(), ptrs[0], ptrs[1], count, n.out[0][0].ptr, count) if count else 0
                n.out[0][1] = _lib.check(got, "%s:process" % b.name)
            else:
                tail = n
                cap = b.max_output(count)
                n.out[0][0].reserve(max(cap, 1) * b.get_output_type().size)
                n.out[0][1] = b.process_device(ptrs[0], count, n.out[0][0].ptr, cap) if count else 0
            for k, port in enumerate(tail.block.signature[1]):
                if not any(oi == k for _, _, oi in tail.consumers):
                    cnt = tail.out[k][1]
                    y = np.empty(cnt, dtype=tail.block.get_output_type(k + 1).dtype)
                    if cnt:
                        _lib.check(L.lrhip_memcpy_d2h(y.ctypes.data_as(C.c_void_p), tail.out[k][0].ptr, y.nbytes), "d2h")
                    key = tail.block.name if len(tail.block.signature[1]) == 1 else "%s.%s" % (tail.block.name, port.name)
                    while key in results:
                        key += "'"
                    results[key] = y
        _lib.check(L.lrhip_synchronize(), "synchronize")
        for n in self._order:                            # staging vectors of this call are no longer in flight
            for st in getattr(n, "_stages", []):
                st.free()
            n._stages = []
        return results
