"""What lua/radio/** needs around it to RUN under tests/helpers/minilua.py - TEST INFRASTRUCTURE.

  * `ffi`: a small LuaJIT-FFI look-alike on ctypes - cdef (the two wire structs of devicefanout.lua are recognised by name, prototypes are
    taken from luaradio_amd/_lib.py's SIGNATURES), new / cast / gc / string / sizeof / copy / errno, ffi.C (getpid, socketpair, read, write,
    close, poll, strerror) and ffi.load("lrhip") -> a library proxy that either forwards every lrhip_* call to the REAL liblrhip.so through
    ctypes (GPU box) or to a recording fake (CPU box);
  * the reference's core modules the glue requires (radio.core.block / pipe / platform, data types, Vector) as minimal stand-ins with the
    same API surface - written from the reference's documentation of that API (docs/0.reference-manual.md, docs/3.creating-blocks.md), not copied.
"""
import ctypes as C
import os
import time

import numpy as np

from . import minilua as ml

LuaTable, LuaError = ml.LuaTable, ml.LuaError


def T(**kw):
    t = LuaTable()
    for k, v in kw.items():
        t.set(k, v)
    return t


def L(*items):
    t = LuaTable()
    for i, v in enumerate(items):
        t.set(i + 1, v)
    return t


# ------------------------------------------------------------------------------------------------------------------------------- cdata
class HelloStruct(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserved", C.c_int32), ("capacity", C.c_uint64), ("mem", (C.c_uint8 * 64) * 2),
                ("filled", (C.c_uint8 * 64) * 2), ("consumed", (C.c_uint8 * 64) * 2)]


class TokenStruct(C.Structure):
    _fields_ = [("k", C.c_int64), ("n", C.c_int64)]


STRUCTS = {"lrhip_fanout_hello_t": HelloStruct, "lrhip_fanout_token_t": TokenStruct}
SCALARS = {"int": C.c_int, "unsigned": C.c_uint, "unsigned int": C.c_uint, "long": C.c_long, "unsigned long": C.c_ulong, "unsigned long long": C.c_ulonglong,
           "uint8_t": C.c_uint8, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float, "double": C.c_double, "char": C.c_char}


class CData:
    """a typed address.  elem = ctypes type of what it points to (None: void / opaque); owner keeps the memory alive"""
    lua_type = "cdata"

    def __init__(self, addr, elem, owner=None, count=None, is_struct=False):
        self.addr, self.elem, self.owner, self.count, self.is_struct = int(addr or 0), elem, owner, count, is_struct
        self.finalizer = None

    # -- helpers
    def _obj(self):
        return self.elem.from_address(self.addr)

    def lua_is_null(self):
        return self.addr == 0

    def lua_tostring(self):
        return "cdata<%s>: 0x%x" % (getattr(self.elem, "__name__", "void"), self.addr)

    def lua_eq(self, other):
        return isinstance(other, CData) and other.addr == self.addr

    def lua_tonumber(self):
        return float(self.addr)

    def _wrap(self, obj, elem):
        if isinstance(obj, C.Array):
            return CData(C.addressof(obj), obj._type_, (self.owner if self.owner is not None else self), len(obj))
        if isinstance(obj, C.Structure):
            return CData(C.addressof(obj), type(obj), (self.owner if self.owner is not None else self), 1, True)
        if isinstance(obj, bytes):
            return float(obj[0])
        return float(obj) if isinstance(obj, (int, float)) else obj

    def lua_index(self, key):
        if isinstance(key, str):
            if not self.is_struct:
                raise LuaError("cdata: no field '%s' on a non-struct" % key)
            return self._wrap(getattr(self._obj(), key), None)
        i = int(key)
        if self.elem is None:
            raise LuaError("cdata: cannot index void *")
        if self.elem is C.c_void_p:                      # array of pointers
            v = C.c_void_p.from_address(self.addr + 8 * i).value
            return CData(v or 0, None)
        if issubclass(self.elem, (C.Array, C.Structure)):
            return self._wrap(self.elem.from_address(self.addr + C.sizeof(self.elem) * i), None)
        return self._wrap(self.elem.from_address(self.addr + C.sizeof(self.elem) * i).value, None)

    def lua_newindex(self, key, val):
        if isinstance(key, str):
            field = dict((f[0], f[1]) for f in self.elem._fields_).get(key)
            setattr(self._obj(), key, float(val) if field in (C.c_float, C.c_double) else (int(val) if isinstance(val, float) else val))
            return
        i = int(key)
        if self.elem is C.c_void_p:
            C.c_void_p.from_address(self.addr + 8 * i).value = val.addr if isinstance(val, CData) else int(val or 0)
            if isinstance(val, CData):
                self.__dict__.setdefault("_keep", []).append(val)
            return
        cell = self.elem.from_address(self.addr + C.sizeof(self.elem) * i)
        cell.value = (float(val) if self.elem in (C.c_float, C.c_double) else int(val))

    def lua_arith(self, op, a, b):
        if op in ("+", "-") and isinstance(a, CData) and not isinstance(b, CData):
            size = C.sizeof(self.elem) if self.elem is not None else 1
            n = int(ml.tonum(b))
            return CData(self.addr + (n if op == "+" else -n) * size, self.elem, (self.owner if self.owner is not None else self), None, self.is_struct)
        if op == "-" and isinstance(a, CData) and isinstance(b, CData):
            size = C.sizeof(self.elem) if self.elem is not None else 1
            return float((a.addr - b.addr) // size)
        raise LuaError("cdata: unsupported arithmetic %s" % op)


def parse_ctype(spec):
    """-> (element ctypes type or None, is pointer/array, count or None ('?' -> -1), is_struct)"""
    s = " ".join(spec.replace("*", " * ").replace("[", " [").split())
    s = s.replace("const ", "").replace(" const", "").replace("struct ", "")
    arr = None
    if "[" in s:
        s, dims = s.split(" [", 1)
        dims = dims.rstrip("]")
        arr = -1 if dims.strip() == "?" else int(dims)
    stars = s.count("*")
    base = s.replace("*", "").strip()
    if base in STRUCTS:
        elem, is_struct = STRUCTS[base], True
    elif base in SCALARS:
        elem, is_struct = SCALARS[base], False
    elif base == "void" or base.startswith("lrhip_"):
        elem, is_struct = None, False
    else:
        raise LuaError("minilua ffi: unknown C type %r" % spec)
    return elem, stars, arr, is_struct


class ComplexFloat32Struct(C.Structure):
    _fields_ = [("real", C.c_float), ("imag", C.c_float)]


class Float32Struct(C.Structure):
    _fields_ = [("value", C.c_float)]


class ByteStruct(C.Structure):
    _fields_ = [("value", C.c_uint8)]


ELEMENT_STRUCTS = {"ComplexFloat32": ComplexFloat32Struct, "Float32": Float32Struct, "Byte": ByteStruct}


class CType:
    """an opaque C type of the reference's tables (radio/utilities/format_utils.lua:82-97 real_ctype / complex_ctype): only its size is looked at"""
    lua_type = "cdata"

    def __init__(self, name, size):
        self.name, self.size = name, size

    def lua_tostring(self):
        return "ctype<%s>" % self.name


class DataType:
    """a sample type of the reference (radio.types.*): ComplexFloat32 / Float32 / Byte / Bit as far as the glue looks at them"""
    lua_type = "cdata"

    def __init__(self, name, dtype):
        self.name, self.dtype = name, np.dtype(dtype)
        self.struct = ELEMENT_STRUCTS.get(name)

    def lua_index(self, key):
        if key == "vector":
            return lambda n=0: Vector(self, int(ml.tonum(n or 0)))
        if key == "type_name":
            return self.name
        return None

    def lua_tostring(self):
        return self.name

    def lua_eq(self, other):
        return isinstance(other, DataType) and other.name == self.name


class Vector:
    """radio.core.vector: page-aligned owning buffer, resize() keeps the capacity and grows by doubling (radio/core/vector.lua:19-37, 108-136)"""
    lua_type = "table"

    def __init__(self, data_type, length=0, array=None):
        self.data_type = data_type
        self.owning = array is None
        if array is not None:
            self.buf, self.length, self.capacity = array, len(array), len(array)
        else:
            self.length, self.capacity = length, max(length, 0)
            self._alloc(self.capacity)
        self.reallocations = 0

    def _alloc(self, count):
        raw = np.zeros(count * self.data_type.dtype.itemsize + 4096, np.uint8)
        off = (-raw.ctypes.data) % 4096
        self._raw = raw
        self.buf = raw[off:off + count * self.data_type.dtype.itemsize].view(self.data_type.dtype)

    def resize(self, n):
        n = int(ml.tonum(n))
        if n > self.capacity:
            old = self.buf[:self.length].copy()
            self.capacity = max(n, 2 * self.capacity)
            self._alloc(self.capacity)
            self.buf[:len(old)] = old
            self.reallocations += 1
        self.length = n
        return self

    def array(self):
        return self.buf[:self.length]

    def lua_index(self, key):
        if key == "data":
            # a typed pointer: `x.data + n` steps whole samples, `x.data[i].value` reads one (radio/types/*.lua structs)
            st = self.data_type.struct
            return CData(self.buf.ctypes.data, st, self, None, st is not None)
        if key == "length":
            return float(self.length)
        if key == "size":
            return float(self.length * self.data_type.dtype.itemsize)
        if key == "_capacity":
            return float(self.capacity)
        if key == "data_type":
            return self.data_type
        if key == "resize":
            return lambda self_, n: self.resize(n)
        return None


# ------------------------------------------------------------------------------------------------------------------------- the library
class FakeLib:
    """CPU box: records every lrhip_* call, hands out fake handles, emits as many samples as it was given (rate 1)"""

    def __init__(self):
        import threading
        self.calls, self.next_handle, self.devices = [], 0x1000, 8
        self.pending = 0
        self._tls = threading.local()          # one "process" per thread: the device binding is per process

    @property
    def device(self):
        return getattr(self._tls, "device", -1)

    @device.setter
    def device(self, v):
        self._tls.device = v

    def call(self, name, args):
        self.calls.append((name, args))
        if name == "lrhip_device_count":
            return self.devices
        if name == "lrhip_init":
            d = args[0]
            if self.device >= 0 and d >= 0 and d != self.device:
                return -1
            if self.device < 0:
                self.device = d if d >= 0 else 0
            return 0
        if name == "lrhip_device":
            return self.device
        if name == "lrhip_strerror":
            return b"fake error"
        if name == "lrhip_version":
            return b"fake 0.0"
        st = self.__dict__.setdefault("stage_info", {})
        if name in ("lrhip_format_convert_create", "lrhip_format_pack_create"):
            fmt = args[0].decode() if isinstance(args[0], bytes) else str(args[0])
            size = {"8": 1, "16": 2, "32": 4, "64": 8}[fmt.strip("usflbe")] * (2 if args[1] else 1)
            self.next_handle += 0x100
            st[self.next_handle] = {"kind": name, "in": size if "convert" in name else (8 if args[1] else 4), "out": (8 if args[1] else 4) if "convert" in name else size}
            return self.next_handle
        if name == "lrhip_welch_create":
            self.next_handle += 0x100
            st[self.next_handle] = {"kind": "welch", "n": int(args[0]), "hop": int(args[0]) - int(args[5]), "pending": 0, "frames": 0, "in": 8 if args[4] else 4, "out": 4}
            return self.next_handle
        if name == "lrhip_welch_read":
            w = st[int(args[0])]
            frames = w["frames"]
            if args[2]:
                w["frames"] = 0
            return frames
        if name == "lrhip_stage_execute" and int(args[0]) in st and st[int(args[0])]["kind"] == "welch":
            w = st[int(args[0])]
            total = w["pending"] + int(args[2])
            nf = (total - w["n"]) // w["hop"] + 1 if total >= w["n"] else 0
            w["frames"] += nf
            w["pending"] = total - nf * w["hop"]
            return 0
        if name == "lrhip_chain_create_ex":
            self.next_handle += 0x100
            handles = list((C.c_void_p * int(args[1])).from_address(int(args[0])))
            st[self.next_handle] = {"kind": "chain", "stages": [int(h or 0) for h in handles], "flags": int(args[2]), "queue": [], "depth": 0, "slots": []}
            return self.next_handle
        if name == "lrhip_chain_set_ring":
            c = st.get(int(args[0]))
            if c is not None:
                c["depth"], c["chunk"] = int(args[1]), int(args[2])
                c["slots"] = [C.create_string_buffer(int(args[2]) * 16 + 64) for _ in range(int(args[1]))]
                c["head"] = 0
            return 0
        if name == "lrhip_chain_ring_input":
            c = st[int(args[0])]
            if len(c["queue"]) >= c["depth"]:
                return 0
            return C.addressof(c["slots"][c["head"] % c["depth"]])
        if name == "lrhip_chain_submit_fd":
            c = st[int(args[0])]
            if getattr(self, "fd_is_fifo", False):
                return -4
            if len(c["queue"]) >= c["depth"]:
                return -3
            rec = st[c["stages"][0]]["in"]
            avail = max(os.fstat(int(args[1])).st_size - int(args[2]), 0) // rec
            n = min(int(args[3]), c["chunk"], avail)
            if n:
                os.pread(int(args[1]), n * rec, int(args[2]))
                c["queue"].append(n)
                c["head"] += 1
            return n
        if name == "lrhip_chain_submit":
            c = st[int(args[0])]
            c["queue"].append(int(args[2]))
            c["head"] += 1
            return int(args[2])
        if name == "lrhip_chain_collect":
            c = st[int(args[0])]
            return c["queue"].pop(0) if c["queue"] else -2
        if name == "lrhip_chain_in_flight":
            return len(st[int(args[0])]["queue"])
        if name == "lrhip_chain_last_launches":
            return 1
        if name == "lrhip_chain_start_at":
            if args[2]:
                C.c_ulonglong.from_address(int(args[2])).value = max(int(args[1]) - 127, 0)
            return 0
        if name == "lrhip_chain_halo":
            return 127
        if name == "lrhip_chain_shard_align":
            return 1
        if name in ("lrhip_stage_input_size", "lrhip_stage_output_size") and int(args[0]) in st and "in" in st[int(args[0])]:
            return st[int(args[0])]["in" if name.endswith("input_size") else "out"]
        if name in ("lrhip_stage_execute2", "lrhip_stage_execute2_device"):
            return int(args[3])
        if name == "lrhip_stage_execute_device":
            return int(args[2])
        if name.endswith("_create") or name.endswith("_create_ex") or name in ("lrhip_malloc", "lrhip_host_alloc", "lrhip_ipc_open", "lrhip_ipc_event_open"):
            self.next_handle += 0x100
            if name in ("lrhip_malloc", "lrhip_host_alloc"):
                buf = C.create_string_buffer(int(args[0]) + 64)
                self.__dict__.setdefault("_bufs", []).append(buf)
                return C.addressof(buf)
            return self.next_handle
        if name in ("lrhip_stage_max_output", "lrhip_chain_max_output"):
            return int(args[1])
        if name == "lrhip_chain_push_bound":
            return int(args[1]) + 4096
        if name == "lrhip_chain_push":
            self.pending += int(args[2])
            return 0
        if name in ("lrhip_chain_flush", "lrhip_chain_poll"):
            n, self.pending = self.pending, 0
            return n
        if name == "lrhip_chain_poll_due":
            return -1.0
        if name in ("lrhip_stage_execute", "lrhip_chain_execute", "lrhip_chain_execute_device"):
            return int(args[2])
        if name in ("lrhip_stage_input_size", "lrhip_stage_output_size"):
            return 8
        return 0


class LibProxy:
    lua_type = "userdata"

    def __init__(self, real=None):
        from luaradio_amd import _lib
        self.sigs = _lib.SIGNATURES
        self.real = real                       # a ctypes CDLL with restype / argtypes set (luaradio_amd._lib.load()) or None
        self.fake = FakeLib() if real is None else None
        self.trace = []

    def lua_index(self, name):
        if name not in self.sigs:
            raise LuaError("lib.%s: not declared in include/lrhip.h" % name)
        restype, argtypes = self.sigs[name]

        def fn(*args):
            if len(args) != len(argtypes):
                raise LuaError("lib.%s: %d arguments given, %d declared" % (name, len(args), len(argtypes)))
            conv, keep = [], []
            for a, t in zip(args, argtypes):
                if isinstance(a, CData):
                    conv.append(a.addr)
                elif isinstance(a, Vector):
                    conv.append(a.buf.ctypes.data)
                elif a is None:
                    conv.append(None if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") else 0)
                elif isinstance(a, str):
                    b = a.encode()
                    keep.append(b)
                    conv.append(b)
                elif isinstance(a, bool):
                    conv.append(int(a))
                elif t in (C.c_float, C.c_double):
                    conv.append(float(a))
                else:
                    conv.append(int(a))
            self.trace.append(name)
            if self.real is not None:
                f = getattr(self.real, name)
                fixed = []
                for v, t in zip(conv, argtypes):
                    if hasattr(t, "contents") and isinstance(v, int):          # POINTER(x) parameters take an address
                        fixed.append(C.cast(C.c_void_p(v), t))
                    else:
                        fixed.append(v)
                r = f(*fixed)
            else:
                r = self.fake.call(name, conv)
            if restype is None:
                return None
            if restype is C.c_void_p:
                return CData(r or 0, None)
            if restype is C.c_char_p:
                return r if isinstance(r, bytes) else (r or b"")
            return float(r)
        return fn


# ------------------------------------------------------------------------------------------------------------------------------- ffi
def make_ffi(interp, lib_proxy, sockets=None):
    ffi = LuaTable()
    state = {"errno": 0}

    def cdef(text):
        return None

    def new(spec, *init):
        elem, stars, arr, is_struct = parse_ctype(spec)
        if arr is not None:
            count = int(ml.tonum(init[0])) if arr == -1 else arr
            et = C.c_void_p if stars >= 1 else elem
            buf = (et * max(count, 1))()
            return CData(C.addressof(buf), et, buf, count)
        if is_struct and stars == 0:
            obj = elem()
            return CData(C.addressof(obj), elem, obj, 1, True)
        raise LuaError("minilua ffi.new: unsupported type %r" % spec)

    def cast(spec, value):
        elem, stars, arr, is_struct = parse_ctype(spec)
        if isinstance(value, CData):
            return CData(value.addr, elem, (value.owner if value.owner is not None else value), None, is_struct)
        if isinstance(value, Vector):
            return CData(value.buf.ctypes.data, elem, value)
        if isinstance(value, (int, float)):
            return CData(int(value), elem)
        raise LuaError("minilua ffi.cast: cannot cast %s" % ml.lua_type(value))

    def gc(obj, fin):
        if isinstance(obj, CData):
            obj.finalizer = fin
        return obj

    def string(v, n=None):
        if isinstance(v, bytes):
            return v.decode(errors="replace")
        if isinstance(v, CData):
            return C.string_at(v.addr).decode(errors="replace") if n is None else C.string_at(v.addr, int(n)).decode("latin-1")     # binary data: one character per byte
        return ml.tostr(v)

    def sizeof(v):
        if isinstance(v, CData):
            if v.count is not None and v.elem is not None and not v.is_struct:
                return float(C.sizeof(v.elem) * v.count)
            return float(C.sizeof(v.elem)) if v.elem is not None else 8.0
        if isinstance(v, DataType):
            return float(v.dtype.itemsize)
        if isinstance(v, CType):
            return float(v.size)
        if isinstance(v, str):
            elem, stars, arr, _ = parse_ctype(v)
            base = 8 if stars else C.sizeof(elem)
            return float(base * (arr if arr and arr > 0 else 1))
        raise LuaError("minilua ffi.sizeof: unsupported operand")

    def istype(ct, obj):
        return False

    def fill(dst, n, c=0):
        C.memset(dst.addr, int(ml.tonum(c or 0)), int(ml.tonum(n)))

    def copy(dst, src, n=None):
        if isinstance(src, str):
            b = src.encode() + b"\0"
            C.memmove(dst.addr, b, len(b))
            return
        C.memmove(dst.addr, src.addr, int(ml.tonum(n)))

    # ---- ffi.C: the POSIX calls the glue makes itself
    Cns = LuaTable()

    def c_socketpair(domain, typ, proto, fds):
        import socket
        a, b = socket.socketpair()
        # raw descriptors, owned by the Lua code from here on (it closes them with ffi.C.close): a Python socket object closing the same number again
        # later would hit whatever file got that number meanwhile
        fds.lua_newindex(0, a.detach())
        fds.lua_newindex(1, b.detach())
        return 0.0

    def c_read(fd, buf, n):
        try:
            data = os.read(int(fd), int(ml.tonum(n)))
        except OSError as e:
            state["errno"] = e.errno
            return -1.0
        C.memmove(buf.addr, data, len(data))
        return float(len(data))

    def c_write(fd, buf, n):
        try:
            if isinstance(buf, str):             # LuaJIT converts a Lua string to const void *
                return float(os.write(int(fd), buf.encode("latin-1")[:int(ml.tonum(n))]))
            return float(os.write(int(fd), C.string_at(buf.addr, int(ml.tonum(n)))))
        except OSError as e:
            state["errno"] = e.errno
            return -1.0

    def c_close(fd):
        try:
            os.close(int(fd))
        except OSError:
            return -1.0
        return 0.0

    def c_poll(pollfds, nfds, timeout_ms):
        hook = state.get("poll_hook")
        if hook is not None:
            return float(hook(pollfds, int(nfds), ml.tonum(timeout_ms)))
        return 0.0

    # ---- stdio, as radio/blocks/sources/iqfile.lua and radio/blocks/sinks/iqfile.lua use it (FILE * = an object holding a Python file)
    class CFile:
        lua_type = "cdata"

        def __init__(self, fh):
            self.fh, self.eof, self.closed = fh, False, False

        def lua_is_null(self):
            return False

        def lua_eq(self, other):
            return other is self

        def lua_tostring(self):
            return "cdata<FILE *>"

    def c_fopen(path, mode):
        try:
            return CFile(open(path, mode if "b" in mode else mode + "b"))
        except OSError as e:
            state["errno"] = e.errno
            return CData(0, None)

    def c_fread(buf, size, count, f):
        size, count = int(ml.tonum(size)), int(ml.tonum(count))
        data = f.fh.read(size * count)
        if len(data) < size * count:
            f.eof = True
        whole = len(data) // size
        C.memmove(buf.addr, data, whole * size)
        state.setdefault("fread_sizes", []).append((size, count, whole))
        return float(whole)

    def c_fwrite(buf, size, count, f):
        size, count = int(ml.tonum(size)), int(ml.tonum(count))
        f.fh.write(C.string_at(buf.addr, size * count))
        return float(count)

    def c_rewind(f):
        f.fh.seek(0)
        f.eof = False

    def c_fclose(f):
        f.fh.close()
        f.closed = True
        return 0.0

    Cns.set("lseek", lambda fd, off, whence: float(os.lseek(int(fd), int(ml.tonum(off)), int(ml.tonum(whence)))))
    for name, f in (("fopen", c_fopen), ("fread", c_fread), ("fwrite", c_fwrite), ("feof", lambda f: 1.0 if f.eof else 0.0), ("ferror", lambda f: 0.0),
                    ("rewind", c_rewind), ("fclose", c_fclose), ("fileno", lambda f: float(f.fh.fileno())), ("ftell", lambda f: float(f.fh.tell())),
                    ("fseek", lambda f, off, whence: (f.fh.seek(int(ml.tonum(off)), int(ml.tonum(whence))), setattr(f, "eof", False), 0.0)[2])):
        Cns.set(name, f)

    for name, f in (("getpid", lambda: float(os.getpid())), ("socketpair", c_socketpair), ("read", c_read), ("write", c_write), ("close", c_close),
                    ("poll", c_poll), ("strerror", lambda e: os.strerror(int(e)).encode())):
        Cns.set(name, f)
    # ---- the helper process of lrhip.in_helper (lua/radio/core/lrhip.lua): REAL pipe / fork / waitpid / _exit - the child is a copy of this interpreter
    def c_pipe(fds):
        r, w = os.pipe()
        fds.lua_newindex(0, r)
        fds.lua_newindex(1, w)
        return 0.0

    def c_fork():
        pid = os.fork()
        if pid == 0:
            state["forked_child"] = True
            import signal
            signal.signal(signal.SIGALRM, signal.SIG_DFL)
            signal.alarm(30)                     # a child that escapes the glue's _exit (a bug under test) must not live on as a second pytest
        else:
            state.setdefault("forked_pids", []).append(pid)
        return float(pid)

    def c_waitpid(pid, status, options):
        p, st = os.waitpid(int(pid), int(ml.tonum(options)))
        if status is not None:
            status.lua_newindex(0, st)
        return float(p)

    def c__exit(code):
        os._exit(int(ml.tonum(code)))

    for name, f in (("pipe", c_pipe), ("fork", c_fork), ("waitpid", c_waitpid), ("_exit", c__exit)):
        Cns.set(name, f)
    for name, v in (("SEEK_SET", 0.0), ("SEEK_CUR", 1.0), ("SEEK_END", 2.0)):
        Cns.set(name, v)
    Cns.set("AF_UNIX", 1.0)
    Cns.set("SOCK_STREAM", 1.0)

    def load(name, *a):
        if "lrhip" not in name:
            raise LuaError("cannot load '%s'" % name)
        return lib_proxy

    for name, f in (("cdef", cdef), ("new", new), ("cast", cast), ("gc", gc), ("string", string), ("sizeof", sizeof), ("copy", copy), ("istype", istype), ("fill", fill),
                    ("errno", lambda: float(state["errno"])), ("load", load)):
        ffi.set(name, f)
    ffi.set("C", Cns)
    ffi.set("_state", state)
    return ffi


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MOCK_LUA = os.path.join(ROOT, "tests", "lua_mocks")
GLUE_LUA = os.path.join(ROOT, "lua")


def make_interpreter(real_lib=None, env_vars=None):
    """an interpreter with ffi, the mocked reference core and the glue on its search path; returns (interp, lib_proxy, ffi table)"""
    interp = ml.Interpreter(search_paths=[MOCK_LUA, GLUE_LUA], env_vars=env_vars)
    proxy = LibProxy(real_lib)
    ffi = make_ffi(interp, proxy)
    interp.register("ffi", ffi)
    interp.register("math", interp.globals.get("math"))
    interp.register("string", interp.globals.get("string"))
    interp.register("table", interp.globals.get("table"))
    types = LuaTable()
    for name, dt in (("ComplexFloat32", np.complex64), ("Float32", np.float32), ("Byte", np.uint8)):
        types.set(name, DataType(name, dt))
    interp.register("radio.types", types)
    # radio/utilities/format_utils.lua:82-97 as far as the file blocks look at it: the ctypes' sizes
    sizes = {"u8": 1, "s8": 1, "u16le": 2, "u16be": 2, "s16le": 2, "s16be": 2, "u32le": 4, "u32be": 4, "s32le": 4, "s32be": 4,
             "f32le": 4, "f32be": 4, "f64le": 8, "f64be": 8}
    formats = LuaTable()
    for name, size in sizes.items():
        formats.set(name, T(real_ctype=CType("format_%s_t" % name, size), complex_ctype=CType("iq_format_%s_t" % name, 2 * size), swap=name.endswith("be")))
    interp.register("radio.utilities.format_utils", T(formats=formats))
    interp.globals.set("__now_us", lambda: float(time.time() * 1e6))
    return interp, proxy, ffi
