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
            return CData(C.addressof(obj), obj._type_, self.owner or self, len(obj))
        if isinstance(obj, C.Structure):
            return CData(C.addressof(obj), type(obj), self.owner or self, 1, True)
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
            setattr(self._obj(), key, int(val) if isinstance(val, float) else val)
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
            return CData(self.addr + (n if op == "+" else -n) * size, self.elem, self.owner or self, None, self.is_struct)
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


class DataType:
    """a sample type of the reference (radio.types.*): ComplexFloat32 / Float32 / Byte / Bit as far as the glue looks at them"""
    lua_type = "cdata"

    def __init__(self, name, dtype):
        self.name, self.dtype = name, np.dtype(dtype)

    def lua_index(self, key):
        if key == "vector":
            return lambda n=0: Vector(self, int(ml.tonum(n or 0)))
        if key == "type_name":
            return self.name
        return None

    def lua_tostring(self):
        return self.name


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
            return CData(self.buf.ctypes.data, None, self)
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
            return CData(value.addr, elem, value.owner or value, None, is_struct)
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
            return (C.string_at(v.addr) if n is None else C.string_at(v.addr, int(n))).decode(errors="replace")
        return ml.tostr(v)

    def sizeof(v):
        if isinstance(v, CData):
            if v.count is not None and v.elem is not None and not v.is_struct:
                return float(C.sizeof(v.elem) * v.count)
            return float(C.sizeof(v.elem)) if v.elem is not None else 8.0
        if isinstance(v, DataType):
            return float(v.dtype.itemsize)
        if isinstance(v, str):
            elem, stars, arr, _ = parse_ctype(v)
            base = 8 if stars else C.sizeof(elem)
            return float(base * (arr if arr and arr > 0 else 1))
        raise LuaError("minilua ffi.sizeof: unsupported operand")

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

    for name, f in (("getpid", lambda: float(os.getpid())), ("socketpair", c_socketpair), ("read", c_read), ("write", c_write), ("close", c_close),
                    ("poll", c_poll), ("strerror", lambda e: os.strerror(int(e)).encode())):
        Cns.set(name, f)
    Cns.set("AF_UNIX", 1.0)
    Cns.set("SOCK_STREAM", 1.0)

    def load(name, *a):
        if "lrhip" not in name:
            raise LuaError("cannot load '%s'" % name)
        return lib_proxy

    for name, f in (("cdef", cdef), ("new", new), ("cast", cast), ("gc", gc), ("string", string), ("sizeof", sizeof), ("copy", copy),
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
    interp.globals.set("__now_us", lambda: float(time.time() * 1e6))
    return interp, proxy, ffi
