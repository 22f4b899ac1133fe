"""File sources feeding the device path - mirrors radio/blocks/sources/iqfile.lua and realfile.lua.

The file is read on the host in the reference's chunks (8192 samples, iqfile.lua:52); the RAW records are handed to
the device unchanged and converted there ((value - offset)/scale after the byte swap, format_utils.lua:82-97), so
an RTL-SDR style `u8` capture crosses PCIe at 2 bytes per complex sample instead of 8.
"""
import ctypes as C
import io

import numpy as np

from . import _lib, types
from .block import Block, Input, Output

# radio/utilities/format_utils.lua:82-97 (bytes per raw scalar)
FORMAT_BYTES = {"u8": 1, "s8": 1, "u16le": 2, "u16be": 2, "s16le": 2, "s16be": 2, "u32le": 4, "u32be": 4,
                "s32le": 4, "s32be": 4, "f32le": 4, "f32be": 4, "f64le": 8, "f64be": 8}


class _FileSource(Block):
    _complex = True

    def instantiate(self, file, format, rate, repeat_on_eof=False):
        assert file is not None, "Missing argument #1 (file)"
        assert format, "Missing argument #2 (format)"
        if format not in FORMAT_BYTES:
            raise AssertionError('Unsupported format ("%s")' % format)       # iqfile.lua:48
        assert rate, "Missing argument #3 (rate)"
        self.file, self.format, self.rate = file, format, rate
        self.repeat_on_eof = repeat_on_eof or False
        self.chunk_size = 8192
        self.record_size = FORMAT_BYTES[format] * (2 if self._complex else 1)
        out_type = types.ComplexFloat32 if self._complex else types.Float32
        self.add_type_signature([], [Output("out", out_type)])
        self.signature = self.type_signatures[0]

    def get_rate(self):
        return self.rate

    def initialize(self):
        if isinstance(self.file, (bytes, bytearray)):
            self._fh = io.BytesIO(bytes(self.file))          # tests/buffer.lua: an in-memory file
        elif isinstance(self.file, str):
            self._fh = open(self.file, "rb")
        else:
            self._fh = self.file
        self._set_stage(_lib.load().lrhip_format_convert_create(self.format.encode(), int(self._complex)),
                        "Creating lrhip format object")

    def process(self):
        """One chunk, or None at end of file (iqfile.lua:82-96)."""
        raw = self._fh.read(self.chunk_size * self.record_size)
        num_samples = len(raw) // self.record_size
        if num_samples == 0 and self.repeat_on_eof:
            self._fh.seek(0)                  # iqfile.lua:86-90: rewind once and read again; a file without one whole record still ends
            raw = self._fh.read(self.chunk_size * self.record_size)
            num_samples = len(raw) // self.record_size
        if num_samples == 0:
            return None
        L = _lib.load()
        buf = np.frombuffer(raw, dtype=np.uint8, count=num_samples * self.record_size)
        out = np.empty(num_samples, dtype=self.get_output_type().dtype)
        n = L.lrhip_stage_execute(self._stage, buf.ctypes.data_as(C.c_void_p), num_samples, out.ctypes.data_as(C.c_void_p), num_samples)
        _lib.check(n, "%s:process" % self.name)
        return out

    def read_all(self):
        """Convenience: the whole file as one vector (chunks concatenated)."""
        parts = []
        while True:
            v = self.process()
            if v is None:
                break
            parts.append(v)
        return np.concatenate(parts) if parts else np.zeros(0, self.get_output_type().dtype)

    def cleanup(self):
        if isinstance(self.file, str):
            self._fh.close()


class IQFileSource(_FileSource):
    """radio/blocks/sources/iqfile.lua. IQFileSource(file, format, rate[, repeat_on_eof])."""
    name = "IQFileSource"
    _complex = True


class RealFileSource(_FileSource):
    """radio/blocks/sources/realfile.lua. RealFileSource(file, format, rate[, repeat_on_eof])."""
    name = "RealFileSource"
    _complex = False


class _FileSink(Block):
    """radio/blocks/sinks/iqfile.lua / realfile.lua: samples -> raw records on the device, written by the host."""
    _complex = True

    def instantiate(self, file, format):
        assert file is not None, "Missing argument #1 (file)"
        assert format, "Missing argument #2 (format)"
        if format not in FORMAT_BYTES:
            raise AssertionError('Unsupported format ("%s")' % format)
        self.file, self.format = file, format
        self.record_size = FORMAT_BYTES[format] * (2 if self._complex else 1)
        self.add_type_signature([Input("in", types.ComplexFloat32 if self._complex else types.Float32)], [])

    def initialize(self):
        self._fh = open(self.file, "wb") if isinstance(self.file, str) else self.file
        self._set_stage(_lib.load().lrhip_format_pack_create(self.format.encode(), int(self._complex)), "Creating lrhip format object")

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=self.get_input_type().dtype)
        out = np.empty(len(x) * self.record_size, np.uint8)
        n = _lib.load().lrhip_stage_execute(self._stage, x.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p), len(x))
        _lib.check(n, "%s:process" % self.name)
        self._fh.write(out.tobytes())

    def cleanup(self):
        if isinstance(self.file, str):
            self._fh.close()
        else:
            self._fh.flush()


class IQFileSink(_FileSink):
    """radio/blocks/sinks/iqfile.lua. IQFileSink(file, format)."""
    name = "IQFileSink"
    _complex = True


class RealFileSink(_FileSink):
    """radio/blocks/sinks/realfile.lua. RealFileSink(file, format)."""
    name = "RealFileSink"
    _complex = False
