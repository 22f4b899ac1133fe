"""The benchmark suite's feeders and meter: ZeroSource, RawFileSource, BenchmarkSink and the suite's trial
protocol - mirrors of radio/blocks/sources/zero.lua, radio/blocks/sources/rawfile.lua, radio/blocks/sinks/benchmark.lua and
benchmarks/luaradio_benchmark.lua:690-738.

None of them computes anything: a source hands over vectors, the sink counts them.  What changes on the device path is WHERE the
vectors live: ZeroSource keeps one zero vector resident in HBM (process_device() returns its pointer) so that a benchmarked block is
fed without a host copy, exactly as the reference's ZeroSource feeds the same memory to every process() call (zero.lua:38-44)."""
import ctypes as C
import io
import json
import math
import sys
import time

import numpy as np

from . import _lib
from .block import Block, Input, Output


class ZeroSource(Block):
    """radio/blocks/sources/zero.lua. ZeroSource(data_type, rate): chunk_size = 8192 zero samples per process()."""
    name = "ZeroSource"

    def instantiate(self, data_type, rate, chunk_size=8192):
        assert data_type is not None, "Missing argument #1 (data_type)"
        assert rate, "Missing argument #2 (rate)"
        self.data_type, self.rate, self.chunk_size = data_type, rate, int(chunk_size)
        self.add_type_signature([], [Output("out", data_type)])
        self.signature = self.type_signatures[0]
        self._dev = None

    def get_rate(self):
        return self.rate

    def initialize(self):
        self.out = np.zeros(self.chunk_size, self.data_type.dtype)
        L = _lib.load()
        self._dev = _lib.check_ptr(L.lrhip_malloc(self.out.nbytes), "ZeroSource: device vector")
        _lib.check(L.lrhip_memcpy_h2d(self._dev, self.out.ctypes.data_as(C.c_void_p), self.out.nbytes), "ZeroSource: upload")

    def process(self):
        return self.out

    def process_device(self):
        """(device pointer, number of samples) of the resident zero vector"""
        return self._dev, self.chunk_size

    def cleanup(self):
        if self._dev:
            _lib.load().lrhip_free(self._dev)
            self._dev = None


class RawFileSource(Block):
    """radio/blocks/sources/rawfile.lua. RawFileSource(file, data_type, rate[, repeat_on_eof]): elements of `data_type` in their in-memory
    layout (ComplexFloat32 = two little-endian floats, complexfloat32.lua:19-24), read through a 262 144-byte buffer; a partial element at
    the end of a read stays in the buffer for the next call (rawfile.lua:77-108)."""
    name = "RawFileSource"

    def instantiate(self, file, data_type, rate, repeat_on_eof=False):
        assert file is not None, "Missing argument #1 (file)"
        assert data_type is not None, "Missing argument #2 (data_type)"
        assert rate, "Missing argument #3 (rate)"
        self.file, self.data_type, self.rate = file, data_type, rate
        self.repeat_on_eof = repeat_on_eof or False
        self.add_type_signature([], [Output("out", data_type)])
        self.signature = self.type_signatures[0]

    def get_rate(self):
        return self.rate

    def initialize(self):
        if isinstance(self.file, (bytes, bytearray)):
            self._fh = io.BytesIO(bytes(self.file))
        elif isinstance(self.file, str):
            self._fh = open(self.file, "rb")
        else:
            self._fh = self.file
        self.buf_capacity = 262144
        self._pending = b""

    def process(self):
        """the elements read this call, or None at end of file"""
        want = self.buf_capacity - len(self._pending)
        data = self._fh.read(want)
        if len(data) < want and len(data) == 0:
            if self.repeat_on_eof:
                self._fh.seek(0)            # rawfile.lua:88: rewind; this call delivers whatever was pending
            else:
                return None
        buf = self._pending + data
        size = self.data_type.size
        count = len(buf) // size
        self._pending = buf[count * size:]
        return np.frombuffer(buf, dtype=self.data_type.dtype, count=count).copy()

    def cleanup(self):
        if isinstance(self.file, str):
            self._fh.close()


def _normalize(amount):
    """benchmark.lua:54-64"""
    if amount > 1e9:
        return amount / 1e9, "G"
    if amount > 1e6:
        return amount / 1e6, "M"
    if amount > 1e3:
        return amount / 1e3, "K"
    return amount, ""


class BenchmarkSink(Block):
    """radio/blocks/sinks/benchmark.lua. BenchmarkSink([file[, use_json[, title]]]): counts the samples it is handed; every 3 s a report line
    "[title] x.xx MS/s (y.yy MB/s)", or with use_json one {"samples_per_second", "bytes_per_second"} object at cleanup().
    process() takes a vector or just its length (a device-resident vector has no host object to pass)."""
    name = "BenchmarkSink"

    def instantiate(self, file=None, use_json=False, title=None, clock=time.perf_counter):
        self.file = sys.stderr if file is None else file
        self.use_json = use_json or False
        self.title = title or "BenchmarkSink"
        self.report_period = 3.0
        self._clock = clock
        self.add_type_signature([Input("in", None)], [])          # accepts every type (benchmark.lua:50)
        self.elem_size = 8

    def differentiate(self, input_types):
        self.input_type = input_types[0]
        self.elem_size = getattr(input_types[0], "size", 8)

    def get_input_type(self, index=1):
        return getattr(self, "input_type", None)

    def initialize(self):
        self._fh = open(self.file, "w") if isinstance(self.file, str) else self.file
        self.count = 0
        self.tic = self._clock()

    def process(self, x):
        self.count += x if isinstance(x, (int, np.integer)) else len(x)
        if not self.use_json:
            toc = self._clock()
            if toc - self.tic > self.report_period:
                sps = self.count / (toc - self.tic)
                s, sp = _normalize(sps)
                b, bp = _normalize(self.elem_size * sps)
                self._fh.write("[%s] %.2f %sS/s (%.2f %sB/s)\n" % (self.title, s, sp, b, bp))
                self._fh.flush()
                self.tic, self.count = toc, 0

    def cleanup(self):
        if self.use_json:
            toc = self._clock()
            sps = self.count / (toc - self.tic)
            self._fh.write(json.dumps({"samples_per_second": sps, "bytes_per_second": self.elem_size * sps}))
        if isinstance(self.file, str):
            self._fh.close()
        else:
            self._fh.flush()


def run_trials(make_top, num_trials=5, trial_duration=1.0, sync=None):
    """benchmarks/luaradio_benchmark.lua:690-738: `num_trials` trials of `trial_duration` seconds each; every trial builds a fresh flow graph
    with make_top(results_file) -> step, where step() moves one source vector through the blocks into a BenchmarkSink(results_file, True)
    and make_top also returns the sink; the trial's JSON is read back and mean / standard deviation (population, :722-738) are reported.
    make_top(results) must return (step, sink).  sync() (e.g. lrhip_synchronize) is called before the sink's cleanup so that queued device
    work is inside the measured interval."""
    sps, bps = [], []
    for _ in range(num_trials):
        results = io.StringIO()
        step, sink = make_top(results)
        t_end = time.perf_counter() + trial_duration
        while time.perf_counter() < t_end:
            step()
        if sync is not None:
            sync()
        sink.cleanup()
        r = json.loads(results.getvalue())
        sps.append(r["samples_per_second"])
        bps.append(r["bytes_per_second"])
    mean_s, mean_b = sum(sps) / num_trials, sum(bps) / num_trials
    return {"samples_per_second": mean_s, "samples_per_second_stdev": math.sqrt(sum((v - mean_s) ** 2 for v in sps) / num_trials),
            "bytes_per_second": mean_b, "bytes_per_second_stdev": math.sqrt(sum((v - mean_b) ** 2 for v in bps) / num_trials),
            "trials": num_trials, "trial_duration": trial_duration}
