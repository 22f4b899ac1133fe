"""ctypes face of the interprocess primitives of include/lrhip.h (lrhip_ipc_* / lrhip_peer_copy): what a one-process-per-GPU host
(LuaRadio forks a process per block, radio/core/composite.lua:569) uses to hand a source's slabs to the branch processes' devices
without a host round trip.  Handles are 64 opaque bytes that travel over whatever control channel the host has (a pipe here)."""
import ctypes as C

from . import _lib

HANDLE_BYTES = 64


def export_memory(dev_ptr):
    h = C.create_string_buffer(HANDLE_BYTES)
    _lib.check(_lib.load().lrhip_ipc_export(dev_ptr, h), "ipc_export")
    return h.raw


def open_memory(handle):
    return _lib.check_ptr(_lib.load().lrhip_ipc_open(C.create_string_buffer(handle, HANDLE_BYTES)), "ipc_open")


def close_memory(dev_ptr):
    _lib.check(_lib.load().lrhip_ipc_close(dev_ptr), "ipc_close")


class Event:
    """an interprocess HIP event: create() in one process, open(handle) in the other"""

    def __init__(self, ptr, handle=None):
        self._e, self.handle = ptr, handle

    @classmethod
    def create(cls):
        h = C.create_string_buffer(HANDLE_BYTES)
        return cls(_lib.check_ptr(_lib.load().lrhip_ipc_event_create(h), "ipc_event_create"), h.raw)

    @classmethod
    def open(cls, handle):
        return cls(_lib.check_ptr(_lib.load().lrhip_ipc_event_open(C.create_string_buffer(handle, HANDLE_BYTES)), "ipc_event_open"), handle)

    def record(self, on_copy_stream=False):
        _lib.check(_lib.load().lrhip_ipc_event_record(self._e, int(on_copy_stream)), "ipc_event_record")

    def wait(self, on_copy_stream=False):
        """make the stream wait (on the GPU); the host does not block"""
        _lib.check(_lib.load().lrhip_ipc_event_wait(self._e, int(on_copy_stream)), "ipc_event_wait")

    def query(self):
        return bool(_lib.check(_lib.load().lrhip_ipc_event_query(self._e), "ipc_event_query"))

    def synchronize(self):
        _lib.check(_lib.load().lrhip_ipc_event_synchronize(self._e), "ipc_event_synchronize")

    def destroy(self):
        if self._e:
            _lib.load().lrhip_ipc_event_destroy(self._e)
            self._e = None


def peer_copy(dst_ptr, dst_device, src_ptr, src_device, nbytes):
    _lib.check(_lib.load().lrhip_peer_copy(dst_ptr, dst_device, src_ptr, src_device, nbytes), "peer_copy")


def copy_stream_synchronize():
    _lib.check(_lib.load().lrhip_copy_stream_synchronize(), "copy_stream_synchronize")
