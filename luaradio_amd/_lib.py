"""ctypes binding of liblrhip.so - exactly the declarations a LuaJIT ffi.cdef of include/lrhip.h makes.

The product path FAILS LOUDLY when the HIP library is missing or no GPU is visible: there is no CPU
fallback anywhere in this package (the CPU oracle under oracle/ is test infrastructure and is never
imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LRHIP_LIB_PATH") or os.path.join(_HERE, "liblrhip.so")   # override: kernel A/B builds


class LrhipError(RuntimeError):
    pass


_lib = None

_vp, _ul, _fp = C.c_void_p, C.c_ulong, C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/lrhip.h one to one
SIGNATURES = {
    "lrhip_init": (C.c_int, [C.c_int]),
    "lrhip_strerror": (C.c_char_p, []),
    "lrhip_device": (C.c_int, []),
    "lrhip_device_count": (C.c_int, []),
    "lrhip_set_stream": (C.c_int, [_vp]),
    "lrhip_synchronize": (C.c_int, []),
    "lrhip_version": (C.c_char_p, []),
    "lrhip_fir_create": (_vp, [_fp, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_int]),
    "lrhip_rotator_create": (_vp, [C.c_double]),
    "lrhip_downsampler_create": (_vp, [C.c_uint, C.c_int]),
    "lrhip_fmdiscrim_create": (_vp, [C.c_double]),
    "lrhip_iir_create": (_vp, [_fp, C.c_uint, _fp, C.c_uint, C.c_int]),
    "lrhip_psd_create": (_vp, [C.c_uint, _fp, C.c_double, C.c_int, C.c_int, C.c_int]),
    "lrhip_dft_create": (_vp, [C.c_uint, C.c_int, C.c_int]),
    "lrhip_format_convert_create": (_vp, [C.c_char_p, C.c_int]),
    "lrhip_format_pack_create": (_vp, [C.c_char_p, C.c_int]),
    "lrhip_binary_create": (_vp, [C.c_char_p, C.c_int]),
    "lrhip_multiply_constant_create": (_vp, [C.c_float, C.c_float, C.c_int, C.c_int]),
    "lrhip_upsampler_create": (_vp, [C.c_uint, C.c_int]),
    "lrhip_channelizer_create": (_vp, [_fp, C.c_uint, C.c_uint]),
    "lrhip_welch_create": (_vp, [C.c_uint, _fp, C.c_double, C.c_int, C.c_int, C.c_uint]),
    "lrhip_welch_read": (C.c_long, [_vp, _fp, C.c_int]),
    "lrhip_fmmod_create": (_vp, [C.c_double]),
    "lrhip_powersquelch_create": (_vp, [C.c_double, C.c_double, C.c_int]),
    "lrhip_agc_create": (_vp, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "lrhip_unary_create": (_vp, [C.c_char_p, C.c_float, C.c_float, C.c_int, C.c_int]),
    "lrhip_delay_create": (_vp, [C.c_uint, C.c_int]),
    "lrhip_hilbert_create": (_vp, [_fp, C.c_uint]),
    "lrhip_stage_execute2": (C.c_long, [_vp, _vp, _vp, _ul, _vp, _ul]),
    "lrhip_stage_execute2_device": (C.c_long, [_vp, _vp, _vp, _ul, _vp, _ul]),
    "lrhip_stage_destroy": (None, [_vp]),
    "lrhip_stage_reset": (C.c_int, [_vp]),
    "lrhip_stage_input_size": (C.c_int, [_vp]),
    "lrhip_stage_output_size": (C.c_int, [_vp]),
    "lrhip_stage_max_output": (_ul, [_vp, _ul]),
    "lrhip_stage_execute": (C.c_long, [_vp, _vp, _ul, _vp, _ul]),
    "lrhip_stage_execute_device": (C.c_long, [_vp, _vp, _ul, _vp, _ul]),
    "lrhip_chain_create": (_vp, [C.POINTER(_vp), C.c_uint]),
    "lrhip_chain_create_ex": (_vp, [C.POINTER(_vp), C.c_uint, C.c_uint]),
    "lrhip_chain_destroy": (None, [_vp]),
    "lrhip_chain_reset": (C.c_int, [_vp]),
    "lrhip_chain_max_output": (_ul, [_vp, _ul]),
    "lrhip_chain_execute": (C.c_long, [_vp, _vp, _ul, _vp, _ul]),
    "lrhip_chain_execute_device": (C.c_long, [_vp, _vp, _ul, _vp, _ul]),
    "lrhip_chain_last_launches": (C.c_int, [_vp]),
    "lrhip_chain_set_ring": (C.c_int, [_vp, C.c_uint, _ul]),
    "lrhip_chain_submit": (C.c_long, [_vp, _vp, _ul]),
    "lrhip_chain_ring_input": (_vp, [_vp]),
    "lrhip_chain_submit_fd": (C.c_long, [_vp, C.c_int, C.c_ulonglong, _ul]),
    "lrhip_chain_collect": (C.c_long, [_vp, _vp, _ul]),
    "lrhip_chain_in_flight": (C.c_int, [_vp]),
    "lrhip_chain_push": (C.c_long, [_vp, _vp, _ul, _vp, _ul]),
    "lrhip_chain_flush": (C.c_long, [_vp, _vp, _ul]),
    "lrhip_chain_push_bound": (_ul, [_vp, _ul]),
    "lrhip_chain_set_latency": (C.c_int, [_vp, C.c_double]),
    "lrhip_chain_poll": (C.c_long, [_vp, _vp, _ul]),
    "lrhip_chain_poll_due": (C.c_double, [_vp]),
    "lrhip_malloc": (_vp, [_ul]),
    "lrhip_free": (None, [_vp]),
    "lrhip_memcpy_h2d": (C.c_int, [_vp, _vp, _ul]),
    "lrhip_memcpy_d2h": (C.c_int, [_vp, _vp, _ul]),
    "lrhip_memcpy_d2d": (C.c_int, [_vp, _vp, _ul]),
    "lrhip_host_alloc": (_vp, [_ul]),
    "lrhip_host_free": (None, [_vp]),
    "lrhip_host_register": (C.c_int, [_vp, _ul]),
    "lrhip_host_unregister": (C.c_int, [_vp]),
    "lrhip_timer_create": (_vp, []),
    "lrhip_timer_destroy": (None, [_vp]),
    "lrhip_timer_start": (C.c_int, [_vp]),
    "lrhip_timer_stop": (C.c_int, [_vp]),
    "lrhip_timer_elapsed_ms": (C.c_double, [_vp]),
    "lrhip_stage_seek": (C.c_int, [_vp, C.c_ulonglong]),
    "lrhip_chain_seek": (C.c_int, [_vp, C.c_ulonglong]),
    "lrhip_chain_halo": (C.c_long, [_vp]),
    "lrhip_chain_shard_align": (_ul, [_vp]),
    "lrhip_chain_start_at": (C.c_int, [_vp, C.c_ulonglong, C.POINTER(C.c_ulonglong)]),
    "lrhip_ipc_export": (C.c_int, [_vp, _vp]),
    "lrhip_ipc_open": (_vp, [_vp]),
    "lrhip_ipc_close": (C.c_int, [_vp]),
    "lrhip_ipc_event_create": (_vp, [_vp]),
    "lrhip_ipc_event_open": (_vp, [_vp]),
    "lrhip_ipc_event_destroy": (None, [_vp]),
    "lrhip_ipc_event_record": (C.c_int, [_vp, C.c_int]),
    "lrhip_ipc_event_wait": (C.c_int, [_vp, C.c_int]),
    "lrhip_ipc_event_query": (C.c_int, [_vp]),
    "lrhip_ipc_event_synchronize": (C.c_int, [_vp]),
    "lrhip_peer_copy": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _ul]),
    "lrhip_copy_stream_synchronize": (C.c_int, []),
}


# lrhip_chain_create_ex flags (include/lrhip.h)
CHAIN_EXACT_ROTATOR, CHAIN_NO_POLYPHASE_TAIL, CHAIN_NO_FUSION, CHAIN_NO_SINGLE_LAUNCH = 1, 2, 4, 8
CHAIN_EXACT = CHAIN_EXACT_ROTATOR | CHAIN_NO_POLYPHASE_TAIL | CHAIN_NO_SINGLE_LAUNCH


def load():
    """dlopen liblrhip.so and declare every entry point. Raises LrhipError if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LrhipError("liblrhip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C luaradio_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError here == missing export
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return load().lrhip_strerror().decode()


def check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise LrhipError("%s: %s" % (what, last_error()))
    return rc


def check_ptr(p, what):
    if not p:
        raise LrhipError("%s: %s" % (what, last_error()))
    return p


def adopt_torch_stream():
    """Enqueue the library's work on torch's CURRENT stream, so that torch tensors handed to process_device() are ordered with the
    kernels both ways.  torch's default stream has handle 0, which lrhip_set_stream() reads as "the library's own stream"; HIP's name
    for that stream as an explicit handle is hipStreamLegacy = (hipStream_t)1."""
    import torch
    h = torch.cuda.current_stream().cuda_stream
    check(load().lrhip_set_stream(C.c_void_p(h if h else 1)), "lrhip_set_stream")


def init(device=-1):
    check(load().lrhip_init(device), "lrhip_init")
