"""CPU-side tests of the product's host logic and of the C-ABI surface (no compute calls: no GPU here)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import luaradio_amd as lr
from luaradio_amd import _lib, filter_utils, types, window_utils
from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "lrhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lrhip_\w+)\s*\(", text)))


def test_abi_exports_every_declared_symbol():
    """liblrhip.so loads and exports exactly what include/lrhip.h declares (and _lib.py binds all of it)."""
    L = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (lrhip_\w+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(_lib.SIGNATURES), sorted(set(declared) ^ set(_lib.SIGNATURES))
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.lrhip_version()


def test_library_is_built_for_gfx950_only():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + _lib.LIB_PATH],
                         capture_output=True, text=True).stdout
    if out.strip():
        gpu_targets = [l for l in out.split() if "amdgcn" in l]
        assert gpu_targets and all("gfx950" in t for t in gpu_targets), gpu_targets


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_product_fails_loudly_without_gpu():
    """No CPU fallback: creating any stage without a device is an error with a message."""
    L = _lib.load()
    blk = lr.FrequencyTranslatorBlock(0.2)
    blk.rate = 2.0
    blk.differentiate([types.ComplexFloat32])
    with pytest.raises(lr.LrhipError) as ei:
        blk.initialize()
    assert "rotator" in str(ei.value) and L.lrhip_strerror()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "luaradio_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".lua")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liblroracle" not in text, f


def test_window_utils_mirror_matches_reference_vectors():
    vals = G.load("window_utils_vectors")["values"]
    for name, want in vals.items():
        kind = name[len("window_"):]
        periodic = kind.endswith("_periodic")
        kind = kind[:-len("_periodic")] if periodic else kind
        got = types.Float32.vector_from_array(window_utils.window(128, kind, periodic))
        assert G.max_abs_err(got, want) < 1e-6, name
    with pytest.raises(ValueError):
        window_utils.window(8, "kaiser")


def test_filter_utils_mirror_matches_reference_vectors():
    vals = G.load("filter_utils_vectors")["values"]
    f32 = types.Float32.vector_from_array
    assert G.max_abs_err(f32(filter_utils.firwin_lowpass(128, 0.5)), vals["firwin_lowpass"]) < 1e-6
    assert G.max_abs_err(f32(filter_utils.firwin_highpass(129, 0.5)), vals["firwin_highpass"]) < 1e-6
    assert G.max_abs_err(f32(filter_utils.firwin_bandpass(129, [0.4, 0.6])), vals["firwin_bandpass"]) < 1e-6
    assert G.max_abs_err(f32(filter_utils.firwin_bandstop(129, [0.4, 0.6])), vals["firwin_bandstop"]) < 1e-6


def test_filter_utils_complex_rrc_hilbert_match_reference_vectors():
    vals = G.load("filter_utils_vectors")["values"]      # args: tests/utilities/filter_utils_spec.lua:29-57
    c64 = types.ComplexFloat32.vector_from_array
    f32 = types.Float32.vector_from_array
    for cut, key in (([0.1, 0.3], "positive"), ([-0.1, -0.3], "negative"), ([-0.2, 0.2], "zero")):
        assert G.max_abs_err(c64(filter_utils.firwin_complex_bandpass(129, cut)), vals["firwin_complex_bandpass_" + key]) < 1e-6
        assert G.max_abs_err(c64(filter_utils.firwin_complex_bandstop(129, cut)), vals["firwin_complex_bandstop_" + key]) < 1e-6
    assert G.max_abs_err(f32(filter_utils.fir_root_raised_cosine(101, 1e6, 0.5, 1e3)), vals["fir_root_raised_cosine"]) < 1e-6
    assert G.max_abs_err(f32(filter_utils.fir_hilbert_transform(129)), vals["fir_hilbert_transform"]) < 1e-6
    with pytest.raises(ValueError):
        filter_utils.fir_hilbert_transform(128)


def test_filter_utils_mirror_is_bit_identical_to_oracle_design():
    """two independent restatements (C double, Python double) of filter_utils.lua give the same Float32 taps"""
    from oracle import oracle as O
    for M, c, w in [(128, 15e3 / 110250, "hamming"), (128, 0.2, "bartlett"), (16, 0.2, "hamming"), (33, 0.7, "blackman")]:
        a = types.Float32.vector_from_array(filter_utils.firwin_lowpass(M, c, w))
        b = O.firwin_lowpass(M, c, w).astype(np.float32)
        assert np.array_equal(a, b), (M, c, w)


def test_block_type_signatures_and_rates():
    b = lr.FIRFilterBlock([1.0, 2.0])
    b.differentiate([types.Float32])
    assert b.get_output_type() is types.Float32
    b.differentiate([types.ComplexFloat32])
    assert b.get_output_type() is types.ComplexFloat32
    c = lr.FIRFilterBlock(np.array([1 + 1j], dtype=np.complex64))
    with pytest.raises(TypeError):
        c.differentiate([types.Float32])          # complex taps need complex input (firfilter.lua:68-70)
    d = lr.DownsamplerBlock(5)
    d.rate = 1102500.0
    assert d.get_rate() == 220500.0                # downsampler.lua:36-38
    t = lr.TunerBlock(-250e3, 200e3, 5)
    t.rate = 1102500.0
    t.differentiate([types.ComplexFloat32])
    t._propagate_rates()
    assert t.get_rate() == 220500.0
    with pytest.raises(AssertionError):
        lr.LowpassFilterBlock(128, None)
    with pytest.raises(RuntimeError):
        lr.FrequencyTranslatorBlock(1.0).get_rate()


def test_lua_glue_declares_the_same_abi():
    """the ffi.cdef in lua/radio/core/lrhip.lua names every function of include/lrhip.h - except the ones a LuaJIT host has no use for, which
    tests/test_lua_glue.py lists with their reasons (NOT_FOR_LUA: the PyTorch stream hook, the HIP-event timers of the measurement harness, ...)"""
    from tests.test_lua_glue import NOT_FOR_LUA
    path = os.path.join(ROOT, "lua", "radio", "core", "lrhip.lua")
    text = open(path).read()
    cdef = text[text.index("ffi.cdef[["):text.index("]]", text.index("ffi.cdef[["))]
    for name in _header_functions():
        assert (name + "(" in cdef) != (name in NOT_FOR_LUA), name


def test_device_graph_construction_errors_need_no_gpu():
    """DeviceGraph mirrors CompositeBlock:connect (composite.lua:140-330): bad port names, double connections, missing
    inputs and feedback loops are rejected before any device object is created"""
    import pytest
    import luaradio_amd as lr
    from luaradio_amd import types
    g = lr.DeviceGraph()
    a = g.input("a", types.ComplexFloat32, 1e6)
    mc = lr.MultiplyConjugateBlock()
    g.connect(a, "out", mc, "nope")
    with pytest.raises(KeyError):
        g.initialize()
    g = lr.DeviceGraph()
    a = g.input("a", types.ComplexFloat32, 1e6)
    mc = lr.MultiplyConjugateBlock()
    g.connect(a, "out", mc, "in1")
    with pytest.raises(ValueError, match="unconnected"):
        g.initialize()
    g = lr.DeviceGraph()
    b1, b2 = lr.ComplexConjugateBlock(), lr.ComplexConjugateBlock()
    g.connect(b1, b2)
    g.connect(b2, b1)
    with pytest.raises(ValueError, match="cycle"):
        g.initialize()


def test_file_source_shorter_than_one_record_ends_even_with_repeat():
    """repeat_on_eof on an empty / truncated file recursed forever; the reference returns nil when the read after the rewind
    still yields nothing (iqfile.lua:86-96)"""
    import luaradio_amd as lr
    for blob in (b"", b"\x01"):
        src = lr.IQFileSource(blob, "s16le", 1e6, True)
        src._fh = __import__("io").BytesIO(blob)       # initialize() needs the device for its format stage; the read loop does not
        src._stage = None
        assert src.process() is None


def test_composite_checks_its_own_type_signatures():
    import luaradio_amd as lr
    from luaradio_amd import types
    import pytest
    t = lr.TunerBlock(1e3, 1e3, 5)
    t.rate = 1e6
    with pytest.raises(TypeError, match="No compatible type signatures"):
        t.differentiate([types.Float32])


def test_bench_recording_is_the_same_whichever_partition_asks():
    """bench.py --workload timeshard: every rank generates only its own partition (+ halo) of ONE synthetic FM recording - closed-form phase, noise from
    one generator per block of 2^22 samples - so the partitions must tile the recording exactly, wherever they are cut"""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    blk = 1 << 12
    a, b = 3 * blk - 100, 6 * blk + 77
    whole = bench.fm_recording(torch, "cpu", a, b, block=blk)
    cuts = [a, a + 1, 4 * blk, 4 * blk + 5, 5 * blk - 1, b]
    parts = torch.cat([bench.fm_recording(torch, "cpu", lo, hi, block=blk) for lo, hi in zip(cuts, cuts[1:])])
    assert whole.shape == (2 * (b - a),) and torch.equal(whole, parts)
    z = whole.view(-1, 2)
    mag = torch.sqrt(z[:, 0] ** 2 + z[:, 1] ** 2)
    assert float(mag.min()) > 0.97 and float(mag.max()) < 1.03          # unit carrier + 1 % noise


def test_fir_mode_table_matches_the_lua_glue():
    """one use_fft table for every front end: luaradio_amd/block.py fir_mode == lrhip.fir_mode in lua/radio/core/lrhip.lua"""
    import re
    from luaradio_amd import block
    lua = open(os.path.join(ROOT, "lua", "radio", "core", "lrhip.lua")).read()
    body = lua[lua.index("function M.fir_mode(use_fft)"):]
    body = body[:body.index("\nend\n")]
    assert re.search(r"package\.loaded\['tests\.jigs'\]\s*then\s*return 0", body) and re.search(r"use_fft == nil.*?return 3", body, re.S)
    assert 'use_fft == "auto" then return 3' in body and 'use_fft == "fast" then return 2' in body and "return use_fft and 1 or 0" in body
    saved, block.TESTS_JIGS_LOADED = block.TESTS_JIGS_LOADED, False
    try:
        assert [block.fir_mode(v) for v in (None, "auto", "fast", True, False)] == [3, 3, 2, 1, 0]
        block.TESTS_JIGS_LOADED = True
        assert block.fir_mode(None) == 0
    finally:
        block.TESTS_JIGS_LOADED = saved


def _bench_plain(*args):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env), env


def test_bench_gpus_n_launches_n_ranks_by_itself():
    """`--gpus` was parsed and dropped.  `python bench.py --gpus 2` with no launcher environment re-executes itself under
    torch.distributed.run with two ranks (the reference forks its own processes, radio/core/composite.lua:568-569); --launch-check makes every
    rank report and exit before it needs a device, so this runs on a box without GPUs"""
    import json
    r, env = _bench_plain("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    seen = sorted((d["rank"], d["local_rank"], d["world"], d["gpus"]) for d in (json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")))
    assert seen == [(0, 0, 2, 2), (1, 1, 2, 2)]
    # one rank: no launcher in between
    r, _ = _bench_plain("--gpus", "1", "--launch-check")
    assert r.returncode == 0 and json.loads(r.stdout.strip())["world"] == 1


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    r, env = _bench_plain("--gpus", "1", "--launch-check")
    env = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_without_enough_devices_exits_nonzero():
    import torch
    if torch.cuda.is_available():
        pytest.skip("box with GPUs: covered by tests/test_gpu_bench.py")
    r, _ = _bench_plain("--gpus", "8", "--steps", "1")
    assert r.returncode != 0 and "8 asked for, 0 device(s) visible" in r.stderr and "n_gpus" not in r.stdout


def test_fft_building_blocks_run_on_the_cpu(tmp_path):
    """radix4 / dft16 / dft64 and the complex helpers of pk_math.h are __host__ __device__ (the device pass compiles the packed-f32 bodies, the host pass
    the scalar ones), so the index algebra of the one-wave-per-block overlap-save kernel (kernels_firfft64.h: 4096 = 64 x 64, two in-register 64-point
    transforms, one transpose, the D x C twiddle) is checked HERE, without a GPU, against double-precision DFTs: tools/host_fft_check.hip"""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "host_fft_check")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-function", "-I", os.path.join(ROOT, "luaradio_amd", "csrc"),
                        "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tools", "host_fft_check.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
