"""Time-axis sharding (luaradio_amd/timeshard.py, include/lrhip.h lrhip_chain_seek / lrhip_chain_halo).

CPU: the partition plan, and the one-process-per-GPU driver under gloo with world size 2 (the chain is a stand-in with the same
seek / halo / process contract built on the oracle's FIR, because the product has no CPU compute path).
GPU: G in {2, 4, 8} virtual partitions of one stream on one device against the uninterrupted run - bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from luaradio_amd import timeshard


def test_bounds_cover_the_stream_on_aligned_cuts():
    for n, parts, align in ((1 << 20, 8, 65536), (1000003, 4, 65536), (70000, 8, 65536), (0, 3, 16), (12345, 1, 64), (4096 * 9 + 5, 2, 4096)):
        b = timeshard.bounds(n, parts, align)
        assert len(b) == parts and b[0][0] == 0 and b[-1][1] == n
        for (a0, b0), (a1, b1) in zip(b, b[1:]):
            assert b0 == a1 and a0 <= b0
        for a, _ in b[1:]:
            assert a % align == 0 or a == n
    assert timeshard.bounds(1 << 20, 4, 65536) == [(0, 262144), (262144, 524288), (524288, 786432), (786432, 1048576)]
    assert timeshard.replay_start(1000, 127) == 873 and timeshard.replay_start(100, 127) == 0 and timeshard.replay_start(1000, 127, 25) == 850
    with pytest.raises(ValueError):
        timeshard.bounds(10, 0)


class _OracleFirChain:
    """seek / halo / process with the contract of luaradio_amd.Chain, on the oracle's FIR + downsampler"""

    def __init__(self, taps, decim):
        self.taps, self.decim = np.asarray(taps, np.float32), decim
        self.seek(0)

    def halo(self):
        return len(self.taps) - 1

    def shard_align(self):
        return 1

    def seek(self, n0):
        from oracle import oracle as O
        self.fir = O.FIR(self.taps, True, O.MODE_FMA)
        self.phase = (-n0) % self.decim

    def process(self, x):
        y = self.fir.process(np.asarray(x, np.complex64))
        out = y[self.phase::self.decim]
        self.phase = (self.phase - len(y)) % self.decim
        return out


def test_run_partition_equals_the_uninterrupted_stream_cpu_stand_in():
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n = 50000
    x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    taps = O.firwin_lowpass(64, 0.2).astype(np.float32)
    whole = _OracleFirChain(taps, 5).process(x)
    for parts in (2, 3, 8):
        got = np.concatenate([timeshard.run_partition(_OracleFirChain(taps, 5), x, a, b) for a, b in timeshard.bounds(n, parts, 4096)])
        assert np.array_equal(got, whole)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    n = 40000
    x = (np.random.default_rng(11).uniform(-1, 1, n) + 1j * np.random.default_rng(12).uniform(-1, 1, n)).astype(np.complex64)   # the recording every rank can read
    taps = O.firwin_lowpass(64, 0.2).astype(np.float32)
    a, b = timeshard.rank_partition(n, world, rank, 4096)
    mine = timeshard.run_partition(_OracleFirChain(taps, 5), x, a, b)
    # no collective on the data path; the counts are gathered only to check the partition sizes add up
    cnt = torch.tensor([len(mine)], dtype=torch.int64)
    allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, cnt)
    np.save(os.path.join(outdir, "part%d.npy" % rank), mine)
    if rank == 0:
        np.save(os.path.join(outdir, "counts.npy"), np.array([int(c.item()) for c in allc]))
    dist.barrier()
    dist.destroy_process_group()


def test_one_partition_per_rank_under_gloo(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n = 40000
    x = (np.random.default_rng(11).uniform(-1, 1, n) + 1j * np.random.default_rng(12).uniform(-1, 1, n)).astype(np.complex64)
    whole = _OracleFirChain(O.firwin_lowpass(64, 0.2).astype(np.float32), 5).process(x)
    parts = [np.load(os.path.join(str(tmp_path), "part%d.npy" % r)) for r in range(world)]
    assert list(np.load(os.path.join(str(tmp_path), "counts.npy"))) == [len(p) for p in parts]
    assert np.array_equal(np.concatenate(parts), whole)


# ------------------------------------------------------------------------------------------------------------ GPU
def _rand_c(seed, n):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def _chains():
    import luaradio_amd as lr
    from luaradio_amd import types
    rate = 1102500.0

    def init(blocks, t=types.ComplexFloat32):
        r = rate
        for b in blocks:
            b.rate = r
            b.differentiate([t])
            b.initialize()
            r, t = b.get_rate(), b.get_output_type()
        return lr.Chain(blocks)

    return {
        "tuner": lambda: init([lr.FrequencyTranslatorBlock(-250e3), lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5)]),
        "tuner+disc": lambda: init([lr.FrequencyTranslatorBlock(123456.0), lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5), lr.FrequencyDiscriminatorBlock(1.25)]),
        "lowpass": lambda: init([lr.LowpassFilterBlock(128, 50e3)]),
        "translate+downsample": lambda: init([lr.FrequencyTranslatorBlock(-1e5), lr.DownsamplerBlock(7)]),
        "wbfm": lambda: lr.wbfm_mono_receiver(rate, -250e3),
        "wbfm-two-launch": lambda: lr.Chain(lr.wbfm_mono_receiver(rate, -250e3)._blocks, lr._lib.CHAIN_NO_SINGLE_LAUNCH),
        "deemphasis": lambda: init([lr.ComplexToRealBlock(), lr.FMDeemphasisFilterBlock(75e-6)]),
        "disc+lowpass+deemphasis": lambda: init([lr.FrequencyDiscriminatorBlock(1.25), lr.FIRFilterBlock(O_taps(), "fast"), lr.FMDeemphasisFilterBlock(75e-6)]),
    }


def O_taps():
    from oracle import oracle as O
    return O.firwin_lowpass(128, 0.2).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tuner", "tuner+disc", "lowpass", "translate+downsample", "wbfm", "wbfm-two-launch", "deemphasis", "disc+lowpass+deemphasis"])
def test_virtual_partitions_on_one_device_bit_equal_to_the_single_stream(name):
    """G in {2, 4, 8} partitions of one stream, each started with lrhip_chain_seek + the chain's halo, concatenated ==
    the uninterrupted run, bit for bit (boundaries on multiples of the chain's shard_align())"""
    n = (1 << 21) + 12345
    x = _rand_c(21, n)
    build = _chains()[name]
    whole = build().process(x)
    for parts in (2, 4, 8):
        chain = build()                      # one chain object re-used for every partition: seek() must clear everything
        h = chain.halo()
        assert 0 < h < 200000 or name == "translate+downsample"
        align = chain.shard_align()
        # tuner + discriminator: the tile grid of the window-relative rotator staging (1024 outputs x 5); the WBFM receiver = lcm with its tail's 64000
        assert align == {"tuner+disc": 5120, "wbfm": 128000, "wbfm-two-launch": 128000, "deemphasis": 4096, "disc+lowpass+deemphasis": 4096 * 7}.get(name, 1)
        got = np.concatenate([timeshard.run_partition(chain, x, a, b) for a, b in timeshard.bounds(n, parts, 4096 * align if align == 1 else align)])
        assert len(got) == len(whole), (name, parts)
        if name == "disc+lowpass+deemphasis":
            # overlap-save arithmetic: the first block of a chunk takes its history from the carried buffer instead of the stream, which
            # rounds differently - same values to Float32 rounding (the block's own parity bar is 1e-6)
            assert float(np.max(np.abs(got - whole))) < 1e-7, (name, parts)
        elif name == "wbfm":
            # the single-launch receiver (kernels_rx.h) deals its tiles out over one round of workgroups, and a run that does not start the
            # chunk warms its recurrence up from zero over 75 audio samples (q^75 = 1.4e-10) instead of carrying state between workgroups:
            # where the runs fall depends on the chunk length, so a partition agrees with the single stream to that warm-up, not bit for
            # bit (tuner, discriminator and audio filter do - their tiles are pure functions of their windows); the two-launch form below does
            assert float(np.max(np.abs(got - whole))) < 1e-7, (name, parts)
        else:
            assert np.array_equal(got, whole), (name, parts, float(np.max(np.abs(got - whole))))


@pytest.mark.gpu
def test_unaligned_partition_boundaries_same_values():
    """any boundary: filters / rotators / downsamplers exactly; the tuner in front of a discriminator and the recurrences of the WBFM tail to
    Float32 rounding (the angle of a small filter output is ill-conditioned: median and a robust maximum)"""
    n = 1 << 20
    x = _rand_c(22, n)
    for name, tol in (("tuner", 0.0), ("tuner+disc", 1e-3), ("wbfm", 5e-5)):
        build = _chains()[name]
        whole = build().process(x)
        chain = build()
        cuts = [0, 100001, 333333, 700007, n]
        got = np.concatenate([timeshard.run_partition(chain, x, a, b) for a, b in zip(cuts, cuts[1:])])
        assert len(got) == len(whole)
        d = np.abs(got - whole)
        d = np.minimum(d, np.abs(2 * np.pi / 1.25 - d)) if name == "tuner+disc" else d      # a whole turn is no difference
        assert float(np.max(d)) <= tol, name
        assert float(np.median(d)) <= min(tol, 1e-7), name


@pytest.mark.gpu
def test_unaligned_partition_boundaries_on_an_fm_signal_are_tight():
    """the same cuts on what the receiver is for - an FM signal, |tuner output| ~ 1: the window-relative rotator staging then moves the angles by
    Float32 rounding of well-conditioned filter outputs, and the bounds are 2e-6 (tuner + discriminator) and 2e-7 (receiver audio); the U(-1, 1)
    noise input of the test above is the documented worst case (small filter outputs, ill-conditioned angles)"""
    n = 1 << 20
    fs = 1102500.0
    rng = np.random.default_rng(23)
    t = np.arange(n) / fs
    m = 0.5 * np.sin(2 * np.pi * 1e3 * t) + 0.5 * np.sin(2 * np.pi * 5e3 * t)
    x = (np.exp(1j * (2 * np.pi * 250e3 * t + 2 * np.pi * 75e3 / fs * np.cumsum(m))) + 0.01 * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))).astype(np.complex64)
    import luaradio_amd as lr
    from luaradio_amd import types

    def tuner_disc():
        blocks = [lr.FrequencyTranslatorBlock(-250e3), lr.LowpassFilterBlock(128, 100e3), lr.DownsamplerBlock(5), lr.FrequencyDiscriminatorBlock(1.25)]
        r, ty = fs, types.ComplexFloat32
        for b in blocks:
            b.rate = r
            b.differentiate([ty])
            b.initialize()
            r, ty = b.get_rate(), b.get_output_type()
        return lr.Chain(blocks)

    for name, build, tol in (("tuner+disc", tuner_disc, 2e-6), ("wbfm", lambda: lr.wbfm_mono_receiver(fs, -250e3), 2e-7)):
        whole = build().process(x)
        chain = build()
        cuts = [0, 100001, 333333, 700007, n]
        got = np.concatenate([timeshard.run_partition(chain, x, a, b) for a, b in zip(cuts, cuts[1:])])
        assert len(got) == len(whole)
        assert float(np.max(np.abs(got - whole))) <= tol, (name, float(np.max(np.abs(got - whole))))


@pytest.mark.gpu
def test_halo_refuses_chains_with_unbounded_memory():
    import luaradio_amd as lr
    from luaradio_amd import types
    agc = lr.AGCBlock("slow")
    agc.rate = 48000.0
    agc.differentiate([types.Float32])
    agc.initialize()
    chain = lr.Chain([agc])
    with pytest.raises(lr.LrhipError, match="unbounded memory"):
        chain.halo()


@pytest.mark.gpu
def test_example_timeshard_wbfm_selftest_and_file_mode(tmp_path):
    """examples/timeshard_wbfm.py: --selftest, and a ComplexFloat32 recording cut into 3 partitions == the receiver on the whole file"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex = os.path.join(root, "examples", "timeshard_wbfm.py")
    r = subprocess.run([sys.executable, ex, "--selftest"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr[-2000:]
    import luaradio_amd as lr
    x = _rand_c(5, 300000)
    rec, out = tmp_path / "rec.cf32", tmp_path / "audio.f32"
    x.tofile(rec)
    r = subprocess.run([sys.executable, ex, str(rec), str(out), "--parts", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = lr.wbfm_mono_receiver(1102500.0, -250e3).process(x)
    got = np.fromfile(out, np.float32)
    assert len(got) == len(want) and float(np.max(np.abs(got - want))) < 1e-7      # single-launch receiver: equal to its recurrence warm-up (1e-10)
