"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: branch plan, slab broadcast, per-rank branch execution
and max-over-ranks timing.  The branch executor is injected (the CPU oracle's Tuner) because the product has no
CPU compute path; on a GPU box the same FanOut object drives DeviceBranch executors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from luaradio_amd import fanout


def test_plan_and_offsets():
    assert fanout.branch_offsets(8) == [-350e3, -250e3, -150e3, -50e3, 50e3, 150e3, 250e3, 350e3]
    assert fanout.plan(8, 8) == list(range(8))
    assert fanout.plan(8, 2) == [0, 1, 0, 1, 0, 1, 0, 1]
    assert fanout.local_branches(8, 4, 3) == [3, 7]
    assert fanout.local_branches(3, 8, 5) == []
    with pytest.raises(ValueError):
        fanout.plan(0, 2)


class _OracleBranch:
    def __init__(self, offset, fs):
        from oracle import oracle as O
        self.chain = O.tuner(offset, 100e3, 5, fs, mode=O.MODE_FMA, rot_mode=O.MODE_F64)

    def process(self, slab):
        x = slab.numpy().view(np.complex64)
        return torch.from_numpy(self.chain.process(x).view(np.float32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nbranches, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs, n = 1102500.0, 6000
    offs = fanout.branch_offsets(nbranches)
    mine = {b: _OracleBranch(offs[b], fs) for b in fanout.local_branches(nbranches, world, rank)}
    fo = fanout.FanOut(dist, rank, world, nbranches, mine, src=0)
    rng = np.random.default_rng(99)
    outs = {b: [] for b in mine}
    for s in range(3):     # three slabs: the branch state (history, phase, index) carries across slabs
        if rank == 0:
            x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
            slab = torch.from_numpy(x.view(np.float32).copy())
        else:
            slab = torch.zeros(2 * n, dtype=torch.float32)
        got = fo.push(slab)
        for b, y in got.items():
            outs[b].append(y.clone())
    dt = fo.timed(lambda: None, dist.barrier)
    assert dt >= 0
    for b, parts in outs.items():
        np.save(os.path.join(outdir, "branch%d.npy" % b), torch.cat(parts).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nbranches", [2, 5])
def test_fanout_two_ranks_gloo(tmp_path, nbranches):
    from oracle import oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), nbranches, str(tmp_path)), nprocs=world, join=True)
    # single-process oracle of the same stream: every branch must equal its own Tuner over the concatenated slabs
    fs, n = 1102500.0, 6000
    rng = np.random.default_rng(99)
    x = np.concatenate([(rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) for _ in range(3)])
    for b, off in enumerate(fanout.branch_offsets(nbranches)):
        want = O.tuner(off, 100e3, 5, fs, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(x)
        got = np.load(os.path.join(str(tmp_path), "branch%d.npy" % b)).view(np.complex64)
        assert np.array_equal(got, want), b


def test_fanout_rejects_wrong_ownership():
    with pytest.raises(ValueError):
        fanout.FanOut(None, 0, 2, 4, {1: object()})


def _stream_worker(rank, world, port, nbranches, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs, n, nslabs = 1102500.0, 6000, 5
    offs = fanout.branch_offsets(nbranches)
    mine = {b: _OracleBranch(offs[b], fs) for b in fanout.local_branches(nbranches, world, rank)}
    fo = fanout.FanOut(dist, rank, world, nbranches, mine, src=0)
    rng = np.random.default_rng(99)

    def source():
        for _ in range(nslabs):
            x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
            yield torch.from_numpy(x.view(np.float32).copy())

    outs = {b: [] for b in mine}
    for got in fo.stream(source() if rank == 0 else range(nslabs), 2 * n):     # double-buffered: slab k+1 travels while slab k is processed
        for b, y in got.items():
            outs[b].append(y.clone())
    assert fo.slabs == nslabs
    for b, parts in outs.items():
        np.save(os.path.join(outdir, "sbranch%d.npy" % b), torch.cat(parts).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_fanout_double_buffered_stream_two_ranks_gloo(tmp_path):
    from oracle import oracle as O
    world, nbranches = 2, 4
    mp.spawn(_stream_worker, args=(world, _free_port(), nbranches, str(tmp_path)), nprocs=world, join=True)
    fs, n = 1102500.0, 6000
    rng = np.random.default_rng(99)
    x = np.concatenate([(rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64) for _ in range(5)])
    for b, off in enumerate(fanout.branch_offsets(nbranches)):
        want = O.tuner(off, 100e3, 5, fs, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(x)
        got = np.load(os.path.join(str(tmp_path), "sbranch%d.npy" % b)).view(np.complex64)
        assert np.array_equal(got, want), b


@pytest.mark.gpu
def test_fanout_stream_on_the_device_equals_push():
    """one process, one GPU: the double-buffered form (communication stream + events) gives the slabs' outputs in order, same bits as
    push(); the branch kernels are ordered behind the copy that fills the buffer (DeviceBranch adopts torch's current stream)"""
    import luaradio_amd as lr
    from luaradio_amd import types
    fs, n, nslabs = 1102500.0, 1 << 18, 6
    g = torch.Generator(device="cuda").manual_seed(3)
    slabs = [torch.rand(2 * n, dtype=torch.float32, device="cuda", generator=g) * 2 - 1 for _ in range(nslabs)]

    def branch():
        tun = lr.TunerBlock(-350e3, 100e3, 5)
        tun.rate = fs
        tun.differentiate([types.ComplexFloat32])
        tun.initialize()
        return fanout.DeviceBranch(tun, n)

    a = fanout.FanOut(None, 0, 1, 1, {0: branch()})
    want = torch.cat([a.push(s)[0].clone() for s in slabs])
    b = fanout.FanOut(None, 0, 1, 1, {0: branch()})
    got = torch.cat([o[0].clone() for o in b.stream(iter(slabs), 2 * n)])
    torch.cuda.synchronize()
    assert torch.equal(got, want)
