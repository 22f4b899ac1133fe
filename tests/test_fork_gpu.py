"""The reference's process model against the real library: the library is loaded in the PARENT, the parent fork()s one child per block
AFTER that, each child closes every descriptor except its own pipes (radio/core/composite.lua:568-611) and only then makes its first device call.  Until
round 5 every multi-process GPU test here used mp.get_context("spawn"); tests/helpers/fork_model.py does what LuaRadio does, in a fresh interpreter."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def run_shape(shape, tmp_path):
    out = tmp_path / (shape + ".npz")
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "fork_model.py"), shape, str(out)], env=env, timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-2000:]
    return np.load(out)


def test_a_forked_block_process_runs_a_stand_alone_stage(tmp_path):
    from oracle import oracle as O
    from tests.helpers.fork_model import FS, stream
    r = run_shape("stage", tmp_path)
    assert list(r["codes"]) == [0], (r["codes"], r["messages"])
    want = O.lowpass(128, 100e3, FS, True, mode=O.MODE_FMA).process(stream())
    assert np.array_equal(r["y"], want)                              # direct form: the bits of the fmaf chain


def test_a_forked_block_process_runs_the_device_chain_call_sequence(tmp_path):
    from oracle import oracle as O
    from tests.helpers.fork_model import FS, stream
    r = run_shape("chain", tmp_path)
    assert list(r["codes"]) == [0], (r["codes"], r["messages"])
    want = O.tuner(-250e3, 100e3, 5, FS, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(stream())
    assert len(r["y"]) == len(want) and float(np.max(np.abs(r["y"] - want))) < 2e-6


def test_fan_out_head_and_three_branches_started_by_fork(tmp_path):
    """socket pairs created before fork() (DeviceFanoutBlock:initialize), one fork() per block, the parent closes its copies of every end"""
    from oracle import oracle as O
    from tests.helpers.fork_model import FS, N, OFFSETS, stream
    r = run_shape("fanout", tmp_path)
    assert list(r["codes"]) == [0, 0, 0, 0], (r["codes"], r["messages"])
    slabs, copies = [int(v) for v in r["head"]]
    assert slabs == (N + 65535) // 65536 and copies == 3 * slabs
    x = stream()
    for k, off in enumerate(OFFSETS):
        want = O.tuner(off, 100e3, 5, FS, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(x)
        y = r["y%d" % k]
        assert len(y) == len(want) == N // 5 and float(np.max(np.abs(y - want))) < 2e-6, k


def test_a_child_forked_after_the_parent_touched_the_device_fails_with_a_message(tmp_path):
    """the mistake the lazy create_stage() of every device block exists to avoid: lrhip_init in the parent, THEN fork.  The child must get an error
    through lrhip_strerror - not a hang, not a crash"""
    r = run_shape("init_then_fork", tmp_path)
    assert list(r["codes"]) == [0]
    rc, count, msg = str(r["messages"][0]).split("|", 2)
    assert int(rc) != 0 and int(count) < 0
    assert "before fork()" in msg and "first process()" in msg


def test_partition_helpers_in_the_parent_then_a_forked_block_process(tmp_path):
    """DeviceChainBlock:halo() / shard_align() / start_at() are called in the flow graph's parent (a time-partitioned graph positions its source
    there), top:run() forks the block processes afterwards.  The glue asks a fork()ed helper process, so the parent never owns a device and the block
    process - forked later, building its own chain - starts where start_at() said: its output is the uninterrupted run's from that sample on"""
    from oracle import oracle as O
    from tests.helpers.fork_model import FS, PARTITION_FIRST, stream
    r = run_shape("lua_partition", tmp_path)
    assert list(r["codes"]) == [0], (r["codes"], r["messages"])
    halo, align, seek, parent_device = [int(v) for v in r["answers"]]
    assert parent_device == -1
    assert halo >= 127 and align >= 1 and seek % align == 0 and seek <= PARTITION_FIRST - halo < seek + align + halo
    want = O.tuner(-250e3, 100e3, 5, FS, mode=O.MODE_FMA, rot_mode=O.MODE_F64).process(stream())[PARTITION_FIRST // 5:]
    assert len(r["y"]) == len(want) and float(np.max(np.abs(r["y"] - want))) < 2e-6
