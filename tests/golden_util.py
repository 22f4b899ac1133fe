"""Loader for the reference golden vectors committed under tests/golden/ (see make_golden.py)."""
import gzip
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _vec(v):
    """{"type","data"} -> numpy array (complex64 / float32); anything else is returned unchanged."""
    if isinstance(v, dict) and "data" in v:
        if v["type"] == "ComplexFloat32":
            a = np.array(v["data"], dtype=np.float64).reshape(-1, 2)
            return (a[:, 0] + 1j * a[:, 1]).astype(np.complex64)
        if v["type"] == "Float32":
            return np.array(v["data"], dtype=np.float64).astype(np.float32)
        return np.array(v["data"])
    if isinstance(v, dict) and "scalar" in v:
        if v["type"] == "ComplexFloat32":
            return np.complex64(complex(v["scalar"][0], v["scalar"][1]))
        return np.float32(v["scalar"][0])
    if isinstance(v, dict) and v.get("type") == "bytes":
        return bytes.fromhex(v["hex"])
    return v


def load(name):
    with gzip.open(os.path.join(GOLDEN_DIR, name + ".json.gz")) as f:
        doc = json.load(f)
    if doc["kind"] == "block":
        eps = doc["epsilon"]
        # composite epsilons are Lua expressions "(liquid and not volk) and 1e-3 or 1e-5": take the non-liquid value
        doc["epsilon"] = float(eps.split(" or ")[-1]) if " or " in eps else float(eps)
        for v in doc["vectors"]:
            v["args"] = [_vec(a) for a in v["args"]]
            v["inputs"] = [_vec(a) for a in v["inputs"]]
            v["outputs"] = [_vec(a) for a in v["outputs"]]
    else:
        doc["values"] = {k: _vec(v) for k, v in doc["values"].items()}
    return doc


def max_abs_err(a, b):
    """jigs.assert_vector_equal metric: |x-y| per sample (complex modulus for ComplexFloat32)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a.astype(np.complex128) - b.astype(np.complex128))))


def run_whole_and_samplewise(make_block, x):
    """The two modes of tests/jigs.lua:191-250: whole vector in one process() call, and one sample per call."""
    whole = make_block().process(x)
    blk = make_block()
    parts = [blk.process(x[i:i + 1]) for i in range(len(x))]
    parts = [p for p in parts if len(p)]
    samplewise = np.concatenate(parts) if parts else whole[:0]
    return whole, samplewise
