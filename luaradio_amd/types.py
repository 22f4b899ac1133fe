"""Sample types - mirrors radio/types/complexfloat32.lua:19-24 and radio/types/float32.lua:17-21.

ComplexFloat32 = struct{float real, imag} (8 B interleaved) == numpy complex64;
Float32 = struct{float value} (4 B) == numpy float32.  Vectors are contiguous numpy arrays, which is the
same raw layout the reference writes on its pipes (radio/types/cstruct.lua:87-126).
"""
import numpy as np


class _SampleType:
    def __init__(self, name, dtype, size):
        self.name, self.dtype, self.size = name, np.dtype(dtype), size

    def vector(self, n=0):
        return np.zeros(n, dtype=self.dtype)

    def vector_from_array(self, arr):
        if self.dtype == np.complex64:
            a = np.asarray(arr, dtype=np.float64)
            if a.ndim == 2:       # {{re, im}, ...} as in the reference
                return (a[:, 0] + 1j * a[:, 1]).astype(np.complex64)
            return np.asarray(arr).astype(np.complex64)
        return np.asarray(arr, dtype=np.float64).astype(np.float32)

    def __repr__(self):
        return self.name


ComplexFloat32 = _SampleType("ComplexFloat32", np.complex64, 8)
Float32 = _SampleType("Float32", np.float32, 4)


def type_of(x):
    """data_type of a vector (numpy array)."""
    x = np.asarray(x)
    if x.dtype == np.complex64:
        return ComplexFloat32
    if x.dtype == np.float32:
        return Float32
    raise TypeError("Unsupported sample dtype %s (expected complex64 or float32)" % x.dtype)
