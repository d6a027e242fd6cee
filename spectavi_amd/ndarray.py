"""NdArray: ctypes mirror of include/NdArray.h (C-allocated out-parameter).

Stands in for `cndarray.ndarray.NdArray`, which the reference imports
(spectavi/feature.py:8, spectavi/mvg.py:8) from a git submodule that is empty in
the reference tree.  Same usage: `NdArray(dtype='uint64')`, pass it (or
`ct.byref` of it) where `ct.POINTER(NdArray)` is declared, `.asarray()` to read
the result back.  Memory is malloc'ed by the C side (ndarray_alloc) and released
by this object (ndarray_free).
"""
import ctypes as ct

import numpy as np

NDARRAY_MAX_DIMS = 4


class NdArray(ct.Structure):
    _fields_ = [
        ("m_data", ct.c_void_p),
        ("m_shape", ct.c_size_t * NDARRAY_MAX_DIMS),
        ("m_ndim", ct.c_int32),
        ("m_itemsize", ct.c_int32),
    ]

    def __init__(self, dtype="float64"):
        super().__init__()
        self._dtype = np.dtype(dtype)
        self.m_data = None
        self.m_ndim = 0
        self.m_itemsize = self._dtype.itemsize

    @property
    def shape(self):
        return tuple(int(self.m_shape[i]) for i in range(self.m_ndim))

    def asarray(self):
        """Copy the C buffer into a fresh numpy array of the declared dtype."""
        shape = self.shape
        count = int(np.prod(shape)) if shape else 0
        if not self.m_data or count == 0:
            return np.empty(shape, dtype=self._dtype)
        buf = (ct.c_char * (count * self._dtype.itemsize)).from_address(self.m_data)
        return np.frombuffer(buf, dtype=self._dtype, count=count).reshape(shape).copy()

    def __del__(self):
        try:
            if self.m_data:
                from spectavi_amd._lib import clib
                clib.ndarray_free(ct.byref(self))
        except Exception:
            pass
