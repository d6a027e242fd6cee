"""Prints the sustained v_sad_hi_u8 rate of the attached GPU (roofline constant)."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectavi_amd._lib import clib, check  # noqa: E402

clib.spv_microbench_sad.restype = ct.c_int
clib.spv_microbench_sad.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_double)]
for blocks in (256, 512, 1024, 2048, 4096):
    r = ct.c_double(0)
    check(clib.spv_microbench_sad(blocks, 20000, ct.byref(r)))
    print("blocks=%5d  %.3f Tlane-op/s  (%.1f %% of 256 CU x 128 lanes x 2.4 GHz)" % (
        blocks, r.value / 1e12, 100 * r.value / (256 * 128 * 2.4e9)))
