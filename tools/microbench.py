"""Prints sustained VALU issue rates of the attached GPU (roofline constants for DESIGN.md)."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectavi_amd._lib import clib, check  # noqa: E402

clib.spv_microbench_valu.restype = ct.c_int
clib.spv_microbench_valu.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_double), ct.POINTER(ct.c_double)]
NAMES = ["v_sad_hi_u8", "v_sad_u8", "v_sad_u16", "v_xor+v_add (2 ops)", "v_fma_f32", "v_dot4_u32_u8", "v_med3_u32"]
for op, name in enumerate(NAMES):
    for blocks in (256, 1024, 4096):
        r, c = ct.c_double(0), ct.c_double(0)
        check(clib.spv_microbench_valu(op, blocks, 20000, ct.byref(r), ct.byref(c)))
        cyc = 256 * 4 * c.value * 1e9 / (r.value / 64.0)  # SIMD-cycles per wave64 instruction at that clock
        print("%-20s blocks=%5d  %7.3f Tlane-op/s  clock %.2f GHz  %.2f cyc/wave-instr/SIMD" % (
            name, blocks, r.value / 1e12, c.value, cyc))

clib.spv_microbench_memory.restype = ct.c_int
clib.spv_microbench_memory.argtypes = [ct.c_int, ct.c_size_t, ct.POINTER(ct.c_double)]
for mode, label, sizes in ((0, "stream copy (read+write)", (1 << 30, 4 << 30)),
                           (1, "random 128-B row gather, 8 lanes/row", (32 << 20, 128 << 20, 512 << 20, 4 << 30))):
    for nbytes in sizes:
        r = ct.c_double(0)
        check(clib.spv_microbench_memory(mode, nbytes, ct.byref(r)))
        print("%-40s table %6d MiB  %7.2f TB/s" % (label, nbytes >> 20, r.value / 1e12))
