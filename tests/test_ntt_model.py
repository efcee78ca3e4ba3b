"""The index algebra of the radix-512 pass (csrc/ntt_radix9.h) on the CPU: tools/model_pass9.py runs the kernel's thread -> element maps, LDS
swizzles and output addressing over a small prime field against the pass formulas, and the pass formulas against a whole transform."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import model_pass9 as mp  # noqa: E402


def test_pass_formulas_compose_to_the_transform():
    for log_n, radices in ((6, (2, 2, 2)), (8, (3, 3, 2)), (9, (3, 3, 3)), (10, (4, 3, 3)), (9, (4, 5))):
        mp.check_plan(log_n, radices)


def test_pass9_thread_maps_and_bank_conflicts():
    # (log_n, lgp, d2, last, tiles): a first pass (one table), a last pass, a middle pass shape, a first pass in front of a radix-256 pass
    for log_n, lgp, d2, last, tiles in ((18, 0, 9, False, (0, 127)), (18, 9, 0, True, (77,)), (20, 9, 2, False, (300,)), (19, 0, 8, False, (200,)), (17, 8, 0, True, (63,)), (20, 8, 3, False, (511,))):  # the last two: behind a radix-256 pass (bit-reversed input plans)
        assert mp.check_pass9(log_n, lgp, d2, last, tiles) == 1  # every LDS access of both exchanges is conflict-free


def test_pass_plan_from_the_library():
    import ctypes as C

    from panda_amd import gpu_ffi as ffi

    lib = ffi.load()
    want = {17: [9, 8], 18: [9, 9], 25: [9, 8, 8], 26: [9, 9, 8], 27: [9, 9, 9], 24: [8, 8, 8], 20: [8, 8, 4], 28: [8, 8, 8, 4], 0: []}
    for log_n, radices in want.items():
        passes, bits = C.c_uint(0), (C.c_uint * 4)()
        assert lib.panda_ntt_pass_plan(log_n, C.byref(passes), bits) == 0
        assert passes.value == len(radices) and [b for b in bits if b] == radices
