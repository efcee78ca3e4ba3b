"""The streamed inter-pass table (panda_ntt_set_streamed_tables) against the two-table output product, interleaved: blocks of transforms
with the option off / on in turn (the chip's clock drifts over a run, so only alternating blocks compare).
usage: ntt_streamed_bench.py <log_n[,log_n..]> [rounds] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402
import numpy as np  # noqa: E402


def root(log_n):
    r = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    w = pow(pow(7, (r - 1) >> 28, r), 1 << (28 - log_n), r)
    return np.frombuffer((w * (1 << 256) % r).to_bytes(32, "little"), dtype=np.uint32).copy()


def main():
    ks = [int(x) for x in sys.argv[1].split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    lib = ffi.load()
    gm = pgm.PandaGpuManager(0)
    for k in ks:
        n = 1 << k
        om = root(k)
        da, dbb = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
        ffi.check(lib.panda_gen_scalars(0, 3, 0, n, da.ptr, NULL_STREAM), "gen")
        flag = C.c_uint(0)
        cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, da.ptr, dbb.ptr, C.c_void_p(om.ctypes.data), k, C.pointer(flag))
        ms = C.c_float(0)
        res = {0: [], 1: []}
        for r in range(rounds):
            for on in ((0, 1) if r % 2 == 0 else (1, 0)):
                lib.panda_ntt_set_streamed_tables(on)
                for _ in range(3):  # the first call after a switch rebuilds the tables
                    ffi.check(lib.panda_ntt_execute_bn254_v1(cfg), "ntt")
                ds = []
                for _ in range(reps):
                    ffi.check(lib.panda_ntt_execute_bn254_v1(cfg), "ntt")
                    ffi.check(lib.panda_ntt_last_device_ms(C.byref(ms)), "ms")
                    ds.append(ms.value)
                ds.sort()
                res[on].append(ds[len(ds) // 2])
        lib.panda_ntt_set_streamed_tables(0xFFFFFFFF)
        for on in (0, 1):
            v = sorted(res[on])
            print(f"NTT bn254 2^{k} streamed table {'on ' if on else 'off'}: median of round medians {v[len(v)//2]:.4f} ms   rounds: " + " ".join(f"{x:.4f}" for x in res[on]), flush=True)
        da.free()
        dbb.free()
    gm.deinit()


if __name__ == "__main__":
    main()
