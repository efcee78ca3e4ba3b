"""NTT timing only (development aid): python tools/ntt_bench.py <log_n[,log_n..]> [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402
import oracle as po  # noqa: E402


def main():
    ks = [int(x) for x in sys.argv[1].split(",")]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = ffi.load()
    gm = pgm.PandaGpuManager(0)
    for k in ks:
        n = 1 << k
        om = po.root_of_unity(po.F_BN254_FR, k)
        da, dbb = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
        ffi.check(lib.panda_gen_scalars(0, 3, 0, n, da.ptr, NULL_STREAM), "gen")
        flag = C.c_uint(0)
        cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, da.ptr, dbb.ptr, C.c_void_p(om.ctypes.data), k, C.pointer(flag))
        ms = C.c_float(0)
        for name, fn in (("natural", lib.panda_ntt_execute_bn254_v1), ("bitrev_out", lib.panda_ntt_execute_bn254_bitrev_out)):
            ts, ds = [], []
            for r in range(reps + 1):
                t = time.time()
                ffi.check(fn(cfg), "ntt")
                w = time.time() - t
                ffi.check(lib.panda_ntt_last_device_ms(C.byref(ms)), "ms")
                if r:
                    ts.append(w)
                    ds.append(ms.value)
            ts.sort()
            ds.sort()
            best, med, dmed = ts[0], ts[len(ts) // 2], ds[len(ds) // 2]
            print(f"NTT bn254 2^{k} {name}: wall best {best*1e3:8.3f} ms median {med*1e3:8.3f} ms | device median {dmed:8.3f} ms (best {ds[0]:.3f})  "
                  f"{n/dmed/1e6:8.3f} Gelem/s  {n*64/dmed/1e6:8.1f} GB/s algorithmic", flush=True)
        da.free()
        dbb.free()
    gm.deinit()


if __name__ == "__main__":
    main()
