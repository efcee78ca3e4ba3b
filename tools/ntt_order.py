"""Forward / inverse 2^24 transforms timed in alternating blocks (development aid): is the forward figure's 5 % over the inverse an order effect?"""
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

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lib = ffi.load()
gm = pgm.PandaGpuManager(0)
n = 1 << k
om = po.root_of_unity(po.F_BN254_FR, k)
da, db = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
ffi.check(lib.panda_gen_scalars(0, 3, 0, n, da.ptr, NULL_STREAM), "gen")
flag = C.c_uint(0)
cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, da.ptr, db.ptr, C.c_void_p(om.ctypes.data), k, C.pointer(flag))
ms = C.c_float(0)
for block in range(6):
    name, fn = (("forward", lib.panda_ntt_execute_bn254_v1), ("inverse", lib.panda_ntt_execute_bn254_inverse))[block & 1]
    ds = []
    for r in range(14):
        ffi.check(fn(cfg), "ntt")
        ffi.check(lib.panda_ntt_last_device_ms(C.byref(ms)), "ms")
        if r >= 3:
            ds.append(ms.value)
    ds.sort()
    print(f"block {block} {name}: median {ds[len(ds)//2]:.3f} ms  min {ds[0]:.3f}  max {ds[-1]:.3f}", flush=True)
    if block == 3:
        time.sleep(2.0)
        print("(slept 2 s)")
