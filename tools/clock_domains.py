"""Which hardware unit owns the counter s_memtime reads?  One marker launch (panda_clock_stamp: a wave per CU stores s_memtime and
s_memrealtime into the slot XCC_ID * 256 + HW_ID[15:8]); per XCD and shader engine the spread of s_memtime over the CUs, after taking
out the few ticks of s_memrealtime between the waves.  (Round 6 first compared stamps per XCD and got deltas off by 10^8 cycles.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from gpu_util import DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402

lib = ffi.load()
s = ffi.PandaStream()
ffi.check(lib.panda_stream_create(C.byref(s), False), "stream")
buf = DeviceBuffer(ffi.CLOCK_STAMP_BYTES)
ffi.check(lib.panda_memset(buf.ptr, 0, ffi.CLOCK_STAMP_BYTES), "memset")
ffi.check(lib.panda_clock_stamp(s, buf.ptr), "stamp")
ffi.check(lib.panda_stream_sync(s), "sync")
b = buf.to_host(np.uint64).reshape(8, 256, 2)
print("slots stamped per XCD:", [(b[x, :, 1] != 0).sum() for x in range(8)])
for x in range(8):
    ks = [k for k in range(256) if b[x, k, 1]]
    t0 = min(int(b[x, k, 1]) for k in ks)
    # HW_ID[15:8]: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    by_se = {}
    for k in ks:
        adj = int(b[x, k, 0]) - (int(b[x, k, 1]) - t0) * 21  # ~2.1 GHz / 100 MHz: take out the time between the waves
        by_se.setdefault(k >> 5, []).append((k & 15, (k >> 4) & 1, adj))
    base = min(v[2] for vs in by_se.values() for v in vs)
    print(f"XCD {x}: " + "   ".join(f"SE {se}: {len(vs)} CUs, s_memtime - min = {min(v[2] for v in vs) - base} .. {max(v[2] for v in vs) - base}" for se, vs in sorted(by_se.items())))
buf.free()
